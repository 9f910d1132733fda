"""tests/golden/openvm_airs.npz: the reference's 13 RV32IM instruction AIRs (the original chips an autoprecompile is built from
and the bulk of a segment's "system" AIRs) as prover tables — constraint programs and bus interactions in the post-fix bytecode
pw_prover_create_logup takes. Source: /root/reference/openvm-riscv/tests/openvm_constraints.txt (a snapshot the reference's own
test `machine_extraction` compares against), parsed by powdr_amd/air_text.py. Run where /root/reference is mounted:
    python tests/golden/make_openvm_airs.py"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from powdr_amd import air_text  # noqa: E402

SRC = Path("/root/reference/openvm-riscv/tests/openvm_constraints.txt")


def short_name(full: str) -> str:
    # "VmAirWrapper<Rv32BaseAluAdapterAir, BaseAluCoreAir<4, 8>" -> "BaseAlu"
    core = full.split(",")[1].strip()
    core = core.split("<")[0].rstrip(">")
    return core.replace("CoreAir", "").replace("Rv32", "")


def main():
    airs = air_text.parse_airs(SRC.read_text())
    out = dict(names=np.array([short_name(a.name) for a in airs]), full_names=np.array([a.name for a in airs]),
               widths=np.array([a.width for a in airs], np.uint32))
    for k, a in enumerate(airs):
        bc, spans, (inter, ispans, ibc) = a.tables()
        out[f"a{k}_columns"] = np.array(a.columns)
        out[f"a{k}_bc"], out[f"a{k}_spans"], out[f"a{k}_inter"], out[f"a{k}_ispans"], out[f"a{k}_ibc"] = bc, spans, inter, ispans, ibc
    np.savez_compressed(Path(__file__).parent / "openvm_airs.npz", **out)
    print({n: int(w) for n, w in zip(out["names"], out["widths"])})
    print("constraints", sum(len(a.constraints) for a in airs), "interactions", sum(len(a.interactions) for a in airs))


if __name__ == "__main__":
    main()
