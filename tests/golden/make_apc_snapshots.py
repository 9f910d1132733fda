"""tests/golden/apc_snapshots.json.gz: the reference's golden APC machines — openvm-riscv/tests/apc_snapshots/**/*.txt, what its
apc_builder_{single_instructions,pseudo_instructions,complex,superblocks} tests compare the optimiser's output against
(openvm-riscv/tests/common/mod.rs:38-56, openvm/src/test_utils.rs:63-139) — parsed by powdr_amd/air_text.py::parse_apc_snapshot:
per snapshot the block's instructions (pc + wire format), the optimised machine's columns, constraint texts and bus interactions.
Run where /root/reference is mounted:    python tests/golden/make_apc_snapshots.py"""
import gzip
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from powdr_amd import air_text  # noqa: E402

SRC = Path("/root/reference/openvm-riscv/tests/apc_snapshots")


def main():
    out = {}
    for f in sorted(SRC.glob("*/*.txt")):
        instrs, air = air_text.parse_apc_snapshot(f.read_text())
        air.tables()  # every expression parses
        out[f"{f.parent.name}/{f.stem}"] = dict(instructions=[[pc] + ins for pc, ins in instrs], columns=air.columns, constraints=air.constraints,
                                               interactions=[[bus, mult, args] for bus, mult, args in air.interactions])
    with gzip.GzipFile(Path(__file__).parent / "apc_snapshots.json.gz", "wb", mtime=0) as g:
        g.write(json.dumps(out, sort_keys=True).encode())
    print(len(out), "snapshots,", sum(len(v["columns"]) for v in out.values()), "columns,", sum(len(v["constraints"]) for v in out.values()), "constraints,",
          sum(len(v["interactions"]) for v in out.values()), "interactions")


if __name__ == "__main__":
    main()
