"""Generate the committed golden fixtures from the reference's own test fixtures.

Run in the build container (where /root/reference is mounted):
    python tests/golden/make_golden.py
Outputs (committed; /root/reference does not exist on the GPU box):
    tests/golden/apc_fixtures_summary.json   structural pins + hashes per fixture
    tests/golden/<fixture>.apc.npz           the APC compiled to this repo's flat tables
                                             (our own format — derived data, not a copy)
Source fixtures: /root/reference/autoprecompiles/tests/*.json.gz, the inputs of the
reference's optimizer tests (autoprecompiles/tests/optimizer.rs:66-281).
"""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import apc_model as om  # noqa: E402

REF = Path("/root/reference/autoprecompiles/tests")
OUT = Path(__file__).resolve().parent
FIXTURES = ["keccak_apc_pre_opt", "ecrecover_apc_pre_opt", "single_div_nondet", "wasm_register_reuse", "apc_reth_op_bug"]
NPZ = ["keccak_apc_pre_opt", "ecrecover_apc_pre_opt", "single_div_nondet"]


def max_depth(bc):
    d = m = 0
    i = 0
    while i < len(bc):
        op = bc[i]
        if op <= 1:
            d += 1; i += 2
        elif op <= 4:
            d -= 1; i += 1
        else:
            i += 1
        m = max(m, d)
    return m


def apc_positions(bc):
    pos, i = [], 0
    while i < len(bc):
        if bc[i] == om.OP_PUSH_APC:
            pos.append(i + 1)
        i += 2 if bc[i] <= 1 else 1
    return np.array(pos, dtype=np.uint32)


def column_bounds(apc, idx):
    """Upper bounds (exclusive) per column so that lookups land in range: 2^bits for direct
    var-range operands, 256 otherwise (limbs are bytes)."""
    b = np.full(len(idx), 256, dtype=np.uint32)
    for bi in apc.bus_interactions:
        if bi.id == 3 and bi.args[0][0] == "ref" and bi.args[1][0] == "num":
            c = idx[bi.args[0][2]]
            b[c] = min(b[c], 1 << min(bi.args[1][1], 17)) if b[c] != 256 else (1 << min(bi.args[1][1], 17))
    for bi in apc.bus_interactions:
        if bi.id == 6:
            for a in bi.args[:2]:
                if a[0] == "ref":
                    b[idx[a[2]]] = min(b[idx[a[2]]], 256)
    return b


def main():
    summary = {}
    for name in FIXTURES:
        apc = om.load_apc_file(REF / f"{name}.json.gz")
        idx = apc.poly_id_to_index()
        inter, spans, bc = om.compile_bus(apc, idx, 1)
        gt = om.build_gpu_tables(apc, idx)
        widths = om.air_widths(apc)
        cons_bc, cons_spans = [], []
        for c in apc.constraints:
            off = len(cons_bc)
            om.emit_expr(cons_bc, c, idx, 1)
            cons_spans.append((off, len(cons_bc) - off))
        depth = max([max_depth(bc[o : o + l].tolist()) for o, l in spans] + [0])
        summary[name] = dict(
            main_columns=len(idx), bus_interactions=len(apc.bus_interactions), constraints=len(apc.constraints),
            derived_columns=len(apc.derived_columns), instructions=len(apc.instructions),
            n_subs=int(sum(len(s) for s in apc.subs)),
            airs={n: [widths[n], b] for n, b in zip(gt.air_names, gt.row_block_size)},
            bus_ids=sorted({int(b.id) for b in apc.bus_interactions}),
            bus_bytecode_words=int(len(bc)), bus_bytecode_sha256=hashlib.sha256(bc.tobytes()).hexdigest(),
            max_stack_depth=int(depth),
            subs_sha256=hashlib.sha256(gt.subs.tobytes()).hexdigest(),
        )
        print(name, summary[name])
        if name in NPZ:
            np.savez_compressed(
                OUT / f"{name}.apc.npz",
                poly_ids=np.array(sorted(idx), dtype=np.uint64),
                air_names=np.array(gt.air_names), air_widths=np.array([widths[n] for n in gt.air_names], dtype=np.int32),
                row_block_size=np.array(gt.row_block_size, dtype=np.int32), subs=gt.subs,
                bus_inter=inter, bus_spans=spans, bus_bc=bc, bus_apc_pos=apc_positions(bc.tolist()),
                cons_spans=np.array(cons_spans, dtype=np.uint32).reshape(-1, 2), cons_bc=np.array(cons_bc, dtype=np.uint32),
                cons_apc_pos=apc_positions(cons_bc), col_bound=column_bounds(apc, idx),
                # the block itself: [opcode, a, b, c, d, e, f, g] per instruction (operands as field elements) and its first pc —
                # the input of the original-chip expanders (powdr_amd/original_chips.py)
                instructions=np.array([[int(x) % om.P for x in ins] for ins in apc.instructions], dtype=np.uint32),
                start_pc=np.array([apc.start_pc], dtype=np.uint64),
            )
    (OUT / "apc_fixtures_summary.json").write_text(json.dumps(summary, indent=1, sort_keys=True) + "\n")


if __name__ == "__main__":
    main()
