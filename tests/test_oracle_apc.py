"""CPU tests of the APC trace-generation oracle (no GPU).

Pins, in order of strength:
  * the reference's own structural test vectors for this path's data model
    (autoprecompiles/tests/optimizer.rs:66-84 `load_machine_json`,
    number/src/baby_bear.rs:46-55, expression/src/lib.rs:238-246) — run against
    the fixtures in /root/reference when it is mounted, and against the committed
    derived golden summary otherwise;
  * self-consistency: C bytecode evaluator == direct AST evaluation;
    GPU-convention restatement (A) == transposed CPU-convention restatement (B).
"""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import apc_model as om
from powdr_amd import synth

GOLDEN = Path(__file__).parent / "golden"


def test_modulus_pins():
    # number/src/baby_bear.rs:46-55: is_in_lower_half boundary (p-1)/2 = 0x3c000000
    assert om.P == 0x78000001 and (om.P - 1) // 2 == 0x3C000000
    assert om.P == 2**31 - 2**27 + 1
    x = np.array([0, 1, 2, om.P - 1, 12345678], dtype=np.uint32)
    assert (om.from_monty(om.to_monty(x)) == x).all()
    assert om.to_monty(np.array([1], dtype=np.uint32))[0] == 0x0FFFFFFE  # R mod p


def test_expression_serde_format():
    # expression/src/lib.rs:238-246: 5*x - 3 serialises as [[5,"*","x"],"-",3]
    e = om.parse_expr(json.loads('[[5,"*","x@7"],"-",3]'))
    assert e == ("bin", "-", ("bin", "*", ("num", 5), ("ref", "x", 7)), ("num", 3))
    assert om.eval_ast(e, lambda pid: 11) == 52
    assert om.parse_expr(json.loads('["-", "a_b@12"]')) == ("neg", ("ref", "a_b", 12))
    bc = []
    om.emit_expr(bc, e, {7: 2}, 8)
    # post-fix: 5 x * 3 -   (cuda/mod.rs:49-81), x at column 2 of a height-8 trace
    assert bc == [1, 5, 0, 16, 4, 1, 3, 3]


def _random_expr(rng, ids, depth):
    if depth == 0 or rng.random() < 0.25:
        return ("num", int(rng.integers(0, om.P))) if rng.random() < 0.4 else ("ref", "c", int(rng.choice(ids)))
    r = rng.random()
    if r < 0.15:
        return ("neg", _random_expr(rng, ids, depth - 1))
    return ("bin", "+-*"[int(rng.integers(3))], _random_expr(rng, ids, depth - 1), _random_expr(rng, ids, depth - 1))


def test_c_evaluator_matches_ast():
    rng = np.random.default_rng(1)
    W, H = 12, 8
    ids = list(range(100, 100 + W))
    id_to_index = {p: i for i, p in enumerate(ids)}
    trace = rng.integers(0, om.P, size=W * H, dtype=np.uint32)  # column-major
    for _ in range(200):
        e = _random_expr(rng, ids, 4)
        bc = []
        om.emit_expr(bc, e, id_to_index, H)
        r = int(rng.integers(H))
        try:
            got = om.c_eval_expr(np.array(bc, dtype=np.uint32), trace, r)
        except ValueError:
            continue  # deeper than the 16-entry stack (expr_eval.cuh:22)
        want = om.eval_ast(e, lambda pid: int(trace[id_to_index[pid] * H + r]))
        assert got == want


def test_inv_or_zero():
    bc = np.array([om.OP_PUSH_CONST, 0, om.OP_INV_OR_ZERO], dtype=np.uint32)
    assert om.c_eval_expr(bc, np.zeros(1, np.uint32), 0) == 0
    for v in (1, 2, 31, om.P - 1, 1234567):
        bc = np.array([om.OP_PUSH_CONST, v, om.OP_INV_OR_ZERO, om.OP_PUSH_CONST, v, om.OP_MUL], dtype=np.uint32)
        assert om.c_eval_expr(bc, np.zeros(1, np.uint32), 0) == 1


def run_oracle_gpu_convention(s, num_calls, seed=0):
    """Oracle (A): column-major trace + histograms for a synthetic APC."""
    apc = om.load_apc(s.doc)
    idx = apc.poly_id_to_index()
    W = len(idx)
    H = synth.next_pow2_or_zero(num_calls)
    bufs, dims = synth.fill_dummy_traces_numpy(s, num_calls, seed)
    air_of = _air_of(s)
    gt = om.build_gpu_tables(apc, idx, air_of)
    name_to = {n: i for i, (n, _, _, _) in enumerate(dims)}
    order = [name_to[n] for n in gt.air_names]
    out = om.c_apc_tracegen(H, W, [bufs[i] for i in order], [dims[i][2] for i in order], gt.row_block_size, gt.subs, num_calls)
    cb, offs, lens, dbc = om.compile_derived(apc, idx, H)
    om.c_apc_apply_derived(out, H, num_calls, cb, offs, lens, dbc)
    inter, spans, bbc = om.compile_bus(apc, idx, H)
    hist = dict(var=np.zeros(1 << 18, np.uint32), tuple=np.zeros(256 * 2048, np.uint32), bitwise=np.zeros(2 * 65536, np.uint32))
    om.c_apc_apply_bus(out, num_calls, bbc, inter, spans, 3, hist["var"], 7, hist["tuple"], 256, 2048, 6, hist["bitwise"])
    return apc, idx, out.reshape(W, H) if H else out.reshape(W, 0), hist, (bufs, dims, gt, order)


def _air_of(s):
    """The synthetic generator draws opcodes from the class of each AIR, so the oracle's
    opcode -> AIR table (om.opcode_air) reproduces s.instr_air; None = use that table."""
    for ins, n in zip(s.doc["block"]["blocks"][0]["instructions"], s.instr_air):
        assert n == "" or n == om.opcode_air(ins[0])
    return None


@pytest.mark.parametrize("shape,num_calls", [("T0", 5), ("T0", 64), ("T1", 37), ("T1", 256)])
def test_gpu_convention_equals_cpu_convention(shape, num_calls):
    s = synth.generate(shape, seed=3)
    apc, idx, out_cm, hist, (bufs, dims, gt, order) = run_oracle_gpu_convention(s, num_calls, seed=5)
    W = len(idx)
    # CPU convention: row-major dummy traces, rows of each AIR in call-major order
    ct = om.build_cpu_tables(apc, idx)
    name_to = {n: i for i, (n, _, _, _) in enumerate(dims)}
    dummy_rm, dummy_w = [], []
    for n in ct.air_names:
        i = name_to[n]
        _, w, h, b = dims[i]
        dummy_rm.append(np.ascontiguousarray(bufs[i].reshape(w, h).T))  # [h, w] row-major
        dummy_w.append(w)
    per = dict(var_bus=3, var_hist=np.zeros(1 << 18, np.uint32), tuple_bus=7, tuple_hist=np.zeros(256 * 2048, np.uint32),
               sz0=256, sz1=2048, bitwise_bus=6, bitwise_hist=np.zeros(2 * 65536, np.uint32))
    vals = om.c_generate_witness(apc, ct, idx, dummy_rm, dummy_w, num_calls, per)
    assert vals.shape == (synth.next_pow2_or_zero(num_calls), W)
    assert (vals.T == out_cm).all()
    assert (per["var_hist"] == hist["var"]).all()
    assert (per["tuple_hist"] == hist["tuple"]).all()
    assert (per["bitwise_hist"] == hist["bitwise"]).all()
    # padding rows are zero, is_valid is one on valid rows
    assert (out_cm[:, num_calls:] == 0).all()
    valid_col = idx[[p for p, k in s.kinds.items() if k[0] == "valid"][0]]
    assert (out_cm[valid_col, :num_calls] == 1).all()
    assert hist["var"].sum() > 0 and hist["bitwise"].sum() > 0 and hist["tuple"].sum() > 0


def test_ast_eval_matches_oracle_rows():
    """Direct AST evaluation of bus arguments (expression.rs:115-147) on a few rows."""
    s = synth.generate("T0", seed=1)
    apc, idx, out_cm, hist, _ = run_oracle_gpu_convention(s, 9, seed=2)
    H = out_cm.shape[1]
    inter, spans, bc = om.compile_bus(apc, idx, H)
    flat = np.ascontiguousarray(out_cm).reshape(-1)
    for r in (0, 3, 8):
        k = 0
        for b in apc.bus_interactions:
            for e in [b.mult] + b.args:
                off, ln = spans[k]
                k += 1
                assert om.c_eval_expr(bc[off : off + ln], flat, r) == om.eval_ast(e, lambda pid: int(out_cm[idx[pid], r]))


def test_synth_shapes():
    for name in ("C1", "C2"):
        s = synth.generate(name, seed=0)
        apc = om.load_apc(s.doc)
        sh = synth.SHAPES[name]
        assert len(apc.main_columns()) == sh.width
        assert len(apc.bus_interactions) == sh.n_interactions
        assert len(apc.constraints) == sh.n_constraints
        assert sum(len(x) for x in apc.subs) == sh.width - 1 - sh.n_quotient


# ---- reference fixtures ------------------------------------------------------------------

FIXTURE_PINS = {
    # autoprecompiles/tests/optimizer.rs:66-84 (load_machine_json)
    "keccak_apc_pre_opt": dict(main_columns=27521, bus_interactions=13262, constraints=28627),
}


@pytest.mark.parametrize("name", list(FIXTURE_PINS))
def test_reference_fixture_pins(name, reference_dir):
    apc = om.load_apc_file(reference_dir / "autoprecompiles/tests" / f"{name}.json.gz")
    pins = FIXTURE_PINS[name]
    assert not apc.derived_columns
    assert len(apc.main_columns()) == pins["main_columns"]
    assert len(apc.bus_interactions) == pins["bus_interactions"]
    assert len(apc.constraints) == pins["constraints"]
    # sum over AIRs of width x instructions = pre-opt column count
    w = om.air_widths(apc)
    gt = om.build_gpu_tables(apc, apc.poly_id_to_index())
    assert sum(w[n] * b for n, b in zip(gt.air_names, gt.row_block_size)) == pins["main_columns"]


def test_committed_golden_summary_matches_pins():
    """The golden summary generated from the reference fixtures (tests/golden/make_golden.py)
    carries the same pins, so they are checked on machines without /root/reference too."""
    summ = json.loads((GOLDEN / "apc_fixtures_summary.json").read_text())
    k = summ["keccak_apc_pre_opt"]
    assert (k["main_columns"], k["bus_interactions"], k["constraints"]) == (27521, 13262, 28627)
    assert k["airs"] == {"BaseAlu": [36, 318], "Shift": [53, 116], "LoadStore": [41, 241], "BranchEqual": [26, 1], "JalLui": [18, 1]}


def test_column_structured_substitutions_shape_and_oracle_gather():
    """synth.column_structured_substitutions: distinct source cells, a permutation of the APC columns, few source columns with
    most of their rows — and the oracle's gather (apc_tracegen.cu:35-66 restated) places exactly those cells."""
    dims = [(36, 318), (53, 116), (41, 241), (26, 1), (18, 1)]
    n_sub, calls = 2017, 3
    subs = synth.column_structured_substitutions(dims, n_sub, seed=5)
    assert subs.shape == (n_sub, 4)
    assert len(set(map(tuple, subs[:, :3].tolist()))) == n_sub
    assert sorted(subs[:, 3].tolist()) == list(range(n_sub))
    for a, col, row, _ in subs.tolist():
        assert 0 <= col < dims[a][0] and 0 <= row < dims[a][1]
    used = {}
    for a, col, row, _ in subs.tolist():
        used.setdefault((a, col), set()).add(row)
    assert len(used) <= 0.15 * sum(w for w, _ in dims)
    big = [len(rows) / dims[a][1] for (a, _), rows in used.items() if dims[a][1] > 1]
    assert min(big) > 0.4 and sum(big) / len(big) > 0.55
    rng = np.random.default_rng(1)
    H = 4
    bufs, hs = [], []
    for w, b in dims:
        h = max(synth.next_pow2_or_zero(b * calls), 4)
        bufs.append(rng.integers(0, om.P, size=w * h, dtype=np.uint32)); hs.append(h)
    got = om.c_apc_tracegen(H, n_sub, bufs, hs, [b for _, b in dims], subs, calls).reshape(n_sub, H)
    for a, col, row, apc_col in subs[::37].tolist():
        for r in range(H):
            want = bufs[a][col * hs[a] + row + r * dims[a][1]] if r < calls else 0
            assert got[apc_col, r] == want
