"""Parity of the HIP trace-generation kernels against the CPU oracle, through the C ABI.

Bit-exact: BabyBear cells compared in canonical form, histograms as u32 counters.
"""
from pathlib import Path

import numpy as np
import pytest

from oracle import apc_model as om
from powdr_amd import synth

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def gpu():
    import torch
    from powdr_amd import abi, tracegen  # raises if libpowdr_gpu.so is missing

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU (run with -m gpu on the GPU box)")
    return torch, abi, tracegen


def to_dev(torch, a_canonical):
    return torch.from_numpy(om.to_monty(a_canonical).view(np.int32)).cuda()


def from_dev(t):
    return om.from_monty(t.cpu().numpy().view(np.uint32))


def hist_np(t):
    return t.cpu().numpy().view(np.uint32)


def run_gpu(gpu, W, H, num_calls, bufs, dims, air_names, rbs, subs, derived, bus, prefill=None):
    torch, abi, tg = gpu
    out = tg.DeviceMatrix.zeros(H, W)
    if prefill is not None:
        out.buf.copy_(torch.from_numpy(prefill.view(np.int32)))
    name_to = {n: i for i, (n, _, _, _) in enumerate(dims)}
    airs = []
    keep = []
    for n, b in zip(air_names, rbs):
        _, w, h, _ = dims[name_to[n]]
        t = to_dev(torch, bufs[name_to[n]])
        keep.append(t)
        airs.append((t, w, h, b))
    keep.append(tg.apc_tracegen(out, airs, subs, num_calls))
    if derived is not None:
        keep.append(tg.apc_apply_derived_expr(out, num_calls, *derived))
    per = tg.Periphery.fresh()
    if bus is not None:
        inter, spans, bc = bus
        keep.append(tg.apc_apply_bus(out, num_calls, bc, inter, spans, per))
    torch.cuda.synchronize()
    return out, per


@pytest.mark.parametrize("shape,num_calls,seed", [("T0", 1, 0), ("T0", 5, 1), ("T0", 64, 2), ("T1", 37, 3),
                                                   ("T1", 1000, 4), ("T1", 4096, 5), ("C1", 300, 6)])
def test_synthetic_apc_parity(gpu, shape, num_calls, seed):
    from tests.test_oracle_apc import run_oracle_gpu_convention

    s = synth.generate(shape, seed=seed)
    apc, idx, want, hist, (bufs, dims, gt, order) = run_oracle_gpu_convention(s, num_calls, seed=seed)
    W, H = want.shape
    derived = om.compile_derived(apc, idx, H)
    inter, spans, bc = om.compile_bus(apc, idx, H)
    out, per = run_gpu(gpu, W, H, num_calls, bufs, dims, gt.air_names, gt.row_block_size, gt.subs, derived, (inter, spans, bc))
    got = from_dev(out.buf).reshape(W, H)
    assert (got == want).all()
    assert (hist_np(per.var_hist) == hist["var"]).all()
    assert (hist_np(per.tuple_hist) == hist["tuple"]).all()
    assert (hist_np(per.bitwise_hist) == hist["bitwise"]).all()
    assert hist["var"].sum() > 0


def _scaled(bc, pos, H):
    bc = bc.copy()
    bc[pos] = (bc[pos].astype(np.uint64) * H).astype(np.uint32)
    return bc


@pytest.mark.parametrize("name,num_calls", [("single_div_nondet", 77), ("keccak_apc_pre_opt", 96)])
def test_reference_fixture_apc_parity(gpu, name, num_calls):
    """The reference's own APC fixtures (autoprecompiles/tests/*.json.gz, compiled to flat
    tables by tests/golden/make_golden.py): full pre-optimisation keccak block = 27 521
    substitutions from 5 AIRs and 13 262 bus interactions."""
    z = np.load(GOLDEN / f"{name}.apc.npz")
    W = len(z["poly_ids"])
    H = synth.next_pow2_or_zero(num_calls)
    rng = np.random.default_rng(42)
    bound = z["col_bound"]
    rbs, widths, subs = z["row_block_size"], z["air_widths"], z["subs"]
    bufs, dims = [], []
    for k, (w, b) in enumerate(zip(widths, rbs)):
        h = max(synth.next_pow2_or_zero(int(b) * num_calls), 4)
        bufs.append(rng.integers(0, 256, size=int(w) * h, dtype=np.uint32))
        dims.append((str(z["air_names"][k]), int(w), h, int(b)))
    # cells feeding tighter-bounded columns
    for a, col, row, apc_col in subs:
        if bound[apc_col] < 256:
            _, w, h, b = dims[a]
            v = bufs[a][col * h + row : col * h + row + b * num_calls : b]
            v[:] = rng.integers(0, bound[apc_col], size=len(v), dtype=np.uint32)
    want = om.c_apc_tracegen(H, W, bufs, [d[2] for d in dims], rbs, subs, num_calls)
    bc = _scaled(z["bus_bc"], z["bus_apc_pos"], H)
    hist = dict(var=np.zeros(1 << 18, np.uint32), tuple=np.zeros(256 * 2048, np.uint32), bitwise=np.zeros(2 * 65536, np.uint32))
    om.c_apc_apply_bus(want, num_calls, bc, z["bus_inter"], z["bus_spans"], 3, hist["var"], 7, hist["tuple"], 256, 2048, 6, hist["bitwise"])
    names = [d[0] for d in dims]
    out, per = run_gpu(gpu, W, H, num_calls, bufs, dims, names, rbs.tolist(), subs, None, (z["bus_inter"], z["bus_spans"], bc))
    assert (from_dev(out.buf) == want).all()
    assert (hist_np(per.var_hist) == hist["var"]).all()
    assert (hist_np(per.tuple_hist) == hist["tuple"]).all()
    assert (hist_np(per.bitwise_hist) == hist["bitwise"]).all()
    assert hist["var"].sum() > 0 and hist["bitwise"].sum() > 0


def test_gather_edge_cases(gpu):
    torch, abi, tg = gpu
    rng = np.random.default_rng(7)
    # (a) H = 0 and n_subs = 0 succeed and touch nothing
    out = tg.DeviceMatrix.zeros(0, 4)
    src = to_dev(torch, rng.integers(0, om.P, size=64, dtype=np.uint32))
    tg.apc_tracegen(out, [(src, 4, 16, 1)], np.zeros((0, 4), np.int32), 0)
    # (b) non-power-of-two height is rejected (reference: assert, apc_tracegen.cu:134)
    out = tg.DeviceMatrix.zeros(24, 2)
    with pytest.raises(abi.HipError):
        tg.apc_tracegen(out, [(src, 4, 16, 1)], np.array([[0, 0, 0, 0]], np.int32), 3)
    # (c) duplicate destination: the later substitution wins, like the sequential reference loop;
    #     padding rows are zeroed even if the caller did not clear the buffer; untouched columns stay
    H, W, calls = 16, 3, 11
    a0 = rng.integers(0, om.P, size=5 * 64, dtype=np.uint32)  # width 5, height 64, block 4
    subs = np.array([[0, 1, 2, 0], [0, 3, 0, 0], [0, 4, 3, 2]], np.int32)
    prefill = om.to_monty(rng.integers(0, om.P, size=H * W, dtype=np.uint32))
    want = om.from_monty(prefill.copy())
    om.c_apc_tracegen(H, W, [a0], [64], [4], subs, calls, out=want)
    out = tg.DeviceMatrix.zeros(H, W)
    out.buf.copy_(torch.from_numpy(prefill.view(np.int32)))
    tg.apc_tracegen(out, [(to_dev(torch, a0), 5, 64, 4)], subs, calls)
    torch.cuda.synchronize()
    assert (from_dev(out.buf) == want).all()


@pytest.mark.parametrize("b,n_sub_rows,calls", [(1, 1, 1000), (2, 2, 513), (7, 3, 300), (64, 64, 200),
                                                  (383, 50, 130), (600, 40, 70), (1500, 9, 33), (318, 318, 257)])
def test_gather_block_shapes(gpu, b, n_sub_rows, calls):
    """Row-block sizes around every planner decision: tiny blocks (big R), blocks that fit one
    LDS tile, blocks larger than a tile (row chunks), dense and sparse substitutions."""
    torch, abi, tg = gpu
    rng = np.random.default_rng(b * 1000 + calls)
    w = 6
    H = synth.next_pow2_or_zero(calls)
    h = synth.next_pow2_or_zero(b * calls)
    h = max(h, 4)
    src = rng.integers(0, om.P, size=w * h, dtype=np.uint32)
    rows = np.sort(rng.choice(b, size=min(n_sub_rows, b), replace=False))
    recs = []
    apc_col = 0
    for col in (0, 2, 5):
        for r in rows:
            if rng.random() < 0.8:
                recs.append((0, col, int(r), apc_col))
                apc_col += 1
    recs = np.array(recs, np.int32)
    recs = recs[rng.permutation(len(recs))]
    W = apc_col
    want = om.c_apc_tracegen(H, W, [src], [h], [b], recs, calls)
    out = tg.DeviceMatrix.zeros(H, W)
    tg.apc_tracegen(out, [(to_dev(torch, src), w, h, b)], recs, calls)
    torch.cuda.synchronize()
    assert (from_dev(out.buf) == want).all()


def test_derived_and_bus_edge_cases(gpu):
    torch, abi, tg = gpu
    H, W, calls = 8, 4, 6
    rng = np.random.default_rng(3)
    trace = rng.integers(0, om.P, size=H * W, dtype=np.uint32)
    trace[0:H] = [0, 1, 2, 3, 300, 255, 0, 0]        # column 0: x
    trace[H : 2 * H] = [0, 7, 255, 256, 1, 2, 0, 0]   # column 1: y
    out = tg.DeviceMatrix.zeros(H, W)
    out.buf.copy_(to_dev(torch, trace))
    # n_cols == 0 returns immediately (apc_tracegen.cu:114); num_calls <= 0 for the bus too
    tg.apc_apply_derived_expr(out, calls, np.zeros(0, np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(1, np.uint32))
    per = tg.Periphery.fresh()
    inter = np.array([[6, 4, 0], [3, 2, 5], [7, 2, 8], [9, 1, 11]], np.uint32)  # bitwise, var, tuple, unknown bus
    PA, PC = om.OP_PUSH_APC, om.OP_PUSH_CONST
    bc, spans = [], []

    def span(words):
        spans.append((len(bc), len(words)))
        bc.extend(words)

    span([PC, 3])              # mult 3
    span([PA, 0 * H])          # x
    span([PA, 1 * H])          # y
    span([PC, 0])              # x^y (never read)
    span([PC, 1])              # selector xor
    span([PC, 2])              # var: mult 2
    span([PA, 0 * H])          # value = x
    span([PC, 9])              # bits
    span([PC, 1])              # tuple: mult
    span([PA, 1 * H])          # v0 = y
    span([PA, 0 * H])          # v1 = x
    span([PC, 1])              # unknown bus: mult
    span([PA, 3 * H])
    tg.apc_apply_bus(out, 0, bc, inter, spans, per)
    torch.cuda.synchronize()
    assert int(per.var_hist.sum()) == 0
    tg.apc_apply_bus(out, calls, bc, inter, spans, per)
    torch.cuda.synchronize()
    want = dict(var=np.zeros(1 << 18, np.uint32), tuple=np.zeros(256 * 2048, np.uint32), bitwise=np.zeros(2 * 65536, np.uint32))
    om.c_apc_apply_bus(trace, calls, np.array(bc, np.uint32), inter, np.array(spans, np.uint32), 3, want["var"], 7, want["tuple"], 256, 2048, 6, want["bitwise"])
    assert (hist_np(per.var_hist) == want["var"]).all()
    assert (hist_np(per.tuple_hist) == want["tuple"]).all()
    assert (hist_np(per.bitwise_hist) == want["bitwise"]).all()
    # x = 300 is not a byte: dropped; multiplicity 3 applied as +3
    assert want["bitwise"][65536 + 1 * 256 + 7] == 3 and want["bitwise"].sum() == 3 * 4
    # derived: later columns read earlier ones of the same row; rows >= calls are zeroed
    col_base = np.array([2 * H, 3 * H], np.uint64)
    dbc = [PA, 1 * H, om.OP_INV_OR_ZERO, PA, 0 * H, om.OP_MUL,   # col2 = x / y or 0
           PA, 2 * H, PA, 2 * H, om.OP_MUL, om.OP_NEG]            # col3 = -(col2^2)
    offs, lens = np.array([0, 6], np.uint32), np.array([6, 6], np.uint32)
    want_t = trace.copy()
    om.c_apc_apply_derived(want_t, H, calls, col_base, offs, lens, np.array(dbc, np.uint32))
    tg.apc_apply_derived_expr(out, calls, col_base, offs, lens, np.array(dbc, np.uint32))
    torch.cuda.synchronize()
    assert (from_dev(out.buf) == want_t).all()
    assert (want_t[2 * H + calls : 3 * H] == 0).all() and want_t[2 * H] == 0  # y = 0 -> 0


def test_large_gather_property(gpu):
    """C2-shaped gather at 2^15 calls (sources ~5 GB): every output cell equals the source cell
    the substitution names, checked on device with an independent torch gather; padding is zero."""
    torch, abi, tg = gpu
    s = synth.generate("C2", seed=0)
    apc = om.load_apc(s.doc)
    idx = apc.poly_id_to_index()
    gt = om.build_gpu_tables(apc, idx)
    calls = (1 << 15) - 123
    H = 1 << 15
    W = len(idx)
    dims = {n: (w, b) for n, w, b in s.airs}
    g = torch.Generator(device="cuda").manual_seed(1)
    airs = []
    for n, b in zip(gt.air_names, gt.row_block_size):
        w, b2 = dims[n]
        assert b2 == b
        h = max(synth.next_pow2_or_zero(b * calls), 4)
        airs.append((torch.randint(0, om.P, (w * h,), dtype=torch.int32, device="cuda", generator=g), w, h, b))
    out = tg.DeviceMatrix(torch.full((H * W,), 0x55, dtype=torch.int32, device="cuda"), H, W)
    keep = tg.apc_tracegen(out, airs, gt.subs, calls)
    torch.cuda.synchronize()
    r = torch.arange(calls, device="cuda", dtype=torch.int64)
    m = out.buf.view(W, H)
    sel = np.random.default_rng(0).choice(len(gt.subs), size=300, replace=False)
    for a, col, row, apc_col in gt.subs[sel]:
        t, w, h, b = airs[a]
        want = t[col * h + row + r * b]
        assert torch.equal(m[apc_col, :calls], want)
        assert int(m[apc_col, calls:].abs().sum()) == 0
    # columns without a substitution (derived ones) are untouched
    untouched = sorted(set(range(W)) - set(gt.subs[:, 3].tolist()))
    assert untouched and all(int((m[c] != 0x55).sum()) == 0 for c in untouched)


@pytest.mark.parametrize("shape,num_calls,seed", [("T1", 4096, 5), ("T1", 1001, 6), ("C1", 20000, 7)])
def test_binned_histogram_path(gpu, shape, num_calls, seed, monkeypatch):
    """Long traces take the binned path (packed items + LDS partition histograms); force it on
    short ones too and compare with the oracle and with the direct-atomic path."""
    from tests.test_oracle_apc import run_oracle_gpu_convention

    s = synth.generate(shape, seed=seed)
    apc, idx, want, hist, (bufs, dims, gt, order) = run_oracle_gpu_convention(s, num_calls, seed=seed)
    W, H = want.shape
    inter, spans, bc = om.compile_bus(apc, idx, H)
    for mode in ("1", "0"):
        monkeypatch.setenv("POWDR_BUS_BINNED", mode)
        out, per = run_gpu(gpu, W, H, num_calls, bufs, dims, gt.air_names, gt.row_block_size, gt.subs,
                           om.compile_derived(apc, idx, H), (inter, spans, bc))
        assert (hist_np(per.var_hist) == hist["var"]).all()
        assert (hist_np(per.tuple_hist) == hist["tuple"]).all()
        assert (hist_np(per.bitwise_hist) == hist["bitwise"]).all()


def test_binned_path_large_multiplicity_and_hot_bins(gpu, monkeypatch):
    """Multiplicities that do not fit the packed item (>= 4096) fall back to a direct atomic; a
    constant lookup (every row hits the same bin) is the hot-bin case."""
    torch, abi, tg = gpu
    monkeypatch.setenv("POWDR_BUS_BINNED", "1")
    H, W, calls = 1024, 2, 1000
    rng = np.random.default_rng(11)
    trace = np.zeros(H * W, np.uint32)
    trace[:calls] = rng.integers(0, 256, calls)
    trace[H : H + calls] = rng.integers(0, 4, calls)
    PA, PC = om.OP_PUSH_APC, om.OP_PUSH_CONST
    bc, spans = [], []

    def span(words):
        spans.append((len(bc), len(words)))
        bc.extend(words)

    inter = np.array([[3, 2, 0], [3, 2, 3], [6, 4, 6]], np.uint32)
    span([PC, 5000]); span([PA, 0]); span([PC, 8])          # var: mult 5000, value = col0, bits 8
    span([PC, 1]); span([PC, 77]); span([PC, 12])             # var: constant lookup, one hot bin
    span([PA, H]); span([PA, 0]); span([PC, 3]); span([PC, 0]); span([PC, 0])  # bitwise range, mult = col1 (0..3)
    out = tg.DeviceMatrix.zeros(H, W)
    out.buf.copy_(to_dev(torch, trace))
    per = tg.Periphery.fresh()
    tg.apc_apply_bus(out, calls, bc, inter, spans, per)
    torch.cuda.synchronize()
    want = dict(var=np.zeros(1 << 18, np.uint32), tuple=np.zeros(256 * 2048, np.uint32), bitwise=np.zeros(2 * 65536, np.uint32))
    om.c_apc_apply_bus(trace, calls, np.array(bc, np.uint32), inter, np.array(spans, np.uint32), 3, want["var"], 7, want["tuple"], 256, 2048, 6, want["bitwise"])
    assert (hist_np(per.var_hist) == want["var"]).all() and want["var"][(1 << 12) + 76] == calls
    assert (hist_np(per.bitwise_hist) == want["bitwise"]).all()


@pytest.mark.parametrize("fast", ["0", "1"])
def test_binned_path_values_outside_the_expected_partition(gpu, monkeypatch, fast):
    """The histogram pass only visits the partitions a slot's constant operands predict (a 12-bit range check: the
    first 32 768 bins; a bitwise lookup with a constant operation: its half of the table). Values a dishonest trace
    puts elsewhere — 17-bit values under a 12-bit check, here — are still counted exactly (direct atomics)."""
    torch, abi, tg = gpu
    monkeypatch.setenv("POWDR_BUS_BINNED", "1")
    monkeypatch.setenv("POWDR_BUS_FAST", fast)
    H, W, calls = 2048, 3, 2000
    rng = np.random.default_rng(12)
    trace = np.zeros(H * W, np.uint32)
    trace[:calls] = rng.integers(0, 1 << 17, calls)            # col0: up to 17 bits
    trace[H : H + calls] = rng.integers(0, 256, calls)         # col1: bytes
    trace[2 * H : 2 * H + calls] = rng.integers(0, 3, calls)   # col2: selector 0, 1 and the invalid 2
    PA, PC = om.OP_PUSH_APC, om.OP_PUSH_CONST
    bc, spans = [], []

    def span(words):
        spans.append((len(bc), len(words)))
        bc.extend(words)

    inter = np.array([[3, 2, 0], [3, 2, 3], [6, 4, 6], [6, 4, 11]], np.uint32)
    span([PC, 1]); span([PA, 0]); span([PC, 12])               # 12-bit check of a 17-bit column: bins up to 2^12 + 2^17
    span([PC, 2]); span([PA, 0]); span([PC, 17])               # the honest 17-bit check of the same column
    span([PC, 1]); span([PA, H]); span([PA, H]); span([PC, 0]); span([PC, 1])     # xor lookups: second half of the table
    span([PC, 1]); span([PA, H]); span([PA, H]); span([PC, 0]); span([PA, 2 * H])  # selector from a column: no prediction
    out = tg.DeviceMatrix.zeros(H, W)
    out.buf.copy_(to_dev(torch, trace))
    per = tg.Periphery.fresh()
    tg.apc_apply_bus(out, calls, bc, inter, spans, per)
    torch.cuda.synchronize()
    want = dict(var=np.zeros(1 << 18, np.uint32), tuple=np.zeros(256 * 2048, np.uint32), bitwise=np.zeros(2 * 65536, np.uint32))
    om.c_apc_apply_bus(trace, calls, np.array(bc, np.uint32), inter, np.array(spans, np.uint32), 3, want["var"], 7, want["tuple"], 256, 2048, 6, want["bitwise"])
    assert want["var"][32768:].sum() > calls  # plenty of lookups beyond partition 0
    assert (hist_np(per.var_hist) == want["var"]).all()
    assert (hist_np(per.bitwise_hist) == want["bitwise"]).all()


def test_column_operand_extensions_match_reference_encoding(gpu):
    """powdr_apc_apply_derived_expr_cols / powdr_apc_apply_bus_cols (PUSH_APC operand = column index, for
    traces with W*H >= 2^32) against the reference encoding (operand = col*H) on the same APC."""
    import ctypes as C

    from tests.test_oracle_apc import run_oracle_gpu_convention

    torch, abi, tg = gpu
    s = synth.generate("T1", seed=17)
    calls = 700
    apc, idx, want, hist, (bufs, dims, gt, order) = run_oracle_gpu_convention(s, calls, seed=17)
    W, H = want.shape
    # start from the gathered trace (no derived columns yet)
    out, _ = run_gpu(gpu, W, H, calls, bufs, dims, gt.air_names, gt.row_block_size, gt.subs, None, None)
    cb, offs, lens, dbc = om.compile_derived(apc, idx, 1)          # column-index operands
    specs = np.zeros(len(offs), dtype=[("col_base", "<u8"), ("off", "<u4"), ("len", "<u4")])
    specs["col_base"], specs["off"], specs["len"] = cb * H, offs, lens
    d_specs, d_bc = tg._dev(specs, "cuda"), tg._dev(dbc, "cuda")
    abi.lib.powdr_apc_apply_derived_expr_cols.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    abi.check(abi.lib.powdr_apc_apply_derived_expr_cols(out.ptr(), H, calls, d_specs.data_ptr(), len(offs), d_bc.data_ptr()), "derived_cols")
    torch.cuda.synchronize()
    assert (from_dev(out.buf).reshape(W, H) == want).all()
    inter, spans, bbc = om.compile_bus(apc, idx, 1)
    per = tg.Periphery.fresh()
    d = [tg._dev(np.ascontiguousarray(a), "cuda") for a in (bbc, inter, spans)]
    vp, sz, u32 = C.c_void_p, C.c_size_t, C.c_uint32
    abi.lib.powdr_apc_apply_bus_cols.argtypes = [vp, sz, C.c_int, vp, sz, vp, sz, vp, sz, u32, vp, sz, u32, vp, u32, u32, u32, vp]
    abi.check(abi.lib.powdr_apc_apply_bus_cols(out.ptr(), H, calls, d[0].data_ptr(), len(bbc), d[1].data_ptr(), len(inter), d[2].data_ptr(),
                                               len(spans), per.var_bus, per.var_hist.data_ptr(), per.var_hist.numel(), per.tuple_bus,
                                               per.tuple_hist.data_ptr(), 256, 2048, per.bitwise_bus, per.bitwise_hist.data_ptr()), "bus_cols")
    torch.cuda.synchronize()
    assert (hist_np(per.var_hist) == hist["var"]).all()
    assert (hist_np(per.tuple_hist) == hist["tuple"]).all()
    assert (hist_np(per.bitwise_hist) == hist["bitwise"]).all()


@pytest.mark.parametrize("seed", range(40))
def test_gather_fuzz(gpu, seed):
    """Random OriginalAir / Subst tables: several AIRs, arbitrary block sizes and heights, unsorted and
    duplicated destinations, block rows beyond row_block_size (the formula reads into the next call's
    block), partial last tiles — against the oracle's line-by-line restatement of the reference kernel."""
    torch, abi, tg = gpu
    rng = np.random.default_rng(1000 + seed)
    n_airs = int(rng.integers(1, 5))
    calls = int(rng.integers(1, 3000))
    H = synth.next_pow2_or_zero(calls)
    W = int(rng.integers(1, 40))
    airs, bufs, heights, rbs = [], [], [], []
    for _ in range(n_airs):
        w = int(rng.integers(1, 9))
        b = int(rng.choice([1, 2, 3, 5, 16, 31, 64, 100, 257, 400, 700]))
        extra = int(rng.integers(0, 4))  # rows a substitution may reach beyond its block
        h = max(4, synth.next_pow2_or_zero(b * calls + extra + 1))
        buf = rng.integers(0, om.P, size=w * h, dtype=np.uint32)
        airs.append((to_dev(torch, buf), w, h, b))
        bufs.append(buf); heights.append(h); rbs.append(b)
    n_subs = int(rng.integers(1, 120))
    recs = []
    for _ in range(n_subs):
        a = int(rng.integers(n_airs))
        _, w, h, b = airs[a]
        slack = h - b * calls
        row = int(rng.integers(0, b + min(3, max(0, slack - 1))))
        recs.append((a, int(rng.integers(w)), row, int(rng.integers(W))))
    recs = np.array(recs, np.int32)
    prefill = om.to_monty(rng.integers(0, om.P, size=H * W, dtype=np.uint32))
    want = om.from_monty(prefill.copy())
    om.c_apc_tracegen(H, W, bufs, heights, rbs, recs, calls, out=want)
    out = tg.DeviceMatrix.zeros(H, W)
    out.buf.copy_(torch.from_numpy(prefill.view(np.int32)))
    keep = tg.apc_tracegen(out, airs, recs, calls)
    torch.cuda.synchronize()
    assert (from_dev(out.buf) == want).all()


@pytest.mark.parametrize("seed", range(10))
def test_expression_kernels_fuzz(gpu, seed, monkeypatch):
    """Random derived columns and bus interactions (random expression trees, byte/small/field columns,
    random multiplicities incl. 0 and large ones) through every evaluator: post-fix interpreter vs xbc,
    direct atomics vs binned histograms — all against the oracle."""
    from tests.test_oracle_apc import _random_expr

    torch, abi, tg = gpu
    rng = np.random.default_rng(7000 + seed)
    W, calls = int(rng.integers(4, 30)), int(rng.integers(1, 2000))
    H = synth.next_pow2_or_zero(calls)
    ids = list(range(W))
    idx = {p: p for p in ids}
    trace = np.zeros((W, H), np.uint32)
    kind = rng.integers(0, 3, W)
    for c in range(W):
        hi = [256, 4, om.P][kind[c]]
        trace[c, :calls] = rng.integers(0, hi, calls)
    # derived columns: the last few columns, each may read earlier derived ones
    n_der = int(rng.integers(0, 4))
    der_cols = list(range(W - n_der, W))
    col_base, offs, lens, dbc = [], [], [], []
    for c in der_cols:
        e = _random_expr(rng, ids[: c], 3)
        off = len(dbc)
        if rng.random() < 0.5:
            om.emit_expr(dbc, _random_expr(rng, ids[: c], 2), idx, H)
            dbc.append(om.OP_INV_OR_ZERO)
            om.emit_expr(dbc, e, idx, H)
            dbc.append(om.OP_MUL)
        else:
            om.emit_expr(dbc, e, idx, H)
        col_base.append(c * H); offs.append(off); lens.append(len(dbc) - off)
    flat = trace.reshape(-1).copy()
    derived = (np.array(col_base, np.uint64), np.array(offs, np.uint32), np.array(lens, np.uint32), np.array(dbc, np.uint32))
    want = flat.copy()
    if n_der:
        om.c_apc_apply_derived(want, H, calls, *derived)
    # bus interactions
    bytes_cols = [c for c in range(W - n_der) if kind[c] == 0] or [0]
    small_cols = [c for c in range(W - n_der) if kind[c] == 1] or [0]
    inter, spans, bc = [], [], []

    def span(e):
        off = len(bc)
        om.emit_expr(bc, e, idx, H)
        spans.append((off, len(bc) - off))

    ref = lambda c: ("ref", "c", int(c))
    for _ in range(int(rng.integers(1, 40))):
        bus = int(rng.choice([3, 6, 7, 1, 9]))
        off_idx = len(spans)
        r = rng.random()
        mult = ("num", int(rng.choice([0, 1, 2, 5000]))) if r < 0.3 else ref(rng.choice(small_cols)) if r < 0.7 else _random_expr(rng, ids, 2)
        span(mult)
        if bus == 3:
            val = ref(rng.choice(bytes_cols)) if rng.random() < 0.7 else _random_expr(rng, ids, 2)
            span(val); span(("num", int(rng.integers(0, 18))))
            n_args = 2
        elif bus == 7:
            span(ref(rng.choice(bytes_cols))); span(("bin", "*", ref(rng.choice(bytes_cols)), ("num", int(rng.integers(1, 9)))))
            n_args = 2
        else:
            span(ref(rng.choice(bytes_cols))); span(ref(rng.choice(bytes_cols)) if rng.random() < 0.8 else _random_expr(rng, ids, 2))
            span(("num", 0)); span(("num", int(rng.integers(0, 3))) if rng.random() < 0.7 else ref(rng.choice(small_cols)))
            n_args = 4
        inter.append((bus, n_args, off_idx))
    inter = np.array(inter, np.uint32); spans = np.array(spans, np.uint32); bc = np.array(bc, np.uint32)
    hist = dict(var=np.zeros(1 << 18, np.uint32), tuple=np.zeros(256 * 2048, np.uint32), bitwise=np.zeros(2 * 65536, np.uint32))
    om.c_apc_apply_bus(want, calls, bc, inter, spans, 3, hist["var"], 7, hist["tuple"], 256, 2048, 6, hist["bitwise"])
    for binned in ("0", "1"):
        for xb in ("0", "1"):
            monkeypatch.setenv("POWDR_BUS_BINNED", binned)
            monkeypatch.setenv("POWDR_BUS_XBC", xb)
            out = tg.DeviceMatrix.zeros(H, W)
            out.buf.copy_(to_dev(torch, flat))
            if n_der:
                tg.apc_apply_derived_expr(out, calls, *derived)
            per = tg.Periphery.fresh()
            tg.apc_apply_bus(out, calls, bc, inter, spans, per)
            torch.cuda.synchronize()
            assert (from_dev(out.buf) == want).all()
            assert (hist_np(per.var_hist) == hist["var"]).all(), (binned, xb)
            assert (hist_np(per.tuple_hist) == hist["tuple"]).all(), (binned, xb)
            assert (hist_np(per.bitwise_hist) == hist["bitwise"]).all(), (binned, xb)


def test_field_helpers_selftest_on_device(gpu):
    """The arithmetic helpers' self-test (csrc/field_selftest.hpp) on the GPU: here the 64-bit multiply-add helpers are
    the inline v_mad_u64_u32 instructions, not their portable fall-backs."""
    import ctypes as C

    torch, abi, tg = gpu
    abi.lib.powdr_field_selftest_gpu.restype = C.c_int
    abi.lib.powdr_field_selftest_gpu.argtypes = [C.c_uint64, C.c_uint32, C.POINTER(C.c_int)]
    bad = C.c_int(-1)
    abi.check(abi.lib.powdr_field_selftest_gpu(7, 200, C.byref(bad)), "powdr_field_selftest_gpu")
    assert bad.value == 0


@pytest.mark.parametrize("shape,num_calls,seed", [("T0", 1, 0), ("T0", 5, 1), ("T0", 64, 2), ("T1", 37, 3), ("T1", 1000, 4),
                                                   ("T1", 4097, 5), ("C1", 300, 6), ("C1", 5000, 7)])
def test_callmajor_sources_give_the_same_trace(gpu, shape, num_calls, seed):
    """SURVEY.md §8 row (f)-1, layout half: sources handed over call-major and compacted (only the cells the APC uses,
    contiguous per call — what an original chip would write given the APC's (row, column) -> slot map) produce the SAME
    APC trace as the reference's column-major dummy traces: oracle (A) on the reference layout == oracle on the compacted
    layout == powdr_apc_tracegen_callmajor on the device; derived columns and the bus replay on top are unchanged."""
    from tests.test_oracle_apc import run_oracle_gpu_convention

    torch, abi, tg = gpu
    s = synth.generate(shape, seed=seed)
    apc, idx, want, hist, (bufs, dims, gt, order) = run_oracle_gpu_convention(s, num_calls, seed=seed)
    W, H = want.shape
    name_to = {n: i for i, (n, _, _, _) in enumerate(dims)}
    dense = [bufs[name_to[n]] for n in gt.air_names]
    heights = [dims[name_to[n]][2] for n in gt.air_names]
    cm_bufs, cells, subs_cm = om.compact_call_major(dense, heights, gt.row_block_size, gt.subs, num_calls)
    assert int(cells.sum()) <= len(gt.subs)  # nothing but used cells
    # the gather alone, against the reference-layout oracle
    ref_gather = om.c_apc_tracegen(H, W, dense, heights, gt.row_block_size, gt.subs, num_calls)
    assert (om.c_apc_tracegen_callmajor(H, W, cm_bufs, cells, subs_cm, num_calls) == ref_gather).all()
    out = tg.DeviceMatrix.zeros(H, W)
    keep = [to_dev(torch, b) for b in cm_bufs]
    tg.apc_tracegen_callmajor(out, list(zip(keep, cells)), subs_cm, num_calls)
    torch.cuda.synchronize()
    assert (from_dev(out.buf) == ref_gather).all()
    # a second call hits the cached plan; then the rest of trace generation gives the full oracle trace
    out.buf.zero_()
    tg.apc_tracegen_callmajor(out, list(zip(keep, cells)), subs_cm, num_calls)
    k2 = tg.apc_apply_derived_expr(out, num_calls, *om.compile_derived(apc, idx, H))
    per = tg.Periphery.fresh()
    inter, spans, bc = om.compile_bus(apc, idx, H)
    k3 = tg.apc_apply_bus(out, num_calls, bc, inter, spans, per)
    torch.cuda.synchronize()
    assert (from_dev(out.buf).reshape(W, H) == want).all()
    assert (hist_np(per.var_hist) == hist["var"]).all()
    del k2, k3


def test_callmajor_edge_cases(gpu):
    torch, abi, tg = gpu
    rng = np.random.default_rng(1)
    # duplicate destination: the last substitution wins; slots that nobody reads; rows beyond the calls are zero
    calls, H, U = 70, 128, 5
    buf = rng.integers(0, om.P, calls * U, dtype=np.uint32)
    subs_cm = np.array([[0, 0, 2], [0, 3, 0], [0, 1, 2], [0, 4, 1]], np.int32)  # column 2 written twice
    want = om.c_apc_tracegen_callmajor(H, 3, [buf], np.array([U], np.int32), subs_cm, calls).reshape(3, H)
    assert (want[2, :calls] == buf.reshape(calls, U)[:, 1]).all() and (want[:, calls:] == 0).all()
    out = tg.DeviceMatrix.zeros(H, 3)
    out.buf.fill_(7)
    t = to_dev(torch, buf)
    tg.apc_tracegen_callmajor(out, [(t, U)], subs_cm, calls)
    torch.cuda.synchronize()
    assert (from_dev(out.buf).reshape(3, H) == want).all()
    with pytest.raises(abi.HipError):
        tg.apc_tracegen_callmajor(out, [(t, U)], np.array([[0, 5, 0]], np.int32), calls)  # slot out of range
    with pytest.raises(abi.HipError):
        tg.apc_tracegen_callmajor(tg.DeviceMatrix.zeros(96, 3), [(t, U)], subs_cm, calls)  # height not a power of two


@pytest.mark.gpu
@pytest.mark.parametrize("calls,seed", [(1, 0), (37, 1), (1000, 2), (5000, 3)])
def test_column_structured_substitutions_parity(gpu, calls, seed):
    """The gather on the substitution structure of optimised APCs (few columns of every original AIR, present in most of its
    instructions: synth.column_structured_substitutions, calibrated on the reference's snapshots) — what bench.py's
    `tracegen_column_structured` record times — equals the oracle's restatement of apc_tracegen.cu:35-66."""
    torch, abi, tg = gpu
    rng = np.random.default_rng(seed)
    dims = [(36, 318), (53, 116), (41, 241), (26, 1), (18, 1)]  # the C2 keccak sources
    n_sub = 400
    subs = synth.column_structured_substitutions(dims, n_sub, seed=seed)
    assert len(set(map(tuple, subs[:, :3].tolist()))) == n_sub and sorted(subs[:, 3].tolist()) == list(range(n_sub))
    used = len(set(map(tuple, subs[:, :2].tolist())))
    assert used <= 0.2 * sum(w for w, _ in dims)  # few source columns
    H = max(synth.next_pow2_or_zero(calls), 2)
    bufs, hs, airs = [], [], []
    for w, b in dims:
        h = max(synth.next_pow2_or_zero(b * calls), 4)
        src = rng.integers(0, om.P, size=w * h, dtype=np.uint32)
        bufs.append(src); hs.append(h)
        airs.append((to_dev(torch, src), w, h, b))
    want = om.c_apc_tracegen(H, n_sub, bufs, hs, [b for _, b in dims], subs, calls)
    out = tg.DeviceMatrix.zeros(H, n_sub)
    tg.apc_tracegen(out, airs, subs, calls)
    torch.cuda.synchronize()
    assert (from_dev(out.buf) == want).all()


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"POWDR_GATHER_SPARSE_PCT": "200"}, {"POWDR_GATHER_SPARSE": "0"}])
@pytest.mark.parametrize("calls", [3, 700])
def test_gather_forms_forced(gpu, monkeypatch, env, calls):
    """The same substitutions through the cell-by-cell jobs everywhere (also where streaming would move fewer bytes: dense
    columns, duplicate source cells, rows reaching into the next call's block) and through the streaming jobs only."""
    torch, abi, tg = gpu
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(1000 * calls + len(env) + sum(map(len, env.values())))  # a plan of its own per case (plans are cached by content)
    dims = [(7, 40), (5, 3), (9, 130)]
    H = max(synth.next_pow2_or_zero(calls), 2)
    bufs, hs, airs = [], [], []
    for w, b in dims:
        h = max(synth.next_pow2_or_zero(b * (calls + 1)), 4)  # room for rows that reach past the last call's block
        src = rng.integers(0, om.P, size=w * h, dtype=np.uint32)
        bufs.append(src); hs.append(h)
        airs.append((to_dev(torch, src), w, h, b))
    recs = []
    for a, (w, b) in enumerate(dims):
        for col in range(w):
            dense = col % 3 == 0
            rows = range(b) if dense else sorted(rng.choice(b + 2, size=min(b, 5), replace=False).tolist())
            for r in rows:
                if dense and rng.random() < 0.3:
                    continue
                recs.append((a, col, int(r)))
                if rng.random() < 0.1:
                    recs.append((a, col, int(r)))  # the same source cell twice
    recs = [(a, c, r, i) for i, (a, c, r) in enumerate(recs)]
    subs = np.array(recs, np.int32)[rng.permutation(len(recs))]
    W = len(recs)
    want = om.c_apc_tracegen(H, W, bufs, hs, [b for _, b in dims], subs, calls)
    out = tg.DeviceMatrix.zeros(H, W)
    tg.apc_tracegen(out, airs, subs, calls)
    torch.cuda.synchronize()
    assert (from_dev(out.buf) == want).all()


def test_host_table_and_callmajor_argument_checks(gpu):
    """powdr_apc_tracegen_host_tables with substitutions but n_airs = 0 (it used to skip the range check and read
    the row-block table out of bounds), a Subst naming an AIR beyond the table, and a call-major slot that feeds two
    different APC columns (it used to drop one silently) are rejected; the inline form (no device table, <= 16 AIRs) equals
    the reference entry point."""
    import ctypes as C

    torch, abi, tg = gpu
    rng = np.random.default_rng(21)
    calls, H, w, b = 300, 512, 5, 3
    h = synth.next_pow2_or_zero(b * calls)
    src = rng.integers(0, om.P, size=w * h, dtype=np.uint32)
    d_src = to_dev(torch, src)
    subs = np.array([[0, 1, 2, 0], [0, 4, 0, 1], [0, 2, 1, 2]], np.int32)
    want = om.c_apc_tracegen(H, 3, [src], [h], [b], subs, calls)
    airs = (abi.OriginalAir * 1)(abi.OriginalAir(w, h, d_src.data_ptr(), b))
    f = abi.lib.powdr_apc_tracegen_host_tables
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    out = tg.DeviceMatrix.zeros(H, 3)
    assert f(out.ptr(), H, None, airs, 1, subs.ctypes.data, len(subs), calls) == 0  # no device table: kernel arguments
    torch.cuda.synchronize()
    assert (from_dev(out.buf) == want).all()
    assert f(out.ptr(), H, None, airs, 0, subs.ctypes.data, len(subs), calls) != 0
    bad = subs.copy()
    bad[1, 0] = 1
    assert f(out.ptr(), H, None, airs, 1, bad.ctypes.data, len(bad), calls) != 0
    # call-major: slot 1 of AIR 0 named for two different columns
    U = 4
    buf = to_dev(torch, rng.integers(0, om.P, calls * U, dtype=np.uint32))
    with pytest.raises(abi.HipError):
        tg.apc_tracegen_callmajor(tg.DeviceMatrix.zeros(H, 3), [(buf, U)], np.array([[0, 1, 0], [0, 1, 2], [0, 3, 1]], np.int32), calls)


def test_gather_knobs_are_part_of_the_plan_key(gpu, monkeypatch):
    """the SAME tables under different planner knobs must not share a cached plan."""
    torch, abi, tg = gpu
    rng = np.random.default_rng(31)
    calls, H, w, b = 2000, 2048, 4, 200
    h = synth.next_pow2_or_zero(b * calls)
    src = rng.integers(0, om.P, size=w * h, dtype=np.uint32)
    d_src = to_dev(torch, src)
    rows = np.sort(rng.choice(b, 9, replace=False))
    subs = np.array([[0, 2, int(r), i] for i, r in enumerate(rows)], np.int32)
    want = om.c_apc_tracegen(H, 9, [src], [h], [b], subs, calls)
    forms = []
    for env in ({}, {"POWDR_GATHER_SPARSE": "0"}, {}):
        for k in ("POWDR_GATHER_SPARSE",):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out = tg.DeviceMatrix.zeros(H, 9)
        abi.call_stats(reset=True)
        tg.apc_tracegen(out, [(d_src, w, h, b)], subs, calls)
        torch.cuda.synchronize()
        st = abi.call_stats()
        forms.append((st["gather_sparse_jobs"], st["gather_whole_jobs"] + st["gather_chunk_jobs"]))
        assert (from_dev(out.buf) == want).all()
    assert forms[0] == forms[2] == (1, 0) and forms[1][0] == 0 and forms[1][1] >= 1, forms
