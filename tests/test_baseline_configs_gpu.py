"""Byte parity on the BASELINE configs' OWN plans ("next round" item 1).

The shapes bench.py times — C2's 2 017 uniformly scattered substitutions over source blocks of 318 / 241 / 116 rows with
its 1 734 bus interactions, C3's 3 731 columns with the tuple bus and column-index operands, the 29-AIR pairing-shaped and
the 64-AIR reth-shaped segments — are compared here with the CPU oracle word for word, at heights the oracle finishes in
seconds, through the same entry points the timed step uses. `powdr_gpu_call_stats` proves that the job forms and kernels the
full-size run depends on (cell-by-cell / whole-block / chunked gather jobs, the small-form bus kernel, the binned histogram
path) are the ones that ran.

Reference: openvm/cuda/src/apc_tracegen.cu:35-66, apc_apply_bus.cu:23-113, shape pins openvm-riscv/src/lib.rs:1377-1458
(keccak APC), :1114-1122 (the 19 system AIRs)."""
import ctypes as C

import numpy as np
import pytest

from oracle import apc_model as om
from oracle import stark_model as sm
from powdr_amd import synth

pytestmark = pytest.mark.gpu
P = om.P


@pytest.fixture(scope="module")
def gpu():
    import torch
    from powdr_amd import abi, host, prover, tracegen

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU (run with -m gpu on the GPU box)")
    return torch, abi, tracegen, host, prover


def to_dev(torch, a):
    return torch.from_numpy(om.to_monty(np.ascontiguousarray(a, dtype=np.uint32)).view(np.int32)).cuda()


def from_dev(t):
    return om.from_monty(t.cpu().numpy().view(np.uint32))


def hist_np(t):
    return t.cpu().numpy().view(np.uint32)


def _check_hists(per, hist):
    assert (hist_np(per.var_hist) == hist["var"]).all()
    assert (hist_np(per.tuple_hist) == hist["tuple"]).all()
    assert (hist_np(per.bitwise_hist) == hist["bitwise"]).all()


def _reference_abi(gpu, s, calls, seed, oracle, operands="offsets"):
    """_apc_tracegen -> _apc_apply_derived_expr -> _apc_apply_bus (operands="cols": the column-index extensions)."""
    torch, abi, tg, host, prover = gpu
    apc, idx, want, hist, (bufs, dims, gt, order) = oracle
    W, H = want.shape
    name_to = {n: i for i, (n, _, _, _) in enumerate(dims)}
    airs, keep = [], []
    for n, b in zip(gt.air_names, gt.row_block_size):
        _, w, h, _ = dims[name_to[n]]
        t = to_dev(torch, bufs[name_to[n]])
        keep.append(t)
        airs.append((t, w, h, b))
    out = tg.DeviceMatrix.zeros(H, W)
    abi.call_stats(reset=True)
    keep.append(tg.apc_tracegen(out, airs, gt.subs, calls))
    per = tg.Periphery.fresh()
    if operands == "offsets":
        keep.append(tg.apc_apply_derived_expr(out, calls, *om.compile_derived(apc, idx, H)))
        inter, spans, bc = om.compile_bus(apc, idx, H)
        keep.append(tg.apc_apply_bus(out, calls, bc, inter, spans, per))
    else:
        cb, offs, lens, dbc = om.compile_derived(apc, idx, 1)
        specs = np.zeros(len(offs), dtype=[("col_base", "<u8"), ("off", "<u4"), ("len", "<u4")])
        specs["col_base"], specs["off"], specs["len"] = cb * H, offs, lens
        d_specs, d_bc = tg._dev(specs, "cuda"), tg._dev(dbc, "cuda")
        vp, sz, u32 = C.c_void_p, C.c_size_t, C.c_uint32
        abi.lib.powdr_apc_apply_derived_expr_cols.argtypes = [vp, sz, C.c_int, vp, sz, vp]
        abi.check(abi.lib.powdr_apc_apply_derived_expr_cols(out.ptr(), H, calls, d_specs.data_ptr(), len(offs), d_bc.data_ptr()), "derived_cols")
        inter, spans, bbc = om.compile_bus(apc, idx, 1)
        d = [tg._dev(np.ascontiguousarray(a), "cuda") for a in (bbc, inter, spans)]
        abi.lib.powdr_apc_apply_bus_cols.argtypes = [vp, sz, C.c_int, vp, sz, vp, sz, vp, sz, u32, vp, sz, u32, vp, u32, u32, u32, vp]
        abi.check(abi.lib.powdr_apc_apply_bus_cols(out.ptr(), H, calls, d[0].data_ptr(), len(bbc), d[1].data_ptr(), len(inter), d[2].data_ptr(),
                                                   len(spans), per.var_bus, per.var_hist.data_ptr(), per.var_hist.numel(), per.tuple_bus,
                                                   per.tuple_hist.data_ptr(), 256, 2048, per.bitwise_bus, per.bitwise_hist.data_ptr()), "bus_cols")
        keep += d + [d_specs, d_bc]
    torch.cuda.synchronize()
    stats = abi.call_stats()
    got = from_dev(out.buf).reshape(W, H)
    assert (got == want).all(), f"{int((got != want).sum())} cells differ"
    _check_hists(per, hist)
    del keep
    return stats


def _host_entry(gpu, s, calls, oracle):
    """powdr_apc_generate_witness_gpu (= try_generate_witness, cuda/mod.rs:201-401): JSON in, trace + histograms out."""
    torch, abi, tg, host, prover = gpu
    apc, idx, want, hist, (bufs, dims, gt, order) = oracle
    W, H = want.shape
    names = [n if n else om.opcode_air(ins[0]) for n, ins in zip(s.instr_air, s.doc["block"]["blocks"][0]["instructions"])]
    air_order, ids = [], []
    for n in names:
        if n not in air_order:
            air_order.append(n)
        ids.append(air_order.index(n))
    name_to = {n: i for i, (n, _, _, _) in enumerate(dims)}
    dev, dummy = [], []
    for n in air_order:
        if n in name_to:
            _, w, h, _ = dims[name_to[n]]
            t = to_dev(torch, bufs[name_to[n]])
            dev.append(t)
            dummy.append((t.data_ptr(), w, h))
        else:
            dummy.append((0, 0, 0))
    h_apc = host.Apc(s.doc)
    out = torch.full((H * W,), 7, dtype=torch.int32, device="cuda")  # zero-filled by the callee
    per = tg.Periphery.fresh()
    abi.call_stats(reset=True)
    h_apc.generate_witness_gpu(ids, dummy, calls, out.data_ptr(), per)
    torch.cuda.synchronize()
    stats = abi.call_stats()
    assert (from_dev(out).reshape(W, H) == want).all()
    _check_hists(per, hist)
    h_apc.close()
    return stats, out


@pytest.mark.parametrize("calls", [1 << 14, 3 * (1 << 12), (1 << 15) + 7])
def test_c2_trace_generation_plan_matches_oracle(gpu, calls):
    """BASELINE configs[1]'s own trace-generation plan: synth.generate("C2") — 2 022 columns gathered from (w, b) = (36, 318),
    (53, 116), (41, 241), (26, 1), (18, 1), 4 QuotientOrZero columns, 1 734 bus interactions — at 2^14 calls, at 0.75 H (zero
    padding) and at 2^15 + 7 calls (H = 2^16, a partial last tile): trace and the three histograms equal the oracle's, through
    the three reference entry points AND through powdr_apc_generate_witness_gpu; the plan holds cell-by-cell, whole-block and
    (with POWDR_GATHER_SPARSE=0) chunked jobs, the bus replay ran its small-form kernel and the binned histogram path."""
    from tests.test_oracle_apc import run_oracle_gpu_convention

    torch, abi, tg, host, prover = gpu
    s = synth.generate("C2", seed=0)
    oracle = run_oracle_gpu_convention(s, calls, seed=0)
    W, H = oracle[2].shape
    assert W == 2022 and H == synth.next_pow2_or_zero(calls) and len(oracle[4][2].subs) == 2017
    # instructions that keep at least one cell, per original AIR (of 116 / 241 / 318 / 1 / 1): the row blocks bench.py's plan has
    assert sorted(oracle[4][2].row_block_size) == [1, 1, 113, 227, 300]
    st = _reference_abi(gpu, s, calls, 0, oracle)
    # the timed plan: scattered survivors -> cell-by-cell jobs for the long blocks, whole-block jobs for the one-row AIRs
    assert st["gather_sparse_jobs"] > 100 and st["gather_whole_jobs"] >= 1 and st["gather_calls"] == 1
    assert st["bus_fast_interactions"] > 900 and st["bus_binned_windows"] == (1 if calls >= 16384 else 0)
    assert st["bus_direct_calls"] == (0 if calls >= 16384 else 1) and st["bus_xbc_calls"] == 1
    st2, _ = _host_entry(gpu, s, calls, oracle)
    assert st2["gather_sparse_jobs"] == st["gather_sparse_jobs"] and st2["bus_fast_interactions"] == st["bus_fast_interactions"]


@pytest.mark.parametrize("env,expect", [({"POWDR_GATHER_SPARSE": "0"}, "gather_chunk_jobs"), ({"POWDR_BUS_BINNED": "1"}, "bus_binned_windows"),
                                        ({"POWDR_BUS_FAST": "0", "POWDR_BUS_BINNED": "1"}, "bus_interpreted_interactions"),
                                        ({"POWDR_BUS_XBC": "0", "POWDR_BUS_BINNED": "0"}, "bus_direct_calls")])
def test_c2_plan_through_every_job_form_and_bus_kernel(gpu, monkeypatch, env, expect):
    """The same C2 plan at 5 000 calls with each alternative forced: streaming gather jobs only (the chunked form: blocks of
    318 / 241 rows do not fit one LDS tile), the binned path below its threshold, the interpreter instead of the small-form
    kernel, the reference post-fix interpreter with direct atomics. All equal the oracle."""
    from tests.test_oracle_apc import run_oracle_gpu_convention

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    # plans are cached by content: one APC per case (seed 33's padded expressions need 17 stack slots: the reference's evaluator,
    # expr_eval.cuh:22, and the oracle stop at 16)
    s = synth.generate("C2", seed={"gather_chunk_jobs": 21, "bus_binned_windows": 18, "bus_interpreted_interactions": 34, "bus_direct_calls": 32}[expect])
    calls = 5000
    oracle = run_oracle_gpu_convention(s, calls, seed=3)
    st = _reference_abi(gpu, s, calls, 3, oracle)
    assert st[expect] >= 1, st
    if expect == "gather_chunk_jobs":
        assert st["gather_sparse_jobs"] == 0 and st["gather_chunk_jobs"] > 40 and st["gather_whole_jobs"] > 40


@pytest.mark.parametrize("calls,operands", [(1 << 12, "offsets"), ((1 << 12) - 77, "cols")])
def test_c3_shape_trace_generation_matches_oracle(gpu, calls, operands):
    """BASELINE configs[2]'s AIR (3 731 columns, 3 114 constraints, 2 314 bus interactions incl. the tuple bus; six source
    AIRs with blocks up to 420 rows) at 2^12 rows: the reference encoding and the column-index encoding the full-size run
    needs (W * H > 2^32 at 2^22 rows: powdr_apc_apply_*_cols) against the oracle."""
    from tests.test_oracle_apc import run_oracle_gpu_convention

    s = synth.generate("C3", seed=0)
    oracle = run_oracle_gpu_convention(s, calls, seed=0)
    assert oracle[2].shape == (3731, 1 << 12) and oracle[3]["tuple"].sum() > 0
    st = _reference_abi(gpu, s, calls, 0, oracle, operands)
    assert st["gather_sparse_jobs"] > 100 and st["bus_fast_interactions"] > 1000
    if operands == "cols":
        _host_entry(gpu, s, calls, oracle)


def test_c3_shape_proof_bytes_match_oracle(gpu):
    """... and its proof (3 114 constraint programs) at 2^12 rows, constraints-only and with the LogUp phase."""
    from tests.test_oracle_apc import run_oracle_gpu_convention

    torch, abi, tg, host, prover = gpu
    s = synth.generate("C3", seed=0)
    apc, idx, trace, _, _ = run_oracle_gpu_convention(s, (1 << 12) - 3, seed=0)
    W, H = trace.shape
    flat = np.ascontiguousarray(trace).reshape(-1)
    bc, spans = sm.compile_constraints(apc, idx)
    assert len(spans) == 3114
    it = sm.compile_interactions(apc, idx)
    d_t = to_dev(torch, flat)
    for logup in (False, True):
        pr = prover.Prover(W, bc, spans, num_queries=6, pow_bits=4, interactions=it if logup else None)
        got = pr.prove(d_t.data_ptr(), 12)
        want = sm.prove_logup(flat, W, 12, bc, spans, *it, num_queries=6, pow_bits=4) if logup else \
            sm.prove(flat, W, 12, bc, spans, num_queries=6, pow_bits=4)
        assert len(got) == len(want) and (got == want).all(), f"logup={logup}: first differing word {int(np.argmax(got != want))}"
        pr.close()


def _segment_airs(kind, cap, seed):
    """(name, width, log_height, n_constraints, n_interactions) of bench.py's C4 / C5 segments with every height reduced by
    the same factor so that the tallest trace has 2^cap rows (widths, AIR count, the number of distinct heights and the
    program sizes are the full-size ones)."""
    shapes = synth.segment_shape(kind, seed=seed, max_log_height=20)
    top = max(s[2] for s in shapes)
    return [(n, w, max(1, lh - (top - cap)), nc, ni) for n, w, lh, nc, ni in shapes]


@pytest.mark.parametrize("kind,cap", [("C4", 12), ("C5", 12)])
@pytest.mark.parametrize("logup", [False, True])
def test_c4_c5_segment_proof_bytes_match_oracle(gpu, kind, cap, logup):
    """BASELINE configs[3] / configs[4] as bench.py builds them (`multi_segment` leg, --shape C4|C5): 10 APC AIRs (widths
    160-520) + the 19 system AIRs (819 columns, 643 constraints, 253 interactions); 45 APC AIRs (widths 30-3 700, eleven
    distinct heights) + the system AIRs. ONE pw-stark v1 proof per segment, heights capped at 2^12: the HIP words equal
    sm.prove_segment's and both verifiers accept."""
    torch, abi, tg, host, prover = gpu
    shapes = _segment_airs(kind, cap, seed=0)
    assert len(shapes) == (29 if kind == "C4" else 64) and max(s[1] for s in shapes) >= (520 if kind == "C4" else 3000)
    assert len({s[2] for s in shapes}) >= (8 if kind == "C4" else 10)
    rng = np.random.default_rng(7)
    airs, provers, traces = [], [], []
    for k, (name, w, lh, nc, ni) in enumerate(shapes):
        bc, sp, it = synth.air_programs(name, w, nc, ni, seed=k)
        flat = rng.integers(0, P, size=w << lh, dtype=np.uint32)
        airs.append((flat, w, lh, bc, sp, it))
        provers.append(prover.Prover(w, bc, sp, num_queries=5, pow_bits=3, interactions=it if logup else None))
        traces.append(to_dev(torch, flat))
    got = prover.prove_segment([(pr, t.data_ptr(), a[2]) for pr, t, a in zip(provers, traces, airs)], logup=logup)
    want = sm.prove_segment(airs, num_queries=5, pow_bits=3, logup=logup)
    assert len(got) == len(want)
    assert (got == want).all(), f"first differing word {int(np.argmax(got != want))} of {len(want)}"
    descs = [(a[1], a[2], a[3], a[4], a[5]) for a in airs]
    # random traces do not satisfy the random constraint programs: both verifiers must name the same failing AIR
    rc = prover.verify_segment(descs, got, 5, 3, logup)[0]
    assert rc == sm.verify_segment(got, airs, 5, 3, logup)[0] and rc & 0xFF == 2
    for pr in provers:
        pr.close()
