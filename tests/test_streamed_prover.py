"""The STREAMED proof path (include/powdr_prover.h "STREAMED proofs"; csrc/prover.hip): traces whose low-degree extension does
not fit in HBM — BASELINE configs[2], 3 731 columns x 2^22 rows with its 2 314 bus interactions — are proven from coefficient
arrays, one sub-coset of the extended domain at a time. Exact field arithmetic, so the proof WORDS must equal the resident
path's and the oracle's: these tests force the streamed mode (POWDR_STREAM_LOG_BLOCKS) on traces the oracle can prove in seconds."""
import numpy as np
import pytest

from oracle import apc_model as om
from oracle import stark_model as sm
from powdr_amd import synth

P = om.P
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    from powdr_amd import abi, prover

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch, abi, prover


def to_dev(torch, a):
    return torch.from_numpy(om.to_monty(np.ascontiguousarray(a, dtype=np.uint32)).view(np.int32)).cuda()


def from_dev(t):
    return om.from_monty(t.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("log_h,W,log_blocks", [(3, 2, 1), (3, 2, 2), (3, 1, 3), (6, 3, 1), (6, 3, 4), (10, 2, 3), (12, 3, 1), (12, 2, 2), (13, 2, 3),
                                                 (14, 2, 5), (16, 3, 3), (18, 2, 2), (20, 2, 3), (20, 1, 1), (21, 1, 4)])
def test_subcoset_lde_is_the_rows_of_the_lde(gpu, log_h, W, log_blocks):
    """pw_lde_subcoset: rows r + 2^b i of the oracle's LDE, from the coefficient arrays pw_lde_batch leaves, for every r (small b)
    or a sample of them."""
    torch, abi, prover = gpu
    rng = np.random.default_rng(100 * log_h + log_blocks)
    H = 1 << log_h
    t = rng.integers(0, P, W * H, dtype=np.uint32)
    want = sm.lde(t, W, log_h).reshape(W, 2 * H)
    d_t = to_dev(torch, t)
    d_c = torch.empty(W * H, dtype=torch.int32, device="cuda")
    d_l = torch.empty(W * 2 * H, dtype=torch.int32, device="cuda")
    abi.check(prover.lib.pw_lde_batch(d_t.data_ptr(), W, log_h, d_c.data_ptr(), d_l.data_ptr()), "pw_lde_batch")
    B = 1 << log_blocks
    m = 2 * H // B
    d_s = torch.empty(H, dtype=torch.int32, device="cuda")
    d_o = torch.empty(W * m, dtype=torch.int32, device="cuda")
    rs = range(B) if B <= 8 else sorted({0, 1, B // 2, B - 1, int(rng.integers(0, B))})
    for r in rs:
        d_o.zero_()
        abi.check(prover.lib.pw_lde_subcoset(d_c.data_ptr(), W, log_h, log_blocks, r, d_s.data_ptr(), d_o.data_ptr()), "pw_lde_subcoset")
        torch.cuda.synchronize()
        got = from_dev(d_o).reshape(W, m)
        assert (got == want[:, r::B]).all(), f"sub-coset {r} of {B}"


def _baseline_case(shape, log_h):
    from tests._oracle_cases import baseline_case

    return baseline_case(shape, log_h)


def _synthetic(shape, calls, seed):
    from tests._oracle_cases import synthetic

    return synthetic(shape, calls, seed)


def _prove(prover, monkeypatch, d_t, W, log_h, bc, spans, it, nq, pow_bits, log_blocks, jit):
    monkeypatch.setenv("POWDR_STREAM_LOG_BLOCKS", str(log_blocks))
    monkeypatch.setenv("POWDR_JIT", "1" if jit else "0")
    pr = prover.Prover(W, bc, spans, num_queries=nq, pow_bits=pow_bits, interactions=it)
    assert pr.stream_log_blocks(log_h) == min(log_blocks, max(log_h - 1, 0))
    got = pr.prove(d_t.data_ptr(), log_h)
    again = pr.prove(d_t.data_ptr(), log_h)  # buffer reuse
    assert (got == again).all()
    state = pr.specialised()["state"]
    pr.close()
    return got, state


@pytest.mark.parametrize("shape,calls,nq,pow_bits", [("T0", 7, 4, 0), ("T0", 64, 5, 5), ("T1", 100, 6, 0), ("T1", 1000, 8, 3), ("T1", 5000, 10, 0)])
@pytest.mark.parametrize("logup", [False, True])
def test_streamed_proof_words_equal_the_oracle(gpu, monkeypatch, shape, calls, nq, pow_bits, logup):
    """Every stage of the streamed path — commitments from sub-cosets, quotient terms per sub-coset + boundary terms, openings of
    the permutation matrix from its coefficients, the DEEP numerator as an extended polynomial, query rows from a last pass —
    against the oracle, for 2, 4 and 8 sub-cosets, with the interpreter and with the run-time specialised kernels."""
    torch, abi, prover = gpu
    flat, W, log_h, bc, spans, it = _synthetic(shape, calls, seed=21)
    want = sm.prove_logup(flat, W, log_h, bc, spans, *it, num_queries=nq, pow_bits=pow_bits) if logup else \
        sm.prove(flat, W, log_h, bc, spans, num_queries=nq, pow_bits=pow_bits)
    d_t = to_dev(torch, flat)
    for log_blocks in (1, 2, 3):
        for jit in (False, True):
            got, state = _prove(prover, monkeypatch, d_t, W, log_h, bc, spans, it if logup else None, nq, pow_bits, log_blocks, jit)
            assert (state == 1) if jit else (state in (0, -1))
            assert len(got) == len(want) and (got == want).all(), \
                f"blocks 2^{log_blocks} jit={jit}: first differing word {int(np.argmax(got != want))} of {len(want)}"
    # the specialised quotient without permutation panels (all permutation columns of a sub-coset at once): same words
    monkeypatch.setenv("POWDR_STREAM_NO_PANELS", "1")
    got, state = _prove(prover, monkeypatch, d_t, W, log_h, bc, spans, it if logup else None, nq, pow_bits, 2, True)
    monkeypatch.delenv("POWDR_STREAM_NO_PANELS")
    assert state == 1 and (got == want).all()
    # the resident path on the same prover inputs (POWDR_STREAM_LOG_BLOCKS=0): same words
    got, _ = _prove(prover, monkeypatch, d_t, W, log_h, bc, spans, it if logup else None, nq, pow_bits, 0, False)
    assert (got == want).all()


@pytest.mark.parametrize("shape,log_h,log_blocks,jit", [("C3", 12, 3, False), ("C3", 12, 2, True), ("C2", 14, 3, True), ("C2", 14, 4, False)])
def test_baseline_shapes_streamed_with_logup(gpu, monkeypatch, shape, log_h, log_blocks, jit):
    """the C3 shape (3 731 columns, 3 114 constraints, 2 314 interactions) at 2^12 rows and the C2 shape at 2^14
    through the streamed path with the LogUp phase: words == sm.prove_logup, both verifiers accept."""
    torch, abi, prover = gpu
    (flat, W, lh, bc, spans, it), want = _baseline_case(shape, log_h)
    d_t = to_dev(torch, flat)
    got, state = _prove(prover, monkeypatch, d_t, W, log_h, bc, spans, it, 6, 4, log_blocks, jit)
    assert (state == 1) if jit else (state in (0, -1))
    assert len(got) == len(want) and (got == want).all(), f"first differing word {int(np.argmax(got != want))} of {len(want)}"
    assert prover.verify_logup(got, W, log_h, bc, spans, it, num_queries=6, pow_bits=4)[0] == 0
    assert sm.verify_logup(got, W, log_h, bc, spans, *it, num_queries=6, pow_bits=4) == 0


@pytest.mark.parametrize("log_blocks,jit", [(1, False), (2, True)])
def test_tall_trace_query_rows_from_the_partial_transform(gpu, monkeypatch, log_blocks, jit):
    """From 2^14-point sub-cosets on, the transform has strided stage groups and the query phase finishes them for the queried rows
    only (subcoset_lde_first_group + subcoset_rows): 2^16 rows, sub-cosets of 2^16 and 2^15 points."""
    torch, abi, prover = gpu
    flat, W, log_h, bc, spans, it = _synthetic("T1", 40000, seed=3)
    assert log_h == 16
    want = sm.prove_logup(flat, W, log_h, bc, spans, *it, num_queries=12, pow_bits=0)
    d_t = to_dev(torch, flat)
    # the three forms of the query pass (ntt.hip subcoset_query_rows): the tiles' terms stored and reduced (default), round 5's 64-bit
    # atomic sums, the stored partial transform + subcoset_rows — the same words
    for select in ("1", "2", "0"):
        monkeypatch.setenv("POWDR_QUERY_SELECT", select)
        got, _ = _prove(prover, monkeypatch, d_t, W, log_h, bc, spans, it, 12, 0, log_blocks, jit)
        assert len(got) == len(want) and (got == want).all(), f"select {select}: first differing word {int(np.argmax(got != want))} of {len(want)}"


def test_streamed_equals_resident_at_2_to_18_rows(gpu, monkeypatch):
    """Beyond what the oracle proves in seconds: the C2 AIR (2 022 columns, 187 constraints, 1 734 interactions) on a 2^18-row trace,
    specialised kernels, 4 and 8 sub-cosets (three stage groups per sub-coset transform, permutation panels by unit) against the
    resident path — the same words, which the product's verifier accepts or rejects alike (the trace is random: code 2)."""
    torch, abi, prover = gpu
    from powdr_amd import host

    s = synth.generate("C2", seed=0)
    apc = host.Apc(s.doc)
    bc, spans = apc.compile_constraints()
    it = apc.compile_bus(1)
    W, log_h = apc.width, 18
    d_t = torch.empty(W << log_h, dtype=torch.int32, device="cuda")
    d_t.random_(0, P)
    monkeypatch.setenv("POWDR_JIT", "1")
    proofs = {}
    for b in (0, 2, 3):
        monkeypatch.setenv("POWDR_STREAM_LOG_BLOCKS", str(b))
        pr = prover.Prover(W, bc, spans, num_queries=20, pow_bits=8, interactions=it)
        proofs[b] = pr.prove(d_t.data_ptr(), log_h)
        assert pr.specialised()["state"] == 1
        pr.close()
    assert (proofs[2] == proofs[0]).all() and (proofs[3] == proofs[0]).all()
    assert prover.verify_logup(proofs[2], W, log_h, bc, spans, it, num_queries=20, pow_bits=8)[0] == 2
    apc.close()


def test_trace_root_then_prove_in_streamed_mode(gpu, monkeypatch):
    """pw_prover_trace_root leaves the streamed commitment (coefficients + tree) for the proof that follows; a shared bus seed
    works as in the resident mode."""
    torch, abi, prover = gpu
    flat, W, log_h, bc, spans, it = _synthetic("T1", 700, seed=5)
    d_t = to_dev(torch, flat)
    monkeypatch.setenv("POWDR_STREAM_LOG_BLOCKS", "0")
    pr = prover.Prover(W, bc, spans, num_queries=5, interactions=it)
    root0 = pr.trace_root(d_t.data_ptr(), log_h)
    pr.set_bus_seed(root0[::-1].copy())
    want = pr.prove(d_t.data_ptr(), log_h)
    pr.close()
    monkeypatch.setenv("POWDR_STREAM_LOG_BLOCKS", "2")
    pr = prover.Prover(W, bc, spans, num_queries=5, interactions=it)
    root = pr.trace_root(d_t.data_ptr(), log_h)
    assert (root == root0).all()
    pr.set_bus_seed(root0[::-1].copy())
    got = pr.prove(d_t.data_ptr(), log_h)
    assert (got == want).all()
    pr.close()


def test_mode_policy(gpu, monkeypatch):
    """Unset, the mode follows the free memory: a small trace is resident; a shape that cannot fit resident is streamed with the
    smallest number of sub-cosets whose buffers fit (here checked through the planner only: nothing is allocated)."""
    torch, abi, prover = gpu
    monkeypatch.delenv("POWDR_STREAM_LOG_BLOCKS", raising=False)
    bc, sp, it = synth.air_programs("apc", 64, 10, 20, seed=1)
    pr = prover.Prover(64, bc, sp, num_queries=4, interactions=it)
    assert pr.stream_log_blocks(12) == 0 and pr.stream_log_blocks(18) == 0
    pr.close()
    free = torch.cuda.mem_get_info()[0]
    # a width whose resident LDE alone exceeds the free memory at 2^22 rows
    W = int(free // (8 << 22)) + 64
    bc, sp, it = synth.air_programs("apc", W, 4, 8, seed=2)
    pr = prover.Prover(W, bc, sp, num_queries=4, interactions=None)
    b = pr.stream_log_blocks(22)
    assert b >= 1 or b == -1
    pr.close()


@pytest.mark.parametrize("workers", [1, 3])
def test_streamed_airs_of_a_segment_share_a_bus_seed_and_balance(gpu, monkeypatch, workers):
    """A segment whose AIRs are proven independently (pw_prove_airs: two phases, shared bus seed — the flow that also shards AIRs over
    GPUs) works unchanged when its proofs are streamed: the phase-1 commitment (pw_prover_trace_root) leaves coefficients + tree for
    phase 2, the proofs equal the resident ones, the APC's lookups and the periphery AIRs' receives cancel."""
    torch, abi, prover = gpu
    from powdr_amd import periphery, tracegen as tg
    from tests.test_oracle_apc import run_oracle_gpu_convention
    from tests.test_tracegen_gpu import run_gpu

    no_cons = (np.zeros(0, np.uint32), np.zeros((0, 2), np.uint32))
    s = synth.generate("T1", seed=6)
    apc, idx, want, hist, (bufs, dims, gt, order) = run_oracle_gpu_convention(s, 3000, seed=6)
    W, H = want.shape
    log_h = H.bit_length() - 1
    out, per = run_gpu((torch, None, tg), W, H, 3000, bufs, dims, gt.air_names, gt.row_block_size, gt.subs,
                       om.compile_derived(apc, idx, H), om.compile_bus(apc, idx, H))
    cons = sm.compile_constraints(apc, idx)
    sends = periphery.select_buses(sm.compile_interactions(apc, idx), {per.var_bus, per.tuple_bus})
    airs = [(out.buf, W, cons, sends, log_h),
            (periphery.var_range_trace(per.var_hist), 3, no_cons, periphery.var_range_interactions(per.var_bus), per.var_hist.numel().bit_length() - 1),
            (periphery.tuple2_trace(per.tuple_hist, per.tuple_sizes), 3, no_cons, periphery.tuple2_interactions(per.tuple_bus),
             per.tuple_hist.numel().bit_length() - 1)]
    proofs = {}
    for b in (0, 2):
        monkeypatch.setenv("POWDR_STREAM_LOG_BLOCKS", str(b))
        provers = [prover.Prover(w, *c, num_queries=7, interactions=it) for (_, w, c, it, _) in airs]
        assert [pr.stream_log_blocks(lh) for pr, (*_, lh) in zip(provers, airs)] == [b] * 3
        proofs[b], seed = prover.prove_airs([(pr, t.data_ptr(), lh) for pr, (t, _, _, _, lh) in zip(provers, airs)], shared_bus_seed=True,
                                            n_workers=workers)
        proofs[b] = [np.array(x, copy=True) for x in proofs[b]]
        for pr in provers:
            pr.close()
    for a, c in zip(proofs[0], proofs[2]):
        assert len(a) == len(c) and (a == c).all()
    descs = [(w, lh, *c, it) for (_, w, c, it, lh) in airs]
    rc, total = prover.verify_airs(descs, proofs[2], num_queries=7, shared_bus_seed=True, check_balance=True)
    assert rc == 0 and (total == 0).all()


# ---- the trace handed over (pw_prover_prove_consuming, VERDICT r4 #2b) -------------------------------------------------------------
def _prove_consuming(torch, prover, monkeypatch, flat, W, log_h, bc, spans, it, nq, pow_bits, log_blocks, jit):
    """One consuming proof on a fresh copy of the trace; returns (words, what is left in the buffer, the buffer restored, state)."""
    monkeypatch.setenv("POWDR_STREAM_LOG_BLOCKS", str(log_blocks))
    monkeypatch.setenv("POWDR_JIT", "1" if jit else "0")
    pr = prover.Prover(W, bc, spans, num_queries=nq, pow_bits=pow_bits, interactions=it)
    assert pr.stream_log_blocks_consuming(log_h) == min(log_blocks, max(log_h - 1, 0))
    streamed = pr.stream_log_blocks_consuming(log_h) > 0  # (a resident proof leaves the trace alone)
    d_t = to_dev(torch, flat)
    got = pr.prove(d_t.data_ptr(), log_h, consume=True)
    torch.cuda.synchronize()
    left = d_t.clone()
    if streamed:
        prover.trace_from_coefficients(d_t.data_ptr(), W, log_h)
    again = pr.prove(d_t.data_ptr(), log_h, consume=True)  # buffer reuse, on the restored trace
    assert (got == again).all()
    # a plain proof on the same prover afterwards (its tcoef buffer comes back): same words
    if streamed:
        prover.trace_from_coefficients(d_t.data_ptr(), W, log_h)
    plain = pr.prove(d_t.data_ptr(), log_h)
    assert (got == plain).all()
    # ... and a consuming proof after the plain streamed one (tcoef is released BEFORE the other buffers grow)
    third = pr.prove(d_t.data_ptr(), log_h, consume=True)
    assert (got == third).all()
    if streamed:
        prover.trace_from_coefficients(d_t.data_ptr(), W, log_h)
    state = pr.specialised()["state"]
    pr.close()
    return got, left, d_t, state


@pytest.mark.parametrize("shape,calls,nq,pow_bits", [("T0", 7, 4, 0), ("T1", 100, 6, 0), ("T1", 1000, 8, 3), ("T1", 5000, 10, 0)])
@pytest.mark.parametrize("logup", [False, True])
def test_consuming_proof_words_equal_the_oracle(gpu, monkeypatch, shape, calls, nq, pow_bits, logup):
    """The caller hands its trace over: streamed, the trace's coefficient arrays end up IN the caller's buffer (no tcoef), with LogUp
    via the permutation buffer while the permutation columns are computed from the trace's values. Same words as the oracle for 2, 4
    and 8 sub-cosets, interpreter and specialised kernels; what is left in the buffer is the H-scaled bit-reversed coefficient arrays,
    and pw_trace_from_coefficients turns it back into the trace exactly. Resident (0 sub-cosets): the trace is left alone."""
    torch, abi, prover = gpu
    flat, W, log_h, bc, spans, it = _synthetic(shape, calls, seed=21)
    want = sm.prove_logup(flat, W, log_h, bc, spans, *it, num_queries=nq, pow_bits=pow_bits) if logup else \
        sm.prove(flat, W, log_h, bc, spans, num_queries=nq, pow_bits=pow_bits)
    H = 1 << log_h
    d_ref = to_dev(torch, flat)
    d_c = torch.empty(W * H, dtype=torch.int32, device="cuda")
    d_l = torch.empty(W * 2 * H, dtype=torch.int32, device="cuda")
    abi.check(prover.lib.pw_lde_batch(d_ref.data_ptr(), W, log_h, d_c.data_ptr(), d_l.data_ptr()), "pw_lde_batch")
    torch.cuda.synchronize()
    for log_blocks in (0, 1, 2, 3):
        for jit in (False, True):
            got, left, restored, state = _prove_consuming(torch, prover, monkeypatch, flat, W, log_h, bc, spans, it if logup else None, nq, pow_bits,
                                                          log_blocks, jit)
            assert len(got) == len(want) and (got == want).all(), \
                f"blocks 2^{log_blocks} jit={jit}: first differing word {int(np.argmax(got != want))} of {len(want)}"
            streamed = min(log_blocks, max(log_h - 1, 0)) > 0
            assert torch.equal(left, d_c if streamed else d_ref)
            assert torch.equal(restored, d_ref)


@pytest.mark.parametrize("shape,log_h,log_blocks,jit", [("C3", 12, 1, True), ("C3", 12, 2, False), ("C2", 14, 1, True)])
def test_baseline_shapes_consuming_with_logup(gpu, monkeypatch, shape, log_h, log_blocks, jit):
    """configs[2]'s mode of round 5 — the C3 shape with its 2 314 interactions, trace handed over, TWO sub-cosets, specialised
    kernels with permutation panels — at 2^12 rows against sm.prove_logup; both verifiers accept."""
    torch, abi, prover = gpu
    (flat, W, lh, bc, spans, it), want = _baseline_case(shape, log_h)
    got, left, restored, state = _prove_consuming(torch, prover, monkeypatch, flat, W, log_h, bc, spans, it, 6, 4, log_blocks, jit)
    assert (state == 1) if jit else (state in (0, -1))
    assert len(got) == len(want) and (got == want).all(), f"first differing word {int(np.argmax(got != want))} of {len(want)}"
    assert torch.equal(restored, to_dev(torch, flat))
    assert prover.verify_logup(got, W, log_h, bc, spans, it, num_queries=6, pow_bits=4)[0] == 0
    assert sm.verify_logup(got, W, log_h, bc, spans, *it, num_queries=6, pow_bits=4) == 0


def test_consuming_tall_trace(gpu, monkeypatch):
    """2^16 rows (strided stage groups in the restore transform and the sub-coset transforms), 2 sub-cosets, LogUp."""
    torch, abi, prover = gpu
    flat, W, log_h, bc, spans, it = _synthetic("T1", 40000, seed=3)
    want = sm.prove_logup(flat, W, log_h, bc, spans, *it, num_queries=12, pow_bits=0)
    got, left, restored, _ = _prove_consuming(torch, prover, monkeypatch, flat, W, log_h, bc, spans, it, 12, 0, 1, True)
    assert (got == want).all() and torch.equal(restored, to_dev(torch, flat))


def test_a_trace_at_an_odd_word_offset(gpu, monkeypatch):
    """from 2^16 rows on the resident DEEP numerator is combined with 8-byte loads of the caller's trace; a trace that is
    only 4-byte aligned (a view into a larger tensor) takes the accumulating kernel instead — same words. The consuming call wants
    16 bytes and says so."""
    torch, abi, prover = gpu
    flat, W, log_h, bc, spans, it = _synthetic("T1", 40000, seed=3)
    assert log_h == 16
    monkeypatch.setenv("POWDR_STREAM_LOG_BLOCKS", "0")
    d_t = to_dev(torch, flat)
    big = torch.empty(d_t.numel() + 8, dtype=torch.int32, device="cuda")
    assert big.data_ptr() % 16 == 0
    odd = big[1:1 + d_t.numel()]
    odd.copy_(d_t)
    for inter in (None, it):
        pr = prover.Prover(W, bc, spans, num_queries=7, pow_bits=0, interactions=inter)
        a = pr.prove(d_t.data_ptr(), log_h)
        b = pr.prove(odd.data_ptr(), log_h)
        assert (a == b).all()
        with pytest.raises(Exception):
            pr.prove(odd.data_ptr(), log_h, consume=True)
        pr.close()
    # the public transforms that stage whole tiles with 16-byte loads refuse a 4-byte-aligned view as well
    H = 1 << log_h
    scale = torch.empty(H, dtype=torch.int32, device="cuda")
    out = torch.empty(W * H, dtype=torch.int32, device="cuda")
    assert prover.lib.pw_lde_subcoset(odd.data_ptr(), W, log_h, 1, 0, scale.data_ptr(), out.data_ptr()) == 1  # hipErrorInvalidValue
    assert prover.lib.pw_lde_subcoset(d_t.data_ptr(), W, log_h, 1, 0, scale.data_ptr(), out.data_ptr()) == 0
    scratch = torch.empty(8192, dtype=torch.int32, device="cuda")
    assert prover.lib.pw_trace_from_coefficients(odd.data_ptr(), W, log_h, scratch.data_ptr()) == 1
    torch.cuda.synchronize()
