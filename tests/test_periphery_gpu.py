"""Periphery chips' traces from the lookup histograms, and the property that ties trace generation (a3: the
bus -> histogram replay) to the proof (a6: the same bus interactions as LogUp terms): in a segment made of the APC
AIR and the periphery AIRs, the cumulative bus sums of all proofs add up to zero."""
import numpy as np
import pytest

from oracle import apc_model as om
from oracle import stark_model as sm
from powdr_amd import synth

pytestmark = pytest.mark.gpu
P = om.P
NO_CONS = (np.zeros(0, np.uint32), np.zeros((0, 2), np.uint32))


@pytest.fixture(scope="module")
def gpu():
    import torch
    from powdr_amd import abi, periphery, prover, sharding, tracegen

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU (run with -m gpu on the GPU box)")
    return torch, periphery, prover, sharding, tracegen


def to_dev(torch, a_canonical):
    return torch.from_numpy(om.to_monty(np.ascontiguousarray(a_canonical, dtype=np.uint32)).view(np.int32)).cuda()


def from_dev(t):
    return om.from_monty(t.cpu().numpy().view(np.uint32))


def test_periphery_traces_match_restatement(gpu):
    torch, periphery, *_ = gpu
    rng = np.random.default_rng(1)
    for bins in (1, 2, 16, 1 << 18):
        h = rng.integers(0, 1 << 32, bins, dtype=np.uint64).astype(np.uint32)  # counters may exceed p: reduced mod p
        got = from_dev(periphery.var_range_trace(torch.from_numpy(h.view(np.int32)).cuda())).reshape(3, bins)
        assert (got == om.var_range_trace(h)).all()
        i = np.arange(bins)
        assert (((1 << got[1].astype(np.int64)) + got[0] - 1) == i).all()  # apc_apply_bus.cu:74
    for sz in ((256, 2048), (4, 8), (1, 1)):
        h = rng.integers(0, 1000, sz[0] * sz[1]).astype(np.uint32)
        got = from_dev(periphery.tuple2_trace(torch.from_numpy(h.view(np.int32)).cuda(), sz)).reshape(3, -1)
        assert (got == om.tuple2_trace(h, *sz)).all()
    h = rng.integers(0, 5, 2 * 65536).astype(np.uint32)
    got = from_dev(periphery.bitwise_trace(torch.from_numpy(h.view(np.int32)).cuda())).reshape(5, 65536)
    assert (got == om.bitwise_trace(h)).all()
    with pytest.raises(RuntimeError):
        periphery.var_range_trace(torch.zeros(12, dtype=torch.int32, device="cuda"))  # not a power of two


def ext_sum(sums):
    acc = np.zeros(4, np.uint64)
    for s in sums:
        acc = (acc + s.astype(np.uint64)) % P
    return acc


def prove_segment(gpu, airs, log_hs, nq=8):
    """airs: [(device trace tensor, width, cons, interactions)] -> per-AIR cumulative sums (all proofs verified
    against the seed recomputed from the trace roots the verifier returns)"""
    torch, periphery, prover, sharding, _ = gpu
    provers = [prover.Prover(w, *cons, num_queries=nq, interactions=it) for (_, w, cons, it) in airs]
    roots = [pr.trace_root(t.data_ptr(), lh) for pr, (t, *_), lh in zip(provers, airs, log_hs)]
    seed = sharding.commitment_digest(np.array(roots))
    sums, seen = [], []
    for pr, (t, w, cons, it), lh in zip(provers, airs, log_hs):
        pr.set_bus_seed(seed)
        proof = pr.prove(t.data_ptr(), lh)
        rc, S, root = prover.verify_logup(proof, w, lh, *cons, it, num_queries=nq, bus_seed=seed, with_root=True)
        assert rc == 0
        sums.append(S)
        seen.append(root)
        pr.close()
    assert (sharding.commitment_digest(np.array(seen)) == seed).all()
    return sums


@pytest.mark.parametrize("shape,calls,seed", [("T0", 50, 1), ("T1", 1000, 2), ("T1", 4096, 3), ("C1", 2000, 4)])
def test_histograms_balance_the_lookup_buses(gpu, shape, calls, seed):
    """APC trace generation fills the var-range and tuple histograms (a3); the APC AIR sends the same lookups
    as LogUp terms (a6); the periphery AIRs receive them with the histogram counts as multiplicities. The three
    proofs' cumulative sums cancel — and stop cancelling when a single histogram bin is off by one."""
    torch, periphery, prover, sharding, tg = gpu
    from tests.test_oracle_apc import run_oracle_gpu_convention
    from tests.test_tracegen_gpu import run_gpu

    s = synth.generate(shape, seed=seed)
    apc, idx, want, hist, (bufs, dims, gt, order) = run_oracle_gpu_convention(s, calls, seed=seed)
    W, H = want.shape
    log_h = H.bit_length() - 1
    tgpu = (torch, None, tg)
    out, per = run_gpu(tgpu, W, H, calls, bufs, dims, gt.air_names, gt.row_block_size, gt.subs, om.compile_derived(apc, idx, H),
                       om.compile_bus(apc, idx, H))
    assert (per.var_hist.cpu().numpy().view(np.uint32) == hist["var"]).all()
    cons = sm.compile_constraints(apc, idx)
    sends = periphery.select_buses(sm.compile_interactions(apc, idx), {per.var_bus, per.tuple_bus})
    assert len(sends[0]) > 0
    var_t = periphery.var_range_trace(per.var_hist)
    tup_t = periphery.tuple2_trace(per.tuple_hist, per.tuple_sizes)
    airs = [(out.buf, W, cons, sends), (var_t, 3, NO_CONS, periphery.var_range_interactions(per.var_bus)),
            (tup_t, 3, NO_CONS, periphery.tuple2_interactions(per.tuple_bus))]
    log_hs = [log_h, per.var_hist.numel().bit_length() - 1, per.tuple_hist.numel().bit_length() - 1]
    sums = prove_segment(gpu, airs, log_hs)
    assert sums[0].any() and sums[1].any()
    assert (ext_sum(sums) == 0).all()
    # one lookup lost in the histogram: the bus no longer balances
    bin_ = int(np.argmax(hist["var"] != 0))
    per.var_hist[bin_] -= 1
    airs[1] = (periphery.var_range_trace(per.var_hist), 3, NO_CONS, periphery.var_range_interactions(per.var_bus))
    assert (ext_sum(prove_segment(gpu, airs, log_hs)) != 0).any()


def test_bitwise_bus_balances(gpu):
    """A sender with honest bitwise lookups (z = x ^ y for the xor operation, z = 0 for the range check) against the
    bitwise periphery AIR built from the histogram the replay kernel filled."""
    torch, periphery, prover, sharding, tg = gpu
    rng = np.random.default_rng(5)
    log_h, H = 12, 1 << 12
    x, y = rng.integers(0, 256, H).astype(np.uint32), rng.integers(0, 256, H).astype(np.uint32)
    sel = rng.integers(0, 2, H).astype(np.uint32)
    m = rng.integers(0, 3, H).astype(np.uint32)
    trace = np.stack([x, y, np.where(sel == 1, x ^ y, 0).astype(np.uint32), sel, m])
    PA = om.OP_PUSH_APC
    # the reference's device encoding for the replay kernel (operands col*H), and column operands for the prover
    spans = np.array([[0, 2], [2, 2], [4, 2], [6, 2], [8, 2]], np.uint32)  # mult, x, y, z, sel
    def prog(stride):
        return np.array([PA, 4 * stride, PA, 0, PA, 1 * stride, PA, 2 * stride, PA, 3 * stride], np.uint32)
    inter = np.array([[6, 4, 0]], np.uint32)
    out = tg.DeviceMatrix.zeros(H, 5)
    out.buf.copy_(to_dev(torch, trace.reshape(-1)))
    per = tg.Periphery.fresh()
    keep = tg.apc_apply_bus(out, H, prog(H), inter, spans, per)
    torch.cuda.synchronize()
    want = np.zeros(2 * 65536, np.uint32)
    np.add.at(want, sel.astype(np.int64) * 65536 + x * 256 + y, m)
    assert (per.bitwise_hist.cpu().numpy().view(np.uint32) == want).all()
    airs = [(out.buf, 5, NO_CONS, (inter, spans, prog(1))),
            (periphery.bitwise_trace(per.bitwise_hist), 5, NO_CONS, periphery.bitwise_interactions(6))]
    sums = prove_segment(gpu, airs, [log_h, 16])
    assert sums[0].any() and (ext_sum(sums) == 0).all()
    del keep


@pytest.mark.parametrize("workers", [1, 3])
def test_prove_segment_matches_the_per_air_flow(gpu, workers):
    """pw_prove_segment (one call, host threads + streams, phase-1 commitments reused in phase 2) produces exactly the
    proofs of the explicit trace_root -> seed -> prove flow, and pw_verify_segment accepts them as a balanced segment."""
    torch, periphery, prover, sharding, tg = gpu
    from tests.test_oracle_apc import run_oracle_gpu_convention
    from tests.test_tracegen_gpu import run_gpu

    s = synth.generate("T1", seed=6)
    apc, idx, want, hist, (bufs, dims, gt, order) = run_oracle_gpu_convention(s, 3000, seed=6)
    W, H = want.shape
    log_h = H.bit_length() - 1
    out, per = run_gpu((torch, None, tg), W, H, 3000, bufs, dims, gt.air_names, gt.row_block_size, gt.subs,
                       om.compile_derived(apc, idx, H), om.compile_bus(apc, idx, H))
    cons = sm.compile_constraints(apc, idx)
    sends = periphery.select_buses(sm.compile_interactions(apc, idx), {per.var_bus, per.tuple_bus})
    airs = [(out.buf, W, cons, sends, log_h),
            (periphery.var_range_trace(per.var_hist), 3, NO_CONS, periphery.var_range_interactions(per.var_bus), per.var_hist.numel().bit_length() - 1),
            (periphery.tuple2_trace(per.tuple_hist, per.tuple_sizes), 3, NO_CONS, periphery.tuple2_interactions(per.tuple_bus),
             per.tuple_hist.numel().bit_length() - 1)]
    nq = 7
    provers = [prover.Prover(w, *c, num_queries=nq, interactions=it) for (_, w, c, it, _) in airs]
    # explicit flow
    roots = [pr.trace_root(t.data_ptr(), lh) for pr, (t, _, _, _, lh) in zip(provers, airs)]
    seed = sharding.commitment_digest(np.array(roots))
    ref = []
    for pr, (t, _, _, _, lh) in zip(provers, airs):
        pr.set_bus_seed(seed)
        ref.append(pr.prove(t.data_ptr(), lh))
    # one call
    got, seed2 = prover.prove_airs([(pr, t.data_ptr(), lh) for pr, (t, _, _, _, lh) in zip(provers, airs)], shared_bus_seed=True,
                                      n_workers=workers)
    assert (seed2 == seed).all()
    for a, b in zip(got, ref):
        assert len(a) == len(b) and (a == b).all()
    descs = [(w, lh, *c, it) for (_, w, c, it, lh) in airs]
    rc, total = prover.verify_airs(descs, got, num_queries=nq, shared_bus_seed=True, check_balance=True)
    assert rc == 0 and (total == 0).all()
    # the rank-sharded orchestration (here with one rank owning every AIR) gives the same bytes again
    commit, prove = sharding.gpu_segment_callables(provers, [t.data_ptr() for t, *_ in airs], [lh for *_, lh in airs])
    seed3, proofs3 = sharding.prove_segment_sharded([0, 1, 2], 3, commit, prove)
    assert (seed3 == seed).all() and all((proofs3[u] == ref[u]).all() for u in range(3))
    for pr in provers:
        pr.close()


def test_prove_segment_many_airs_of_mixed_heights(gpu):
    """A reth-shaped segment: many AIRs, heights 2^4..2^15, constraints only; four workers produce
    the same bytes as proving the AIRs one by one, whatever order the workers finish in."""
    torch, periphery, prover, sharding, tg = gpu
    rng = np.random.default_rng(8)
    PA, PC = om.OP_PUSH_APC, om.OP_PUSH_CONST
    airs, provers = [], []
    for k in range(14):
        log_h = int(rng.integers(4, 16))
        W = int(rng.integers(2, 40))
        a, b = (int(x) for x in rng.integers(0, W, 2))
        bc = np.array([PA, a, PA, b, om.OP_MUL, PC, int(rng.integers(0, P)), om.OP_SUB], np.uint32)  # unsatisfied: parity does not care
        spans = np.array([[0, len(bc)]], np.uint32)
        t = to_dev(torch, rng.integers(0, P, W << log_h, dtype=np.uint32))
        airs.append((t, W, log_h, bc, spans))
        provers.append(prover.Prover(W, bc, spans, num_queries=6, pow_bits=3))
    ref = [pr.prove(t.data_ptr(), lh) for pr, (t, _, lh, _, _) in zip(provers, airs)]
    for workers in (4, 0, 1):
        got, _ = prover.prove_airs([(pr, t.data_ptr(), lh) for pr, (t, _, lh, _, _) in zip(provers, airs)], n_workers=workers)
        for a, b in zip(got, ref):
            assert len(a) == len(b) and (a == b).all()
    for k in (0, 5, 13):
        t, W, lh, bc, spans = airs[k]
        if lh <= 10:
            assert (ref[k] == sm.prove(from_dev(t), W, lh, bc, spans, num_queries=6, pow_bits=3)).all()
    for pr in provers:
        pr.close()
