"""The honest multi-AIR segment of the C4 / C5 bench legs (powdr_amd/segment_workload.py): generated traces,
real constraints on the instruction AIRs, periphery AIRs from the histograms. At capped heights: every constraint holds, the
segment proof is accepted by the product's verifier and by the oracle's, its words equal sm.prove_segment's on the same traces,
and the lookup buses balance (APC + instruction AIR sends = periphery AIR receives)."""
import numpy as np
import pytest

from oracle import apc_model as om
from oracle import stark_model as sm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from powdr_amd import segment_workload as sw

    return torch, sw


def _host_airs(seg):
    airs = []
    for a in seg.airs:
        flat = om.from_monty(a["trace"].cpu().numpy().view(np.uint32))
        airs.append((flat, a["width"], a["log_h"], a["cons"][0], a["cons"][1], a["inter"]))
    return airs


@pytest.mark.parametrize("kind,cap,n_apc", [("C4", 10, 3), ("C5", 11, 4)])
def test_honest_segment_verifies_balances_and_matches_the_oracle(gpu, kind, cap, n_apc):
    torch, sw = gpu
    seg = sw.HonestSegment(kind, max_log_height=cap, seed=1, queries=5, pow_bits=3, logup=True, max_apc_airs=n_apc)
    roles = [a["role"] for a in seg.airs]
    assert roles.count("apc") == n_apc and roles.count("instruction") == 13 and roles.count("periphery") == 3
    seg.generate_traces()
    torch.cuda.synchronize()
    assert int(seg.per.var_hist.sum()) > 0 and int(seg.per.bitwise_hist.sum()) > 0 and int(seg.per.tuple_hist.sum()) > 0
    assert seg.check_constraints() == 0  # the device's mock prover: all constraints of all 19 AIRs on all rows
    proof = seg.prove(copy=True)
    assert seg.verify(proof) == 0
    # the same words as the oracle's segment prover on the same traces; the oracle's verifier accepts them
    airs = _host_airs(seg)
    want = sm.prove_segment(airs, num_queries=5, pow_bits=3, logup=True)
    assert len(proof) == len(want) and (proof == want).all(), f"first differing word {int(np.argmax(proof != want))} of {len(want)}"
    assert sm.verify_segment(proof, airs, 5, 3, True)[0] == 0
    # lookup buses: senders and receivers cancel
    rc, total = seg.balance_witness()
    assert rc == 0 and (np.asarray(total) == 0).all()
    # ... and stop cancelling when one histogram bin is off by one
    seg.per.var_hist[(1 << 12) + 5] += 1
    from powdr_amd import abi

    abi.check(abi.lib.powdr_periphery_var_range_trace(seg.per.var_hist.data_ptr(), seg.per.var_hist.numel(), seg.per_traces["var_range"].data_ptr()), "trace")
    assert seg.balance_witness()[0] == 14
    seg.close()


def test_a_second_generation_gives_the_same_proof(gpu):
    """generate_traces() is what the bench repeats inside its timed region: it must rebuild everything (histograms zeroed, every
    trace rewritten) — two generations, two identical proofs."""
    torch, sw = gpu
    seg = sw.HonestSegment("C4", max_log_height=9, seed=2, queries=4, pow_bits=0, logup=True, max_apc_airs=2)
    seg.generate_traces()
    p1 = seg.prove(copy=True)
    for a in seg.airs:
        if a["role"] != "instruction":  # (instruction-AIR padding rows are written once, at allocation)
            a["trace"].fill_(12345)
    seg.generate_traces()
    p2 = seg.prove(copy=True)
    assert (p1 == p2).all() and seg.verify(p2) == 0
    seg.close()


def test_staged_inputs_give_other_segments_of_the_same_shape(gpu):
    """HonestSegment.stage_inputs(u): the inputs of segment u — dummy traces behind the APC AIRs (libpowdr_synth.so, one
    write-only launch per matrix), the instruction AIRs' records — replace the resident ones. Same AIRs, other rows: the traces and the
    commitment change, every constraint still holds, the proof verifies and equals the oracle's on the traces read back, the lookup
    buses balance (every bounded cell was drawn below its bound), and the same index gives the same segment again."""
    torch, sw = gpu
    seg = sw.HonestSegment("C4", max_log_height=10, seed=1, queries=5, pow_bits=3, logup=True, max_apc_airs=3)
    hdr = 5 + 4 * len(seg.airs)
    seen, traces = {}, {}
    for u in (0, 1, 5, 1):
        seg.stage_inputs(u)
        seg.generate_traces()
        torch.cuda.synchronize()
        assert seg.check_constraints() == 0
        proof = seg.prove(copy=True)
        assert seg.verify(proof) == 0
        root = tuple(int(x) for x in proof[hdr:hdr + 8])
        snap = [a["trace"].clone() for a in seg.airs]
        if u in seen:
            assert seen[u] == root and all(torch.equal(x, y) for x, y in zip(traces[u], snap))
            continue
        seen[u], traces[u] = root, snap
        if u == 5:
            airs = _host_airs(seg)
            want = sm.prove_segment(airs, num_queries=5, pow_bits=3, logup=True)
            assert len(proof) == len(want) and (proof == want).all()
            rc, total = seg.balance_witness()
            assert rc == 0 and (np.asarray(total) == 0).all()
    assert len(set(seen.values())) == 3
    # APC and instruction traces differ between segments (periphery tables differ through their multiplicities)
    for i, a in enumerate(seg.airs):
        if a["role"] != "periphery":
            assert not torch.equal(traces[0][i], traces[1][i]), a["name"]
    seg.close()


def test_segments_with_their_own_trace_heights(gpu, monkeypatch):
    """the reference's segments each carry their own `trace_heights` (openvm/src/trace_generation.rs:113-131). One
    resident HonestSegment is re-shaped per unit (draw_shape: every chip at its cap / one chip at its cap and the others log-uniform
    over two octaves / the execution's tail at <= 1/8), its inputs staged for that shape, traces generated, ONE segment proof — for
    three differently-shaped segments: every constraint holds, both verifiers accept, the words equal sm.prove_segment's on the traces
    read back, the lookup buses balance; going back to an earlier shape gives that segment's proof again (plan caches keyed by
    height, buffers that only grow, padding rows rewritten)."""
    torch, sw = gpu
    from powdr_amd import prover

    seg = sw.HonestSegment("C4", max_log_height=10, seed=3, queries=5, pow_bits=3, logup=True, max_apc_airs=3)
    n_segments = 6
    shapes = {u: seg.draw_shape(u, n_segments) for u in range(n_segments)}
    caps = [wl["max_calls"] for wl in seg.apcs]
    assert shapes[0]["apc_calls"] == caps and shapes[0]["instr_calls"] == seg.max_calls
    assert all(c <= cap // 8 for c, cap in zip(shapes[n_segments - 1]["apc_calls"], caps)) and shapes[n_segments - 1]["instr_calls"] <= max(1, seg.max_calls // 8)
    for u in range(1, n_segments - 1):
        f = [c / cap for c, cap in zip(shapes[u]["apc_calls"], caps)] + [shapes[u]["instr_calls"] / seg.max_calls]
        assert max(f) == 1.0 and min(f) >= 0.24
    roots, heights, proofs = {}, {}, {}
    for u in (0, 2, n_segments - 1, 0, 2):
        seg.stage_inputs(u, shapes[u])
        seg.generate_traces()
        torch.cuda.synchronize()
        assert seg.check_constraints() == 0, u
        proof = seg.prove(copy=True)
        assert seg.verify(proof) == 0, u
        hdr = 5 + 4 * len(seg.airs)
        root = tuple(int(x) for x in proof[hdr:hdr + 8])
        if u in roots:
            assert roots[u] == root and heights[u] == seg.heights() and (proofs[u] == proof).all()
            continue
        roots[u], heights[u], proofs[u] = root, seg.heights(), proof
        airs = _host_airs_shaped(seg)
        want = sm.prove_segment(airs, num_queries=5, pow_bits=3, logup=True)
        assert len(proof) == len(want) and (proof == want).all(), f"segment {u}: first differing word {int(np.argmax(proof != want))} of {len(want)}"
        assert sm.verify_segment(proof, airs, 5, 3, True)[0] == 0
        if u != 2:  # (the capped segment and the tail: the lookup buses balance at every shape; the middle one is covered by the budget test)
            rc, total = seg.balance_witness()
            assert rc == 0 and (np.asarray(total) == 0).all(), u
        # the same segment with its traces handed over (pw_prove_segment_consuming; nothing is streamed at this size): same words
        assert (seg.prove(copy=True, hand_over=True) == proof).all()
    assert len(set(roots.values())) == 3 and len({tuple(h) for h in heights.values()}) == 3
    # the tail is short: every APC and instruction AIR at most 1/8 of the capped segment's rows
    for h0, ht, a in zip(heights[0], heights[n_segments - 1], seg.airs):
        if a["role"] == "apc":
            assert ht <= h0 - 3
        elif a["role"] == "instruction":  # (four block executions at this cap: the tail has one)
            assert ht < h0 or h0 <= 2
    seg.close()


def _host_airs_shaped(seg):
    """like _host_airs, for a re-shaped segment: a trace is the prefix of its buffer"""
    airs = []
    for a in seg.airs:
        n = a["width"] << a["log_h"]
        flat = om.from_monty(a["trace"][:n].cpu().numpy().view(np.uint32))
        airs.append((flat, a["width"], a["log_h"], a["cons"][0], a["cons"][1], a["inter"]))
    return airs


def test_a_budget_makes_the_capped_segment_stream_and_the_words_stay(gpu, monkeypatch):
    """The bench's `multi_segment` leg under a device budget of 0.8 x the capped segment's resident plan: segment 0 (every chip at its
    cap) streams its largest AIR with the trace handed over, the shorter segments stay resident — decisions made per segment, inside
    the run — and every proof equals the one made without a budget."""
    torch, sw = gpu
    from powdr_amd import prover

    monkeypatch.setenv("POWDR_STREAM_MIN_LOG_HEIGHT", "9")
    seg = sw.HonestSegment("C4", max_log_height=11, seed=4, queries=4, pow_bits=0, logup=True, max_apc_airs=3)
    shapes = {u: seg.draw_shape(u, 4) for u in range(4)}
    plain = {}
    for u in range(4):
        seg.stage_inputs(u, shapes[u])
        seg.generate_traces()
        plain[u] = seg.prove(copy=True)
        if u == 0:
            resident = prover.segment_last_plan()[0]
            assert resident > 0 and all(b == 0 for b, _ in prover.segment_last_modes())
    try:
        prover.set_device_budget(int(0.8 * resident))
        streamed_in = {}
        for u in (3, 0, 1, 0, 2):
            seg.stage_inputs(u, shapes[u])
            seg.generate_traces()
            got = seg.prove(copy=True, hand_over=True)
            assert (got == plain[u]).all(), u
            modes = prover.segment_last_modes()
            streamed_in[u] = [a["name"] for a, (b, eaten) in zip(seg.airs, modes) if b]
            assert all(eaten == (b > 0 and a["role"] != "instruction") for a, (b, eaten) in zip(seg.airs, modes))
        assert streamed_in[0] and streamed_in[0][0] == "apc0" and streamed_in[3] == []
    finally:
        prover.set_device_budget(0)
    seg.close()
