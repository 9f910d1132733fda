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
