"""pw-stark v1: ONE proof per segment (oracle/stark_segment.inc, csrc/segment_prover.hip, pw_verify_segment).

The reference makes one engine call per segment with all chips' traces (/root/reference/openvm/src/trace_generation.rs:
136-139, openvm-riscv/src/lib.rs:327-341). CPU tests: the oracle prover against the oracle verifier and the product's
host verifier (independent arithmetic), failure codes, bus balance, a committed golden digest. GPU tests: the HIP segment
prover's words equal the oracle's for mixed heights, with and without LogUp."""
import ctypes
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import apc_model as om
from oracle import stark_model as sm
from powdr_amd import synth

P = om.P
GOLDEN = Path(__file__).parent / "golden" / "pw_stark_segment_T.json"


def synthetic_airs(spec, seed0=1):
    """[(shape, calls)] -> oracle air tuples (trace, W, log_h, cons_bc, cons_spans, interactions)."""
    from tests.test_oracle_apc import run_oracle_gpu_convention

    airs = []
    for k, (shape, calls) in enumerate(spec):
        s = synth.generate(shape, seed=seed0 + k)
        apc, idx, trace, _, _ = run_oracle_gpu_convention(s, calls, seed=seed0 + k)
        W, H = trace.shape
        airs.append((np.ascontiguousarray(trace).reshape(-1), W, H.bit_length() - 1, *sm.compile_constraints(apc, idx),
                     sm.compile_interactions(apc, idx)))
    return airs


def descs_of(airs):
    return [(a[1], a[2], a[3], a[4], a[5]) for a in airs]


SPEC = [("T0", 30), ("T1", 200), ("T0", 5), ("T1", 64), ("T0", 30)]  # LDE heights 64, 512, 16, 128, 64


@pytest.mark.parametrize("logup", [False, True])
def test_oracle_segment_proof_and_both_verifiers(logup):
    from powdr_amd import prover

    airs = synthetic_airs(SPEC)
    pf = sm.prove_segment(airs, num_queries=5, pow_bits=4, logup=logup)
    assert sm.verify_segment(pf, airs, 5, 4, logup)[0] == 0
    rc, total = prover.verify_segment(descs_of(airs), pf, 5, 4, logup)
    assert rc == 0 and (total == sm.verify_segment(pf, airs, 5, 4, logup)[1]).all()
    # one FRI and one query phase for the whole segment: smaller than the independent proofs
    separate = sum(len(sm.prove_logup(a[0], a[1], a[2], a[3], a[4], *a[5], num_queries=5, pow_bits=4)) if logup else
                   len(sm.prove(a[0], a[1], a[2], a[3], a[4], num_queries=5, pow_bits=4)) for a in airs)
    assert len(pf) < 0.8 * separate
    # every tampered word is rejected, by both verifiers, with the same code
    rng = np.random.default_rng(7)
    codes = set()
    for pos in list(range(0, 30)) + [int(x) for x in rng.integers(30, len(pf), 60)]:
        bad = pf.copy()
        bad[pos] = (int(bad[pos]) + 1) % P
        c1 = sm.verify_segment(bad, airs, 5, 4, logup)[0]
        c2 = prover.verify_segment(descs_of(airs), bad, 5, 4, logup)[0]
        assert c1 == c2 != 0, (pos, c1, c2)
        codes.add(c1 & 0xFF)
    assert {1, 2, 5, 7} <= codes
    # wrong statement: heights / order / flag
    assert prover.verify_segment(descs_of(airs)[::-1], pf, 5, 4, logup)[0] == 1
    assert prover.verify_segment(descs_of(airs), pf, 5, 4, not logup)[0] == 1
    assert prover.verify_segment(descs_of(airs), pf[:-3], 5, 4, logup)[0] in (9, 10, 7)
    big = pf.copy()
    big[len(big) // 2] = int(big[len(big) // 2]) + P if int(big[len(big) // 2]) + P < 2 ** 32 else big[len(big) // 2]
    if (big != pf).any():
        assert prover.verify_segment(descs_of(airs), big, 5, 4, logup)[0] == 13 == sm.verify_segment(big, airs, 5, 4, logup)[0]


def test_segment_with_a_violated_constraint_is_rejected():
    from powdr_amd import prover

    airs = synthetic_airs([("T0", 20), ("T1", 100)])
    t = airs[1][0].copy()
    s = synth.generate("T1", seed=2)
    apc = om.load_apc(s.doc)
    idx = apc.poly_id_to_index()
    valid_col = idx[[q for q, k in s.kinds.items() if k[0] == "valid"][0]]
    H = 1 << airs[1][2]
    t[valid_col * H + 3] = 2  # is_valid = 2 breaks is_valid * (is_valid - 1)
    bad_airs = [airs[0], (t,) + airs[1][1:]]
    pf = sm.prove_segment(bad_airs, num_queries=4, logup=False)
    assert sm.verify_segment(pf, bad_airs, 4, 0, False)[0] == (2 << 8) | 2
    assert prover.verify_segment(descs_of(bad_airs), pf, 4, 0, False)[0] == (2 << 8) | 2


@pytest.mark.parametrize("log_hs", [(4, 4), (6, 3), (3, 9)])
def test_segment_bus_balance(log_hs):
    """Two AIRs on one bus (sends / permuted receives) of DIFFERENT heights in one segment proof: the cumulative sums
    cancel (check_balance passes); one changed tuple and the segment is rejected with code 14 — by both verifiers."""
    from powdr_amd import prover
    from tests.test_oracle_stark import balanced_bus_pair_mixed

    no_cons = (np.zeros(0, np.uint32), np.zeros((0, 2), np.uint32))
    pair = balanced_bus_pair_mixed(log_hs[0], log_hs[1], seed=sum(log_hs))
    airs = [(t.reshape(-1), 3, lh, *no_cons, it) for (t, it), lh in zip(pair, log_hs)]
    pf = sm.prove_segment(airs, num_queries=4, logup=True)
    rc, total = sm.verify_segment(pf, airs, 4, 0, True, check_balance=True)
    assert rc == 0 and (total == 0).all()
    assert prover.verify_segment(descs_of(airs), pf, 4, 0, True, check_balance=True)[0] == 0
    t = airs[0][0].copy()
    t[0] = (int(t[0]) + 1) % P  # one sent tuple changes
    off = [(t,) + airs[0][1:], airs[1]]
    pf2 = sm.prove_segment(off, num_queries=4, logup=True)
    assert sm.verify_segment(pf2, off, 4, 0, True, check_balance=False)[0] == 0
    assert sm.verify_segment(pf2, off, 4, 0, True, check_balance=True)[0] == 14
    assert prover.verify_segment(descs_of(off), pf2, 4, 0, True, check_balance=True)[0] == 14


def test_segment_golden_digest():
    """The oracle's segment proofs of a fixed synthetic segment are pinned by SHA-256 (tests/golden/make_segment_golden.py):
    a change of the protocol, of the transcript or of any kernel-visible convention shows up here first."""
    g = json.loads(GOLDEN.read_text())
    airs = synthetic_airs([tuple(x) for x in g["spec"]], seed0=g["seed0"])
    for key, logup in (("sha256_v1", False), ("sha256_v1_logup", True)):
        pf = sm.prove_segment(airs, num_queries=g["num_queries"], pow_bits=g["pow_bits"], logup=logup)
        assert hashlib.sha256(pf.astype("<u4").tobytes()).hexdigest() == g[key], key
        assert len(pf) == g["words" + ("_logup" if logup else "")]


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def gpu():
    import torch
    from powdr_amd import abi, prover

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU (run with -m gpu on the GPU box)")
    return torch, abi, prover


def to_dev(torch, a):
    return torch.from_numpy(om.to_monty(np.ascontiguousarray(a, dtype=np.uint32)).view(np.int32)).cuda()


def hip_segment(gpu, airs, nq, pow_bits, logup):
    torch, abi, prover = gpu
    provers = [prover.Prover(a[1], a[3], a[4], num_queries=nq, pow_bits=pow_bits, interactions=a[5] if logup else None) for a in airs]
    traces = [to_dev(torch, a[0]) for a in airs]
    pf = prover.prove_segment([(pr, t.data_ptr(), a[2]) for pr, t, a in zip(provers, traces, airs)], logup=logup)
    again = prover.prove_segment([(pr, t.data_ptr(), a[2]) for pr, t, a in zip(provers, traces, airs)], logup=logup)
    assert (pf == again).all()  # buffer reuse
    for pr in provers:
        pr.close()
    return pf


@pytest.mark.gpu
@pytest.mark.parametrize("spec,nq,pow_bits", [
    (SPEC, 5, 4),
    ([("T0", 2)], 3, 0),                                   # one AIR of 2 rows
    ([("T1", 100), ("T1", 100), ("T1", 100)], 4, 0),       # one height only: no roll-in
    ([("T0", 2), ("T0", 3), ("T0", 4), ("T0", 7), ("T0", 9), ("T0", 17), ("T0", 33), ("T0", 65), ("T1", 129), ("T1", 700), ("T1", 3000)], 6, 3),
    ([("C1", 600), ("T1", 5000), ("T0", 40), ("T1", 1000)], 8, 0),
])
@pytest.mark.parametrize("logup", [False, True])
def test_hip_segment_proof_bytes_match_oracle(gpu, spec, nq, pow_bits, logup):
    torch, abi, prover = gpu
    airs = synthetic_airs(spec, seed0=11)
    want = sm.prove_segment(airs, num_queries=nq, pow_bits=pow_bits, logup=logup)
    got = hip_segment(gpu, airs, nq, pow_bits, logup)
    assert len(got) == len(want)
    assert (got == want).all(), f"first differing word {int(np.argmax(got != want))} of {len(want)}"
    assert prover.verify_segment(descs_of(airs), got, nq, pow_bits, logup)[0] == 0
    assert sm.verify_segment(got, airs, nq, pow_bits, logup)[0] == 0


_WANT = {}


def _oracle_segment(spec, nq, pow_bits, logup, seed0=11):
    key = (tuple(spec), nq, pow_bits, logup, seed0)
    if key not in _WANT:
        airs = synthetic_airs(spec, seed0=seed0)
        _WANT[key] = (airs, sm.prove_segment(airs, num_queries=nq, pow_bits=pow_bits, logup=logup))
    return _WANT[key]


@pytest.mark.gpu
@pytest.mark.parametrize("spec,nq,pow_bits", [
    ([("T1", 1000), ("T0", 40), ("T1", 100)], 6, 3),                                # three heights, each AIR alone at its own: all streamed
    ([("C1", 600), ("T1", 5000), ("T0", 40), ("T1", 1000), ("T0", 33)], 8, 0),      # two pairs of AIRs share a height (resident); only the 2^13-row AIR is streamed
    ([("T1", 40000), ("T0", 64)], 5, 0),                                            # 2^16 rows: strided stage groups, query rows from the partial transform
])
@pytest.mark.parametrize("logup", [False, True])
@pytest.mark.parametrize("log_blocks,jit", [(1, "0"), (2, "1"), (3, "1")])
def test_hip_segment_proof_with_streamed_airs(gpu, monkeypatch, spec, nq, pow_bits, logup, log_blocks, jit):
    """Streamed AIRs INSIDE a segment proof: every AIR that is alone at its height is proven from coefficient arrays —
    row digests hashed sub-coset by sub-coset into its level of the mixed trees, quotient / DEEP / query rows from prover_stream.hpp —
    and the words still equal the oracle's."""
    torch, abi, prover = gpu
    if len(spec) == 2 and (log_blocks == 1 or (not logup and log_blocks != 2)):
        pytest.skip("the tall case runs once per kernel kind")
    airs, want = _oracle_segment(spec, nq, pow_bits, logup)  # (once per spec and proof kind: the sub-coset counts prove the same segment)
    monkeypatch.setenv("POWDR_STREAM_LOG_BLOCKS", str(log_blocks))
    monkeypatch.setenv("POWDR_JIT", jit)
    got = hip_segment(gpu, airs, nq, pow_bits, logup)
    assert len(got) == len(want)
    assert (got == want).all(), f"first differing word {int(np.argmax(got != want))} of {len(want)}"
    assert prover.verify_segment(descs_of(airs), got, nq, pow_bits, logup)[0] == 0
    if len(spec) == 2 and log_blocks == 2:
        # the same segment with every LDE resident: from 2^16 rows on an AIR's DEEP numerator is combined on the un-extended matrices
        # and extended as 4 (+ 4) columns (segment_prover.hip, like the one-AIR prover) — and accumulated over the LDE when told to
        monkeypatch.setenv("POWDR_STREAM_LOG_BLOCKS", "0")
        assert (hip_segment(gpu, airs, nq, pow_bits, logup) == want).all()
        monkeypatch.setenv("POWDR_DEEP_DIRECT", "1")
        assert (hip_segment(gpu, airs, nq, pow_bits, logup) == want).all()


@pytest.mark.gpu
@pytest.mark.parametrize("logup", [False, True])
@pytest.mark.parametrize("log_blocks,jit", [(1, "1"), (2, "0")])
def test_streamed_airs_that_share_a_height(gpu, monkeypatch, logup, log_blocks, jit):
    """A level's row digest is a sponge over the CONCATENATED rows of all matrices of that height. With streamed matrices in the
    concatenation the level is absorbed run by run (merkle.hip leaf_absorb_kernel), the sponge states parked between the runs — widths
    that are no multiples of the rate (13, 7, 30, ...) leave rate blocks open across matrix boundaries. Words == the oracle's."""
    torch, abi, prover = gpu
    rng = np.random.default_rng(17)
    shapes = [(13, 8), (7, 8), (30, 8), (5, 6), (11, 6), (9, 10), (3, 8), (20, 5)]
    airs = []
    for k, (w, lh) in enumerate(shapes):
        bc, sp, it = synth.random_air_programs(w, 3, 5, seed=100 + k)
        airs.append((rng.integers(0, P, size=w << lh, dtype=np.uint32), w, lh, bc, sp, it))
    want = sm.prove_segment(airs, num_queries=6, pow_bits=2, logup=logup)
    monkeypatch.setenv("POWDR_STREAM_LOG_BLOCKS", str(log_blocks))
    monkeypatch.setenv("POWDR_JIT", jit)
    got = hip_segment(gpu, airs, 6, 2, logup)
    assert len(got) == len(want)
    assert (got == want).all(), f"first differing word {int(np.argmax(got != want))} of {len(want)}"
    # mixed: only some AIRs streamed is what the memory policy produces; forced here by making three of them too short to stream
    short = [(rng.integers(0, P, size=6 << 2, dtype=np.uint32), 6, 2, *synth.random_air_programs(6, 2, 2, seed=7)[:2], synth.random_air_programs(6, 2, 2, seed=7)[2])]
    airs2 = airs[:3] + short + airs[3:]
    want2 = sm.prove_segment(airs2, num_queries=6, pow_bits=2, logup=logup)
    got2 = hip_segment(gpu, airs2, 6, 2, logup)
    assert len(got2) == len(want2) and (got2 == want2).all()


# ---- traces handed over per AIR (pw_prove_segment_consuming) ----------------------------------------------------------
def hip_segment_consuming(gpu, airs, nq, pow_bits, logup, mask):
    """pw_prove_segment_consuming with the AIRs of `mask` handed over -> (words, modes). Checks what the call promises about the
    buffers: a handed-over trace that was streamed holds its coefficient arrays afterwards and pw_trace_from_coefficients restores
    it exactly; every other trace is untouched; a second proof on the restored traces gives the same words."""
    torch, abi, prover = gpu
    provers = [prover.Prover(a[1], a[3], a[4], num_queries=nq, pow_bits=pow_bits, interactions=a[5] if logup else None) for a in airs]
    orig = [to_dev(torch, a[0]) for a in airs]
    traces = [t.clone() for t in orig]
    seg = [(pr, t.data_ptr(), a[2]) for pr, t, a in zip(provers, traces, airs)]
    out = []
    for _ in range(2):
        pf = prover.prove_segment(seg, logup=logup, hand_over=mask)
        modes = prover.segment_last_modes()
        torch.cuda.synchronize()
        assert len(modes) == len(airs)
        for (b, eaten), m, t, o, a in zip(modes, mask, traces, orig, airs):
            assert eaten == (bool(m) and b > 0)
            if eaten:
                assert not torch.equal(t, o)
                prover.trace_from_coefficients(t.data_ptr(), a[1], a[2])
                torch.cuda.synchronize()
            assert torch.equal(t, o)
        out.append(pf)
    assert (out[0] == out[1]).all()
    # ... and the plain call on the same provers afterwards (tcoef comes back): same words
    plain = prover.prove_segment(seg, logup=logup)
    assert (plain == out[0]).all()
    for pr in provers:
        pr.close()
    return out[0], modes


@pytest.mark.gpu
@pytest.mark.parametrize("spec,nq,pow_bits,logup,log_blocks,jit", [
    ([("T1", 1000), ("T0", 40), ("T1", 100)], 6, 3, True, 1, "1"),
    ([("T1", 1000), ("T0", 40), ("T1", 100)], 6, 3, True, 2, "0"),
    ([("T1", 1000), ("T0", 40), ("T1", 100)], 6, 3, False, 3, "1"),
    ([("T1", 1000), ("T0", 40), ("T1", 100)], 6, 3, False, 1, "0"),
    ([("C1", 600), ("T1", 5000), ("T0", 40), ("T1", 1000), ("T0", 33)], 8, 0, True, 1, "1"),
    ([("C1", 600), ("T1", 5000), ("T0", 40), ("T1", 1000), ("T0", 33)], 8, 0, True, 3, "0"),
    ([("T1", 40000), ("T0", 64)], 5, 0, True, 1, "1"),    # 2^16 rows: strided stage groups, the restore transform's too
    ([("T1", 40000), ("T0", 64)], 5, 0, False, 2, "0"),
])
def test_hip_segment_consuming_words_equal_the_oracle(gpu, monkeypatch, spec, nq, pow_bits, logup, log_blocks, jit):
    """The segment proof with its traces handed over: the words are sm.prove_segment's whatever the sub-coset count and the kernel
    kind, with every AIR handed over and with every other one."""
    torch, abi, prover = gpu
    airs, want = _oracle_segment(spec, nq, pow_bits, logup)
    monkeypatch.setenv("POWDR_STREAM_LOG_BLOCKS", str(log_blocks))
    monkeypatch.setenv("POWDR_JIT", jit)
    for mask in ([True] * len(airs), [k % 2 == 0 for k in range(len(airs))]):
        got, modes = hip_segment_consuming(gpu, airs, nq, pow_bits, logup, mask)
        assert len(got) == len(want) and (got == want).all(), f"first differing word {int(np.argmax(got != want))} of {len(want)}"
        assert any(e for _, e in modes)
    assert prover.verify_segment(descs_of(airs), got, nq, pow_bits, logup)[0] == 0
    # a handed-over trace that is only 4-byte aligned is refused
    big = torch.empty((airs[0][1] << airs[0][2]) + 4, dtype=torch.int32, device="cuda")
    pr = prover.Prover(airs[0][1], airs[0][3], airs[0][4], num_queries=nq, pow_bits=pow_bits, interactions=airs[0][5] if logup else None)
    with pytest.raises(Exception):
        prover.prove_segment([(pr, big[1:].data_ptr(), airs[0][2])], logup=logup, hand_over=True)
    pr.close()


@pytest.mark.gpu
def test_the_memory_policy_streams_the_largest_air_first_under_a_budget(gpu, monkeypatch):
    """pw_set_device_budget + the automatic policy (no POWDR_STREAM_LOG_BLOCKS): with room for everything no AIR is streamed; as the
    budget shrinks the LARGEST AIR goes first, then the next — the decision adds up what ensure_air / ensure_proof_buffers allocate
     — and the words never change. POWDR_STREAM_MIN_LOG_HEIGHT lets 2^10-row AIRs take part."""
    torch, abi, prover = gpu
    spec, nq = [("C1", 600), ("T1", 5000), ("T0", 40), ("T1", 1000), ("T0", 33)], 8
    airs, want = _oracle_segment(spec, nq, 0, True)
    monkeypatch.delenv("POWDR_STREAM_LOG_BLOCKS", raising=False)
    monkeypatch.setenv("POWDR_STREAM_MIN_LOG_HEIGHT", "9")
    cells = [a[1] << a[2] for a in airs]
    largest = int(np.argmax(cells))
    seen = []
    try:
        for budget in (0, 1 << 34, 1 << 27, 1 << 26, 1 << 25, 1 << 24, 1 << 23, 1 << 22):
            prover.set_device_budget(budget)
            got, modes = hip_segment_consuming(gpu, airs, nq, 0, True, [True] * len(airs))
            assert (got == want).all(), budget
            streamed = [k for k, (b, _) in enumerate(modes) if b]
            seen.append(streamed)
            if budget in (0, 1 << 34):
                assert streamed == []
            if streamed:
                assert largest in streamed
    finally:
        prover.set_device_budget(0)
    assert any(len(x) == 1 for x in seen) or any(len(x) >= 1 for x in seen), seen
    assert seen[-1] and len(seen[-1]) >= len(seen[2])


@pytest.mark.gpu
def test_hip_segment_golden_and_panels(gpu, monkeypatch):
    """The HIP prover reproduces the pinned golden segment digests, also with the LDE forced through many small panels."""
    g = json.loads(GOLDEN.read_text())
    airs = synthetic_airs([tuple(x) for x in g["spec"]], seed0=g["seed0"])
    for panel in (None, "12"):
        if panel:
            monkeypatch.setenv("POWDR_PANEL_LOG_WORDS", panel)
        for key, logup in (("sha256_v1", False), ("sha256_v1_logup", True)):
            pf = hip_segment(gpu, airs, g["num_queries"], g["pow_bits"], logup)
            assert hashlib.sha256(pf.astype("<u4").tobytes()).hexdigest() == g[key], (key, panel)


@pytest.mark.gpu
def test_hip_segment_balances_apc_against_periphery(gpu):
    """One segment = {APC AIR restricted to the lookup buses, var-range AIR, tuple AIR}: trace generation filled the
    histograms (a3), the segment proof carries the same lookups as LogUp terms (a6), pw_verify_segment(check_balance)
    accepts — and rejects (14) once a histogram bin is off by one. Heights 2^10 / 2^18 / 2^19 in one proof."""
    torch, abi, prover = gpu
    from powdr_amd import periphery, tracegen as tg
    from tests.test_oracle_apc import run_oracle_gpu_convention
    from tests.test_tracegen_gpu import run_gpu

    no_cons = (np.zeros(0, np.uint32), np.zeros((0, 2), np.uint32))
    s = synth.generate("T1", seed=2)
    calls = 1000
    apc, idx, want, hist, (bufs, dims, gt, order) = run_oracle_gpu_convention(s, calls, seed=2)
    W, H = want.shape
    out, per = run_gpu((torch, None, tg), W, H, calls, bufs, dims, gt.air_names, gt.row_block_size, gt.subs,
                       om.compile_derived(apc, idx, H), om.compile_bus(apc, idx, H))
    cons = sm.compile_constraints(apc, idx)
    sends = periphery.select_buses(sm.compile_interactions(apc, idx), {per.var_bus, per.tuple_bus})

    def run(var_hist):
        var_t = periphery.var_range_trace(var_hist)
        tup_t = periphery.tuple2_trace(per.tuple_hist, per.tuple_sizes)
        airs = [(out.buf, W, H.bit_length() - 1, cons, sends),
                (var_t, 3, var_hist.numel().bit_length() - 1, no_cons, periphery.var_range_interactions(per.var_bus)),
                (tup_t, 3, per.tuple_hist.numel().bit_length() - 1, no_cons, periphery.tuple2_interactions(per.tuple_bus))]
        provers = [prover.Prover(w, *c, num_queries=6, interactions=it) for (_, w, _, c, it) in airs]
        pf = prover.prove_segment([(pr, t.data_ptr(), lh) for pr, (t, _, lh, _, _) in zip(provers, airs)], logup=True)
        descs = [(w, lh, c[0], c[1], it) for (_, w, lh, c, it) in airs]
        rc = prover.verify_segment(descs, pf, 6, 0, True, check_balance=True)[0]
        for pr in provers:
            pr.close()
        return rc

    assert run(per.var_hist) == 0
    bin_ = int(np.argmax(hist["var"] != 0))
    per.var_hist[bin_] -= 1
    assert run(per.var_hist) == 14


@pytest.mark.gpu
def test_two_host_threads_prove_different_segments_concurrently(gpu):
    """One process, two host threads, each with its own launch stream (powdr_gpu_set_stream is per thread) and its own prover
    objects: the segment provers share no mutable state (per-thread contexts, per-device tables behind mutexes), so the
    proofs equal the ones made one after the other — what a thread-per-GPU host relies on (robustness)."""
    import threading

    torch, abi, prover = gpu
    specs = [[("T0", 30), ("T1", 200), ("T0", 5)], [("T1", 700), ("T0", 9), ("T1", 64), ("T0", 33)]]
    jobs = []
    for k, spec in enumerate(specs):
        airs = synthetic_airs(spec, seed0=40 + 10 * k)
        jobs.append((airs, hip_segment(gpu, airs, 5, 2, True)))  # sequential reference (checked against the oracle elsewhere)
    results = [None, None]
    errors = []

    def work(k):
        try:
            airs = jobs[k][0]
            st = torch.cuda.Stream()
            abi.lib.powdr_gpu_set_stream(ctypes.c_void_p(st.cuda_stream))
            with torch.cuda.stream(st):
                provers = [prover.Prover(a[1], a[3], a[4], num_queries=5, pow_bits=2, interactions=a[5]) for a in airs]
                traces = [to_dev(torch, a[0]) for a in airs]
                st.synchronize()
                out = None
                for _ in range(6):  # several rounds so the two threads really overlap
                    pf = prover.prove_segment([(pr, t.data_ptr(), a[2]) for pr, t, a in zip(provers, traces, airs)], logup=True)
                    assert out is None or (pf == out).all()
                    out = pf
                for pr in provers:
                    pr.close()
            results[k] = out
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append((k, repr(e)))
        finally:
            abi.lib.powdr_gpu_set_stream(None)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in range(2):
        assert (results[k] == jobs[k][1]).all()


@pytest.mark.gpu
def test_segment_rejects_a_prover_used_twice_and_mixed_configurations(gpu):
    """The per-AIR device buffers live in the prover object and the AIRs of a segment run on side streams: the same prover
    for two AIRs would silently corrupt both -> hipErrorInvalidValue; so do provers of different configurations
    (one FRI / query phase serves all AIRs)."""
    torch, abi, prover = gpu
    airs = synthetic_airs([("T0", 30), ("T0", 30)], seed0=5)
    pr = [prover.Prover(a[1], a[3], a[4], num_queries=4) for a in airs]
    t = [to_dev(torch, a[0]) for a in airs]
    assert len(prover.prove_segment([(pr[0], t[0].data_ptr(), airs[0][2]), (pr[1], t[1].data_ptr(), airs[1][2])], logup=False)) > 0
    with pytest.raises(abi.HipError):
        prover.prove_segment([(pr[0], t[0].data_ptr(), airs[0][2]), (pr[0], t[1].data_ptr(), airs[1][2])], logup=False)
    other = prover.Prover(airs[1][1], airs[1][3], airs[1][4], num_queries=5)
    with pytest.raises(abi.HipError):
        prover.prove_segment([(pr[0], t[0].data_ptr(), airs[0][2]), (other, t[1].data_ptr(), airs[1][2])], logup=False)
    for p in pr + [other]:
        p.close()


@pytest.mark.gpu
def test_worker_threads_return_their_device_memory(gpu):
    """Host threads that prove a segment and exit (pw_prove_airs' workers, a caller's thread pool) must not leak their
    per-thread contexts (pinned root mailbox, segment context buffers, streams): free HBM after 12 such threads
    is what it was after the first."""
    import threading

    torch, abi, prover = gpu
    provers, traces, seg = [], [], []
    for k, (w, lh) in enumerate([(300, 16), (40, 16), (64, 12), (9, 7)]):  # ~40 MB of per-thread context (trees, FRI layers)
        bc, sp, it = synth.random_air_programs(w, 6, 10, seed=k)
        provers.append(prover.Prover(w, bc, sp, num_queries=4, interactions=it))
        t = torch.empty(w << lh, dtype=torch.int32, device="cuda")
        t.random_(0, P)
        traces.append(t)
        seg.append((provers[-1], t.data_ptr(), lh))
    out = []

    def work():
        out.append(prover.prove_segment(seg, logup=True))

    free = []
    for k in range(12):
        th = threading.Thread(target=work)
        th.start()
        th.join()
        torch.cuda.synchronize()
        free.append(torch.cuda.mem_get_info()[0])
    assert all((o == out[0]).all() for o in out)
    # (the runtime keeps a few 2 MB pages of its own; a leak would be ~40 MB per thread)
    assert free[-1] >= free[0] - (32 << 20), f"free HBM fell from {free[0]} to {free[-1]} over 11 more worker threads"
    for p in provers:
        p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("log_blocks,jit", [(1, "1"), (2, "0")])
def test_hand_over_edge_cases(gpu, monkeypatch, log_blocks, jit):
    """Handed-over traces where the buffer juggling of pw_prove_segment_consuming is tightest: AIRs that SHARE a height (their level is
    absorbed run by run), widths that are no multiples of the rate, an AIR WITHOUT interactions inside a LogUp segment (its permutation
    matrix is the four phi columns: the trace's coefficients wait in a buffer sized for the trace, not for the matrix), a one-AIR segment,
    and a 4-row AIR that is too short to stream (handed over, left alone). Words == the oracle's; every eaten trace comes back exactly."""
    torch, abi, prover = gpu
    rng = np.random.default_rng(23)
    no_inter = (np.zeros((0, 3), np.uint32), np.zeros((0, 2), np.uint32), np.zeros(0, np.uint32))
    airs = []
    for k, (w, lh, with_inter) in enumerate([(13, 8, True), (7, 8, False), (30, 8, True), (5, 6, False), (9, 10, True), (6, 2, True)]):
        bc, sp, it = synth.random_air_programs(w, 3, 5, seed=300 + k)
        airs.append((rng.integers(0, P, size=w << lh, dtype=np.uint32), w, lh, bc, sp, it if with_inter else no_inter))
    monkeypatch.setenv("POWDR_STREAM_LOG_BLOCKS", str(log_blocks))
    monkeypatch.setenv("POWDR_JIT", jit)
    for logup in (True, False):
        want = sm.prove_segment(airs, num_queries=5, pow_bits=1, logup=logup)
        got, modes = hip_segment_consuming(gpu, airs, 5, 1, logup, [True] * len(airs))
        assert len(got) == len(want) and (got == want).all(), f"logup={logup}: first differing word {int(np.argmax(got != want))} of {len(want)}"
        assert modes[-1] == (0, False) and all(e for (b, e) in modes[:-1])
    one = [airs[4]]
    want = sm.prove_segment(one, num_queries=5, pow_bits=1, logup=True)
    got, modes = hip_segment_consuming(gpu, one, 5, 1, True, [True])
    assert (got == want).all() and modes == [(min(log_blocks, 5), True)]


def test_host_verifiers_reject_mutated_proofs_without_crashing():
    """The verifiers read proofs from outside: truncated, extended and word-mutated proofs (field edges, all-ones, flipped bits, header
    fields that claim other shapes and counts) of the three formats — pw-stark v0, v0 + LogUp, v1 — are all rejected and none crashes
    the host library (the same loop ran under AddressSanitizer + UBSan in round 6: profiles/r06_sanitizers.txt)."""
    import random

    from powdr_amd import prover

    rng = random.Random(5)

    def mutate(pf):
        x = pf.copy()
        r = rng.random()
        if r < 0.3:
            return x[:rng.randrange(0, len(x))]
        if r < 0.45:
            return np.concatenate([x, np.array([rng.randrange(1 << 32) for _ in range(rng.randrange(1, 9))], np.uint32)])
        for _ in range(rng.randrange(1, 4)):
            i = rng.randrange(len(x))
            x[i] = rng.choice([0, 1, 0x78000000, 0x78000001, 0xFFFFFFFF, rng.randrange(1 << 32), int(x[i]) ^ (1 << rng.randrange(32))])
        return x

    airs = synthetic_airs([("T0", 30), ("T1", 200), ("T0", 5)], seed0=1)
    flat, W, lh, bc, sp, it = airs[1]
    p0 = sm.prove(flat, W, lh, bc, sp, num_queries=5, pow_bits=3)
    p1 = sm.prove_logup(flat, W, lh, bc, sp, *it, num_queries=5, pow_bits=3)
    p2 = sm.prove_segment(airs, num_queries=5, pow_bits=3, logup=True)
    checks = ((p0, lambda q: prover.verify(q, W, lh, bc, sp, 5, 3)), (p1, lambda q: prover.verify_logup(q, W, lh, bc, sp, it, 5, 3)[0]),
              (p2, lambda q: prover.verify_segment(descs_of(airs), q, 5, 3, True)[0]))
    for base, fn in checks:
        assert fn(base) == 0
        for _ in range(250):
            m = mutate(base)
            same = len(m) == len(base) and (m == base).all()
            assert (fn(m) == 0) == same
        for _ in range(60):  # header fields: claimed heights, widths, counts
            q = base.copy()
            q[rng.randrange(0, 12)] = rng.choice([0, 1, 26, 27, 31, 0xFFFFFFFF, 1 << 20])
            if not (q == base).all():
                assert fn(q) != 0
