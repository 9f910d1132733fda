"""Run-time specialised expression kernels (csrc/jit.hpp, jit_codegen.hpp, prover_jit.hip): the AIR's
constraint / interaction programs emitted as straight-line HIP, compiled with hiprtc and run instead of the interpreter
kernels. Exact field arithmetic => the proof words must not depend on the path: every test proves with POWDR_JIT=1 and
POWDR_JIT=0 and compares both with the CPU oracle. CPU part: code generation + hiprtc cross-compilation need no GPU."""
import numpy as np
import pytest

from oracle import apc_model as om
from oracle import stark_model as sm
from powdr_amd import synth

P = om.P
PA, PC, ADD, SUB, MUL, NEG = om.OP_PUSH_APC, om.OP_PUSH_CONST, om.OP_ADD, om.OP_SUB, om.OP_MUL, om.OP_NEG


def _tables(shape, seed=0):
    s = synth.generate(shape, seed=seed)
    apc = om.load_apc(s.doc)
    idx = apc.poly_id_to_index()
    return len(idx), sm.compile_constraints(apc, idx), sm.compile_interactions(apc, idx)


def hand_made_air(W=9):
    """Interaction shapes the synthetic APCs do not have: no arguments at all, constant-only arguments (degree-0 denominators:
    more than two members per LogUp group), a multiplicity of degree 2, a group of one, nested expressions that need the stack."""
    bc, spans = [], []

    def span(words, sink, spans_):
        spans_.append((len(sink), len(words)))
        sink.extend(words)

    span([PA, 0, PA, 1, MUL, PA, 2, SUB], bc, spans)                       # a*b - c
    span([PA, 3, PA, 3, PC, 1, SUB, MUL], bc, spans)                       # d*(d-1)
    span([PA, 0, PA, 1, ADD, PA, 2, PA, 3, ADD, MUL, PA, 4, NEG, ADD], bc, spans)  # (a+b)*(c+d) - e: both operands compound
    ibc, ispans, inter = [], [], []

    def interaction(bus, mult, args):
        inter.append((bus, len(args), len(ispans)))
        span(mult, ibc, ispans)
        for a in args:
            span(a, ibc, ispans)

    interaction(5, [PA, 3], [])                                             # no arguments
    interaction(5, [PC, 1], [[PC, 7], [PC, 9]])                             # constants only: joins the group of the previous ones
    interaction(5, [PA, 4], [[PC, 3]])
    interaction(5, [PA, 5], [[PC, 11], [PC, 12], [PC, 13]])
    interaction(3, [PA, 3, PA, 4, MUL], [[PA, 5], [PC, 12]])                # degree-2 multiplicity
    interaction(6, [PA, 3], [[PA, 0, PA, 1, ADD, PA, 2, PA, 6, SUB, MUL], [PA, 7], [PA, 8], [PC, 1]])  # degree-2 argument: a group of its own
    interaction(1, [PC, 2, NEG], [[PA, 1], [PA, 2], [PA, 3], [PA, 4], [PA, 5], [PA, 6], [PA, 7]])      # seven arguments
    return (W, (np.array(bc, np.uint32), np.array(spans, np.uint32).reshape(-1, 2)),
            (np.array(inter, np.uint32).reshape(-1, 3), np.array(ispans, np.uint32).reshape(-1, 2), np.array(ibc, np.uint32)))


@pytest.mark.parametrize("shape", ["T0", "T1", "hand"])
@pytest.mark.parametrize("logup", [False, True])
def test_specialised_kernels_compile_without_a_gpu(shape, logup):
    from powdr_amd import prover

    W, (bc, spans), it = hand_made_air() if shape == "hand" else _tables(shape)
    r = prover.jit_compile_check(W, bc, spans, it if logup else None)
    assert r["rc"] == 0, r["error"]
    assert r["kernels"] >= 1 and r["chunks"] >= r["kernels"] and r["code_bytes"] > 4000
    if logup and shape == "hand":
        assert len(prover.logup_group_starts(it)) - 1 < len(it[0]) - 2  # the constant-only interactions share a group


def test_code_objects_are_kept_on_disk_across_provers(tmp_path, monkeypatch):
    """The on-disk cache of compiled units ($POWDR_JIT_CACHE_DIR): the first compilation of an AIR's kernels writes one entry per
    translation unit, the same AIR again (its programs released in between: the in-process cache only holds live programs) loads
    them instead of compiling; a truncated or foreign entry is ignored and replaced; POWDR_JIT_CACHE=0 switches the cache off."""
    from powdr_amd import prover

    monkeypatch.setenv("POWDR_JIT_CACHE_DIR", str(tmp_path / "cache" / "nested"))
    W, (bc, spans), it = hand_made_air()
    s0 = prover.jit_cache_stats()
    r1 = prover.jit_compile_check(W, bc, spans, it)
    s1 = prover.jit_cache_stats()
    assert r1["rc"] == 0 and s1["compiled"] - s0["compiled"] == r1["kernels"] and s1["from_disk"] == s0["from_disk"]
    entries = sorted((tmp_path / "cache" / "nested").glob("*.pwjc"))
    assert len(entries) == r1["kernels"] and not list((tmp_path / "cache" / "nested").glob("*.tmp*"))
    r2 = prover.jit_compile_check(W, bc, spans, it)
    s2 = prover.jit_cache_stats()
    assert r2 == r1 and s2["compiled"] == s1["compiled"] and s2["from_disk"] - s1["from_disk"] == r1["kernels"]
    # a damaged entry: ignored, compiled again, rewritten whole
    whole = entries[0].read_bytes()
    entries[0].write_bytes(whole[: len(whole) // 2])
    r3 = prover.jit_compile_check(W, bc, spans, it)
    s3 = prover.jit_cache_stats()
    assert r3 == r1 and s3["compiled"] - s2["compiled"] == 1 and s3["from_disk"] - s2["from_disk"] == r1["kernels"] - 1
    assert entries[0].read_bytes()[:8] == b"PWJC0001" and len(entries[0].read_bytes()) > len(whole) // 2  # (code objects are not byte-reproducible)
    # an entry whose stored source is another unit's (a file-name collision): not used
    entries[0].write_bytes(entries[-1].read_bytes() if len(entries) > 1 else whole[:24] + b"x" * (len(whole) - 24))
    r4 = prover.jit_compile_check(W, bc, spans, it)
    assert r4 == r1 and prover.jit_cache_stats()["compiled"] - s3["compiled"] == 1
    monkeypatch.setenv("POWDR_JIT_CACHE", "0")
    s4 = prover.jit_cache_stats()
    r5 = prover.jit_compile_check(W, bc, spans, it)
    s5 = prover.jit_cache_stats()
    assert r5 == r1 and s5["compiled"] - s4["compiled"] == r1["kernels"] and s5["from_disk"] == s4["from_disk"]


def test_jit_is_off_with_POWDR_JIT_0(monkeypatch):
    from powdr_amd import prover

    monkeypatch.setenv("POWDR_JIT", "0")
    W, (bc, spans), it = _tables("T0")
    r = prover.jit_compile_check(W, bc, spans, it)
    assert r["rc"] == 1 and r["kernels"] == 0


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def gpu():
    import torch
    from powdr_amd import abi, prover

    assert torch.cuda.is_available()
    return torch, abi, prover


def to_dev(torch, a):
    return torch.from_numpy(om.to_monty(np.ascontiguousarray(a, dtype=np.uint32)).view(np.int32)).cuda()


def _prove_both_paths(gpu, monkeypatch, flat, W, log_h, bc, spans, it, nq=5, pow_bits=3):
    torch, abi, prover = gpu
    d_t = to_dev(torch, flat)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("POWDR_JIT", mode)
        pr = prover.Prover(W, bc, spans, num_queries=nq, pow_bits=pow_bits, interactions=it)
        abi.call_stats(reset=True)
        out[mode] = pr.prove(d_t.data_ptr(), log_h)
        st = abi.call_stats()
        info = pr.specialised()
        if mode == "1":
            assert info["state"] == 1 and info["kernels"] >= 1, info
            assert st["jit_launches"] >= 1 and st["interpreter_launches"] == 0, st
            assert (pr.prove(d_t.data_ptr(), log_h) == out[mode]).all()  # second proof: the same kernels again
        else:
            assert info["state"] <= 0 and st["jit_launches"] == 0 and st["interpreter_launches"] >= 1, (info, st)
        pr.close()
    assert len(out["1"]) == len(out["0"]) and (out["1"] == out["0"]).all(), f"first differing word {int(np.argmax(out['1'] != out['0']))}"
    return out["1"]


@pytest.mark.gpu
@pytest.mark.parametrize("shape,calls", [("T0", 7), ("T0", 64), ("T1", 1000), ("T1", 4000), ("C1", 700)])
@pytest.mark.parametrize("logup", [False, True])
def test_proof_bytes_do_not_depend_on_the_path(gpu, monkeypatch, shape, calls, logup):
    """Specialised kernels == interpreter == oracle, constraints-only and with the LogUp phase, with zero-padding rows."""
    from tests.test_oracle_apc import run_oracle_gpu_convention

    s = synth.generate(shape, seed=5)
    apc, idx, trace, _, _ = run_oracle_gpu_convention(s, calls, seed=5)
    W, H = trace.shape
    log_h = H.bit_length() - 1
    flat = np.ascontiguousarray(trace).reshape(-1)
    bc, spans = sm.compile_constraints(apc, idx)
    it = sm.compile_interactions(apc, idx) if logup else None
    got = _prove_both_paths(gpu, monkeypatch, flat, W, log_h, bc, spans, it)
    want = sm.prove_logup(flat, W, log_h, bc, spans, *it, num_queries=5, pow_bits=3) if logup else sm.prove(flat, W, log_h, bc, spans, num_queries=5, pow_bits=3)
    assert len(got) == len(want) and (got == want).all()


@pytest.mark.gpu
@pytest.mark.parametrize("log_h", [1, 5, 11])
def test_hand_made_interaction_shapes(gpu, monkeypatch, log_h):
    """Groups of one, of four (constant-only denominators), interactions without arguments, degree-2 multiplicities and
    arguments, stack-using expressions — on random traces (byte parity does not need satisfied constraints)."""
    W, (bc, spans), it = hand_made_air()
    rng = np.random.default_rng(log_h)
    flat = rng.integers(0, P, W << log_h, dtype=np.uint32)
    flat[3 << log_h: (3 << log_h) + max(1, (1 << log_h) // 3)] = 0  # rows with zero multiplicities (no inversion there)
    got = _prove_both_paths(gpu, monkeypatch, flat, W, log_h, bc, spans, it, nq=4, pow_bits=0)
    assert (got == sm.prove_logup(flat, W, log_h, bc, spans, *it, num_queries=4, pow_bits=0)).all()
    got = _prove_both_paths(gpu, monkeypatch, flat, W, log_h, bc, spans, None, nq=4, pow_bits=0)
    assert (got == sm.prove(flat, W, log_h, bc, spans, num_queries=4, pow_bits=0)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("logup", [False, True])
def test_segment_proof_with_specialised_kernels(gpu, monkeypatch, logup):
    """pw_prove_segment: all AIRs' kernels compiled in one batch; mixed heights; an AIR without interactions inside a LogUp
    segment; the same words as the interpreter path and the oracle."""
    torch, abi, prover = gpu
    from tests.test_segment_proof import descs_of, synthetic_airs

    airs = synthetic_airs([("T0", 30), ("T1", 900), ("T0", 5), ("T1", 64)], seed0=21)
    W, (bc, spans), it = hand_made_air()
    rng = np.random.default_rng(3)
    airs.append((rng.integers(0, P, W << 6, dtype=np.uint32), W, 6, bc, spans, it))
    no_inter = (np.zeros((0, 3), np.uint32), np.zeros((0, 2), np.uint32), np.zeros(0, np.uint32))
    airs.append((rng.integers(0, P, 5 << 4, dtype=np.uint32), 5, 4, np.array([PA, 0, PA, 1, MUL], np.uint32), np.array([[0, 5]], np.uint32), no_inter))
    want = sm.prove_segment(airs, num_queries=4, pow_bits=2, logup=logup)
    traces = [to_dev(torch, a[0]) for a in airs]
    for mode in ("1", "0"):
        monkeypatch.setenv("POWDR_JIT", mode)
        provers = [prover.Prover(a[1], a[3], a[4], num_queries=4, pow_bits=2, interactions=a[5] if logup else None) for a in airs]
        abi.call_stats(reset=True)
        got = prover.prove_segment([(pr, t.data_ptr(), a[2]) for pr, t, a in zip(provers, traces, airs)], logup=logup)
        st = abi.call_stats()
        assert (st["jit_launches"] > 0) == (mode == "1") and (st["interpreter_launches"] > 0) == (mode == "0"), st
        assert len(got) == len(want) and (got == want).all(), (mode, int(np.argmax(got != want)))
        for pr in provers:
            pr.close()


@pytest.mark.gpu
def test_default_policy_specialises_tall_traces_only(gpu, monkeypatch):
    """Unset POWDR_JIT: a 2^10-row proof keeps the interpreter (compiling costs seconds), the same prover is specialised once
    it meets a trace of 2^POWDR_JIT_MIN_LOG_HEIGHT rows; pw_prover_specialise does it at set-up time."""
    torch, abi, prover = gpu
    monkeypatch.delenv("POWDR_JIT", raising=False)
    monkeypatch.setenv("POWDR_JIT_MIN_LOG_HEIGHT", "12")
    W, (bc, spans), it = _tables("T0")
    rng = np.random.default_rng(0)
    pr = prover.Prover(W, bc, spans, num_queries=3, interactions=it)
    small = to_dev(torch, rng.integers(0, P, W << 10, dtype=np.uint32))
    tall = to_dev(torch, rng.integers(0, P, W << 12, dtype=np.uint32))
    pr.prove(small.data_ptr(), 10)
    assert pr.specialised()["state"] == 0
    p_tall = pr.prove(tall.data_ptr(), 12)
    assert pr.specialised()["state"] == 1
    flat = om.from_monty(tall.cpu().numpy().view(np.uint32))
    assert (p_tall == sm.prove_logup(flat, W, 12, bc, spans, *it, num_queries=3, pow_bits=0)).all()
    pr.close()
    pr2 = prover.Prover(W, bc, spans, num_queries=3)
    assert pr2.specialise() and pr2.specialised()["state"] == 1
    pr2.close()
