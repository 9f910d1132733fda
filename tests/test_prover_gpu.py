"""Parity of the HIP prover against the pw-stark v0 oracle: every stage, then the proof
bytes. Bit-exact (integer field arithmetic)."""
import numpy as np
import pytest

from oracle import apc_model as om
from oracle import stark_model as sm
from powdr_amd import synth

P = om.P


def test_host_transcript_permutation_matches_oracle():
    """CPU-only: the host-side Poseidon2 of libpowdr_gpu (transcript) vs the oracle."""
    from powdr_amd import prover

    rng = np.random.default_rng(0)
    for _ in range(20):
        s = rng.integers(0, P, 16, dtype=np.uint32)
        assert (prover.poseidon2_host(s) == sm.poseidon2(s)).all()
    z = np.zeros(16, np.uint32)
    assert (prover.poseidon2_host(z) == sm.poseidon2(z)).all()


@pytest.fixture(scope="module")
def gpu():
    import torch
    from powdr_amd import abi, prover

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU (run with -m gpu on the GPU box)")
    return torch, abi, prover


def to_dev(torch, a):
    return torch.from_numpy(om.to_monty(np.ascontiguousarray(a, dtype=np.uint32)).view(np.int32)).cuda()


def from_dev(t):
    return om.from_monty(t.cpu().numpy().view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("log_h,W", [(1, 3), (2, 2), (3, 5), (4, 1), (5, 2), (6, 4), (7, 1), (8, 2), (9, 3), (10, 2), (11, 2),
                                      (12, 1), (13, 3), (14, 2), (15, 1), (16, 2), (17, 1), (18, 1), (19, 1), (20, 1),
                                      (21, 1), (22, 1), (23, 1),
                                      # many columns through the multi-pass plans (blockIdx.y > 0 in every stage group)
                                      (17, 9), (18, 5), (20, 3), (22, 3)])
def test_lde_matches_oracle(gpu, log_h, W):
    torch, abi, prover = gpu
    rng = np.random.default_rng(log_h)
    H = 1 << log_h
    t = rng.integers(0, P, W * H, dtype=np.uint32)
    want = sm.lde(t, W, log_h)
    d_t = to_dev(torch, t)
    d_c = torch.empty(W * H, dtype=torch.int32, device="cuda")
    d_l = torch.empty(W * 2 * H, dtype=torch.int32, device="cuda")
    abi.check(prover.lib.pw_lde_batch(d_t.data_ptr(), W, log_h, d_c.data_ptr(), d_l.data_ptr()), "pw_lde_batch")
    torch.cuda.synchronize()
    assert (from_dev(d_l) == want).all()
    # the fused schedule the provers use (contiguous DIF + DIT stage groups in one kernel): same LDE
    d_l2 = torch.empty(W * 2 * H, dtype=torch.int32, device="cuda")
    d_tmp = torch.empty(W * H, dtype=torch.int32, device="cuda")
    abi.check(prover.lib.pw_lde_fused(d_t.data_ptr(), W, log_h, d_tmp.data_ptr(), d_l2.data_ptr()), "pw_lde_fused")
    torch.cuda.synchronize()
    assert torch.equal(d_l2, d_l)
    # the coefficient buffer: H * coefficient[bitrev(q)]
    coef = sm.dft(t[:H], inverse=True)
    got = from_dev(d_c)[:H]
    br = np.array([int(format(q, f"0{log_h}b")[::-1], 2) for q in range(H)])
    assert (got == (coef[br].astype(np.uint64) * H % P)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("height,W", [(2, 1), (4, 8), (8, 9), (64, 16), (256, 23), (1024, 7), (4096, 100)])
def test_merkle_commit_matches_oracle(gpu, height, W):
    torch, abi, prover = gpu
    rng = np.random.default_rng(W)
    m = rng.integers(0, P, W * height, dtype=np.uint32)
    root, dig = sm.merkle_commit(m, height, W, want_digests=True)
    d_m = to_dev(torch, m)
    d_d = torch.empty((2 * height - 1) * 8, dtype=torch.int32, device="cuda")
    abi.check(prover.lib.pw_merkle_commit(d_m.data_ptr(), height, W, d_d.data_ptr()), "pw_merkle_commit")
    torch.cuda.synchronize()
    assert (from_dev(d_d) == dig).all()


def _synthetic(shape, calls, seed):
    from tests.test_oracle_apc import run_oracle_gpu_convention

    s = synth.generate(shape, seed=seed)
    apc, idx, trace, hist, _ = run_oracle_gpu_convention(s, calls, seed=seed)
    bc, spans = sm.compile_constraints(apc, idx)
    return s, np.ascontiguousarray(trace).reshape(-1), trace.shape, bc, spans


@pytest.mark.gpu
@pytest.mark.parametrize("shape,calls,nq,pow_bits", [("T0", 2, 3, 0), ("T0", 7, 4, 0), ("T0", 64, 5, 6), ("T1", 100, 8, 0),
                                                      ("T1", 1000, 6, 10), ("T1", 5000, 20, 0), ("C1", 600, 10, 0)])
def test_proof_bytes_match_oracle(gpu, shape, calls, nq, pow_bits):
    torch, abi, prover = gpu
    s, flat, (W, H), bc, spans = _synthetic(shape, calls, seed=9)
    log_h = H.bit_length() - 1
    want = sm.prove(flat, W, log_h, bc, spans, num_queries=nq, pow_bits=pow_bits)
    assert sm.verify(want, W, log_h, bc, spans, num_queries=nq, pow_bits=pow_bits) == 0
    pr = prover.Prover(W, bc, spans, num_queries=nq, pow_bits=pow_bits)
    d_t = to_dev(torch, flat)
    got = pr.prove(d_t.data_ptr(), log_h)
    assert len(got) == len(want)
    assert (got == want).all(), f"first differing word {int(np.argmax(got != want))}"
    # a second proof from the same object (buffer reuse) is identical
    assert (pr.prove(d_t.data_ptr(), log_h) == want).all()
    pr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,calls,nq,pow_bits", [("T0", 2, 3, 0), ("T0", 7, 4, 0), ("T0", 64, 5, 5), ("T1", 100, 6, 0), ("T1", 1000, 8, 0),
                                                      ("T1", 5000, 10, 0)])
def test_logup_proof_bytes_match_oracle(gpu, monkeypatch, shape, calls, nq, pow_bits):
    """pw-stark v0 + LogUp: the HIP proof (permutation columns, prefix scan, extended quotient, openings at
    zeta and g*zeta, third Merkle tree) equals the oracle's byte for byte, and the oracle's verifier accepts it."""
    torch, abi, prover = gpu
    from tests.test_oracle_apc import run_oracle_gpu_convention

    s = synth.generate(shape, seed=13)
    apc, idx, trace, _, _ = run_oracle_gpu_convention(s, calls, seed=13)
    W, H = trace.shape
    log_h = H.bit_length() - 1
    bc, spans = sm.compile_constraints(apc, idx)
    inter, ispans, ibc = sm.compile_interactions(apc, idx)
    flat = np.ascontiguousarray(trace).reshape(-1)
    want = sm.prove_logup(flat, W, log_h, bc, spans, inter, ispans, ibc, num_queries=nq, pow_bits=pow_bits)
    assert sm.verify_logup(want, W, log_h, bc, spans, inter, ispans, ibc, num_queries=nq, pow_bits=pow_bits) == 0
    pr = prover.Prover(W, bc, spans, num_queries=nq, pow_bits=pow_bits, interactions=(inter, ispans, ibc))
    d_t = to_dev(torch, flat)
    got = pr.prove(d_t.data_ptr(), log_h)
    assert len(got) == len(want)
    assert (got == want).all(), f"first differing word {int(np.argmax(got != want))} of {len(want)}"
    assert (pr.prove(d_t.data_ptr(), log_h) == want).all()
    assert pr.logup_path() == 2  # the synthetic APCs' multiplicities and arguments are small forms, like the real ones
    pr.close()
    # the interpreter path (what an interaction with a wider expression falls back to) gives the same words
    monkeypatch.setenv("POWDR_LOGUP_INTERPRET", "1")
    pri = prover.Prover(W, bc, spans, num_queries=nq, pow_bits=pow_bits, interactions=(inter, ispans, ibc))
    monkeypatch.delenv("POWDR_LOGUP_INTERPRET")
    assert pri.logup_path() == 1
    assert (pri.prove(d_t.data_ptr(), log_h) == want).all()
    pri.close()
    # the constraints-only prover is unaffected
    pr0 = prover.Prover(W, bc, spans, num_queries=nq, pow_bits=pow_bits)
    assert (pr0.prove(d_t.data_ptr(), log_h) == sm.prove(flat, W, log_h, bc, spans, num_queries=nq, pow_bits=pow_bits)).all()
    pr0.close()


def _prove_both_and_compare(torch, prover, flat, W, log_h, bc, spans, it, nq, pow_bits, satisfied=True):
    """HIP proof == oracle proof, word for word; both verifiers accept it (satisfied=False: a random trace that does not
    satisfy its random constraints — byte parity does not need it to — is rejected by both at the constraint identity)."""
    if it is None:
        want = sm.prove(flat, W, log_h, bc, spans, num_queries=nq, pow_bits=pow_bits)
    else:
        want = sm.prove_logup(flat, W, log_h, bc, spans, *it, num_queries=nq, pow_bits=pow_bits)
    pr = prover.Prover(W, bc, spans, num_queries=nq, pow_bits=pow_bits, interactions=it)
    d_t = to_dev(torch, flat)
    got = pr.prove(d_t.data_ptr(), log_h)
    pr.close()
    assert len(got) == len(want)
    assert (got == want).all(), f"first differing word {int(np.argmax(got != want))} of {len(want)}"
    if it is None:
        assert prover.verify(got, W, log_h, bc, spans, num_queries=nq, pow_bits=pow_bits) == (0 if satisfied else 2)
        assert sm.verify(got, W, log_h, bc, spans, num_queries=nq, pow_bits=pow_bits) == (0 if satisfied else 2)
    else:
        assert prover.verify_logup(got, W, log_h, bc, spans, it, num_queries=nq, pow_bits=pow_bits)[0] == 0
        assert sm.verify_logup(got, W, log_h, bc, spans, *it, num_queries=nq, pow_bits=pow_bits) == 0
    return got


@pytest.mark.gpu
@pytest.mark.parametrize("logup", [pytest.param(False, marks=pytest.mark.slow), True])
def test_c2_shape_proof_bytes_match_oracle(gpu, monkeypatch, logup):
    """BASELINE configs[1]'s AIR itself — 2 022 columns, 187 constraints, 1 734 bus interactions (867 LogUp groups,
    3 472 permutation columns) — at 2^14 rows, the largest height the CPU oracle proves in seconds: the trace comes
    from the oracle's trace generation, the proof bytes of the HIP prover equal the oracle's, constraints-only (the
    bench headline) and with the LogUp phase (the bench's `logup` leg). Shape pins of the real keccak APC:
    /root/reference/openvm-riscv/src/lib.rs:1377-1458."""
    torch, abi, prover = gpu
    from tests.test_oracle_apc import run_oracle_gpu_convention
    from tests._oracle_cases import baseline_case as _baseline_case

    if logup:
        # the trace (2^14 - 5 calls: a few zero-padding rows) and the oracle's proof of it, made once per suite run: the streamed and the
        # consuming tests of tests/test_streamed_prover.py prove the same trace against the same words
        (flat, W, log_h, bc, spans, it), want = _baseline_case("C2", 14)
        nq, pow_bits = 6, 4
        assert (W, log_h) == (2022, 14) and len(np.asarray(spans).reshape(-1, 2)) == 187
        assert len(it[0]) == 1734 and len(prover.logup_group_starts(it)) - 1 == 867
        pr = prover.Prover(W, bc, spans, num_queries=nq, pow_bits=pow_bits, interactions=it)
        d_t = to_dev(torch, flat)
        got = pr.prove(d_t.data_ptr(), 14)
        pr.close()
        assert len(got) == len(want) and (got == want).all(), f"first differing word {int(np.argmax(got != want))} of {len(want)}"
        assert prover.verify_logup(got, W, 14, bc, spans, it, num_queries=nq, pow_bits=pow_bits)[0] == 0
        assert sm.verify_logup(got, W, 14, bc, spans, *it, num_queries=nq, pow_bits=pow_bits) == 0
    else:
        s = synth.generate("C2", seed=0)
        calls = (1 << 14) - 5
        apc, idx, trace, _, _ = run_oracle_gpu_convention(s, calls, seed=0)
        W, H = trace.shape
        assert (W, H) == (2022, 1 << 14)
        bc, spans = sm.compile_constraints(apc, idx)
        assert len(spans) == 187
        it, nq, pow_bits = None, 8, 8
        flat = np.ascontiguousarray(trace).reshape(-1)
        got = _prove_both_and_compare(torch, prover, flat, W, 14, bc, spans, it, nq=nq, pow_bits=pow_bits)
    # the opened values reach the host through host-mapped memory, the permutation matrix's in four slices that the host absorbs
    # while the next is computed; the plain path (one copy after the last kernel) gives the same words
    monkeypatch.setenv("POWDR_OPENINGS_OVERLAP", "0")
    pr = prover.Prover(W, bc, spans, num_queries=nq, pow_bits=pow_bits, interactions=it)
    d_t = to_dev(torch, flat)
    assert (pr.prove(d_t.data_ptr(), 14) == got).all()
    pr.close()


_C1_TRACE = {}


@pytest.mark.gpu
@pytest.mark.parametrize("logup", [pytest.param(False, marks=pytest.mark.slow), True])
def test_c1_full_size_proof_bytes_match_oracle(gpu, logup):
    """BASELINE configs[0] (sha256-shaped single segment, 2^16 rows, 1 204 columns, 377 constraints, 954 interactions)
    at its FULL size: oracle trace generation -> oracle proof == HIP proof (tools/run_c1_oracle.py records the
    oracle's timings of the same run under profiles/)."""
    torch, abi, prover = gpu
    from tests.test_oracle_apc import run_oracle_gpu_convention

    if "c1" not in _C1_TRACE:  # the oracle's single-threaded row loop over 2^16 rows: once for both proof kinds
        s = synth.generate("C1", seed=1)
        calls = (1 << 16) - 3
        _C1_TRACE["c1"] = run_oracle_gpu_convention(s, calls, seed=1)[:3]
    apc, idx, trace = _C1_TRACE["c1"]
    W, H = trace.shape
    assert (W, H) == (1204, 1 << 16)
    bc, spans = sm.compile_constraints(apc, idx)
    it = sm.compile_interactions(apc, idx) if logup else None
    _prove_both_and_compare(torch, prover, np.ascontiguousarray(trace).reshape(-1), W, 16, bc, spans, it, nq=6, pow_bits=4)


@pytest.mark.gpu
@pytest.mark.parametrize("W,log_h,panel_log_words,logup", [(300, 16, 21, False), (300, 16, 12, False), (1204, 12, 12, True),
                                                            (1204, 12, 19, True), (257, 17, 20, False)])
def test_multi_panel_lde_proof_bytes_match_oracle(gpu, monkeypatch, W, log_h, panel_log_words, logup):
    """The LDE of a wide trace runs panel by panel through a bounded coefficient buffer (1 GB panels at C2 = 8 panels,
    C3 = 15). POWDR_PANEL_LOG_WORDS (read per call) forces many small panels here — 8 columns per panel at the floor —
    for the main trace and, with LogUp, for the permutation matrix; the proof bytes must not depend on the panelling."""
    torch, abi, prover = gpu
    monkeypatch.setenv("POWDR_PANEL_LOG_WORDS", str(panel_log_words))
    H = 1 << log_h
    expect_panels = -(-W // max(8, (1 << panel_log_words) // H))
    assert expect_panels >= 5
    if logup:
        from tests.test_oracle_apc import run_oracle_gpu_convention

        s = synth.generate("C1", seed=4)
        apc, idx, trace, _, _ = run_oracle_gpu_convention(s, H - 9, seed=4)
        assert trace.shape == (W, H)
        bc, spans = sm.compile_constraints(apc, idx)
        it = sm.compile_interactions(apc, idx)
        flat = np.ascontiguousarray(trace).reshape(-1)
    else:
        rng = np.random.default_rng(W + log_h + panel_log_words)
        flat = rng.integers(0, P, W * H, dtype=np.uint32)
        PA, PC = om.OP_PUSH_APC, om.OP_PUSH_CONST
        bc, spans = [], []
        for k in range(7):
            off = len(bc)
            a, b, c = (int(x) for x in rng.integers(0, W, 3))
            bc += [PA, a, PA, b, om.OP_MUL, PA, c, om.OP_MUL, PC, int(rng.integers(0, P)), om.OP_ADD]
            spans.append((off, len(bc) - off))
        bc, spans, it = np.array(bc, np.uint32), np.array(spans, np.uint32).reshape(-1, 2), None
    got = _prove_both_and_compare(torch, prover, flat, W, log_h, bc, spans, it, nq=5, pow_bits=0, satisfied=logup)
    # and the default panelling gives the same bytes
    monkeypatch.delenv("POWDR_PANEL_LOG_WORDS")
    pr = prover.Prover(W, bc, spans, num_queries=5, interactions=it)
    assert (pr.prove(to_dev(torch, flat).data_ptr(), log_h) == got).all()
    pr.close()


@pytest.mark.gpu
def test_non_canonical_proof_words_are_rejected(gpu):
    """A proof word w and w + p encode the same field element; both verifiers reject the second encoding (code 13),
    so proofs are not malleable."""
    torch, abi, prover = gpu
    s, flat, (W, H), bc, spans = _synthetic("T0", 30, seed=2)
    log_h = H.bit_length() - 1
    pr = prover.Prover(W, bc, spans, num_queries=3)
    proof = pr.prove(to_dev(torch, flat).data_ptr(), log_h)
    pr.close()
    assert prover.verify(proof, W, log_h, bc, spans, num_queries=3) == 0
    for pos in (6, 20, len(proof) - 1):
        bad = proof.copy()
        if int(bad[pos]) + P < (1 << 32):
            bad[pos] = int(bad[pos]) + P
            assert prover.verify(bad, W, log_h, bc, spans, num_queries=3) == 13
            assert sm.verify(bad, W, log_h, bc, spans, num_queries=3) == 13


@pytest.mark.gpu
@pytest.mark.parametrize("log_h", [4, 13, 17])
def test_logup_bus_balances_across_airs(gpu, log_h):
    """Two AIRs on one bus (sends / permuted receives), proven on the device, verified on the host: each proof is
    valid and the cumulative sums cancel; 2^17 rows crosses the scan's block boundaries (4096 rows per workgroup)."""
    torch, abi, prover = gpu
    from powdr_amd import sharding
    from tests.test_oracle_stark import balanced_bus_pair, ext_add_canonical

    no_cons = (np.zeros(0, np.uint32), np.zeros((0, 2), np.uint32))
    airs = [(to_dev(torch, t.reshape(-1)), t, it, prover.Prover(3, *no_cons, num_queries=6, interactions=it))
            for t, it in balanced_bus_pair(log_h, seed=log_h)]
    roots = [pr.trace_root(d.data_ptr(), log_h) for d, _, _, pr in airs]  # phase 1
    seed = sharding.commitment_digest(np.array(roots))
    sums = []
    for (d, t, it, pr), root in zip(airs, roots):
        pr.set_bus_seed(seed)
        proof = pr.prove(d.data_ptr(), log_h)
        assert (proof[7:15] == root).all() and (proof[15:23] == seed).all()
        rc, S, vroot = prover.verify_logup(proof, 3, log_h, *no_cons, it, num_queries=6, bus_seed=seed, with_root=True)
        assert rc == 0 and (vroot == root).all()
        assert prover.verify_logup(proof, 3, log_h, *no_cons, it, num_queries=6)[0] == 12
        if log_h <= 13:
            assert sm.verify_logup(proof, 3, log_h, *no_cons, *it, num_queries=6, bus_seed=seed) == 0
            assert (proof == sm.prove_logup(t.reshape(-1), 3, log_h, *no_cons, *it, num_queries=6, bus_seed=seed)).all()
        pr.set_bus_seed(None)  # back to the lone-AIR seed
        assert prover.verify_logup(pr.prove(d.data_ptr(), log_h), 3, log_h, *no_cons, it, num_queries=6)[0] == 0
        pr.close()
        sums.append(S)
    assert sums[0].any() and (ext_add_canonical(sums[0], sums[1]) == 0).all()


@pytest.mark.gpu
def test_logup_large_proof_verifies(gpu):
    """2^16-row, 160-column synthetic APC trace with all its bus interactions inside the proof: the HIP proof is
    accepted by the product verifier and the oracle's; breaking one interaction operand in the trace is rejected
    by neither (the trace is still self-consistent) but changes S."""
    torch, abi, prover = gpu
    from powdr_amd import tracegen as tg

    s = synth.generate("T1", seed=22)
    apc = om.load_apc(s.doc)
    idx = apc.poly_id_to_index()
    bc, spans = sm.compile_constraints(apc, idx)
    it = sm.compile_interactions(apc, idx)
    gt = om.build_gpu_tables(apc, idx)
    calls = (1 << 16) - 3
    H, W, log_h = 1 << 16, len(idx), 16
    bufs, dims = synth.fill_dummy_traces_numpy(s, calls, seed=22)
    name_to = {n: i for i, (n, _, _, _) in enumerate(dims)}
    airs = [(to_dev(torch, bufs[name_to[n]]), dims[name_to[n]][1], dims[name_to[n]][2], b) for n, b in zip(gt.air_names, gt.row_block_size)]
    out = tg.DeviceMatrix.zeros(H, W)
    keep = [tg.apc_tracegen(out, airs, gt.subs, calls), tg.apc_apply_derived_expr(out, calls, *om.compile_derived(apc, idx, H))]
    pr = prover.Prover(W, bc, spans, num_queries=20, interactions=it)
    proof = pr.prove(out.ptr(), log_h)
    rc, S = prover.verify_logup(proof, W, log_h, bc, spans, it, num_queries=20)
    assert rc == 0
    assert sm.verify_logup(proof, W, log_h, bc, spans, *it, num_queries=20) == 0
    bad = proof.copy()
    bad[31] = (int(bad[31]) + 1) % P  # claimed S
    assert prover.verify_logup(bad, W, log_h, bc, spans, it, num_queries=20)[0] == 2
    pr.close()
    del keep


@pytest.mark.gpu
def test_golden_proofs_from_the_device(gpu):
    """The HIP prover reproduces the committed golden proofs (tests/golden/pw_stark_proofs_T0.npz) word for word."""
    from pathlib import Path

    torch, abi, prover = gpu
    z = np.load(Path(__file__).parent / "golden" / "pw_stark_proofs_T0.npz")
    W, log_h = int(z["width"]), int(z["log_h"])
    d_t = to_dev(torch, z["trace"])
    pr = prover.Prover(W, z["cons_bc"], z["cons_spans"], num_queries=4, pow_bits=5)
    assert (pr.prove(d_t.data_ptr(), log_h) == z["proof_v0"]).all()
    pr.close()
    pr = prover.Prover(W, z["cons_bc"], z["cons_spans"], num_queries=4, pow_bits=5,
                       interactions=(z["inter"], z["inter_spans"], z["inter_bc"]))
    assert (pr.prove(d_t.data_ptr(), log_h) == z["proof_logup"]).all()
    pr.close()


@pytest.mark.gpu
def test_reference_keccak_constraints_proof_bytes(gpu):
    """The reference's own pre-optimisation keccak APC (autoprecompiles/tests/keccak_apc_pre_opt.json.gz through
    tests/golden/make_golden.py): 27 521 columns and its 28 627 real constraint programs (post-fix, deep stacks), on a
    random trace. Byte parity of the proof does not need the constraints to hold; the mock prover counts the same
    violations as a direct evaluation of the programs."""
    from pathlib import Path

    torch, abi, prover = gpu
    z = np.load(Path(__file__).parent / "golden" / "keccak_apc_pre_opt.apc.npz")
    W, log_h = len(z["poly_ids"]), 6
    H = 1 << log_h
    bc, spans = z["cons_bc"], z["cons_spans"]
    rng = np.random.default_rng(3)
    flat = rng.integers(0, 4, W * H).astype(np.uint32)  # small values: a fair share of the constraints does vanish
    want = sm.prove(flat, W, log_h, bc, spans, num_queries=3)
    pr = prover.Prover(W, bc, spans, num_queries=3)
    assert pr.max_constraint_degree() <= 3
    d_t = to_dev(torch, flat)
    got = pr.prove(d_t.data_ptr(), log_h)
    assert len(got) == len(want) and (got == want).all()
    n_bad, row, cons = pr.check_constraints(d_t.data_ptr(), log_h)
    trace = flat.reshape(W, H)
    bad = 0
    for k in range(0, len(spans), 97):  # a sample of the programs, evaluated directly on every row
        off, ln = spans[k]
        prog = bc[off:off + ln].copy()
        pos = 0
        while pos < len(prog):
            if prog[pos] == om.OP_PUSH_APC:
                prog[pos + 1] *= H
            pos += 2 if prog[pos] <= 1 else 1
        bad += sum(om.c_eval_expr(prog, flat, r) != 0 for r in range(H))
    assert n_bad > 0 and bad > 0 and bad <= n_bad
    pr.close()
    # the same AIR with its 13 262 real bus interactions inside the proof (execution bridge, memory with 7 arguments,
    # range checks, bitwise lookups, pc lookup): grouping, permutation trace, extended quotient
    it = (z["bus_inter"], z["bus_spans"], z["bus_bc"])
    groups = prover.logup_group_starts(it)
    assert (groups == sm.group_starts(*it)).all() and len(groups) - 1 < len(it[0])
    want = sm.prove_logup(flat, W, log_h, bc, spans, *it, num_queries=3)
    pr = prover.Prover(W, bc, spans, num_queries=3, interactions=it)
    got = pr.prove(d_t.data_ptr(), log_h)
    assert len(got) == len(want) and (got == want).all()
    pr.close()


@pytest.mark.gpu
def test_reserve_allocates_what_the_proof_needs(gpu):
    """pw_prover_reserve sizes the buffers exactly like pw_prover_prove: a proof after it allocates nothing more, and
    is the same proof."""
    torch, abi, prover = gpu
    s, flat, (W, H), bc, spans = _synthetic("T1", 3000, seed=5)
    log_h = H.bit_length() - 1
    it = None
    for interactions in (None, "logup"):
        if interactions:
            from tests.test_oracle_apc import run_oracle_gpu_convention

            apc, idx, _, _, _ = run_oracle_gpu_convention(synth.generate("T1", seed=5), 3000, seed=5)
            it = sm.compile_interactions(apc, idx)
        pr = prover.Prover(W, bc, spans, num_queries=9, interactions=it)
        assert 1 <= pr.max_constraint_degree() <= 3  # the degree bound of a blow-up-2 quotient
        assert pr.device_bytes() == 0
        pr.reserve(log_h)
        reserved = pr.device_bytes()
        assert reserved > 12 * W * H
        d_t = to_dev(torch, flat)
        proof = pr.prove(d_t.data_ptr(), log_h)
        assert pr.device_bytes() == reserved
        pr2 = prover.Prover(W, bc, spans, num_queries=9, interactions=it)
        assert (pr2.prove(d_t.data_ptr(), log_h) == proof).all() and pr2.device_bytes() == reserved
        pr.close()
        pr2.close()


@pytest.mark.gpu
def test_large_proof_verifies(gpu):
    """2^16-row, 160-column trace: too slow to prove on the CPU oracle in a test, so the HIP
    proof is checked with the oracle's VERIFIER (accept) and a corrupted trace (reject)."""
    torch, abi, prover = gpu
    from powdr_amd import tracegen as tg

    s = synth.generate("T1", seed=21)
    apc = om.load_apc(s.doc)
    idx = apc.poly_id_to_index()
    bc, spans = sm.compile_constraints(apc, idx)
    gt = om.build_gpu_tables(apc, idx)
    calls = (1 << 16) - 5
    H, W, log_h = 1 << 16, len(idx), 16
    bufs, dims = synth.fill_dummy_traces_numpy(s, calls, seed=21)
    name_to = {n: i for i, (n, _, _, _) in enumerate(dims)}
    airs = [(to_dev(torch, bufs[name_to[n]]), dims[name_to[n]][1], dims[name_to[n]][2], b) for n, b in zip(gt.air_names, gt.row_block_size)]
    out = tg.DeviceMatrix.zeros(H, W)
    keep = [tg.apc_tracegen(out, airs, gt.subs, calls), tg.apc_apply_derived_expr(out, calls, *om.compile_derived(apc, idx, H))]
    pr = prover.Prover(W, bc, spans, num_queries=30, pow_bits=0)
    proof = pr.prove(out.ptr(), log_h)
    assert sm.verify(proof, W, log_h, bc, spans, num_queries=30) == 0
    valid_col = idx[[q for q, k in s.kinds.items() if k[0] == "valid"][0]]
    out.buf[valid_col * H + 17] = 2 * 0x0FFFFFFE % P  # is_valid = 2 (Montgomery) breaks is_valid*(is_valid-1)
    bad = pr.prove(out.ptr(), log_h)
    assert sm.verify(bad, W, log_h, bc, spans, num_queries=30) != 0


@pytest.mark.gpu
def test_c2_full_size_trace_generation_and_proof(gpu):
    """BASELINE configs[1] at full size (2 022 columns x 2^20 rows, ~185 GB of HBM): the whole hot path
    through the C ABI, checked by size-independent properties — the proof is accepted by the oracle's
    verifier (constraints hold on the generated trace, commitments/openings/FRI consistent), the binned
    histogram path equals the direct-atomic path, histogram mass equals the number of lookups, gathered
    cells equal their sources, padding rows are zero."""
    import os

    torch, abi, prover = gpu
    if torch.cuda.mem_get_info()[1] < 250e9:
        pytest.skip("needs a 288 GB MI355X")
    import bench
    from powdr_amd import host

    wl = bench.build_workload("C2", 20, False, seed=0)
    H, W, calls = wl["H"], wl["W"], wl["calls"]
    per = wl["per"]
    hists = {}
    for mode in ("0", "1"):
        os.environ["POWDR_BUS_BINNED"] = mode
        for t in (per.var_hist, per.tuple_hist, per.bitwise_hist):
            t.zero_()
        wl["apc"].generate_witness_gpu(wl["instr_air"], wl["dummy"], calls, wl["out"].data_ptr(), per)
        torch.cuda.synchronize()
        hists[mode] = [t.clone() for t in (per.var_hist, per.tuple_hist, per.bitwise_hist)]
    os.environ.pop("POWDR_BUS_BINNED")
    for a, b in zip(hists["0"], hists["1"]):
        assert torch.equal(a, b)
    assert all(int(h.sum()) > 0 for h in hists["1"])
    # gathered cells == source cells for a sample of substitutions
    s = wl["synth"]
    m = wl["out"].view(W, H)
    idx = {pid: i for i, pid in enumerate(s.poly_ids)}
    r = torch.arange(calls, device="cuda", dtype=torch.int64)
    for pid in list(s.source_of)[:: max(1, len(s.source_of) // 40)]:
        name, row, col = s.source_of[pid]
        t, w, h, b = wl["tensors"][name]
        assert torch.equal(m[idx[pid]], t[col * h + row + r * b])
    # proof of the full-size trace, verified on the CPU
    bc, spans = wl["cons"]
    pr = prover.Prover(W, bc, spans, num_queries=24, pow_bits=12)
    proof = pr.prove(wl["out"].data_ptr(), 20)
    assert sm.verify(proof, W, 20, bc, spans, num_queries=24, pow_bits=12) == 0
    assert prover.verify(proof, W, 20, bc, spans, num_queries=24, pow_bits=12) == 0  # product-side verifier
    proof2 = proof.copy()
    proof2[len(proof2) // 2] ^= 1
    assert sm.verify(proof2, W, 20, bc, spans, num_queries=24, pow_bits=12) != 0
    pr.close()


@pytest.mark.gpu
def test_multi_air_segment_commitment_merge(gpu):
    """C4/C5-shaped plumbing on one GPU: several AIRs of different shapes are assigned to ranks by cell
    count (sharding.assign_units), proven, and their commitments merged into one digest."""
    torch, abi, prover = gpu
    from powdr_amd import sharding

    units = [("T0", 60), ("T1", 900), ("T0", 7), ("T1", 300), ("C1", 200)]
    proofs, roots, cells = [], [], []
    for shape, calls in units:
        s, flat, (W, H), bc, spans = _synthetic(shape, calls, seed=3)
        pr = prover.Prover(W, bc, spans, num_queries=4)
        p = pr.prove(to_dev(torch, flat).data_ptr(), H.bit_length() - 1)
        assert sm.verify(p, W, H.bit_length() - 1, bc, spans, num_queries=4) == 0
        roots.append(p[6:14])
        cells.append(W * H)
        pr.close()
    parts = sharding.assign_units(cells, 4)
    assert sorted(u for p in parts for u in p) == list(range(len(units)))
    merged = sharding.merge_commitments(list(range(len(units))), np.array(roots), len(units))
    assert (merged == np.array(roots)).all()
    d = sharding.commitment_digest(merged)
    assert d.shape == (8,) and (d != sharding.commitment_digest(merged[::-1])).any()


@pytest.mark.gpu
@pytest.mark.parametrize("W,log_h,n_cons,nq,pow_bits", [(1, 1, 0, 2, 0), (1, 2, 1, 3, 0), (2, 1, 2, 1, 4), (7, 3, 3, 4, 0), (8, 4, 5, 5, 0),
                                                         (9, 5, 4, 6, 3), (17, 8, 9, 7, 0), (33, 12, 6, 9, 0), (3, 14, 2, 5, 0)])
def test_proof_bytes_on_arbitrary_traces(gpu, W, log_h, n_cons, nq, pow_bits):
    """Byte parity does not depend on the trace satisfying its constraints: random traces, random
    (unsatisfied) constraint programs, widths that are not multiples of the sponge rate, tiny heights,
    zero constraints."""
    torch, abi, prover = gpu
    rng = np.random.default_rng(W * 100 + log_h)
    H = 1 << log_h
    flat = rng.integers(0, P, W * H, dtype=np.uint32)
    bc, spans = [], []
    PA, PC = om.OP_PUSH_APC, om.OP_PUSH_CONST
    for _ in range(n_cons):
        off = len(bc)
        a, b, c = (int(x) for x in rng.integers(0, W, 3))
        bc += [PA, a, PA, b, om.OP_MUL, PC, int(rng.integers(0, P)), om.OP_SUB, PA, c, om.OP_NEG, om.OP_ADD]
        spans.append((off, len(bc) - off))
    bc = np.array(bc, np.uint32)
    spans = np.array(spans, np.uint32).reshape(-1, 2)
    want = sm.prove(flat, W, log_h, bc, spans, num_queries=nq, pow_bits=pow_bits)
    pr = prover.Prover(W, bc, spans, num_queries=nq, pow_bits=pow_bits)
    assert pr.max_constraint_degree() == (2 if n_cons else 0)
    got = pr.prove(to_dev(torch, flat).data_ptr(), log_h)
    assert len(got) == len(want) and (got == want).all()
    pr.close()


@pytest.mark.gpu
def test_c3_scale_trace_and_proof(gpu):
    """BASELINE configs[2] scale on one GPU: 3 731 columns x 2^22 rows (15.6 G cells, W*H > 2^32, ~190 GB peak):
    trace generation through the column-operand path, proof, host verification (tools/run_c3_scale.py)."""
    torch, abi, prover = gpu
    import gc

    gc.collect()  # provers of earlier tests release their device buffers in __del__
    torch.cuda.empty_cache()
    if torch.cuda.mem_get_info()[0] < 230e9:
        pytest.skip(f"needs ~190 GB of free HBM, {torch.cuda.mem_get_info()[0] / 1e9:.0f} GB free")
    import importlib.util
    from pathlib import Path

    spec = importlib.util.spec_from_file_location("run_c3_scale", Path(__file__).resolve().parents[1] / "tools" / "run_c3_scale.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rep = mod.run(22, 16, 8, verbose=False)
    assert rep["verify_rc"] == 0 and rep["cells"] == 3731 << 22
    torch.cuda.empty_cache()


@pytest.mark.gpu
def test_mock_prover_reports_violations(gpu):
    """pw_prover_check_constraints = the reference's prove_mock / debug_proving_ctx: a generated trace has no
    violation (padding rows included); a corrupted cell is located exactly."""
    torch, abi, prover = gpu
    s, flat, (W, H), bc, spans = _synthetic("T1", 500, seed=5)
    log_h = H.bit_length() - 1
    pr = prover.Prover(W, bc, spans, num_queries=2)
    d_t = to_dev(torch, flat)
    assert pr.check_constraints(d_t.data_ptr(), log_h) == (0, None, None)
    apc = om.load_apc(s.doc)
    idx = apc.poly_id_to_index()
    valid_col = idx[[q for q, k in s.kinds.items() if k[0] == "valid"][0]]
    d_t[valid_col * H + 123] = int(om.to_monty(np.array([2], np.uint32))[0])
    n, row, c = pr.check_constraints(d_t.data_ptr(), log_h)
    assert n >= 1 and row == 123
    # the oracle's view of the same row: constraint c is the first non-zero one
    trace = flat.reshape(W, H).copy()
    trace[valid_col, 123] = 2
    vals = [om.eval_ast(e, lambda pid: int(trace[idx[pid], 123])) for e in apc.constraints]
    assert vals[c] != 0 and all(v == 0 for v in vals[:c])
    pr.close()
