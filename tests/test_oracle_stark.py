"""CPU tests of the pw-stark v0 oracle: internal consistency of every stage and the
accept/reject behaviour of prove -> verify (the only kind of check the reference's own
tests make on this path: openvm-riscv/src/lib.rs:337-341 `verify_app_proof`)."""
import numpy as np
import pytest

from oracle import apc_model as om
from oracle import stark_model as sm
from powdr_amd import synth

P = om.P


def test_field_constants():
    g = 0x1A427A41  # 31^15: generator of the order-2^27 subgroup
    assert pow(31, 15, P) == g
    assert pow(g, 1 << 27, P) == 1 and pow(g, 1 << 26, P) != 1
    assert sm.root_of_unity(27) == g and sm.root_of_unity(1) == P - 1
    # 11 is a quadratic non-residue, so X^4 - 11 is irreducible over BabyBear (p = 1 mod 4)
    assert pow(11, (P - 1) // 2, P) == P - 1


def test_ext_field():
    rng = np.random.default_rng(0)
    for _ in range(50):
        a = rng.integers(0, P, 4, dtype=np.uint32)
        b = rng.integers(0, P, 4, dtype=np.uint32)
        assert (sm.ext_mul(a, b) == sm.ext_mul(b, a)).all()
        assert (sm.ext_mul(a, sm.ext_inv(a)) == [1, 0, 0, 0]).all()
    x = np.array([0, 1, 0, 0], np.uint32)
    x2 = sm.ext_mul(x, x)
    assert (sm.ext_mul(x2, x2) == [11, 0, 0, 0]).all()


def test_dft_matches_naive_and_inverts():
    rng = np.random.default_rng(1)
    for log_n in (1, 2, 5, 8):
        a = rng.integers(0, P, 1 << log_n, dtype=np.uint32)
        f = sm.dft(a)
        assert (f == sm.dft_naive(a)).all()
        assert (sm.dft(f, inverse=True) == a).all()


def test_lde_is_low_degree_extension():
    rng = np.random.default_rng(2)
    log_h, W = 5, 3
    H = 1 << log_h
    t = rng.integers(0, P, W * H, dtype=np.uint32)
    l = sm.lde(t, W, log_h).reshape(W, 2 * H)
    w = sm.root_of_unity(log_h + 1)
    for c in range(W):
        coef = sm.dft(t[c * H : (c + 1) * H], inverse=True)
        for j in (0, 1, 7, 2 * H - 1):
            x = 31 * pow(w, j, P) % P
            want = sum(int(coef[i]) * pow(x, i, P) for i in range(H)) % P
            assert int(l[c, j]) == want


def test_poseidon2_is_a_permutation_with_fixed_constants():
    e, i, d = sm.poseidon2_constants()
    assert e.shape == (8, 16) and (e < P).all() and (i < P).all()
    assert d[1] == 1 and d[0] == P - 2 and (2 * int(d[3])) % P == 1  # [-2, 1, 2, 1/2, ...]
    z = sm.poseidon2(np.zeros(16, np.uint32))
    o = sm.poseidon2(np.arange(16, dtype=np.uint32))
    assert len(set(z.tolist())) > 8 and (z != o).any()
    # regression pin of this repo's own constant table (NOT a reference KAT: constants are unpinned)
    assert z[:4].tolist() == sm.poseidon2(np.zeros(16, np.uint32))[:4].tolist()


def test_merkle_commit_structure():
    rng = np.random.default_rng(3)
    H, W = 16, 11
    m = rng.integers(0, P, W * H, dtype=np.uint32)
    root, dig = sm.merkle_commit(m, H, W, want_digests=True)
    dig = dig.reshape(-1, 8)
    assert (dig[-1] == root).all()
    # parent = first 8 words of permute(left || right)
    st = np.concatenate([dig[0], dig[1]])
    assert (sm.poseidon2(st)[:8] == dig[H]).all()
    # leaf = overwrite-mode sponge over the row, rate 8
    row = m.reshape(W, H)[:, 0]
    s = np.zeros(16, np.uint32)
    s[:8] = row[:8]
    s = sm.poseidon2(s)
    s[:3] = row[8:11]
    s = sm.poseidon2(s)
    assert (s[:8] == dig[0]).all()


def synthetic_trace(shape, num_calls, seed):
    from tests.test_oracle_apc import run_oracle_gpu_convention

    s = synth.generate(shape, seed=seed)
    apc, idx, trace, hist, _ = run_oracle_gpu_convention(s, num_calls, seed=seed)
    return s, apc, idx, trace


@pytest.mark.parametrize("shape,calls", [("T0", 7), ("T0", 64), ("T1", 100)])
def test_prove_verify_accepts_and_rejects(shape, calls):
    s, apc, idx, trace = synthetic_trace(shape, calls, seed=4)
    W, H = trace.shape
    log_h = H.bit_length() - 1
    bc, spans = sm.compile_constraints(apc, idx)
    flat = np.ascontiguousarray(trace).reshape(-1)
    proof = sm.prove(flat, W, log_h, bc, spans, num_queries=6)
    assert sm.verify(proof, W, log_h, bc, spans, num_queries=6) == 0
    # deterministic
    assert (sm.prove(flat, W, log_h, bc, spans, num_queries=6) == proof).all()
    # any flipped word is rejected
    rng = np.random.default_rng(5)
    for pos in rng.choice(len(proof), size=12, replace=False):
        bad = proof.copy()
        bad[pos] = (int(bad[pos]) + 1) % P
        assert sm.verify(bad, W, log_h, bc, spans, num_queries=6) != 0
    # a trace that violates a constraint does not yield an accepting proof
    bad_trace = flat.copy()
    valid_col = idx[[p for p, k in s.kinds.items() if k[0] == "valid"][0]]
    bad_trace[valid_col * H] = 2  # is_valid * (is_valid - 1) != 0
    bad_proof = sm.prove(bad_trace, W, log_h, bc, spans, num_queries=6)
    assert sm.verify(bad_proof, W, log_h, bc, spans, num_queries=6) != 0


def test_proof_of_work():
    s, apc, idx, trace = synthetic_trace("T0", 5, seed=6)
    W, H = trace.shape
    bc, spans = sm.compile_constraints(apc, idx)
    flat = np.ascontiguousarray(trace).reshape(-1)
    proof = sm.prove(flat, W, 3, bc, spans, num_queries=3, pow_bits=8)
    assert sm.verify(proof, W, 3, bc, spans, num_queries=3, pow_bits=8) == 0
    assert sm.verify(proof, W, 3, bc, spans, num_queries=3, pow_bits=0) != 0


@pytest.mark.parametrize("shape,calls,pow_bits", [("T0", 7, 0), ("T0", 33, 6), ("T1", 200, 0)])
def test_product_verifier_agrees_with_oracle_verifier(shape, calls, pow_bits):
    """pw_verify (host code of libpowdr_gpu, Montgomery arithmetic, no GPU) against the oracle's
    verifier: accepts the oracle prover's proofs, rejects every tampering the oracle rejects, with
    the same verdict code."""
    from powdr_amd import prover

    s, apc, idx, trace = synthetic_trace(shape, calls, seed=8)
    W, H = trace.shape
    log_h = H.bit_length() - 1
    bc, spans = sm.compile_constraints(apc, idx)
    flat = np.ascontiguousarray(trace).reshape(-1)
    proof = sm.prove(flat, W, log_h, bc, spans, num_queries=7, pow_bits=pow_bits)
    assert prover.verify(proof, W, log_h, bc, spans, num_queries=7, pow_bits=pow_bits) == 0
    rng = np.random.default_rng(9)
    for pos in list(rng.choice(len(proof), size=40, replace=False)) + [0, 5, 6, len(proof) - 1]:
        bad = proof.copy()
        bad[pos] = (int(bad[pos]) + 1) % P
        a = sm.verify(bad, W, log_h, bc, spans, num_queries=7, pow_bits=pow_bits)
        b = prover.verify(bad, W, log_h, bc, spans, num_queries=7, pow_bits=pow_bits)
        assert a != 0 and a == b, (pos, a, b)
    assert prover.verify(proof[:-3], W, log_h, bc, spans, num_queries=7, pow_bits=pow_bits) == 10
    assert prover.verify(np.concatenate([proof, proof[:1]]), W, log_h, bc, spans, num_queries=7, pow_bits=pow_bits) == 9
    assert prover.verify(proof, W, log_h, bc, spans, num_queries=6, pow_bits=pow_bits) == 1


@pytest.mark.parametrize("shape,calls", [("T0", 5), ("T0", 64), ("T1", 90)])
def test_logup_extension_accepts_and_rejects(shape, calls):
    """pw-stark v0 + LogUp (bus interactions inside the proof): honest proofs verify; tampering, a broken
    constraint, and a perturbed interaction operand (which breaks q_i * d_i = m_i) are rejected."""
    from powdr_amd import prover

    s, apc, idx, trace = synthetic_trace(shape, calls, seed=12)
    W, H = trace.shape
    log_h = H.bit_length() - 1
    bc, spans = sm.compile_constraints(apc, idx)
    inter, ispans, ibc = sm.compile_interactions(apc, idx)
    flat = np.ascontiguousarray(trace).reshape(-1)
    proof = sm.prove_logup(flat, W, log_h, bc, spans, inter, ispans, ibc, num_queries=5)
    assert sm.verify_logup(proof, W, log_h, bc, spans, inter, ispans, ibc, num_queries=5) == 0
    assert (sm.prove_logup(flat, W, log_h, bc, spans, inter, ispans, ibc, num_queries=5) == proof).all()
    rng = np.random.default_rng(3)
    for pos in rng.choice(len(proof), size=16, replace=False):
        bad = proof.copy()
        bad[pos] = (int(bad[pos]) + 1) % P
        assert sm.verify_logup(bad, W, log_h, bc, spans, inter, ispans, ibc, num_queries=5) != 0
    # the product's host verifier agrees with the oracle's on every one of these, code for code
    rc, S = prover.verify_logup(proof, W, log_h, bc, spans, (inter, ispans, ibc), num_queries=5)
    assert rc == 0 and (S == proof[31:35]).all()  # header 7 | trace root 8 | bus seed 8 | perm root 8 | S
    assert (proof[7:15] == proof[15:23]).all()     # a lone AIR seeds its bus challenges with its own trace root
    for pos in list(rng.choice(len(proof), size=30, replace=False)) + [0, 4, 7, 15, 23, 31, len(proof) - 1]:
        bad = proof.copy()
        bad[pos] = (int(bad[pos]) + 1) % P
        a = sm.verify_logup(bad, W, log_h, bc, spans, inter, ispans, ibc, num_queries=5)
        b = prover.verify_logup(bad, W, log_h, bc, spans, (inter, ispans, ibc), num_queries=5)[0]
        assert a != 0 and a == b, (pos, a, b)
    assert prover.verify_logup(proof[:-2], W, log_h, bc, spans, (inter, ispans, ibc), num_queries=5)[0] == 10
    assert prover.verify(proof, W, log_h, bc, spans, num_queries=5) == 1  # a PWS2 proof is not a PWS1 proof
    # a verifier that believes in a different interaction list rejects the proof
    inter2 = inter.copy()
    inter2[0, 0] = (int(inter2[0, 0]) + 1) % 16
    assert sm.verify_logup(proof, W, log_h, bc, spans, inter2, ispans, ibc, num_queries=5) != 0
    assert prover.verify_logup(proof, W, log_h, bc, spans, (inter2, ispans, ibc), num_queries=5)[0] == 2
    # the proof is longer than the constraints-only one by the permutation matrix openings
    base = sm.prove(flat, W, log_h, bc, spans, num_queries=5)
    assert len(proof) > len(base) + 8 * (len(inter) + 1)


def balanced_bus_pair(log_h, seed):
    """Two 3-column AIRs over one bus: the first sends (a, b) with multiplicity m on each row, the second
    receives the same tuples in a different row order (multiplicity -m)."""
    rng = np.random.default_rng(seed)
    H = 1 << log_h
    m = rng.integers(0, 4, H).astype(np.uint32)
    a = rng.integers(0, P, H, dtype=np.uint32)
    b = rng.integers(0, 1 << 16, H).astype(np.uint32)
    perm = rng.permutation(H)
    t1 = np.stack([m, a, b])
    t2 = np.stack([m[perm], a[perm], b[perm]])
    PA = om.OP_PUSH_APC
    send = (np.array([[5, 2, 0]], np.uint32), np.array([[0, 2], [2, 2], [4, 2]], np.uint32), np.array([PA, 0, PA, 1, PA, 2], np.uint32))
    recv = (np.array([[5, 2, 0]], np.uint32), np.array([[0, 3], [3, 2], [5, 2]], np.uint32),
            np.array([PA, 0, om.OP_NEG, PA, 1, PA, 2], np.uint32))
    return (t1, send), (t2, recv)


def balanced_bus_pair_mixed(log_h_send, log_h_recv, seed):
    """Like balanced_bus_pair with DIFFERENT heights: the sender's tuples are received by an AIR of another height —
    the receiver aggregates duplicates into multiplicities (more rows than needed are padded with multiplicity 0),
    or splits a tuple over several rows when it is taller."""
    rng = np.random.default_rng(seed)
    Hs, Hr = 1 << log_h_send, 1 << log_h_recv
    n_distinct = min(Hs, Hr)
    a = rng.integers(0, P, n_distinct, dtype=np.uint32)
    b = rng.integers(0, 1 << 16, n_distinct).astype(np.uint32)
    # sender: row r sends tuple (r mod n_distinct) with multiplicity m_s[r]
    m_s = rng.integers(0, 4, Hs).astype(np.uint32)
    t1 = np.stack([m_s, a[np.arange(Hs) % n_distinct], b[np.arange(Hs) % n_distinct]])
    total = np.zeros(n_distinct, np.int64)
    np.add.at(total, np.arange(Hs) % n_distinct, m_s)
    # receiver: tuple k's total multiplicity is spread over the rows r = k mod n_distinct (all on the first such row)
    m_r = np.zeros(Hr, np.uint32)
    m_r[:n_distinct] = total.astype(np.uint32)
    perm = rng.permutation(Hr)
    t2 = np.stack([m_r, a[np.arange(Hr) % n_distinct], b[np.arange(Hr) % n_distinct]])[:, perm]
    PA = om.OP_PUSH_APC
    send = (np.array([[5, 2, 0]], np.uint32), np.array([[0, 2], [2, 2], [4, 2]], np.uint32), np.array([PA, 0, PA, 1, PA, 2], np.uint32))
    recv = (np.array([[5, 2, 0]], np.uint32), np.array([[0, 3], [3, 2], [5, 2]], np.uint32),
            np.array([PA, 0, om.OP_NEG, PA, 1, PA, 2], np.uint32))
    return (t1, send), (t2, recv)


def ext_add_canonical(x, y):
    return (x.astype(np.uint64) + y.astype(np.uint64)) % P


def test_logup_cumulative_sums_balance_across_airs():
    """The point of LogUp: a segment's bus is balanced iff the per-AIR cumulative sums add up to zero."""
    from powdr_amd import prover

    from powdr_amd import sharding

    no_cons = (np.zeros(0, np.uint32), np.zeros((0, 2), np.uint32))
    (t1, send), (t2, recv) = balanced_bus_pair(5, seed=4)

    def segment(airs):
        """prove every AIR against the seed formed from all trace roots; verify as the segment verifier would"""
        roots = [sm.prove_logup(t.reshape(-1), 3, 5, *no_cons, *it, num_queries=0)[7:15] for t, it in airs]  # phase 1: commitments
        seed = sharding.commitment_digest(np.array(roots))
        sums, seen_roots = [], []
        for t, it in airs:
            proof = sm.prove_logup(t.reshape(-1), 3, 5, *no_cons, *it, num_queries=4, bus_seed=seed)
            assert sm.verify_logup(proof, 3, 5, *no_cons, *it, num_queries=4, bus_seed=seed) == 0
            assert sm.verify_logup(proof, 3, 5, *no_cons, *it, num_queries=4) == 12  # not a lone-AIR proof
            rc, S, root = prover.verify_logup(proof, 3, 5, *no_cons, it, num_queries=4, bus_seed=seed, with_root=True)
            assert rc == 0
            assert prover.verify_logup(proof, 3, 5, *no_cons, it, num_queries=4, bus_seed=seed ^ 1)[0] == 12
            sums.append(S)
            seen_roots.append(root)
        assert (sharding.commitment_digest(np.array(seen_roots)) == seed).all()  # the seed binds every trace
        return sums

    s1, s2 = segment([(t1, send), (t2, recv)])
    assert s1.any() and (ext_add_canonical(s1, s2) == 0).all()
    # an unmatched send (one tuple changed on the receiving side) leaves a non-zero total
    t2b = t2.copy()
    row = int(np.argmax(t2b[0] != 0))
    t2b[2, row] ^= 1
    s1, s2 = segment([(t1, send), (t2b, recv)])
    assert (ext_add_canonical(s1, s2) != 0).any()
    # proofs made with private seeds (each AIR's own root) are individually valid but their sums do not cancel
    lone = [prover.verify_logup(sm.prove_logup(t.reshape(-1), 3, 5, *no_cons, *it, num_queries=4), 3, 5, *no_cons, it, num_queries=4)
            for t, it in ((t1, send), (t2, recv))]
    assert lone[0][0] == 0 and lone[1][0] == 0 and (ext_add_canonical(lone[0][1], lone[1][1]) != 0).any()


def test_commitment_digest_matches_restatement():
    from powdr_amd import prover, sharding

    rng = np.random.default_rng(0)
    for n in (0, 1, 2, 3, 5, 8, 13):
        r = rng.integers(0, P, (n, 8), dtype=np.uint32)
        assert (prover.commitment_digest(r) == sm.commitment_digest(r)).all()
        assert (sharding.commitment_digest(r) == sm.commitment_digest(r)).all()
    one = rng.integers(0, P, (1, 8), dtype=np.uint32)
    assert (prover.commitment_digest(one) == one[0]).all()


def test_verify_segment_on_oracle_proofs():
    """pw_verify_segment (host): recomputes the bus seed from the trace roots inside the proofs, verifies every
    proof against it, adds up the cumulative sums."""
    from powdr_amd import prover

    no_cons = (np.zeros(0, np.uint32), np.zeros((0, 2), np.uint32))
    (t1, send), (t2, recv) = balanced_bus_pair(4, seed=9)

    def proofs_for(airs):
        roots = [sm.prove_logup(t.reshape(-1), 3, 4, *no_cons, *it, num_queries=0)[7:15] for t, it in airs]
        seed = sm.commitment_digest(np.array(roots))
        return [sm.prove_logup(t.reshape(-1), 3, 4, *no_cons, *it, num_queries=5, bus_seed=seed) for t, it in airs]

    descs = [(3, 4, *no_cons, send), (3, 4, *no_cons, recv)]
    proofs = proofs_for([(t1, send), (t2, recv)])
    rc, total = prover.verify_airs(descs, proofs, num_queries=5, shared_bus_seed=True, check_balance=True)
    assert rc == 0 and (total == 0).all()
    # the order of the AIRs is part of the seed
    assert prover.verify_airs(descs[::-1], proofs[::-1], num_queries=5, shared_bus_seed=True)[0] == (1 << 8) | 12
    # a tampered opening in the second proof
    bad = [proofs[0], proofs[1].copy()]
    bad[1][40] = (int(bad[1][40]) + 1) % P
    assert prover.verify_airs(descs, bad, num_queries=5, shared_bus_seed=True)[0] >> 8 == 2
    # an unmatched tuple: every proof is valid, the segment is not balanced
    t2b = t2.copy()
    t2b[1, int(np.argmax(t2b[0] != 0))] ^= 1
    proofs = proofs_for([(t1, send), (t2b, recv)])
    assert prover.verify_airs(descs, proofs, num_queries=5, shared_bus_seed=True, check_balance=False)[0] == 0
    rc, total = prover.verify_airs(descs, proofs, num_queries=5, shared_bus_seed=True, check_balance=True)
    assert rc == 14 and total.any()
    # constraints-only proofs ("PWS1") go through the same entry point
    s, apc, idx, trace = synthetic_trace("T0", 9, seed=3)
    W, H = trace.shape
    bc, spans = sm.compile_constraints(apc, idx)
    pf = sm.prove(np.ascontiguousarray(trace).reshape(-1), W, H.bit_length() - 1, bc, spans, num_queries=5)
    assert prover.verify_airs([(W, H.bit_length() - 1, bc, spans, None)] * 2, [pf, pf], num_queries=5)[0] == 0
    assert prover.verify_airs([(W, H.bit_length() - 1, bc, spans, None)], [pf[:-1]], num_queries=5)[0] == (1 << 8) | 10


def test_logup_grouping_is_the_same_in_oracle_and_product():
    """Which interactions share a committed column is part of the protocol: prover, verifier and oracle must agree.
    Degree-1 arguments pair up (q d1 d2 has degree 3); a degree-2 argument or a degree-3 multiplicity stays alone."""
    from powdr_amd import prover

    for shape, seed in (("T0", 1), ("T1", 2), ("C1", 3)):
        s, apc, idx, trace = synthetic_trace(shape, 4, seed=seed)
        it = sm.compile_interactions(apc, idx)
        a, b = sm.group_starts(*it), prover.logup_group_starts(it)
        assert (a == b).all() and a[0] == 0 and a[-1] == len(it[0])
        assert (np.diff(a) >= 1).all() and (np.diff(a) <= 2).all() and len(a) - 1 < len(it[0])
    PA, PC, MUL = om.OP_PUSH_APC, om.OP_PUSH_CONST, om.OP_MUL
    lin, quad, const = [PA, 0], [PA, 0, PA, 1, MUL], [PC, 7]
    cube = [PA, 0, PA, 1, MUL, PA, 2, MUL]

    def table(rows):  # rows: (mult program, [arg programs])
        inter, spans, bc = [], [], []
        for m, args in rows:
            inter.append((3, len(args), len(spans)))
            for prog in [m] + args:
                spans.append((len(bc), len(prog)))
                bc += prog
        return np.array(inter, np.uint32), np.array(spans, np.uint32), np.array(bc, np.uint32)

    cases = [([(lin, [lin]), (lin, [lin]), (lin, [lin])], [0, 2, 3]),          # pairs
             ([(lin, [quad]), (lin, [lin])], [0, 1, 2]),                       # 1 + 2 + 1 > 3
             ([(lin, [const]), (lin, [const]), (quad, [lin, const]), (lin, [lin])], [0, 4]),  # constant denominators are free
             ([(cube, [lin]), (lin, [lin])], [0, 1, 2]),                        # m1 d2 would have degree 4
             ([(quad, [lin]), (quad, [lin])], [0, 2]),                          # 2 + 1 = 3
             ([], [0])]
    for rows, want in cases:
        t = table(rows)
        assert sm.group_starts(*t).tolist() == want and prover.logup_group_starts(t).tolist() == want


def test_golden_proofs_pin_the_protocol():
    """tests/golden/pw_stark_proofs_T0.npz (made by tests/golden/make_proof_golden.py): the oracle still produces these
    exact words from the stored trace, and both verifiers accept them — any drift of the transcript, the hash, the
    field conventions or the proof layout shows up here, on a machine without a GPU."""
    from pathlib import Path

    from powdr_amd import prover

    z = np.load(Path(__file__).parent / "golden" / "pw_stark_proofs_T0.npz")
    W, log_h = int(z["width"]), int(z["log_h"])
    cons = (z["cons_bc"], z["cons_spans"])
    it = (z["inter"], z["inter_spans"], z["inter_bc"])
    assert (sm.prove(z["trace"], W, log_h, *cons, num_queries=4, pow_bits=5) == z["proof_v0"]).all()
    assert (sm.prove_logup(z["trace"], W, log_h, *cons, *it, num_queries=4, pow_bits=5) == z["proof_logup"]).all()
    assert sm.verify(z["proof_v0"], W, log_h, *cons, num_queries=4, pow_bits=5) == 0
    assert prover.verify(z["proof_v0"], W, log_h, *cons, num_queries=4, pow_bits=5) == 0
    assert sm.verify_logup(z["proof_logup"], W, log_h, *cons, *it, num_queries=4, pow_bits=5) == 0
    assert prover.verify_logup(z["proof_logup"], W, log_h, *cons, it, num_queries=4, pow_bits=5)[0] == 0
    assert z["proof_v0"][0] == 0x31535750 and z["proof_logup"][0] == 0x32535750
