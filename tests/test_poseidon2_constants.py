"""The hash parameters are one table installed at run time: pw_set_poseidon2_constants and the oracle's
twin or_set_poseidon2_constants. The reference's round constants live in the un-vendored p3-baby-bear 0.5.2
(/root/reference/number/Cargo.toml:16-19); loading them through this entry makes the permutation the reference's, and
pw_poseidon2_permute_host is where a maintainer checks that crate's known-answer vector. Here: the signed-representative
device/host form (csrc/poseidon2.hpp, every derived table rebuilt) equals the oracle's plain `% p` form under arbitrary and
extreme constant sets, proofs are byte-identical under a second set, and proofs do not verify across sets."""
import numpy as np
import pytest

from oracle import apc_model as om
from oracle import stark_model as sm

P = om.P


@pytest.fixture
def second_set():
    from powdr_amd import prover

    rng = np.random.default_rng(0xC0FFEE)
    E, I = rng.integers(0, P, (8, 16), dtype=np.uint32), rng.integers(0, P, 13, dtype=np.uint32)
    prover.set_poseidon2_constants(E, I)
    sm.set_poseidon2_constants(E, I)
    try:
        yield E, I
    finally:
        prover.set_poseidon2_constants()
        sm.set_poseidon2_constants()


def _edge_states(rng):
    yield np.zeros(16, np.uint32)
    yield np.full(16, P - 1, np.uint32)
    yield np.full(16, (P - 1) // 2, np.uint32)
    for _ in range(40):
        yield rng.integers(0, P, 16, dtype=np.uint32)


def test_host_permutation_equals_oracle_under_any_constants():
    """Random tables and the extreme ones (all 0, all p - 1, both neighbours of p / 2 — the largest centred magnitudes the
    bounds of tools/poseidon2_bounds.py allow for an additive constant): signed host form == oracle."""
    from powdr_amd import prover

    rng = np.random.default_rng(5)
    tables = [(np.full((8, 16), v, np.uint32), np.full(13, v, np.uint32)) for v in (0, P - 1, (P - 1) // 2, (P + 1) // 2, 1)]
    tables += [(rng.integers(0, P, (8, 16), dtype=np.uint32), rng.integers(0, P, 13, dtype=np.uint32)) for _ in range(6)]
    try:
        for E, I in tables:
            prover.set_poseidon2_constants(E, I)
            sm.set_poseidon2_constants(E, I)
            e, i, d = prover.poseidon2_constants()
            assert (e == E).all() and (i == I).all() and (d == sm.poseidon2_constants()[2]).all()
            for s in _edge_states(rng):
                assert (prover.poseidon2_host(s) == sm.poseidon2(s)).all()
    finally:
        prover.set_poseidon2_constants()
        sm.set_poseidon2_constants()
    with pytest.raises(ValueError):
        prover.set_poseidon2_constants(np.full((8, 16), P, np.uint32), np.zeros(13, np.uint32))
    z = np.zeros(16, np.uint32)
    assert (prover.poseidon2_host(z) == sm.poseidon2(z)).all()  # the placeholder is back


def test_proofs_do_not_verify_across_constant_sets(second_set):
    """Oracle proof under the second table: accepted by the product's host verifier under the same table, rejected under the
    placeholder (Merkle paths and transcript change)."""
    from powdr_amd import prover
    from tests.test_prover_gpu import _synthetic

    s, flat, (W, H), bc, spans = _synthetic("T0", 20, seed=3)
    log_h = H.bit_length() - 1
    pf = sm.prove(flat, W, log_h, bc, spans, num_queries=4, pow_bits=2)
    assert prover.verify(pf, W, log_h, bc, spans, 4, 2) == 0 and sm.verify(pf, W, log_h, bc, spans, 4, 2) == 0
    prover.set_poseidon2_constants()
    assert prover.verify(pf, W, log_h, bc, spans, 4, 2) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("logup", [False, True])
def test_hip_proof_bytes_under_a_second_constant_set(second_set, logup):
    """Device kernels (leaf hash, compress, FRI trees, proof of work) under the second table: single-AIR and segment proofs
    equal the oracle's; back on the placeholder the golden behaviour returns."""
    import torch
    from powdr_amd import prover
    from tests.test_prover_gpu import _synthetic, to_dev
    from tests.test_segment_proof import hip_segment, synthetic_airs

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU (run with -m gpu on the GPU box)")
    s, flat, (W, H), bc, spans = _synthetic("T1", 900, seed=4)
    log_h = H.bit_length() - 1
    apc = om.load_apc(s.doc)
    it = sm.compile_interactions(apc, apc.poly_id_to_index()) if logup else None
    pr = prover.Prover(W, bc, spans, num_queries=5, pow_bits=6, interactions=it)
    got = pr.prove(to_dev(torch, flat).data_ptr(), log_h)
    want = sm.prove_logup(flat, W, log_h, bc, spans, *it, num_queries=5, pow_bits=6) if logup else sm.prove(flat, W, log_h, bc, spans, num_queries=5, pow_bits=6)
    assert len(got) == len(want) and (got == want).all()
    pr.close()
    airs = synthetic_airs([("T0", 30), ("T1", 200), ("T0", 5)], seed0=70)
    seg = hip_segment((torch, None, prover), airs, 4, 3, logup)
    assert (seg == sm.prove_segment(airs, num_queries=4, pow_bits=3, logup=logup)).all()
    # the placeholder table again: a different proof, and the oracle agrees there too
    prover.set_poseidon2_constants()
    sm.set_poseidon2_constants()
    seg0 = hip_segment((torch, None, prover), airs, 4, 3, logup)
    assert (seg0 != seg).any() and (seg0 == sm.prove_segment(airs, num_queries=4, pow_bits=3, logup=logup)).all()
