"""ctypes binding of the pw-stark v0 prover entry points (include/powdr_prover.h)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import abi

lib = abi.lib


class PwStarkConfig(C.Structure):
    _fields_ = [("num_queries", C.c_uint32), ("pow_bits", C.c_uint32)]


PROVER_SYMBOLS = ["pw_prover_check_constraints", "pw_verify", "pw_prover_create", "pw_prover_create_logup", "pw_verify_logup", "pw_prover_trace_root", "pw_prover_set_bus_seed", "pw_prover_logup_path", "pw_prove_airs", "pw_verify_airs", "pw_prove_segment", "pw_verify_segment", "pw_commitment_digest", "pw_logup_group_starts", "pw_prover_width", "pw_prover_reserve", "pw_prover_stream_log_blocks", "pw_prover_max_constraint_degree", "pw_prover_destroy", "pw_prover_prove", "pw_prover_prove_consuming", "pw_prover_stream_log_blocks_consuming", "pw_trace_from_coefficients", "pw_prover_device_bytes",
                  "pw_lde_batch", "pw_lde_fused", "pw_lde_subcoset", "pw_merkle_commit", "pw_poseidon2_permute_host",
                  "pw_set_poseidon2_constants", "pw_get_poseidon2_constants", "pw_prover_specialise", "pw_prover_specialised", "pw_jit_compile_check", "pw_jit_cache_stats", "pw_jit_generated_source",
                  "pw_prove_segments_multi", "pw_multi_last_merge", "pw_assign_units",
                  "pw_prove_segment_consuming", "pw_segment_last_modes", "pw_segment_last_plan", "pw_set_device_budget", "pw_get_device_budget", "pw_provers_specialise"]

lib.pw_prover_create.restype = C.c_void_p
lib.pw_prover_create.argtypes = [C.POINTER(PwStarkConfig), C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
lib.pw_prover_create_logup.restype = C.c_void_p
lib.pw_prover_create_logup.argtypes = [C.POINTER(PwStarkConfig), C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                       C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
lib.pw_prover_destroy.argtypes = [C.c_void_p]
lib.pw_prover_prove.restype = C.c_int
lib.pw_prover_prove.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_size_t)]
lib.pw_prover_check_constraints.restype = C.c_int
lib.pw_prover_check_constraints.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_uint32)]
lib.pw_prover_device_bytes.restype = C.c_size_t
lib.pw_prover_device_bytes.argtypes = [C.c_void_p]
lib.pw_lde_batch.restype = C.c_int
lib.pw_lde_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
lib.pw_lde_fused.restype = C.c_int
lib.pw_lde_fused.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
lib.pw_prover_prove_consuming.restype = C.c_int
lib.pw_prover_prove_consuming.argtypes = lib.pw_prover_prove.argtypes
lib.pw_prover_stream_log_blocks_consuming.restype = C.c_int
lib.pw_prover_stream_log_blocks_consuming.argtypes = [C.c_void_p, C.c_uint32]
lib.pw_trace_from_coefficients.restype = C.c_int
lib.pw_trace_from_coefficients.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
lib.pw_lde_subcoset.restype = C.c_int
lib.pw_lde_subcoset.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
lib.pw_segment_last_modes.restype = C.c_size_t
lib.pw_segment_last_modes.argtypes = [C.c_void_p, C.c_size_t]
lib.pw_set_device_budget.restype = None
lib.pw_set_device_budget.argtypes = [C.c_size_t]
lib.pw_get_device_budget.restype = C.c_size_t
lib.pw_merkle_commit.restype = C.c_int
lib.pw_merkle_commit.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
lib.pw_poseidon2_permute_host.argtypes = [C.c_void_p]


lib.pw_verify.restype = C.c_int
lib.pw_verify.argtypes = [C.POINTER(PwStarkConfig), C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                          C.c_void_p, C.c_size_t]


def verify(proof, width: int, log_height: int, cons_bytecode, cons_spans, num_queries: int = 100, pow_bits: int = 0) -> int:
    """Host-side verification (no GPU). 0 = valid, otherwise the code of the first failed check."""
    pr = np.ascontiguousarray(proof, dtype=np.uint32)
    bc = np.ascontiguousarray(cons_bytecode, dtype=np.uint32)
    sp = np.ascontiguousarray(cons_spans, dtype=np.uint32).reshape(-1, 2)
    cfg = PwStarkConfig(num_queries, pow_bits)
    return int(lib.pw_verify(C.byref(cfg), width, log_height, bc.ctypes.data_as(C.c_void_p), len(bc),
                             sp.ctypes.data_as(C.c_void_p), len(sp), pr.ctypes.data_as(C.c_void_p), len(pr)))


lib.pw_verify_logup.restype = C.c_int
lib.pw_verify_logup.argtypes = [C.POINTER(PwStarkConfig), C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                C.c_void_p, C.c_void_p]
lib.pw_prover_trace_root.restype = C.c_int
lib.pw_prover_trace_root.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
lib.pw_prover_logup_path.restype = C.c_int
lib.pw_prover_logup_path.argtypes = [C.c_void_p]
lib.pw_prover_set_bus_seed.restype = C.c_int
lib.pw_prover_set_bus_seed.argtypes = [C.c_void_p, C.c_void_p]


def verify_logup(proof, width: int, log_height: int, cons_bytecode, cons_spans, interactions, num_queries: int = 100,
                 pow_bits: int = 0, bus_seed=None, with_root: bool = False):
    """Host-side verification of a LogUp proof: (code, cumulative bus sum S as 4 canonical words or None)
    [, trace root]. bus_seed: the 8 words the proof must have drawn its bus challenges from (None: its own
    trace root)."""
    pr = np.ascontiguousarray(proof, dtype=np.uint32)
    bc = np.ascontiguousarray(cons_bytecode, dtype=np.uint32)
    sp = np.ascontiguousarray(cons_spans, dtype=np.uint32).reshape(-1, 2)
    it = np.ascontiguousarray(interactions[0], dtype=np.uint32).reshape(-1, 3)
    isp = np.ascontiguousarray(interactions[1], dtype=np.uint32).reshape(-1, 2)
    ibc = np.ascontiguousarray(interactions[2], dtype=np.uint32)
    cfg = PwStarkConfig(num_queries, pow_bits)
    s, root = np.zeros(4, np.uint32), np.zeros(8, np.uint32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    seed = None if bus_seed is None else np.ascontiguousarray(bus_seed, dtype=np.uint32)
    rc = int(lib.pw_verify_logup(C.byref(cfg), width, log_height, p(bc), len(bc), p(sp), len(sp), p(it), len(it), p(isp), len(isp),
                                 p(ibc), len(ibc), None if seed is None else p(seed), p(pr), len(pr), p(s), p(root)))
    if with_root:
        return rc, (s if rc == 0 else None), (root if rc == 0 else None)
    return rc, (s if rc == 0 else None)


class PwSegmentAir(C.Structure):
    _fields_ = [("prover", C.c_void_p), ("d_trace", C.c_void_p), ("log_height", C.c_uint32), ("flags", C.c_uint32)]


PW_AIR_HAND_OVER = 1


class PwAirDescription(C.Structure):
    _fields_ = [("width", C.c_uint32), ("log_height", C.c_uint32), ("logup", C.c_uint32),
                ("cons_bytecode", C.c_void_p), ("bytecode_len", C.c_size_t), ("cons_spans", C.c_void_p), ("n_constraints", C.c_size_t),
                ("interactions", C.c_void_p), ("n_interactions", C.c_size_t), ("inter_spans", C.c_void_p), ("n_inter_spans", C.c_size_t),
                ("inter_bytecode", C.c_void_p), ("inter_bytecode_len", C.c_size_t)]


lib.pw_prove_airs.restype = C.c_int
lib.pw_prove_airs.argtypes = [C.POINTER(PwSegmentAir), C.c_size_t, C.c_int, C.c_uint, C.POINTER(C.POINTER(C.c_uint32)),
                                 C.POINTER(C.c_size_t), C.c_void_p]
lib.pw_prove_segment.restype = C.c_int
lib.pw_prove_segment.argtypes = [C.POINTER(PwSegmentAir), C.c_size_t, C.c_int, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_size_t)]
lib.pw_prove_segment_consuming.restype = C.c_int
lib.pw_prove_segment_consuming.argtypes = lib.pw_prove_segment.argtypes
lib.pw_verify_segment.restype = C.c_int
lib.pw_verify_segment.argtypes = [C.POINTER(PwStarkConfig), C.POINTER(PwAirDescription), C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
lib.pw_verify_airs.restype = C.c_int
lib.pw_verify_airs.argtypes = [C.POINTER(PwStarkConfig), C.POINTER(PwAirDescription), C.c_size_t, C.POINTER(C.c_void_p),
                                  C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_void_p]
lib.pw_commitment_digest.restype = None
lib.pw_commitment_digest.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]


lib.pw_logup_group_starts.restype = C.c_size_t
lib.pw_logup_group_starts.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]


def logup_group_starts(interactions) -> np.ndarray:
    """Boundaries of the LogUp groups (one committed extension column each) for an interaction table."""
    it = np.ascontiguousarray(interactions[0], dtype=np.uint32).reshape(-1, 3)
    isp = np.ascontiguousarray(interactions[1], dtype=np.uint32).reshape(-1, 2)
    ibc = np.ascontiguousarray(interactions[2], dtype=np.uint32)
    out = np.zeros(len(it) + 2, np.uint32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    k = lib.pw_logup_group_starts(p(it), len(it), p(isp), len(isp), p(ibc), len(ibc), p(out), len(out))
    if k == 0:
        raise ValueError("malformed interaction table")
    return out[:k].copy()


def commitment_digest(roots) -> np.ndarray:
    """Digest over an ordered list of 8-word commitments (canonical words)."""
    r = np.ascontiguousarray(roots, dtype=np.uint32).reshape(-1, 8)
    out = np.zeros(8, np.uint32)
    lib.pw_commitment_digest(r.ctypes.data_as(C.c_void_p), len(r), out.ctypes.data_as(C.c_void_p))
    return out


def prove_airs(airs, shared_bus_seed: bool = False, n_workers: int = 0, copy: bool = True):
    """airs: [(Prover, device trace pointer, log_height)] -> ([proof words per AIR], bus seed).
    INDEPENDENT proofs, one per AIR, on `n_workers` host threads / HIP streams (0 = 4) — the flow AIR-level sharding
    over ranks uses; `prove_segment` is the one-proof-per-segment form."""
    n = len(airs)
    recs = (PwSegmentAir * max(n, 1))()
    for i, (pr, ptr, lh) in enumerate(airs):
        recs[i] = PwSegmentAir(pr._h, ptr, lh)
    proofs = (C.POINTER(C.c_uint32) * max(n, 1))()
    lens = (C.c_size_t * max(n, 1))()
    seed = np.zeros(8, np.uint32)
    rc = lib.pw_prove_airs(recs, n, int(shared_bus_seed), n_workers, proofs, lens, seed.ctypes.data_as(C.c_void_p))
    abi.check(rc, "pw_prove_airs")
    out = []
    for i in range(n):
        a = np.ctypeslib.as_array(proofs[i], shape=(lens[i],))
        out.append(a.copy() if copy else a)
    return out, seed


def _air_descriptions(descs):
    n = len(descs)
    keep, recs = [], (PwAirDescription * max(n, 1))()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    for i, (w, lh, bc, sp, it) in enumerate(descs):
        bc = np.ascontiguousarray(bc, dtype=np.uint32)
        sp = np.ascontiguousarray(sp, dtype=np.uint32).reshape(-1, 2)
        keep += [bc, sp]
        if it is None:
            recs[i] = PwAirDescription(w, lh, 0, vp(bc), len(bc), vp(sp), len(sp), None, 0, None, 0, None, 0)
        else:
            a = np.ascontiguousarray(it[0], dtype=np.uint32).reshape(-1, 3)
            b = np.ascontiguousarray(it[1], dtype=np.uint32).reshape(-1, 2)
            c = np.ascontiguousarray(it[2], dtype=np.uint32)
            keep += [a, b, c]
            recs[i] = PwAirDescription(w, lh, 1, vp(bc), len(bc), vp(sp), len(sp), vp(a), len(a), vp(b), len(b), vp(c), len(c))
    return recs, keep


def prove_segment(airs, logup: bool = False, copy: bool = True, hand_over=None) -> np.ndarray:
    """ONE proof for all AIRs of a segment (pw-stark v1, magic PWS3): airs = [(Prover, device trace pointer, log_height)],
    every Prover created with `interactions=` when logup. The counterpart of the reference's one engine call per segment.
    hand_over: None = pw_prove_segment; True / a list of bools = pw_prove_segment_consuming with those AIRs' traces handed over
    (a STREAMED AIR then leaves its coefficient arrays in the caller's buffer: segment_last_modes() says which did)."""
    n = len(airs)
    recs = (PwSegmentAir * max(n, 1))()
    flags = [0] * n if hand_over is None else ([PW_AIR_HAND_OVER] * n if hand_over is True else [PW_AIR_HAND_OVER if f else 0 for f in hand_over])
    for i, (pr, ptr, lh) in enumerate(airs):
        recs[i] = PwSegmentAir(pr._h, ptr, lh, flags[i])
    words = C.POINTER(C.c_uint32)()
    nw = C.c_size_t()
    if hand_over is None:
        abi.check(lib.pw_prove_segment(recs, n, int(logup), C.byref(words), C.byref(nw)), "pw_prove_segment")
    else:
        abi.check(lib.pw_prove_segment_consuming(recs, n, int(logup), C.byref(words), C.byref(nw)), "pw_prove_segment_consuming")
    a = np.ctypeslib.as_array(words, shape=(nw.value,))
    return a.copy() if copy else a


def segment_last_modes():
    """[(log2 sub-cosets, trace overwritten by its coefficients)] per AIR of this thread's last segment proof."""
    n = int(lib.pw_segment_last_modes(None, 0))
    out = (C.c_uint32 * max(n, 1))()
    lib.pw_segment_last_modes(out, n)
    return [(int(out[i]) & 0xFF, bool(int(out[i]) & 0x100)) for i in range(n)]


def segment_last_plan():
    """(bytes with every AIR resident, bytes as planned, bytes the policy had) of this thread's last segment proof."""
    a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
    lib.pw_segment_last_plan(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def specialise_all(provers) -> int:
    """pw_provers_specialise: the run-time specialised kernels of all `provers` (Prover objects) in one concurrent compile batch,
    whatever their traces' heights. Returns how many run specialised kernels afterwards."""
    n = len(provers)
    arr = (C.c_void_p * max(n, 1))(*[p._h for p in provers])
    lib.pw_provers_specialise.restype = C.c_size_t
    lib.pw_provers_specialise.argtypes = [C.c_void_p, C.c_size_t]
    return int(lib.pw_provers_specialise(arr, n))


def set_device_budget(n_bytes: int) -> None:
    """pw_set_device_budget: bytes the provers of this process may plan for (0 = whatever the device has free)."""
    lib.pw_set_device_budget(int(n_bytes))


def verify_segment(descs, proof, num_queries: int = 100, pow_bits: int = 0, logup: bool = False, check_balance: bool = False):
    """Host verification of a segment proof. descs: [(width, log_height, cons_bytecode, cons_spans, interactions-or-None)]
    -> (code, sum of the AIRs' cumulative bus sums). 0 = valid; ((i+1) << 8) | 2 = constraint identity of AIR i;
    14 = the bus sums do not cancel (check_balance)."""
    recs, keep = _air_descriptions(descs)
    pr = np.ascontiguousarray(proof, dtype=np.uint32)
    cfg = PwStarkConfig(num_queries, pow_bits)
    total = np.zeros(4, np.uint32)
    rc = int(lib.pw_verify_segment(C.byref(cfg), recs, len(descs), int(logup), pr.ctypes.data_as(C.c_void_p), len(pr), int(check_balance),
                                   total.ctypes.data_as(C.c_void_p)))
    return rc, total


def verify_airs(descs, proofs, num_queries: int = 100, pow_bits: int = 0, shared_bus_seed: bool = False, check_balance: bool = False):
    """descs: [(width, log_height, cons_bytecode, cons_spans, interactions-or-None)] -> (code, total bus sum).
    code 0 = every proof valid (and balanced if asked); ((i+1) << 8) | c = proof i failed check c; 14 = unbalanced."""
    n = len(descs)
    keep, recs = [], (PwAirDescription * max(n, 1))()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    for i, (w, lh, bc, sp, it) in enumerate(descs):
        bc = np.ascontiguousarray(bc, dtype=np.uint32)
        sp = np.ascontiguousarray(sp, dtype=np.uint32).reshape(-1, 2)
        keep += [bc, sp]
        if it is None:
            recs[i] = PwAirDescription(w, lh, 0, vp(bc), len(bc), vp(sp), len(sp), None, 0, None, 0, None, 0)
        else:
            a = np.ascontiguousarray(it[0], dtype=np.uint32).reshape(-1, 3)
            b = np.ascontiguousarray(it[1], dtype=np.uint32).reshape(-1, 2)
            c = np.ascontiguousarray(it[2], dtype=np.uint32)
            keep += [a, b, c]
            recs[i] = PwAirDescription(w, lh, 1, vp(bc), len(bc), vp(sp), len(sp), vp(a), len(a), vp(b), len(b), vp(c), len(c))
    prs = [np.ascontiguousarray(p, dtype=np.uint32) for p in proofs]
    ptrs = (C.c_void_p * max(n, 1))(*[p.ctypes.data for p in prs])
    lens = (C.c_size_t * max(n, 1))(*[len(p) for p in prs])
    cfg = PwStarkConfig(num_queries, pow_bits)
    total = np.zeros(4, np.uint32)
    rc = int(lib.pw_verify_airs(C.byref(cfg), recs, n, ptrs, lens, int(shared_bus_seed), int(check_balance), vp(total)))
    return rc, total


def poseidon2_host(state) -> np.ndarray:
    s = np.ascontiguousarray(state, dtype=np.uint32).copy()
    lib.pw_poseidon2_permute_host(s.ctypes.data_as(C.c_void_p))
    return s


lib.pw_set_poseidon2_constants.restype = C.c_int
lib.pw_set_poseidon2_constants.argtypes = [C.c_void_p, C.c_void_p]
lib.pw_get_poseidon2_constants.restype = None
lib.pw_get_poseidon2_constants.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]


def jit_compile_check(width: int, cons_bytecode, cons_spans, interactions=None) -> dict:
    """pw_jit_compile_check: generate + compile the specialised kernels of an AIR on the host (no GPU needed)."""
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    bc = np.ascontiguousarray(cons_bytecode, dtype=np.uint32)
    sp = np.ascontiguousarray(cons_spans, dtype=np.uint32).reshape(-1, 2)
    k, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
    err = C.create_string_buffer(4096)
    f = lib.pw_jit_compile_check
    f.restype = C.c_int
    f.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                  C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
    if interactions is not None:
        it, isp, ibc = (np.ascontiguousarray(a, dtype=np.uint32) for a in interactions)
        isp = isp.reshape(-1, 2)
        rc = f(width, vp(bc), len(bc), vp(sp), len(sp), vp(it), len(it.reshape(-1, 3)), vp(isp), len(isp), vp(ibc), len(ibc), C.byref(k), C.byref(b), C.byref(c), err, 4096)
    else:
        rc = f(width, vp(bc), len(bc), vp(sp), len(sp), None, 0, None, 0, None, 0, C.byref(k), C.byref(b), C.byref(c), err, 4096)
    return dict(rc=int(rc), kernels=k.value, code_bytes=b.value, chunks=c.value, error=err.value.decode(errors="replace"))


def jit_generated_sources(width: int, cons_bytecode, cons_spans, interactions=None, which: int = 0, chunk_cost: int = 0, chunks_per_unit: int = 0):
    """pw_jit_generated_source for every translation unit: ([{source, kernel, first_chunk, n_chunks}], total chunks). which: 0 = the
    quotient numerator, 1 = the LogUp permutation columns. No compilation, no GPU (a test hook)."""
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    bc = np.ascontiguousarray(cons_bytecode, dtype=np.uint32)
    sp = np.ascontiguousarray(cons_spans, dtype=np.uint32).reshape(-1, 2)
    f = lib.pw_jit_generated_source
    f.restype = C.c_size_t
    f.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                  C.c_int, C.c_uint32, C.c_uint32, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32),
                  C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    if interactions is not None:
        it, isp, ibc = (np.ascontiguousarray(a, dtype=np.uint32) for a in interactions)
        tables = (vp(it), len(it.reshape(-1, 3)), vp(isp), len(isp.reshape(-1, 2)), vp(ibc), len(ibc))
    else:
        tables = (None, 0, None, 0, None, 0)
    units, total = [], C.c_uint32()
    while True:
        first, n, name = C.c_uint32(), C.c_uint32(), C.create_string_buffer(64)
        size = f(width, vp(bc), len(bc), vp(sp), len(sp), *tables, which, chunk_cost, chunks_per_unit, len(units), None, 0, name, 64, C.byref(first), C.byref(n),
                 C.byref(total))
        if size == 0:
            break
        buf = C.create_string_buffer(size + 1)
        f(width, vp(bc), len(bc), vp(sp), len(sp), *tables, which, chunk_cost, chunks_per_unit, len(units), buf, size + 1, name, 64, C.byref(first), C.byref(n),
          C.byref(total))
        units.append(dict(source=buf.value.decode(), kernel=name.value.decode(), first_chunk=first.value, n_chunks=n.value))
    return units, total.value


def jit_cache_stats() -> dict:
    """pw_jit_cache_stats: translation units this process compiled / loaded from the on-disk code-object cache."""
    lib.pw_jit_cache_stats.restype = None
    lib.pw_jit_cache_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    a, b = C.c_uint64(), C.c_uint64()
    lib.pw_jit_cache_stats(C.byref(a), C.byref(b))
    return dict(compiled=a.value, from_disk=b.value)


_SEGMENT_PROVE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_uint32))
lib.pw_prove_segments_multi.restype = C.c_int
lib.pw_prove_segments_multi.argtypes = [C.POINTER(C.c_int), C.c_size_t, C.POINTER(C.c_uint64), C.c_size_t, _SEGMENT_PROVE_FN, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
lib.pw_multi_last_merge.restype = C.c_int
lib.pw_assign_units.restype = C.c_size_t
lib.pw_assign_units.argtypes = [C.POINTER(C.c_uint64), C.c_size_t, C.c_size_t, C.c_void_p]


def assign_units(cells, n_workers: int) -> np.ndarray:
    """pw_assign_units: worker of every unit (largest first, each to the least loaded worker)."""
    n = len(cells)
    out = np.zeros(n, np.uint32)
    arr = (C.c_uint64 * max(n, 1))(*[int(c) for c in cells])
    assert lib.pw_assign_units(arr, n, n_workers, out.ctypes.data_as(C.c_void_p)) == n
    return out


def prove_segments_multi(devices, segment_cells, prove_segment):
    """pw_prove_segments_multi: one host thread per entry of `devices` proves the segments placed on it by calling
    prove_segment(segment, worker, device) -> 8 commitment words; the commitments are merged over RCCL.
    Returns (uint32 [n_segments, 8], worker of every segment, merge kind: 1 = RCCL all-gather, 2 = host)."""
    n = len(segment_cells)
    errors = []

    def cb(_user, segment, worker, device, out):
        try:
            c = np.asarray(prove_segment(int(segment), int(worker), int(device)), dtype=np.uint32).reshape(8)
            for i in range(8):
                out[i] = int(c[i])
            return 0
        except Exception as e:  # noqa: BLE001 - reported to the caller below
            errors.append(e)
            return 1

    fn = _SEGMENT_PROVE_FN(cb)
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    cells = (C.c_uint64 * max(n, 1))(*[int(c) for c in segment_cells])
    commitments = np.zeros((n, 8), np.uint32)
    owner = np.zeros(n, np.uint32)
    rc = lib.pw_prove_segments_multi(devs, len(devices), cells, n, fn, None, commitments.ctypes.data_as(C.c_void_p), owner.ctypes.data_as(C.c_void_p))
    if errors:
        raise errors[0]
    abi.check(rc, "pw_prove_segments_multi")
    return commitments, owner, int(lib.pw_multi_last_merge())


def set_poseidon2_constants(ext_rc=None, int_rc=None) -> None:
    """pw_set_poseidon2_constants: 8 x 16 external + 13 internal round constants (canonical); None, None = the placeholder."""
    if ext_rc is None:
        rc = lib.pw_set_poseidon2_constants(None, None)
    else:
        e = np.ascontiguousarray(ext_rc, dtype=np.uint32).reshape(8, 16)
        i = np.ascontiguousarray(int_rc, dtype=np.uint32).reshape(13)
        rc = lib.pw_set_poseidon2_constants(e.ctypes.data_as(C.c_void_p), i.ctypes.data_as(C.c_void_p))
    if rc:
        raise ValueError("round constants must be canonical field elements")


def poseidon2_constants():
    e, i, d = np.zeros((8, 16), np.uint32), np.zeros(13, np.uint32), np.zeros(16, np.uint32)
    lib.pw_get_poseidon2_constants(e.ctypes.data_as(C.c_void_p), i.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p))
    return e, i, d


class Prover:
    """One AIR = one prover (constraint programs fixed at construction).

    interactions = (inter[n x 3] = {bus, n_args, first span}, spans[m x 2], bytecode) — the output of
    host.compile_bus(apc, 1) — switches the prover to "pw-stark v0 + LogUp" (proof magic PWS2)."""

    def __init__(self, width: int, cons_bytecode, cons_spans, num_queries: int = 100, pow_bits: int = 0, interactions=None):
        bc = np.ascontiguousarray(cons_bytecode, dtype=np.uint32)
        sp = np.ascontiguousarray(cons_spans, dtype=np.uint32).reshape(-1, 2)
        cfg = PwStarkConfig(num_queries, pow_bits)
        self.width = width
        if interactions is None:
            self._h = lib.pw_prover_create(C.byref(cfg), width, bc.ctypes.data_as(C.c_void_p), len(bc),
                                           sp.ctypes.data_as(C.c_void_p), len(sp))
        else:
            it = np.ascontiguousarray(interactions[0], dtype=np.uint32).reshape(-1, 3)
            isp = np.ascontiguousarray(interactions[1], dtype=np.uint32).reshape(-1, 2)
            ibc = np.ascontiguousarray(interactions[2], dtype=np.uint32)
            self._h = lib.pw_prover_create_logup(C.byref(cfg), width, bc.ctypes.data_as(C.c_void_p), len(bc),
                                                 sp.ctypes.data_as(C.c_void_p), len(sp), it.ctypes.data_as(C.c_void_p), len(it),
                                                 isp.ctypes.data_as(C.c_void_p), len(isp), ibc.ctypes.data_as(C.c_void_p), len(ibc))
        if not self._h:
            raise RuntimeError("pw_prover_create failed")

    def prove(self, d_trace_ptr: int, log_height: int, copy: bool = True, consume: bool = False) -> np.ndarray:
        """pw_prover_prove; consume=True: pw_prover_prove_consuming — the trace is handed over (a streamed proof leaves its
        coefficient arrays there: trace_from_coefficients restores it)."""
        words = C.POINTER(C.c_uint32)()
        n = C.c_size_t()
        fn = lib.pw_prover_prove_consuming if consume else lib.pw_prover_prove
        rc = fn(self._h, d_trace_ptr, log_height, C.byref(words), C.byref(n))
        abi.check(rc, "pw_prover_prove_consuming" if consume else "pw_prover_prove")
        a = np.ctypeslib.as_array(words, shape=(n.value,))
        return a.copy() if copy else a

    def trace_root(self, d_trace_ptr: int, log_height: int) -> np.ndarray:
        """Commitment to the trace alone (8 canonical words) — phase 1 of a multi-AIR LogUp segment."""
        root = np.zeros(8, np.uint32)
        abi.check(lib.pw_prover_trace_root(self._h, d_trace_ptr, log_height, root.ctypes.data_as(C.c_void_p)), "pw_prover_trace_root")
        return root

    def set_bus_seed(self, seed) -> None:
        """Seed of the bus challenges shared by all AIRs of a segment (None: back to the AIR's own trace root)."""
        a = None if seed is None else np.ascontiguousarray(seed, dtype=np.uint32)
        abi.check(lib.pw_prover_set_bus_seed(self._h, None if a is None else a.ctypes.data_as(C.c_void_p)), "pw_prover_set_bus_seed")

    def check_constraints(self, d_trace_ptr: int, log_height: int):
        """Mock prover: (number of violated (row, constraint) pairs, first row, first constraint)."""
        n, row, c = C.c_uint64(), C.c_uint64(), C.c_uint32()
        rc = lib.pw_prover_check_constraints(self._h, d_trace_ptr, log_height, C.byref(n), C.byref(row), C.byref(c))
        abi.check(rc, "pw_prover_check_constraints")
        return (n.value, row.value, c.value) if n.value else (0, None, None)

    def reserve(self, log_height: int) -> None:
        """Allocate the device buffers of a 2^log_height-row proof now (the first prove otherwise pays for it)."""
        lib.pw_prover_reserve.restype = C.c_int
        lib.pw_prover_reserve.argtypes = [C.c_void_p, C.c_uint32]
        abi.check(lib.pw_prover_reserve(self._h, log_height), "pw_prover_reserve")

    def stream_log_blocks(self, log_height: int) -> int:
        """pw_prover_stream_log_blocks: 0 = a proof of this height keeps the LDE resident, b >= 1 = streamed over 2^b sub-cosets of
        the extended domain (the memory free now decides; POWDR_STREAM_LOG_BLOCKS forces), -1 = does not fit."""
        lib.pw_prover_stream_log_blocks.restype = C.c_int
        lib.pw_prover_stream_log_blocks.argtypes = [C.c_void_p, C.c_uint32]
        return int(lib.pw_prover_stream_log_blocks(self._h, log_height))

    def stream_log_blocks_consuming(self, log_height: int) -> int:
        """the same for prove(..., consume=True): the trace's coefficients need no buffer of their own"""
        return int(lib.pw_prover_stream_log_blocks_consuming(self._h, log_height))

    def specialise(self) -> bool:
        """pw_prover_specialise: compile the run-time specialised kernels now. True if the prover has them."""
        lib.pw_prover_specialise.restype = C.c_int
        lib.pw_prover_specialise.argtypes = [C.c_void_p]
        return int(lib.pw_prover_specialise(self._h)) == 0

    def specialised(self) -> dict:
        """pw_prover_specialised: {"state": 1 specialised | 0 not tried | -1 interpreter only, "kernels", "code_bytes", "chunks"}."""
        lib.pw_prover_specialised.restype = C.c_int
        lib.pw_prover_specialised.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        k, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
        st = int(lib.pw_prover_specialised(self._h, C.byref(k), C.byref(b), C.byref(c)))
        return dict(state=st, kernels=k.value, code_bytes=b.value, chunks=c.value)

    def max_constraint_degree(self) -> int:
        lib.pw_prover_max_constraint_degree.restype = C.c_int
        lib.pw_prover_max_constraint_degree.argtypes = [C.c_void_p]
        return int(lib.pw_prover_max_constraint_degree(self._h))

    def device_bytes(self) -> int:
        return int(lib.pw_prover_device_bytes(self._h))

    def logup_path(self) -> int:
        """0 = no LogUp extension, 1 = interpreter, 2 = small forms (pw_prover_logup_path)."""
        return int(lib.pw_prover_logup_path(self._h))

    def close(self):
        if self._h:
            lib.pw_prover_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def trace_from_coefficients(d_ptr: int, width: int, log_height: int) -> None:
    """pw_trace_from_coefficients, in place: what a streamed consuming proof left in the caller's trace buffer -> the trace."""
    import torch

    scratch = torch.empty(1 << 13, dtype=torch.int32, device="cuda")
    abi.check(lib.pw_trace_from_coefficients(d_ptr, width, log_height, scratch.data_ptr()), "pw_trace_from_coefficients")
    torch.cuda.synchronize()  # (the scratch goes out of scope)
