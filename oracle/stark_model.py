"""TEST INFRASTRUCTURE — Python bindings of oracle/stark_oracle.cpp (pw-stark v0 CPU oracle).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import apc_model as om

P = om.P


def _lib():
    lib = om.c_oracle()
    lib.or_prove.restype = C.c_size_t
    lib.or_prove_logup.restype = C.c_size_t
    lib.or_verify_logup.restype = C.c_int
    lib.or_verify.restype = C.c_int
    lib.or_root_of_unity.restype = C.c_uint32
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def poseidon2(state) -> np.ndarray:
    s = np.ascontiguousarray(state, dtype=np.uint32).copy()
    assert s.shape == (16,)
    _lib().or_poseidon2_permute(_p(s))
    return s


def poseidon2_constants():
    e, i, d = np.zeros((8, 16), np.uint32), np.zeros(13, np.uint32), np.zeros(16, np.uint32)
    _lib().or_poseidon2_constants(_p(e), _p(i), _p(d))
    return e, i, d


def set_poseidon2_constants(ext_rc=None, int_rc=None) -> None:
    """Install another round-constant table in the oracle (canonical words, 8 x 16 and 13); None, None = the placeholder."""
    if ext_rc is None:
        rc = _lib().or_set_poseidon2_constants(None, None)
    else:
        e = np.ascontiguousarray(ext_rc, dtype=np.uint32).reshape(8, 16)
        i = np.ascontiguousarray(int_rc, dtype=np.uint32).reshape(13)
        rc = _lib().or_set_poseidon2_constants(_p(e), _p(i))
    if rc:
        raise ValueError("round constants must be canonical field elements")


def root_of_unity(log_n: int) -> int:
    return int(_lib().or_root_of_unity(C.c_int(log_n)))


def ext_mul(a, b):
    a, b = (np.ascontiguousarray(x, dtype=np.uint32) for x in (a, b))
    o = np.zeros(4, np.uint32)
    _lib().or_ext_mul(_p(a), _p(b), _p(o))
    return o


def ext_inv(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    o = np.zeros(4, np.uint32)
    _lib().or_ext_inv(_p(a), _p(o))
    return o


def dft(a, inverse=False):
    a = np.ascontiguousarray(a, dtype=np.uint32).copy()
    log_n = int(len(a)).bit_length() - 1
    _lib().or_dft(_p(a), C.c_int(log_n), C.c_int(1 if inverse else 0))
    return a


def dft_naive(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    out = np.zeros_like(a)
    _lib().or_dft_naive(_p(a), _p(out), C.c_int(int(len(a)).bit_length() - 1))
    return out


def lde(trace_cm: np.ndarray, width: int, log_h: int) -> np.ndarray:
    """trace_cm: flat column-major W x H canonical. Returns flat column-major W x 2H (natural order)."""
    t = np.ascontiguousarray(trace_cm, dtype=np.uint32)
    out = np.zeros(width * (2 << log_h), np.uint32)
    _lib().or_lde(_p(t), C.c_uint32(width), C.c_int(log_h), _p(out))
    return out


def merkle_commit(m_cm: np.ndarray, height: int, width: int, want_digests=False):
    m = np.ascontiguousarray(m_cm, dtype=np.uint32)
    root = np.zeros(8, np.uint32)
    dig = np.zeros((2 * height - 1) * 8, np.uint32) if want_digests else None
    _lib().or_merkle_commit(_p(m), C.c_size_t(height), C.c_size_t(width), _p(dig) if want_digests else None, _p(root))
    return (root, dig) if want_digests else root


def compile_constraints(apc: om.Apc, idx: dict):
    """Constraint programs with column-index operands -> (bytecode u32[], spans u32[n,2])."""
    bc, spans = [], []
    for c in apc.constraints:
        off = len(bc)
        om.emit_expr(bc, c, idx, 1)
        spans.append((off, len(bc) - off))
    return np.array(bc, np.uint32), np.array(spans, np.uint32).reshape(-1, 2)


def _proof_cap(num_queries, shapes) -> int:
    """An upper bound of a proof's words for `shapes` = [(width, permutation columns, log_h)]: the output buffer is sized ONCE — a buffer
    that turns out too small makes or_prove* return the size it needs and the wrapper prove again, which doubled the CPU time of every
    proof above 2^18 words (the bench's `cpu_baseline` with 100 queries, round 6) without changing a word of it."""
    log_n = max(lh for _, _, lh in shapes) + 1
    opened = sum(w + 2 * wp + 8 for w, wp, _ in shapes)
    rows = sum(w + wp + 8 for w, wp, _ in shapes)
    per_query = 1 + rows + 3 * 8 * log_n + log_n * (4 + 8 * log_n)
    return 4096 + 16 * len(shapes) + 4 * opened + 12 * log_n + num_queries * per_query


def prove(trace_cm, width, log_h, cons_bc, cons_spans, num_queries=8, pow_bits=0) -> np.ndarray:
    lib = _lib()
    t = np.ascontiguousarray(trace_cm, dtype=np.uint32)
    bc = np.ascontiguousarray(cons_bc, dtype=np.uint32)
    sp = np.ascontiguousarray(cons_spans, dtype=np.uint32)
    args = (C.c_uint32(num_queries), C.c_uint32(pow_bits), _p(t), C.c_uint32(width), C.c_uint32(log_h), _p(bc), _p(sp), C.c_size_t(len(sp)))
    cap = max(1 << 16, _proof_cap(num_queries, [(width, 0, log_h)]))
    while True:
        buf = np.zeros(cap, np.uint32)
        n = lib.or_prove(*args, _p(buf), C.c_size_t(cap))
        if n <= cap:
            return buf[:n].copy()
        cap = int(n)


def verify(proof, width, log_h, cons_bc, cons_spans, num_queries=8, pow_bits=0) -> int:
    pr = np.ascontiguousarray(proof, dtype=np.uint32)
    bc = np.ascontiguousarray(cons_bc, dtype=np.uint32)
    sp = np.ascontiguousarray(cons_spans, dtype=np.uint32)
    return int(_lib().or_verify(C.c_uint32(num_queries), C.c_uint32(pow_bits), _p(pr), C.c_size_t(len(pr)), C.c_uint32(width),
                                C.c_uint32(log_h), _p(bc), _p(sp), C.c_size_t(len(sp))))


def compile_interactions(apc: om.Apc, idx: dict):
    """All bus interactions with column-index operands: (inter[n,3], spans[m,2], bytecode) — compile_bus at height 1."""
    return om.compile_bus(apc, idx, 1)


def group_starts(inter, ispans, ibc) -> np.ndarray:
    """Boundaries of the LogUp groups (n_groups + 1 interaction indices)."""
    it, isp, ib = (np.ascontiguousarray(a, dtype=np.uint32) for a in (inter, ispans, ibc))
    n = len(it.reshape(-1, 3))
    out = np.zeros(n + 2, np.uint32)
    lib = _lib()
    lib.or_group_starts.restype = C.c_size_t
    k = lib.or_group_starts(_p(it), C.c_size_t(n), _p(isp), _p(ib), _p(out), C.c_size_t(len(out)))
    return out[:k].copy()


def _seed(bus_seed):
    if bus_seed is None:
        return None, C.c_void_p(None)
    a = np.ascontiguousarray(bus_seed, dtype=np.uint32)
    assert a.shape == (8,)
    return a, _p(a)


def prove_logup(trace_cm, width, log_h, cons_bc, cons_spans, inter, ispans, ibc, num_queries=8, pow_bits=0, bus_seed=None) -> np.ndarray:
    """bus_seed: 8 canonical words shared by all AIRs of a segment (None: the AIR's own trace root)."""
    lib = _lib()
    keep, seed_p = _seed(bus_seed)
    arrs = [np.ascontiguousarray(a, dtype=np.uint32) for a in (trace_cm, cons_bc, cons_spans, inter, ispans, ibc)]
    t, bc, sp, it, isp, ib = arrs
    args = (C.c_uint32(num_queries), C.c_uint32(pow_bits), _p(t), C.c_uint32(width), C.c_uint32(log_h), _p(bc), _p(sp),
            C.c_size_t(len(sp.reshape(-1, 2))), _p(it), C.c_size_t(len(it.reshape(-1, 3))), _p(isp), _p(ib), seed_p)
    cap = max(1 << 18, _proof_cap(num_queries, [(width, 4 * (len(it.reshape(-1, 3)) + 1), log_h)]))
    while True:
        buf = np.zeros(cap, np.uint32)
        n = lib.or_prove_logup(*args, _p(buf), C.c_size_t(cap))
        if n <= cap:
            return buf[:n].copy()
        cap = int(n)


def verify_logup(proof, width, log_h, cons_bc, cons_spans, inter, ispans, ibc, num_queries=8, pow_bits=0, bus_seed=None) -> int:
    keep, seed_p = _seed(bus_seed)
    arrs = [np.ascontiguousarray(a, dtype=np.uint32) for a in (proof, cons_bc, cons_spans, inter, ispans, ibc)]
    pr, bc, sp, it, isp, ib = arrs
    return int(_lib().or_verify_logup(C.c_uint32(num_queries), C.c_uint32(pow_bits), _p(pr), C.c_size_t(len(pr)), C.c_uint32(width),
                                      C.c_uint32(log_h), _p(bc), _p(sp), C.c_size_t(len(sp.reshape(-1, 2))), _p(it),
                                      C.c_size_t(len(it.reshape(-1, 3))), _p(isp), _p(ib), seed_p))


def commitment_digest(roots) -> np.ndarray:
    """Restatement of pw_commitment_digest: binary Poseidon2 tree over the ordered 8-word commitments, an odd node is
    paired with zeros, a single commitment is its own digest, none gives zeros."""
    level = [np.asarray(r, dtype=np.uint32) for r in np.asarray(roots, dtype=np.uint32).reshape(-1, 8)]
    if not level:
        return np.zeros(8, np.uint32)
    while len(level) > 1:
        nxt = []
        for i in range(0, len(level), 2):
            right = level[i + 1] if i + 1 < len(level) else np.zeros(8, np.uint32)
            nxt.append(poseidon2(np.concatenate([level[i], right]))[:8])
        level = nxt
    return level[0]


# ---- pw-stark v1: one proof per segment (oracle/stark_segment.inc) ----------------------------------------------------
class OrSegAir(C.Structure):
    _fields_ = [("trace", C.c_void_p), ("width", C.c_uint32), ("log_h", C.c_uint32), ("cons_bc", C.c_void_p), ("cons_spans", C.c_void_p),
                ("n_constraints", C.c_size_t), ("inter", C.c_void_p), ("n_inter", C.c_size_t), ("ispans", C.c_void_p), ("ibc", C.c_void_p)]


def _seg_airs(airs, with_traces=True):
    """airs: [(trace_cm or None, width, log_h, cons_bc, cons_spans, interactions-or-None)] -> (ctypes array, keep-alive list)."""
    keep, recs = [], (OrSegAir * max(len(airs), 1))()
    for i, (t, w, lh, bc, sp, it) in enumerate(airs):
        bc = np.ascontiguousarray(bc, dtype=np.uint32)
        sp = np.ascontiguousarray(sp, dtype=np.uint32).reshape(-1, 2)
        tt = np.ascontiguousarray(t, dtype=np.uint32) if (with_traces and t is not None) else None
        if it is None:
            a, b, c = np.zeros((0, 3), np.uint32), np.zeros((0, 2), np.uint32), np.zeros(0, np.uint32)
        else:
            a = np.ascontiguousarray(it[0], dtype=np.uint32).reshape(-1, 3)
            b = np.ascontiguousarray(it[1], dtype=np.uint32).reshape(-1, 2)
            c = np.ascontiguousarray(it[2], dtype=np.uint32)
        keep += [bc, sp, tt, a, b, c]
        recs[i] = OrSegAir(None if tt is None else tt.ctypes.data, w, lh, bc.ctypes.data, sp.ctypes.data, len(sp), a.ctypes.data, len(a),
                           b.ctypes.data, c.ctypes.data)
    return recs, keep


def prove_segment(airs, num_queries=8, pow_bits=0, logup=False) -> np.ndarray:
    """ONE proof for all AIRs of a segment. airs: [(trace_cm, width, log_h, cons_bc, cons_spans, interactions-or-None)]."""
    lib = _lib()
    lib.or_prove_segment.restype = C.c_size_t
    recs, keep = _seg_airs(airs)
    cap = max(1 << 18, _proof_cap(num_queries, [(w, 4 * ((0 if it is None else len(np.asarray(it[0]).reshape(-1, 3))) + 1) if logup else 0, lh)
                                                  for (_, w, lh, _, _, it) in airs]))
    while True:
        buf = np.zeros(cap, np.uint32)
        n = lib.or_prove_segment(C.c_uint32(num_queries), C.c_uint32(pow_bits), C.c_int(int(logup)), recs, C.c_size_t(len(airs)), _p(buf), C.c_size_t(cap))
        if n <= cap:
            return buf[:n].copy()
        cap = int(n)


def verify_segment(proof, airs, num_queries=8, pow_bits=0, logup=False, check_balance=False):
    """(code, sum of the AIRs' cumulative bus sums). airs as for prove_segment (traces ignored)."""
    lib = _lib()
    lib.or_verify_segment.restype = C.c_int
    recs, keep = _seg_airs(airs, with_traces=False)
    pr = np.ascontiguousarray(proof, dtype=np.uint32)
    total = np.zeros(4, np.uint32)
    rc = lib.or_verify_segment(C.c_uint32(num_queries), C.c_uint32(pow_bits), C.c_int(int(logup)), recs, C.c_size_t(len(airs)), _p(pr),
                               C.c_size_t(len(pr)), C.c_int(int(check_balance)), _p(total))
    return int(rc), total
