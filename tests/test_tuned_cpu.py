"""oracle/tuned_cpu.cpp — the TUNED CPU baseline bench.py quotes next to the naive `u64 %` oracle: Montgomery
+ AVX-512 LDE and Poseidon2 Merkle commitment. It must compute what the oracle computes."""
import ctypes as C

import numpy as np
import pytest

from oracle import apc_model as om
from oracle import stark_model as sm

P = om.P


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("W,log_h", [(1, 4), (16, 5), (17, 6), (40, 9), (3, 12)])
def test_tuned_lde_and_merkle_equal_the_oracle(W, log_h):
    lib = om.tuned_cpu()
    rng = np.random.default_rng(W + log_h)
    H = 1 << log_h
    t = rng.integers(0, P, W * H, dtype=np.uint32)
    want = sm.lde(t, W, log_h)
    got = np.zeros(W * 2 * H, np.uint32)
    lib.tc_lde(_p(t), C.c_uint32(W), C.c_int(log_h), _p(got))
    assert (got == want).all()
    root = np.zeros(8, np.uint32)
    lib.tc_merkle_root(_p(want), C.c_size_t(2 * H), C.c_uint32(W), _p(root))
    assert (root == sm.merkle_commit(want, 2 * H, W)).all()


def test_tuned_build_matches_the_cpu():
    lib = om.tuned_cpu()
    flags = next((l for l in open("/proc/cpuinfo") if l.startswith("flags")), "")
    assert bool(lib.tc_has_avx512()) == all(f" {f}" in flags for f in ("avx512f", "avx512bw", "avx512dq", "avx512vl"))
