"""The oracle's side of the BASELINE-shape GPU tests, computed once per suite run: several test files prove the same trace (the shape's
APC at 2^log_h - 5 calls, seed 0) against the same oracle proof (LogUp, 6 queries, 4 grinding bits). Always imported as
`tests._oracle_cases` so that there is ONE memo whatever name pytest gives the test modules."""
import numpy as np

from oracle import stark_model as sm
from powdr_amd import synth

_MEMO = {}


def synthetic(shape, calls, seed):
    from tests.test_oracle_apc import run_oracle_gpu_convention

    s = synth.generate(shape, seed=seed)
    apc, idx, trace, _, _ = run_oracle_gpu_convention(s, calls, seed=seed)
    W, H = trace.shape
    bc, spans = sm.compile_constraints(apc, idx)
    it = sm.compile_interactions(apc, idx)
    return np.ascontiguousarray(trace).reshape(-1), W, H.bit_length() - 1, bc, spans, it


def baseline_case(shape, log_h):
    """((flat trace, W, log_h, cons_bc, cons_spans, interactions), sm.prove_logup words with num_queries=6, pow_bits=4)"""
    key = (shape, log_h)
    if key not in _MEMO:
        case = synthetic(shape, (1 << log_h) - 5, seed=0)
        flat, W, lh, bc, spans, it = case
        assert lh == log_h
        _MEMO[key] = (case, sm.prove_logup(flat, W, log_h, bc, spans, *it, num_queries=6, pow_bits=4))
    return _MEMO[key]
