"""Golden proofs of the pw-stark v0 protocol (constraints-only "PWS1" and LogUp "PWS2") for the synthetic T0 AIR,
produced by the CPU oracle. They pin the proof FORMAT across rounds: oracle prover, product host verifier and (on the
GPU) the HIP prover must keep producing / accepting exactly these words. Regenerate only for a deliberate protocol
change:    python tests/golden/make_proof_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import stark_model as sm  # noqa: E402
from powdr_amd import synth  # noqa: E402
from tests.test_oracle_apc import run_oracle_gpu_convention  # noqa: E402

OUT = Path(__file__).resolve().parent


def build():
    s = synth.generate("T0", seed=31)
    apc, idx, trace, _, _ = run_oracle_gpu_convention(s, 50, seed=31)
    W, H = trace.shape
    log_h = H.bit_length() - 1
    bc, spans = sm.compile_constraints(apc, idx)
    inter, ispans, ibc = sm.compile_interactions(apc, idx)
    flat = np.ascontiguousarray(trace).reshape(-1)
    v0 = sm.prove(flat, W, log_h, bc, spans, num_queries=4, pow_bits=5)
    lu = sm.prove_logup(flat, W, log_h, bc, spans, inter, ispans, ibc, num_queries=4, pow_bits=5)
    return dict(trace=flat.astype(np.uint32), width=np.uint32(W), log_h=np.uint32(log_h), cons_bc=bc, cons_spans=spans,
                inter=inter, inter_spans=ispans, inter_bc=ibc, proof_v0=v0, proof_logup=lu)


if __name__ == "__main__":
    d = build()
    np.savez_compressed(OUT / "pw_stark_proofs_T0.npz", **d)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in d.items()})
