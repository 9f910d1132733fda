"""Full-size pin of BASELINE configs[2] WITH its bus argument: the C3p AIR (3 731 columns x 2^22 rows, 3 114 constraints,
2 314 interactions = 4 632 permutation columns), generated trace, proven three ways —
    (a) trace handed over, 2 sub-cosets (pw_prover_prove_consuming: the mode the bench's `c3` leg runs in),
    (b) plain streamed proof over 4 sub-cosets (round 4's mode),  (c) plain streamed proof over 8 sub-cosets —
the three proofs must be the same words (SHA-256), accepted by the product's host verifier AND by the ORACLE's verifier
(oracle/stark_oracle.cpp through oracle.stark_model.verify_logup: a second implementation, canonical u64 arithmetic).
The oracle's PROVER cannot produce this proof for comparison: 15.6 G cells at its 2.2 M cells/s on 256 cores are two hours and
~900 GB of host memory; its byte parity with the HIP prover is pinned at the C3 shape with 2^12 rows (tests/test_streamed_prover.py)
and at the full C2 size (profiles/r03_full_size_parity_c2_logup.json).
usage: python tools/full_size_parity_c3.py [queries=100] [pow_bits=16]   -> one JSON line"""
import hashlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from oracle import stark_model as sm  # noqa: E402  (the CHECKER: this is a parity tool, not a product path)
from powdr_amd import prover  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 100
pow_bits = int(sys.argv[2]) if len(sys.argv) > 2 else 16
log_h = 22
wl = bench.build_workload("C3p", log_h, False, seed=0)
W = wl["W"]
wl["apc"].generate_witness_gpu(wl["instr_air"], wl["dummy"], wl["calls"], wl["out"].data_ptr(), wl["per"])
torch.cuda.synchronize()
wl["dummy"].clear(); wl["tensors"].clear()
torch.cuda.empty_cache()
bc, spans = wl["cons"]
it = wl["apc"].compile_bus(1)
checksum0 = int(wl["out"].view(torch.int64).sum().item())
out, proofs = {}, {}
for name, consume, b in (("handed_over_2_subcosets", True, None), ("streamed_4_subcosets", False, 2), ("streamed_8_subcosets", False, 3)):
    if b is None:
        os.environ.pop("POWDR_STREAM_LOG_BLOCKS", None)
    else:
        os.environ["POWDR_STREAM_LOG_BLOCKS"] = str(b)
    pr = prover.Prover(W, bc, spans, num_queries=nq, pow_bits=pow_bits, interactions=it)
    pr.specialise()
    mode = pr.stream_log_blocks_consuming(log_h) if consume else pr.stream_log_blocks(log_h)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    proof = pr.prove(wl["out"].data_ptr(), log_h, consume=consume)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    if consume and mode > 0:
        prover.trace_from_coefficients(wl["out"].data_ptr(), W, log_h)
    restored = int(wl["out"].view(torch.int64).sum().item()) == checksum0
    out[name] = dict(first_prove_s_including_allocation=t, stream_log_blocks=int(mode), words=int(len(proof)), sha256=hashlib.sha256(proof.tobytes()).hexdigest(),
                     device_bytes=pr.device_bytes(), trace_intact_afterwards=bool(restored),
                     product_verifier_rc=int(prover.verify_logup(proof, W, log_h, bc, spans, it, nq, pow_bits)[0]))
    proofs[name] = proof
    pr.close()
    torch.cuda.empty_cache()
same = len({v["sha256"] for v in out.values()}) == 1
t0 = time.perf_counter()
oracle_rc = int(sm.verify_logup(proofs["handed_over_2_subcosets"], W, log_h, bc, spans, *it, num_queries=nq, pow_bits=pow_bits))
t_or = time.perf_counter() - t0
print(json.dumps(dict(shape="C3p", cols=W, log_height=log_h, logup=True, interactions=int(len(it[0])), perm_cols=4 * len(prover.logup_group_starts(it)),
                      num_queries=nq, pow_bits=pow_bits, proofs_identical=same, oracle_verifier_rc=oracle_rc, oracle_verify_s=t_or, **out)))
