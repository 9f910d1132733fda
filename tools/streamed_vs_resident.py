"""Full-size check of the streamed proof path without the oracle's eight minutes: BASELINE configs[1] (C2: 2 022
columns x 2^20 rows, generated trace, LogUp) proven resident and streamed over 4 and 8 sub-cosets — the three proofs must be the
same words, accepted by the product's host verifier. (Resident == oracle at this size: profiles/r03_full_size_parity_c2_logup.json;
streamed == oracle at the sizes the oracle proves in seconds: tests/test_streamed_prover.py.)
usage: python tools/streamed_vs_resident.py [log_h=20] [shape=C2] [queries=16] [pow_bits=8]   -> one JSON line"""
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
from powdr_amd import prover  # noqa: E402

log_h = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shape = sys.argv[2] if len(sys.argv) > 2 else "C2"
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 16
pow_bits = int(sys.argv[4]) if len(sys.argv) > 4 else 8

wl = bench.build_workload(shape, log_h, True, seed=0, calls_fraction=1.0)
W = wl["W"]
wl["apc"].generate_witness_gpu(wl["instr_air"], wl["dummy"], wl["calls"], wl["out"].data_ptr(), wl["per"])
torch.cuda.synchronize()
wl["dummy"].clear(); wl["tensors"].clear()
torch.cuda.empty_cache()
bc, spans = wl["cons"]
it = wl["apc"].compile_bus(1)
out = {}
for b in (0, 2, 3):
    os.environ["POWDR_STREAM_LOG_BLOCKS"] = str(b)
    pr = prover.Prover(W, bc, spans, num_queries=nq, pow_bits=pow_bits, interactions=it)
    pr.specialise()
    pr.prove(wl["out"].data_ptr(), log_h, copy=False)  # buffers
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    proof = pr.prove(wl["out"].data_ptr(), log_h)
    t = time.perf_counter() - t0
    out[b] = dict(prove_s=t, words=int(len(proof)), sha256=hashlib.sha256(proof.tobytes()).hexdigest(), device_bytes=pr.device_bytes(),
                  verify_rc=int(prover.verify_logup(proof, W, log_h, bc, spans, it, nq, pow_bits)[0]))
    assert pr.check_constraints(wl["out"].data_ptr(), log_h)[0] == 0
    pr.close()
    torch.cuda.empty_cache()
same = len({v["sha256"] for v in out.values()}) == 1
print(json.dumps(dict(shape=shape, cols=W, log_height=log_h, logup=True, num_queries=nq, pow_bits=pow_bits, proofs_identical=same,
                      resident=out[0], streamed_4_subcosets=out[2], streamed_8_subcosets=out[3])))
