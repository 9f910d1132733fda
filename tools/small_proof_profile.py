"""50 sequential proofs of one small AIR (default 446 x 2^10): wall time per proof, for use under
rocprofv3 --kernel-trace --stats (kernel time vs. host overhead of a latency-bound proof).
usage: python tools/small_proof_profile.py [W] [log_h] [n]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from powdr_amd import prover

W = int(sys.argv[1]) if len(sys.argv) > 1 else 446
log_h = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = int(sys.argv[3]) if len(sys.argv) > 3 else 50
bc = np.array([0, 0, 0, 1, 4, 0, 0, 0, 1, 4, 3], np.uint32)
pr = prover.Prover(W, bc, np.array([[0, len(bc)]], np.uint32), num_queries=100, pow_bits=16)
t = torch.randint(0, 0x78000001, (W << log_h,), dtype=torch.int32, device="cuda")
for _ in range(3):
    pr.prove(t.data_ptr(), log_h, copy=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    pr.prove(t.data_ptr(), log_h, copy=False)
torch.cuda.synchronize()
print(f"W={W} log_h={log_h}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per proof")
