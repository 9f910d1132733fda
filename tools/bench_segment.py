"""A reth-shaped segment (tens of APC AIRs, log-uniform heights 2^10..2^20, widths 30..4000): cells/s of
pw_prove_segment as a function of the worker count. Random traces, one product constraint per AIR, constraints only.
usage: python tools/bench_segment.py [n_airs] [max_total_Gcells] [seed]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from powdr_amd import prover

n_airs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
budget = float(sys.argv[2]) * 1e9 if len(sys.argv) > 2 else 3e9
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = np.random.default_rng(seed)
P = 0x78000001
PA, MUL, SUB = 0, 4, 3
airs, total = [], 0
while len(airs) < n_airs:
    log_h = int(rng.integers(10, 21))
    W = int(np.exp(rng.uniform(np.log(30), np.log(4000))))
    if total + (W << log_h) > budget:
        if all((30 << lh) + total > budget for lh in range(10, 21)):
            break
        continue
    total += W << log_h
    bc = np.array([PA, 0, PA, 1, MUL, PA, 0, PA, 1, MUL, SUB], np.uint32)
    t = torch.randint(0, P, (W << log_h,), dtype=torch.int32, device="cuda")
    airs.append((prover.Prover(W, bc, np.array([[0, len(bc)]], np.uint32), num_queries=100, pow_bits=16), t, log_h, W))
hist = np.bincount([a[2] for a in airs], minlength=21)[10:]
print(f"{len(airs)} AIRs, {total/1e9:.2f} G cells, heights 2^10..2^20 counts {hist.tolist()}, widths {min(a[3] for a in airs)}..{max(a[3] for a in airs)}")
seg = [(pr, t.data_ptr(), lh) for pr, t, lh, _ in airs]
prover.prove_airs(seg, n_workers=4, copy=False)  # warm-up: buffers, NTT tables
for workers in (1, 2, 4, 8, 16):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        prover.prove_airs(seg, n_workers=workers, copy=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"workers={workers:2d}: {dt*1e3:8.1f} ms per segment, {total/dt/1e9:6.2f} G cells/s", flush=True)
# the one-proof-per-segment form (pw-stark v1): all AIRs of a phase in one mixed-height commitment, one FRI, one query phase
pf = prover.prove_segment(seg, logup=False, copy=False)
for _ in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        pf = prover.prove_segment(seg, logup=False, copy=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"pw_prove_segment (one proof): {dt*1e3:8.1f} ms per segment, {total/dt/1e9:6.2f} G cells/s, proof {len(pf)*4/1e6:.2f} MB", flush=True)
per_air = prover.prove_airs(seg, n_workers=8, copy=False)[0]
print(f"independent proofs: {sum(len(p) for p in per_air)*4/1e6:.2f} MB in total")
print(f"prover buffers: {sum(pr.device_bytes() for pr, *_ in airs)/1e9:.1f} GB")
