"""Proof latency / throughput of pw_prover_prove as a function of the trace height (reth-shaped segments hold
30-100 APC AIRs with heights 2^10..2^20): one prover per height, random traces, no constraints
beyond a product check, sequential proofs, then T host threads with their own streams.
usage: python tools/bench_air_heights.py [W] [threads]"""
import sys
import threading
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from powdr_amd import abi, prover

W = int(sys.argv[1]) if len(sys.argv) > 1 else 446
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
P = 0x78000001
PA, MUL, SUB = 0, 4, 3
bc = np.array([PA, 0, PA, 1, MUL, PA, 0, PA, 1, MUL, SUB], np.uint32)  # x0*x1 - x0*x1 == 0
spans = np.array([[0, len(bc)]], np.uint32)
print(f"W={W}; columns: log_h, ms/proof sequential, Mcells/s, ms/proof with {T} threads, Mcells/s")
for log_h in (10, 12, 14, 16, 18, 20):
    H = 1 << log_h
    reps = max(3, min(50, (1 << 22) // H))
    workers = []
    for i in range(T):
        t = torch.randint(0, P, (W * H,), dtype=torch.int32, device="cuda")
        workers.append((torch.cuda.Stream(), t, prover.Prover(W, bc, spans, num_queries=100, pow_bits=16)))

    def run(w, n):
        st, t, pr = w
        with torch.cuda.stream(st):
            abi.lib.powdr_gpu_set_stream(st.cuda_stream)
            for _ in range(n):
                pr.prove(t.data_ptr(), log_h, copy=False)

    for w in workers:
        run(w, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(workers[0], reps)
    torch.cuda.synchronize()
    seq = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(w, reps)) for w in workers]
    [x.start() for x in th]
    [x.join() for x in th]
    torch.cuda.synchronize()
    par = (time.perf_counter() - t0) / (reps * T)
    print(f"{log_h:3d} {seq*1e3:9.3f} {W*H/seq/1e6:10.1f} {par*1e3:9.3f} {W*H/par/1e6:10.1f}", flush=True)
    for _, _, pr in workers:
        pr.close()
    del workers
    torch.cuda.empty_cache()
