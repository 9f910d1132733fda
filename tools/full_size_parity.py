"""One-off: byte parity of the HIP proof with the CPU oracle's at the FULL size of BASELINE configs[1] (C2: 2 022 columns x
2^20 rows). The trace is generated on the device by the product path (its parity with the oracle is tested separately at
sizes the single-threaded reference row loop finishes), copied to the host, proven by the oracle (minutes on the box's
cores, ~45 GB of host memory) and by the HIP prover; the two proofs must be the same words.
usage: python tools/full_size_parity.py [log_h=20] [shape=C2] [queries=16] [pow_bits=8] [logup=0]   -> one JSON line"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

import bench
from oracle import apc_model as om
from oracle import stark_model as sm
from powdr_amd import prover

log_h = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shape = sys.argv[2] if len(sys.argv) > 2 else "C2"
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 16
pow_bits = int(sys.argv[4]) if len(sys.argv) > 4 else 8
logup = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False

free_kb = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1])
wl = bench.build_workload(shape, log_h, True, seed=0, calls_fraction=1.0)
need_gb = wl["W"] * wl["H"] * 4 * (6 if not logup else 14) / 1e9
if free_kb / 1e6 < need_gb + 16:
    print(json.dumps(dict(skipped=f"host has {free_kb / 1e6:.0f} GB available, the oracle needs ~{need_gb:.0f} GB")))
    sys.exit(0)
W, H = wl["W"], wl["H"]
wl["apc"].generate_witness_gpu(wl["instr_air"], wl["dummy"], wl["calls"], wl["out"].data_ptr(), wl["per"])
torch.cuda.synchronize()
bc, spans = wl["cons"]
it = wl["apc"].compile_bus(1) if logup else None
pr = prover.Prover(W, bc, spans, num_queries=nq, pow_bits=pow_bits, interactions=it)
t0 = time.perf_counter()
got = pr.prove(wl["out"].data_ptr(), log_h)
t_gpu = time.perf_counter() - t0
assert pr.check_constraints(wl["out"].data_ptr(), log_h)[0] == 0
flat = om.from_monty(wl["out"].cpu().numpy().view(np.uint32))  # canonical, column-major
del wl["tensors"], wl["dummy"]
torch.cuda.empty_cache()
t0 = time.perf_counter()
want = sm.prove(flat, W, log_h, bc, spans, num_queries=nq, pow_bits=pow_bits) if not logup else \
    sm.prove_logup(flat, W, log_h, bc, spans, *it, num_queries=nq, pow_bits=pow_bits)
t_cpu = time.perf_counter() - t0
same = len(got) == len(want) and bool((got == want).all())
rc = prover.verify(got, W, log_h, bc, spans, nq, pow_bits) if not logup else prover.verify_logup(got, W, log_h, bc, spans, it, nq, pow_bits)[0]
import os
print(json.dumps(dict(shape=shape, cols=W, log_height=log_h, logup=logup, num_queries=nq, pow_bits=pow_bits, proof_words=int(len(got)),
                      proofs_identical=same, first_difference=None if same else int(np.argmax(got[:min(len(got), len(want))] != want[:min(len(got), len(want))])),
                      product_verifier_rc=int(rc), hip_prove_s=t_gpu, oracle_prove_s=t_cpu, host_cores=os.cpu_count(),
                      oracle_cells_per_s=W * H / t_cpu)))
