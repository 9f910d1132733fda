"""Experiment: prove independent segments from P host threads / HIP streams on ONE GPU so that the
HBM-bound stages of one segment overlap the VALU-bound Poseidon2 hashing of another.
usage: python tools/pipeline_experiment.py [P] [steps] [log_h]"""
import sys
import threading
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

import bench
from powdr_amd import abi, host, prover, tracegen as tg

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
log_h = int(sys.argv[3]) if len(sys.argv) > 3 else 20
wl = bench.build_workload("C2", log_h, False, seed=0)
workers = []
for i in range(P):
    w = dict(stream=torch.cuda.Stream(), apc=host.Apc(wl["synth"].doc), per=tg.Periphery.fresh(),
             out=torch.empty_like(wl["out"]), pr=prover.Prover(wl["W"], *wl["cons"], num_queries=100, pow_bits=16))
    workers.append(w)


def run(w, n, roots):
    with torch.cuda.stream(w["stream"]):
        abi.lib.powdr_gpu_set_stream(w["stream"].cuda_stream)
        for _ in range(n):
            for t in (w["per"].var_hist, w["per"].tuple_hist, w["per"].bitwise_hist):
                t.zero_()
            w["apc"].generate_witness_gpu(wl["instr_air"], wl["dummy"], wl["calls"], w["out"].data_ptr(), w["per"])
            proof = w["pr"].prove(w["out"].data_ptr(), log_h, copy=False)
            roots.append(proof[6:14].copy())


for label, nthreads in (("warmup", P), ("sequential", 1), (f"{P} threads", P)):
    roots = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if nthreads == 1:
        run(workers[0], steps, roots)
    else:
        th = [threading.Thread(target=run, args=(workers[i], steps // P, roots)) for i in range(P)]
        [t.start() for t in th]
        [t.join() for t in th]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = len(roots)
    print(f"{label}: {n} segments in {dt*1e3:.1f} ms -> {dt/n*1e3:.1f} ms/segment, {wl['W']*wl['H']*n/dt/1e9:.2f} Gcells/s; "
          f"all roots equal: {all((r == roots[0]).all() for r in roots)}", flush=True)
