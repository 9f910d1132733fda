"""Scratch timing of the trace-generation stage alone (gather + derived + bus replay) on a synthetic
shape, through the C++ host mirror. usage: python tools/bench_tracegen.py [shape] [log_height] [reps]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

import bench
from powdr_amd import abi

shape = sys.argv[1] if len(sys.argv) > 1 else "C2"
logh = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
wl = bench.build_workload(shape, logh, False, seed=0)
W, H = wl["W"], wl["H"]
print(f"{shape}: W={W} H=2^{logh} sources {wl['src_bytes']/1e9:.2f} GB, trace {W*H*4/1e9:.2f} GB", flush=True)
for rep in range(reps):
    for t in (wl["per"].var_hist, wl["per"].tuple_hist, wl["per"].bitwise_hist):
        t.zero_()
    abi.lib.powdr_gpu_timing_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wl["apc"].generate_witness_gpu(wl["instr_air"], wl["dummy"], wl["calls"], wl["out"].data_ptr(), wl["per"])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    rep_t = abi.timing_report()
    cells = W * H
    print(f"rep {rep}: wall {1e3*(t1-t0):.2f} ms  {cells/(t1-t0)/1e9:.2f} Gcells/s ; " +
          " ; ".join(f"{k} {v[1]:.3f} ms" for k, v in rep_t.items()), flush=True)
    g = rep_t.get("apc_gather_tile_kernel", (0, 1))[1] * 1e-3
    print(f"   gather: algorithmic {8*cells/g/1e9:.0f} GB/s")
print("hist sums", int(wl["per"].var_hist.sum()), int(wl["per"].tuple_hist.sum()), int(wl["per"].bitwise_hist.sum()))
