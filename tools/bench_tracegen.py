"""Scratch timing of the three trace-generation kernels on a synthetic shape.
usage: python tools/bench_tracegen.py [shape] [log_height] [reps]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from oracle import apc_model as om  # table building only (scratch tool, not the product bench)
from powdr_amd import abi, synth, tracegen as tg

shape = sys.argv[1] if len(sys.argv) > 1 else "C2"
logh = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
s = synth.generate(shape, seed=0)
apc = om.load_apc(s.doc)
idx = apc.poly_id_to_index()
gt = om.build_gpu_tables(apc, idx)
H = 1 << logh
calls = H
W = len(idx)
dims = {n: (w, b) for n, w, b in s.airs}
airs = []
tot = 0
for n, b in zip(gt.air_names, gt.row_block_size):
    w, _ = dims[n]
    h = max(synth.next_pow2_or_zero(b * calls), 4)
    t = torch.randint(0, om.P, (w * h,), dtype=torch.int32, device="cuda")
    tot += t.numel() * 4
    airs.append((t, w, h, b))
# bounded kinds
for pid, (name, row, col) in s.source_of.items():
    kind, bound = s.kinds[pid]
    if bound >= om.P:
        continue
    a = gt.air_names.index(name)
    t, w, h, b = airs[a]
    v = torch.randint(0, bound, (calls,), dtype=torch.int64, device="cuda")
    t[col * h + row : col * h + row + b * calls : b] = ((v << 32) % om.P).to(torch.int32)
print(f"{shape}: W={W} H=2^{logh} sources {tot/1e9:.2f} GB, trace {W*H*4/1e9:.2f} GB", flush=True)
out = tg.DeviceMatrix.zeros(H, W)
derived = om.compile_derived(apc, idx, H)
inter, spans, bc = om.compile_bus(apc, idx, H)
print(f"bus bytecode words {len(bc)}, interactions {len(inter)}")
for rep in range(reps):
    per = tg.Periphery.fresh()
    abi.lib.powdr_gpu_timing_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k1 = tg.apc_tracegen(out, airs, gt.subs, calls)
    k2 = tg.apc_apply_derived_expr(out, calls, *derived)
    k3 = tg.apc_apply_bus(out, calls, bc, inter, spans, per)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    rep_t = abi.timing_report()
    cells = W * H
    print(f"rep {rep}: wall {1e3*(t1-t0):.2f} ms  {cells/(t1-t0)/1e9:.2f} Gcells/s ; " +
          " ; ".join(f"{k} {v[1]:.3f} ms" for k, v in rep_t.items()), flush=True)
    g = rep_t.get("apc_gather_tile_kernel", (0, 1))[1] * 1e-3
    print(f"   gather: algorithmic {8*cells/g/1e9:.0f} GB/s, source-stream {tot/g/1e9:.0f} GB/s")
print("hist sums", int(per.var_hist.sum()), int(per.tuple_hist.sum()), int(per.bitwise_hist.sum()))
