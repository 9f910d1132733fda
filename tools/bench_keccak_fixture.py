"""The reference's own pre-optimisation keccak APC (autoprecompiles/tests/keccak_apc_pre_opt.json.gz, through
tests/golden/keccak_apc_pre_opt.apc.npz): 27 521 columns gathered 1:1 from 5 original AIRs, 13 262 real bus interactions,
28 627 real constraint programs. Trace generation + proof on random dummy traces, per-kernel times.
usage: python tools/bench_keccak_fixture.py [log_height] [reps]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

from powdr_amd import abi, prover, synth, tracegen as tg

log_h = int(sys.argv[1]) if len(sys.argv) > 1 else 14
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
z = np.load(ROOT / "tests" / "golden" / "keccak_apc_pre_opt.apc.npz")
W, H, calls = len(z["poly_ids"]), 1 << log_h, 1 << log_h
airs = []
for w, b in zip(z["air_widths"], z["row_block_size"]):
    h = synth.next_pow2_or_zero(int(b) * calls)
    airs.append((torch.randint(0, 256, (int(w) * h,), dtype=torch.int32, device="cuda"), int(w), h, int(b)))
print(f"keccak pre-opt APC: W={W} H=2^{log_h} ({W * H / 1e9:.2f} G cells), sources {sum(a[0].numel() for a in airs) * 4 / 1e9:.1f} GB, "
      f"{len(z['bus_inter'])} interactions, {len(z['cons_spans'])} constraints", flush=True)
bus_bc = z["bus_bc"].copy()
bus_bc[z["bus_apc_pos"]] *= H  # the reference's device encoding: PUSH_APC operand = col * H
out = tg.DeviceMatrix.zeros(H, W)
per = tg.Periphery.fresh()
pr = prover.Prover(W, z["cons_bc"], z["cons_spans"], num_queries=100, pow_bits=16)
print("max constraint degree", pr.max_constraint_degree())
for rep in range(reps + 1):
    for t in (per.var_hist, per.tuple_hist, per.bitwise_hist):
        t.zero_()
    abi.lib.powdr_gpu_timing_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    keep = [tg.apc_tracegen(out, airs, z["subs"], calls), tg.apc_apply_bus(out, calls, bus_bc, z["bus_inter"], z["bus_spans"], per)]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    proof = pr.prove(out.ptr(), log_h, copy=False)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    tm = abi.timing_report()
    if rep:
        print(f"rep {rep}: trace generation {(t1 - t0) * 1e3:.1f} ms (host tables re-uploaded each call), proof {(t2 - t1) * 1e3:.1f} ms, "
              f"{W * H / (t2 - t0) / 1e9:.2f} G cells/s; " + ", ".join(f"{k} {v[1]:.2f}" for k, v in tm.items() if v[1] > 0.3), flush=True)
print("proof words", len(proof), "prover buffers %.1f GB" % (pr.device_bytes() / 1e9))
