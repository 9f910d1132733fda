"""Where a SHORT segment's time goes (round 6: the execution's tail, HonestSegment.draw_shape(n - 1, n): every chip at <= 1/8 of its cap):
per-kernel sums of the library's event timers over `reps` proofs of the tail segment, next to the wall time of the proof.
usage: python tools/tail_segment_profile.py [C4|C5] [reps=3]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from powdr_amd import abi, segment_workload as sw

kind = sys.argv[1] if len(sys.argv) > 1 else "C5"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
seg = sw.HonestSegment(kind, max_log_height=20, seed=0, queries=100, pow_bits=16, logup=True)
for which, u in (("capped", 0), ("tail", 7)):
    shape = seg.draw_shape(u, 8)
    seg.stage_inputs(u, shape)
    seg.generate_traces()
    seg.prove()
    torch.cuda.synchronize()
    abi.lib.powdr_gpu_timing_enable(1)
    t0 = time.perf_counter()
    for _ in range(reps):
        seg.generate_traces()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        seg.prove()
        torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    rep = abi.timing_report()
    abi.lib.powdr_gpu_timing_enable(0)
    ksum = sum(ms for _, ms in rep.values()) / reps
    n_launch = sum(c for c, _ in rep.values()) / reps
    print(f"{kind} {which}: {seg.cells / 1e9:.3f} G cells, {len(seg.airs)} AIRs, heights 2^{min(seg.heights())}..2^{max(seg.heights())}: wall {wall * 1e3:.1f} ms per "
          f"generate + prove, kernel-time sum {ksum:.1f} ms over {n_launch:.0f} timed launches (side streams overlap: the sum may exceed the wall)")
    for k, (c, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"    {k:40s} {ms / reps:8.2f} ms  x{c / reps:.0f}")
seg.close()
