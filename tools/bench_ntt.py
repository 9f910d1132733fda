"""LDE only (iNTT + coset NTT, pw_lde_batch): time per call and VALU-relevant rates.
usage: python tools/bench_ntt.py [W] [log_h] [reps]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from powdr_amd import abi, prover

W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
log_h = int(sys.argv[2]) if len(sys.argv) > 2 else 20
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
H = 1 << log_h
t = torch.randint(0, 0x78000001, (W * H,), dtype=torch.int32, device="cuda")
c = torch.empty(W * H, dtype=torch.int32, device="cuda")
l = torch.empty(2 * W * H, dtype=torch.int32, device="cuda")
for _ in range(2):
    abi.check(prover.lib.pw_lde_batch(t.data_ptr(), W, log_h, c.data_ptr(), l.data_ptr()), "pw_lde_batch")
torch.cuda.synchronize()
abi.lib.powdr_gpu_timing_enable(1)
t0 = time.perf_counter()
for _ in range(reps):
    abi.check(prover.lib.pw_lde_batch(t.data_ptr(), W, log_h, c.data_ptr(), l.data_ptr()), "pw_lde_batch")
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
tm = abi.timing_report()
abi.lib.powdr_gpu_timing_enable(0)
# the fused schedule (pw_lde_fused): what the provers run
for _ in range(2):
    abi.check(prover.lib.pw_lde_fused(t.data_ptr(), W, log_h, c.data_ptr(), l.data_ptr()), "pw_lde_fused")
torch.cuda.synchronize()
abi.lib.powdr_gpu_timing_enable(1)
t0 = time.perf_counter()
for _ in range(reps):
    abi.check(prover.lib.pw_lde_fused(t.data_ptr(), W, log_h, c.data_ptr(), l.data_ptr()), "pw_lde_fused")
torch.cuda.synchronize()
dtf = (time.perf_counter() - t0) / reps
tmf = abi.timing_report()
abi.lib.powdr_gpu_timing_enable(0)
print(f"W={W} log_h={log_h}: FUSED {dtf*1e3:.3f} ms per LDE ({dt/dtf:.2f}x), {W*H/dtf/1e9:.2f} Gcells/s; per kernel: "
      + ", ".join(f"{k} {ms/reps:.3f} ms" for k, (n, ms) in tmf.items()))
elst = W * H * (log_h + 2 * log_h)  # element-stages: iNTT log_h on H, forward log_h (after the duplication) on 2H
print(f"W={W} log_h={log_h}: {dt*1e3:.3f} ms per LDE, {elst/dt/1e12:.3f} T element-stages/s, "
      f"{W*H/dt/1e9:.2f} Gcells/s; per kernel: " + ", ".join(f"{k} {ms/reps:.3f} ms" for k, (n, ms) in tm.items()))
