"""the query pass of a streamed proof in SELECT mode (ntt.hip subcoset_query_rows) in isolation — one constraint-free AIR
of --cols random columns x 2^--log-h rows, proven STREAMED (2 sub-cosets, 100 queries), timed. POWDR_QUERY_SELECT chooses the form:
2 = round 5's 64-bit atomic sums (the kernel whose `rocprofv3 --pmc FETCH_SIZE` pass did not return at configs[2]), 1 = the tiles' terms
stored + select_reduce_kernel (round 6 default), 0 = the stored partial transform. tools/repro_select_atomics.sh runs the three under
--kernel-trace and under --pmc, each behind its own `timeout`."""
import argparse
import hashlib
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
ap = argparse.ArgumentParser()
ap.add_argument("--cols", type=int, default=512)
ap.add_argument("--log-h", type=int, default=20)
ap.add_argument("--proofs", type=int, default=2)
args = ap.parse_args()
os.environ["POWDR_STREAM_LOG_BLOCKS"] = "1"
import numpy as np
import torch

from powdr_amd import abi, prover

torch.manual_seed(6)
torch.cuda.manual_seed_all(6)  # the same trace for every form: the digests below must agree
t = torch.empty(args.cols << args.log_h, dtype=torch.int32, device="cuda")
t.random_(0, 0x78000001)
pr = prover.Prover(args.cols, np.zeros(0, np.uint32), np.zeros((0, 2), np.uint32), num_queries=100, pow_bits=0)
for k in range(args.proofs):
    torch.cuda.synchronize()
    abi.lib.powdr_gpu_timing_enable(1)
    t0 = time.perf_counter()
    pf = pr.prove(t.data_ptr(), args.log_h)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rep = abi.timing_report()
    abi.lib.powdr_gpu_timing_enable(0)
print(f"select={os.environ.get('POWDR_QUERY_SELECT', '1')} cols={args.cols} log_h={args.log_h} proof {dt * 1e3:.1f} ms  "
      f"sha256 {hashlib.sha256(pf.astype('<u4').tobytes()).hexdigest()[:16]}  "
      + "  ".join(f"{k}: {c} x {ms / max(c, 1):.3f} ms" for k, (c, ms) in rep.items() if "dit" in k or "rows" in k))
