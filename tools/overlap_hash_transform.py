"""VERDICT r4 #2a, measured: the leaf hash of sub-coset k on one stream while sub-coset k+1 is transformed on another.
Two host threads, each with its own HIP stream (powdr_gpu_set_stream is per thread): thread A hashes the rows of a resident
sub-coset (pw_merkle_commit: leaf hash + inner levels of W columns x 2^22 rows), thread B evaluates the next sub-coset from
coefficient arrays (pw_lde_subcoset, 2 sub-cosets of a 2^22-row trace: the configs[2] mode of round 5). Each alone, then both.
usage: python tools/overlap_hash_transform.py [W] [log_h] [reps]"""
import sys
import threading
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from powdr_amd import abi, prover

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
log_h = int(sys.argv[2]) if len(sys.argv) > 2 else 22
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
H = 1 << log_h
coef = torch.randint(0, 0x78000001, (W * H,), dtype=torch.int32, device="cuda")
blk_a = torch.randint(0, 0x78000001, (W * H,), dtype=torch.int32, device="cuda")  # the sub-coset being hashed (m = H rows at b = 1)
blk_b = torch.empty(W * H, dtype=torch.int32, device="cuda")                      # the sub-coset being evaluated
dig = torch.empty(2 * H * 8 + 64, dtype=torch.int32, device="cuda")
scr = torch.empty(1 << 15, dtype=torch.int32, device="cuda")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def hash_job(n):
    abi.lib.powdr_gpu_set_stream(streams[0].cuda_stream)
    for _ in range(n):
        abi.check(prover.lib.pw_merkle_commit(blk_a.data_ptr(), H, W, dig.data_ptr()), "pw_merkle_commit")
    streams[0].synchronize()


def lde_job(n):
    abi.lib.powdr_gpu_set_stream(streams[1].cuda_stream)
    for _ in range(n):
        abi.check(prover.lib.pw_lde_subcoset(coef.data_ptr(), W, log_h, 1, 1, scr.data_ptr(), blk_b.data_ptr()), "pw_lde_subcoset")
    streams[1].synchronize()


def timed(jobs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=j, args=(reps,)) for j in jobs]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


timed([hash_job, lde_job])  # warm-up: tables, code objects
h = timed([hash_job])
l = timed([lde_job])
both = timed([hash_job, lde_job])
print(f"W={W} x 2^{log_h} rows, per sub-coset: leaf hash + tree alone {h:.2f} ms, sub-coset transform alone {l:.2f} ms, "
      f"sum {h + l:.2f} ms; on two streams together {both:.2f} ms = {both / (h + l):.3f} of the sum "
      f"({'gain' if both < h + l else 'loss'} {abs(h + l - both):.2f} ms, {abs(1 - both / (h + l)) * 100:.1f} %)")
