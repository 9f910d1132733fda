"""The dominant kernel alone, for bench.py's live PMC pass: Merkle-commit a random [width x 2^log_n] matrix twice (pw_merkle_commit:
leaf_hash_kernel + the inner levels). usage: python tools/leaf_hash_once.py WIDTH LOG_N"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from powdr_amd import abi, prover

W, log_n = int(sys.argv[1]), int(sys.argv[2])
N = 1 << log_n
m = torch.randint(0, 0x78000001, (W * N,), dtype=torch.int32, device="cuda")
dig = torch.empty(2 * N * 8, dtype=torch.int32, device="cuda")
for _ in range(2):
    abi.check(prover.lib.pw_merkle_commit(m.data_ptr(), N, W, dig.data_ptr()), "pw_merkle_commit")
torch.cuda.synchronize()
print("ok")
