"""The N>1 path on CPU: unit assignment and the commitment merge with world_size 2 over gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from powdr_amd import sharding


def test_assign_units_balances_and_covers():
    cells = [2022 << 20] * 5 + [819 << 16] * 3 + [446 << 10] * 9
    for world in (1, 2, 4, 8):
        parts = sharding.assign_units(cells, world)
        assert sorted(u for p in parts for u in p) == list(range(len(cells)))
        loads = [sum(cells[u] for u in p) for p in parts]
        assert max(loads) - min(loads) <= max(cells)
    assert sharding.assign_units([], 4) == [[], [], [], []]
    assert sharding.assign_units([5, 5, 5], 8)[3:] == [[]] * 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_units, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cells = [(u % 3 + 1) << 12 for u in range(n_units)]
    mine = sharding.assign_units(cells, world)[rank]
    # a deterministic stand-in for the trace root of unit u (what pw_prover_prove puts at proof[6:14])
    roots = np.array([[(u * 8 + k + 1) * 2654435761 % 0x78000001 for k in range(8)] for u in mine], dtype=np.uint32).reshape(-1, 8)
    merged = sharding.merge_commitments(mine, roots, n_units)
    np.save(os.path.join(out_dir, f"merged_{rank}.npy"), merged)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_units", [1, 5, 8])
def test_commitment_merge_world_size_2_gloo(tmp_path, n_units):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_units, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "merged_0.npy")
    b = np.load(tmp_path / "merged_1.npy")
    want = np.array([[(u * 8 + k + 1) * 2654435761 % 0x78000001 for k in range(8)] for u in range(n_units)], dtype=np.uint32)
    assert (a == want).all() and (b == want).all()
    # the merged list hashes to the same digest on every rank
    assert (sharding.commitment_digest(a) == sharding.commitment_digest(b)).all()
    assert sharding.commitment_digest(a).shape == (8,)


def _hist_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    hs = [torch.randint(0, 2**31 - 1, (n,), dtype=torch.int32, generator=g) for n in (1 << 10, 1 << 11, 1 << 9)]
    sharding.allreduce_histograms(hs)
    torch.save(hs, os.path.join(out_dir, f"h_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_histogram_allreduce_world_size_2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_hist_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(tmp_path / "h_0.pt")
    b = torch.load(tmp_path / "h_1.pt")
    for k, n in enumerate((1 << 10, 1 << 11, 1 << 9)):
        parts = [torch.randint(0, 2**31 - 1, (n,), dtype=torch.int32, generator=torch.Generator().manual_seed(100 + r)) if k == 0 else None
                 for r in range(2)]
        assert torch.equal(a[k], b[k])
    # exact u32 wrap-around sum of the two ranks' first histograms
    g0, g1 = torch.Generator().manual_seed(100), torch.Generator().manual_seed(101)
    x0 = torch.randint(0, 2**31 - 1, (1 << 10,), dtype=torch.int32, generator=g0)
    x1 = torch.randint(0, 2**31 - 1, (1 << 10,), dtype=torch.int32, generator=g1)
    want = ((x0.to(torch.int64) + x1.to(torch.int64)) & 0xFFFFFFFF).numpy().astype(np.uint32)
    assert (a[0].numpy().view(np.uint32) == want).all()
