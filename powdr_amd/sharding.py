"""Multi-GPU sharding of the proving path: independent segments / AIR proofs across ranks.

The reference proves segments strictly sequentially on one device
(/root/reference/openvm/src/trace_generation.rs:111-141); once metered execution has fixed the
segment boundaries (:107-109) the per-segment proofs are independent, so the MI355X design shards
them one process per GPU with NO data-path collective. The only exchange is the final
"commitment merge": every rank contributes the 8-word trace commitment of each proof it made and
all ranks receive the full, segment-ordered list (RCCL all-gather over xGMI on GPUs: 32 bytes
per proof; gloo in the CPU tests). `torch.distributed` is plumbing here.
"""
from __future__ import annotations

import numpy as np
import torch


def assign_units(cells_per_unit, world_size: int):
    """Largest-first greedy balance of proof units (segments or AIRs) by cell count.
    Returns a list of unit-index lists, one per rank; deterministic."""
    loads = [0] * world_size
    out = [[] for _ in range(world_size)]
    for u in sorted(range(len(cells_per_unit)), key=lambda i: (-cells_per_unit[i], i)):
        r = min(range(world_size), key=lambda k: (loads[k], k))
        out[r].append(u)
        loads[r] += cells_per_unit[u]
    for lst in out:
        lst.sort()
    return out


def merge_commitments(local_units, local_roots, n_units: int, group=None) -> np.ndarray:
    """All-gather the per-proof commitments. local_units: indices this rank proved;
    local_roots: uint32 [len(local_units), 8]. Returns uint32 [n_units, 8] on every rank,
    rows of units nobody proved are zero."""
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    roots = np.zeros((n_units, 8), dtype=np.uint32)
    mine = np.asarray(local_roots, dtype=np.uint32).reshape(-1, 8)
    for u, r in zip(local_units, mine):
        roots[u] = r
    if world == 1:
        return roots
    backend = dist.get_backend(group)
    device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    # every rank owns disjoint rows, so a sum-free gather of full tables + OR-merge is exact
    t = torch.from_numpy(roots.view(np.int32).reshape(-1).copy()).to(device)
    gathered = torch.empty(world * t.numel(), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(gathered, t, group=group)
    g = gathered.cpu().numpy().view(np.uint32).reshape(world, n_units, 8)
    return np.bitwise_or.reduce(g, axis=0)


def commitment_digest(roots: np.ndarray) -> np.ndarray:
    """One 8-word digest over the ordered list of commitments (binary Poseidon2 tree, an odd node paired with
    zeros; `pw_commitment_digest` of libpowdr_gpu, host code; canonical words). Also the bus seed of a segment."""
    from . import prover

    return prover.commitment_digest(roots)


def allreduce_histograms(histograms, group=None):
    """AIR-level sharding inside one segment (SURVEY.md §8e level 2): the APC chips of a segment share the
    periphery histograms, which are plain integer sums, so ranks that generate different AIRs' traces sum
    their partial histograms once per segment — ~3.5 MiB of u32 counters (2^18 var-range + 2^19 tuple +
    2^17 bitwise), one all-reduce each. In place; tensors may live on the GPU (RCCL) or the CPU (gloo)."""
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return histograms
    for h in histograms:
        # u32 counters are stored as int32 words; wrap-around addition is the same operation
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
    return histograms
