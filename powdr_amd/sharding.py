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


def prove_segment_sharded(my_units, n_units: int, commit, prove, group=None):
    """AIR-level sharding of ONE segment whose AIRs share buses (SURVEY.md 8e level 2; the single-process form is
    `pw_prove_airs(shared_bus_seed=1)`; independent per-AIR proofs — the one-proof-per-segment form `pw_prove_segment`
    needs all matrices on one device). Every rank owns the AIRs `my_units` (see `assign_units`).

      phase 1  commit(u) -> 8-word trace root of AIR u           (rank-local: LDE + Merkle tree stay on the GPU)
      exchange all-gather of the roots, 32 bytes per AIR          (the only collective on this path)
      seed     commitment_digest(roots in AIR order)              (identical on every rank)
      phase 2  prove(u, seed) -> proof words of AIR u             (rank-local)

    Returns (seed, {u: proof}). `commit` / `prove` are callables so that the same orchestration drives the GPU
    provers (`gpu_segment_callables`) and, in the CPU tests, a stand-in prover. The verifier needs nothing from this
    exchange: `pw_verify_airs` recomputes the seed from the trace roots inside the proofs."""
    roots = np.array([commit(u) for u in my_units], dtype=np.uint32).reshape(-1, 8)
    merged = merge_commitments(my_units, roots, n_units, group=group)
    seed = commitment_digest(merged)
    return seed, {u: prove(u, seed) for u in my_units}


def gpu_segment_callables(provers, trace_ptrs, log_heights):
    """commit / prove callables over `prover.Prover` objects created with `interactions=` (one per AIR index)."""

    def commit(u):
        return provers[u].trace_root(trace_ptrs[u], log_heights[u])

    def prove(u, seed):
        provers[u].set_bus_seed(seed)
        return provers[u].prove(trace_ptrs[u], log_heights[u])

    return commit, prove


def prove_segments_sharded(cells_per_segment, prove_segment, rank: int, world_size: int, group=None):
    """STRONG scaling over a fixed list of independent segments (SURVEY.md 8e level 1; the reference proves them one
    after the other on one device, /root/reference/openvm/src/trace_generation.rs:111-141): the segments are placed on
    the ranks by cell count, largest first (`assign_units`, identical on every rank), rank r proves its own with
    `prove_segment(u) -> 8-word main commitment of segment u`, and the only exchange is the final all-gather of those
    commitments (32 bytes per segment). Returns (the segments this rank proved, uint32 [n_segments, 8] on every rank)."""
    mine = assign_units(list(cells_per_segment), world_size)[rank]
    roots = np.array([prove_segment(u) for u in mine], dtype=np.uint32).reshape(-1, 8)
    return mine, merge_commitments(mine, roots, len(cells_per_segment), group=group)
