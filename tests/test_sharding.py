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


def _segment_worker(rank, world, port, out_dir):
    """One LogUp segment, two AIRs on one bus, one AIR per rank; the oracle stands in for the GPU prover."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import stark_model as sm
    from tests.test_oracle_stark import balanced_bus_pair

    no_cons = (np.zeros(0, np.uint32), np.zeros((0, 2), np.uint32))
    airs = balanced_bus_pair(4, seed=21)
    mine = sharding.assign_units([3 << 4, 3 << 4], world)[rank]

    def commit(u):
        t, it = airs[u]
        return sm.prove_logup(t.reshape(-1), 3, 4, *no_cons, *it, num_queries=0)[7:15]

    def prove(u, seed):
        t, it = airs[u]
        return sm.prove_logup(t.reshape(-1), 3, 4, *no_cons, *it, num_queries=4, bus_seed=seed)

    seed, proofs = sharding.prove_segment_sharded(mine, 2, commit, prove)
    for u, pf in proofs.items():
        np.save(os.path.join(out_dir, f"proof_{u}.npy"), pf)
    np.save(os.path.join(out_dir, f"seed_{rank}.npy"), seed)
    dist.barrier()
    dist.destroy_process_group()


def test_logup_segment_sharded_over_two_ranks_gloo(tmp_path):
    """N>1 path of a bus-sharing segment: trace roots all-gathered, same seed on both ranks, the host verifier
    accepts the two proofs as one balanced segment."""
    from powdr_amd import prover
    from tests.test_oracle_stark import balanced_bus_pair

    port = _free_port()
    mp.spawn(_segment_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    s0, s1 = np.load(tmp_path / "seed_0.npy"), np.load(tmp_path / "seed_1.npy")
    assert (s0 == s1).all() and s0.any()
    proofs = [np.load(tmp_path / f"proof_{u}.npy") for u in range(2)]
    no_cons = (np.zeros(0, np.uint32), np.zeros((0, 2), np.uint32))
    descs = [(3, 4, *no_cons, it) for _, it in balanced_bus_pair(4, seed=21)]
    rc, total = prover.verify_airs(descs, proofs, num_queries=4, shared_bus_seed=True, check_balance=True)
    assert rc == 0 and (total == 0).all()
    assert (proofs[0][15:23] == s0).all() and (proofs[1][15:23] == s0).all()


def _strong_scaling_worker(rank, world, port, out_dir):
    """The bench's C4 / C5 path (bench.py segment_bench) on CPU: a fixed list of segments, each proven as ONE segment proof
    (the oracle stands in for pw_prove_segment), placed by cells, main commitments all-gathered."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import stark_model as sm
    from tests.test_segment_proof import synthetic_airs

    segments = [synthetic_airs([("T0", 10 + u), ("T1", 40 * (u + 1))], seed0=50 + u) for u in range(5)]
    cells = [sum(a[1] << a[2] for a in seg) for seg in segments]
    proofs = {}

    def prove_one(u):
        pf = sm.prove_segment(segments[u], num_queries=3, logup=False)
        proofs[u] = pf
        hdr = 5 + 4 * len(segments[u])
        return pf[hdr:hdr + 8]

    mine, merged = sharding.prove_segments_sharded(cells, prove_one, rank, world)
    np.save(os.path.join(out_dir, f"merged_{rank}.npy"), merged)
    np.save(os.path.join(out_dir, f"mine_{rank}.npy"), np.array(mine))
    for u, pf in proofs.items():
        np.save(os.path.join(out_dir, f"seg_{u}.npy"), pf)
    dist.barrier()
    dist.destroy_process_group()


def test_strong_scaling_over_segments_world_size_2_gloo(tmp_path):
    from powdr_amd import prover
    from tests.test_segment_proof import descs_of, synthetic_airs

    port = _free_port()
    mp.spawn(_strong_scaling_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    m0, m1 = np.load(tmp_path / "merged_0.npy"), np.load(tmp_path / "merged_1.npy")
    assert m0.shape == (5, 8) and (m0 == m1).all() and m0.any(axis=1).all()
    mine = [set(np.load(tmp_path / f"mine_{r}.npy").tolist()) for r in range(2)]
    assert mine[0] | mine[1] == set(range(5)) and not (mine[0] & mine[1]) and all(mine)
    # balance by cells: neither rank carries more than the other plus the largest segment
    for u in range(5):
        seg = synthetic_airs([("T0", 10 + u), ("T1", 40 * (u + 1))], seed0=50 + u)
        pf = np.load(tmp_path / f"seg_{u}.npy")
        assert prover.verify_segment(descs_of(seg), pf, 3, 0, False)[0] == 0  # the product's host verifier accepts every segment
        assert (pf[5 + 4 * 2:5 + 4 * 2 + 8] == m0[u]).all()  # the merged row IS that segment's main commitment


def test_c_abi_placement_equals_python_placement():
    """pw_assign_units (the placement pw_prove_segments_multi uses) == sharding.assign_units, ties included."""
    from powdr_amd import prover

    rng = np.random.default_rng(3)
    for n, w in [(0, 3), (1, 1), (5, 8), (9, 2), (40, 8), (17, 5)]:
        cells = [int(x) for x in rng.integers(1, 50, n)]  # many ties
        owner = prover.assign_units(cells, w)
        want = sharding.assign_units(cells, w)
        assert [sorted(np.flatnonzero(owner == k).tolist()) for k in range(w)] == want


@pytest.mark.gpu
@pytest.mark.parametrize("n_workers,steal", [(1, True), (3, False), (3, True)])
def test_prove_segments_multi_on_one_gpu(n_workers, steal, monkeypatch):
    """pw_prove_segments_multi with every worker on GPU 0 (one-GPU box): host threads with their own streams and prover
    replicas prove 7 segments of different sizes concurrently; commitments and proofs equal the ones made one after the
    other; the merge went through an RCCL communicator of size 1 (or says that RCCL is missing)."""
    import threading

    import torch
    from powdr_amd import prover, synth

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU (run with -m gpu on the GPU box)")
    P = 0x78000001
    shapes = [(40, 12), (9, 7), (120, 10), (33, 13), (5, 4), (64, 11), (17, 9)]
    progs = [synth.random_air_programs(w, 5, 8, seed=k) for k, (w, lh) in enumerate(shapes)]
    traces = []
    for w, lh in shapes:
        t = torch.empty(w << lh, dtype=torch.int32, device="cuda")
        t.random_(0, P)
        traces.append(t)
    torch.cuda.synchronize()

    def make_provers():
        return [prover.Prover(w, bc, sp, num_queries=4, pow_bits=2, interactions=it) for (w, lh), (bc, sp, it) in zip(shapes, progs)]

    # a "segment" here = two AIRs (s and its neighbour): one pw_prove_segment proof each
    def prove_with(provers, s):
        a, b = s, (s + 1) % len(shapes)
        return prover.prove_segment([(provers[a], traces[a].data_ptr(), shapes[a][1]), (provers[b], traces[b].data_ptr(), shapes[b][1])], logup=True)

    seq = make_provers()
    want = [prove_with(seq, s) for s in range(len(shapes))]
    hdr = 5 + 4 * 2
    per_worker = [make_provers() for _ in range(n_workers)]
    proofs, lock = {}, threading.Lock()

    def prove_segment(s, worker, device):
        assert device == 0
        pf = prove_with(per_worker[worker], s)
        with lock:
            proofs[s] = (worker, pf)
        return pf[hdr:hdr + 8]

    cells = [(shapes[s][0] << shapes[s][1]) + (shapes[(s + 1) % 7][0] << shapes[(s + 1) % 7][1]) for s in range(7)]
    monkeypatch.setenv("POWDR_MULTI_STEAL", "1" if steal else "0")
    commitments, owner, merge = prover.prove_segments_multi([0] * n_workers, cells, prove_segment)
    assert merge in (1, 2)
    if steal and n_workers > 1:
        # the placement by cells is the plan; a worker that runs dry steals the smallest unstarted segment of the busiest queue: whoever
        # proved a segment is reported as its owner, every segment is proven exactly once (the proofs below), by a worker that exists
        assert len(proofs) == 7 and all(0 <= int(o) < n_workers for o in owner)
    else:
        assert (owner == prover.assign_units(cells, n_workers)).all() and len(set(owner.tolist())) == n_workers
    for s in range(7):
        assert proofs[s][0] == owner[s]
        assert (proofs[s][1] == want[s]).all()
        assert (commitments[s] == want[s][hdr:hdr + 8]).all()
    for ps in per_worker + [seq]:
        for p in ps:
            p.close()


def test_placement_of_segments_with_their_own_shapes():
    """The cell counts of the bench's eight C4 segments with their OWN trace heights (profiles/r06_c4_budget_sweep.txt: the capped
    segment, six with one chip at its cap, the short tail) placed on 1 / 2 / 4 / 8 ranks: unequal segment counts, loads within the
    largest segment of each other; on 8 ranks the step is the capped segment (what strong scaling over unequal segments costs)."""
    cells = [3267841024, 2474124800, 2476166144, 2145864704, 2578982400, 2605196800, 2542282240, 299164864]
    for world in (1, 2, 4, 8):
        parts = sharding.assign_units(cells, world)
        loads = [sum(cells[u] for u in p) for p in parts]
        assert sorted(u for p in parts for u in p) == list(range(8))
        assert max(loads) - min(loads) <= max(cells)
        if world == 2:
            assert max(loads) / (sum(cells) / 2) < 1.07  # (the best split of these eight is 1.05)
        if world == 4:
            assert sorted(len(p) for p in parts) == [2, 2, 2, 2] and max(loads) / (sum(cells) / 4) < 1.10
        if world == 8:
            assert max(loads) == cells[0]  # 8 ranks: 18.39 G cells in the time of 3.27 G -> at most 5.6x of one rank's rate


def test_segment_shape_draws():
    """HonestSegment.draw_shape's rule (powdr_amd/segment_workload.draw_segment_shape) without a GPU: segment 0 at the caps, the last one
    a tail at <= 1/8, the others with one chip at its cap and the rest within the two octaves below theirs; deterministic per (seed, u);
    never fewer than 3 calls for an APC chip (its trace is next_pow2(calls) rows and the provers want >= 4)."""
    from powdr_amd.segment_workload import draw_segment_shape

    caps, icap, n = [1 << 20] * 10, 1 << 10, 8
    shapes = [draw_segment_shape(0, u, n, caps, icap) for u in range(n)]
    assert shapes[0]["apc_calls"] == caps and shapes[0]["instr_calls"] == icap
    assert all(c <= cap // 8 for c, cap in zip(shapes[n - 1]["apc_calls"], caps)) and shapes[n - 1]["instr_calls"] <= icap // 8
    for u in range(1, n - 1):
        f = [c / cap for c, cap in zip(shapes[u]["apc_calls"], caps)] + [shapes[u]["instr_calls"] / icap]
        assert max(f) == 1.0 and min(f) >= 0.2499 and sum(x == 1.0 for x in f) == 1
    assert shapes == [draw_segment_shape(0, u, n, caps, icap) for u in range(n)]
    assert shapes[3] != draw_segment_shape(1, 3, n, caps, icap)
    tiny = draw_segment_shape(0, 7, 8, [4, 4, 64], 4)
    assert all(c >= 3 for c in tiny["apc_calls"]) and tiny["instr_calls"] >= 1
