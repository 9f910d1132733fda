"""bench.py prints ONE JSON line with the fields the driver reads (small trace so that it runs in seconds)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.gpu


LINE_LIMIT = 6000


def _no_constants(name):
    raise AssertionError(f"non-strict JSON constant {name} on the bench line")


def check_line(stdout, full_path):
    """The stdout contract: ONE JSON line, the last one (RCCL prints a version banner through C stdio when a communicator
    is created), under LINE_LIMIT bytes, strict JSON (no NaN / Infinity), every string under 120 characters, the contract's fields
    present — and equal (to the 6 digits the line keeps) to the full record written next to it, which is what the tests below read."""
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len([l for l in lines if l.lstrip().startswith("{")]) == 1 and lines[-1].lstrip().startswith("{"), lines
    line = lines[-1]
    assert len(line.encode()) < LINE_LIMIT, len(line)
    c = json.loads(line, parse_constant=_no_constants)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in c, k

    def strings(x):
        if isinstance(x, dict):
            for v in x.values():
                yield from strings(v)
        elif isinstance(x, list):
            for v in x:
                yield from strings(v)
        elif isinstance(x, str):
            yield x

    assert max(len(s_) for s_ in strings(c)) < 120
    assert "workload" in c["config"] and "model" not in c["config"]
    full = json.loads(Path(full_path).read_text(), parse_constant=_no_constants)
    assert c["full_record"] and Path(c["full_record"]).name == Path(full_path).name
    for k in ("value", "ms_per_step"):
        assert abs(c[k] - full[k]) <= 1e-5 * abs(full[k])
    for k in ("unit", "n_gpus", "steps", "warmup", "scaling", "dtype", "data"):
        assert c[k] == full[k]
    if full.get("roofline"):
        for k in ("bound", "kernel", "unit", "peak"):
            assert c["roofline"][k] == full["roofline"][k]
        assert abs(c["roofline"]["frac"] - full["roofline"]["frac"]) <= 1e-5 * full["roofline"]["frac"]
        assert "traffic" in c["roofline"]
    if full.get("cpu_baseline"):
        assert c["cpu_baseline"]["kind"] == full["cpu_baseline"]["kind"] and c["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"]
        assert c["cpu_baseline"]["sample"]
    return c, full


def run_bench(*extra, env=None, argv_prefix=None, base=("--log-height", "12", "--steps", "2", "--warmup", "1", "--cpu-log-height", "8"), timeout=900):
    """Runs bench.py, checks the stdout line (check_line) and returns the FULL record of the run (bench_full.json, --full-out)."""
    import tempfile

    if "torch" in sys.modules:  # this process's cached HBM back to the device: the ranks of the run below share it with us
        import gc

        gc.collect()
        sys.modules["torch"].cuda.empty_cache()
    with tempfile.TemporaryDirectory() as td:
        fp = Path(td) / "bench_full.json"
        cmd = (argv_prefix or [sys.executable]) + [str(ROOT / "bench.py"), *base, "--full-out", str(fp), *extra]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        assert len(out.stderr) < 4000, out.stderr[-1000:]  # the driver's tail holds stdout AND stderr: nothing large may follow the line
        c, full = check_line(out.stdout, fp)
        full["_compact"] = c
        return full


def test_default_line_has_the_contract_fields():
    d = run_bench()
    assert d["metric"].startswith("STARK cells/sec") and d["unit"] == "cells/s"
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "u32" in d["dtype"]
    assert d["value"] > 0 and abs(d["value"] - d["config"]["rows"] * d["config"]["cols"] / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["avg_launch_ms"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "cells/s" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert d["stage_ms"]["leaf_hash_kernel"] > 0 and d["gauges"]["trace_gen_time_ms"] > 0
    # the headline proof includes the AIR's bus interactions (LogUp): SURVEY 8d's 48 + 4 + 44 rho algorithmic bytes per main cell
    assert "LogUp" in d["metric"] and "WITH the LogUp phase" in d["config"]["workload"]
    ws = r["whole_step"]
    assert ws["algo_bytes_per_cell"] > 100 and abs(ws["achieved_GBps"] - d["value"] * ws["algo_bytes_per_cell"] / 1e9) < 1e-6 * ws["achieved_GBps"]
    assert d["stage_ms"].get("logup_perm_kernel", 0) + d["stage_ms"].get("logup_perm_jit_kernel", 0) > 0
    # the second timed leg: the same step with the constraints-only proof (the round-1/2 headline)
    co = d["constraints_only"]
    assert d["logup"] is None and co["value"] > d["value"] and co["perm_cols"] == 0 and co["algo_bytes_per_main_cell"] == 48.0
    assert co["proof_bytes"] < d["config"]["proof_bytes"] and "CONSTRAINTS-ONLY" in co["note"]
    # the third leg: multi-AIR segments (C4 shape), one proof per segment, strong scaling over a fixed number of segments
    ms = d["multi_segment"]
    assert ms["shape"] == "C4" and ms["scaling"] == "strong" and ms["n_segments"] == 8 and ms["airs_per_segment"] == 26
    assert ms["airs_by_role"] == {"apc": 10, "instruction": 13, "periphery": 3}
    assert ms["value"] > 0 and ms["proof_bytes_per_segment"] > 0 and ms["segments_on_rank0"] == 8
    # an honest workload: traces generated inside the timed region, the proof verifies, the constraints hold, the lookup buses balance
    assert ms["verify_rc"] == 0 and ms["constraint_violations"] == 0 and ms["lookup_balance"]["verify_rc"] == 0
    # ... for EVERY one of the 8 segments, which differ in their inputs: eight different commitments in the merge
    assert ms["distinct_segments"] is True and ms["distinct_commitments_in_merge"] == 8 and ms["distinct_commitments"] == 8
    assert ms["checked_segments"] == list(range(8)) and ms["verify_rc_by_segment"] == [0] * 8 and ms["lookup_balance"]["verify_rc_by_segment"] == [0] * 8
    assert ms["input_staging_ms_per_segment"] > 0
    # ... and in their SHAPE: per-segment trace heights in the full record, a short tail, the proofs checked per segment
    assert ms["segment_shapes"] == "own" and len(ms["heights_by_segment"]) == 8 and len({tuple(h) for h in ms["heights_by_segment"]}) >= 7
    apc = [i for i, n in enumerate(ms["air_names"]) if n.startswith("apc")]
    assert ms["cells_by_segment"][0] == ms["cells_per_segment"] and all(ms["heights_by_segment"][7][i] <= ms["heights_by_segment"][0][i] - 3 for i in apc)
    assert d["_compact"]["multi_segment"]["cells_by_segment"] == ms["cells_by_segment"]
    assert ms["trace_gen_ms_per_segment"] > 0 and ms["prove_ms_per_segment"] > 0
    # the dominant kernel's HBM bytes and VALU instructions are measured in the run itself (two rocprofv3 --pmc passes)
    if r["kernel"] == "leaf_hash_kernel":
        import shutil
        if shutil.which("rocprofv3"):
            lde_bytes = 2 * d["config"]["rows"] * (d["config"]["cols"] + d["config"]["perm_cols"]) * 4  # trace + permutation LDEs
            assert r["traffic"] is not None and "IN THIS RUN" in r["traffic_unit"]
            assert 0.5 * lde_bytes < r["traffic"] < 2.0 * lde_bytes  # the kernel reads the LDE matrix once
            assert r["valu"] and "measured in this run" in r["valu"]["source"] and 2500 < r["valu"]["valu_instr_per_perm"] < 6000
    # no per-kernel HBM fraction for the quotient kernel (it reads only the referenced columns)
    assert "quotient_kernel" not in d["roofline_by_kernel"]
    # BASELINE configs[2] rides in the same line (skipped with the reason on a small trace / little free HBM), and how the library was built
    assert d["build"]["lib"].endswith("libpowdr_gpu.so") and d["build"]["built_after_newest_source"] in (True, False)
    assert "c3" in d


def test_constraints_only_headline():
    d = run_bench("--constraints-only", "--no-cpu-baseline", "--no-segment-leg")
    assert "constraints-only" in d["metric"] and "CONSTRAINTS-ONLY" in d["config"]["workload"] and d["roofline"]["whole_step"]["algo_bytes_per_cell"] == 48.0
    lg = d["logup"]
    assert d["constraints_only"] is None and lg["value"] > 0 and lg["value"] < d["value"] and lg["perm_cols"] == 4 * (lg["interaction_groups"] + 1)
    assert lg["proof_bytes"] > d["config"]["proof_bytes"]


def test_logup_and_partial_calls_modes_run():
    d = run_bench("--logup", "--no-cpu-baseline", "--calls-fraction", "0.75")
    assert d["cpu_baseline"] is None and "LogUp" in d["config"]["workload"] and "3072 APC calls" in d["config"]["workload"]
    assert d["stage_ms"].get("logup_perm_kernel", 0) + d["stage_ms"].get("logup_perm_jit_kernel", 0) > 0 and d["gauges"]["perm_trace_time_ms"] > 0


def test_segment_shapes_run_as_the_main_workload():
    d = run_bench("--shape", "C4", "--segments", "3", "--segment-log-height", "11", "--no-cpu-baseline")
    assert d["scaling"] == "strong" and "multi-segment" in d["metric"] and d["multi_segment"]["n_segments"] == 3
    ms = d["multi_segment"]
    assert d["value"] > 0 and abs(d["value"] - sum(ms["cells_by_segment"]) / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    # every segment has its OWN trace heights: segment 0 at the caps, the last one the short tail
    assert ms["segment_shapes"] == "own" and len(ms["heights_by_segment"]) == 3 and len({tuple(h) for h in ms["heights_by_segment"]}) == 3
    apc = [i for i, n in enumerate(ms["air_names"]) if n.startswith("apc")]
    assert ms["cells_by_segment"][0] == ms["cells_per_segment"] and all(ms["heights_by_segment"][2][i] <= ms["heights_by_segment"][0][i] - 3 for i in apc)
    assert ms["verify_rc_by_segment"] == [0] * 3 and ms["constraint_violations"] == 0 and ms["lookup_balance"]["verify_rc"] == 0
    assert ms["traces_handed_over"] is True and ms["device_budget_bytes"] > 0
    # round 5's form (equal shapes) is still there
    e = run_bench("--shape", "C4", "--segments", "2", "--segment-log-height", "10", "--no-cpu-baseline", "--segment-shapes", "equal")["multi_segment"]
    assert e["segment_shapes"] == "equal" and e["cells_by_segment"] == [e["cells_per_segment"]] * 2 and e["verify_rc"] == 0
    d = run_bench("--shape", "C5", "--segments", "2", "--segment-log-height", "12", "--logup")
    assert d["multi_segment"]["logup"] is True and d["multi_segment"]["airs_per_segment"] >= 25 and d["multi_segment"]["verify_rc"] == 0


def test_inproc_multi_device_form():
    """bench.py --inproc: the multi-segment workload through pw_prove_segments_multi (one process, a host thread per worker,
    RCCL all-gather of the commitments) — here two workers that share the box's one GPU."""
    import os

    # (POWDR_MULTI_STEAL=0: the placement assertions below are about the PLAN; two workers that share one GPU steal from each other at random)
    d = run_bench("--shape", "C4", "--segments", "5", "--segment-log-height", "11", "--gpus", "2", "--inproc", "--no-cpu-baseline",
                  env=dict(os.environ, POWDR_MULTI_STEAL="0"))
    ms = d["multi_segment"]
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and ms["workers"] == 2 and ms["devices"] == [0, 0]
    assert sum(ms["segments_per_worker"]) == 5 and min(ms["segments_per_worker"]) >= 1 and ms["logup"] is True
    assert "RCCL" in ms["commitment_merge"] or "host" in ms["commitment_merge"]
    assert abs(d["value"] - sum(ms["cells_by_segment"]) / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    # placement by CELLS of segments that differ in shape: the workers' cell counts are close, their segment counts need not be
    assert sum(ms["cells_per_worker"]) == sum(ms["cells_by_segment"]) and max(ms["cells_per_worker"]) <= 1.5 * min(ms["cells_per_worker"])
    assert ms["verify_rc"] == 0 and ms["constraint_violations"] == 0


def test_two_ranks_on_one_gpu_weak_and_strong_legs():
    """The N > 1 code path of bench.py as the driver launches it (torch.distributed.run, one process per rank), with both
    ranks on GPU 0 over gloo (POWDR_DIST_BACKEND): weak scaling of the single-AIR step in `value`, strong scaling of the
    multi-segment leg (8 segments -> 4 per rank), commitments all-gathered inside the timed regions."""
    import os
    import socket

    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, POWDR_DIST_BACKEND="gloo")
    d = run_bench("--gpus", "2", "--no-cpu-baseline", env=env, base=("--log-height", "12", "--steps", "2", "--warmup", "1"),
                  argv_prefix=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                               "--master-port", str(port)])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * d["config"]["rows"] * d["config"]["cols"] / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    ms = d["multi_segment"]
    assert ms["scaling"] == "strong" and ms["n_segments"] == 8 and ms["value"] > 0
    assert ms["segments_on_rank0"] == len(ms["placement"][0]) and sorted(ms["placement"][0] + ms["placement"][1]) == list(range(8))
    assert max(ms["cells_per_rank"]) <= 1.25 * min(ms["cells_per_rank"])  # largest first by cells: unequal counts, balanced work
    assert d["constraints_only"]["value"] > 0


def test_plain_command_with_gpus_2_produces_a_two_rank_line():
    """`python bench.py --gpus 2 ...` as a plain subprocess — no torch.distributed.run around it — launches its two ranks
    itself (here both on GPU 0 over gloo, the one-GPU test hook) and the line says so: n_gpus, the communicator's rank count, per-rank
    times, the weak C2 value AND the strong multi-segment value in one record."""
    import os

    d = run_bench("--gpus", "2", "--no-cpu-baseline", env=dict(os.environ, POWDR_DIST_BACKEND="gloo"), base=("--log-height", "12", "--steps", "2", "--warmup", "1"))
    c = d["_compact"]
    assert c["n_gpus"] == 2 and c["rccl_ranks"] == 2 and len(c["per_rank_ms"]) == 2 and c["strong_scaling_value"] > 0 and c["multi_segment"]["value"] > 0
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["comm"]["launch"] == "self" and d["comm"]["backend"] == "gloo"
    assert len(d["per_rank_ms"]) == 2 and abs(max(d["per_rank_ms"]) - d["ms_per_step"]) < 1e-6 * d["ms_per_step"]
    assert d["scaling"] == "weak" and d["strong_scaling_value"] == d["multi_segment"]["value"] > 0
    assert d["multi_segment"]["ranks"] == 2 and len(d["multi_segment"]["per_rank_ms"]) == 2
    # asked for more GPUs than the box has, without the hook: refused, not relabelled
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "64", "--log-height", "12"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 2 and "only" in out.stderr


def test_eight_ranks_on_one_gpu_with_a_cold_kernel_cache(tmp_path):
    """`python bench.py --gpus 8 ...` as typed — eight ranks (all on GPU 0 over gloo, the one-GPU test hook), every
    prover specialised (POWDR_JIT=1) against ONE cold on-disk cache: the line says n_gpus 8 / 8 ranks / 8 per-rank times, the eight
    segments of the strong-scaling leg land one per rank with eight different commitments, and every translation unit was compiled by
    exactly one rank (the others loaded it). Not a multi-GPU measurement — RCCL has still not seen two devices."""
    import os

    # (specialised from 2^11 rows on: the T1 headline — the SAME AIR on every rank, so rank 0 compiles its units and the seven others load
    # them — and not the 26 AIRs of every segment shape, whose ~250 translation units cost a cold box ~30 s of hiprtc; the cross-process
    # sharing of a cold cache is also tested without a GPU: tests/test_jit.py::test_ranks_that_share_a_cold_cache_compile_every_unit_once)
    env = dict(os.environ, POWDR_DIST_BACKEND="gloo", POWDR_JIT_MIN_LOG_HEIGHT="11", POWDR_JIT_CACHE_DIR=str(tmp_path / "jit"))
    env.pop("POWDR_JIT", None)
    # (headline shape T1: on a box whose hiprtc / comgr caches are cold too, the C2 headline's 350 translation units alone take ~115 s
    # of the ranks' first step — profiles/r06_eight_ranks_phases.txt; the segment leg's 291 units keep the cold-cache claim honest)
    d = run_bench("--gpus", "8", "--shape", "T1", "--no-cpu-baseline", "--no-c3-leg", "--no-logup-leg", "--no-callmajor-leg", "--segment-log-height", "10",
                  env=env, base=("--log-height", "12", "--steps", "2", "--warmup", "1"), timeout=1500)
    c = d["_compact"]
    assert c["n_gpus"] == 8 and c["rccl_ranks"] == 8 and len(c["per_rank_ms"]) == 8 and c["comm"]["backend"] == "gloo"
    assert d["comm"]["launch"] == "self" and d["scaling"] == "weak"
    assert abs(d["value"] - 8 * d["config"]["rows"] * d["config"]["cols"] / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    ms = d["multi_segment"]
    assert ms["ranks"] == 8 and ms["n_segments"] == 8 and ms["segments_on_rank0"] == 1 and len(ms["per_rank_ms"]) == 8
    assert ms["distinct_commitments_in_merge"] == 8 and ms["verify_rc"] == 0 and ms["lookup_balance"]["verify_rc"] == 0
    jc = d["jit_cache"]
    assert len(jc["units_compiled"]) == 8 and sum(jc["units_compiled"]) > 0
    n_entries = len(list((tmp_path / "jit").glob("*.pwjc")))
    assert sum(jc["units_compiled"]) == n_entries, (jc, n_entries)  # one compile per distinct unit over the eight ranks
    # ... and every rank got its kernels: compiled here or loaded from what another rank compiled
    assert all(a + b > 0 for a, b in zip(jc["units_compiled"], jc["units_from_disk"])) and sum(jc["units_from_disk"]) >= n_entries
    assert c["jit_cache"]["units_compiled"] == n_entries


def test_inproc_eight_workers_on_one_device():
    """pw_prove_segments_multi with EIGHT worker threads on one device (the in-process form of an 8-GPU node): 8 segments, one each."""
    d = run_bench("--shape", "C4", "--segments", "8", "--segment-log-height", "10", "--gpus", "8", "--inproc", "--no-cpu-baseline")
    ms = d["multi_segment"]
    assert d["n_gpus"] == 8 and ms["workers"] == 8 and ms["devices"] == [0] * 8 and sum(ms["segments_per_worker"]) == 8
    assert ms["verify_rc"] == 0 and ms["constraint_violations"] == 0 and ms["distinct_segments"] is True
