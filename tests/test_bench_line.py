"""The stdout line of bench.py is the COMPACT record (the driver could not read round 4's 23 KB line): built here from
canned full records — round 4's own line, with non-finite numbers and oversized strings injected — no GPU involved."""
import copy
import json
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def _no_constants(name):
    raise AssertionError(name)


@pytest.fixture(scope="module")
def bench():
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        import bench as b
    finally:
        sys.argv = argv
    return b


@pytest.fixture(scope="module")
def canned():
    return json.loads((ROOT / "profiles" / "r04c_bench_c2.json").read_text())


def test_round4_record_fits(bench, canned):
    assert len(json.dumps(canned)) > 20000  # the line the driver could not read
    c = bench.compact_line(canned)
    text = json.dumps(c, allow_nan=False)
    assert len(text.encode()) < bench.LINE_LIMIT
    d = json.loads(text, parse_constant=_no_constants)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in d
    assert abs(d["value"] - canned["value"]) < 1e-5 * canned["value"] and d["config"]["rows"] == 1 << 20 and d["config"]["cols"] == 2022
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["kernel"] == "leaf_hash_kernel" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    assert r["traffic"] > 0 and r["whole_step"]["frac"] > 0 and r["valu"]["frac"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 256 and cb["value"] > 0 and cb["tuned"]["upper_bound_commit_stages_only"] > 0
    assert d["c3"]["verify_rc"] == 0 and d["c3"]["prove_ms"] > 0 and d["multi_segment"]["verify_rc"] == 0
    assert d["tracegen_from_records"]["timed_step"]["ms_per_step"] > 0 and d["stage_ms"]["leaf_hash_kernel"] > 0
    assert "roofline_by_kernel" not in d and "build" not in d


def test_non_finite_numbers_and_long_strings(bench, canned):
    full = copy.deepcopy(canned)
    full["hbm_copy_GBps_measured"] = float("nan")
    full["roofline"]["traffic"] = float("inf")
    full["config"]["workload"] = "w" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["metric"] = "m" * 500
    full["stage_ms"] = {f"kernel_with_a_rather_long_name_{i}": 1.0 + i for i in range(400)}
    d = json.loads(json.dumps(bench.compact_line(full), allow_nan=False), parse_constant=_no_constants)
    assert d["hbm_copy_GBps_measured"] is None and d["roofline"]["traffic"] is None
    assert len(d["config"]["workload"]) < 120 and len(d["cpu_baseline"]["sample"]) < 120 and len(d["metric"]) < 120
    assert len(json.dumps(d).encode()) < bench.LINE_LIMIT and len(d["stage_ms"]) == 5


def test_emit_writes_the_full_record_and_prints_the_compact_one(bench, canned, tmp_path, capsys, monkeypatch):
    monkeypatch.setattr(bench, "FULL_OUT", str(tmp_path / "full.json"))
    bench.emit(copy.deepcopy(canned))
    out = capsys.readouterr()
    lines = [l for l in out.out.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0].encode()) < bench.LINE_LIMIT and out.err == ""
    d = json.loads(lines[0], parse_constant=_no_constants)
    full = json.loads((tmp_path / "full.json").read_text(), parse_constant=_no_constants)
    assert d["full_record"].endswith("full.json") and full["roofline_by_kernel"] and full["value"] == canned["value"]


def test_segment_shape_and_launch_check_lines(bench):
    seg = dict(metric="STARK cells/sec (trace rows x cols), multi-segment reth-shaped [with the bus argument]", value=5.9e9, unit="cells/s", n_gpus=8,
               steps=2, warmup=1, ms_per_step=100.0, higher_is_better=True, scaling="strong", vs_baseline=None, dtype="u32", data="synthetic",
               rccl_ranks=8, per_rank_ms=[100.0] * 8, config=dict(workload="C5: " + "x" * 400, proof_bytes=1), roofline=dict(bound="hbm", kernel="whole step",
               achieved=1.0, peak=8000.0, unit="GB/s", frac=1 / 8000.0, traffic=None), cpu_baseline=None,
               multi_segment=dict(value=5.9e9, verify_rc=0, n_segments=8, kernels={f"k{i}": dict(ms=1.0, note="n" * 200) for i in range(100)}))
    d = bench.compact_line(seg)
    assert d["n_gpus"] == 8 and len(d["per_rank_ms"]) == 8 and d["cpu_baseline"] is None and "kernels" not in d["multi_segment"]
    assert len(json.dumps(d).encode()) < 2000
    lc = bench.compact_line(dict(launch_check=True, n_gpus=3, ranks=[0, 1, 2], launch="self"))
    assert lc == dict(launch_check=True, n_gpus=3, ranks=[0, 1, 2], launch="self")


def test_a_record_that_cannot_be_shrunk_still_prints_a_line(bench, canned, tmp_path, capsys, monkeypatch):
    """a record that stays above the limit after every optional block went must not turn a 200 s run into an
    AssertionError without a line (simulated with a limit below what the fixed blocks need)."""
    monkeypatch.setattr(bench, "LINE_LIMIT", 1400)
    full = copy.deepcopy(canned)
    d = bench.compact_line(full)
    assert len(json.dumps(d, allow_nan=False).encode()) <= 1400
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data", "config"):
        assert k in d
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0
    assert "c3" not in d and "multi_segment" not in d
    monkeypatch.setattr(bench, "FULL_OUT", str(tmp_path / "full.json"))
    bench.emit(full)
    out = capsys.readouterr()
    assert len([l for l in out.out.splitlines() if l.strip()]) == 1 and "contract fields only" in out.err
