"""`python bench.py --gpus N` typed as it is (no launcher around it) must start N ranks itself (it used to run one
GPU silently). --launch-check starts the ranks exactly as a real run does — self-launch under torch.distributed.run, 127.0.0.1
rendezvous — lets them meet over gloo on the CPU and reports who is there, without touching a GPU: this runs here."""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _run(*args, env=None):
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.lstrip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_plain_command_with_gpus_2_starts_two_ranks():
    d = _run("--gpus", "2", "--launch-check")
    assert d["launch_check"] is True and d["n_gpus"] == 2 and d["ranks"] == [0, 1] and d["launch"] == "self"


def test_plain_command_with_gpus_3_starts_three_ranks():
    d = _run("--gpus", "3", "--launch-check")
    assert d["n_gpus"] == 3 and d["ranks"] == [0, 1, 2]


def test_single_process_and_external_launcher_are_left_alone():
    assert _run("--launch-check")["n_gpus"] == 1
    import os
    import socket

    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT, env=dict(os.environ))
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.lstrip().startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["launch"] == "external"


def test_rank_count_must_match_gpus():
    """A launcher that started a different number of ranks than --gpus says is an error, not a silently relabelled line."""
    import os

    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300,
                         cwd=ROOT, env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert out.returncode == 2 and "--gpus 4" in out.stderr
