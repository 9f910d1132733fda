"""Build libpowdr_gpu.so (HIP, gfx950) in-tree with hipcc.

`python -m powdr_amd.build` or `__graft_entry__.build()`. hipcc cross-compiles
without a GPU. The shared object lands in powdr_amd/lib/ (git-ignored, but it
travels with gpurun snapshots).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
LIBDIR = ROOT / "lib"
LIB = LIBDIR / "libpowdr_gpu.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-I", str(ROOT.parent / "include")]


def _sources():
    return sorted(CSRC.glob("*.hip")) + sorted((CSRC / "host").glob("*.cpp"))


def _headers():
    return (sorted(CSRC.glob("*.hpp")) + sorted(CSRC.glob("*.inc")) + sorted((CSRC / "host").glob("*.hpp"))
            + sorted((ROOT.parent / "include").glob("*.h")))


def _compile(src: Path, obj: Path):
    cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.suffix == ".cpp" else []) + ["-c", str(src), "-o", str(obj)]
    subprocess.check_call(cmd)


def build(force: bool = False, verbose: bool = False) -> Path:
    LIBDIR.mkdir(exist_ok=True)
    objdir = LIBDIR / "obj"
    objdir.mkdir(exist_ok=True)
    hdr_m = max((h.stat().st_mtime for h in _headers()), default=0.0)
    jobs = []
    objs = []
    for s in _sources():
        o = objdir / (s.stem + ".o")
        objs.append(o)
        if force or not o.exists() or o.stat().st_mtime < max(s.stat().st_mtime, hdr_m):
            jobs.append((s, o))
    if jobs:
        if verbose:
            print("hipcc:", ", ".join(s.name for s, _ in jobs), file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(lambda so: _compile(*so), jobs))
    if jobs or not LIB.exists():
        subprocess.check_call([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB)] + [str(o) for o in objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
