#!/usr/bin/env python3
"""BASELINE configs[2] with its bus argument, alone: bench.py's `c3` leg (streamed proof of 3 731 x 2^22 + 4 632 permutation columns)
without the rest of the bench. Writes gpurun_out/c3_logup.json."""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    torch.cuda.set_device(0)
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    rec = bench.c3_leg(100, 16, steps=steps, constraints_only_too="--no-constraints-only" not in sys.argv, segment_too="--no-segment" not in sys.argv)
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "c3_logup.json").write_text(json.dumps(rec, indent=1))
    slim = {k: v for k, v in rec.items() if k not in ("kernels", "constraints_only")}
    print(json.dumps(slim, indent=1))
    for k, e in sorted(rec.get("kernels", {}).items(), key=lambda kv: -kv[1]["ms"])[:24]:
        print(f"{k:40s} {e['ms']:10.2f} ms  x{e['launches_per_step']:.0f}")
