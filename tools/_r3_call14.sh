#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_prover_gpu.py -m gpu -x -q -k "golden or oracle or logup" 2>&1 | tail -4
timeout 600 python bench.py --steps 3 --warmup 1 --no-segment-leg --no-cpu-baseline --no-callmajor-leg > gpurun_out/r3_call14_bench.json 2> gpurun_out/r3_call14_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3_call14_bench.json") if l.startswith("{")][-1])
print("headline", d["value"], d["ms_per_step"])
print({k: round(v, 2) for k, v in d["stage_ms"].items() if v > 1})
PY
tail -3 gpurun_out/r3_call14_bench.err
