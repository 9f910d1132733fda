#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_prover_gpu.py tests/test_original_chips.py -m gpu -x -q 2>&1 | tail -6
timeout 600 python bench.py --steps 3 --warmup 1 --no-segment-leg > gpurun_out/r3_call13_bench.json 2> gpurun_out/r3_call13_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3_call13_bench.json") if l.startswith("{")][-1])
print("headline", d["value"], d["ms_per_step"])
print({k: round(v, 2) for k, v in d["stage_ms"].items() if v > 1})
print("records", d["tracegen_from_records"]["fused_ms"], d["tracegen_from_records"].get("step_ms_with_trace_from_records"))
PY
tail -3 gpurun_out/r3_call13_bench.err
