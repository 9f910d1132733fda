#!/bin/bash
# round 3, call 8: the thirteen chips on the device + the records leg of the bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_original_chips.py tests/test_abi_load.py -m gpu -x -q 2>&1 | tail -15
timeout 600 python bench.py --steps 2 --warmup 1 --no-segment-leg > gpurun_out/r3_call8_bench.json 2> gpurun_out/r3_call8_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3_call8_bench.json") if l.startswith("{")][-1])
print("headline", d["value"], d["ms_per_step"])
print("records", d.get("tracegen_from_records"))
t = d.get("kernel_ms_per_step", {})
print({k: round(v, 2) for k, v in t.items() if "ext_dot" in k or "deep" in k})
PY
tail -3 gpurun_out/r3_call8_bench.err
