mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q -x -k "logup or selftest or segment or periphery" 2>&1 | tail -4 ) > gpurun_out/r02_pytest14.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-segment-leg --no-callmajor-leg --no-copy-ceiling > gpurun_out/r02_bench_logup.json 2>/dev/null
python - <<P
import json
d=json.load(open('gpurun_out/r02_bench_logup.json')); l=d['logup']; print(d['ms_per_step'], l['ms_per_step'], l['value']/1e9); print(sorted(l['stage_ms'].items(), key=lambda kv:-kv[1])[:10])
P
tail -3 gpurun_out/r02_pytest14.log
