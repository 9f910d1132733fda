mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_tracegen_gpu.py -m gpu -q -x -k "callmajor" 2>&1 | tail -15 ) > gpurun_out/r02_pytest5.log
( timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-logup-leg --no-segment-leg ) > gpurun_out/r02_bench_c2_d.json 2> gpurun_out/r02_bench_d_err.log
tail -4 gpurun_out/r02_pytest5.log; python - <<'P'
import json
d=json.load(open('gpurun_out/r02_bench_c2_d.json'))
print(d['value']/1e9, d['ms_per_step']); print(d['tracegen_callmajor']); print({k:round(v,2) for k,v in d['stage_ms'].items()})
P
tail -3 gpurun_out/r02_bench_d_err.log
