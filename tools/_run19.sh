mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > gpurun_out/r02_pytest_final.log
( timeout 900 python bench.py --steps 10 --warmup 3 ) > gpurun_out/r02_bench_c2.json 2> gpurun_out/r02_bench_c2.err
tail -4 gpurun_out/r02_pytest_final.log
python - <<P
import json
d=json.load(open('gpurun_out/r02_bench_c2.json'))
print(d['value']/1e9, d['ms_per_step'], d['logup']['ms_per_step'], d['multi_segment']['value']/1e9, d['tracegen_callmajor']['gather_ms'], d['tracegen_column_structured']['gather_ms'], d['cpu_baseline']['value'])
r=d['roofline']; print(r['frac'], r['traffic'], r['valu']['frac'], r['whole_step']['frac'])
P
