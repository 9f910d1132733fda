mkdir -p gpurun_out
( timeout 640 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -40 ) > gpurun_out/r04b_gpu_suite.txt
tail -5 gpurun_out/r04b_gpu_suite.txt
T0=$(date +%s)
( timeout 420 python bench.py --steps 10 --warmup 3 ) > gpurun_out/r04b_bench_c2.json 2> gpurun_out/r04b_bench_c2.err
echo "default bench wall seconds: $(( $(date +%s) - T0 ))" > gpurun_out/r04b_bench_wall.txt
cat gpurun_out/r04b_bench_wall.txt
python - <<P
import json
try:
    d=json.loads([l for l in open('gpurun_out/r04b_bench_c2.json') if l.startswith('{')][-1])
    print("headline", d['value']/1e9, d['ms_per_step'], "co", d['constraints_only']['ms_per_step'])
    m=d['multi_segment']; print("multi", m['value']/1e9, m.get('verify_rc'), m.get('constraint_violations'), (m.get('lookup_balance') or {}).get('verify_rc'), m.get('prove_ms_per_segment'))
    c=d['c3'] or {}; print("c3", c.get('value'), c.get('prove_ms'), c.get('verify_rc'), (c.get('segment') or {}).get('prove_ms'), (c.get('segment') or {}).get('verify_rc'))
    t=d['tracegen_from_records']; ts=t.get('timed_step') or {}; print("records", t.get('fused_ms'), t.get('step_ms_with_trace_from_records'), ts.get('ms_per_step'), ts.get('error'), {k:v for k,v in (ts.get('stage_ms') or {}).items() if 'apc' in k or 'bus' in k})
except Exception as e:
    print("bench parse failed", e); print(open('gpurun_out/r04b_bench_c2.err').read()[-2000:])
P
bash tools/gaps_c2.sh > /dev/null 2>&1; head -12 gpurun_out/r04_gaps_c2.txt
