# Segment prover checks after a change to segment_prover.hip: its parity tests (oracle words), the honest workload, the C4 / C5 shapes, one C5 bench
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_segment_proof.py tests/test_segment_workload_gpu.py tests/test_baseline_configs_gpu.py -q -m gpu -p no:cacheprovider -k "segment or honest or second_generation or streamed or worker" 2>&1 | tail -15 ) > gpurun_out/r04c_segment_tests.txt
tail -4 gpurun_out/r04c_segment_tests.txt
timeout 150 python bench.py --shape C5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04c_bench_c5.json 2>/dev/null
python - <<P
import json
d=json.loads([l for l in open('gpurun_out/r04c_bench_c5.json') if l.startswith('{')][-1]); m=d.get('multi_segment') or d
print('C5', d['value']/1e9, d['ms_per_step'], m.get('prove_ms_per_segment'), m.get('trace_gen_ms_per_segment'), m.get('verify_rc'), m.get('constraint_violations'), (m.get('lookup_balance') or {}).get('verify_rc'))
P
