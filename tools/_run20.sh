mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_prover_gpu.py tests/test_segment_proof.py -m gpu -q -x 2>&1 | tail -4 ) > gpurun_out/r02_pytest20.log
( timeout 600 python tools/bench_ntt.py 2>&1 ) > gpurun_out/r02_bench_ntt.txt
LEGS="--no-cpu-baseline --no-logup-leg --no-segment-leg --no-callmajor-leg --no-copy-ceiling"
timeout 300 python bench.py --steps 5 --warmup 2 $LEGS > gpurun_out/r02_bench_tw.json 2>/dev/null
python - <<P
import json
d=json.load(open('gpurun_out/r02_bench_tw.json')); s=d['stage_ms']; print(d['value']/1e9, d['ms_per_step'], s['lde_fused_kernel'], s['ntt_group_kernel<dif>'], s['ntt_group_kernel<dit>'], s['leaf_hash_kernel'])
P
grep FUSED gpurun_out/r02_bench_ntt.txt
tail -3 gpurun_out/r02_pytest20.log
