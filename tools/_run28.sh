mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_prover_gpu.py tests/test_segment_proof.py -m gpu -q -x 2>&1 | tail -4 ) > gpurun_out/r02_pytest28.log
: > gpurun_out/r02_bench_ntt.txt
for a in "256 20" "2022 14" "64 12" "512 16" "128 22" "1024 10" "1024 13" "2022 20 3" "600 21 3" "300 18" "300 19"; do
  timeout 300 python tools/bench_ntt.py $a 2>/dev/null >> gpurun_out/r02_bench_ntt.txt
done
LEGS="--no-cpu-baseline --no-logup-leg --no-segment-leg --no-callmajor-leg --no-copy-ceiling --no-live-pmc"
timeout 300 python bench.py --steps 5 --warmup 2 $LEGS > /tmp/b.json 2>/dev/null
python - <<P
import json
d=json.load(open('/tmp/b.json')); s=d['stage_ms']; print(d['value']/1e9, d['ms_per_step'], s['ntt_group_kernel<dif>'], s['lde_fused_kernel'], s['ntt_group_kernel<dit>'])
P
grep FUSED gpurun_out/r02_bench_ntt.txt | cut -c1-90
tail -3 gpurun_out/r02_pytest28.log
