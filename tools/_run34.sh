mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_prover_gpu.py -m gpu -q -x -k "lde or LDE or golden or c2_shape or multi_panel" 2>&1 | tail -3 ) > gpurun_out/r02_pytest34.log
LEGS="--no-cpu-baseline --no-logup-leg --no-segment-leg --no-callmajor-leg --no-copy-ceiling --no-live-pmc"
timeout 300 python bench.py --steps 6 --warmup 2 $LEGS > /tmp/b.json 2>/dev/null
python - <<P
import json
d=json.load(open('/tmp/b.json')); s=d['stage_ms']; print(d['value']/1e9, d['ms_per_step'], s['ntt_group_kernel<dif>'], s['lde_fused_kernel'], s['ntt_group_kernel<dit>'], s['leaf_hash_kernel'])
P
tail -2 gpurun_out/r02_pytest34.log
