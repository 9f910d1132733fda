LEGS="--no-cpu-baseline --no-logup-leg --no-segment-leg --no-callmajor-leg --no-copy-ceiling --no-live-pmc"
for c in 4 5 6 3; do
POWDR_NTT_C=$c timeout 300 python bench.py --steps 4 --warmup 2 $LEGS > /tmp/b_$c.json 2>/dev/null
python - <<P
import json
d=json.load(open('/tmp/b_$c.json')); s=d['stage_ms']; print('c=$c', d['ms_per_step'], s['ntt_group_kernel<dif>'], s['lde_fused_kernel'], s['ntt_group_kernel<dit>'])
P
done
