mkdir -p gpurun_out
LEGS="--no-cpu-baseline --no-logup-leg --no-segment-leg --no-copy-ceiling"
for sp in 60 70 80 90 100 120; do
POWDR_GATHER_SPARSE_PCT=$sp timeout 300 python bench.py --steps 4 --warmup 2 $LEGS > gpurun_out/r02_bench_sp$sp.json 2>/dev/null
python - <<P
import json
d=json.load(open('gpurun_out/r02_bench_sp$sp.json')); s=d['stage_ms']; print('sparse pct $sp', d['value']/1e9, d['ms_per_step'], s['apc_gather_tile_kernel'], d['tracegen_column_structured']['gather_ms'])
P
done
