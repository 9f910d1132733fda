mkdir -p gpurun_out
LEGS="--no-cpu-baseline --no-logup-leg --no-segment-leg --no-callmajor-leg --no-copy-ceiling"
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 ) > gpurun_out/r02_pytest12.log
timeout 300 python bench.py --steps 5 --warmup 2 $LEGS > gpurun_out/r02_bench_signed.json 2>/dev/null
python - <<P
import json
d=json.load(open('gpurun_out/r02_bench_signed.json')); print(d['value']/1e9, d['ms_per_step'], d['stage_ms']['leaf_hash_kernel'], d['stage_ms']['compress_kernel'], d['stage_ms']['compress_tail_kernel'])
P
timeout 300 python bench.py --shape C4 --steps 2 --warmup 1 > gpurun_out/r02_bench_c4_signed.json 2>/dev/null
python - <<P
import json
d=json.load(open('gpurun_out/r02_bench_c4_signed.json')); print('C4', d['value']/1e9, d['multi_segment']['stage_ms_rank0']['leaf_hash_kernel'])
P
tail -4 gpurun_out/r02_pytest12.log
