mkdir -p gpurun_out
LEGS="--no-cpu-baseline --no-logup-leg --no-segment-leg --no-callmajor-leg --no-copy-ceiling"
( timeout 300 python -m pytest tests/test_abi_load.py tests/test_prover_gpu.py -m gpu -q -x -k "selftest or golden or proof_bytes_match or c2_shape" 2>&1 | tail -4 ) > gpurun_out/r02_pytest9.log
for w in 8 6; do
  POWDR_HASH_WAVES=$w timeout 300 python bench.py --steps 5 --warmup 2 $LEGS > gpurun_out/r02_bench_hash_w$w.json 2>/dev/null
  python - <<P
import json
d=json.load(open('gpurun_out/r02_bench_hash_w$w.json')); print('waves $w', d['ms_per_step'], d['stage_ms']['leaf_hash_kernel'], d['stage_ms']['compress_kernel'], d['stage_ms']['compress_tail_kernel'])
P
done
POWDR_HASH_WAVES=6 timeout 300 python bench.py --shape C4 --steps 2 --warmup 1 > gpurun_out/r02_bench_c4_w6.json 2>/dev/null
POWDR_HASH_WAVES=8 timeout 300 python bench.py --shape C4 --steps 2 --warmup 1 > gpurun_out/r02_bench_c4_w8.json 2>/dev/null
python - <<P
import json
for w in (6,8):
    d=json.load(open(f'gpurun_out/r02_bench_c4_w{w}.json')); print('C4 waves',w, d['value']/1e9, d['multi_segment']['stage_ms_rank0']['leaf_hash_kernel'])
P
tail -3 gpurun_out/r02_pytest9.log
