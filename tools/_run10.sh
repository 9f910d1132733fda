mkdir -p gpurun_out
LEGS="--no-cpu-baseline --no-logup-leg --no-segment-leg --no-callmajor-leg --no-copy-ceiling"
( timeout 300 python -m pytest tests/test_abi_load.py tests/test_prover_gpu.py tests/test_segment_proof.py -m gpu -q -x -k "selftest or golden or proof_bytes_match or c2_shape or segment" 2>&1 | tail -4 ) > gpurun_out/r02_pytest10.log
for w in 8 6; do
  POWDR_HASH_WAVES=$w timeout 300 python bench.py --steps 5 --warmup 2 $LEGS > gpurun_out/r02_bench_hash_w$w.json 2>/dev/null
  python - <<P
import json
d=json.load(open('gpurun_out/r02_bench_hash_w$w.json')); print('waves $w', d['ms_per_step'], d['stage_ms']['leaf_hash_kernel'], d['stage_ms']['compress_kernel'], d['stage_ms']['compress_tail_kernel'])
P
done
tail -3 gpurun_out/r02_pytest10.log
