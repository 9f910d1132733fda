mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_segment_proof.py tests/test_periphery_gpu.py tests/test_bench_contract_gpu.py -m gpu -q -x 2>&1 | tail -4 ) > gpurun_out/r02_pytest18.log
for k in 0 4 8; do
POWDR_SEGMENT_STREAMS=$k timeout 600 python bench.py --shape C5 --steps 2 --warmup 1 > gpurun_out/r02_bench_c5_k$k.json 2>/dev/null
python - <<P
import json
d=json.load(open('gpurun_out/r02_bench_c5_k$k.json')); m=d['multi_segment']; print('C5 streams $k', d['value']/1e9, d['ms_per_step'], sum(m['stage_ms_rank0'].values()))
P
done
POWDR_SEGMENT_STREAMS=4 timeout 600 python bench.py --shape C4 --steps 2 --warmup 1 > gpurun_out/r02_bench_c4.json 2>/dev/null
python - <<P
import json
d=json.load(open('gpurun_out/r02_bench_c4.json')); m=d['multi_segment']; print('C4', d['value']/1e9, d['ms_per_step'], sum(m['stage_ms_rank0'].values()))
P
( timeout 600 python tools/bench_segment.py 2>&1 ) | tail -5
tail -3 gpurun_out/r02_pytest18.log
