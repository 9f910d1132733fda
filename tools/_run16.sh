mkdir -p gpurun_out
( timeout 600 python __graft_entry__.py --smoke 2>&1 | tail -2 ) > gpurun_out/r02_smoke.log
timeout 600 python bench.py --shape C4 --steps 2 --warmup 1 > gpurun_out/r02_bench_c4.json 2>/dev/null
timeout 600 python bench.py --shape C5 --steps 2 --warmup 1 > gpurun_out/r02_bench_c5.json 2>/dev/null
( timeout 600 python tools/bench_segment.py 2>&1 ) > gpurun_out/r02_segment_bench_reth_shaped.txt
( timeout 600 python tools/bench_keccak_fixture.py ) > gpurun_out/r02_keccak_preopt_fixture.txt 2>&1
python - <<P
import json
for k in ('c4','c5'):
    d=json.load(open(f'gpurun_out/r02_bench_{k}.json')); m=d['multi_segment']; print(k, d['value']/1e9, d['ms_per_step'], m['airs_per_segment'], m['proof_bytes_per_segment'], sum(m['stage_ms_rank0'].values()))
P
cat gpurun_out/r02_smoke.log; tail -6 gpurun_out/r02_segment_bench_reth_shaped.txt; tail -5 gpurun_out/r02_keccak_preopt_fixture.txt
