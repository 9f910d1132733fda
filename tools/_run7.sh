mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_segment_proof.py -m gpu -q -x 2>&1 | tail -6 ) > gpurun_out/r02_pytest7.log
( timeout 600 python bench.py --shape C4 --steps 2 --warmup 1 ) > gpurun_out/r02_bench_c4.json 2> gpurun_out/r02_bench_c4_err.log
( timeout 600 python bench.py --shape C5 --steps 2 --warmup 1 ) > gpurun_out/r02_bench_c5.json 2> gpurun_out/r02_bench_c5_err.log
( timeout 600 python tools/bench_segment.py 57 3 0 ) > gpurun_out/r02_segment_bench_reth_shaped.txt 2>&1
tail -3 gpurun_out/r02_pytest7.log; tail -6 gpurun_out/r02_segment_bench_reth_shaped.txt
python - <<'P'
import json
for f in ("c4","c5"):
    d=json.load(open(f"gpurun_out/r02_bench_{f}.json")); ms=d["multi_segment"]
    print(f, d["value"]/1e9, d["ms_per_step"], {k:round(v,1) for k,v in ms["stage_ms_rank0"].items() if v>15})
P
