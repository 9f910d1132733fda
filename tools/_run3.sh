mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_segment_proof.py tests/test_bench_contract_gpu.py -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r02_pytest3.log
( timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > gpurun_out/r02_bench_c2_b.json 2> gpurun_out/r02_bench_b_err.log
( timeout 600 python bench.py --shape C4 --steps 2 --warmup 1 ) > gpurun_out/r02_bench_c4.json 2> gpurun_out/r02_bench_c4_err.log
( timeout 600 python bench.py --shape C5 --steps 2 --warmup 1 ) > gpurun_out/r02_bench_c5.json 2> gpurun_out/r02_bench_c5_err.log
( timeout 600 python tools/bench_segment.py 57 3 0 ) > gpurun_out/r02_segment_bench_reth_shaped.txt 2>&1
tail -4 gpurun_out/r02_pytest3.log; tail -12 gpurun_out/r02_segment_bench_reth_shaped.txt; head -c 300 gpurun_out/r02_bench_c4.json; tail -2 gpurun_out/r02_bench_c4_err.log
