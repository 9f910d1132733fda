mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r02_pytest1.log
( timeout 120 ./tools/microbench_opcodes ) > gpurun_out/r02_microbench_opcodes.txt 2>&1
( timeout 120 ./tools/microbench_mfma_mds ) > gpurun_out/r02_microbench_mfma_mds.txt 2>&1
( timeout 600 python bench.py --steps 5 --warmup 2 ) > gpurun_out/r02_bench_c2_a.json 2> gpurun_out/r02_bench_err.log
tail -3 gpurun_out/r02_pytest1.log; tail -5 gpurun_out/r02_microbench_mfma_mds.txt; head -c 600 gpurun_out/r02_bench_c2_a.json; tail -3 gpurun_out/r02_bench_err.log
