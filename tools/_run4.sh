mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_prover_gpu.py tests/test_segment_proof.py tests/test_bench_contract_gpu.py -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r02_pytest4.log
for cfg in "256 20" "2022 14" "64 12" "512 16" "128 22" "1024 10"; do ( timeout 120 python tools/bench_ntt.py $cfg 5 ) >> gpurun_out/r02_bench_ntt.txt 2>&1; done
( timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-logup-leg ) > gpurun_out/r02_bench_c2_c.json 2> gpurun_out/r02_bench_c_err.log
tail -4 gpurun_out/r02_pytest4.log; grep -v amdgpu.ids gpurun_out/r02_bench_ntt.txt; head -c 400 gpurun_out/r02_bench_c2_c.json
