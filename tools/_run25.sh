mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_prover_gpu.py tests/test_segment_proof.py -m gpu -q -x 2>&1 | tail -4 ) > gpurun_out/r02_pytest25.log
for a in "128 22" "1024 13" "2022 14" "256 20"; do
  timeout 300 python tools/bench_ntt.py $a 2>/dev/null
  POWDR_NTT_NO_TABLES=1 timeout 300 python tools/bench_ntt.py $a 2>/dev/null | tail -1 | sed 's/^/NO TABLES: /'
done
tail -3 gpurun_out/r02_pytest25.log
