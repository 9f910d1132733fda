mkdir -p gpurun_out
: > gpurun_out/r02_bench_ntt.txt
for a in "256 20" "2022 14" "64 12" "512 16" "128 22" "1024 10" "1024 13" "2022 20 3" "600 21 3"; do
  timeout 300 python tools/bench_ntt.py $a 2>/dev/null >> gpurun_out/r02_bench_ntt.txt
done
cat gpurun_out/r02_bench_ntt.txt
