for a in "256 20" "2022 14" "512 16" "128 22" "1024 13" "600 21 3" "300 18" "2000 15" "1000 17" "300 19"; do
  for c in 4 5; do
    POWDR_NTT_C=$c timeout 300 python tools/bench_ntt.py $a 2>/dev/null | sed "s/^/c=$c /" | cut -c1-110
  done
done
