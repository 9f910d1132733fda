mkdir -p gpurun_out
( timeout 1500 python tools/full_size_parity.py 20 C2 16 8 0 ) > gpurun_out/r02_full_size_parity_c2.json 2> gpurun_out/r02_full_size_parity_c2.err
cat gpurun_out/r02_full_size_parity_c2.json; tail -3 gpurun_out/r02_full_size_parity_c2.err; free -g | head -2
