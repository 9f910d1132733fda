mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > gpurun_out/r02_pytest_final.log
( timeout 900 python bench.py --steps 10 --warmup 3 ) > gpurun_out/r02_bench_c2.json 2> gpurun_out/r02_bench_c2.err
cd /tmp && export TMPDIR=/tmp
LEGS="--no-cpu-baseline --no-logup-leg --no-segment-leg --no-callmajor-leg --no-copy-ceiling"
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_prof_stats -- python $R/bench.py --steps 10 --warmup 3 $LEGS ) > $R/gpurun_out/r02_bench_c2_under_rocprofv3.json 2> $R/gpurun_out/r02_prof_stats.err
( timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r02_pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 $LEGS ) > /dev/null 2> $R/gpurun_out/r02_pmc_fetch.err
( timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r02_pmc_write -- python $R/bench.py --steps 2 --warmup 1 $LEGS ) > /dev/null 2> $R/gpurun_out/r02_pmc_write.err
cd $R
python tools/pmc_traffic_json.py gpurun_out/r02_pmc_fetch gpurun_out/r02_pmc_write 2 > gpurun_out/r02_pmc_traffic_c2.json 2> gpurun_out/r02_pmc_traffic.err
find gpurun_out/r02_prof_stats -name "*stats*.csv" | head -3
for f in $(find gpurun_out/r02_prof_stats -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/r02_kernel_stats_c2.csv; done
rm -rf gpurun_out/r02_pmc_fetch gpurun_out/r02_pmc_write gpurun_out/r02_pmc_valu
find gpurun_out/r02_prof_stats -name "*kernel_trace.csv" -delete
( timeout 900 python tools/run_c1_oracle.py ) > gpurun_out/r02_c1_oracle.json 2> gpurun_out/r02_c1_oracle.err
( timeout 900 python tools/run_c3_scale.py ) > gpurun_out/r02_c3_scale_report.txt 2>&1
( timeout 600 python tools/bench_keccak_fixture.py ) > gpurun_out/r02_keccak_preopt_fixture.txt 2>&1
tail -3 gpurun_out/r02_pytest_final.log; head -c 300 gpurun_out/r02_bench_c2.json; echo; cat gpurun_out/r02_c1_oracle.json; tail -5 gpurun_out/r02_c3_scale_report.txt; head -12 gpurun_out/r02_kernel_stats_c2.csv; head -c 600 gpurun_out/r02_pmc_traffic_c2.json
