mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
( timeout 900 python bench.py --steps 10 --warmup 3 ) > gpurun_out/r02_bench_c2.json 2> gpurun_out/r02_bench_c2.err
( timeout 120 tools/microbench_opcodes ) > gpurun_out/r02_microbench_opcodes.txt 2>&1
cd /tmp && export TMPDIR=/tmp
LEGS="--no-cpu-baseline --no-logup-leg --no-segment-leg --no-callmajor-leg --no-copy-ceiling"
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_prof_stats -- python $R/bench.py --steps 10 --warmup 3 $LEGS ) > $R/gpurun_out/r02_bench_c2_under_rocprofv3.json 2> $R/gpurun_out/r02_prof_stats.err
( timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r02_pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 $LEGS ) > /dev/null 2> $R/gpurun_out/r02_pmc_fetch.err
( timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r02_pmc_write -- python $R/bench.py --steps 2 --warmup 1 $LEGS ) > /dev/null 2> $R/gpurun_out/r02_pmc_write.err
( timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/r02_pmc_valu -- python $R/bench.py --log-height 18 --steps 1 --warmup 1 --no-cpu-baseline --no-logup-leg --no-copy-ceiling --no-segment-leg --no-callmajor-leg ) > $R/gpurun_out/r02_pmc_valu.log 2>&1
cd $R
python tools/pmc_traffic_json.py gpurun_out/r02_pmc_fetch gpurun_out/r02_pmc_write 3 > gpurun_out/r02_pmc_traffic_c2.json 2> gpurun_out/r02_pmc_traffic.err
python tools/pmc_valu_json.py gpurun_out/r02_pmc_valu gpurun_out/r02_microbench_opcodes.txt 18 2022 2 > gpurun_out/r02_valu_model.json 2> gpurun_out/r02_valu_model.err
for f in $(find gpurun_out/r02_prof_stats -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/r02_kernel_stats_c2.csv; done
find gpurun_out/r02_prof_stats -name "*kernel_trace.csv" -delete
find gpurun_out/r02_pmc_fetch gpurun_out/r02_pmc_write gpurun_out/r02_pmc_valu -name "*.csv" -size +20M -delete
head -c 400 gpurun_out/r02_bench_c2.json; echo; head -8 gpurun_out/r02_kernel_stats_c2.csv; head -c 500 gpurun_out/r02_pmc_traffic_c2.json; echo; head -c 600 gpurun_out/r02_valu_model.json; cat gpurun_out/r02_valu_model.err | tail -3; grep -i "mad_i64\|signed" gpurun_out/r02_microbench_opcodes.txt | head
