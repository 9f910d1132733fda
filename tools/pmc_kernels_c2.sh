# PMC issue counters of every kernel of one C2 step -> gpurun_out/r02_pmc_kernels_c2.txt
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_n
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_WAVES --output-format csv -d /tmp/pmc_n -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-logup-leg --no-segment-leg --no-callmajor-leg --no-copy-ceiling --no-live-pmc > /dev/null 2>&1
OUT=$R/gpurun_out/r02_pmc_kernels_c2.txt
: > $OUT
for k in lde_fused_kernel ntt_group_kernel leaf_hash_kernel apc_gather_tile_kernel apc_apply_bus bus_histogram deep_kernel ext_dot_partial quotient_kernel compress; do python $R/tools/pmc_csv_summary.py /tmp/pmc_n $k >> $OUT; done
cat $OUT
