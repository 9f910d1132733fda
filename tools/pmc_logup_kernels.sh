# PMC counters (two passes) of the LogUp kernels at 2^18 rows -> gpurun_out/r02_pmc_logup_kernels.txt
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
ARGS="--log-height 18 --steps 1 --warmup 1 --no-cpu-baseline --no-segment-leg --no-callmajor-leg --no-copy-ceiling --logup-steps 1"
rm -rf /tmp/pmc_a /tmp/pmc_b
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d /tmp/pmc_a -- python $R/bench.py $ARGS > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_WAVES SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmc_b -- python $R/bench.py $ARGS > /dev/null 2>&1
OUT=$R/gpurun_out/r02_pmc_logup_kernels.txt
: > $OUT
for k in quotient_logup_kernel logup_perm_kernel deep_logup_kernel ext_dot_partial_kernel quotient_kernel apc_apply_bus; do
  python $R/tools/pmc_csv_summary.py /tmp/pmc_a $k >> $OUT
  python $R/tools/pmc_csv_summary.py /tmp/pmc_b $k >> $OUT
done
cat $OUT
