#!/bin/bash
# Is the rocprofv3 --pmc hang of round 5 (configs[2], FETCH_SIZE pass, query phase with 64-bit atomic sums) a profiler limit or a fault?
# The SELECT pass in isolation, every form, plain / --kernel-trace / --pmc, every run behind its own timeout (a hang costs 150 s, not the box).
cd "$(dirname "$0")/.."; ROOT=$PWD; export TMPDIR=/tmp; OUT=$ROOT/gpurun_out/r06_select; mkdir -p $OUT
COLS=${COLS:-1024}; LOGH=${LOGH:-22}
run() { # label, env select, profiler args...
  local label=$1 sel=$2; shift 2
  SECONDS=0
  ( cd /tmp && POWDR_QUERY_SELECT=$sel timeout -k 10 ${TMO:-150} "$@" python $ROOT/tools/repro_select_atomics.py --cols $COLS --log-h $LOGH ) > $OUT/$label.log 2>&1
  local rc=$?
  echo "$label: rc=$rc wall=${SECONDS} s | $(grep '^select=' $OUT/$label.log | tail -1)"
}
if [ "${STAGE:-small}" = small ]; then
run plain_sel1 1
run plain_sel2 2
run plain_sel0 0
run ktrace_sel2 2 rocprofv3 --kernel-trace --stats -d $OUT/ktrace_sel2 --
run pmc_fetch_sel2 2 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_sel2 --
run pmc_fetch_sel1 1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_sel1 --
run pmc_write_sel1 1 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_sel1 --
else  # STAGE=wide COLS=4632: the width of configs[2]'s permutation matrix, where round 5's FETCH_SIZE pass did not return
run wide_plain_sel2 2
run wide_plain_sel1 1
run wide_pmc_fetch_sel1 1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/wide_pmc_fetch_sel1 --
run wide_pmc_fetch_sel2 2 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/wide_pmc_fetch_sel2 --
fi
rocm-smi --showuse 2>/dev/null | head -8
