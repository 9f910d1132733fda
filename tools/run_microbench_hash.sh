# Times several builds of tools/microbench_hash.hip against each other (+ PMC instruction counts). Build them first, e.g.
#   B="hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ipowdr_amd/csrc -Iinclude tools/microbench_hash.hip"; mkdir -p tools/_hb
#   $B -o tools/_hb/scaled; $B -DHB_MINW=8 -o tools/_hb/scaled_w8; (older variants: check out the revision, build under another name)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_microbench_hash2.txt
: > $OUT
for v in unsigned signed scaled scaled_w8 signed scaled; do
  echo "== $v" >> $OUT
  timeout 120 $R/tools/_hb/$v 19 512 5 >> $OUT 2>&1
done
cd /tmp && export TMPDIR=/tmp
for v in signed scaled; do
  rm -rf /tmp/pmc_$v
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVE_CYCLES --output-format csv -d /tmp/pmc_$v -- $R/tools/_hb/$v 19 512 1 > /dev/null 2>&1
  echo "== pmc $v" >> $OUT
  python $R/tools/pmc_csv_summary.py /tmp/pmc_$v hash_kernel >> $OUT 2>&1
done
cat $OUT
