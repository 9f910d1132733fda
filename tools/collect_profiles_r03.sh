# Round 3's profile set in one GPU visit (gpurun -- 'bash tools/collect_profiles_r03.sh'): the bench line (LogUp headline), rocprofv3
# kernel stats of the headline leg, FETCH_SIZE / WRITE_SIZE PMC passes, PMC issue counters of the specialised (JIT) kernels and their
# interpreter twins at 2^18 rows, C4 / C5 (+ in-process multi-device form), the keccak fixture with and without specialised kernels
# -> gpurun_out/r03_*; copy what is to be kept into profiles/.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
( timeout 1200 python bench.py --steps 10 --warmup 3 ) > gpurun_out/r03_bench_c2.json 2> gpurun_out/r03_bench_c2.err
cd /tmp && export TMPDIR=/tmp
LEGS="--no-cpu-baseline --no-logup-leg --no-segment-leg --no-callmajor-leg --no-copy-ceiling --no-live-pmc --no-c3-leg"
rm -rf $R/gpurun_out/r03_prof_stats $R/gpurun_out/r03_pmc_fetch $R/gpurun_out/r03_pmc_write
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_prof_stats -- python $R/bench.py --steps 10 --warmup 3 $LEGS ) > $R/gpurun_out/r03_bench_c2_under_rocprofv3.json 2> $R/gpurun_out/r03_prof_stats.err
( timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r03_pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 $LEGS ) > /dev/null 2> $R/gpurun_out/r03_pmc_fetch.err
( timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r03_pmc_write -- python $R/bench.py --steps 2 --warmup 1 $LEGS ) > /dev/null 2> $R/gpurun_out/r03_pmc_write.err
cd $R
python tools/pmc_traffic_json.py gpurun_out/r03_pmc_fetch gpurun_out/r03_pmc_write 3 > gpurun_out/r03_pmc_traffic_c2.json 2> gpurun_out/r03_pmc_traffic.err
for f in $(find gpurun_out/r03_prof_stats -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/r03_kernel_stats_c2.csv; done
find gpurun_out/r03_prof_stats -name "*kernel_trace.csv" -delete
rm -rf gpurun_out/r03_pmc_fetch gpurun_out/r03_pmc_write
# PMC issue counters of the expression kernels: specialised vs interpreter, 2^18 rows
cd /tmp
ARGS="--log-height 18 --steps 1 --warmup 1 --no-cpu-baseline --no-logup-leg --no-segment-leg --no-callmajor-leg --no-copy-ceiling --no-live-pmc --no-c3-leg"
OUT=$R/gpurun_out/r03_pmc_expression_kernels.txt
: > $OUT
for J in 1 0; do
  rm -rf /tmp/pmc_a
  POWDR_JIT=$J timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVES --output-format csv -d /tmp/pmc_a -- python $R/bench.py $ARGS > /dev/null 2>&1
  echo "== POWDR_JIT=$J (2022 cols x 2^18 rows, 1 warm-up + 1 timed step)" >> $OUT
  for k in quotient logup_perm deep_logup ext_dot_partial; do python $R/tools/pmc_csv_summary.py /tmp/pmc_a $k >> $OUT; done
done
cd $R
timeout 900 python bench.py --shape C4 --steps 2 --warmup 1 > gpurun_out/r03_bench_c4.json 2>/dev/null
timeout 900 python bench.py --shape C5 --steps 2 --warmup 1 > gpurun_out/r03_bench_c5.json 2>/dev/null
timeout 900 python bench.py --shape C4 --steps 2 --warmup 1 --gpus 1 --inproc > gpurun_out/r03_bench_c4_inproc.json 2>/dev/null
( POWDR_JIT=0 timeout 600 python tools/bench_keccak_fixture.py 2>&1 ) | grep -v amdgpu.ids > gpurun_out/r03_keccak_preopt_fixture.txt
( POWDR_JIT=1 timeout 900 python tools/bench_keccak_fixture.py 2>&1 ) | grep -v amdgpu.ids > gpurun_out/r03_keccak_preopt_fixture_jit.txt
python - <<P
import json
load=lambda p: json.loads([l for l in open(p) if l.startswith('{')][-1])
d=load('gpurun_out/r03_bench_c2.json')
print("headline", d['value']/1e9, d['ms_per_step'], "constraints-only", d['constraints_only']['ms_per_step'], d['constraints_only']['value']/1e9, "multi", d['multi_segment']['value']/1e9, "c3", (d['c3'] or {}).get('value'))
r=d['roofline']; print(r['frac'], r['traffic'], (r.get('valu') or {}).get('frac'), r['whole_step']['frac'])
print(sorted(d['stage_ms'].items(), key=lambda kv:-kv[1])[:12])
print("records", d['tracegen_from_records'].get('fused_ms'), "cpu", d['cpu_baseline']['value'], (d['cpu_baseline'].get('tuned') or {}))
u=load('gpurun_out/r03_bench_c2_under_rocprofv3.json'); print('under rocprof', u['ms_per_step'])
for k in ('c4','c5','c4_inproc'):
    try:
        x=load(f'gpurun_out/r03_bench_{k}.json'); print(k, x['value']/1e9, x['ms_per_step'], x['multi_segment'].get('commitment_merge'))
    except Exception as e: print(k, 'ERR', e)
P
head -12 gpurun_out/r03_kernel_stats_c2.csv | cut -c1-160; tail -3 gpurun_out/r03_keccak_preopt_fixture.txt | cut -c1-400; tail -3 gpurun_out/r03_keccak_preopt_fixture_jit.txt | cut -c1-400
cat gpurun_out/r03_pmc_expression_kernels.txt | cut -c1-300
