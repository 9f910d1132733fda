# Round 5's profile set in one GPU visit (gpurun -- 'bash tools/collect_profiles_r05.sh'): the driver's own bench command (compact
# line + full record), rocprofv3 kernel stats of the headline leg and of configs[2] proven with the trace handed over, FETCH_SIZE /
# WRITE_SIZE PMC passes of that proof, C4 / C5 as the main workload (distinct segments), the 8-rank plain command on one GPU.
# -> gpurun_out/r05_*; copy what is to be kept into profiles/.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
T0=$(date +%s)
( timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --full-out gpurun_out/r05_bench_full.json ) > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench.err
echo "driver's bench command, wall seconds: $(( $(date +%s) - T0 )); line bytes: $(wc -c < gpurun_out/r05_bench_line.json); stderr bytes: $(wc -c < gpurun_out/r05_bench.err)" > gpurun_out/r05_bench_wall.txt
cd /tmp && export TMPDIR=/tmp
LEGS="--no-cpu-baseline --no-logup-leg --no-segment-leg --no-callmajor-leg --no-copy-ceiling --no-live-pmc --no-c3-leg"
rm -rf $R/gpurun_out/r05_prof_stats $R/gpurun_out/r05_prof_c3 /tmp/c3_fetch /tmp/c3_write
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_prof_stats -- python $R/bench.py --steps 10 --warmup 3 $LEGS --full-out $R/gpurun_out/r05_bench_c2_under_rocprofv3_full.json ) > $R/gpurun_out/r05_bench_c2_under_rocprofv3.json 2> $R/gpurun_out/r05_prof_stats.err
( timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_prof_c3 -- python $R/tools/run_c3_logup.py 2 --no-constraints-only --no-segment ) > $R/gpurun_out/r05_c3_logup_under_rocprofv3.txt 2> $R/gpurun_out/r05_prof_c3.err
cp $R/gpurun_out/c3_logup.json $R/gpurun_out/r05_c3_logup_profiled.json
# (PMC passes: POWDR_QUERY_SELECT=0 — on the round's last build the FETCH_SIZE pass did not come back within its 600 s with the query phase's
#  64-bit atomic sums under counter collection; the two-kernel query path profiles like round 4's. The kernel-trace passes above are unaffected.)
export POWDR_QUERY_SELECT=0
( timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/c3_fetch -- python $R/tools/run_c3_logup.py 1 --no-constraints-only --no-segment ) > /dev/null 2> $R/gpurun_out/r05_pmc_c3_fetch.err
( timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/c3_write -- python $R/tools/run_c3_logup.py 1 --no-constraints-only --no-segment ) > /dev/null 2> $R/gpurun_out/r05_pmc_c3_write.err
unset POWDR_QUERY_SELECT
cd $R
python tools/pmc_traffic_json.py /tmp/c3_fetch /tmp/c3_write 2 "C3 3731 cols x 2^22 rows with LogUp, trace handed over, streamed over 2 sub-cosets (bytes per PROOF: 1 warm-up + 1 timed proof under the counters; trace generation and the two restoring transforms run once each)" > gpurun_out/r05_pmc_traffic_c3_logup.json 2> gpurun_out/r05_pmc_traffic_c3.err
for f in $(find gpurun_out/r05_prof_stats -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/r05_kernel_stats_c2.csv; done
for f in $(find gpurun_out/r05_prof_c3 -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/r05_kernel_stats_c3_logup.csv; done
find gpurun_out/r05_prof_stats gpurun_out/r05_prof_c3 -name "*kernel_trace.csv" -delete
timeout 900 python bench.py --shape C4 --steps 2 --warmup 1 --full-out gpurun_out/r05_bench_c4_full.json > gpurun_out/r05_bench_c4.json 2>/dev/null
timeout 900 python bench.py --shape C5 --steps 2 --warmup 1 --full-out gpurun_out/r05_bench_c5_full.json > gpurun_out/r05_bench_c5.json 2>/dev/null
POWDR_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --log-height 12 --segment-log-height 10 --steps 2 --warmup 1 --no-cpu-baseline --no-c3-leg --full-out gpurun_out/r05_bench_8ranks_one_gpu_full.json > gpurun_out/r05_bench_8ranks_one_gpu.json 2>/dev/null
python - <<P
import json
load=lambda p: json.loads([l for l in open(p) if l.startswith('{')][-1])
print(open('gpurun_out/r05_bench_wall.txt').read().strip())
d=load('gpurun_out/r05_bench_line.json')
print(json.dumps({k: d[k] for k in ('value','ms_per_step','roofline','cpu_baseline','c3','multi_segment','constraints_only','tracegen_from_records') if k in d})[:1800])
u=load('gpurun_out/r05_bench_c2_under_rocprofv3.json'); print('under rocprof', u['ms_per_step'])
for k in ('c4','c5','8ranks_one_gpu'):
    try:
        x=load(f'gpurun_out/r05_bench_{k}.json'); print(k, x['value']/1e9, x['ms_per_step'], x.get('n_gpus'), x.get('multi_segment'))
    except Exception as e: print(k, 'ERR', e)
t=json.load(open('gpurun_out/r05_pmc_traffic_c3_logup.json'))
for k,v in sorted(t['kernels'].items(), key=lambda kv:-(kv[1]['fetch_bytes_corrected']+kv[1]['write_bytes']))[:8]:
    print(k[:70], v['dispatches'], round(v['fetch_bytes_corrected']/1e9,1), round(v['write_bytes']/1e9,1))
P
head -6 gpurun_out/r05_kernel_stats_c2.csv | cut -c1-170; head -8 gpurun_out/r05_kernel_stats_c3_logup.csv | cut -c1-170
