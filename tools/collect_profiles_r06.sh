# Round 6's profile set in one GPU visit (gpurun -- 'bash tools/collect_profiles_r06.sh'): the driver's own bench command (compact line +
# full record), rocprofv3 kernel stats of the headline leg and of configs[2] (trace handed over), C4 / C5 as the main workload with the
# segments' OWN shapes, the 8-rank plain command on one GPU. (The PMC passes of configs[2] — tools/pmc_c3_logup_r06.sh — the budget sweep —
# tools/c4_budget_sweep.sh — and the SELECT repro — tools/repro_select_atomics.sh — are separate visits.)
# -> gpurun_out/r06_*; copy what is to be kept into profiles/.
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$PWD}
SECONDS=0
( timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --full-out gpurun_out/r06_bench_full.json ) > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench.err
echo "driver's bench command, wall seconds: $SECONDS; line bytes: $(wc -c < gpurun_out/r06_bench_line.json); stderr bytes: $(wc -c < gpurun_out/r06_bench.err)" > gpurun_out/r06_bench_wall.txt
cd /tmp && export TMPDIR=/tmp
LEGS="--no-cpu-baseline --no-logup-leg --no-segment-leg --no-callmajor-leg --no-copy-ceiling --no-live-pmc --no-c3-leg"
rm -rf $R/gpurun_out/r06_prof_stats $R/gpurun_out/r06_prof_c3
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_prof_stats -- python $R/bench.py --steps 10 --warmup 3 $LEGS --full-out $R/gpurun_out/r06_bench_c2_under_rocprofv3_full.json ) > $R/gpurun_out/r06_bench_c2_under_rocprofv3.json 2> $R/gpurun_out/r06_prof_stats.err
( timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_prof_c3 -- python $R/tools/run_c3_logup.py 2 --no-constraints-only ) > $R/gpurun_out/r06_c3_logup_under_rocprofv3.txt 2> $R/gpurun_out/r06_prof_c3.err
cp $R/gpurun_out/c3_logup.json $R/gpurun_out/r06_c3_logup_profiled.json
cd $R
for f in $(find gpurun_out/r06_prof_stats -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/r06_kernel_stats_c2.csv; done
for f in $(find gpurun_out/r06_prof_c3 -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/r06_kernel_stats_c3_logup.csv; done
find gpurun_out/r06_prof_stats gpurun_out/r06_prof_c3 -name "*kernel_trace.csv" -delete
timeout 900 python bench.py --shape C4 --steps 2 --warmup 1 --full-out gpurun_out/r06_bench_c4_full.json > gpurun_out/r06_bench_c4.json 2>/dev/null
timeout 900 python bench.py --shape C5 --steps 2 --warmup 1 --full-out gpurun_out/r06_bench_c5_full.json > gpurun_out/r06_bench_c5.json 2>/dev/null
POWDR_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --log-height 12 --segment-log-height 10 --steps 2 --warmup 1 --no-cpu-baseline --no-c3-leg --full-out gpurun_out/r06_bench_8ranks_one_gpu_full.json > gpurun_out/r06_bench_8ranks_one_gpu.json 2>/dev/null
python - <<P
import json
load=lambda p: json.loads([l for l in open(p) if l.startswith('{')][-1])
print(open('gpurun_out/r06_bench_wall.txt').read().strip())
d=load('gpurun_out/r06_bench_line.json')
print(json.dumps({k: d[k] for k in ('value','ms_per_step','roofline','cpu_baseline','c3','multi_segment','constraints_only','tracegen_from_records') if k in d})[:2400])
u=load('gpurun_out/r06_bench_c2_under_rocprofv3.json'); print('under rocprof', u['ms_per_step'])
for k in ('c4','c5','8ranks_one_gpu'):
    try:
        x=load(f'gpurun_out/r06_bench_{k}.json'); print(k, x['value']/1e9, x['ms_per_step'], x.get('n_gpus'), x.get('multi_segment'))
    except Exception as e: print(k, 'ERR', e)
P
head -6 gpurun_out/r06_kernel_stats_c2.csv | cut -c1-170; head -8 gpurun_out/r06_kernel_stats_c3_logup.csv | cut -c1-170
