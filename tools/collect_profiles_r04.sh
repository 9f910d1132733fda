# Round 4's profile set in one GPU visit (gpurun -- 'bash tools/collect_profiles_r04.sh'): the bench line (LogUp headline, honest C4
# leg, configs[2] with its bus argument streamed, timed step from records), rocprofv3 kernel stats of the headline leg and of the C3
# streamed proof, FETCH_SIZE / WRITE_SIZE PMC passes of the headline leg, C4 / C5 as the main workload, the 2-rank plain command.
# -> gpurun_out/r04_*; copy what is to be kept into profiles/.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
T0=$(date +%s)
( timeout 1500 python bench.py --steps 10 --warmup 3 ) > gpurun_out/r04_bench_c2.json 2> gpurun_out/r04_bench_c2.err
echo "default bench wall seconds: $(( $(date +%s) - T0 ))" > gpurun_out/r04_bench_wall.txt
cd /tmp && export TMPDIR=/tmp
LEGS="--no-cpu-baseline --no-logup-leg --no-segment-leg --no-callmajor-leg --no-copy-ceiling --no-live-pmc --no-c3-leg"
rm -rf $R/gpurun_out/r04_prof_stats $R/gpurun_out/r04_pmc_fetch $R/gpurun_out/r04_pmc_write $R/gpurun_out/r04_prof_c3
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_prof_stats -- python $R/bench.py --steps 10 --warmup 3 $LEGS ) > $R/gpurun_out/r04_bench_c2_under_rocprofv3.json 2> $R/gpurun_out/r04_prof_stats.err
( timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r04_pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 $LEGS ) > /dev/null 2> $R/gpurun_out/r04_pmc_fetch.err
( timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r04_pmc_write -- python $R/bench.py --steps 2 --warmup 1 $LEGS ) > /dev/null 2> $R/gpurun_out/r04_pmc_write.err
( timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_prof_c3 -- python $R/tools/run_c3_logup.py 2 --no-constraints-only ) > $R/gpurun_out/r04_c3_logup_under_rocprofv3.txt 2> $R/gpurun_out/r04_prof_c3.err
cd $R
python tools/pmc_traffic_json.py gpurun_out/r04_pmc_fetch gpurun_out/r04_pmc_write 3 > gpurun_out/r04_pmc_traffic_c2.json 2> gpurun_out/r04_pmc_traffic.err
for f in $(find gpurun_out/r04_prof_stats -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/r04_kernel_stats_c2.csv; done
for f in $(find gpurun_out/r04_prof_c3 -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/r04_kernel_stats_c3_logup.csv; done
cp gpurun_out/c3_logup.json gpurun_out/r04_c3_logup.json
find gpurun_out/r04_prof_stats gpurun_out/r04_prof_c3 -name "*kernel_trace.csv" -delete
rm -rf gpurun_out/r04_pmc_fetch gpurun_out/r04_pmc_write
timeout 900 python bench.py --shape C4 --steps 2 --warmup 1 > gpurun_out/r04_bench_c4.json 2>/dev/null
timeout 900 python bench.py --shape C5 --steps 2 --warmup 1 > gpurun_out/r04_bench_c5.json 2>/dev/null
POWDR_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --log-height 12 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_2ranks_one_gpu.json 2>/dev/null
python - <<P
import json
load=lambda p: json.loads([l for l in open(p) if l.startswith('{')][-1])
d=load('gpurun_out/r04_bench_c2.json')
print(open('gpurun_out/r04_bench_wall.txt').read().strip())
print("headline", d['value']/1e9, d['ms_per_step'], "constraints-only", d['constraints_only']['ms_per_step'], d['constraints_only']['value']/1e9)
m=d['multi_segment']; print("multi", m['value']/1e9, m.get('verify_rc'), m.get('constraint_violations'), (m.get('lookup_balance') or {}).get('verify_rc'), m.get('trace_gen_ms_per_segment'), m.get('prove_ms_per_segment'))
c=d['c3'] or {}; print("c3", c.get('value'), c.get('prove_ms'), c.get('verify_rc'), c.get('stream_log_blocks'), c.get('prover_plus_trace_bytes'), (c.get('constraints_only') or {}).get('prove_ms'))
r=d['roofline']; print(r['frac'], r['traffic'], (r.get('valu') or {}).get('frac'), r['whole_step']['frac'])
print(sorted(d['stage_ms'].items(), key=lambda kv:-kv[1])[:12])
t=d['tracegen_from_records']; print("records", t.get('fused_ms'), (t.get('timed_step') or {}).get('ms_per_step'), "cpu", d['cpu_baseline']['value'])
u=load('gpurun_out/r04_bench_c2_under_rocprofv3.json'); print('under rocprof', u['ms_per_step'])
for k in ('c4','c5','2ranks_one_gpu'):
    try:
        x=load(f'gpurun_out/r04_bench_{k}.json'); print(k, x['value']/1e9, x['ms_per_step'], x.get('n_gpus'), (x.get('multi_segment') or {}).get('verify_rc'))
    except Exception as e: print(k, 'ERR', e)
P
head -8 gpurun_out/r04_kernel_stats_c2.csv | cut -c1-160; head -8 gpurun_out/r04_kernel_stats_c3_logup.csv | cut -c1-160
