# The round's profile set in one GPU visit (gpurun -- 'bash tools/collect_profiles.sh'): bench line, rocprofv3 kernel stats of the
# headline leg, FETCH_SIZE / WRITE_SIZE PMC passes, C4, C5, reth-shaped segment, keccak fixture -> gpurun_out/r02_*; copy what is
# to be kept into profiles/.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
( timeout 900 python bench.py --steps 10 --warmup 3 ) > gpurun_out/r02_bench_c2.json 2> gpurun_out/r02_bench_c2.err
cd /tmp && export TMPDIR=/tmp
LEGS="--no-cpu-baseline --no-logup-leg --no-segment-leg --no-callmajor-leg --no-copy-ceiling"
rm -rf $R/gpurun_out/r02_prof_stats $R/gpurun_out/r02_pmc_fetch $R/gpurun_out/r02_pmc_write
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_prof_stats -- python $R/bench.py --steps 10 --warmup 3 $LEGS ) > $R/gpurun_out/r02_bench_c2_under_rocprofv3.json 2> $R/gpurun_out/r02_prof_stats.err
( timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r02_pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 $LEGS ) > /dev/null 2> $R/gpurun_out/r02_pmc_fetch.err
( timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r02_pmc_write -- python $R/bench.py --steps 2 --warmup 1 $LEGS ) > /dev/null 2> $R/gpurun_out/r02_pmc_write.err
cd $R
python tools/pmc_traffic_json.py gpurun_out/r02_pmc_fetch gpurun_out/r02_pmc_write 3 > gpurun_out/r02_pmc_traffic_c2.json 2> gpurun_out/r02_pmc_traffic.err
for f in $(find gpurun_out/r02_prof_stats -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/r02_kernel_stats_c2.csv; done
find gpurun_out/r02_prof_stats -name "*kernel_trace.csv" -delete
rm -rf gpurun_out/r02_pmc_fetch gpurun_out/r02_pmc_write
timeout 600 python bench.py --shape C4 --steps 2 --warmup 1 > gpurun_out/r02_bench_c4.json 2>/dev/null
timeout 600 python bench.py --shape C5 --steps 2 --warmup 1 > gpurun_out/r02_bench_c5.json 2>/dev/null
( timeout 600 python tools/bench_segment.py 2>&1 ) | grep -v amdgpu.ids > gpurun_out/r02_segment_bench_reth_shaped.txt
( timeout 600 python tools/bench_keccak_fixture.py 2>&1 ) | grep -v amdgpu.ids > gpurun_out/r02_keccak_preopt_fixture.txt
python - <<P
import json
d=json.load(open('gpurun_out/r02_bench_c2.json'))
print(d['value']/1e9, d['ms_per_step'], d['logup']['ms_per_step'], d['logup']['value']/1e9, d['multi_segment']['value']/1e9, d['tracegen_column_structured']['gather_ms'])
r=d['roofline']; print(r['frac'], r['traffic'], r['valu']['frac'], r['whole_step']['frac'])
print(sorted(d['stage_ms'].items(), key=lambda kv:-kv[1])[:8])
u=json.load(open('gpurun_out/r02_bench_c2_under_rocprofv3.json')); print('under rocprof', u['ms_per_step'])
for k in ('c4','c5'):
    x=json.load(open(f'gpurun_out/r02_bench_{k}.json')); print(k, x['value']/1e9, x['ms_per_step'])
P
head -4 gpurun_out/r02_kernel_stats_c2.csv | cut -c1-200; tail -3 gpurun_out/r02_segment_bench_reth_shaped.txt; tail -2 gpurun_out/r02_keccak_preopt_fixture.txt | cut -c1-250
