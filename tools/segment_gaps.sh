# rocprofv3 --kernel-trace of bench.py --shape C5 + tools/gap_analysis.py -> gpurun_out/r02_segment_gaps.txt
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --shape C5 --steps 2 --warmup 1 > /tmp/kt.json 2>/tmp/kt.log
python - <<P
import json
d=json.load(open('/tmp/kt.json')); m=d['multi_segment']; print('C5 under rocprof', d['value']/1e9, d['ms_per_step'], sum(m['stage_ms_rank0'].values()))
P
python $R/tools/gap_analysis.py /tmp/kt --max-gap-us 3000 > $R/gpurun_out/r02_segment_gaps.txt 2>&1
cat $R/gpurun_out/r02_segment_gaps.txt
