# Where the GPU idles inside the C2 headline step: rocprofv3 --kernel-trace of the headline leg + tools/gap_analysis.py
# -> gpurun_out/r04_gaps_c2.txt
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
LEGS="--no-cpu-baseline --no-logup-leg --no-segment-leg --no-callmajor-leg --no-copy-ceiling --no-live-pmc --no-c3-leg"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --steps 10 --warmup 2 $LEGS > /tmp/kt.json 2>/tmp/kt.log
( echo "# C2 headline leg (LogUp), 2 warm-up + 10 timed steps under rocprofv3 --kernel-trace; gaps above 3 ms (between steps: trace generation is host-driven) ignored";
  python -c "
import json
d=json.loads([l for l in open('/tmp/kt.json') if l.startswith('{')][-1]); print('# ms_per_step under the profiler', d['ms_per_step'], 'sum of kernel times per step', sum(d['stage_ms'].values()))";
  python $R/tools/gap_analysis.py /tmp/kt --max-gap-us 3000 ) > $R/gpurun_out/r04_gaps_c2.txt 2>&1
cat $R/gpurun_out/r04_gaps_c2.txt | head -30
