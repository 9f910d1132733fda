# Where the GPU idles inside the honest multi-AIR segment proofs (bench.py --shape C5 / C4): rocprofv3 --kernel-trace + tools/gap_analysis.py
# -> gpurun_out/r04b_gaps_<shape>.txt       usage: bash tools/gaps_segments.sh [C5 [C4]]
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for SHAPE in ${@:-C5}; do
  rm -rf /tmp/kt_$SHAPE
  timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$SHAPE -- python $R/bench.py --shape $SHAPE --steps 2 --warmup 1 --no-cpu-baseline > /tmp/kt_$SHAPE.json 2>/tmp/kt_$SHAPE.log
  ( echo "# bench.py --shape $SHAPE --steps 2 --warmup 1 under rocprofv3 --kernel-trace (8 honest segments per step, trace generation + one segment proof each); gaps above 3 ms ignored (host work between segments)";
    python -c "
import json
d=json.loads([l for l in open('/tmp/kt_$SHAPE.json') if l.startswith('{')][-1]); m=d.get('multi_segment') or d
print('# value', d['value'], 'ms_per_step', d['ms_per_step'], 'trace_gen_ms_per_segment', m.get('trace_gen_ms_per_segment'), 'prove_ms_per_segment', m.get('prove_ms_per_segment'))";
    python $R/tools/gap_analysis.py /tmp/kt_$SHAPE --max-gap-us 3000 ) > $R/gpurun_out/r04b_gaps_$SHAPE.txt 2>&1
  head -24 $R/gpurun_out/r04b_gaps_$SHAPE.txt
done
