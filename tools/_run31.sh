R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-logup-leg --no-segment-leg --no-callmajor-leg --no-copy-ceiling --no-live-pmc > /tmp/kt.json 2>/tmp/kt.log
python $R/tools/gap_analysis.py /tmp/kt --max-gap-us 3000 | head -24
