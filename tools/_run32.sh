mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_prover_gpu.py tests/test_segment_proof.py -m gpu -q -x 2>&1 | tail -4 ) > gpurun_out/r02_pytest32.log
LEGS="--no-cpu-baseline --no-logup-leg --no-segment-leg --no-callmajor-leg --no-copy-ceiling --no-live-pmc"
timeout 300 python bench.py --steps 8 --warmup 2 $LEGS > /tmp/b.json 2>/dev/null
python - <<P
import json
d=json.load(open('/tmp/b.json')); s=d['stage_ms']; print(d['value']/1e9, d['ms_per_step'], sum(s.values()))
P
bash tools/_run31.sh 2>&1 | head -8
tail -3 gpurun_out/r02_pytest32.log
