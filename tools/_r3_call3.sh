set -x
python -m pytest tests/test_jit.py tests/test_segment_proof.py::test_worker_threads_return_their_device_memory -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3_call3_pytest.txt
B="python bench.py --logup --steps 3 --warmup 1 --no-cpu-baseline --no-callmajor-leg --no-segment-leg --no-live-pmc --no-copy-ceiling"
( time $B ) > gpurun_out/r3_c3_logup_jit.json 2> gpurun_out/r3_c3_logup_jit.err
( time POWDR_JIT=0 $B ) > gpurun_out/r3_c3_logup_nojit.json 2> gpurun_out/r3_c3_logup_nojit.err
( time POWDR_JIT_CHUNK_COST=1700 $B ) > gpurun_out/r3_c3_logup_jit_c1700.json 2> gpurun_out/r3_c3_logup_jit_c1700.err
( time POWDR_JIT_CHUNK_COST=8000 $B ) > gpurun_out/r3_c3_logup_jit_c8000.json 2> gpurun_out/r3_c3_logup_jit_c8000.err
tail -3 gpurun_out/r3_call3_pytest.txt
for f in jit nojit jit_c1700 jit_c8000; do python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3_c3_logup_$f.json").read().strip().splitlines()[-1])
    s=d["stage_ms"]; print("$f", round(d["ms_per_step"],1), {k:round(v,2) for k,v in s.items() if v>1.0})
except Exception as e: print("$f", "ERR", e)
PY
tail -4 gpurun_out/r3_c3_logup_$f.err
done
