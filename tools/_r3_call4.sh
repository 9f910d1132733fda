python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r3_call4_pytest.txt
( time python bench.py ) > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err
tail -5 gpurun_out/r3_call4_pytest.txt; tail -5 gpurun_out/r3_bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_bench_default.json").read().strip().splitlines()[-1])
print("headline", d["value"]/1e9, d["ms_per_step"], {k:round(v,2) for k,v in d["stage_ms"].items() if v>0.8})
co=d["constraints_only"]; print("constraints_only", co.get("value",0)/1e9 if co else None, co.get("ms_per_step") if co else None, {k:round(v,2) for k,v in (co or {}).get("stage_ms",{}).items() if v>0.8})
ms=d["multi_segment"]; print("multi", {k:ms.get(k) for k in ("value","ms_per_step","error","logup")})
c3=d["c3"]; print("c3", {k:c3.get(k) for k in ("value","trace_gen_ms","prove_ms","error","skipped","specialised_kernels")} if c3 else None)
if c3 and "kernels" in c3: print({k:round(v["ms"],1) for k,v in c3["kernels"].items() if v["ms"]>5})
print("roof", d["roofline"]["frac"], d["roofline"]["kernel"], d["roofline"].get("traffic"), d["roofline"].get("valu"))
print("cpu", d["cpu_baseline"])
PY
