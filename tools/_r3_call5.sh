python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r3_call5_pytest.txt
( time python bench.py ) > gpurun_out/r3_bench_default2.json 2> gpurun_out/r3_bench_default2.err
tail -8 gpurun_out/r3_call5_pytest.txt; tail -4 gpurun_out/r3_bench_default2.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_bench_default2.json").read().strip().splitlines()[-1])
print("headline", d["value"]/1e9, d["ms_per_step"])
co=d["constraints_only"]; print("constraints_only", co.get("value",0)/1e9 if co else None, co.get("ms_per_step") if co else None)
print("records", d["tracegen_from_records"])
ms=d["multi_segment"]; print("multi", {k:ms.get(k) for k in ("value","ms_per_step","error","logup")})
c3=d["c3"]; print("c3", {k:c3.get(k) for k in ("value","trace_gen_ms","prove_ms","error","skipped","specialised_kernels","verify_rc")} if c3 else None)
if c3 and "kernels" in c3: print({k:round(v["ms"],1) for k,v in c3["kernels"].items() if v["ms"]>5})
print("build", d["build"])
PY
