#!/bin/bash
# C4 multi-segment leg with the segments' own shapes under three device budgets (fractions of the capped segment's resident plan):
# per-segment times and which AIRs the policy streamed. Output: gpurun_out/r06_c4_budget_*.json
mkdir -p gpurun_out
for f in 0 0.95 0.8; do
  python bench.py --shape C4 --segments 8 --steps 2 --warmup 1 --no-cpu-baseline --segment-budget-frac $f --full-out gpurun_out/r06_c4_budget_$f.json > /dev/null 2> gpurun_out/r06_c4_budget_$f.err
  python - <<PY
import json
ms=json.load(open("gpurun_out/r06_c4_budget_$f.json"))["multi_segment"]
print("budget frac $f value %.3f G cells/s  ms/step %.1f  verify %s" % (ms["value"]/1e9, ms["ms_per_step"], ms["verify_rc"]))
for u,t in ms["ms_by_segment_rank0"].items():
    print("  seg", u, "cells %.2f G" % (ms["cells_by_segment"][int(u)]/1e9), "prove %.1f ms" % t["prove"], "gen %.1f" % t["trace_gen"], ms["streamed_airs_by_segment"].get(u))
PY
done
