"""Per-kernel sums of PMC counters from a rocprofv3 rocpd database (run on the GPU box; the DB is
too large to ship back). usage: rocpd_pmc_summary.py results.db"""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
views = [r[0] for r in c.execute("select name from sqlite_master where type='view'")]
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
print("# counters_collection columns:", cols, file=sys.stderr)
name_col = "kernel_name" if "kernel_name" in cols else "name"
cnt_col = "counter_name" if "counter_name" in cols else "pmc_name"
val_col = "value" if "value" in cols else "counter_value"
disp_col = "dispatch_id" if "dispatch_id" in cols else "id"
agg = defaultdict(lambda: defaultdict(float))
ndisp = defaultdict(set)
for kname, cname, val, d in c.execute(f"select {name_col}, {cnt_col}, {val_col}, {disp_col} from counters_collection"):
    agg[kname][cname] += val
    ndisp[kname].add(d)
dur = {}
try:
    for kname, s, n in c.execute("select name, sum(end-start), count(*) from kernels group by name"):
        dur[kname] = (s, n)
except Exception:
    pass
counters = sorted({cn for k in agg for cn in agg[k]})
print("kernel\tdispatches\ttotal_ms\t" + "\t".join(counters))
for k in sorted(agg, key=lambda k: -dur.get(k, (0, 0))[0]):
    short = k.replace("pw::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:48]
    print(f"{short}\t{len(ndisp[k])}\t{dur.get(k, (0, 0))[0] / 1e6:.3f}\t" + "\t".join(f"{agg[k].get(cn, 0):.4g}" for cn in counters))
