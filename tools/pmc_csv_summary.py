"""Per-kernel sums of PMC counters from rocprofv3 --output-format csv (*_counter_collection.csv files under a dir).
usage: pmc_csv_summary.py DIR [kernel-substring]"""
import csv
import glob
import sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name") or r.get("kernel_name")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r.get("Dispatch_Id"))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(agg):
    if flt in k:
        print(k[:70], len(disp[k]), {c: f"{v:.4g}" for c, v in sorted(agg[k].items())})
