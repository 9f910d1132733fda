"""HBM bytes per bench step and kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output):
FETCH_SIZE is reported in KiB and, on gfx950, counts wide coalesced reads at half their bytes (MI355X_MICROARCH.md,
HBM section) -> x 2; WRITE_SIZE (KiB) as reported. usage: pmc_traffic_json.py FETCH_DIR WRITE_DIR N_STEPS [WORKLOAD LABEL] > out.json   (N_STEPS = warm-up + timed steps)"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def sums(d, counter):
    agg, disp = defaultdict(float), defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            k = k.replace("pw::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")
            k = re.sub(r"^void ", "", k)
            k = re.sub(r"\(.*$", "", k)
            k = re.sub(r"<\d+>$", "", k)
            agg[k] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    return agg, disp


fetch, disp = sums(sys.argv[1], "FETCH_SIZE")
write, _ = sums(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3])
out = {}
for k in sorted(fetch, key=lambda k: -fetch[k]):
    if k.startswith("at::") or "rocclr" in k and fetch[k] + write.get(k, 0) == 0:
        continue
    out[k] = dict(dispatches=len(disp[k]) / steps, fetch_bytes_reported=fetch[k] * 1024 / steps, fetch_bytes_corrected=fetch[k] * 1024 * 2 / steps,
                  write_bytes=write.get(k, 0.0) * 1024 / steps)
label = sys.argv[4] if len(sys.argv) > 4 else "C2 2022 cols x 2^20 rows"
print(json.dumps(dict(workload=label + ", bytes per step (N_STEPS = warm-up + timed steps of the profiled run)", kernels=out,
                      note="bytes per step; fetch_bytes_corrected = FETCH_SIZE x2: the gfx950 correction of MI355X_MICROARCH.md (HBM section) is calibrated for wide "
                           "coalesced streaming reads (16 B per lane: leaf hash, DEEP, openings, gather, fused LDE); kernels that read 64-byte segments "
                           "(the strided NTT groups) are described by fetch_bytes_reported instead; WRITE_SIZE as reported (uncalibrated)"), indent=1))
