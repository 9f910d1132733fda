"""Where the GPU idles inside a run: from a rocprofv3 --kernel-trace csv, the gaps between the end of everything dispatched so
far and the start of the next kernel, summed by the kernel that started late (gaps above `--max-gap-us` are pauses between
proofs and ignored).   usage: gap_analysis.py DIR [--max-gap-us 2000] [--skip-first N]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
max_gap = 2000.0
if "--max-gap-us" in sys.argv:
    max_gap = float(sys.argv[sys.argv.index("--max-gap-us") + 1])
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
busy = 0.0
gaps = defaultdict(lambda: [0, 0.0])
end = None
total_gap = 0.0
for s, e, n in rows:
    name = n.replace("void ", "").replace("pw::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    if end is not None and s > end:
        g = (s - end) / 1000.0
        if g <= max_gap:
            gaps[name][0] += 1
            gaps[name][1] += g
            total_gap += g
    busy += (e - s) / 1000.0
    end = e if end is None else max(end, e)
# wall time covered by at least one kernel (kernels of different streams overlap in the segment prover) and by at least two
cover = over = 0.0
ev = sorted([(s, 1) for s, e, n in rows] + [(e, -1) for s, e, n in rows])
depth, last = 0, None
for t, dlt in ev:
    if last is not None and depth >= 1:
        cover += (t - last) / 1000.0
    if last is not None and depth >= 2:
        over += (t - last) / 1000.0
    depth += dlt
    last = t
print(f"{len(rows)} dispatches, kernel time {busy / 1000:.1f} ms, idle (gaps <= {max_gap:.0f} us) {total_gap / 1000:.1f} ms; "
      f"wall with a kernel running {cover / 1000:.1f} ms, with two or more {over / 1000:.1f} ms")
for name, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {t / 1000:8.2f} ms in {c:6d} gaps (avg {t / c:7.1f} us) before {name[:90]}")
