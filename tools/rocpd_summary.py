"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel stats table
(the equivalent of `--stats` CSV): calls, total/avg/min/max duration, % of GPU kernel time."""
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print(f"{'kernel':<70} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}")
for n, cnt, s, a, mn, mx in rows:
    n = n if len(n) <= 68 else n[:65] + "..."
    print(f"{n:<70} {cnt:>7} {s/1e6:>10.3f} {a/1e3:>10.2f} {mn/1e3:>10.2f} {mx/1e3:>10.2f} {100*s/tot:>6.2f}")
