"""Per-kernel duration summary of a rocprofv3 results database (rocpd sqlite): name, calls, total / average / min us."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
scols = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
name_col = "display_name" if "display_name" in scols else "kernel_name"
rows = cur.execute(f"select s.{name_col}, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id").fetchall()
agg = defaultdict(list)
for n, s, e in rows:
    agg[n].append((e - s) / 1e3)
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tot = sum(sum(v) for v in agg.values())
print(f"{'kernel':90s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'min_us':>9s} {'%':>6s}")
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n[:90]:90s} {len(v):6d} {sum(v):10.1f} {sum(v) / len(v):9.2f} {min(v):9.2f} {100 * sum(v) / tot:6.2f}")
