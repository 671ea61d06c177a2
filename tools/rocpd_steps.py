"""Per-step timeline of a rocprofv3 results database (rocpd sqlite) of bench.py: steps are cut at every launch of the
kernel whose name contains argv[2] (default: gather_rows); per step the wall span, the sum of kernel durations, the largest
idle gap between consecutive kernels and the kernel that followed it."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
scols = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
name_col = "display_name" if "display_name" in scols else "kernel_name"
rows = cur.execute(f"select s.{name_col}, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id order by d.start").fetchall()
mark = sys.argv[2] if len(sys.argv) > 2 else "gather_rows"
cuts = [i for i, r in enumerate(rows) if mark in r[0]]
print(f"{len(rows)} dispatches, {len(cuts)} steps")
for a, b in zip(cuts, cuts[1:] + [len(rows)]):
    seg = rows[a:b]
    busy = sum(e - s for _, s, e in seg) / 1e3
    end = max(e for _, _, e in seg)
    span = ((rows[b][1] if b < len(rows) else end) - seg[0][1]) / 1e3
    gaps = [(seg[i + 1][1] - max(x[2] for x in seg[:i + 1]), seg[i + 1][0]) for i in range(len(seg) - 1)]
    g, n = max(gaps) if gaps else (0, "")
    slow = max(seg, key=lambda r: r[2] - r[1])
    print(f"step at {seg[0][1] / 1e6:12.3f} ms: span {span:9.1f} us, kernels {len(seg):4d}, busy {busy:9.1f} us, largest gap {g / 1e3:8.1f} us before {n[:50]}; longest {slow[0][:50]} {(slow[2] - slow[1]) / 1e3:.1f} us")

# argv[3] (optional): index of a step whose launches are listed one by one (negative: from the end)
if len(sys.argv) > 3:
    k = int(sys.argv[3])
    a = cuts[k]
    b = cuts[k + 1] if k + 1 < len(cuts) and k != -1 else len(rows)
    t0 = rows[a][1]
    prev_end = t0
    print(f"--- step {k}: {b - a} launches")
    for n, s, e in rows[a:b]:
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f}  {n[:110]}")
        prev_end = max(prev_end, e)
