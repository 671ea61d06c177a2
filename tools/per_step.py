"""Per-step table from a tools/rocpd_stats.py summary: kernels launched at least once per step, launches and microseconds per step.
usage: python tools/per_step.py <kernel_stats.txt> [steps]   (steps: default = the most common call count)"""
import collections
import re
import sys

rows = []
for line in open(sys.argv[1]).read().splitlines()[1:]:
    m = re.match(r"(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)$", line)
    if m:
        rows.append((m.group(1)[:96], int(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5))))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else collections.Counter(c for _, c, _, _, _ in rows).most_common(1)[0][0]
per = [(n, c / steps, t / steps, a, mi) for n, c, t, a, mi in rows if c >= steps and c % steps == 0]
print(f"steps {steps}  launches/step {sum(x[1] for x in per):.1f}  us/step {sum(x[2] for x in per):.1f}")
for x in per:
    print(f"{x[0]:96s} {x[1]:5.1f} {x[2]:7.1f} avg {x[3]:7.2f} min {x[4]:7.2f}")
