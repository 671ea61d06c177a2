"""Per-launch kernel durations of ONE training step, from a rocprofv3 --kernel-trace csv of bench.py.
usage: python tools/step_trace.py <kernel_trace.csv> [step_index_from_end]"""
import csv, sys

def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a step starts at each gather_rows launch.  bench.py times its K headline steps and then (single GPU) 2 + K steps
    # of the all-terms-in-step pipeline, whose gather instance writes fp32 copies (template argument `float` with the
    # counts copied): the default picks a step from the middle of the headline loop.
    starts = [i for i, r in enumerate(rows) if "gather_rows" in r["Kernel_Name"]]
    dur = [int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i in starts]
    light = [j for j, d in enumerate(dur) if d < 3 * min(dur)]  # headline steps: the gather moves per-row scalars only
    k = int(sys.argv[2]) if len(sys.argv) > 2 else (len(starts) - light[len(light) // 2] if len(light) > 3 else 2)
    lo, hi = starts[-k - 1], starts[-k]
    t0 = int(rows[lo]["Start_Timestamp"])
    tot = 0
    for r in rows[lo:hi]:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        tot += d
        g = [int(r[f"Grid_Size_{a}"]) // max(1, int(r[f"Workgroup_Size_{a}"])) for a in "XYZ"] if "Grid_Size_X" in r else [r.get("Grid_Size")]
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d / 1e3:8.1f} us  grid={g}  {r['Kernel_Name'][:100]}")
    print(f"sum of kernel time {tot / 1e3:.1f} us; wall {(int(rows[hi]['Start_Timestamp']) - t0) / 1e3:.1f} us")

if __name__ == "__main__":
    main()
