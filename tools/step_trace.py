"""Per-launch kernel durations of ONE training step, from a rocprofv3 --kernel-trace csv of bench.py.
usage: python tools/step_trace.py <kernel_trace.csv> [step_index_from_end]"""
import csv, sys

def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a step starts at each gather_rows launch
    starts = [i for i, r in enumerate(rows) if "gather_rows" in r["Kernel_Name"]]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
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
