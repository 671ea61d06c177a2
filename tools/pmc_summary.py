"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per dispatch, per kernel name.
usage: python tools/pmc_summary.py <dir-or-csv> [name-filter]"""
import csv, glob, os, sys
from collections import defaultdict

def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else "c2::"
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    for f in files:
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if flt not in n:
                continue
            key = n[:90] + f" grid={r['Grid_Size']}"
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[key][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k in sorted(acc):
        d = list(dur[k].values())
        print(f"{k}\n   dispatches={len(d)} mean_dur_us={sum(d)/len(d)/1e3:.1f}")
        for c, v in sorted(acc[k].items()):
            print(f"   {c:32s} {sum(v)/len(v):.4g}")

if __name__ == "__main__":
    main()
