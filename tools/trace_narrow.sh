#!/bin/bash
# device-side median durations of the narrow-GEMM micro-benchmark (GPU box only). usage: tools/trace_narrow.sh [dbg...]
cd "$(dirname "$0")/.." || exit 1
for d in "${@:-0}"; do
  C2_GEMM_DBG=$d timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pn$d -- python3 tools/bench_gemm_narrow.py > /dev/null 2>&1 || exit 1
  echo "dbg=$d"
  python3 - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob("gpurun_out/pn$d/*/*kernel_trace.csv")[0])))
rows=[r for r in rows if "gemm_wide" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
i=0
out=[]
while i < len(rows):
    g=rows[i:i+23]; dd=sorted(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in g)
    out.append("%s g%d %.1f" % (g[0]["Kernel_Name"][31:42].replace(" ",""), int(g[0]["Grid_Size_X"])//int(g[0]["Workgroup_Size_X"]), dd[len(dd)//2]/1e3))
    i+=23
print(" | ".join(out))
PY
  rm -rf gpurun_out/pn$d
done
