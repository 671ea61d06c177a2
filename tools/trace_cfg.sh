#!/bin/bash
# kernel trace + stats of one bench configuration's profile pass (GPU box only).  usage: tools/trace_cfg.sh <cfg> <steps> <tag>
set -o pipefail
cfg=$1; steps=$2; tag=$3
cd "$(dirname "$0")/.." || exit 1
root=$PWD
out=$root/gpurun_out/trace_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/t -- python3 $root/bench.py --profile-pass --config $cfg --steps $steps --warmup 5 > $out/run.log 2>&1 || exit 3
db=$(find $out/t -name "*.db" | head -1)
python3 $root/tools/rocpd_stats.py $db > $out/${cfg}_kernel_stats.txt || exit 4
python3 $root/tools/rocpd_steps.py $db gather_rows -5 > $out/${cfg}_steps.txt || exit 5
rm -rf $out/t
head -40 $out/${cfg}_kernel_stats.txt | cut -c1-150
