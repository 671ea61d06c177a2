#!/bin/bash
# SQ counters of one tool script's kernels: tools/pmc_kernel.sh <tag> <python script> -- one rocprofv3 --pmc pass (8 SQ
# counters), csv output under gpurun_out/<tag>/.  GPU box only.
set -o pipefail
tag=$1; script=$2
cd "$(dirname "$0")/.." || exit 1
out=$PWD/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $out/pmc -- python3 $GRAFT_REPO_ROOT/$script > $out/pmc.log 2>&1 || exit 4
find $out/pmc -name "*counter_collection.csv" | head -1
