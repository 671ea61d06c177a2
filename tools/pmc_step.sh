#!/bin/bash
# HBM traffic of the gene-width GEMM launches of the REAL training step: two counter passes over bench.py (FETCH_SIZE and
# WRITE_SIZE need separate runs), then tools/pmc_step_json.py.  GPU box only.  usage: tools/pmc_step.sh <tag>
set -o pipefail
tag=$1
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/pmc_step_$tag
mkdir -p $out
W=2; K=3
timeout -s KILL 200 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $out/fetch -- python3 bench.py --profile-pass --graph 0 --steps $K --warmup $W > $out/fetch.log 2>&1 || exit 4
timeout -s KILL 200 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $out/write -- python3 bench.py --profile-pass --graph 0 --steps $K --warmup $W > $out/write.log 2>&1 || exit 5
python3 tools/pmc_step_json.py $out $((W + K)) > $out/r03_gemm_pmc.json || exit 6
rm -rf $out/fetch $out/write
cat $out/r03_gemm_pmc.json
