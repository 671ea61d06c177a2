#!/bin/bash
# Three counter passes over the GEMM micro-benchmark (GPU box only).  usage: tools/pmc_gemm.sh <tag> [case]
set -o pipefail
tag=$1; which=${2:-all}
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/pmc_$tag
mkdir -p $out
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $out/sq -- python3 tools/bench_gemm.py 2 $which > $out/sq.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE -d $out/tcc -- python3 tools/bench_gemm.py 2 $which > $out/tcc.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $out/fetch -- python3 tools/bench_gemm.py 2 $which > $out/fetch.log 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $out/write -- python3 tools/bench_gemm.py 2 $which > $out/write.log 2>&1 || exit 5
python3 tools/pmc_summary.py $out "gemm" > $out/summary.txt
rm -rf $out/sq $out/tcc $out/fetch $out/write
cat $out/summary.txt
