#!/bin/bash
# Counter passes over the NB loss micro-benchmark (GPU box only): SQ wait / VMEM level, then the vector memory path
# (TA / TCP).  usage: tools/pmc_nb.sh <tag>
set -o pipefail
tag=$1
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/pmc_nb_$tag
mkdir -p $out
run() {  # name, counters...
    local n=$1; shift
    timeout -s KILL 90 rocprofv3 --output-format csv --pmc "$@" -d $out/$n -- python3 tools/bench_nb.py > $out/$n.log 2>&1 || return 1
    python3 tools/pmc_table.py "$(find $out/$n -name '*counter_collection.csv' | head -1)" nb_rowreg >> $out/summary.txt
    rm -rf $out/$n
}
: > $out/summary.txt
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_WAIT_ANY || exit 2
run tcp1 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE || exit 3
run tcp2 TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum || exit 4
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum || exit 5
cat $out/summary.txt
