#!/bin/bash
# The round's profiles in one GPU call: kernel trace + stats of the bench command (C4, C2 and C3, step replayed as a HIP
# graph), the HBM-traffic passes of the gene-width GEMMs, and the bench lines themselves.  Every GPU step has its own
# time limit and the script stops at the first failure.  usage: tools/profile_round.sh <tag>   (GPU box only)
set -o pipefail
tag=$1
cd "$(dirname "$0")/.." || exit 1
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for cfg in c4 c2 c3; do
  steps=20; warm=5
  [ $cfg = c2 ] && { steps=200; warm=20; }
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/trace_$cfg -- python3 $root/bench.py --profile-pass --config $cfg --steps $steps --warmup $warm > $out/trace_$cfg.log 2>&1 || exit 3
  db=$(find $out/trace_$cfg -name "*.db" | head -1)
  python3 $root/tools/rocpd_stats.py $db > $out/${cfg}_kernel_stats.txt || exit 4
  echo "# python3 bench.py --profile-pass --config $cfg --steps $steps --warmup $warm  (rocprofv3 --kernel-trace --stats, summarised by tools/rocpd_stats.py; $((steps + warm + 3)) steps: 3 set-up steps -- two launched kernel by kernel, then the one that captures the graph -- + $warm warm-up + $steps timed replays; the poisson / compare / reduce kernels at the top build the synthetic dataset)" >> $out/${cfg}_kernel_stats.txt
  rm -rf $out/trace_$cfg
done
cd $root
bash tools/pmc_step.sh $tag || exit 5
cp gpurun_out/pmc_step_$tag/r03_gemm_pmc.json $out/ || exit 6
timeout -k 10 300 python3 bench.py > $out/bench_c4.json 2> $out/bench_c4.err || exit 7
timeout -k 10 300 python3 bench.py --config c2 --steps 200 --warmup 20 > $out/bench_c2.json 2> $out/bench_c2.err || exit 8
for cfg in c1 c3 c5 c4g1 c4z; do  # the other configurations of SURVEY.md 8(d) (+ 19 001 genes, + ZINB): bench lines only
  steps=200; [ $cfg = c3 -o $cfg = c4g1 -o $cfg = c4z ] && steps=20
  extra=""; [ $cfg = c5 -o $cfg = c4g1 -o $cfg = c4z ] && extra="--no-cpu-baseline"  # (the numpy port has no frozen-parameter step)
  timeout -k 10 300 python3 bench.py --config $cfg --steps $steps --warmup 10 $extra > $out/bench_$cfg.json 2> $out/bench_$cfg.err || exit 9
done
C2_DP_SINGLE_RANK=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline > $out/bench_c4_dp1_rehearsal.json 2> $out/bench_c4_dp1_rehearsal.err || exit 10
echo done
