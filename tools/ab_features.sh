#!/bin/bash
# Same-box A/B of the round-3 features of bf16 mode (environment switches read by engine.StepEngine): step time of the C4
# and C2 benchmark configurations with each switched off.  GPU box only.  usage: tools/ab_features.sh [out file]
cd "$(dirname "$0")/.." || exit 1
out=${1:-gpurun_out/ab_features.log}
: > $out
for cfg in c4 c2; do
  st=20; [ $cfg = c2 ] && st=200
  for v in "X=1" "C2_CENTER=0" "C2_FWD16=0" "C2_HALF_LOGITS=0" "C2_CENTER=0 C2_FWD16=0 C2_HALF_LOGITS=0" "X=1"; do
    line=$(env $v timeout -k 10 150 python3 bench.py --config $cfg --steps $st --warmup 10 --no-cpu-baseline --steady-seconds 2 2>/dev/null |
           python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['steady_state']['ms_per_step'],4))") || exit 3
    echo "$cfg [$v] ms/step (K steps, steady) $line" | tee -a $out
  done
done
