set -o pipefail
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  C2_GEMM_BF16_128=$v timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d /tmp/tr$v -- python3 $GRAFT_REPO_ROOT/bench.py --profile-pass --config c2 --steps 200 --warmup 20 > /tmp/out$v.txt 2>&1 || { echo "run $v failed"; tail -5 /tmp/out$v.txt; exit 1; }
  echo "== C2_GEMM_BF16_128=$v: $(tail -1 /tmp/out$v.txt | cut -c1-200)"
  python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/tr$v -name "*.db" | head -1) | grep -i "gemm" > /tmp/g$v.txt; cat /tmp/g$v.txt | cut -c1-130
done
