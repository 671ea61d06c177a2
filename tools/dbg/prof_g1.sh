root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/prof_g1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/trace -- python3 $root/bench.py --profile-pass --config c4g1 --steps 20 --warmup 5 > $out/trace.log 2>&1 || exit 3
db=$(find $out/trace -name "*.db" | head -1)
python3 $root/tools/rocpd_stats.py $db > $out/c4g1_kernel_stats.txt || exit 4
rm -rf $out/trace
