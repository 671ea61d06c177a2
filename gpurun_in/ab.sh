set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_ops_gpu.py -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/rot_tests.log 2>&1
for i in 1 2 3; do
  C2MI_LIB=$GRAFT_REPO_ROOT/gpurun_in/libc2mi_rot1a.so timeout -k 10 150 python bench.py --config c4 --steps 30 --warmup 5 > gpurun_out/rot1a_$i.json 2> gpurun_out/rot1a_$i.err
  timeout -k 10 150 python bench.py --config c4 --steps 30 --warmup 5 > gpurun_out/rot1b_$i.json 2> gpurun_out/rot1b_$i.err
done
