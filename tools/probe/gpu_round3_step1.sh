set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python tools/probe/time_grid_shapes.py > gpurun_out/grid_shapes.log 2>&1 || { tail -20 gpurun_out/grid_shapes.log; exit 1; }
tail -3 gpurun_out/grid_shapes.log
timeout -k 10 500 python -m pytest tests/test_shoot_gpu.py tests/test_configs_gpu.py tests/test_mixed_gpu.py tests/test_exchange_gpu.py -m gpu -x -q > gpurun_out/t1.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/t1.log
for w in config1 config2 config4 config3; do
  timeout -k 10 240 python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/b_$w.json 2> gpurun_out/b_$w.err; echo "$w rc $?"; tail -c 600 gpurun_out/b_$w.err; head -c 1500 gpurun_out/b_$w.json; echo
done
