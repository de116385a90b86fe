set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_shoot_gpu.py tests/test_configs_gpu.py -m gpu -x -q > gpurun_out/t16.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/t16.log
for rep in 1 2; do
for V in default 2,3 3,3; do
  if [ $V = default ]; then unset ES_GRID_SHAPE; else export ES_GRID_SHAPE=$V; fi
  timeout -k 10 200 python bench.py --workload config2 --no-cpu-baseline --steps 30 > gpurun_out/j_c2_$V.json 2>/dev/null
  python -c "import json;j=json.loads(open('gpurun_out/j_c2_$V.json').read().strip().splitlines()[-1]);print('config2 shape $V ms/step %.3f launch %.3f alone %.3f %s' % (j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['avg_launch_ms_unshared'], j['roofline']['kernel']))"
done
done
unset ES_GRID_SHAPE
