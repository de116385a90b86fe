set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2 3; do
for V in default 4,2; do
  if [ $V = default ]; then unset ES_GRID_SHAPE; else export ES_GRID_SHAPE=$V; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-mode --steps 40 --warmup 3 > gpurun_out/ab_shape_${V}_$rep.json 2> gpurun_out/ab_shape_${V}_$rep.err || { echo "A/B $V failed"; tail -3 gpurun_out/ab_shape_${V}_$rep.err; }
  python -c "import json;j=json.loads(open('gpurun_out/ab_shape_${V}_$rep.json').read().strip().splitlines()[-1]);print('shape $V rep $rep: ms/step %.3f grid alone %.3f shared %.3f value %.4e' % (j['ms_per_step'], j['roofline']['avg_launch_ms_unshared'], j['roofline']['avg_launch_ms'], j['value']))"
done
done
export ES_GRID_SHAPE=4,2
timeout -k 10 200 python bench.py --share-of 8 --no-cpu-baseline --no-extra-mode --steps 60 --warmup 3 > gpurun_out/share8_wpe2.json 2>/dev/null
python -c "import json;j=json.loads(open('gpurun_out/share8_wpe2.json').read().strip().splitlines()[-1]);print('E=8 wpe2 ms/step %.3f' % j['ms_per_step'])"
unset ES_GRID_SHAPE
timeout -k 10 200 python bench.py --share-of 8 --no-cpu-baseline --no-extra-mode --steps 60 --warmup 3 > gpurun_out/share8_wpe3.json 2>/dev/null
python -c "import json;j=json.loads(open('gpurun_out/share8_wpe3.json').read().strip().splitlines()[-1]);print('E=8 wpe3 ms/step %.3f' % j['ms_per_step'])"
BENCH_ARGS="--workload config4" bash tools/rehearse_multi_gpu.sh "2 3" 2>&1 | tail -3
BENCH_ARGS="--workload config1" bash tools/rehearse_multi_gpu.sh "2" 2>&1 | tail -2
BENCH_ARGS="--workload config2" bash tools/rehearse_multi_gpu.sh "3" 2>&1 | tail -2
