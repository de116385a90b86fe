# GPU box: (1) headline A/B of the register cap of the wide-row shape (3 waves per SIMD with 6 spilled values around the
# exterior against 2 waves per SIMD without any), (2) strong-scaling projection E = 1, 2, 4, 8 with the default 3 streams,
# (3) rehearsal of the N > 1 paths on one GPU (gloo ranks sharing cuda:0): merged tables identical to N = 1
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
for V in default 4,2; do
  if [ $V = default ]; then unset ES_GRID_SHAPE; else export ES_GRID_SHAPE=$V; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-mode --steps 40 --warmup 3 > gpurun_out/ab_shape_${V}_$rep.json 2> gpurun_out/ab_shape_${V}_$rep.err || { echo "A/B $V failed"; tail -3 gpurun_out/ab_shape_${V}_$rep.err; }
  python -c "import json;j=json.loads(open('gpurun_out/ab_shape_${V}_$rep.json').read().strip().splitlines()[-1]);print('shape $V rep $rep: ms/step %.3f grid alone %.3f shared %.3f value %.4e' % (j['ms_per_step'], j['roofline']['avg_launch_ms_unshared'], j['roofline']['avg_launch_ms'], j['value']))"
done
done
unset ES_GRID_SHAPE
bash tools/project_scaling.sh 2>&1 | tail -5
BENCH_ARGS="" bash tools/rehearse_multi_gpu.sh "2 4" 2>&1 | tail -3
mv gpurun_out/roots_n1.npy gpurun_out/roots_c3_n1.npy
BENCH_ARGS="--workload config4" bash tools/rehearse_multi_gpu.sh "2 3" 2>&1 | tail -3
BENCH_ARGS="--workload config1" bash tools/rehearse_multi_gpu.sh "2" 2>&1 | tail -2
