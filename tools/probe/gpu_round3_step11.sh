# GPU box: launch shape of the headline family on the 512- and 1024-row tiles of a multi-GPU run (tail quantisation:
# 2048 four-point tiles on 768 resident workgroup slots are 2.67 rounds)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for E in 8 4 2 1; do
for V in default 2,3 1,4; do
  if [ $V = default ]; then unset ES_GRID_SHAPE; else export ES_GRID_SHAPE=$V; fi
  timeout -k 10 200 python bench.py --share-of $E --no-cpu-baseline --no-extra-mode --steps 40 --warmup 3 > gpurun_out/g_${E}_$V.json 2>/dev/null
  python -c "import json;j=json.loads(open('gpurun_out/g_${E}_$V.json').read().strip().splitlines()[-1]);print('E=$E shape $V ms/step %.3f launch %.3f alone %.3f' % (j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['avg_launch_ms_unshared']))"
done
done
unset ES_GRID_SHAPE
