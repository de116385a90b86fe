# GPU box: the full-size parity tests alone (timing them), then the E = 8 tile with 1 / 2 / 3 pipelined streams
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_full_size_parity_gpu.py tests/test_shoot_gpu.py -m gpu -x -q -s > gpurun_out/t4.log 2>&1; echo "pytest rc $?"; grep -E "configs\[|DOP853|passed|failed|Error" gpurun_out/t4.log | cut -c1-400
for S in 1 2 3 4; do
  timeout -k 10 200 python bench.py --share-of 8 --streams $S --no-cpu-baseline --no-extra-mode --steps 60 --warmup 3 > gpurun_out/share8_s$S.json 2> gpurun_out/share8_s$S.err || { echo "streams $S failed"; tail -3 gpurun_out/share8_s$S.err; }
  python -c "import json;j=json.loads(open('gpurun_out/share8_s$S.json').read().strip().splitlines()[-1]);print('E=8 streams $S ms/step %.3f grid alone %.3f shared %.3f' % (j['ms_per_step'], j['roofline']['avg_launch_ms_unshared'], j['roofline']['avg_launch_ms']))"
done
for S in 2 3; do
  timeout -k 10 200 python bench.py --streams $S --no-cpu-baseline --no-extra-mode --steps 30 --warmup 3 > gpurun_out/n1_s$S.json 2> gpurun_out/n1_s$S.err
  python -c "import json;j=json.loads(open('gpurun_out/n1_s$S.json').read().strip().splitlines()[-1]);print('E=1 streams $S ms/step %.3f grid alone %.3f shared %.3f' % (j['ms_per_step'], j['roofline']['avg_launch_ms_unshared'], j['roofline']['avg_launch_ms']))"
done
