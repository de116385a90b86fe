set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# twisted-family refinement: wave-shared node entries (16 steps per chunk) against per-lane entries, same box
for i in 1 2; do
python bench.py --workload config4 --steps 10 --warmup 3 > gpurun_out/s14_priv_$i.json 2> gpurun_out/s14_priv_$i.err &&
ES_REFINE_SHARED_TWIST=1 python bench.py --workload config4 --steps 10 --warmup 3 > gpurun_out/s14_shared_$i.json 2> gpurun_out/s14_shared_$i.err || exit 1
done
ES_REFINE_SHARED_TWIST=1 timeout -k 10 600 python -m pytest tests/test_mixed_gpu.py tests/test_shoot_gpu.py -m gpu -x -q 2>&1 | tail -3
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s14_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["ms_per_step"], j["value"])
PY
