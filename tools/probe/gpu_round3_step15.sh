set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_mixed_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -3 &&
python bench.py --workload config4 --steps 10 --warmup 3 > gpurun_out/s15_c4_1.json 2> gpurun_out/s15_c4_1.err &&
python bench.py --workload config4 --steps 10 --warmup 3 > gpurun_out/s15_c4_2.json 2> gpurun_out/s15_c4_2.err &&
timeout -k 10 900 python tools/fuzz_mixed.py 800 77 > gpurun_out/s15_fuzz_mixed.txt 2>&1
tail -3 gpurun_out/s15_fuzz_mixed.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s15_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["ms_per_step"], j["value"], j["roofline"].get("kernel"), j["roofline"].get("avg_ms"))
PY
