set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_shoot_gpu.py tests/test_configs_gpu.py tests/test_fuzz_gpu.py tests/test_errors_gpu.py -m gpu -x -q 2>&1 | tail -5 || exit 1
python tools/probe/time_row_width.py 2>&1 | tail -5
ES_GRID_ROWS2=0 python tools/probe/time_row_width.py 2>&1 | tail -5
B="--no-cpu-baseline --no-extra-mode --steps 30 --warmup 5 --workload config2"
python bench.py $B > gpurun_out/s42_r2_a.json 2> gpurun_out/s42_r2_a.err &&
ES_GRID_ROWS2=0 python bench.py $B > gpurun_out/s42_r1_a.json 2> gpurun_out/s42_r1_a.err &&
python bench.py $B > gpurun_out/s42_r2_b.json 2> gpurun_out/s42_r2_b.err &&
ES_GRID_ROWS2=0 python bench.py $B > gpurun_out/s42_r1_b.json 2> gpurun_out/s42_r1_b.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s42_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(j["ms_per_step"],3), "%.4e"%j["value"], j["roofline"]["kernel"], round(j["roofline"]["avg_launch_ms"],3), round(j["roofline"]["avg_launch_ms_unshared"],3))
PY
