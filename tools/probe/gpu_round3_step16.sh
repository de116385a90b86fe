set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 &&
python bench.py --workload config4 --precision f64 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/s16_c4_f64.json 2> gpurun_out/s16_c4_f64.err &&
bash tools/profile_bench.sh r3e config4 > gpurun_out/s16_profile.log 2>&1
tail -3 gpurun_out/s16_profile.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s16_*.json"))+["gpurun_out/bench_r3e_config4.json"]:
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["ms_per_step"], j["value"])
PY
