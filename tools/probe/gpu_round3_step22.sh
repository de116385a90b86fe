set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 &&
for wl in config3 config1 config2 config4; do
bash tools/profile_bench.sh r3g $wl > gpurun_out/s22_profile_$wl.log 2>&1 || { tail -5 gpurun_out/s22_profile_$wl.log; exit 1; }
done
python bench.py --workload config4 --precision f64 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/s22_c4_f64.json 2> gpurun_out/s22_c4_f64.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/bench_r3g*.json"))+["gpurun_out/s22_c4_f64.json"]:
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["ms_per_step"], j["value"], j["roofline"].get("kernel"), j["roofline"].get("avg_launch_ms"), j["roofline"].get("traffic_source"))
PY
