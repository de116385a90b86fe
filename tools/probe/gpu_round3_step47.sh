set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for wl in config3 config1 config2 config4; do
bash tools/profile_bench.sh r3j $wl > gpurun_out/s47_profile_$wl.log 2>&1 || { tail -5 gpurun_out/s47_profile_$wl.log; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/bench_r3j*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["ms_per_step"], j["value"], j["roofline"].get("kernel"), j["roofline"].get("avg_launch_ms"), j["roofline"].get("avg_launch_ms_unshared"), j["valu_issue"]["frac"])
PY
