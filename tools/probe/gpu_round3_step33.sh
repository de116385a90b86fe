set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for wl in config3 config1 config2 config4; do
bash tools/profile_bench.sh r3h $wl > gpurun_out/s33_profile_$wl.log 2>&1 || { tail -5 gpurun_out/s33_profile_$wl.log; exit 1; }
done
python bench.py --workload config4 --precision f64 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/s33_c4_f64.json 2> gpurun_out/s33_c4_f64.err
bash tools/project_scaling.sh > gpurun_out/s33_scaling.txt 2>&1; cat gpurun_out/s33_scaling.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/bench_r3h*.json"))+["gpurun_out/s33_c4_f64.json"]:
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["ms_per_step"], j["value"], j["roofline"].get("kernel"), j["roofline"].get("avg_launch_ms"), j["roofline"].get("avg_launch_ms_unshared"), j["roofline"].get("traffic_source"))
PY
