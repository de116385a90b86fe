set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 &&
for wl in config3 config1 config2 config4; do
python bench.py --workload $wl --no-cpu-baseline > gpurun_out/s17_$wl.json 2> gpurun_out/s17_$wl.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s17_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["ms_per_step"], j["value"], j["roofline"].get("avg_launch_ms"), j["roofline"].get("avg_launch_ms_unshared"))
PY
