set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 || exit 1
B="--no-cpu-baseline --no-extra-mode"
for wl in config3 config2; do
python bench.py $B --workload $wl > gpurun_out/s26_$wl.json 2> gpurun_out/s26_$wl.err || exit 1
done
ES_REFINE_ROUNDS_IN_KERNEL=1 python bench.py $B --workload config3 > gpurun_out/s26_config3_inkernel.json 2> gpurun_out/s26_config3_inkernel.err
ES_REFINE_PRIVATE_ENTRIES=1 python bench.py $B --workload config3 > gpurun_out/s26_config3_private.json 2> gpurun_out/s26_config3_private.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s26_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(j["ms_per_step"],3), "%.4e"%j["value"], j["roofline"]["avg_launch_ms"], j["roofline"]["avg_launch_ms_unshared"])
PY
