set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="--no-cpu-baseline --no-extra-mode --steps 30 --warmup 5"
for wl in config3 config2; do python bench.py $B --workload $wl > gpurun_out/s46_new_$wl.json 2> gpurun_out/s46_new_$wl.err || exit 1; done
ES_BUILD_EXTRA_FLAGS="-DES_STAGE_FAR_NODE_ALWAYS" python -m eigensolver_amd.build --force > gpurun_out/s46_build.log 2>&1 || exit 2
for wl in config3 config2; do python bench.py $B --workload $wl > gpurun_out/s46_old_$wl.json 2> gpurun_out/s46_old_$wl.err || exit 1; done
python -m eigensolver_amd.build --force > gpurun_out/s46_build2.log 2>&1 || exit 2
for wl in config3 config2; do python bench.py $B --workload $wl > gpurun_out/s46_newb_$wl.json 2> gpurun_out/s46_newb_$wl.err || exit 1; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s46_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(j["ms_per_step"],3), "%.4e"%j["value"], round(j["roofline"]["avg_launch_ms"],3), round(j["roofline"]["avg_launch_ms_unshared"],3))
PY
