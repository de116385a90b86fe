set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 || exit 1
B="--no-cpu-baseline --no-extra-mode --steps 30 --warmup 5"
for wl in config1 config3 config2; do
python bench.py $B --workload $wl > gpurun_out/s32_rcp4_$wl.json 2> gpurun_out/s32_rcp4_$wl.err || exit 1
done
ES_BUILD_EXTRA_FLAGS="-DES_NO_RCP4" python -m eigensolver_amd.build --force > gpurun_out/s32_build.log 2>&1 || exit 2
for wl in config1 config3 config2; do
python bench.py $B --workload $wl > gpurun_out/s32_base_$wl.json 2> gpurun_out/s32_base_$wl.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s32_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(j["ms_per_step"],3), "%.4e"%j["value"], j["roofline"]["kernel"], round(j["roofline"]["avg_launch_ms"],3), round(j["roofline"]["avg_launch_ms_unshared"],3))
PY
