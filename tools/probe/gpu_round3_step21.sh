set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_shoot_gpu.py tests/test_mixed_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tail -3 || exit 1
B="--no-cpu-baseline --no-extra-mode"
run() { tag=$1; shift; env "$@" python bench.py $B --workload $WL > gpurun_out/s21_${WL}_$tag.json 2> gpurun_out/s21_${WL}_$tag.err || exit 1; }
for WL in config3 config4 config2 config1; do
run new X=1
run old ES_REFINE_ROUNDS_IN_KERNEL=1
done
WL=config4; run wpe2 ES_REFINE_TWIST_WPE2=1
WL=config3; run private ES_REFINE_PRIVATE_ENTRIES=1
WL=config2; run private ES_REFINE_PRIVATE_ENTRIES=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s21_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(j["ms_per_step"],3), "%.4e"%j["value"])
PY
