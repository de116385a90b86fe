set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="--no-cpu-baseline --no-extra-mode --steps 30 --warmup 5 --workload config2"
python bench.py $B > gpurun_out/s37_def_a.json 2> gpurun_out/s37_def_a.err &&
ES_GRID_SHAPE=3,3 python bench.py $B > gpurun_out/s37_p3_a.json 2> gpurun_out/s37_p3_a.err &&
python bench.py $B > gpurun_out/s37_def_b.json 2> gpurun_out/s37_def_b.err &&
ES_GRID_SHAPE=3,3 python bench.py $B > gpurun_out/s37_p3_b.json 2> gpurun_out/s37_p3_b.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s37_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(j["ms_per_step"],3), "%.4e"%j["value"], j["roofline"]["kernel"], round(j["roofline"]["avg_launch_ms"],3), round(j["roofline"]["avg_launch_ms_unshared"],3))
PY
