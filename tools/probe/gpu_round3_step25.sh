set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="--no-cpu-baseline --no-extra-mode --steps 30 --warmup 5"
# the library in the snapshot is the experiment build (-DES_RCP4_EXPERIMENT)
python bench.py $B > gpurun_out/s25_rcp4_a.json 2> gpurun_out/s25_rcp4_a.err &&
ES_GRID_SHAPE=4,2 python bench.py $B > gpurun_out/s25_rcp4_wpe2.json 2> gpurun_out/s25_rcp4_wpe2.err &&
python bench.py $B > gpurun_out/s25_rcp4_b.json 2> gpurun_out/s25_rcp4_b.err &&
python -m eigensolver_amd.build --force > gpurun_out/s25_build.log 2>&1 &&
python bench.py $B > gpurun_out/s25_base_a.json 2> gpurun_out/s25_base_a.err &&
python bench.py $B > gpurun_out/s25_base_b.json 2> gpurun_out/s25_base_b.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s25_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(j["ms_per_step"],3), "%.4e"%j["value"], j["roofline"]["avg_launch_ms"], j["roofline"]["avg_launch_ms_unshared"])
PY
