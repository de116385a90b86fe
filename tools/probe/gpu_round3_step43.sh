set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 || exit 1
python tools/probe/time_row_width.py 2>&1 | tail -5
ES_GRID_ROWS2=0 python tools/probe/time_row_width.py 2>&1 | tail -5
B="--no-cpu-baseline --no-extra-mode --steps 30 --warmup 5"
python bench.py $B --workload config2 > gpurun_out/s43_c2.json 2> gpurun_out/s43_c2.err &&
python bench.py $B > gpurun_out/s43_c3.json 2> gpurun_out/s43_c3.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s43_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(j["ms_per_step"],3), "%.4e"%j["value"], j["roofline"]["kernel"], round(j["roofline"]["avg_launch_ms"],3), round(j["roofline"]["avg_launch_ms_unshared"],3), j["valu_issue"]["frac_at_2p1_ghz_unshared"])
PY
