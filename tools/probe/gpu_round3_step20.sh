set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# bench lines again with the PMC summaries of the same build in place (profiles/pmc_latest_*.json = r3f)
python bench.py > gpurun_out/bench_r3f.json 2> gpurun_out/bench_r3f.err &&
python bench.py --workload config1 > gpurun_out/bench_r3f_config1.json 2> gpurun_out/bench_r3f_config1.err &&
python bench.py --workload config2 > gpurun_out/bench_r3f_config2.json 2> gpurun_out/bench_r3f_config2.err &&
python bench.py --workload config4 > gpurun_out/bench_r3f_config4.json 2> gpurun_out/bench_r3f_config4.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/bench_r3f*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["ms_per_step"], j["value"], j["roofline"]["traffic"], j["valu_issue"]["frac"], j["valu_issue"]["source"])
PY
