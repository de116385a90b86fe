set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_mixed_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -2 || exit 1
timeout -k 10 500 python tools/fuzz_mixed.py 800 77 > gpurun_out/s51_fuzz.txt 2>&1; tail -1 gpurun_out/s51_fuzz.txt
B="--no-cpu-baseline --no-extra-mode --workload config4 --steps 12 --warmup 3"
python bench.py $B > gpurun_out/s51_new_a.json 2> gpurun_out/s51_new_a.err
python bench.py $B > gpurun_out/s51_new_b.json 2> gpurun_out/s51_new_b.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s51_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(j["ms_per_step"],3), "%.4e"%j["value"], round(j["roofline"]["avg_launch_ms"],3), round(j["roofline"]["avg_launch_ms_unshared"],3))
PY
