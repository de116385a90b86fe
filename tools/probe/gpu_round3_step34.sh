set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_mixed_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -3 || exit 1
timeout -k 10 900 python tools/fuzz_mixed.py 800 77 > gpurun_out/s34_fuzz_mixed.txt 2>&1
tail -1 gpurun_out/s34_fuzz_mixed.txt
python bench.py --workload config4 --no-cpu-baseline --no-extra-mode --steps 10 --warmup 3 > gpurun_out/s34_c4.json 2> gpurun_out/s34_c4.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/s34_c4.json").read().strip().splitlines()[-1]); print(round(j["ms_per_step"],3), "%.4e"%j["value"], round(j["roofline"]["avg_launch_ms_unshared"],3))
PY
