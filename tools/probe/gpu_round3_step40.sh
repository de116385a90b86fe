set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_shoot_gpu.py tests/test_mixed_gpu.py tests/test_configs_gpu.py tests/test_workers_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -3 || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s40_config4 -- python3 $R/tools/probe/one_unit_mixed.py > $R/gpurun_out/s40_config4.log 2>&1 || { tail -5 $R/gpurun_out/s40_config4.log; exit 2; }
cd $R
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/s40_config4/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:5]:
    print("  ", r["Name"][28:80], r["Calls"], "total %.2f ms avg %.3f min %.3f"%(float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6, float(r["MinNs"])/1e6))
PY
B="--no-cpu-baseline --no-extra-mode --workload config4 --steps 12 --warmup 3"
python bench.py $B > gpurun_out/s40_c4_a.json 2> gpurun_out/s40_c4_a.err
python bench.py $B --precision f64 > gpurun_out/s40_c4_f64.json 2> gpurun_out/s40_c4_f64.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s40_c4*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(j["ms_per_step"],3), "%.4e"%j["value"])
PY
