set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_shoot_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tail -3 || exit 1
cd /tmp && export TMPDIR=/tmp
for u in "config2 2" "config3 0"; do
set -- $u
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s30_$1 -- python3 $R/tools/probe/one_unit_f64.py $1 $2 > $R/gpurun_out/s30_$1.log 2>&1 || { tail -5 $R/gpurun_out/s30_$1.log; exit 2; }
done
cd $R
python - <<'PY'
import csv,glob
for tag in ("config2","config3"):
    f=glob.glob(f"gpurun_out/s30_{tag}/*/*_kernel_stats.csv")[0]
    print(tag)
    for r in list(csv.DictReader(open(f)))[:3]:
        print("  ", r["Name"][28:80], r["Calls"], "total %.2f ms avg %.3f min %.3f"%(float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6, float(r["MinNs"])/1e6))
PY
