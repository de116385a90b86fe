set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s31_config2 -- python3 $R/tools/probe/one_unit_f64.py config2 2 > $R/gpurun_out/s31_config2.log 2>&1 || { tail -5 $R/gpurun_out/s31_config2.log; exit 2; }
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s31_config1 -- python3 $R/tools/probe/one_unit_f64.py config1 1 > $R/gpurun_out/s31_config1.log 2>&1 || { tail -5 $R/gpurun_out/s31_config1.log; exit 2; }
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s31_config4 -- python3 $R/tools/probe/one_unit_mixed.py > $R/gpurun_out/s31_config4.log 2>&1 || { tail -5 $R/gpurun_out/s31_config4.log; exit 2; }
cd $R
python - <<'PY'
import csv,glob
for tag in ("config2","config1","config4"):
    f=glob.glob(f"gpurun_out/s31_{tag}/*/*_kernel_stats.csv")[0]
    print(tag)
    for r in list(csv.DictReader(open(f)))[:5]:
        print("  ", r["Name"][28:80], r["Calls"], "total %.2f ms avg %.3f min %.3f"%(float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6, float(r["MinNs"])/1e6))
PY
B="--no-cpu-baseline --no-extra-mode"
for wl in config3 config1 config2 config4; do
python bench.py $B --workload $wl > gpurun_out/s31_$wl.json 2> gpurun_out/s31_$wl.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s31_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(j["ms_per_step"],3), "%.4e"%j["value"], round(j["roofline"]["avg_launch_ms"],3), round(j["roofline"]["avg_launch_ms_unshared"],3))
PY
