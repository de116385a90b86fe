set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 || exit 1
B="--no-cpu-baseline --no-extra-mode --steps 30 --warmup 5"
for wl in config3 config2 config1; do python bench.py $B --workload $wl > gpurun_out/s45_$wl.json 2> gpurun_out/s45_$wl.err || exit 1; done
python bench.py --no-cpu-baseline --no-extra-mode --steps 12 --warmup 3 --workload config4 > gpurun_out/s45_config4.json 2> gpurun_out/s45_config4.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s45_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(j["ms_per_step"],3), "%.4e"%j["value"], round(j["roofline"]["avg_launch_ms"],3), round(j["roofline"]["avg_launch_ms_unshared"],3))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s45_unit2 -- python3 $R/tools/probe/one_unit_f64.py config2 2 > $R/gpurun_out/s45_unit2.log 2>&1
cd $R
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/s45_unit2/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:3]:
    print("  ", r["Name"][28:80], r["Calls"], "avg %.3f min %.3f"%(float(r["AverageNs"])/1e6, float(r["MinNs"])/1e6))
PY
