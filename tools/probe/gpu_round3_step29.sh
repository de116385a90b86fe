set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
P="python3 $R/tools/probe/one_unit_f64.py config2 2"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s29_rcp4 -- $P > $R/gpurun_out/s29_rcp4.log 2>&1 || { tail -5 $R/gpurun_out/s29_rcp4.log; exit 2; }
cd $R
ES_BUILD_EXTRA_FLAGS="-DES_NO_RCP4" python -m eigensolver_amd.build --force > gpurun_out/s29_build.log 2>&1 || exit 3
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s29_base -- $P > $R/gpurun_out/s29_base.log 2>&1 || exit 2
cd $R
python - <<'PY'
import csv,glob
for tag in ("rcp4","base"):
    f=glob.glob(f"gpurun_out/s29_{tag}/*/*_kernel_stats.csv")[0]
    print(tag)
    for r in list(csv.DictReader(open(f)))[:4]:
        print("  ", r["Name"][28:80], r["Calls"], "total %.2f ms avg %.3f min %.3f"%(float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6, float(r["MinNs"])/1e6))
PY
