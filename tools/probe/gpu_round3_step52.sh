set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="--no-cpu-baseline --no-extra-mode --workload config4 --steps 12 --warmup 3"
python bench.py $B > gpurun_out/s52_dup_a.json 2> gpurun_out/s52_dup_a.err || exit 1
ES_BUILD_EXTRA_FLAGS="-DES_F32_NO_DUP" python -m eigensolver_amd.build --force > gpurun_out/s52_build.log 2>&1 || exit 2
python bench.py $B > gpurun_out/s52_nodup_a.json 2> gpurun_out/s52_nodup_a.err || exit 1
python bench.py $B > gpurun_out/s52_nodup_b.json 2> gpurun_out/s52_nodup_b.err || exit 1
python -m eigensolver_amd.build --force > gpurun_out/s52_build2.log 2>&1 || exit 2
python bench.py $B > gpurun_out/s52_dup_b.json 2> gpurun_out/s52_dup_b.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s52_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(j["ms_per_step"],3), "%.4e"%j["value"], round(j["roofline"]["avg_launch_ms"],3), round(j["roofline"]["avg_launch_ms_unshared"],3))
PY
