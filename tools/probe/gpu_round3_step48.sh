set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="--no-cpu-baseline --no-extra-mode --steps 60 --warmup 5"
for E in 8 4 1; do
python bench.py $B --share-of $E > gpurun_out/s48_gate_$E.json 2> gpurun_out/s48_gate_$E.err || exit 1
ES_BENCH_NO_GATE=1 python bench.py $B --share-of $E > gpurun_out/s48_nogate_$E.json 2> gpurun_out/s48_nogate_$E.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s48_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(j["ms_per_step"],3), "%.4e"%j["value"])
PY
