set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="--no-cpu-baseline --no-extra-mode"
for L in 1 2; do ES_BENCH_UNIT_LANES=$L python bench.py $B --workload config4 --steps 12 --warmup 3 > gpurun_out/s38_c4_L$L.json 2> gpurun_out/s38_c4_L$L.err || exit 1; done
for L in 2 3 4; do ES_BENCH_UNIT_LANES=$L python bench.py $B --workload config2 --steps 30 --warmup 5 > gpurun_out/s38_c2_L$L.json 2> gpurun_out/s38_c2_L$L.err || exit 1; done
for L in 3 4 6; do ES_BENCH_UNIT_LANES=$L python bench.py $B --workload config1 --steps 40 --warmup 5 > gpurun_out/s38_c1_L$L.json 2> gpurun_out/s38_c1_L$L.err || exit 1; done
for S in 3 4; do python bench.py $B --streams $S --steps 30 --warmup 5 > gpurun_out/s38_c3_S$S.json 2> gpurun_out/s38_c3_S$S.err || exit 1; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s38_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(j["ms_per_step"],3), "%.4e"%j["value"])
PY
