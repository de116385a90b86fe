#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: bench line + rocprofv3 kernel stats + HBM PMC passes.
# Usage: bash tools/profile_bench.sh <round-tag>        -> files under gpurun_out/ to be copied into profiles/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-rX}
mkdir -p $R/gpurun_out
python $R/bench.py > $R/gpurun_out/bench_${TAG}.json 2> $R/gpurun_out/bench_${TAG}.err || exit 1
tail -c 2500 $R/gpurun_out/bench_${TAG}.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_${TAG} -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra-mode > $R/gpurun_out/prof_stats_${TAG}.log 2>&1 || exit 2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch_${TAG} -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-mode > $R/gpurun_out/prof_fetch_${TAG}.log 2>&1 || exit 3
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write_${TAG} -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-mode > $R/gpurun_out/prof_write_${TAG}.log 2>&1 || exit 4
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_sq_${TAG} -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-mode > $R/gpurun_out/prof_sq_${TAG}.log 2>&1 || echo "SQ pass failed (non-fatal)"
ls $R/gpurun_out
