#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: rocprofv3 kernel stats + HBM / SQ PMC passes of ONE workload of
# bench.py, then its bench line.
# Usage: bash tools/profile_bench.sh <round-tag> [config3|config1|config2|config4]  -> files under gpurun_out/, turned into
#        profiles/<tag>[_<workload>]_* by python tools/summarize_profiles.py <tag> [workload]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-rX}
WL=${2:-config3}
SUF=""; [ "$WL" != "config3" ] && SUF="_$WL"
ARGS="--workload $WL"
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
P="python3 $R/bench.py $ARGS --warmup 1 --no-cpu-baseline --no-extra-mode"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_${TAG}${SUF} -- $P --steps 5 > $R/gpurun_out/prof_stats_${TAG}${SUF}.log 2>&1 || exit 2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch_${TAG}${SUF} -- $P --steps 2 > $R/gpurun_out/prof_fetch_${TAG}${SUF}.log 2>&1 || exit 3
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write_${TAG}${SUF} -- $P --steps 2 > $R/gpurun_out/prof_write_${TAG}${SUF}.log 2>&1 || exit 4
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_sq_${TAG}${SUF} -- $P --steps 2 > $R/gpurun_out/prof_sq_${TAG}${SUF}.log 2>&1 || echo "SQ pass failed (non-fatal)"
# the PMC summary of THIS box first (profiles/pmc_latest_<workload>.json), then the bench line that quotes it: traffic,
# valu_issue and the launch times of one line come from one box and one build
cd $R && python $R/tools/summarize_profiles.py $TAG $WL > /dev/null || exit 5
python $R/bench.py $ARGS > $R/gpurun_out/bench_${TAG}${SUF}.json 2> $R/gpurun_out/bench_${TAG}${SUF}.err || { tail -5 $R/gpurun_out/bench_${TAG}${SUF}.err; exit 1; }
tail -c 1500 $R/gpurun_out/bench_${TAG}${SUF}.json
ls $R/gpurun_out | grep "${TAG}${SUF}" | head -20
