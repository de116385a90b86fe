# GPU box: whole -m gpu suite, then bench + rocprofv3 (kernel trace, FETCH / WRITE / SQ passes) of the four workloads
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t2.log 2>&1; echo "pytest rc $?"; tail -6 gpurun_out/t2.log
TAG=${1:-r3a}
for w in config3 config1 config2 config4; do
  timeout -k 10 600 bash tools/profile_bench.sh $TAG $w > gpurun_out/profile_${TAG}_$w.log 2>&1; echo "profile $w rc $?"; tail -2 gpurun_out/profile_${TAG}_$w.log | cut -c1-300
done
