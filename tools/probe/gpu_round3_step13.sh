set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
BENCH_ARGS="" bash tools/rehearse_multi_gpu.sh "2 3" 2>&1 | tail -2
BENCH_ARGS="--workload config4" bash tools/rehearse_multi_gpu.sh "2" 2>&1 | tail -1
BENCH_ARGS="--workload config1" bash tools/rehearse_multi_gpu.sh "2" 2>&1 | tail -1
BENCH_ARGS="--workload config2" bash tools/rehearse_multi_gpu.sh "4" 2>&1 | tail -1
BENCH_ARGS="--mode weak-m" bash tools/rehearse_multi_gpu.sh "2" 2>&1 | tail -2
