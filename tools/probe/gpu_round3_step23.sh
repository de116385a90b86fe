set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/project_scaling.sh > gpurun_out/s23_scaling.txt 2>&1 || { tail -5 gpurun_out/s23_scaling.txt; exit 1; }
cat gpurun_out/s23_scaling.txt
BENCH_ARGS="" bash tools/rehearse_multi_gpu.sh "2 4" 2>&1 | tail -2 &&
BENCH_ARGS="--workload config4" bash tools/rehearse_multi_gpu.sh "2 3" 2>&1 | tail -2 &&
BENCH_ARGS="--workload config1" bash tools/rehearse_multi_gpu.sh "2" 2>&1 | tail -1 &&
BENCH_ARGS="--workload config2" bash tools/rehearse_multi_gpu.sh "3" 2>&1 | tail -1 &&
timeout -k 10 1500 python tools/fuzz_mixed.py 3700 2025 > gpurun_out/s23_fuzz_mixed.txt 2>&1
tail -2 gpurun_out/s23_fuzz_mixed.txt
