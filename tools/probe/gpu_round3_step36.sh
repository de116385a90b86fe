set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_bench.sh r3i config4 > gpurun_out/s36_profile_config4.log 2>&1 || { tail -5 gpurun_out/s36_profile_config4.log; exit 1; }
tail -c 400 gpurun_out/bench_r3i_config4.json; echo
timeout -k 10 1100 python tools/fuzz_grid.py 2000 7 24 > gpurun_out/s36_fuzz_grid.txt 2>&1
tail -3 gpurun_out/s36_fuzz_grid.txt
