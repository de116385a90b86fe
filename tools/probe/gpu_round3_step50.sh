set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( while true; do sleep 60; echo "[progress] $(date +%T)"; done ) &
HB=$!
timeout -k 10 600 python tools/fuzz_grid.py 3000 23 32 > gpurun_out/s50_fuzz_grid.txt 2>&1; echo "grid rc $?"; tail -2 gpurun_out/s50_fuzz_grid.txt
timeout -k 10 500 python tools/fuzz_mixed.py 3000 41 > gpurun_out/s50_fuzz_mixed.txt 2>&1; echo "mixed rc $?"; tail -1 gpurun_out/s50_fuzz_mixed.txt
kill $HB
