set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( while true; do sleep 60; echo "[progress] $(date +%T)"; done ) &
HB=$!
timeout -k 10 500 python tools/full_size_parity.py configs12 16 > gpurun_out/s39_fsp_configs12.txt 2>&1; echo "configs12 rc $?"; tail -3 gpurun_out/s39_fsp_configs12.txt
timeout -k 10 900 python tools/full_size_parity.py config4 16 > gpurun_out/s39_fsp_config4.txt 2>&1; echo "config4 rc $?"; tail -3 gpurun_out/s39_fsp_config4.txt
kill $HB
