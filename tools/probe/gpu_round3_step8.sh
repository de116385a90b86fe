set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in config4 config1 config2; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/d_$w.json 2> gpurun_out/d_$w.err; echo "$w rc $?"; tail -c 300 gpurun_out/d_$w.err
  python -c "import json;j=json.loads(open('gpurun_out/d_$w.json').read().strip().splitlines()[-1]);print('$w', j['ms_per_step'], j['value'], j['roofline']['avg_launch_ms'], j['roofline']['avg_launch_ms_unshared'], j['config']['brackets_per_step'], j['config']['roots_per_step'])"
done
timeout -k 10 300 python bench.py --workload config4 --precision f64 --no-cpu-baseline > gpurun_out/d_config4_f64.json 2> gpurun_out/d_config4_f64.err
python -c "import json;j=json.loads(open('gpurun_out/d_config4_f64.json').read().strip().splitlines()[-1]);print('config4 f64', j['ms_per_step'], j['value'])"
timeout -k 10 600 python -m pytest tests/test_mixed_gpu.py tests/test_abi.py -m gpu -x -q > gpurun_out/t8.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/t8.log
BENCH_ARGS="--workload config4" bash tools/rehearse_multi_gpu.sh "2" 2>&1 | tail -2
BENCH_ARGS="--workload config1" bash tools/rehearse_multi_gpu.sh "3" 2>&1 | tail -2
