set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { # label, workload, env...
  local label=$1 w=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --steps 30 > gpurun_out/e_${w}_$label.json 2>/dev/null
  python -c "import json;j=json.loads(open('gpurun_out/e_${w}_$label.json').read().strip().splitlines()[-1]);print('$w $label ms/step %.3f launch %.3f alone %.3f' % (j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['avg_launch_ms_unshared']))"
}
for rep in 1 2; do
run gate_L2 config2 X=1
run nogate_L2 config2 ES_BENCH_NO_GATE=1
run gate_L1 config2 ES_BENCH_UNIT_LANES=1
run nogate_L1 config2 ES_BENCH_NO_GATE=1 ES_BENCH_UNIT_LANES=1
run gate_L3 config2 ES_BENCH_UNIT_LANES=3
done
for rep in 1 2; do
run gate_L3 config1 X=1
run nogate_L3 config1 ES_BENCH_NO_GATE=1
run gate_L1 config1 ES_BENCH_UNIT_LANES=1
run nogate_L1 config1 ES_BENCH_NO_GATE=1 ES_BENCH_UNIT_LANES=1
done
run gate_L1 config4 X=1
run nogate_L1 config4 ES_BENCH_NO_GATE=1
run gate_L2 config4 ES_BENCH_UNIT_LANES=2
run nogate_L2 config4 ES_BENCH_NO_GATE=1 ES_BENCH_UNIT_LANES=2
