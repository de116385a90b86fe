set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { local label=$1 E=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --share-of $E --no-cpu-baseline --no-extra-mode --steps 40 --warmup 3 > gpurun_out/f_${E}_$label.json 2>/dev/null
  python -c "import json;j=json.loads(open('gpurun_out/f_${E}_$label.json').read().strip().splitlines()[-1]);print('E=$E $label ms/step %.3f launch %.3f alone %.3f' % (j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['avg_launch_ms_unshared']))"
}
for rep in 1 2; do
run gate 1 X=1
run nogate 1 ES_BENCH_NO_GATE=1
run gate 8 X=1
run nogate 8 ES_BENCH_NO_GATE=1
done
