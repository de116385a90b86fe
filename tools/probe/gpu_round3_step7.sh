# GPU box: section rule of the refinement (17 / 9 / 5) with three pipelined streams, whole grid and the 512-row tile; configs 2 and 4 as well
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for S in 17 9 5; do
  export ES_REFINE_SECTIONS=$S
  for E in 1 8; do
    timeout -k 10 200 python bench.py --share-of $E --no-cpu-baseline --no-extra-mode --steps 40 --warmup 3 > gpurun_out/sec_${S}_$E.json 2>/dev/null
    python -c "import json;j=json.loads(open('gpurun_out/sec_${S}_$E.json').read().strip().splitlines()[-1]);print('sections $S E=$E ms/step %.3f' % j['ms_per_step'])"
  done
  timeout -k 10 200 python bench.py --workload config4 --no-cpu-baseline --steps 10 > gpurun_out/sec_${S}_c4.json 2>/dev/null
  python -c "import json;j=json.loads(open('gpurun_out/sec_${S}_c4.json').read().strip().splitlines()[-1]);print('sections $S config4 ms/step %.3f' % j['ms_per_step'])"
  timeout -k 10 200 python bench.py --workload config2 --no-cpu-baseline --steps 20 > gpurun_out/sec_${S}_c2.json 2>/dev/null
  python -c "import json;j=json.loads(open('gpurun_out/sec_${S}_c2.json').read().strip().splitlines()[-1]);print('sections $S config2 ms/step %.3f' % j['ms_per_step'])"
done
unset ES_REFINE_SECTIONS
