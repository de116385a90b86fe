# GPU box: -m gpu suite, config4 / config1 / config2 lines, strong-scaling projection of the headline (rank 0's tile of an
# E-GPU run on one GPU) with wave-shared and per-lane node entries in the refinement
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t3.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/t3.log
for w in config4 config1 config2; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/c_$w.json 2> gpurun_out/c_$w.err; echo "$w rc $?"
  python -c "import json;j=json.loads(open('gpurun_out/c_$w.json').read().strip().splitlines()[-1]);print('$w', j['ms_per_step'], j['value'], j['roofline']['avg_launch_ms'], j['roofline']['avg_launch_ms_unshared'])"
done
timeout -k 10 300 python bench.py --workload config4 --precision f64 --no-cpu-baseline > gpurun_out/c_config4_f64.json 2> gpurun_out/c_config4_f64.err
python -c "import json;j=json.loads(open('gpurun_out/c_config4_f64.json').read().strip().splitlines()[-1]);print('config4 f64', j['ms_per_step'], j['value'], j['roofline']['avg_launch_ms'], j['roofline']['avg_launch_ms_unshared'])"
for E in 1 2 4 8; do
  for V in shared private; do
    if [ $V = private ]; then export ES_REFINE_PRIVATE_ENTRIES=1; else unset ES_REFINE_PRIVATE_ENTRIES; export ES_REFINE_SHARED_MIN=0; fi
    timeout -k 10 200 python bench.py --share-of $E --no-cpu-baseline --no-extra-mode --steps 40 --warmup 3 > gpurun_out/share_${V}_$E.json 2> gpurun_out/share_${V}_$E.err || { echo "share $E $V failed"; tail -3 gpurun_out/share_${V}_$E.err; }
    python -c "import json;j=json.loads(open('gpurun_out/share_${V}_$E.json').read().strip().splitlines()[-1]);print('E=$E $V ms/step %.3f grid alone %.3f brackets %d' % (j['ms_per_step'], j['roofline']['avg_launch_ms_unshared'], j['config']['brackets_per_step']))"
  done
done
unset ES_REFINE_PRIVATE_ENTRIES ES_REFINE_SHARED_MIN
