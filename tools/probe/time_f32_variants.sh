cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 3; do
  ES_F32_VARIANT=$v rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_f32v$v -- python3 $GRAFT_REPO_ROOT/tools/time_mixed_vs_f64.py > $GRAFT_REPO_ROOT/gpurun_out/prof_f32v$v.log 2>&1
done
echo done
