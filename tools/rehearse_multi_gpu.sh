#!/bin/bash
# Rehearsal of the N > 1 bench path on a ONE-GPU box (through gpurun): `python bench.py --gpus N` starts its own ranks
# (gloo, all ranks on cuda:0), k-tiles the ONE 4096x4096 grid and gathers the root tables; the merged root table of every
# N must equal the N = 1 table record for record.   Usage: bash tools/rehearse_multi_gpu.sh "2 4"
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export ES_BENCH_BACKEND=gloo ES_BENCH_SHARE_GPU=1
python $R/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline $BENCH_ARGS --dump-roots $R/gpurun_out/roots_n1.npy > $R/gpurun_out/rehearse_n1.json 2> $R/gpurun_out/rehearse_n1.err || exit 1
for N in ${1:-2 4}; do
  python $R/bench.py --gpus $N --steps 2 --warmup 1 --no-cpu-baseline $BENCH_ARGS --dump-roots $R/gpurun_out/roots_n$N.npy > $R/gpurun_out/rehearse_n$N.json 2> $R/gpurun_out/rehearse_n$N.err || { tail -5 $R/gpurun_out/rehearse_n$N.err; exit 2; }
  python - <<PY || exit 3
import json, numpy as np
a = np.load("$R/gpurun_out/roots_n1.npy"); b = np.load("$R/gpurun_out/roots_n$N.npy")
j = json.loads(open("$R/gpurun_out/rehearse_n$N.json").read().strip().splitlines()[-1])
same = a.shape == b.shape and np.array_equal(a, b, equal_nan=True)
print(f"N=$N: {b.shape[0]} gathered records, identical to N=1: {same}; value {j['value']:.3e} {j['unit']}, "
      f"{j['ms_per_step']:.2f} ms/step, scaling {j['scaling']}, rows/GPU {j['config'].get('k_rows_per_gpu', j['config'].get('k_rows_per_gpu_per_unit'))}")
assert same
PY
done
