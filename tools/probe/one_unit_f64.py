"""GPU box: ONE unit of a workload (default config2, m = 2), fp64 search six times on one stream -- under rocprofv3
--kernel-trace --stats this gives the exclusive duration of every kernel of the pipeline."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from eigensolver_amd import ShootProblem, _lib  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
idx = int(sys.argv[2]) if len(sys.argv) > 2 else 2
_, units = bench.workload_units(wl)
label, uid, eq, mode, m, k, W = units[idx]
ctx = _lib.Context(0)
gp = ShootProblem(eq, mode, m=m, ctx=ctx)
kt, Wt = torch.as_tensor(k, device="cuda"), torch.as_tensor(W, device="cuda")
tab = gp.alloc_root_table(1 << 15)
for _ in range(6):
    D, st = gp.eval_grid(kt, Wt)
    out = gp.find_roots(kt, Wt, D, st, n_bisect=bench.N_BISECT, tol_percent=bench.TOL_PERCENT, table=tab)
torch.cuda.synchronize()
print(label, out[1] if len(out) > 1 else None)
