"""GPU box: ONE order of configs[4] (m = 3, 1024 x 1024, N = 2000), mixed search five times on one stream -- under
rocprofv3 --kernel-trace --stats this gives the exclusive duration of every kernel of the pipeline."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from eigensolver_amd import ShootProblem, _lib  # noqa: E402

_, units = bench.workload_units("config4")
label, uid, eq, mode, m, k, W = units[3]
ctx = _lib.Context(0)
gp = ShootProblem(eq, mode, m=m, ctx=ctx)
kt, Wt = torch.as_tensor(k, device="cuda"), torch.as_tensor(W, device="cuda")
tab = gp.alloc_root_table(1 << 14)
for _ in range(6):
    r, n, D, st, stats = gp.find_roots_mixed(kt, Wt, n_bisect=bench.N_BISECT, tol_percent=bench.TOL_PERCENT, table=tab)
torch.cuda.synchronize()
print(n, stats)
