"""GPU box: ns per point-step of the fp64 grid launch of the untwisted cylinder against the row width (4096 rows, N = 2001):
384 frequencies run as 2 points x 192 lanes (three-wave workgroups), 512 as 2 x 256, 1024 as 4 x 256 (four-wave workgroups)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eigensolver_amd import ShootProblem, _lib, equilibrium as q  # noqa: E402

ctx = _lib.Context(0)
gp = ShootProblem(q.CylinderDensity(width=0.95, n_nodes=2001), "kink", 2, ctx=ctx)
k = torch.as_tensor(np.linspace(0.01, 4.5, 4096), device="cuda")
for nw in (256, 384, 512, 768, 1024):
    W = torch.as_tensor(2.05 + (np.arange(nw) + 0.5) * (4.95 - 2.05) / nw, device="cuda")
    for _ in range(2):
        gp.eval_grid(k, W)
    torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); gp.eval_grid(k, W); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    t = min(ts)
    print(nw, gp.grid_kernel_name(nw), f"{t:.3f} ms, {t * 1e6 / (4096 * nw * 2000):.4f} ns per point-step", flush=True)
