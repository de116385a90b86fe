"""GPU box: time of the fp64 grid launch against the node count N (same grid): t = a + b N; a = what a point costs before and
after its march (exterior closed form, first staging, boundary algebra, stores), b N = the march."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eigensolver_amd import ShootProblem, _lib, equilibrium as q  # noqa: E402

ctx = _lib.Context(0)
k2 = np.linspace(0.01, 4.5, 4096)
W2 = 2.05 + (np.arange(384) + 0.5) * (4.95 - 2.05) / 384
k1 = np.linspace(0.05, 3.5, 1024)
W1 = 1.4 + (np.arange(1024) + 0.5) * (2.45 - 1.4) / 1024
for name, mk, mode, m, k, W in (("config2 m=2", lambda n: q.CylinderDensity(width=0.95, n_nodes=n), "kink", 2, k2, W2),
                                 ("config2 m=0", lambda n: q.CylinderDensity(width=0.95, n_nodes=n), "sausage", 0, k2, W2),
                                 ("config1 kink", lambda n: q.SlabFlow(U_i0=0.35, width=1.5, n_nodes=n), "kink", None, k1, W1)):
    kt, Wt = torch.as_tensor(k, device="cuda"), torch.as_tensor(W, device="cuda")
    res = []
    for n in (126, 251, 501, 1001, 2001):
        gp = ShootProblem(mk(n), mode, m=m, ctx=ctx)
        D = torch.empty((len(k), len(W)), dtype=torch.float64, device="cuda")
        for _ in range(2):
            gp.eval_grid(kt, Wt)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); gp.eval_grid(kt, Wt); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        res.append((n - 1, min(ts)))
        gp.close()
    x = np.array([r[0] for r in res], float); y = np.array([r[1] for r in res])
    b_, a_ = np.polyfit(x, y, 1)
    print(name, " ".join(f"N={int(n)}:{t:.3f}ms" for n, t in res), f"| fit a = {a_:.3f} ms, b = {b_ * 1e3:.4f} us per step; a / t(500) = {a_ / (a_ + 500 * b_):.2f}")
