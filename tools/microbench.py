"""Scratch timing of the grid kernel variants on the GPU box (not the judged bench; see bench.py)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigensolver_amd import ShootProblem, _lib, equilibrium as q  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = _lib.Context(0)
eq = q.CylinderFlow(U_i0=0.6, width=1.0)
gp = ShootProblem(eq, "kink", ctx=ctx)
k = torch.linspace(0.01, 4.0, n, dtype=torch.float64, device="cuda")
W = 2.7 + (torch.arange(n, dtype=torch.float64, device="cuda") + 0.5) * (4.95 - 2.7) / n
for variant in os.environ.get("VARIANTS", "0,1,2").split(","):
    os.environ["ES_GRID_VARIANT"] = variant
    D, st = gp.eval_grid(k, W)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t = time.time()
        D, st = gp.eval_grid(k, W)
        torch.cuda.synchronize()
        ts.append(time.time() - t)
    t = min(ts)
    print(f"variant {variant}: {n}x{n} grid eval {t*1e3:.1f} ms  -> {n*n/t/1e6:.1f} M det-evals/s", flush=True)
del os.environ["ES_GRID_VARIANT"]
D, st = gp.eval_grid(k, W)
torch.cuda.synchronize()
t = time.time()
roots, cnt = gp.find_roots(k, W, D, st, n_bisect=40, capacity=1 << 18)
torch.cuda.synchronize()
t = time.time() - t
acc = int((roots["flag"] == 1).sum())
print(f"find_roots: {cnt} brackets, {acc} accepted, {t*1e3:.1f} ms ({cnt*41/t/1e6:.2f} M evals/s in refine)")
print("status histogram", torch.bincount(st.flatten().to(torch.int64), minlength=4).tolist())
