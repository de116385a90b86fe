"""GPU box: status histogram of configs[4] per order and the share of 256-point omega chunks (= one wave of the fp32
screening kernel, 4 points per lane) whose points ALL carry the continuum status."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from eigensolver_amd import ShootProblem, _lib  # noqa: E402

_, units = bench.workload_units(sys.argv[1] if len(sys.argv) > 1 else "config4")
ctx = _lib.Context(0)
for label, uid, eq, mode, m, k, W in units:
    gp = ShootProblem(eq, mode, m=m, ctx=ctx)
    kt, Wt = torch.as_tensor(k, device="cuda"), torch.as_tensor(W, device="cuda")
    D, st = gp.eval_grid(kt, Wt)[:2]
    st = st.cpu().numpy().reshape(len(k), len(W))
    hist = np.bincount(st.ravel(), minlength=6)
    for chunk in (64, 128, 256):
        c = (st.reshape(len(k), -1, chunk) == 2).all(axis=2).mean()
        print(label, "chunk", chunk, "all-continuum share %.3f" % c, end="; ")
    print("hist", hist.tolist(), "continuum share %.3f" % (hist[2] / st.size))
