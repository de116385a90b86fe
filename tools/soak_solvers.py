"""Reference-size driver runs of every per-geometry solver on the GPU box (robustness + timing; not a test)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eigensolver_amd as E  # noqa: E402
from eigensolver_amd import _lib  # noqa: E402

ctx = _lib.Context(0)
runs = [
    ("CylinderNonUniformFlow  (CF:1134-1153, 150 k x 70)", lambda: E.CylinderNonUniformFlow(U_i0=0.6, width=1.0, ctx=ctx), np.linspace(0.01, 4.0, 150), 70),
    ("CylinderNonUniformDensity (CD-C:1126-1153, 90 k x 90)", lambda: E.CylinderNonUniformDensity(width=0.95, ctx=ctx), np.linspace(0.01, 4.5, 90), 90),
    ("CylinderNonUniformDensity photospheric (100 k x 60)", lambda: E.CylinderNonUniformDensity(width=0.9, photospheric=True, ctx=ctx), np.linspace(0.01, 4.5, 100), 60),
    ("CylinderRotationalFlow kink_fast (CR-KF:743, 20 k x 50)", lambda: E.CylinderRotationalFlow(variant="kink_fast", ctx=ctx), np.linspace(0.25, 0.37, 20), 50),
    ("CylinderRotationalFlow sausage (CR-SF:748, 110 k x 50)", lambda: E.CylinderRotationalFlow(v_twist=0.15, power=1.25, variant="sausage", ctx=ctx), np.linspace(0.75, 4.0, 110), 50),
    ("SlabNonUniformFlow (SF-G:758, 100 k x 60)", lambda: E.SlabNonUniformFlow(U_i0=0.35, width=1.5, ctx=ctx), np.linspace(0.01, 4.5, 100), 60),
    ("SlabNonUniformDensity (SD-P, 25 k x 100)", lambda: E.SlabNonUniformDensity(width=1.5, ctx=ctx), np.linspace(0.01, 3.5, 25), 100),
    ("SlabUniformFlow (SF-U:812-845, 350 k, logspace 80 + body 100, p_tol 1e-6)", lambda: E.SlabUniformFlow(ctx=ctx), np.linspace(0.01, 3.5, 350), None),
]
for name, make, ks, n in runs:
    s = make()
    ts = []
    for _ in range(2):                 # the first call of a family also loads its kernels; report the second
        torch.cuda.synchronize()
        t = time.time()
        out = s.solve(ks, n) if n is not None else s.solve(ks)
        torch.cuda.synchronize()
        ts.append(time.time() - t)
    print(f"{name}: {ts[1]*1e3:8.1f} ms (first call {ts[0]*1e3:.1f} ms)   roots "
          + ", ".join(f"{m} {len(v[0])}" for m, v in out.items()), flush=True)
    s.close()
