"""GPU timing aid: fp32-screened grid search (es_shoot_find_roots_mixed) against the fp64 path on the bench workload
(Cylinder / Gaussian axial flow, 4096 x 4096) and on one order of BASELINE configs[4]; checks that the root tables are
bit-identical.  Run on the GPU box:  python tools/time_mixed_vs_f64.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigensolver_amd import ShootProblem, _lib, equilibrium as q  # noqa: E402


def run(name, eq, mode, m, k, W, reps=5):
    ctx = _lib.Context(0)
    gp = ShootProblem(eq, mode, m=m, ctx=ctx)
    table = gp.alloc_root_table(1 << 18)

    def f64():
        D, st = gp.eval_grid(k, W)
        return gp.find_roots(k, W, D, st, n_bisect=16, tol_percent=1e-3, table=table)

    def mixed():
        return gp.find_roots_mixed(k, W, n_bisect=16, tol_percent=1e-3, table=table)

    out = {}
    for label, fn in (("f64", f64), ("mixed", mixed)):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        out[label] = ((time.perf_counter() - t) / reps * 1e3, {a: v.clone() for a, v in r[0].items()}, r[1], r[4] if len(r) > 2 else None)
    same = out["f64"][2] == out["mixed"][2] and all(
        np.array_equal(out["f64"][1][a].cpu().numpy(), out["mixed"][1][a].cpu().numpy(), equal_nan=True) for a in out["f64"][1])
    print(f"{name}: f64 {out['f64'][0]:.2f} ms, mixed {out['mixed'][0]:.2f} ms, brackets {out['f64'][2]}, "
          f"stats {out['mixed'][3]}, root tables identical: {same}")
    gp.close()


if __name__ == "__main__":
    k = np.linspace(0.01, 4.0, 4096)
    W = 0.8944271909999159 + (np.arange(4096) + 0.5) * ((5.0 - 0.8944271909999159) / 4096)
    run("bench workload (untwisted cylinder, 4096^2, N = 1000)", q.CylinderFlow(U_i0=0.7, width=0.9), "kink", 1, k, W)
    k4 = np.linspace(0.25, 4.0, 1024)
    W4 = 0.7 + (np.arange(1024) + 0.5) * ((1.45 - 0.7) / 1024)
    run("configs[4], m = 3 (twisted cylinder, 1024^2, N = 2000)", q.CylinderRotation(v_twist=0.1, power=1.0), "kink", 3, k4, W4)
