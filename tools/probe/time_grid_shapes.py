"""GPU box: time every launch shape (points per lane, waves per SIMD) of the fp64 grid kernel for each family on the
grids of the BASELINE configs, through ES_GRID_SHAPE (needs the measuring build: ES_BUILD_ALL_SHAPES=1 python -m
eigensolver_amd.build --force).  Writes gpurun_out/grid_shapes.json; the chosen shapes and costs go into ShapeTable
(csrc/es_shoot.hip) and profiles/r3_grid_shapes.json."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from eigensolver_amd import ShootProblem, _lib, equilibrium as q  # noqa: E402


def cases():
    k1 = np.linspace(0.05, 3.5, 1024)
    W1 = 1.4 + (np.arange(1024) + 0.5) * (2.45 - 1.4) / 1024
    k2 = np.linspace(0.01, 4.5, 4096)
    W2 = 2.05 + (np.arange(384) + 0.5) * (4.95 - 2.05) / 384
    k3 = np.linspace(0.01, 4.0, 1024)                        # a quarter of the rows of configs[3]
    W3 = 0.8944271909999159 + (np.arange(4096) + 0.5) * ((5.0 - 0.8944271909999159) / 4096)
    k4 = np.linspace(0.25, 4.0, 1024)
    W4 = 0.7 + (np.arange(1024) + 0.5) * ((1.45 - 0.7) / 1024)
    return [
        ("config1 slab flow kink 1024x1024", q.SlabFlow(U_i0=0.35, width=1.5), "kink", None, k1, W1),
        ("slab density kink 1024x1024", q.SlabDensity(width=1.5, n_nodes=1001), "kink", None, k1, 0.9 + (np.arange(1024) + 0.5) * 0.35 / 1024),
        ("config2 cyl density m=1 4096x384", q.CylinderDensity(width=0.95), "kink", 1, k2, W2),
        ("config3 cyl flow m=1 1024x4096", q.CylinderFlow(U_i0=0.7, width=0.9), "kink", 1, k3, W3),
        ("config4 cyl rotation m=3 1024x1024", q.CylinderRotation(v_twist=0.1, power=1.0, r_axis=0.001), "kink", 3, k4, W4),
        ("config4 cyl rotation m=0 1024x1024", q.CylinderRotation(v_twist=0.1, power=1.0, r_axis=0.01), "sausage", 0, k4, W4),
    ]


def main():
    ctx = _lib.Context(0)
    out = {}
    for name, eq, mode, m, k, W in cases():
        gp = ShootProblem(eq, mode, m=m, ctx=ctx)
        kt = torch.as_tensor(k, device="cuda")
        Wt = torch.as_tensor(W, device="cuda")
        res, ref = {}, None
        for pts in (4, 2, 1):
            for wpe in (2, 3, 4):
                os.environ["ES_GRID_SHAPE"] = f"{pts},{wpe}"
                try:
                    D, st = gp.eval_grid(kt, Wt)
                except Exception as e:                      # shape not built
                    res[f"{pts},{wpe}"] = str(e)
                    continue
                torch.cuda.synchronize()
                ts = []
                for _ in range(4):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    D, st = gp.eval_grid(kt, Wt)
                    b.record()
                    torch.cuda.synchronize()
                    ts.append(a.elapsed_time(b))
                Dn = D.cpu().numpy()
                if ref is None:
                    ref = Dn
                same = bool(np.array_equal(Dn, ref, equal_nan=True))
                res[f"{pts},{wpe}"] = {"ms": float(np.min(ts[1:])), "bit_identical_to_first": same}
                print(name, pts, wpe, res[f"{pts},{wpe}"], flush=True)
        os.environ.pop("ES_GRID_SHAPE", None)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gp.eval_grid(kt, Wt)
        a.record()
        gp.eval_grid(kt, Wt)
        b.record()
        torch.cuda.synchronize()
        res["default"] = a.elapsed_time(b)
        out[name] = {"points": int(len(k) * len(W)), "n_nodes": int(eq.n_nodes), "shapes": res}
        gp.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "grid_shapes.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
