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
for variant in os.environ.get("SHAPES", "4,3;2,3;1,4").split(";"):      # ES_GRID_SHAPE = points per lane, waves per SIMD
    os.environ["ES_GRID_SHAPE"] = variant
    D, st = gp.eval_grid(k, W)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t = time.time()
        D, st = gp.eval_grid(k, W)
        torch.cuda.synchronize()
        ts.append(time.time() - t)
    t = min(ts)
    print(f"shape {variant}: {n}x{n} grid eval {t*1e3:.1f} ms  -> {n*n/t/1e6:.1f} M det-evals/s", flush=True)
del os.environ["ES_GRID_SHAPE"]
D, st = gp.eval_grid(k, W)
torch.cuda.synchronize()
t = time.time()
roots, cnt = gp.find_roots(k, W, D, st, n_bisect=40, capacity=1 << 18)
torch.cuda.synchronize()
t = time.time() - t
acc = int((roots["flag"] == 1).sum())
print(f"find_roots: {cnt} brackets, {acc} accepted, {t*1e3:.1f} ms ({cnt*41/t/1e6:.2f} M evals/s in refine)")
print("status histogram", torch.bincount(st.flatten().to(torch.int64), minlength=4).tolist())

# K1 (closed-form slab) and K2 (closed-form cylinder) and the slab-flow propagator at BASELINE sizes
from eigensolver_amd import SlabSteadyFlow, CylinderUniform  # noqa: E402


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.time()
        fn()
        torch.cuda.synchronize()
        ts.append(time.time() - t)
    return min(ts)


s = SlabSteadyFlow(ctx=ctx)
K = torch.linspace(3.5 / n, 3.5, n, dtype=torch.float64, device="cuda")
Wv = (torch.arange(n, dtype=torch.float64, device="cuda") + 0.5) * (3.0 / n)
for mode in (0, 2):
    t = timeit(lambda: s.disp_rel(mode, Wv, K))
    print(f"K1 slab analytic mode {mode}: {n}x{n} eval {t*1e3:.2f} ms -> {n*n/t/1e9:.2f} G det-evals/s, {n*n*8/t/1e9:.0f} GB/s written")
t = timeit(lambda: s.scan(0, K, Wv, 3.0 / n, capacity=1 << 20))
print(f"K1 scan (2 evals per cell + compaction): {t*1e3:.2f} ms -> {2*n*n/t/1e9:.2f} G det-evals/s")
cu = CylinderUniform(q.CylinderFlow(), "kink", ctx=ctx)
t = timeit(lambda: cu.eval_grid(k, W))
print(f"K2 uniform cylinder closed form: {n}x{n} eval {t*1e3:.2f} ms -> {n*n/t/1e6:.1f} M det-evals/s")
n1 = 1024
sf = ShootProblem(q.SlabFlow(U_i0=0.35, width=1.5), "sausage", ctx=ctx)
k1 = torch.linspace(0.05, 3.5, n1, dtype=torch.float64, device="cuda")
W1 = 1.4 + (torch.arange(n1, dtype=torch.float64, device="cuda") + 0.5) * (2.45 - 1.4) / n1
t = timeit(lambda: sf.eval_grid(k1, W1))
print(f"K3 slab flow (config 2, 1024x1024, N=500): {t*1e3:.2f} ms -> {n1*n1/t/1e6:.1f} M det-evals/s")

# twisted / rotational family (configs[4]): general coefficient set, N = 2000 nodes
n2 = 2048
rot = ShootProblem(q.CylinderRotation(v_twist=0.1, power=1.0), "kink", ctx=ctx)
k2 = torch.linspace(0.25, 4.0, n2, dtype=torch.float64, device="cuda")
W2 = 0.7 + (torch.arange(n2, dtype=torch.float64, device="cuda") + 0.5) * (1.45 - 0.7) / n2
for variant in ("4,2", "2,2", "1,3"):
    os.environ["ES_GRID_SHAPE"] = variant
    t = timeit(lambda: rot.eval_grid(k2, W2), reps=3)
    print(f"K3 rotational (FAM_CYLT, {n2}x{n2}, N=2000) shape {variant}: {t*1e3:.1f} ms -> {n2*n2/t/1e6:.1f} M det-evals/s")
del os.environ["ES_GRID_SHAPE"]

# K6 complex-frequency flow slab: 256 k x (64 x 64) (Re, Im) grid, N = 500, + root search
from eigensolver_amd import SlabComplexFlow  # noqa: E402
from eigensolver_amd.shooting import W_PHASE_SPEED  # noqa: E402

cxs = SlabComplexFlow(width=0.9, ctx=ctx)
kc = np.linspace(0.05, 2.5, 256)
wr, wi = np.linspace(-0.5, 2.5, 64), np.linspace(-0.3, 0.3, 64)
t = timeit(lambda: cxs.eval_grid("kink", kc, wr, wi, W_PHASE_SPEED), reps=3)
print(f"K6 complex flow slab (256 k x 64 x 64, N=500): {t*1e3:.1f} ms -> {256*64*64/t/1e6:.1f} M det-evals/s")
Dc, sc, rc = cxs.eval_grid("kink", kc, wr, wi, W_PHASE_SPEED)
torch.cuda.synchronize()
tt = time.time()
roots, cnt = cxs.find_roots("kink", kc, wr, wi, Dc, sc, W_PHASE_SPEED)
torch.cuda.synchronize()
print(f"K6 root search: {cnt} candidate cells, {int((roots['flag'] == 1).sum())} accepted, {(time.time()-tt)*1e3:.1f} ms")
