"""GPU probe: what of the grid launch is NOT the march -- the same 4096 x 4096 launch with 2, 130 and 1000 interior nodes."""
import os, sys, dataclasses
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from eigensolver_amd import ShootProblem, _lib
eq0 = bench.workload_equilibrium()
k_np, W_np = bench.workload_grid()
ctx = _lib.Context(0)
k = torch.as_tensor(k_np, device="cuda"); W = torch.as_tensor(W_np, device="cuda")
for n in (2, 3, 130, 1000):
    eq = dataclasses.replace(eq0, n_nodes=n)
    pr = ShootProblem(eq, "kink", m=1, ctx=ctx)
    for _ in range(2):
        pr.eval_grid(k, W)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        pr.eval_grid(k, W)
    e1.record(); torch.cuda.synchronize()
    print(f"n_nodes {n:5d}: {e0.elapsed_time(e1) / 5:8.3f} ms per launch", flush=True)
    pr.close()
