"""GPU probe: duration of the grid launch for few workgroups (bench problem, 4096 omega = 4 segments per k-row): tells how
the dispatcher spreads a partly filled round over the CUs (tail of narrow k-tiles in multi-GPU runs)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from eigensolver_amd import ShootProblem, _lib

eq = bench.workload_equilibrium()
k_np, W_np = bench.workload_grid()
ctx = _lib.Context(0)
pr = ShootProblem(eq, "kink", m=1, ctx=ctx)
W = torch.as_tensor(W_np, dtype=torch.float64, device="cuda")
for nk in [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128, 192, 256, 320, 384, 512, 576, 768, 1024]:
    k = torch.as_tensor(k_np[:: max(1, len(k_np) // nk)][:nk], dtype=torch.float64, device="cuda")
    for _ in range(2):
        pr.eval_grid(k, W)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        pr.eval_grid(k, W)
    e1.record()
    torch.cuda.synchronize()
    print(f"nk {nk:5d}  workgroups {4 * nk:5d} = {4 * nk / 768:5.2f} x 768   {e0.elapsed_time(e1) / 5:7.3f} ms", flush=True)

# cold launches: the chip idle before each one (what a synchronising caller sees)
import time
for nk in (512, 4096):
    k = torch.as_tensor(k_np[:: max(1, len(k_np) // nk)][:nk], dtype=torch.float64, device="cuda")
    for idle_ms in (0.0, 1.0, 20.0):
        ts = []
        for _ in range(6):
            torch.cuda.synchronize()
            time.sleep(idle_ms * 1e-3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            pr.eval_grid(k, W)
            e1.record()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            ts.append((e0.elapsed_time(e1), (t1 - t0) * 1e3))
        print(f"nk {nk:5d} idle {idle_ms:5.1f} ms before: events {np.mean([a for a, _ in ts[1:]]):7.3f} ms, host time inside the call {np.mean([b for _, b in ts[1:]]):6.3f} ms", flush=True)
