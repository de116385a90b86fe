"""Prints, per golden trace, how many reference worker calls the GPU path reproduces root-for-root (GPU box)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_workers_gpu import _solvers, Sink, G  # noqa: E402
from eigensolver_amd import _lib  # noqa: E402

ctx = _lib.Context(0)
tot = [0, 0, 0]
for name, (solver, key) in _solvers(ctx).items():
    tr = json.load(open(os.path.join(G, f"trace_{name}.json")))
    same = calls = roots = 0
    worst = 0.0
    for call in tr["calls"]:
        ws, ks = Sink(), Sink()
        getattr(solver, call["fn"])(call["k"], ws, ks, np.array(call["freq"]))
        mine, ref = ws.items[0], call["roots_w"]
        calls += 1
        if len(mine) == len(ref):
            dw = max([abs(a - b) / abs(b) for a, b in zip(mine, ref)], default=0.0)
            if dw <= 1e-10:
                same += 1
                roots += len(ref)
                worst = max(worst, dw)
    print(f"{name:12s} calls {calls:2d}  identical root lists {same:2d}  roots {roots:3d}  max|dw/w| {worst:.1e}")
    tot[0] += calls; tot[1] += same; tot[2] += roots
print("total", tot)

# larger driver-style sweeps
from tests.test_workers_gpu import ROOTSET_REPORT_ONLY, ROOTSET_SOLVERS  # noqa: E402
for name, (skey, key) in {**ROOTSET_SOLVERS, **ROOTSET_REPORT_ONLY}.items():
    path = os.path.join(G, f"roots_{name}.json")
    if not os.path.exists(path):
        continue
    solver, _ = _solvers(ctx)[skey]
    rs = json.load(open(path))
    calls = same = rref = rmatch = fails = 0
    for c in rs["calls"]:
        fr = np.linspace(c["band"][0] * c["k"], c["band"][1] * c["k"], c["n"])[None, :]
        mine = solver.run_batch(c["fn"], [c["k"]], fr)[0]
        ref = c["roots_w"]
        calls += 1
        fails += c["n_fsolve_fail"] > 0
        rref += len(ref)
        same += len(mine) == len(ref) and all(abs(a - b) <= 1e-10 * abs(b) for a, b in zip(mine, ref))
        rmatch += sum(1 for b in ref if any(abs(a - b) <= 1e-10 * abs(b) for a in mine))
    print(f"sweep {name:12s} calls {calls:3d} identical lists {same:3d}  reference roots {rref:3d} reproduced {rmatch:3d}"
          f"  (calls with fsolve failures in the reference: {fails})")

# the reference's full driver run for the cylinder-flow script (CF:1134-1153): 150 k x 4 bands x 70 points x 2 modes
import time  # noqa: E402
import torch  # noqa: E402
import eigensolver_amd as E  # noqa: E402
s = E.CylinderNonUniformFlow(U_i0=0.6, width=1.0, ctx=ctx)
kk = np.linspace(0.01, 4.0, 150)
s.solve(kk[:4], 70)
torch.cuda.synchronize()
t = time.time()
out = s.solve(kk, 70)
torch.cuda.synchronize()
t = time.time() - t
print(f"driver run CF (150 k x 4 bands x 70 pts x 2 modes = {150*4*70*2} grid evaluations + refinements): {t*1e3:.1f} ms, "
      f"roots sausage {len(out['sausage'][0])} kink {len(out['kink'][0])}")
