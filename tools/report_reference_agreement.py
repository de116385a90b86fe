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
