"""The fixed-grid RK4 propagator converges with 4th order to the adaptive DOP853 oracle (CPU port; the GPU runs the
same arithmetic): halving the step divides the error by ~16 until rounding takes over."""
import dataclasses

import numpy as np

from eigensolver_amd import equilibrium as q
from tests import cases


def test_fourth_order_convergence_cylinder_flow():
    base = q.CylinderFlow(U_i0=0.6, width=1.0)
    truth = cases.truth_problem(base, "kink")
    pts = [(1.3, 3.1), (2.7, 4.4), (3.6, 3.9)]
    ref = [truth.mismatch(k, k * W) for k, W in pts]
    errs = []
    for N in (126, 251, 501, 1001):
        eq = dataclasses.replace(base, n_nodes=N)
        port = cases.port_problem(eq, "kink")
        D, rel, st = port.eval_points([k for k, W in pts], [k * W for k, W in pts])
        e = max(abs(D[i] - ref[i][0]) / max(abs(ref[i][1]), abs(ref[i][2])) for i in range(len(pts)))
        errs.append(e)
    # monotone, and 4th order overall (the 1/r behaviour at the axis makes the individual ratios uneven)
    assert all(errs[i] > errs[i + 1] for i in range(len(errs) - 1)), errs
    order = np.log(errs[0] / errs[-1]) / np.log(8.0)
    assert order > 3.5, (errs, order)
    assert errs[-1] < 1e-9


def test_fourth_order_convergence_slab_density():
    base = q.SlabDensity(width=1.5)
    truth = cases.truth_problem(base, "kink")
    pts = [(1.0, 1.05), (2.2, 1.12)]
    ref = [truth.mismatch(k, k * W) for k, W in pts]
    errs = []
    for N in (51, 101, 201, 401):
        eq = dataclasses.replace(base, n_nodes=N)
        port = cases.port_problem(eq, "kink")
        D, rel, st = port.eval_points([k for k, W in pts], [k * W for k, W in pts])
        errs.append(max(abs(D[i] - ref[i][0]) / max(abs(ref[i][1]), abs(ref[i][2])) for i in range(len(pts))))
    ratios = [errs[i] / errs[i + 1] for i in range(len(errs) - 1)]
    assert all(8.0 < r < 32.0 for r in ratios), (errs, ratios)
