"""Seeded fuzz on the CPU: the C port (the algorithm the GPU runs) against the adaptive DOP853 oracle over random
equilibria of the cylinder families (both radial sign conventions, orders up to 8, both initial-value conventions)."""
import numpy as np
import pytest

from eigensolver_amd import equilibrium as q
from tests import cases


@pytest.mark.parametrize("seed", range(4))
def test_port_vs_truth_random_cylinders(seed):
    rng = np.random.default_rng(500 + seed)
    n = 0
    for _ in range(3):
        ic = [(1e-8, 1e-8), (1e-8, 1e-15)][rng.integers(0, 2)]
        kind = rng.integers(0, 3)
        common = dict(r_sign=float(rng.choice([-1.0, 1.0])), n_nodes=1000, ic=ic)
        if kind == 0:
            eq = q.CylinderFlow(U_i0=rng.uniform(-0.5, 0.8), width=rng.uniform(0.7, 2.5), **common)
        elif kind == 1:
            eq = q.CylinderDensity(width=rng.uniform(0.8, 2.5), c1_power=int(rng.choice([1, 2])), **common)
        else:
            eq = q.CylinderRotation(v_twist=rng.uniform(0.05, 0.25), power=rng.uniform(0.8, 1.3), n_nodes=2000, ic=ic)
        mode, m = ("sausage", 0) if rng.random() < 0.3 else ("kink", int(rng.integers(1, 9)))
        port = cases.port_problem(eq, mode, m)
        truth = cases.truth_problem(eq, mode, m)
        lo, hi = (0.6, 1.45) if kind == 2 else (2.2, 4.9)
        for _ in range(5):
            k, W = rng.uniform(0.3, 4.0), rng.uniform(lo, hi)
            D, rel, st = port.eval_points([k], [k * W])
            d, a, b, s = truth.mismatch(k, k * W)
            if s != 0 or st[0] != 0:
                assert s == st[0] or 3 in (s, st[0])
                continue
            n += 1
            assert abs(D[0] - d) <= 5e-8 * max(abs(a), abs(b)), (type(eq).__name__, mode, m, k, W, D[0], d)
    assert n >= 6
