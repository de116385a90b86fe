"""The C port (oracle/c/shoot_port.c, fixed-grid RK4 as on the GPU) against the DOP853 oracle for every family.
Bounds the discretisation error of the algorithm the product runs."""
import numpy as np
import pytest

from tests import cases


@pytest.mark.parametrize("name", list(cases.all_cases()))
def test_port_vs_truth(name):
    case = cases.all_cases()[name]
    eq, mode, m, _ = case
    port = cases.port_problem(eq, mode, m)
    truth = cases.truth_problem(eq, mode, m)
    k, W = cases.sample_kw(case, nk=3, nw=6, seed=1)
    kk = np.repeat(k, len(W))
    ww = (k[:, None] * W[None, :]).ravel()
    D, rel, st = port.eval_points(kk, ww, nthreads=4)
    n_ok = n_cont = 0
    for i in range(len(kk)):
        d, a, b, s = truth.mismatch(kk[i], ww[i])
        if s in (1, 2):
            assert st[i] == s, (name, kk[i], ww[i], st[i], s)
            continue
        if s == 3 or st[i] == 3:
            # the oracle samples 4001 points, the port the 2N-1 grid points: they may disagree only at the very
            # edge of a continuum band
            n_cont += 1
            continue
        n_ok += 1
        scale = max(abs(a), abs(b))
        # RK4 on the reference's grid vs adaptive DOP853: 4th-order discretisation error
        tol = 3e-8 * max(1.0, (1000.0 / eq.n_nodes) ** 4)
        assert abs(D[i] - d) <= tol * scale, (name, kk[i], ww[i], D[i], d)
    assert n_ok >= 4, (name, n_ok, n_cont)
