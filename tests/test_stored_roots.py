"""Oracle (C port) against the roots the reference authors stored with their code: known-answer test of the whole
physics chain (equilibrium, coefficient set, exterior, axis condition, mismatch) for every geometry family."""
import numpy as np
import pytest

from tests import cases, stored_sets as S


@pytest.mark.parametrize("tag", list(S.SETS))
def test_port_accepts_stored_roots(tag):
    eq, tol, fmin = S.SETS[tag]
    for mode, w, k in S.pairs(tag):
        port = cases.port_problem(eq, mode)
        D, rel, st = port.eval_points(k, w, nthreads=8)
        frac = float(np.mean(rel < tol))
        need = fmin[0] if mode == "sausage" else fmin[1]
        assert frac >= need, (tag, mode, frac, need)
