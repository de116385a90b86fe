"""The closed-form (Bessel) determinant of the uniform cylinder against (a) the DOP853 oracle run with profile
width 1e5, (b) the reference's own uniform-limit traces, (c) the stored uniform-limit roots."""
import json
import math
import os

import numpy as np
import pytest

from oracle import cylinder as oc
from tests import stored_sets as S

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("mode,m", [("kink", 1), ("sausage", 0), ("kink", 3)])
def test_closed_form_vs_dop853(mode, m):
    eq = oc.CylinderEquilibrium("flow")            # uniform: U_i0 = 0, width 1e5
    prob = oc.CylinderProblem(eq, m, axis_bc=mode if mode == "sausage" else "kink")
    n = 0
    for k in (0.4, 1.5, 3.7):
        for W in (0.93, 1.3, 1.9, 2.4, 3.3, 4.6):      # body (m_i < 0) and surface (m_i > 0) ranges
            w = k * W
            d1, a1, b1, s1 = oc.uniform_closed_form(eq, k, w, m, axis_bc=mode)
            d2, a2, b2, s2 = prob.mismatch(k, w)
            if s1 != 0 or s2 != 0:
                continue
            n += 1
            assert abs(d1 - d2) <= 2e-8 * max(abs(a2), abs(b2)), (mode, m, k, W, d1, d2)
    assert n >= 12


@pytest.mark.parametrize("case,ic", [("CF_uniform", (1e-8, 1e-8)), ("CDC_uniform", (1e-8, 1e-15))])
def test_closed_form_vs_reference_uniform_traces(case, ic):
    tr = json.load(open(os.path.join(G, f"trace_{case}.json")))
    eq = oc.CylinderEquilibrium("flow")
    n = 0
    for call in tr["calls"]:
        mode = call["fn"]
        m = 1 if mode == "kink" else 0
        for ev in call["evals"][:14]:
            if ev["d"] is None or ev["ier"] != 1:
                continue
            A = abs(ev["ext_end"][0])
            k, w = call["k"], ev["omega"]
            d, a, b, st = oc.uniform_closed_form(eq, k, w, m, ic=ic, axis_bc=mode)
            prob = oc.CylinderProblem(eq, m, ic=ic)
            mu = math.sqrt(prob.exterior(k, w)[0])
            if st != 0 or math.exp(-2 * mu * (3 * 2 * math.pi / k - 1)) > 1e-7:
                continue
            n += 1
            # bound by the reference's LSODA exterior error (see tests/test_oracle_golden.py)
            assert abs(d - ev["d"] / A) <= 2e-2 * max(abs(a), abs(b)), (case, mode, k, w, d, ev["d"] / A)
    assert n >= 8


@pytest.mark.parametrize("tag", ["cyl_density_coronal_w1e5", "cyl_flow_coronal_noflow"])
def test_closed_form_accepts_stored_uniform_roots(tag):
    eq_p, tol, fmin = S.SETS[tag]
    eq = oc.CylinderEquilibrium("flow")
    for mode, w, k in S.pairs(tag):
        m = 1 if mode == "kink" else 0
        ok = 0
        for kk, ww in zip(k, w):
            d, a, b, st = oc.uniform_closed_form(eq, kk, ww, m, ic=eq_p.ic, axis_bc=mode)
            if st == 0 and abs(d) * 100 / max(abs(a), abs(b)) < tol:
                ok += 1
        frac = ok / len(k)
        assert frac >= (fmin[0] if mode == "sausage" else fmin[1]) - 0.02, (tag, mode, frac)
