"""a9 / a10: the closed-form equilibrium profiles and characteristic-speed lists (product: eigensolver_amd/
equilibrium.py + solvers.py; oracle: oracle/cylinder.py, oracle/slab.py) against values of the reference's own
sympy/lambdify functions and `speeds` lists (tests/golden/equilibria.json, tools/gen_golden.py)."""
import json
import os

import numpy as np
import pytest

from eigensolver_amd import equilibrium as q
from oracle import cylinder as oc
from oracle import slab as osl

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "equilibria.json")))
PHOTO = dict(c_e=1.5, vA_e=0.5)

PRODUCT = {
    "CD-C": q.CylinderDensity(width=0.95), "CD-C_w15": q.CylinderDensity(width=1.5),
    "CD-P": q.CylinderDensity(width=0.9, r_sign=1.0, **PHOTO), "CF": q.CylinderFlow(),
    "CF_flow": q.CylinderFlow(U_i0=0.6, width=1.0),
    "CR-KF": q.CylinderRotation(v_twist=0.25, power=0.8), "CR-KS": q.CylinderRotation(v_twist=0.1, power=0.8),
    "CR-SF": q.CylinderRotation(v_twist=0.15, power=1.25), "CR-SS": q.CylinderRotation(v_twist=0.15, power=1.25),
    "SD-P_w15": q.SlabDensity(width=1.5), "SD-C": q.SlabDensity(width=0.9, vA_i0=1.2, vA_e=3.0, c_e=0.4),
    "SF-G_flow": q.SlabFlow(U_i0=0.35, width=1.5),
}
ORACLE = {
    "CD-C": oc.CylinderEquilibrium("density", width=0.95), "CD-C_w15": oc.CylinderEquilibrium("density", width=1.5),
    "CD-P": oc.CylinderEquilibrium("density", width=0.9, **PHOTO), "CF": oc.CylinderEquilibrium("flow"),
    "CF_flow": oc.CylinderEquilibrium("flow", U_i0=0.6, width=1.0),
    "CR-KF": oc.CylinderEquilibrium("rotation", v_twist=0.25, power=0.8, **PHOTO),
    "CR-KS": oc.CylinderEquilibrium("rotation", v_twist=0.1, power=0.8, **PHOTO),
    "CR-SF": oc.CylinderEquilibrium("rotation", v_twist=0.15, power=1.25, **PHOTO),
    "CR-SS": oc.CylinderEquilibrium("rotation", v_twist=0.15, power=1.25, **PHOTO),
    "SD-P_w15": osl.SlabEquilibrium("density", c_i0=1.0, vA_i0=1.9, c_e=1.3, vA_e=0.8, width=1.5),
    "SD-C": osl.SlabEquilibrium("density", c_i0=1.0, vA_i0=1.2, c_e=0.4, vA_e=3.0, width=0.9),
    "SF-G_flow": osl.SlabEquilibrium("flow", c_i0=0.3, vA_i0=1.0, c_e=0.2, vA_e=2.5, U_i0=0.35, width=1.5),
}
RTOL = 2e-14


def _close(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.allclose(a, b, rtol=RTOL, atol=1e-15)


@pytest.mark.parametrize("tag", list(G))
def test_profiles_match_reference_lambdified_functions(tag):
    g, pe, oe = G[tag], PRODUCT[tag], ORACLE[tag]
    x = np.array(g["points"])
    assert abs(pe.rho_e - g["rho_e"]) <= RTOL * g["rho_e"] and abs(oe.rho_e - g["rho_e"]) <= RTOL * g["rho_e"]
    assert abs(pe.cT_e - g["cT_e"]) <= RTOL and abs(oe.cT_e - g["cT_e"]) <= RTOL
    if "c_kink" in g:
        assert abs(pe.c_kink - g["c_kink"]) <= RTOL * g["c_kink"]
    if tag.startswith("C"):
        assert _close(pe.rho(x), g["rho_i_np"]) and _close(oe.rho(x), g["rho_i_np"])
        assert _close(np.sqrt(pe.c2(x)), g["c_i_np"]) and _close(np.sqrt(oe.c2(x)), g["c_i_np"])
        assert _close(pe.vA(x), g["vA_i_np"]) and _close(oe.vA(x), g["vA_i_np"])
        if "v_iphi_np" in g:
            assert _close(pe.v_phi(x), g["v_iphi_np"]) and _close(oe.v_phi(x), g["v_iphi_np"])
        if "v_z" in g:
            assert _close(pe.v_z(x), g["v_z"]) and _close(oe.v_z(x), g["v_z"])
        if "B_i_np" in g:
            assert _close(pe.B_z(x), g["B_i_np"])
    elif tag.startswith("SD"):
        # product samples its own grid: compare through an interpolation-free re-evaluation at the golden points
        rho = pe.rho_e + (pe.rho_i0 - pe.rho_e) * np.exp(-(x - pe.x0) ** 2 / pe.width ** 2)
        assert _close(rho, g["rho_i_np"]) and _close(oe.rho(x), g["rho_i_np"])
        assert _close(np.sqrt(oe.c2(x)), g["c_i_np"]) and _close(np.sqrt(oe.vA2(x)), g["vA_i_np"])
        pe2 = q.SlabDensity(**{**pe.__dict__, "n_nodes": 2, "x_boundary": -1.0, "x_end": 1.0})
        pr = pe2.profiles()                                     # points -1, 0, +1
        ref = oe.c2(np.array([-1.0, 0.0, 1.0]))
        assert _close(pr["c2"], ref) and _close(pr["vA2"], oe.vA2(np.array([-1.0, 0.0, 1.0])))
    else:
        assert _close(oe.U(x), g["U_i_np"]) and _close(oe.dU(x), g["dU_i_np"]) and _close(oe.ddU(x), g["ddU_i_np"])
        pe2 = q.SlabFlow(**{**pe.__dict__, "n_nodes": 2})
        pr = pe2.profiles()
        xs = np.array([-1.0, 0.0, 1.0])
        assert _close(pr["U"], oe.U(xs)) and _close(pr["dU"], oe.dU(xs)) and _close(pr["ddU"], oe.ddU(xs))


def test_speeds_lists_match_reference(monkeypatch):
    """a9: the characteristic-speed lists the band builder uses, per script (including CD-C's missing comma)."""
    import eigensolver_amd.solvers as S

    class NoGPU(S._WorkerSolver):
        def __init__(self):        # solvers only need `eq` for speeds(); skip the GPU context
            pass
    checks = {
        "CD-C": (S.CylinderNonUniformDensity, dict(width=0.95)),
        "CD-P": (S.CylinderNonUniformDensity, dict(width=0.9, photospheric=True)),
        "CF": (S.CylinderNonUniformFlow, dict()),
        "CR-KF": (S.CylinderRotationalFlow, dict(v_twist=0.25, power=0.8, variant="kink_fast")),
        "CR-KS": (S.CylinderRotationalFlow, dict(v_twist=0.1, power=0.8, variant="kink_slow")),
        "CR-SF": (S.CylinderRotationalFlow, dict(v_twist=0.15, power=1.25, variant="sausage")),
        "CR-SS": (S.CylinderRotationalFlow, dict(v_twist=0.15, power=1.25, variant="sausage_slow")),
        "SF-G_flow": (S.SlabNonUniformFlow, dict(U_i0=0.35, width=1.5)),
    }
    monkeypatch.setattr(S._WorkerSolver, "__init__", lambda self, eq, ctx=None: setattr(self, "eq", eq))
    for tag, (cls, kw) in checks.items():
        s = cls(**kw)
        mine = sorted(s.speeds())
        ref = sorted(G[tag]["speeds"])
        assert len(mine) == len(ref), tag
        assert np.allclose(mine, ref, rtol=1e-14, atol=1e-15), (tag, mine, ref)
        bands = s.bands(2.0, 5)
        assert len(bands) == len(ref) - 1 and bands[0][0] == ref[0] * 2.0 and bands[-1][-1] == ref[-1] * 2.0
