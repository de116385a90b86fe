"""Shared test configurations: product-side equilibrium objects and the matching oracle problems."""
import numpy as np

from eigensolver_amd import equilibrium as q
from eigensolver_amd import shooting as s


def desc_dict(d):
    return {f[0]: getattr(d, f[0]) for f in d._fields_}


def port_problem(eq, mode, m=None):
    from oracle.port import PortProblem
    d, p = s.make_desc(eq, mode, m)
    return PortProblem(desc_dict(d), p)


def truth_problem(eq, mode, m=None):
    """The DOP853 oracle object equivalent to the product equilibrium `eq` (built from the same parameters)."""
    from oracle import cylinder as oc, slab as osl
    if isinstance(eq, q._CylinderBase):
        kind = {q.CylinderDensity: "density", q.CylinderFlow: "flow", q.CylinderRotation: "rotation"}[type(eq)]
        kw = dict(c_i0=eq.c_i0, vA_i0=eq.vA_i0, c_e=eq.c_e, vA_e=eq.vA_e, rho_i0=eq.rho_i0)
        if kind == "density":
            kw.update(width=eq.width, r0=eq.r0)
        elif kind == "flow":
            kw.update(width=eq.width, r0=eq.r0, U_i0=eq.U_i0, U_e=eq.U_e)
        else:
            kw.update(v_twist=eq.v_twist, power=eq.power)
        oeq = oc.CylinderEquilibrium(kind, **kw)
        mm = (1 if mode == "kink" else 0) if m is None else m
        bc = "sausage" if mode == "sausage" else ("rotation_kink" if kind == "rotation" else "kink")
        return oc.CylinderProblem(oeq, mm, r_sign=eq.r_sign, r_axis=eq.r_axis, L_factor=eq.L_factor, ic=eq.ic,
                                  c1_power=eq.c1_power, axis_bc=bc)
    if isinstance(eq, q.SlabDensity):
        oeq = osl.SlabEquilibrium("density", c_i0=eq.c_i0, vA_i0=eq.vA_i0, c_e=eq.c_e, vA_e=eq.vA_e,
                                  rho_i0=eq.rho_i0, width=eq.width, x0=eq.x0)
    else:
        import math
        kind = "uniform_flow" if math.isinf(eq.width) else "flow"
        oeq = osl.SlabEquilibrium(kind, c_i0=eq.c_i0, vA_i0=eq.vA_i0, c_e=eq.c_e, vA_e=eq.vA_e, rho_i0=eq.rho_i0,
                                  width=eq.width, x0=eq.x0, U_i0=eq.U_i0, U_e=eq.U_e)
    return osl.SlabProblem(oeq, mode, L_factor=eq.L_factor, ic=eq.ic)


# name -> (equilibrium, mode, m, (W_lo, W_hi) phase-speed window used by the tests)
def all_cases():
    return {
        "CF_uniform_kink": (q.CylinderFlow(), "kink", None, (2.05, 4.95)),
        "CF_flow_kink": (q.CylinderFlow(U_i0=0.6, width=1.0), "kink", None, (2.7, 4.95)),
        "CF_flow_sausage": (q.CylinderFlow(U_i0=0.6, width=1.0), "sausage", None, (2.7, 4.95)),
        "CF_flow_m3": (q.CylinderFlow(U_i0=0.35, width=0.9), "kink", 3, (2.7, 4.95)),
        "CDC_w095_kink": (q.CylinderDensity(width=0.95), "kink", None, (2.05, 4.95)),
        "CDC_w095_sausage": (q.CylinderDensity(width=0.95), "sausage", None, (2.05, 4.95)),
        "CDP_kink": (q.CylinderDensity(width=1.5, c_e=1.5, vA_e=0.5, r_sign=1.0, n_nodes=1000, ic=(1e-8, 1e-8)),
                     "kink", None, (0.52, 1.48)),
        "CR_kink": (q.CylinderRotation(v_twist=0.25, power=0.8), "kink", None, (1.2, 1.45)),
        "CR_sausage": (q.CylinderRotation(v_twist=0.15, power=1.25, r_axis=0.01), "sausage", None, (1.05, 1.4)),
        "SD_w15_sausage": (q.SlabDensity(width=1.5, n_nodes=1001), "sausage", None, (0.9, 1.25)),
        "SD_w15_kink": (q.SlabDensity(width=1.5, n_nodes=1001), "kink", None, (0.9, 1.25)),
        "SFG_flow_sausage": (q.SlabFlow(U_i0=0.35, width=1.5), "sausage", None, (1.4, 2.45)),
        "SFG_flow_kink": (q.SlabFlow(U_i0=0.35, width=1.5), "kink", None, (1.4, 2.45)),
        "SFU_sausage": (q.SlabFlow(c_i0=2.0 / 3.0, vA_i0=1.0, c_e=0.75, vA_e=0.0, U_i0=0.0, U_e=-0.15,
                                   width=float("inf"), L_factor=7.0), "sausage", None, (0.3, 0.6)),
    }


def sample_kw(case, nk=5, nw=24, seed=0):
    rng = np.random.default_rng(seed)
    _, _, _, (lo, hi) = case
    k = np.sort(rng.uniform(0.3, 4.0, nk))
    W = np.sort(rng.uniform(lo, hi, nw))
    return k, W
