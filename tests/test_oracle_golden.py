"""The DOP853 oracle (oracle/cylinder.py, oracle/slab.py) pinned against the reference itself: traces of the
reference workers executed in the build container (tests/golden/trace_*.json, made by tools/gen_golden.py).

Two comparisons per evaluation, both on the amplitude-normalised mismatch d / |y_e(boundary)|:
  * interior only: the oracle is fed the reference's own exterior end state (LSODA), so the difference is the
    interior ODE + shooting + mismatch algebra.  Bound: LSODA's rtol = 1.5e-8 amplified by the (ill-conditioned near
    poles) shooting and its ABSOLUTE atol = 1.5e-8 on an interior solution of size A = |y_e(boundary)| (1e-6..1):
    max(3e-4, 40 * 1.5e-8 / A) of max(|outer|, |inner|), wherever fsolve converged (ier == 1);
  * full: closed-form exterior.  The reference starts LSODA at P = 1e-8 with atol 1.5e-8 (absolute tolerance as
    large as the solution), so its I_m admixture is wrong by O(1) -- an error ~ exp(-2 mu (R - 1)) of the
    log-derivative -- and even where that is negligible its exterior log-derivative is only good to ~1e-4
    (measured: 1.32023 vs 1.32016 at mu = 0.9; up to 2e-3 in the traces).  Bound used: 2e-2 of
    max(|outer|, |inner|) where exp(-2 mu (R - 1)) < 1e-7 -- a sanity check, not a precision claim.
"""
import json
import math
import os

import numpy as np
import pytest

from oracle import cylinder as oc
from oracle import slab as osl

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cyl(kind, fn, **kw):
    eqkw = {k: kw.pop(k) for k in list(kw) if k in ("c_e", "vA_e", "width", "U_i0", "v_twist", "power")}
    eq = oc.CylinderEquilibrium(kind, **eqkw)
    m = 1 if fn == "kink" else 0
    bc = "sausage" if fn == "sausage" else ("rotation_kink" if kind == "rotation" else "kink")
    return oc.CylinderProblem(eq, m, axis_bc=bc, **kw)


def _slab(kind, fn, L, ic, **eqkw):
    return osl.SlabProblem(osl.SlabEquilibrium(kind, **eqkw), fn, L_factor=L, ic=ic)


PHOTO = dict(c_e=1.5, vA_e=0.5)
PROBLEMS = {
    "CF_uniform": lambda fn: _cyl("flow", fn, r_sign=-1, r_axis=1e-3, ic=(1e-8, 1e-8), c1_power=2),
    "CF_flow": lambda fn: _cyl("flow", fn, width=1.0, U_i0=0.6, r_sign=-1, r_axis=1e-3, ic=(1e-8, 1e-8), c1_power=2),
    "CDC_w095": lambda fn: _cyl("density", fn, width=0.95, r_sign=-1, r_axis=1e-3, ic=(1e-8, 1e-15), c1_power=1),
    "CDC_uniform": lambda fn: _cyl("density", fn, width=1e5, r_sign=-1, r_axis=1e-3, ic=(1e-8, 1e-15), c1_power=1),
    "CDP": lambda fn: _cyl("density", fn, width=0.9, r_sign=1, r_axis=1e-3, ic=(1e-8, 1e-8), c1_power=1, **PHOTO),
    "CRSF": lambda fn: _cyl("rotation", fn, v_twist=0.15, power=1.25, r_sign=1, r_axis=1e-2, ic=(1e-8, 1e-8),
                            c1_power=2, **PHOTO),
    "CRKS": lambda fn: _cyl("rotation", fn, v_twist=0.1, power=0.8, r_sign=1, r_axis=1e-3, ic=(1e-8, 1e-8),
                            c1_power=2, **PHOTO),
    "CRKF": lambda fn: _cyl("rotation", fn, v_twist=0.25, power=0.8, r_sign=1, r_axis=1e-3, ic=(1e-8, 1e-8),
                            c1_power=2, **PHOTO),
    "SFU": lambda fn: _slab("uniform_flow", fn, 7.0, (1e-8, 1e-15), c_i0=2.0 / 3.0, vA_i0=1.0, c_e=0.75, vA_e=0.0,
                            U_i0=0.0, U_e=-0.15),
    "SFG_uniform": lambda fn: _slab("flow", fn, 3.0, (1e-8, 1e-15), c_i0=0.3, vA_i0=1.0, c_e=0.2, vA_e=2.5,
                                    U_i0=0.9, U_e=0.0, width=1e5),
    "SFG_flow": lambda fn: _slab("flow", fn, 3.0, (1e-8, 1e-15), c_i0=0.3, vA_i0=1.0, c_e=0.2, vA_e=2.5,
                                 U_i0=0.35, U_e=0.0, width=1.5),
    "SDP_uniform": lambda fn: _slab("density", fn, 7.0, (1e-8, 1e-8), c_i0=1.0, vA_i0=1.9, c_e=1.3, vA_e=0.8,
                                    width=1e5),
    "SDP_w15": lambda fn: _slab("density", fn, 7.0, (1e-8, 1e-8), c_i0=1.0, vA_i0=1.9, c_e=1.3, vA_e=0.8,
                                width=1.5),
    # SD-C as checked in: coronal constants SD-C:72-75, dx = 0.9 (:110), L = 3 (:422), V0 = [1e-8, 1e-8] (:470)
    "SDC_w09": lambda fn: _slab("density", fn, 3.0, (1e-8, 1e-8), c_i0=1.0, vA_i0=1.2, c_e=0.4, vA_e=3.0,
                                width=0.9),
    "SDC_uniform": lambda fn: _slab("density", fn, 3.0, (1e-8, 1e-8), c_i0=1.0, vA_i0=1.2, c_e=0.4, vA_e=3.0,
                                    width=1e5),
    # CR-SS as checked in: v_twist 0.15, power 1.25 (CR-SS:176-177), ix down to r = 0.01 (:157)
    "CRSS": lambda fn: _cyl("rotation", fn, v_twist=0.15, power=1.25, r_sign=1, r_axis=1e-2, ic=(1e-8, 1e-8),
                            c1_power=2, **PHOTO),
}


def _traces():
    return sorted(f[6:-5] for f in os.listdir(G) if f.startswith("trace_") and f.endswith(".json"))


@pytest.mark.parametrize("case", _traces())
def test_oracle_vs_reference_trace(case):
    """`<name>_conv`: the second trace set (tools/gen_golden.py --converge) -- the same calls with the reference's unconverged
    fsolve calls re-solved to convergence on its own objective; there the evaluations whose slope the re-solve converged
    (conv 1 / 5) are compared as well, which the first set has to skip (ier != 1)."""
    tr = json.load(open(os.path.join(G, f"trace_{case}.json")))
    conv_set = case.endswith("_conv")
    case = case[:-5] if conv_set else case
    n_cmp = n_full = 0
    worst_int = 0.0
    for call in tr["calls"]:
        prob = PROBLEMS[case](call["fn"])
        k = call["k"]
        seen = set()
        for ev in call["evals"]:
            w, d = ev["omega"], ev["d"]
            # second set: only what the first set had to skip -- evaluations whose slope the harness re-solved
            converged = ev["ier"] == 1 if not conv_set else (ev.get("conv") in (1, 5) and (ev["ier"] != 1 or ev.get("conv") == 5))
            if w is None or d is None or not math.isfinite(d) or not converged or w in seen:
                continue
            seen.add(w)
            if len(seen) > (8 if conv_set else 14):            # enough per call; DOP853 is slow
                break
            ext = ev["ext_end"]
            if len(ext) == 4:            # odeintz (CR-*): complex state viewed as (re, im) pairs; im = 0
                assert ext[1] == 0.0 and ext[3] == 0.0
                ext = [ext[0], ext[2]]
            A = abs(ext[0])
            d_ref = d / A
            do, a, b, st = prob.mismatch(k, w, ext_override=tuple(ext))
            if st != 0:
                continue
            scale = max(abs(a), abs(b))
            err = abs(do - d_ref) / scale
            worst_int = max(worst_int, err)
            # LSODA runs the interior with atol = 1.5e-8 ABSOLUTE on a solution of size A = |y_e(boundary)|
            tol = max(3e-4, 40 * 1.5e-8 / A)
            if conv_set and abs(a) <= tol * abs(b):
                continue             # next to a pole of D (|outer| << |inner|: the refinement chains of the reference home in on
                                     # them) the mismatch is 1 / (omega - omega_pole): no bound on its relative discrepancy
            assert err < tol, (case, call["fn"], k, w, do, d_ref, A)
            assert (do > 0) == (d_ref > 0) or abs(d_ref) < tol * scale     # determinant sign
            n_cmp += 1
            # full comparison (closed-form exterior) where the far-field admixture is negligible
            df, af, bf, stf = prob.mismatch(k, w)
            m_e = prob.exterior(k, w)[0]
            mu = math.sqrt(m_e)
            R = prob.L_factor * 2 * math.pi / k
            if math.exp(-2 * mu * (R - 1)) < 1e-7 and stf == 0:
                assert abs(df - d_ref) / max(abs(af), abs(bf)) < 2e-2, (case, call["fn"], k, w, df, d_ref)
                n_full += 1
    # CR-KF as checked in: fsolve returns ier = 5 (no convergence, silenced by the reference) at 16 of its 21
    # evaluations, so only the converged ones can be compared
    # SD-C as checked in (dx = 0.9): every band between its `speeds` (SD-C:202) lies inside the Alfven / cusp continuum
    # of the profile (v_A rises from 1.2 to 1.71 towards the boundary) or has fsolve ier = 5: no evaluation is
    # comparable there, the determinant VALUES of that configuration are pinned through SDC_uniform only
    print(f"{case}{'_conv' if conv_set else ''}: {n_cmp} evaluations compared (interior behind the reference's exterior), "
          f"{n_full} with the closed-form exterior, worst interior discrepancy {worst_int:.2e} of the scale")
    if conv_set:
        return                       # the re-solved evaluations only: as many as the reference had left unconverged
    need = {"CRKF": 2, "SDC_w09": 0}.get(case, 6)
    assert n_cmp >= need, (case, n_cmp)
    assert n_full >= 1 or case.startswith(("SDP", "SDC", "SFU", "CRKF")), (case, n_full)
