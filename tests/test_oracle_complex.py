"""Complex-frequency flow slab: the oracle's coefficient functions against the values the
reference's own lambdified functions give (tests/golden/complex_coefficients.json, tools/gen_golden_complex.py), the
NumPy restatement of the kernel algorithm against DOP853, the uniform-flow limit against the closed-form complex
dispersion function, and the Im(omega) = 0 limit against the real flow-slab oracle (pinned to reference traces)."""
import json
import os

import numpy as np
import pytest

from oracle import slab as osl
from oracle.slab_complex import ComplexFlowSlab

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "complex_coefficients.json")))


def cx(v):
    return complex(v[0], v[1])


def test_constants_are_the_reference_ones():
    P = ComplexFlowSlab()
    c = GOLD["constants"]
    assert (P.vA_i, P.c_i, P.vA_e, P.rho_i, P.rho_e, P.U_i0, P.U_e) == (c["vA_i"], c["c_i"], c["vA_e"], c["rho_i"], c["rho_e"], c["U_i0"], c["U_e"])
    assert abs(P.c_e - c["c_e"]) < 1e-15


@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: f'{c["mode"]}-w{c["width"]}-k{c["k"]}')
def test_coefficient_functions_match_reference(case):
    P = ComplexFlowSlab(width=case["width"], mode=case["mode"])
    k, w = case["k"], cx(case["w"])
    m_e, _ = P.exterior_constants(k, w)
    assert abs(m_e - cx(case["m_e"])) <= 1e-14 * abs(m_e)
    for row in case["rows"]:
        _, D, coeff, _, _ = P.interior_coefficients(row["x"], k, w)
        Dr, cr = cx(row["D"]), cx(row["coeff"])
        assert abs(D - Dr) <= 1e-12 * max(abs(Dr), 1e-3), (row["x"], D, Dr)     # uniform cases: D ~ 1e-9 by cancellation
        assert abs(coeff - cr) <= 1e-12 * abs(cr), (row["x"], coeff, cr)


@pytest.mark.parametrize("mode", ["kink", "sausage"])
def test_uniform_limit_is_the_closed_form(mode):
    P = ComplexFlowSlab(mode=mode, width=1e5)
    w = np.array([0.9 + 0.2j, 1.5 - 0.1j, 0.3 + 0.05j, 2.2 + 0.3j, 0.7 + 0.0j, 1.1 + 0.6j])
    for k in (0.4, 1.3, 2.6):
        d, rel, st = P.eval_rk4(k, w)
        dc, relc, stc = P.closed_form_uniform(k, w)
        assert np.array_equal(st, stc)
        ok = st == 0
        assert ok.sum() >= 2
        scale = np.abs(dc[ok]) * 100.0 / relc[ok]
        assert np.max(np.abs(d[ok] - dc[ok]) / scale) < 5e-9        # width 1e5 is uniform to 1e-10 only


def test_rk4_restatement_against_dop853():
    P = ComplexFlowSlab(mode="kink", width=0.9)
    w = np.array([0.9 + 0.2j, 1.5 - 0.1j, 0.3 + 0.05j, 2.2 + 0.3j])
    for k in (0.5, 1.3):
        d, rel, st = P.eval_rk4(k, w)
        dt, relt, stt = P.eval_truth(k, w)
        ok = st == 0
        scale = np.abs(dt[ok]) * 100.0 / relt[ok]
        assert np.max(np.abs(d[ok] - dt[ok]) / scale) < 2e-7        # RK4 at N = 500, one step per interval


def test_real_axis_reduces_to_the_real_oracle():
    """variant "sfg" (D of SF-G:421 as written, no U' term in P_T) at Im(omega) = 0 is the real flow-slab path."""
    eq = osl.SlabEquilibrium("flow", c_i0=1.0, vA_i0=1.0, c_e=0.75, vA_e=0.0, U_i0=0.35, width=1.5)
    n = 0
    for mode in ("kink", "sausage"):
        rp = osl.SlabProblem(eq, mode, L_factor=3.0, ic=(1e-8, 1e-15))
        Pc = ComplexFlowSlab(c_i=eq.c_i0, vA_i=eq.vA_i0, c_e=eq.c_e, vA_e=eq.vA_e, rho_i=eq.rho_i0, rho_e=eq.rho_e,
                             U_i0=0.35, U_e=0.0, width=1.5, mode=mode, variant="sfg")
        for k, w in ((1.2, 0.55), (1.2, 0.8), (2.0, 1.1), (0.6, 0.3)):
            d, Pe, Pi, st = rp.mismatch(k, w)
            dc, rel, stc = Pc.eval_truth(k, np.array([w + 0j]))
            if st != 0:
                continue
            n += 1
            # the real oracle scales the exterior to |V_e(-1)| = 1 keeping its sign; here V_e(-1) = 1
            assert abs(abs(dc[0].real) - abs(d)) < 1e-8 * max(abs(Pe), abs(Pi)) and abs(dc[0].imag) < 1e-12 * abs(d)
    assert n >= 5


def test_unstable_root_of_the_closed_form_is_a_root_of_the_shooting_function():
    """Known answer from physics: a Kelvin-Helmholtz unstable kink mode of the uniform-flow slab (U_i0 = 1.4 vA_i is
    super-critical).  The complex root of the closed-form dispersion function must zero the shooting D_c too."""
    P = ComplexFlowSlab(mode="kink", width=1e5)
    for k, want in ((0.3, 0.14293571535248753 + 0.09060650450374286j), (0.5, 0.3128068480162605 + 0.09428139628771133j)):
        root = closed_form_root(P, k, 0.65 * k + 0.18j * k)
        assert abs(root - want) < 1e-12 and root.imag > 0.05            # growth rate Im(omega) > 0: unstable
        assert P.closed_form_uniform(k, [root])[1][0] < 1e-10
        d, rel, st = P.eval_rk4(k, np.array([root]))
        assert st[0] == 0 and rel[0] < 1e-6                             # the Gaussian of width 1e5 is uniform to 1e-10


def closed_form_root(P, k, guess):
    """Secant iterations on the closed-form complex dispersion function."""
    w0, w1 = guess, guess * (1 + 1e-3)
    f0, f1 = P.closed_form_uniform(k, [w0])[0][0], P.closed_form_uniform(k, [w1])[0][0]
    for _ in range(60):
        w2 = w1 - f1 * (w1 - w0) / (f1 - f0)
        w0, f0, w1 = w1, f1, w2
        f1 = P.closed_form_uniform(k, [w1])[0][0]
        if abs(w1 - w0) < 1e-15:
            break
    return w1


# ---- the reference's own evaluation: SF-X `kink` executed in the build container at Im(omega) = 0 ----------------
SFX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sfx_kink_real_axis.json")))


def sfx_bound(A):
    """LSODA's tolerance on the reference's side: interior error 40 * atol / A for a solution of amplitude A (as for the
    real workers, tests/test_oracle_golden.py) on top of the exterior's own error (started at 1e-8 < atol; 2e-3 of the
    scale covers every traced evaluation of amplitude A > 1e-5)."""
    return max(2e-3, 40 * 1.5e-8 / abs(A))


def test_sfx_kink_worker_trace_at_real_frequencies():
    """tests/golden/sfx_kink_real_axis.json (tools/gen_golden_sfx.py): 48 mismatch values (35 of them with an exterior amplitude above 1e-5, the ones compared) the reference's complex
    worker `kink` (SF-X:737) computed at real omega -- uniform flow as checked in (U_i0 = 1.4) and the Gaussian profile
    (dx = 0.9, U_i0 = 0.2; with U_i0 = 1.4 every evanescent real frequency lies in the flow continuum).  The oracle's
    variant "sfx" (D of SF-X:940, total pressure with the U' term SF-X:955-960) reproduces them to LSODA's tolerance:
    this pins the complex path's FORMULAS to the reference's outputs on the real axis.  Off the real axis SF-X mixes
    real and imaginary parts and `sausage` / `locate_*` raise ValueError: parity unpinned there."""
    n = 0
    worst = 0.0
    for s in SFX["sets"]:
        P = ComplexFlowSlab(width=s["width"], U_i0=s["U_i0"], mode="kink", variant="sfx")
        for e in s["evals"]:
            A = e["ext_value"]
            if e["ier"] != 1 or abs(A) < 1e-5:
                continue
            d, rel, st = P.eval_rk4(s["k"], complex(e["w"], 0.0))
            assert st[0] == 0 and abs(d[0].imag) <= 1e-12 * abs(d[0])
            scale = abs(d[0]) * 100.0 / rel[0]
            err = abs(d[0].real - e["d"] / A) / scale
            worst = max(worst, err)
            assert err < sfx_bound(A), (s["width"], s["k"], e["w"], d[0], e["d"] / A, err)
            assert (d[0].real > 0) == (e["d"] / A > 0) or abs(d[0].real) < sfx_bound(A) * scale
            n += 1
    assert n >= 32 and worst < 5e-3, (n, worst)
