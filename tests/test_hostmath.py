"""Device math headers compiled for the host (tests/hostmath/hostmath.cpp, test-only harness) against
scipy.special: the fp64 Bessel routines of eigensolver_amd/csrc/es_bessel.hpp (scaled I/K, J/Y)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
from scipy import special as sp

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hm():
    src = os.path.join(HERE, "hostmath", "hostmath.cpp")
    so = os.path.join(HERE, "hostmath", "libhostmath.so")
    hdr = os.path.join(HERE, "..", "eigensolver_amd", "csrc", "es_bessel.hpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, src], check=True)
    lib = ctypes.CDLL(so)
    for f in (lib.hm_ke_pair, lib.hm_ie_pair, lib.hm_ie_pair_from_k, lib.hm_jy_pair):
        f.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
    return lib


def test_scaled_K(hm):
    out = (ctypes.c_double * 2)()
    xs = np.concatenate([np.logspace(-6, np.log10(2), 120), np.linspace(2.0000001, 3, 30), np.logspace(np.log10(3), np.log10(700), 120)])
    for n in (0, 1, 2, 5, 11):
        for x in xs:
            hm.hm_ke_pair(n, x, out)
            r0, r1 = sp.kve(n, x), sp.kve(n + 1, x)
            if np.isfinite(r1):
                assert abs(out[0] / r0 - 1) < 4e-15 and abs(out[1] / r1 - 1) < 4e-15, (n, x)


def test_scaled_I(hm):
    out = (ctypes.c_double * 2)()
    for n in (0, 1, 2, 5, 11):
        for x in np.concatenate([np.logspace(-3, 0, 40), np.linspace(1, 45, 150)]):
            hm.hm_ie_pair(n, x, out)
            r0, r1 = sp.ive(n, x), sp.ive(n + 1, x)
            assert abs(out[0] / r0 - 1) < 5e-14 and abs(out[1] / r1 - 1) < 5e-14, (n, x)


def test_scaled_I_from_K(hm):
    """Miller ratio + Wronskian normalisation (the form the exterior solution uses): as accurate as the series."""
    out = (ctypes.c_double * 2)()
    worst = 0.0
    for n in (0, 1, 2, 5, 11, 40):
        for x in np.concatenate([np.logspace(-3, np.log10(0.5), 30), np.linspace(0.5, 80, 400), np.logspace(np.log10(80), np.log10(700), 40)]):
            if n > 11 and x < 0.5:
                continue                               # series fallback: its (x/2)^n/n! prefactor loses ~n ulp
            hm.hm_ie_pair_from_k(n, x, out)
            r0, r1 = sp.ive(n, x), sp.ive(n + 1, x)
            tol = 5e-14 if x < 0.5 else (6e-15 if n <= 11 else 1e-13)   # series fallback below 0.5; K_40 by 40 upward recurrences
            if x >= 0.5 and n <= 11:
                worst = max(worst, abs(out[0] / r0 - 1), abs(out[1] / r1 - 1))
            assert abs(out[0] / r0 - 1) < tol and abs(out[1] / r1 - 1) < tol, (n, x, out[0] / r0 - 1)
    assert worst < 6e-15


def test_J_and_Y(hm):
    out = (ctypes.c_double * 4)()
    for n in (0, 1, 2, 5, 11):
        for x in np.concatenate([np.logspace(-5, 0, 30), np.linspace(1, 80, 300)]):
            hm.hm_jy_pair(n, x, out)
            rj0, rj1, ry0, ry1 = sp.jv(n, x), sp.jv(n + 1, x), sp.yv(n, x), sp.yv(n + 1, x)
            e0, e1 = np.hypot(rj0, ry0), np.hypot(rj1, ry1)          # envelope: errors relative to it (zeros of J, Y)
            assert abs(out[0] - rj0) < 2e-14 * e0 and abs(out[1] - rj1) < 2e-14 * e1, (n, x)
            assert abs(out[2] - ry0) < 2e-14 * e0 and abs(out[3] - ry1) < 2e-14 * e1, (n, x)
