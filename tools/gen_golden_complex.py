"""Capture the coefficient functions of the reference's complex-frequency worker (SF-X) as numbers:

    python tools/gen_golden_complex.py      ->  tests/golden/complex_coefficients.json

The reference builds m_e, m0(x), D(x), coeff(x) with sympy inside its worker loops and hands them to `odeintz` as
the right-hand sides  V_e'' = m_e V_e  (SF-X:419-421)  and  Vx'' = -D Vx' - coeff Vx  (SF-X:440-441).  This script
executes the worker (slices 1-346 and 348-1100 of the file, read as text; shims of tools/ref_harness.py) with a
stand-in `odeint` that does not integrate: it evaluates the right-hand side the worker passes at unit states and at
a few x, which yields exactly m_e, D(x) and coeff(x) as complex numbers, records them and stops the worker.
Only these numbers are committed.  (The rest of SF-X is not reproducible: see oracle/slab_complex.py.)
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness as H  # noqa: E402

H.FILES["SF-X"] = "Slab/Non uniform flow/COMPLEX ANALYSIS/flow_multiprocessor_complex_coronal.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "complex_coefficients.json")
XS = [-1.0, -0.6, -0.25, 0.0, 0.4, 1.0]


class Captured(Exception):
    pass


class Probe:
    def __init__(self):
        self.records = []

    def __call__(self, realfunc, y0, t, *a, **kw):
        """Stand-in for scipy's odeint as called by the reference's odeintz: realfunc(x_as_floats, t) -> floats."""
        def cx(state, x):
            out = realfunc(np.array(state, dtype=complex).view(np.float64), x)
            return np.asarray(out, dtype=np.float64).view(np.complex128)
        if t[-1] == -1.0 and t[0] < -1.0:                      # exterior solve: [V', m_e V]
            self.records.append(("exterior", complex(cx([1.0, 0.0], t[0])[1])))
            # return something shaped like a solution so the worker reaches the interior solve
            return np.tile(np.array(y0, dtype=float), (len(t), 1))
        rows = []
        for x in XS:                                           # interior: [Vx', -D Vx' - coeff Vx]
            rows.append((x, complex(-cx([0.0, 1.0], x)[1]), complex(-cx([1.0, 0.0], x)[1])))
        self.records.append(("interior", rows))
        raise Captured()


def capture(mode, width, k, w):
    ns = H.load_slices("SF-X", [(1, 346), (348, 1100)], replacements=[("dx=1e5", f"dx={width!r}")])
    probe = Probe()
    ns["odeint"] = probe
    orig_z = ns["odeintz"]

    def odeintz(func, z0, t, **kw):              # y0 flattening as in ref_harness (numpy >= 1.24 rejects ragged y0)
        return orig_z(func, [complex(np.ravel(np.asarray(v))[0]) for v in z0], t, **kw)
    ns["odeintz"] = odeintz
    sinks = [H.Sink() for _ in range(4)]
    freq = np.array([w / (1.0 + 1.0j)])          # the worker forms freq[j] + 1j*freq[m]  (SF-X:516, :553)
    try:
        ns[mode](k, *sinks, freq)
    except Captured:
        pass
    ext = [r[1] for r in probe.records if r[0] == "exterior"]
    inner = [r[1] for r in probe.records if r[0] == "interior"]
    assert ext and inner, (mode, k, w, probe.records)
    return {"mode": mode, "width": width, "k": k, "w": [w.real, w.imag], "m_e": [ext[0].real, ext[0].imag],
            "rows": [{"x": x, "D": [D.real, D.imag], "coeff": [c.real, c.imag]} for x, D, c in inner[0]]}


def main():
    cases = []
    for mode in ("kink", "sausage"):
        for width in (0.9, 1e5):
            for k, w in ((1.3, 0.9 + 0.2j), (0.4, 0.35 - 0.1j), (2.2, 2.6 + 0.25j)):
                cases.append(capture(mode, width, k, w))
                print(mode, width, k, w, "m_e", cases[-1]["m_e"], "D(-1)", cases[-1]["rows"][0]["D"])
    consts = {}
    ns = H.load_slices("SF-X", [(1, 346)])
    for name in ("vA_i", "c_i", "vA_e", "c_e", "rho_i", "rho_e", "U_i0", "U_e", "p_tol"):
        consts[name] = float(ns[name])
    with open(OUT, "w") as f:
        json.dump({"constants": consts, "cases": cases}, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
