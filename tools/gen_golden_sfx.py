"""Trace of the reference's complex-frequency worker (SF-X) where it is self-consistent: Im(omega) = 0.

    python tools/gen_golden_sfx.py      ->  tests/golden/sfx_kink_real_axis.json

`kink(wavenumber, ws, ks, ws_imag, ks_imag, freq)` of
Slab/Non uniform flow/COMPLEX ANALYSIS/flow_multiprocessor_complex_coronal.py (SF-X:737; the only worker its driver
starts, SF-X:1127-1131) evaluates the mismatch at omega = freq[j] + 1j*freq[m] for every pair (j, m) of ITS OWN 1-D
`freq` argument (SF-X:921-925).  Called with freq = [w, 0.0] the pair (j, m) = (0, 1) is the real frequency omega = w:
there the real-part-only handling of the main loop (boundary value from freq[j] alone, `p_e_const.real`,
`p_i_const.real`, real unknown slope: SF-X:974-1026) IS the whole computation, so the recorded mismatch is a
well-defined number -- the reference's own evaluation of its complex script's formulas (D of SF-X:940, total pressure
with the U' term SF-X:955-960, :1024).  At complex omega the script mixes real and imaginary parts (same lines) and
`sausage` / `locate_*` do not run (ValueError "too many values to unpack" at SF-X:446: one unknown unpacked from a
two-unknown fsolve): nothing is recorded there.  The acceptance branch would raise NameError (`left_P_solution_imag`,
SF-X:1047, is commented out above it), so frequencies are chosen off the roots and p_tol is lowered in memory.

The worker is executed from the text of the file (slices 1-346 and 348-1100) with the shims of tools/ref_harness.py;
only numbers are written."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness as H  # noqa: E402

H.FILES["SF-X"] = "Slab/Non uniform flow/COMPLEX ANALYSIS/flow_multiprocessor_complex_coronal.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sfx_kink_real_axis.json")
LISTS = ["P_diff_check_kink", "all_ws_kink", "all_ws_kink_imag", "sign_check_kink", "all_ks_kink", "all_ks_kink_imag",
         "test_p_diff_kink"]


def run(width, k, ws, U_i0=None):
    repl = [("dx=1e5", f"dx={width!r}"), ("p_tol = 4.", "p_tol = 1e-9")]
    if U_i0 is not None:
        repl.append(("U_i0 = 1.4*vA_i", f"U_i0 = {U_i0!r}*vA_i"))
    ns = H.load_slices("SF-X", [(1, 346), (348, 1100)], replacements=repl)
    orig_z = ns["odeintz"]

    def odeintz(func, z0, t, **kw):              # y0 flattening as in ref_harness (numpy >= 1.24 rejects ragged y0)
        return orig_z(func, [complex(np.ravel(np.asarray(v))[0]) for v in z0], t, **kw)
    ns["odeintz"] = odeintz
    init = {n: list(ns[n]) for n in LISTS if n in ns}
    out = []
    for w in ws:
        for n, v in init.items():
            ns[n] = H.LogList(n, v)
        del H.TRACE[:]
        sinks = [H.Sink() for _ in range(4)]
        ns["kink"](float(k), *sinks, np.array([float(w), 0.0]))
        # group the trace into evaluations: exterior solve (far field -> -1), interior solves, appended mismatch, (freq[j], freq[m])
        cur, evs = None, []
        for ev in H.TRACE:
            if ev[0] == "odeint" and ev[1] < -1.0 - 1e-12 and abs(ev[2] + 1.0) < 1e-12:
                cur = {"ext_end": ev[5], "ier": None, "d": None, "wj": None, "wm": None}
                evs.append(cur)
            elif ev[0] == "fsolve" and cur is not None:
                cur["ier"], cur["slope"] = ev[3], ev[2]
            elif ev[0] == "append" and cur is not None:
                if ev[1] == "P_diff_check_kink" and cur["d"] is None:
                    cur["d"] = ev[2]
                elif ev[1] == "all_ws_kink":
                    cur["wj"] = ev[2]
                elif ev[1] == "all_ws_kink_imag":
                    cur["wm"] = ev[2]
        hit = [e for e in evs if e["wj"] == float(w) and e["wm"] == 0.0 and e["d"] is not None]
        assert len(hit) == 1, (k, w, evs)
        e = hit[0]
        ext = e["ext_end"]                       # complex state viewed as (re, im) pairs
        assert ext[1] == 0.0 and ext[3] == 0.0, ext
        out.append({"k": float(k), "w": float(w), "d": e["d"], "ext_value": ext[0], "ext_slope": ext[2], "ier": e["ier"]})
        print(width, k, w, e["d"], ext[0], ext[2], e["ier"], flush=True)
    return out


def main():
    res = {"file": H.FILES["SF-X"], "worker": "kink", "note": "freq = [w, 0.0]; recorded: the (j, m) = (0, 1) evaluation, omega = w real",
           "sets": []}
    ns = H.load_slices("SF-X", [(1, 346)])
    res["constants"] = {n: float(ns[n]) for n in ("vA_i", "c_i", "vA_e", "c_e", "rho_i", "rho_e", "U_i0", "U_e")}
    # uniform flow as checked in (dx = 1e5, U_i0 = 1.4): real frequencies below the exterior sound speed
    for k, band in ((0.5, (0.35, 0.9)), (1.0, (0.7, 1.9)), (2.0, (1.6, 3.9))):
        ws = np.linspace(band[0], band[1], 9)
        res["sets"].append({"width": 1e5, "U_i0": 1.4, "k": k, "evals": run(1e5, k, ws)})
    # sheared flow: with the checked-in U_i0 = 1.4 every evanescent real frequency lies inside the flow continuum
    # (omega = k U(x) somewhere in the slab: both the reference and any other integrator return noise there), so the
    # Gaussian profile (dx = 0.9) is traced with U_i0 = 0.2, where the window U_max + c_i < omega/k < c_e is regular:
    # this exercises D(x) with U' != 0 (SF-X:940), coeff (SF-X:948) and the U' term of the total pressure (SF-X:955-960)
    for k, band in ((1.0, (1.55, 1.92)), (2.0, (3.1, 3.85)), (0.5, (0.78, 0.96))):
        ws = np.linspace(band[0], band[1], 7)
        res["sets"].append({"width": 0.9, "U_i0": 0.2, "k": k, "evals": run(0.9, k, ws, U_i0=0.2)})
    with open(OUT, "w") as f:
        json.dump(res, f, indent=0)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
