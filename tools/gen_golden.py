"""Generate the golden fixtures under tests/golden/ by running the reference here (build container only).

    python tools/gen_golden.py [--only NAME] [--jobs N]

Outputs (all small, committed):
  tests/golden/slab_analytic.npz      SF-U:107-127 values on a (K, W) grid + the SF-U:166-303 scan results
  tests/golden/trace_<case>.json      per-evaluation trace of a reference worker call: omega, mismatch d,
                                      exterior end state (amplitude, slope), fsolve ier, where (main / loop),
                                      the 3-point refinement calls, and the accepted roots the worker `put`s.
The reference sources are read as text from /root/reference and executed in memory through
tools/ref_harness.py; nothing of them is copied into the repository.
"""
import argparse
import json
import os
import sys
import time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


# ----------------------------------------------------------------------------------------------------
def gen_slab_analytic():
    import ref_harness as H
    ns = H.load_worker_module("SF-U")       # runs the SF-U:166-303 scans as written
    # re-create the *normalised* tube speeds used by the analytic part (the module redefines cT later, SF-U:406)
    path = os.path.join(H.REF, H.FILES["SF-U"])
    lines = open(path).read().split("\n")
    ns2 = {"np": np}
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        exec(compile("\n".join(lines[62:149]), "<SF-U:63-149>", "exec"), ns2)
    K = np.linspace(0.05, 3.5, 24)
    W = np.concatenate([np.linspace(0.0005, 2.9995, 200), np.linspace(ns2["cT_i"]() + 1e-5, ns2["c_i"] - 1e-5, 56)])
    out = {"K": K, "W": W}
    names = ["disp_rel_sausage", "disp_rel_kink", "disp_rel_sausage_body", "disp_rel_kink_body"]
    with np.errstate(all="ignore"):
        for n in names:
            D = np.empty((len(K), len(W)))
            for i, x in enumerate(K):
                for j, v in enumerate(W):
                    D[i, j] = ns2[n](v, x)
            out[n] = D
    out["R1"] = np.array(ns2["R1"])
    out["cT_i"] = np.array(ns2["cT_i"]())
    out["cT_e"] = np.array(ns2["cT_e"]())
    for a in ["x_out_sausage", "W_array_sausage", "x_out_kink", "W_array_kink",
              "x_out_sausage_body", "W_array_sausage_body", "x_out_kink_body", "W_array_kink_body",
              "x_out_sausage_body1", "W_array_sausage_body1", "x_out_kink_body1", "W_array_kink_body1",
              "D_range", "W_range", "W_body_range", "W_body_range2"]:
        out["scan_" + a] = np.asarray(ns[a], dtype=float)
    out["step"] = np.array(ns["step"])
    np.savez_compressed(os.path.join(GOLD, "slab_analytic.npz"), **out)
    return "slab_analytic.npz"


# ----------------------------------------------------------------------------------------------------
# worker trace cases: name -> (file key, replacements, [(fn, k, freq-spec)], notes)
def band(lo, hi, n):
    return ("band", lo, hi, n)


SDC_IX = ("1e6)  # inside slab x values     #was 500", "2001)  # inside slab x values")

CASES = {
    # cylinder, non-uniform flow file as checked in: uniform (dr = 1e5, U_i0 = 0)
    "CF_uniform": ("CF", [], [("kink", 1.5, band(2.05, 4.95, 14)), ("sausage", 1.5, band(2.05, 4.95, 14)),
                               ("kink", 0.3, band(2.05, 4.95, 10)), ("sausage", 3.9, band(2.05, 4.95, 10)),
                               ("kink", 2.0, band(0.9, 1.99, 12)), ("sausage", 2.0, band(0.9, 1.99, 12))]),
    # Gaussian axial flow (edit of CF:126 / CF:130 as the author does by hand)
    "CF_flow": ("CF", [("dr=1e5", "dr=1."), ("U_i0 = 0.*c_i0", "U_i0 = 0.6*c_i0")],
                [("kink", 1.5, band(2.7, 4.95, 12)), ("sausage", 1.5, band(2.7, 4.95, 12)),
                 ("kink", 3.0, band(2.7, 4.95, 10)), ("kink", 0.5, band(2.7, 4.95, 8))]),
    # cylinder, Gaussian density, as checked in (dr = 0.95) and uniform limit
    "CDC_w095": ("CD-C", [], [("kink", 2.0, band(2.05, 4.95, 12)), ("sausage", 2.0, band(2.05, 4.95, 12)),
                               ("kink", 0.7, band(2.05, 4.95, 8)), ("sausage", 3.5, band(2.05, 4.95, 8))]),
    "CDC_uniform": ("CD-C", [("dr=0.95", "dr=1e5")], [("kink", 2.0, band(2.05, 4.95, 10)),
                                                       ("sausage", 2.0, band(2.05, 4.95, 10))]),
    "CDP": ("CD-P", [], [("kink", 2.0, band(0.52, 1.48, 10)), ("sausage", 2.0, band(0.52, 1.48, 10)),
                         ("kink", 0.8, band(0.52, 1.48, 8))]),
    # rotational flow files as checked in
    "CRKF": ("CR-KF", [], [("kink", 1.0, band(1.21, 1.44, 10)), ("kink", 2.0, band(1.21, 1.44, 8))]),
    "CRSF": ("CR-SF", [], [("sausage", 1.5, band(1.05, 1.4, 10)), ("sausage", 3.0, band(1.05, 1.4, 8))]),
    "CRKS": ("CR-KS", [], [("kink", 1.0, band(0.7, 0.99, 10))]),
    # slabs
    "SFU": ("SF-U", [], [("sausage", 1.0, ("lin", 0.3, 0.6, 12)), ("kink", 1.0, ("lin", 0.3, 0.6, 12)),
                         ("sausage", 2.5, ("lin", 0.8, 1.6, 10)), ("kink", 0.4, ("lin", 0.1, 0.3, 10))]),
    "SFG_uniform": ("SF-G", [], [("sausage", 1.0, band(1.05, 2.45, 10)), ("kink", 1.0, band(1.05, 2.45, 10))]),
    "SFG_flow": ("SF-G", [("dx=1e5", "dx=1.5"), ("U_i0 = 0.9*vA_i", "U_i0 = 0.35*vA_i")],
                 [("sausage", 1.0, band(1.4, 2.45, 10)), ("kink", 1.0, band(1.4, 2.45, 10)),
                  ("kink", 2.5, band(1.4, 2.45, 8))]),
    "SDP_uniform": ("SD-P", [("1e5)  # inside slab x values", "2001)  # inside slab x values")],
                    [("sausage", 1.0, band(0.9, 0.99, 8)), ("kink", 1.0, band(0.9, 0.99, 8))]),
    "SDP_w15": ("SD-P", [("1e5)  # inside slab x values", "2001)  # inside slab x values"), ("dx=1e5", "dx=1.5")],
                [("sausage", 1.0, band(0.9, 1.25, 8)), ("kink", 1.0, band(0.9, 1.25, 8)),
                 ("kink", 2.5, band(0.9, 1.25, 8))]),
    # coronal density slab as checked in (dx = 0.9, SD-C:110; p_tol = 1, SD-C:378), interior grid 2001 nodes instead of
    # the checked-in 1e6 (SD-C:106); bands between the script's sorted `speeds` (SD-C:202), end points included as in
    # its driver (SD-C:895)
    "SDC_w09": ("SD-C", [SDC_IX], [("sausage", 1.0, band(0.9, 1.0, 8)), ("kink", 1.0, band(0.9, 1.0, 8)),
                                   ("kink", 0.5, band(1.2, 1.3, 8)), ("sausage", 2.0, band(1.0923, 1.2, 8))]),
    # SD-C in the uniform limit (dx = 1e5, the author's benchmark case; stored output width1e5_coronal.pickle): body
    # modes between cT_i0 = 0.768 and c_i0 = 1 -- outside every continuum, unlike the checked-in dx = 0.9 bands
    "SDC_uniform": ("SD-C", [SDC_IX, ("dx=0.9", "dx=1e5")], [("sausage", 1.0, band(0.78, 0.99, 8)),
                                                             ("kink", 1.0, band(0.78, 0.99, 8)),
                                                             ("kink", 2.5, band(0.78, 0.99, 8))]),
    # slow sausage modes of the rotating cylinder as checked in (xi_tol = 4.5, CR-SS:423; speeds CR-SS:232)
    "CRSS": ("CR-SS", [], [("sausage", 1.5, band(0.9, 0.92, 10)), ("sausage", 3.0, band(0.96, 0.98, 8)),
                            ("sausage", 0.8, band(0.98, 1.0, 8))]),
}


def make_freq(spec, k):
    if spec[0] == "band":
        return np.linspace(spec[1] * k, spec[2] * k, spec[3])
    return np.linspace(spec[1], spec[2], spec[3])


SUFFIX = ""          # "_conv" with --converge


def gen_case(name):
    import ref_harness as H
    H.CONVERGE = bool(SUFFIX)
    key, repl, calls = CASES[name]
    t0 = time.time()
    ns = H.load_worker_module(key, repl)
    init = H.snapshot_initial(ns)
    out = {"case": name, "file": H.FILES[key], "replacements": repl, "calls": []}
    for k_ in ("xi_tol", "p_tol", "P_tol"):
        if k_ in ns:
            out[k_] = float(ns[k_])
    for fn, k, spec in calls:
        freq = make_freq(spec, k)
        rw, rk, tr = H.run_worker(ns, init, fn, float(k), freq)
        evs = H.evaluations(tr)
        rec = {"fn": fn, "k": float(k), "freq": [float(x) for x in freq], "roots_w": rw, "roots_k": rk,
               "linspace3": [[e[1], e[2]] for e in tr if e[0] == "linspace3"],
               "evals": [{"omega": e["omega"], "d": e["d"], "ext_end": e["ext_end"], "ier": e["ier"],
                          "where": e["where"], "int_y0": e.get("int_y0"), "int_end": e.get("int_end"),
                          "slope": e.get("slope"), **({"conv": e.get("conv"), "unc": e.get("unc")} if SUFFIX else {})} for e in evs]}
        out["calls"].append(rec)
    out["seconds"] = round(time.time() - t0, 1)
    if SUFFIX:
        out["converged_mode"] = ("fsolve calls with ier != 1 re-solved to convergence on the reference's own objective "
                                 "(tools/ref_harness.py CONVERGE); `ier` is the flag fsolve returned, `conv` the outcome of the re-solve "
                                 "(1 converged, 4 converged to the noise of the objective, 0 / 2 / 3 not), `unc` the relative accuracy "
                                 "to which the objective's own noise defines the slope")
    with open(os.path.join(GOLD, f"trace_{name}{SUFFIX}.json"), "w") as f:
        json.dump(out, f, indent=0)
    return f"trace_{name}{SUFFIX}.json ({out['seconds']} s, {sum(len(c['evals']) for c in out['calls'])} evals)"


def _run(name):
    try:
        if name == "slab_analytic":
            return gen_slab_analytic()
        if name == "equilibria":
            return gen_equilibrium_fixture()
        if name.startswith("roots:"):
            return gen_rootset(name[6:])
        return gen_case(name)
    except Exception as e:  # noqa
        import traceback
        return f"{name}: FAILED {type(e).__name__}: {e}\n{traceback.format_exc()}"


# ----------------------------------------------------------------------------------------------------
# a9 / a10 fixtures: the reference's `speeds` lists and lambdified equilibrium profiles at sample points
def gen_equilibrium_fixture():
    import ref_harness as H
    out = {}
    cfgs = {
        "CD-C": [], "CD-C_w15": [("dr=0.95", "dr=1.5")], "CD-P": [], "CF": [],
        "CF_flow": [("dr=1e5", "dr=1."), ("U_i0 = 0.*c_i0", "U_i0 = 0.6*c_i0")],
        "CR-KF": [], "CR-KS": [], "CR-SF": [], "CR-SS": [],
        "SD-P_w15": [("dx=1e5", "dx=1.5"), ("1e5)  # inside slab x values", "2001)  # inside slab x values")],
        "SD-C": [("1e6)  # inside slab x values     #was 500", "2001)  # inside slab x values")],
        "SF-G_flow": [("dx=1e5", "dx=1.5"), ("U_i0 = 0.9*vA_i", "U_i0 = 0.35*vA_i")],
    }
    for tag, repl in cfgs.items():
        key = tag.split("_")[0]
        ns = H.load_worker_module(key, repl)
        rec = {"file": H.FILES[key], "replacements": repl}
        if "speeds" in ns:
            rec["speeds"] = [float(x) for x in ns["speeds"]]
        for name in ("rho_e", "cT_e", "c_kink", "cT_i0", "R1"):
            if name in ns:
                v = ns[name]
                rec[name] = float(v() if callable(v) else v)
        cyl = key.startswith("C")
        sign = -1.0 if key in ("CD-C", "CF") else 1.0
        pts = (sign * np.array([1.0, 0.75, 0.5, 0.25, 0.1, 0.01])) if cyl else np.array([-1.0, -0.6, -0.2, 0.0, 0.3, 0.9])
        rec["points"] = [float(x) for x in pts]
        for fn in ("rho_i_np", "c_i_np", "vA_i_np", "cT_i_np", "v_iphi_np", "P_i_np", "B_i_np", "U_i_np", "dU_i_np",
                   "ddU_i_np"):
            if fn in ns:
                with np.errstate(all="ignore"):
                    v = np.broadcast_to(np.asarray(ns[fn](pts), dtype=float), pts.shape)
                rec[fn] = [float(x) for x in v]
        if "v_z" in ns and callable(ns["v_z"]):
            import sympy as sym
            rr = sym.symbols("r")
            f = sym.lambdify(rr, ns["v_z"](rr), "numpy")
            rec["v_z"] = [float(x) for x in np.broadcast_to(np.asarray(f(pts), dtype=float), pts.shape)]
        out[tag] = rec
    with open(os.path.join(GOLD, "equilibria.json"), "w") as f:
        json.dump(out, f, indent=1)
    return "equilibria.json"


# ----------------------------------------------------------------------------------------------------
# Larger root sets: whole driver-style sweeps (k x band x mode) of the reference workers, roots only.
ROOTSETS = {
    "CF_flow": ("CF", [("dr=1e5", "dr=1."), ("U_i0 = 0.*c_i0", "U_i0 = 0.6*c_i0")],
                [0.4, 0.9, 1.4, 1.9, 2.4, 2.9, 3.4, 3.9], [(2.7, 4.95)], 40, ("kink", "sausage")),
    "CF_uniform": ("CF", [], [0.3, 0.8, 1.3, 1.8, 2.3, 2.8, 3.3, 3.8], [(2.05, 4.95), (0.9, 0.99)], 40, ("kink", "sausage")),
    "CDC_w095": ("CD-C", [], [0.6, 1.2, 1.8, 2.4, 3.0, 3.6, 4.2], [(2.05, 4.95)], 40, ("kink", "sausage")),
    "SFG_flow": ("SF-G", [("dx=1e5", "dx=1.5"), ("U_i0 = 0.9*vA_i", "U_i0 = 0.35*vA_i")],
                 [0.5, 1.0, 1.5, 2.0, 2.5, 3.0], [(1.4, 2.45)], 40, ("kink", "sausage")),
    "CRKS": ("CR-KS", [], [0.6, 1.0, 1.5, 2.0, 3.0], [(0.7, 0.99), (1.21, 1.44)], 30, ("kink",)),
    "CRSF": ("CR-SF", [], [0.8, 1.2, 1.6, 2.0, 2.5, 3.0, 3.5], [(1.05, 1.4), (0.7, 0.99)], 30, ("sausage",)),
    "SDP_w15": ("SD-P", [("1e5)  # inside slab x values", "2001)  # inside slab x values"), ("dx=1e5", "dx=1.5")],
                [0.5, 1.0, 1.5, 2.0, 2.5, 3.0], [(0.9, 1.25)], 30, ("kink", "sausage")),
    "CDP": ("CD-P", [], [0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 3.5], [(0.52, 1.48)], 30, ("kink", "sausage")),
    "SFG_uniform": ("SF-G", [], [0.5, 1.0, 1.5, 2.0, 2.5, 3.0], [(1.05, 2.45)], 30, ("kink", "sausage")),
    "CRKF": ("CR-KF", [], [0.6, 1.0, 1.5, 2.0, 2.5, 3.0], [(1.21, 1.44)], 30, ("kink",)),
    # the scripts the round-1 fixtures left out: SD-C (bands = consecutive sorted speeds of SD-C:202), CR-SS (CR-SS:232),
    # and SF-U's own driver grid (SF-U:813: logspace(0.001, 0.55, 80) - 1 for every k, :838: the body band)
    "SDC_w09": ("SD-C", [SDC_IX], [0.4, 0.9, 1.4, 2.0, 2.6, 3.2], [(0.9, 1.0), (1.2, 1.3)], 25, ("kink", "sausage")),
    "SDC_uniform": ("SD-C", [SDC_IX, ("dx=0.9", "dx=1e5")], [0.5, 1.0, 1.5, 2.0, 2.5, 3.0], [(0.78, 0.9), (0.9, 1.0)], 25,
                    ("kink", "sausage")),
    "CRSS": ("CR-SS", [], [0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 3.5], [(0.9, 0.92), (0.94, 0.96), (0.98, 1.0)], 40, ("sausage",)),
    "SFU": ("SF-U", [], [0.3, 0.9, 1.5, 2.1, 2.7, 3.3], ["sfu_log", "sfu_body"], 0, ("kink", "sausage")),
    # NEGATIVE frequencies: the sorted `speeds` of SF-G (-vA_e first, SF-G:180) and CD-C (-vA_e ... -cT_i0, CD-C:225) make
    # the reference drivers scan omega < 0 bands too; with a flow they are not the mirror image of the positive ones
    "SFG_flow_neg": ("SF-G", [("dx=1e5", "dx=1.5"), ("U_i0 = 0.9*vA_i", "U_i0 = 0.35*vA_i")],
                     [0.5, 1.5, 2.5], [(-2.45, -1.4), (-1.3, -0.4)], 30, ("kink", "sausage")),
    "CDC_w095_neg": ("CD-C", [], [0.6, 1.8, 3.0], [(-4.95, -2.05)], 40, ("kink", "sausage")),
    # CR-KF with the parameters of one of its stored result files (vtwist01_power1): the checked-in 0.25 / 0.8 makes fsolve
    # fail at most evaluations
    "CRKF_v01p1": ("CR-KF", [("v_twist = 0.25", "v_twist = 0.1"), ("power = 0.8", "power = 1.0")],
                   [0.6, 1.0, 1.5, 2.0, 2.5, 3.0], [(1.21, 1.44)], 30, ("kink",)),
    # SD-P in the uniform limit (its benchmark case): body modes between cT_i0 and c_i0, away from every continuum
    "SDP_uniform": ("SD-P", [("1e5)  # inside slab x values", "2001)  # inside slab x values")],
                    [1.0, 2.0, 3.0], [(0.9, 0.99)], 20, ("kink", "sausage")),
}


def rootset_freq(ns, spec, k, n):
    """Frequencies of one driver task: (lo, hi) = band between two characteristic speeds (linspace(lo k, hi k, n), e.g.
    CD-C:1145), "sfu_log" / "sfu_body" = the two task kinds of SF-U:813, :838."""
    if spec == "sfu_log":
        return np.logspace(0.001, 0.55, 80) - 1
    if spec == "sfu_body":
        return np.linspace(ns["cT_i"]() * k, (ns["c_e"] + ns["U_e"]) * k, 100)
    return np.linspace(spec[0] * k, spec[1] * k, n)


def gen_rootset(name):
    """roots_<name>.json: per call the reference's root list and counters; roots_<name>_evals.npz: every determinant
    evaluation of every call (call index, omega, mismatch d, exterior end state (value, slope), fsolve ier,
    0 = main loop / 1 = inside locate_*) -- what tests/test_reference_agreement.py needs to explain every call whose
    root list differs."""
    import ref_harness as H
    H.CONVERGE = bool(SUFFIX)
    key, repl, ks, bands, n, modes = ROOTSETS[name]
    t0 = time.time()
    ns = H.load_worker_module(key, repl)
    init = H.snapshot_initial(ns)
    out = {"case": name, "file": H.FILES[key], "replacements": repl, "calls": []}
    cols = {c: [] for c in ("call", "omega", "d", "ext_value", "ext_slope", "ier", "where", "conv", "unc")}
    for k in ks:
        for b in bands:
            freq = rootset_freq(ns, b, float(k), n)
            for fn in modes:
                rw, rk, tr = H.run_worker(ns, init, fn, float(k), freq)
                iers = [e[3] for e in tr if e[0] == "fsolve"]
                unconv = [e for e in tr if e[0] == "fsolve" and (e[4] if len(e) > 4 else int(e[3] == 1)) not in (1, 4, 5)]
                evs = H.evaluations(tr)
                rec = {"fn": fn, "k": float(k), "n": len(freq), "roots_w": rw, "n_evals": len(evs),
                       "n_fsolve_fail": int(sum(1 for i in iers if i != 1))}
                if SUFFIX:
                    rec["n_unconverged"] = len(unconv)
                if isinstance(b, str):
                    rec["freq_kind"] = b
                    rec["freq"] = [float(x) for x in freq]
                else:
                    rec["band"] = [b[0], b[1]]
                ci = len(out["calls"])
                for e in evs:
                    ext = e["ext_end"]
                    if len(ext) == 4:               # odeintz: (re, im) pairs, im = 0
                        ext = [ext[0], ext[2]]
                    cols["call"].append(ci)
                    cols["omega"].append(np.nan if e["omega"] is None else e["omega"])
                    cols["d"].append(np.nan if e["d"] is None else e["d"])
                    cols["ext_value"].append(ext[0])
                    cols["ext_slope"].append(ext[1])
                    cols["ier"].append(-1 if e["ier"] is None else e["ier"])
                    cols["conv"].append(-1 if e.get("conv") is None else e["conv"])
                    cols["unc"].append(0.0 if e.get("unc") is None else e["unc"])
                    cols["where"].append(1 if e["where"] == "loop" else (0 if e["where"] == "main" else -1))
                out["calls"].append(rec)
    out["seconds"] = round(time.time() - t0, 1)
    for k_ in ("xi_tol", "p_tol", "P_tol"):
        if k_ in ns:
            out[k_] = float(ns[k_])
    with open(os.path.join(GOLD, f"roots_{name}{SUFFIX}.json"), "w") as f:
        json.dump(out, f, indent=0)
    np.savez_compressed(os.path.join(GOLD, f"roots_{name}{SUFFIX}_evals.npz"), conv=np.array(cols["conv"], dtype=np.int8), unc=np.array(cols["unc"], dtype=np.float32),
                        call=np.array(cols["call"], dtype=np.int32), omega=np.array(cols["omega"]),
                        d=np.array(cols["d"]), ext_value=np.array(cols["ext_value"]),
                        ext_slope=np.array(cols["ext_slope"]), ier=np.array(cols["ier"], dtype=np.int8),
                        where=np.array(cols["where"], dtype=np.int8))
    return f"roots_{name}{SUFFIX}.json ({out['seconds']} s, {sum(len(c['roots_w']) for c in out['calls'])} roots, {len(cols['call'])} evals)"


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--jobs", type=int, default=6)
    ap.add_argument("--converge", action="store_true",
                    help="second fixture set (*_conv*): fsolve calls the reference leaves unconverged (ier != 1) are re-solved "
                         "to convergence on the reference's own objective; worker traces and root sets only")
    a = ap.parse_args()
    if a.converge:
        SUFFIX = "_conv"
    os.makedirs(GOLD, exist_ok=True)
    names = ["slab_analytic", "equilibria"] + list(CASES) + ["roots:" + n for n in ROOTSETS]
    if a.converge:
        names = [n for n in names if n not in ("slab_analytic", "equilibria")]
    if a.only:
        names = [n for n in names if n in a.only.split(",")]
    import multiprocessing as mp
    with mp.Pool(a.jobs) as pool:
        for msg in pool.imap_unordered(_run, names):
            print(msg, flush=True)

