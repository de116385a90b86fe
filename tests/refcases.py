"""Reference-run fixtures (tests/golden/trace_*.json, roots_*.json made by tools/gen_golden.py) and the product
solver object that reproduces each of them: fixture name -> (reference script key, solver factory).

The factories take an optional GPU context; without one the solver object still offers `eq`, `WORKER`, `speeds()` and
`bands()` (the context is created on first GPU use), which is all the CPU-side tests need."""
import json
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def solver_factories():
    import eigensolver_amd.solvers as E
    return {
        "CF_uniform": ("CF", lambda ctx=None: E.CylinderNonUniformFlow(ctx=ctx)),
        "CF_flow": ("CF", lambda ctx=None: E.CylinderNonUniformFlow(U_i0=0.6, width=1.0, ctx=ctx)),
        "CDC_w095": ("CD-C", lambda ctx=None: E.CylinderNonUniformDensity(width=0.95, ctx=ctx)),
        "CDC_uniform": ("CD-C", lambda ctx=None: E.CylinderNonUniformDensity(width=1e5, ctx=ctx)),
        "CDP": ("CD-P", lambda ctx=None: E.CylinderNonUniformDensity(width=0.9, photospheric=True, ctx=ctx)),
        "CRKF": ("CR-KF", lambda ctx=None: E.CylinderRotationalFlow(v_twist=0.25, power=0.8, variant="kink_fast", ctx=ctx)),
        "CRKF_v01p1": ("CR-KF", lambda ctx=None: E.CylinderRotationalFlow(v_twist=0.1, power=1.0, variant="kink_fast", ctx=ctx)),
        "CRKS": ("CR-KS", lambda ctx=None: E.CylinderRotationalFlow(v_twist=0.1, power=0.8, variant="kink_slow", ctx=ctx)),
        "CRSF": ("CR-SF", lambda ctx=None: E.CylinderRotationalFlow(v_twist=0.15, power=1.25, variant="sausage", ctx=ctx)),
        "CRSS": ("CR-SS", lambda ctx=None: E.CylinderRotationalFlow(v_twist=0.15, power=1.25, variant="sausage_slow", ctx=ctx)),
        "SFU": ("SF-U", lambda ctx=None: E.SlabUniformFlow(ctx=ctx)),
        "SFG_uniform": ("SF-G", lambda ctx=None: E.SlabNonUniformFlow(U_i0=0.9, width=1e5, ctx=ctx)),
        "SFG_flow": ("SF-G", lambda ctx=None: E.SlabNonUniformFlow(U_i0=0.35, width=1.5, ctx=ctx)),
        "SFG_flow_neg": ("SF-G", lambda ctx=None: E.SlabNonUniformFlow(U_i0=0.35, width=1.5, ctx=ctx)),
        "CDC_w095_neg": ("CD-C", lambda ctx=None: E.CylinderNonUniformDensity(width=0.95, ctx=ctx)),
        "SDP_uniform": ("SD-P", lambda ctx=None: E.SlabNonUniformDensity(width=1e5, ctx=ctx)),
        "SDP_w15": ("SD-P", lambda ctx=None: E.SlabNonUniformDensity(width=1.5, ctx=ctx)),
        "SDC_w09": ("SD-C", lambda ctx=None: E.SlabNonUniformDensity(width=0.9, coronal=True, ctx=ctx)),
        "SDC_uniform": ("SD-C", lambda ctx=None: E.SlabNonUniformDensity(width=1e5, coronal=True, ctx=ctx)),
    }


def trace_names(conv=False):
    """Worker-trace fixtures; conv=True: the second set (trace_*_conv.json, tools/gen_golden.py --converge), in which the
    reference's unconverged fsolve calls were re-solved to convergence on its own objective."""
    names = sorted(f[6:-5] for f in os.listdir(G) if f.startswith("trace_") and f.endswith(".json"))
    return [n[:-5] for n in names if n.endswith("_conv")] if conv else [n for n in names if not n.endswith("_conv")]


def rootset_names(conv=False):
    names = sorted(f[6:-5] for f in os.listdir(G) if f.startswith("roots_") and f.endswith(".json"))
    return [n[:-5] for n in names if n.endswith("_conv")] if conv else [n for n in names if not n.endswith("_conv")]


def call_freq(c):
    """The frequency array of one recorded driver task (the reference's linspace(lo k, hi k, n) or SF-U's grids)."""
    if "freq" in c:
        return np.array(c["freq"], dtype=np.float64)
    return np.linspace(c["band"][0] * c["k"], c["band"][1] * c["k"], c["n"])


def load_calls(kind, name, conv=False):
    """Uniform view of a fixture: list of dicts {fn, k, freq, roots_w, n_fsolve_fail, evals}, evals = list of
    (where, omega, d, ext_value, ext_slope, ier) in the order the reference evaluated them (None if not recorded).
    conv=True reads the *_conv fixture; there `ier` is 1 for an evaluation whose slope is converged -- by fsolve itself
    or by the harness's re-solve of the reference's own objective -- and fsolve's flag otherwise."""
    sfx = "_conv" if conv else ""

    def eff(ier, cv):
        # conv codes of tools/ref_harness.py: 1 converged, 4 converged to the noise floor of the reference's own objective
        return 1 if (conv and cv in (1, 4, 5)) else ier
    if kind == "trace":
        tr = json.load(open(os.path.join(G, f"trace_{name}{sfx}.json")))
        out = []
        for c in tr["calls"]:
            evs = []
            for e in c["evals"]:
                if e["omega"] is None or e["d"] is None:
                    continue
                ext = e["ext_end"]
                if len(ext) == 4:
                    ext = [ext[0], ext[2]]
                evs.append((e["where"], e["omega"], e["d"], ext[0], ext[1], eff(e["ier"], e.get("conv")), e.get("conv"), e.get("unc") or 0.0))
            out.append({"fn": c["fn"], "k": c["k"], "freq": np.array(c["freq"]), "roots_w": c["roots_w"],
                        "n_fsolve_fail": sum(1 for e in c["evals"] if eff(e["ier"], e.get("conv")) != 1), "evals": evs})
        return out
    rs = json.load(open(os.path.join(G, f"roots_{name}{sfx}.json")))
    ev_path = os.path.join(G, f"roots_{name}{sfx}_evals.npz")
    cols = None
    if os.path.exists(ev_path):
        with np.load(ev_path) as z:
            cols = {n: z[n] for n in z.files}          # NpzFile decompresses on every access: read each array once
    out = []
    for ci, c in enumerate(rs["calls"]):
        evs = None
        if cols is not None:
            sel = np.nonzero(cols["call"] == ci)[0]
            evs = []
            for i in sel:
                if np.isnan(cols["omega"][i]) or cols["where"][i] < 0:
                    continue
                evs.append(("loop" if cols["where"][i] == 1 else "main", float(cols["omega"][i]), float(cols["d"][i]),
                            float(cols["ext_value"][i]), float(cols["ext_slope"][i]),
                            eff(int(cols["ier"][i]), int(cols["conv"][i]) if "conv" in cols else None),
                            int(cols["conv"][i]) if "conv" in cols else None,
                            float(cols["unc"][i]) if "unc" in cols else 0.0))
        out.append({"fn": c["fn"], "k": c["k"], "freq": call_freq(c), "roots_w": c["roots_w"],
                    "n_fsolve_fail": c.get("n_unconverged", c["n_fsolve_fail"]) if conv else c["n_fsolve_fail"], "evals": evs})
    return out
