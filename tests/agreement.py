"""Per-call comparison of a worker run with the reference's own run of the same call (helper, no tests here).

For one recorded call `worker(k, ws, ks, freq)` of a reference script (tests/golden/trace_*.json / roots_*.json +
roots_*_evals.npz) two runs of the worker state machine (oracle/workers.py -- the accept / bracket / recursive-refine
logic, pinned to the reference traces by tests/test_oracle_workers.py) are compared decision by decision:

  REF   the reference's own numbers replayed in order: mismatch d, exterior end state (value, slope) of its LSODA
        solve and the fsolve flag `ier` of every evaluation it performed;
  OURS  the algorithm the GPU runs (CPU port oracle/c/shoot_port.c; the GPU is compared with the same state machine
        bit for bit in tests/test_workers_gpu.py).

If every decision (accept / open a refinement) coincides, the two root lists are the same linspace points:
category "identical".  Otherwise the FIRST evaluation at which the decisions differ is looked at (both runs evaluate
the same frequency there) and the call is put into exactly one of

  "fsolve"          the reference's fsolve did not converge (ier != 1, silently used: SURVEY section 5) at that
                    evaluation, or at the evaluation that supplied the previous mismatch of the sign product;
  "singular"        OURS flags the point (or the previous one) ES_PT_NONFINITE / ES_PT_CONTINUUM: a coefficient of
                    the ODE changes sign inside the domain and the reference's LSODA integrates through the pole;
  "exterior"        feeding the reference's OWN exterior end state to our interior (port_eval_ext) reproduces the
                    reference's decision: the difference is the error of its LSODA exterior solve (started at
                    1e-8 <= atol = 1.5e-8), measured for this very evaluation;
  "interior_noise"  with the reference's exterior the acceptance measure is within 100 * eps_int of the tolerance
                    (acceptance flip) or the mismatch within eps_int * scale of zero (sign flip), eps_int =
                    max(3e-4, 40 * 1.5e-8 / A) being LSODA's interior error for a solution of amplitude A
                    (the bound tests/test_oracle_golden.py holds the DOP853 oracle to on every trace); in the converged
                    fixture set also the measured noise of the reference's own objective at that evaluation (`unc`);
  "objective_noise" (converged fixture set only) the reference's re-solved slope is converged only to within the noise of
                    its own objective (measured by the harness at that evaluation: the scatter of the objective at three
                    abscissae 1e-6 apart, conv code 4) and the discrepancy of the mismatches is within the relative slope
                    accuracy `unc` that noise leaves;
  "unexplained"     anything else -- the tests fail on it.
"""
import math

import numpy as np

from oracle import workers as OW
from tests import cases

NOISE_FLOOR = 3e-4
ATOL = 1.5e-8


def port_for(solver, mode):
    """CPU port of the product's problem for `mode` (same es_shoot_desc, same profile samples), no GPU involved."""
    from eigensolver_amd import shooting as sh
    from oracle.port import PortProblem
    d, prof = sh.make_desc(solver.eq, mode)
    d.accept_norm = int(solver.WORKER[mode][5])
    return PortProblem(cases.desc_dict(d), prof)


def ours_evaluator(port):
    def evaluate(k, w, w_cst=None):
        st, d, rel, outer, inner = port.eval_one(k, w, w_cst)
        if st == OW.ST_LEAKY:
            return OW.ST_LEAKY, float("nan"), float("nan"), float("nan")
        return st, d, outer, inner
    return evaluate


class RefReplay:
    """The reference's evaluations replayed in order.  Its numbers are normalised by its own exterior amplitude
    |value| (so they are comparable with OURS): d / A, outer = cst * slope / A (cst from the closed form, as the
    reference computes it), inner = outer - d / A."""

    def __init__(self, port, evals):
        self.port = port
        self.queue = {}
        for rec in evals:
            self.queue.setdefault(rec[1], []).append(rec)
        self.used = []

    def __call__(self, k, w, w_cst=None):
        q = self.queue.get(w)
        if q is None:                          # never evaluated by the reference: must be a skipped (m_e < 0) point
            st = self.port.eval_one(k, w, w_cst)[0]
            if st != OW.ST_LEAKY:
                raise KeyError(("the state machine asked for a point the reference never evaluated", k, w))
            return OW.ST_LEAKY, float("nan"), float("nan"), float("nan")
        rec = q.pop(0) if len(q) > 1 else q[0]
        where, _, d, ev, es, ier = rec[:6]
        code = rec[6] if len(rec) > 6 else None
        unc = rec[7] if len(rec) > 7 else 0.0
        A = abs(ev)
        stx, d_mix, rel_mix, outer_mix, inner_mix = self.port.eval_one(k, w, w_cst, ext=(ev, es))
        outer = outer_mix                      # cst * slope / A with the reference's slope
        dn = d / A
        self.used.append({"w": w, "ier": ier, "conv": code, "unc": max(0.0, unc or 0.0), "A": A, "d_mix": d_mix, "rel_mix": rel_mix, "outer_mix": outer_mix,
                          "inner_mix": inner_mix, "st_mix": stx})
        return OW.ST_OK, dn, outer, outer - dn


def _rel(spec, d, outer, inner):
    with np.errstate(all="ignore"):
        den = abs(outer) if spec.accept_norm_outer_only else max(abs(outer), abs(inner))
        return abs(d) * 100.0 / den if den != 0 else float("nan")


def classify_call(solver, key, call, port=None):
    """-> dict(category, same_roots, ours_roots, detail...) for one recorded reference call."""
    mode = call["fn"]
    spec = OW.SPECS[(key, mode)]
    port = port if port is not None else port_for(solver, mode)
    ours = OW.WorkerRun(spec, ours_evaluator(port), call["k"])
    ours.run(np.asarray(call["freq"], dtype=float))
    same = len(ours.roots) == len(call["roots_w"]) and all(
        abs(a - b) <= 1e-10 * abs(b) for a, b in zip(ours.roots, call["roots_w"]))
    out = {"fn": mode, "k": call["k"], "same_roots": same, "ours_roots": list(ours.roots), "n_evals_ours": len(ours.log)}
    if call["evals"] is None:
        out["category"] = "identical" if same else "no_trace"
        return out
    replay = RefReplay(port, call["evals"])
    ref = OW.WorkerRun(spec, replay, call["k"])
    ref.run(np.asarray(call["freq"], dtype=float))
    # the replayed state machine IS the reference (tests/test_oracle_workers.py); keep that honest here as well
    assert ref.roots == call["roots_w"], ("replay does not reproduce the reference", key, mode, call["k"])
    n = min(len(ref.log), len(ours.log))
    first = None
    for i in range(n):
        a, b = ref.log[i], ours.log[i]
        if (a["where"], a["w"]) != (b["where"], b["w"]):
            raise AssertionError(("frequency sequences differ without a differing decision", key, mode, call["k"], i))
        if (a["accepted"], a["refined"]) != (b["accepted"], b["refined"]):
            first = i
            break
    if first is None:
        if len(ref.log) != len(ours.log):
            raise AssertionError(("one run stops early without a differing decision", key, mode, call["k"]))
        assert same, ("identical decisions but different roots", key, mode, call["k"])
        out["category"] = "identical"
        return out
    a, b, u = ref.log[first], ours.log[first], replay.used[first]
    tol = spec.tol
    rel_ref, rel_our = _rel(spec, a["d"], a["outer"], a["inner"]), _rel(spec, b["d"], b["outer"], b["inner"])
    out.update({"first": first, "where": a["where"], "w": a["w"], "rel_ref": rel_ref, "rel_ours": rel_our, "tol": tol,
                "ier": u["ier"], "decision_ref": (a["accepted"], a["refined"]),
                "decision_ours": (b["accepted"], b["refined"])})
    pi = a["prev_idx"]
    sign_matters = (a["accepted"] == b["accepted"])          # the difference is in `refined`: a sign-product flip
    up = replay.used[pi] if pi is not None else None
    bp = ours.log[pi] if pi is not None else None
    flagged = b["st"] in (OW.ST_NONFINITE, OW.ST_CONTINUUM) or (sign_matters and bp is not None and
                                                                bp["st"] in (OW.ST_NONFINITE, OW.ST_CONTINUUM))
    # converged-fixture set: a point our evaluation flags (a coefficient of the ODE changes sign inside the domain) is the
    # REASON the reference's solver cannot converge there -- its objective, mathematically affine, is integrator noise
    # through the pole -- so it is looked at first; in the first fixture set the reference's own flag comes first
    if u.get("conv") is not None and flagged:
        out["category"] = "singular"
        return out
    # 1. non-converged fsolve in the reference
    if u["ier"] != 1 or (sign_matters and up is not None and up["ier"] != 1):
        out["category"] = "fsolve"
        # converged-fixture set: WHY the reference's own objective could not be solved at that evaluation (conv code of
        # tools/ref_harness.py), and how far its mismatch is from what our interior gives behind its own exterior
        bad = u if u["ier"] != 1 else up
        out["fsolve_reason"] = {0: "no root within the noise of its objective (next to a pole of D the objective's slope vanishes)",
                                2: "its objective does not depend on the slope (the interior LSODA solve fails at once)",
                                3: "its objective is non-finite", None: "ier != 1 (first fixture set)"}.get(bad["conv"], str(bad["conv"]))
        sc = max(abs(bad["outer_mix"]), abs(bad["inner_mix"]))
        out["fsolve_pole"] = bool(abs(bad["outer_mix"]) <= max(NOISE_FLOOR, 40 * ATOL / bad["A"]) * abs(bad["inner_mix"])) if sc > 0 else None
        return out
    # 2. singular point (ours flags it)
    if flagged:
        out["category"] = "singular"
        return out
    # 3. the reference's exterior error: our interior behind ITS exterior end state
    rel_mix = u["rel_mix"]
    acc_mix = rel_mix < tol
    prev_mix = up["d_mix"] if up is not None else 0.0
    len_ok = a["refined"] or b["refined"]                    # the length condition holds for both (same history so far)
    ref_mix = (not acc_mix) and (u["d_mix"] * prev_mix < 0) and len_ok
    scale = max(abs(u["outer_mix"]), abs(u["inner_mix"]))
    eps = max(NOISE_FLOOR, 40 * ATOL / u["A"])
    out.update({"rel_mix": rel_mix, "eps_int": eps,
                "ext_shift_percent": abs(rel_mix - rel_our)})   # what the reference's exterior error moves the measure by
    if (acc_mix, ref_mix) == (a["accepted"], a["refined"]):
        out["category"] = "exterior"
        return out
    # 4. LSODA noise of the interior solve: with the same exterior, the reference's mismatch differs from ours by no
    #    more than its integrator tolerance allows (discrepancy <= eps_int of the scale) at the evaluation -- or, for a
    #    sign flip, at the evaluation that supplied the previous mismatch -- and that difference is what flips the
    #    decision.  Next to a POLE of D (|outer| <= eps_int |inner|: the interior solution that meets the far-end
    #    condition vanishes at the boundary, the mismatch changes sign through infinity) the sign is within the same
    #    noise.
    def noise(log_rec, used):
        sc = max(abs(used["outer_mix"]), abs(used["inner_mix"]))
        e = max(NOISE_FLOOR, 40 * ATOL / used["A"])
        disc = abs(log_rec["d"] - used["d_mix"]) / sc if sc > 0 else float("inf")
        pole = abs(used["outer_mix"]) <= e * abs(used["inner_mix"])
        return disc, e, pole
    disc, e, pole = noise(a, u)
    out.update({"discrepancy": disc, "pole": bool(pole)})
    if a["accepted"] != acc_mix:
        if disc <= e:
            out["category"] = "interior_noise"
            return out
    elif a["refined"] != ref_mix:
        ok_here = disc <= e or pole
        ok_prev = False
        if up is not None:
            dp, ep, pp = noise(ref.log[pi], up)
            ok_prev = dp <= ep or pp
            out.update({"discrepancy_prev": dp, "eps_int_prev": ep, "pole_prev": bool(pp)})
        # the sign product flips through the point whose two mismatches (reference / ours behind the same exterior) have
        # opposite signs; that point must be within noise
        flip_here = (a["d"] * u["d_mix"] < 0)
        flip_prev = up is not None and (ref.log[pi]["d"] * up["d_mix"] < 0)
        if (flip_here and ok_here) or (flip_prev and ok_prev):
            out["category"] = "interior_noise"
            return out
    # 5. (converged fixture set) the noise of the reference's OWN objective, measured by the harness at this evaluation (or at
    #    the one that supplied the previous mismatch): its re-solve converged only to within that noise (conv code 4) and the
    #    noise leaves the slope -- in which the inner term of the mismatch is linear -- defined to `unc` relative; the
    #    discrepancy between its mismatch and ours behind the same exterior must be within that
    cand = [(u, disc)]
    if up is not None and "discrepancy_prev" in out:
        cand.append((up, out["discrepancy_prev"]))
    for used, dsc in cand:
        if used.get("conv") == 4 and used.get("unc", 0.0) > 0.0 and dsc <= max(used["unc"], eps):
            out["category"] = "objective_noise"
            out["objective_unc"] = used["unc"]
            return out
    out["category"] = "unexplained"
    return out


def classify_fixture(kind, name, solver=None, conv=False):
    """conv=True: the *_conv fixture of the same name (reference run with its unconverged fsolve calls re-solved to
    convergence on its own objective); the "fsolve" category is then left to the evaluations even the re-solve could not
    converge (a non-finite objective)."""
    from tests import refcases
    key, factory = refcases.solver_factories()[name]
    solver = solver if solver is not None else factory()
    ports = {}
    res = []
    for call in refcases.load_calls(kind, name, conv=conv):
        if call["fn"] not in ports:
            ports[call["fn"]] = port_for(solver, call["fn"])
        res.append(classify_call(solver, key, call, ports[call["fn"]]))
    return res


def summarize(results):
    cats = {}
    for r in results:
        cats[r["category"]] = cats.get(r["category"], 0) + 1
    return cats
