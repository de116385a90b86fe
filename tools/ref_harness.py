"""Reference-run harness (container only; never travels to the GPU box, never imported by the product).

Executes *slices* of the reference solver scripts (read as text from /root/reference) in a scratch
namespace, with the process-local shims described in SURVEY.md section 8c, and records a trace of every
determinant ("mismatch") evaluation the reference worker performs:

  * np.linspace wrapper that casts a float `num` to int (the scripts target an old NumPy),
  * odeint wrapper that flattens y0 to a float vector and logs the end state of each solve,
  * fsolve wrapper that logs `ier`,
  * the module-global history lists (xi_diff_check, all_ws, loop_ws, ...) replaced by logging lists.

Only derived numbers (JSON / NPZ fixtures under tests/golden/) are committed; no reference source is
copied.  Used by tools/gen_golden.py.
"""
import io
import os
import sys
import contextlib
import numpy as np
import scipy.integrate
import scipy.optimize

REF = "/root/reference"

FILES = {
    "SF-U": "Slab/Non uniform flow/Solver/flow_multiprocessor.py",
    "SF-G": "Slab/Non uniform flow/Solver/flow_multiprocessor_coronal.py",
    "SD-P": "Slab/Non uniform density/Photospheric/Solvers/multiprocessor_Inhomogeneous_method.py",
    "SD-C": "Slab/Non uniform density/Coronal/Solvers/multiprocessor_Inhomogeneous_method_coronal.py",
    "CD-C": "Cylinder/Non-uniform density/Coronal/solvers/Density_cylinder.py",
    "CD-P": "Cylinder/Non-uniform density/Photospheric/Solvers/Density_cylinder_photospheric.py",
    "CF": "Cylinder/Non-uniform flow/Coronal/solvers/Cylinder_method_flow_testing.py",
    "CR-KF": "Cylinder/Rotational flow/Photospheric/Solvers/Twisted_photospheric_nonlinear_flow_kink_fast.py",
    "CR-KS": "Cylinder/Rotational flow/Photospheric/Solvers/Twisted_photospheric_nonlinear_flow_kink_slow.py",
    "CR-SF": "Cylinder/Rotational flow/Photospheric/Solvers/Twisted_photospheric_flow_sausage.py",
    "CR-SS": "Cylinder/Rotational flow/Photospheric/Solvers/Twisted_photospheric_flow_sausage_slow.py",
}

TRACE = []          # global event log of the current run


class LogList(list):
    """list that logs append / slice-clear events (module-global history lists of the workers)."""

    def __init__(self, name, init=()):
        super().__init__(init)
        self._name = name

    def append(self, v):
        try:
            fv = float(np.real(v))
        except Exception:
            fv = float("nan")
        TRACE.append(("append", self._name, fv))
        super().append(v)

    def __setitem__(self, key, value):
        if isinstance(key, slice):
            TRACE.append(("assign", self._name, [float(np.real(x)) for x in value]))
        super().__setitem__(key, value)


_real_linspace = np.linspace


def _linspace(start, stop, num=50, *a, **kw):
    num = int(num)
    if num == 3:
        TRACE.append(("linspace3", float(np.real(start)), float(np.real(stop))))
    return _real_linspace(start, stop, num, *a, **kw)


def _odeint(func, y0, t, *a, **kw):
    y0f = np.array([np.ravel(np.asarray(v, dtype=float))[0] for v in y0], dtype=float) \
        if not (isinstance(y0, np.ndarray) and y0.ndim == 1) else y0
    kw.pop("printmessg", None)
    out = scipy.integrate.odeint(func, y0f, t, *a, **kw)
    res = out[0] if isinstance(out, tuple) else out
    TRACE.append(("odeint", float(t[0]), float(t[-1]), int(len(t)),
                  [float(x) for x in y0f], [float(x) for x in res[-1]]))
    return out


# CONVERGE = True (tools/gen_golden.py --converge): when the reference's fsolve gives up (ier != 1; the scripts use the
# returned slope all the same, SF-U:575, SD-P:487, CD-C:26 silences the warning), the reference's OWN objective -- the
# callback it handed to fsolve, still integrating with its own LSODA calls -- is solved to convergence instead and the worker
# continues with that slope.  The shooting condition is affine in the slope, so this is a secant iteration that converges
# in one step up to LSODA noise; it is repeated until two iterates agree to 1e-9.  The fixtures written in this mode
# (*_conv*) answer "what does the reference compute when its solver does what the author intended", so that a call whose
# decisions differ can no longer be excused by ier != 1.
CONVERGE = False
LSODA_ATOL = 1.5e-8          # scipy.integrate.odeint default (the scripts pass none)


def _resolve_affine(func, x_fail, x0, args):
    """Root of the reference's objective `func` -- mathematically affine in the slope (a linear ODE), numerically affine plus
    the noise of its LSODA solves.  The derivative d is taken ONCE from two abscissae far enough apart that the difference
    of the objective stands clear of that noise; then Newton steps with the fixed d.  At every estimate s the objective is
    also evaluated at s (1 +- 1e-6): the scatter sigma of the three values IS the noise of the reference's own objective
    there (the affine part changes by 1e-6 |d s|), so the slope is defined to unc = max(|f(s)|, sigma) / |d s| and no better.
    Returns (slope, objective calls, code, unc):
      1  unc <= 1e-5: converged;
      4  residual within three sigma of the objective's own noise: as converged as the reference's integrator allows, unc says
         how well that is (percent-level where LSODA marches a solution that grows by many decades);
      0  neither after four steps;  2  the objective does not depend on the slope;  3  non-finite objective."""
    def f(s):
        return float(np.real(np.ravel(func(np.array([s], dtype=float), *args))[0]))
    sa = float(np.ravel(x0)[0])
    fa = f(sa)
    n = 1
    if not np.isfinite(fa):
        return sa, n, 3, float("inf")
    step = max(abs(sa), abs(float(x_fail)) if np.isfinite(x_fail) else 0.0, 1e-6)
    d = None
    for _ in range(10):
        sb = sa + step
        fb = f(sb)
        n += 1
        if not np.isfinite(fb):
            return sa, n, 3, float("inf")
        # the difference must stand clear of the noise of the objective, which is absolute (LSODA's atol) near its root and
        # relative away from it
        if abs(fb - fa) > 1e3 * LSODA_ATOL + 1e-3 * max(abs(fa), abs(fb)):
            d = (fb - fa) / (sb - sa)
            break
        step *= 100.0
    if d is None:
        return sa, n, 2, float("inf")
    s1 = sa - fa / d
    unc = float("inf")
    for _ in range(4):
        h = 1e-6 * abs(s1) if s1 != 0.0 else 1e-12
        v = [f(s1), f(s1 + h), f(s1 - h)]
        n += 3
        if not all(np.isfinite(x) for x in v):
            return s1, n, 3, float("inf")
        mean = sum(v) / 3.0
        sigma = max(abs(x - mean) for x in v)
        unc = max(abs(v[0]), sigma) / max(abs(d * s1), 1e-300)
        if unc <= 1e-5:
            return s1 - v[0] / d, n, 1, unc
        if abs(v[0]) <= 3.0 * sigma:
            return s1, n, 4, unc
        s1 = s1 - v[0] / d
    return s1, n, 0, unc


def _fsolve(func, x0, *a, **kw):
    x, info, ier, msg = scipy.optimize.fsolve(func, x0, *a, full_output=True, **kw)
    if CONVERGE and ier != 1 and np.size(x) == 1:
        try:
            xs, n_extra, code, unc = _resolve_affine(func, x[0], x0, kw.get("args", ()))
        except Exception:
            xs, n_extra, code, unc = float(x[0]), 0, 3, float("inf")
        ok = code in (1, 4)
        # conv: 1 converged to 1e-5 relative (5: the same after fsolve had claimed convergence at a slope that is no root); 4 converged to the noise of the reference's own objective (unc = the relative
        # accuracy to which that noise defines the slope); 0 neither; 2 the objective does not depend on the slope at all
        # (the reference's interior solve fails at once); 3 non-finite objective
        TRACE.append(("fsolve", float(np.ravel(x0)[0]), float(xs if ok else x[0]), int(ier), int(code), int(n_extra),
                      float(unc) if np.isfinite(unc) else -1.0))
        if ok:
            return np.array([xs], dtype=float)
        return x
    if CONVERGE and ier == 1 and np.size(x) == 1:
        # fsolve's ier = 1 only says that its step fell below xtol; verify the root on the objective itself: if the Newton
        # correction the objective asks for at the returned slope (two calls, abscissae 0.1 % apart) exceeds 1e-5 of the
        # slope and the residual stands above LSODA's absolute tolerance, the "converged" slope is not a root (seen in SF-U's
        # driver sweep: a mismatch of the wrong sign between two re-solved neighbours) and is re-solved like the others
        try:
            args = kw.get("args", ())
            xv = float(x[0])
            h = 1e-3 * abs(xv) if xv != 0.0 else 1e-9
            f0 = float(np.real(np.ravel(func(np.array([xv], dtype=float), *args))[0]))
            f1 = float(np.real(np.ravel(func(np.array([xv + h], dtype=float), *args))[0]))
            dloc = (f1 - f0) / h
            if np.isfinite(f0) and np.isfinite(dloc) and dloc != 0.0 and abs(f0) > 4.0 * LSODA_ATOL and \
                    abs(f0 / dloc) > 1e-5 * max(abs(xv), 1e-300):
                xs, n_extra, code, unc = _resolve_affine(func, xv, x0, args)
                if code in (1, 4) and abs(xs - xv) > 1e-5 * max(abs(xs), abs(xv)):
                    # conv 5: fsolve claimed convergence, the objective disagrees, re-solved
                    TRACE.append(("fsolve", float(np.ravel(x0)[0]), float(xs), int(ier), 5, int(n_extra) + 2,
                                  float(unc) if np.isfinite(unc) else -1.0))
                    return np.array([xs], dtype=float)
        except Exception:
            pass
    TRACE.append(("fsolve", float(np.ravel(x0)[0]), float(x[0]), int(ier), 1 if ier == 1 else 0, 0, 0.0))
    return x


class Sink:
    def __init__(self):
        self.items = []

    def put(self, x):
        self.items.append(list(x))


def load_slices(key, slices, replacements=()):
    """exec the 1-indexed inclusive line slices of reference file `key`; return the namespace."""
    path = os.path.join(REF, FILES[key])
    with open(path, "r") as f:
        lines = f.read().split("\n")
    ns = {"__name__": "ref_slice"}
    np.linspace = _linspace
    try:
        for i, (a, b) in enumerate(slices):
            src = "\n".join(lines[a - 1:b])
            for old, new in replacements:
                if old in src:
                    src = src.replace(old, new)
            with contextlib.redirect_stdout(io.StringIO()):
                exec(compile(src, f"<{key}:{a}-{b}>", "exec"), ns)
            if i == 0:
                ns["odeint"] = _odeint
                ns["fsolve"] = _fsolve
    finally:
        pass
    ns["odeint"] = _odeint
    ns["fsolve"] = _fsolve
    return ns


HISTORY_LISTS = [
    "xi_diff_check", "xi_diff_loop_check", "loop_sign_check_kink", "sign_check_kink",
    "all_ws", "all_ks", "loop_ws", "P_diff_check", "P_diff_loop_check", "loop_sign_check",
    "sign_check", "P_diff_check_kink", "P_diff_loop_check_kink", "loop_sign_check_kink",
    "sign_check_kink", "loop_ws_kink", "all_ws_kink", "all_ks_kink",
    "P_diff_check_sausage", "P_diff_loop_check_sausage", "loop_ws_sausage", "all_ws_sausage",
    "all_ks_sausage", "loop_sign_check_sausage", "sign_check_sausage",
    "sol_omegas", "sol_ks", "sol_omegas1", "sol_ks1", "sol_omegas_kink", "sol_ks_kink",
    "sol_omegas_kink1", "sol_ks_kink1",
]


def fresh_state(ns, initial):
    """Reset the module-global history lists to their import-time values (a forked worker gets a fresh copy)."""
    for name, init in initial.items():
        ns[name] = LogList(name, init)


def snapshot_initial(ns):
    init = {}
    for name in HISTORY_LISTS:
        if name in ns and isinstance(ns[name], list):
            init[name] = list(ns[name])
    return init


def run_worker(ns, initial, fn, k, freq):
    """Call ns[fn](k, ws_sink, ks_sink, freq) with fresh global state; return (roots_w, roots_k, trace)."""
    fresh_state(ns, initial)
    del TRACE[:]
    ws, ks = Sink(), Sink()
    np.linspace = _linspace
    ns[fn](k, ws, ks, np.asarray(freq, dtype=float))
    tr = list(TRACE)
    rw = [float(np.real(x)) for x in (ws.items[0] if ws.items else [])]
    rk = [float(np.real(x)) for x in (ks.items[0] if ks.items else [])]
    return rw, rk, tr


def evaluations(trace, n_ext=500):
    """Group a trace into determinant evaluations.

    An evaluation = one exterior solve (the odeint whose grid has `n_ext` points, starts in the far field and
    ends on the boundary |x| = 1), the fsolve-driven interior solves, and the mismatch appended to a *_check
    list.  The frequency is taken from the all_ws / loop_ws append that belongs to the evaluation (some workers
    append before the exterior solve, some after the mismatch).
    """
    evs = []
    cur = None
    pending = None
    for ev in trace:
        if ev[0] == "odeint" and ev[3] == n_ext and abs(ev[1]) > 1.0 + 1e-12 and abs(abs(ev[2]) - 1.0) < 1e-12:
            cur = {"ext_y0": ev[4], "ext_end": ev[5], "x_far": ev[1], "ier": None, "conv": None, "unc": None, "omega": pending,
                   "d": None, "where": None, "int_end": None, "n_int": 0}
            pending = None
            evs.append(cur)
        elif ev[0] == "odeint":
            if cur is not None:
                cur["int_y0"] = ev[4]
                cur["int_end"] = ev[5]
                cur["n_int"] += 1
        elif ev[0] == "fsolve":
            if cur is not None:
                cur["ier"] = ev[3]
                cur["slope"] = ev[2]
                cur["conv"] = ev[4] if len(ev) > 4 else (1 if ev[3] == 1 else 0)
                cur["unc"] = ev[6] if len(ev) > 6 else 0.0
        elif ev[0] == "append":
            name = ev[1]
            if name.startswith(("all_ws", "loop_ws")):
                if cur is not None and cur["d"] is not None and cur["omega"] is None:
                    cur["omega"] = ev[2]
                else:
                    pending = ev[2]
            elif ("diff_check" in name or "diff_loop_check" in name) and cur is not None and cur["d"] is None:
                cur["d"] = ev[2]
                cur["where"] = "loop" if "loop" in name else "main"
    return evs


_PLOT_PREFIXES = ("plt.", "ax.", "ax2.", "ax3.", "fig", "exit()", "ax =", "ax2 =", "ax3 =", "box =", "gs =",
                  "image", "print(")


def load_worker_module(key, replacements=()):
    """exec everything of reference file `key` above its driver block (`wavenumber = ...` / `if __name__`),
    dropping top-level plotting / print statements (they do not influence the workers)."""
    path = os.path.join(REF, FILES[key])
    with open(path, "r") as f:
        lines = f.read().split("\n")
    end = None
    for i, ln in enumerate(lines):
        if ln.startswith("if __name__"):
            end = i
            break
    # driver grid lines directly above `if __name__` (wavenumber = ..., freq = ...) are harmless; keep them
    body = []
    for ln in lines[:end]:
        if ln.startswith(_PLOT_PREFIXES):
            body.append("")
        else:
            body.append(ln)
    src = "\n".join(body)
    for old, new in replacements:
        if old not in src:
            raise KeyError(f"replacement target not found in {key}: {old!r}")
        src = src.replace(old, new)
    ns = {"__name__": "ref_slice"}
    np.linspace = _linspace
    cwd = os.getcwd()
    os.makedirs("/tmp/ref_scratch", exist_ok=True)
    os.chdir("/tmp/ref_scratch")
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            exec(compile(src, f"<{key}>", "exec"), ns)
    finally:
        os.chdir(cwd)
    ns["odeint"] = _odeint
    ns["fsolve"] = _fsolve
    if "odeintz" in ns:
        _orig_z = ns["odeintz"]

        def _odeintz(func, z0, t, **kw):
            # same y0 flattening as for odeint (CR-KF:78 builds np.array(z0) from [scalar, array([x])])
            z0f = [complex(np.ravel(np.asarray(v))[0]) for v in z0]
            return _orig_z(func, z0f, t, **kw)
        ns["odeintz"] = _odeintz
    return ns
