"""ORACLE (test infrastructure only): the accept / bracket / recursive-refine logic of the reference workers
 and their frequency-grid builders (row a9), restated over an abstract determinant evaluator.

Reference code restated (all workers share this skeleton; differences are the WorkerSpec fields):
  main loop     SF-U:533-625, SF-G:400-519, SD-P:421-531, CD-C:694-821, CF:702-829, CR-KF:601-734
  locate_*      SF-U:452-526, SF-G:290-397, SD-P:312-417, CD-C:548-686, CF:556-694, CR-KF:456-596
The module-global history lists of the scripts are kept as object state for ONE worker call (the reference forks a
fresh process per call, so they start from their import-time values: *_check = [0], all_ws = loop_ws = []).

Quirks that are reproduced on purpose (they decide which roots the reference reports):
  * a point with m_e < 0 is skipped entirely; a point whose m_e is NaN/inf is *evaluated* (NaN < 0 is False) and
    counts in all_ws / loop_ws although its mismatch is NaN;
  * the sign test uses the product with the previously evaluated mismatch, across refinement calls as well
    (P_diff_loop_check / xi_diff_loop_check are never reset);
  * a refinement is only started when more than `min_len` points were evaluated since the last reset
    (`len(loop_ws) > 1` for slabs, `> 2` for cylinders: a sign change between the first two points of a 3-point
    cylinder refinement is ignored, CD-C:680);
  * after a recursive locate_* call returns, the caller's for-loop continues with the *rebound* `omega` array and
    incremented `itt_num`;
  * loop_ws is not cleared when a locate_* chain ends without acceptance, so later chains see stale entries
    (slab sausage workers clear it at every main-loop iteration, SF-U:536);
  * rotational kink workers stop the main loop at the first accepted grid point (`break`, CR-KF:722).
  * CR-SF / CR-SS: locate_sausage() never assigns xi_e_const and therefore reads the enclosing main loop's value
    (CR-SF:617), i.e. the constant at the grid frequency that opened the refinement (`stale_ext_const`).

Pinned by tests/test_oracle_workers.py: replaying the reference's own mismatch values (golden traces) through this
state machine must request exactly the reference's sequence of frequencies and report exactly its roots.
"""
from dataclasses import dataclass

import numpy as np

ST_OK, ST_LEAKY, ST_NONFINITE, ST_CONTINUUM = 0, 1, 2, 3


@dataclass
class WorkerSpec:
    tol: float                 # p_tol / xi_tol / P_tol in percent
    min_len: int               # refinement needs len(ws) > min_len : 1 slabs, 2 cylinders
    itt_cap: int               # locate_* returns when itt_num > itt_cap
    reset_loop_ws_each_iter: bool = False   # slab sausage workers: loop_ws[:] = [] at every main-loop iteration
    break_on_accept: bool = False           # CR kink workers: break the main loop at the first accepted grid point
    accept_norm_outer_only: bool = False    # CR-KS:722 divides by |xi_e| instead of max(|xi_e|, |xi_i|)
    stale_ext_const: bool = False           # CR-SF/SS: locate_sausage reads the enclosing loop's xi_e_const (CR-SF:558 vs :617)
    main_double_append: bool = False        # CR-SF/SS: the main loop appends freq[j] to all_ws TWICE per evaluation
                                            # (CR-SF:684 and :726; CR-SS:688, :730), so len(all_ws) > 2 holds after two
                                            # evaluations and linspace(all_ws[-2], all_ws[-1], 3) is the degenerate
                                            # interval [w, w, w]: these workers never narrow a bracket


# the checked-in values of every worker (file:line of tolerance / cap)
SPECS = {
    ("SF-U", "sausage"): WorkerSpec(1e-6, 1, 200, reset_loop_ws_each_iter=True),      # SF-U:419, :459
    ("SF-U", "kink"): WorkerSpec(1e-6, 1, 200),
    ("SF-G", "sausage"): WorkerSpec(1.0, 1, 100, reset_loop_ws_each_iter=True),       # SF-G:250, :297
    ("SF-G", "kink"): WorkerSpec(1.0, 1, 100),
    ("SD-P", "sausage"): WorkerSpec(3.0, 1, 100, reset_loop_ws_each_iter=True),       # SD-P:275, :354
    ("SD-P", "kink"): WorkerSpec(3.0, 1, 100),
    ("SD-C", "sausage"): WorkerSpec(1.0, 1, 100, reset_loop_ws_each_iter=True),       # SD-C:378, :457
    ("SD-C", "kink"): WorkerSpec(1.0, 1, 100),
    ("CD-C", "kink"): WorkerSpec(1.0, 2, 150), ("CD-C", "sausage"): WorkerSpec(1.0, 2, 150),   # CD-C:522, :558
    ("CD-P", "kink"): WorkerSpec(1.0, 2, 300), ("CD-P", "sausage"): WorkerSpec(1.0, 2, 300),   # CD-P:525, :561
    ("CF", "kink"): WorkerSpec(6.0, 2, 250), ("CF", "sausage"): WorkerSpec(6.0, 2, 250),       # CF:530, :566
    ("CR-KF", "kink"): WorkerSpec(2.5, 2, 500, break_on_accept=True),                 # CR-KF:435, :464
    ("CR-KS", "kink"): WorkerSpec(3.0, 2, 500, break_on_accept=True, accept_norm_outer_only=True),  # CR-KS:441, :722
    ("CR-SF", "sausage"): WorkerSpec(1.5, 2, 250, stale_ext_const=True, main_double_append=True),   # CR-SF:419, :475, :558, :684/:726
    ("CR-SS", "sausage"): WorkerSpec(4.5, 2, 250, stale_ext_const=True, main_double_append=True),   # CR-SS:423, :479, :562, :688/:730
}


class WorkerRun:
    """One call `worker(wavenumber, ws_sink, ks_sink, freq)`.

    evaluate(k, w) -> (status, d, outer, inner):  status ST_LEAKY means m_e < 0 (point skipped by the reference);
    any other status means the reference evaluates the point (d may be NaN).  With spec.stale_ext_const the evaluator
    is called as evaluate(k, w, w_cst) inside locate(), w_cst = grid frequency that opened the refinement."""

    def __init__(self, spec, evaluate, k):
        self.spec, self.evaluate, self.k = spec, evaluate, float(k)
        self.roots = []
        self.requested = []            # every frequency the worker evaluates, in order (main and locate)
        self.main_prev = 0.0           # *_diff_check[-1]
        self.loop_prev = 0.0           # *_diff_loop_check[-1]
        self.all_ws = []
        self.loop_ws = []
        self.w_stale = None
        # one record per evaluated point, in order: what was evaluated and what the worker decided there
        # (tests/agreement.py compares two runs decision by decision)
        self.log = []
        self._last_idx = {"main": None, "loop": None}

    def _record(self, where, w, st, d, outer, inner, prev, accepted, refined):
        self.log.append({"where": where, "w": float(w), "st": int(st), "d": d, "outer": outer, "inner": inner,
                         "prev": prev, "prev_idx": self._last_idx[where], "accepted": bool(accepted),
                         "refined": bool(refined), "w_cst": self.w_stale if where == "loop" else None})
        self._last_idx[where] = len(self.log) - 1

    def _accepts(self, d, outer, inner):
        s = self.spec
        with np.errstate(all="ignore"):
            den = abs(outer) if s.accept_norm_outer_only else max(abs(outer), abs(inner))
            rel = abs(d) * 100.0 / den if den != 0 else float("nan")
        return rel < s.tol

    def _eval(self, w, where):
        if where == "loop" and self.spec.stale_ext_const:
            st, d, outer, inner = self.evaluate(self.k, float(w), self.w_stale)
        else:
            st, d, outer, inner = self.evaluate(self.k, float(w))
        if st != ST_LEAKY:
            self.requested.append((where, float(w)))
        return st, d, outer, inner

    def locate(self, omega, itt):
        s = self.spec
        omega = [float(x) for x in omega]
        for kk in range(3):
            if itt > s.itt_cap:
                break
            st, d, outer, inner = self._eval(omega[kk], "loop")
            if st == ST_LEAKY:
                continue
            self.loop_ws.append(omega[kk])
            prev = self.loop_prev
            sign = d * self.loop_prev
            self.loop_prev = d
            acc = self._accepts(d, outer, inner)
            ref = (not acc) and sign < 0 and len(self.loop_ws) > s.min_len
            self._record("loop", omega[kk], st, d, outer, inner, prev, acc, ref)
            if acc:
                self.roots.append(omega[kk])
                self.loop_ws = []
                break
            elif ref:
                omega = list(np.linspace(self.loop_ws[-2], self.loop_ws[-1], 3))
                itt = itt + 1
                self.loop_ws = []
                self.locate(omega, itt)

    def run(self, freq):
        s = self.spec
        for w in freq:
            if s.reset_loop_ws_each_iter:
                self.loop_ws = []
            st, d, outer, inner = self._eval(w, "main")
            if st == ST_LEAKY:
                continue
            self.all_ws.append(float(w))
            if s.main_double_append:
                self.all_ws.append(float(w))
            prev = self.main_prev
            sign = d * self.main_prev
            self.main_prev = d
            acc = self._accepts(d, outer, inner)
            ref = (not acc) and sign < 0 and len(self.all_ws) > s.min_len
            self._record("main", w, st, d, outer, inner, prev, acc, ref)
            if acc:
                self.roots.append(float(w))
                self.all_ws = []
                if s.break_on_accept:
                    break
            elif ref:
                omega = np.linspace(self.all_ws[-2], self.all_ws[-1], 3)
                self.all_ws = []
                self.w_stale = float(w)
                self.locate(omega, 0)
        return self.roots


def run_worker(spec, evaluate, k, freq):
    """Returns (roots_w, roots_k, requested) exactly as the reference `put`s them (k list first, then omega list)."""
    r = WorkerRun(spec, evaluate, k)
    roots = r.run(np.asarray(freq, dtype=float))
    return roots, [float(k)] * len(roots), r.requested


# ---- a9: frequency-grid builders --------------------------------------------------------------------------------
def band_frequencies(speeds, k, n):
    """CD-C:1142-1145: for consecutive sorted characteristic speeds, linspace(s_i*k, s_{i+1}*k, n). Band end points
    sit exactly on the singular speeds, as in the reference."""
    sp = sorted(speeds)
    return [np.linspace(sp[i] * k, sp[i + 1] * k, n) for i in range(len(sp) - 1)]


def sfu_frequencies(k, cT_i, c_e, U_e):
    """SF-U:813, :838: freq = logspace(0.001, 0.55, 80) - 1 for every k, plus a 100-point body band."""
    return [np.logspace(0.001, 0.55, 80) - 1, np.linspace(cT_i * k, (c_e + U_e) * k, 100)]
