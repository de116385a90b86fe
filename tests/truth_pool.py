"""Worker side of the DOP853 leg of tests/test_full_size_parity_gpu.py: evaluates oracle/cylinder.py (adaptive DOP853 at
rtol 1e-12, scipy Bessel exterior -- nothing of the RK4 grid, nothing of the C port) at a list of (k, omega) in a process
pool started with `spawn`, so that no child inherits the parent's HIP state.  Test infrastructure only."""
import numpy as np

_TRUTH = None


def _init(eq_kind, eq_kwargs, mode, m):
    global _TRUTH
    from eigensolver_amd import equilibrium as q
    from tests import cases
    eq = getattr(q, eq_kind)(**eq_kwargs)
    _TRUTH = cases.truth_problem(eq, mode, m)


def _one(kw):
    d, a, b, s = _TRUTH.mismatch(kw[0], kw[1])
    return float(d), float(a), float(b), int(s)


def evaluate(eq_kind, eq_kwargs, mode, m, k, w, procs):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    with ctx.Pool(procs, initializer=_init, initargs=(eq_kind, eq_kwargs, mode, m)) as pool:
        res = pool.map(_one, list(zip(np.asarray(k, dtype=float).tolist(), np.asarray(w, dtype=float).tolist())), chunksize=8)
    return np.array(res, dtype=float)
