"""The reference worker signature on the GPU (eigensolver_amd.solvers -> es_worker_run) against
 (a) the oracle state machine (oracle/workers.py) driven by the CPU port's determinant, and
 (b) the roots the reference itself `put` in the golden traces (tests/golden/trace_*.json)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import workers as OW  # noqa: E402
from tests import cases  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Sink:
    def __init__(self):
        self.items = []

    def put(self, x):
        self.items.append(list(x))


def _solvers(es_ctx):
    from tests import refcases
    return {name: (factory(es_ctx), key) for name, (key, factory) in refcases.solver_factories().items()}


def _port_evaluator(solver, mode):
    from oracle.port import PortProblem, lib
    import ctypes as C
    prob = solver.problem(mode)
    pp = PortProblem(cases.desc_dict(prob.desc), prob._prof_np)
    L = lib()
    accept_outer = bool(prob.desc.accept_norm)

    def evaluate(k, w, w_cst=None):
        d, rel = C.c_double(), C.c_double()
        st = L.port_eval2(pp.h, k, w, w if w_cst is None else w_cst, C.byref(d), C.byref(rel))
        if st == 1:
            return OW.ST_LEAKY, float("nan"), float("nan"), float("nan")
        # the state machine only needs rel = 100|d|/norm: hand it (d, outer, inner) with that ratio
        dv, rv = d.value, rel.value
        norm = abs(dv) * 100.0 / rv if rv == rv and rv != 0 else float("nan")
        return st, dv, norm, (0.0 if not accept_outer else norm)
    evaluate._keep = pp
    return evaluate


def _trace_names():
    from tests import refcases
    return refcases.trace_names()


def _rootset_names():
    from tests import refcases
    return refcases.rootset_names()


@pytest.mark.parametrize("name", _trace_names())
def test_worker_vs_oracle_state_machine(es_ctx, name):
    """Every traced call: the GPU worker equals the oracle state machine driven by the CPU port -- same roots bit for
    bit, same number of evaluations the reference worker would perform."""
    solver, key = _solvers(es_ctx)[name]
    tr = json.load(open(os.path.join(G, f"trace_{name}.json")))
    for call in tr["calls"]:
        mode = call["fn"]
        spec = OW.SPECS[(key, mode)]
        ref_roots, _, requested = OW.run_worker(spec, _port_evaluator(solver, mode), call["k"], call["freq"])
        got, nev = solver.run_batch(mode, [call["k"]], np.array([call["freq"]]), return_evals=True)
        assert len(got[0]) == len(ref_roots), (name, mode, call["k"], got[0], ref_roots)
        assert int(nev[0]) == len(requested), (name, mode, call["k"], int(nev[0]), len(requested))
        # roots are grid / linspace points: identical decisions give bit-identical values
        assert got[0] == ref_roots, (name, mode, call["k"])
    solver.close()


AGREEMENT = json.load(open(os.path.join(G, "agreement_table.json")))


@pytest.mark.parametrize("name", _trace_names())
def test_worker_signature_vs_reference_roots(es_ctx, name):
    """Same call as the reference made: sausage/kink(wavenumber, ws_sink, ks_sink, freq), one `put` per sink (k list
    first).  Per call (no fraction): where tests/test_reference_agreement.py classifies the call "identical", the GPU
    returns the reference's root list value for value, |d omega / omega| <= 1e-10 (north star); every other call is in
    a committed, explained category (fsolve ier != 1 / singular point / measured LSODA exterior error / LSODA interior
    tolerance) and the GPU returns exactly what the explained algorithm returns (previous test)."""
    solver, key = _solvers(es_ctx)[name]
    tr = json.load(open(os.path.join(G, f"trace_{name}.json")))
    cats = AGREEMENT[f"trace:{name}"]
    assert len(cats) == len(tr["calls"])
    for call, cat in zip(tr["calls"], cats):
        ws, ks = Sink(), Sink()
        getattr(solver, call["fn"])(call["k"], ws, ks, np.array(call["freq"]))
        assert len(ws.items) == 1 and len(ks.items) == 1 and len(ws.items[0]) == len(ks.items[0])
        assert all(k == call["k"] for k in ks.items[0])
        mine, ref = ws.items[0], call["roots_w"]
        if cat == "identical":
            assert len(mine) == len(ref) and all(abs(a - b) <= 1e-10 * abs(b) for a, b in zip(mine, ref)), \
                (name, call["fn"], call["k"], mine, ref)
        else:
            assert cat in ("fsolve", "singular", "exterior", "interior_noise"), (name, cat)
    solver.close()


def test_worker_edge_cases(es_ctx):
    import eigensolver_amd as E
    s = E.CylinderNonUniformFlow(U_i0=0.6, width=1.0, ctx=es_ctx)
    assert s.run_batch("kink", [], np.zeros((0, 5))) == []
    # all points leaky (m_e < 0): nothing evaluated, no roots
    roots, nev = s.run_batch("kink", [1.0], np.array([[5.2, 5.4, 5.6]]), return_evals=True)
    assert roots == [[]] and int(nev[0]) == 0
    # a single frequency, ragged task counts
    roots = s.run_batch("kink", [1.0, 2.0, 3.0], np.array([[3.0], [6.5], [9.9]]))
    assert len(roots) == 3
    # capacity smaller than the number of roots of a task: loose tolerance accepts every grid point
    f = np.linspace(2.8, 4.9, 40)[None, :] * 1.5
    full = s.run_batch("kink", [1.5], f, tol=1e9)
    assert len(full[0]) == 40
    cut = s.run_batch("kink", [1.5], f, tol=1e9, max_roots=4)
    assert cut[0] == full[0][:4]
    # many tasks at once equals one call per task
    ks = np.linspace(0.5, 3.5, 7)
    fr = np.stack([np.linspace(2.75 * k, 4.9 * k, 30) for k in ks])
    batch = s.run_batch("sausage", ks, fr)
    single = [s.run_batch("sausage", [k], fr[i:i + 1])[0] for i, k in enumerate(ks)]
    assert batch == single
    s.close()


def test_driver_block_layout(es_ctx):
    """solve() mirrors the reference `__main__` fan-out / fan-in and returns the pickle layout arrays."""
    import eigensolver_amd as E
    s = E.CylinderNonUniformFlow(U_i0=0.6, width=1.0, ctx=es_ctx)
    out = s.solve(np.linspace(0.5, 3.5, 6), n_per_band=20)
    assert set(out) == {"sausage", "kink"}
    for mode in out:
        w, k = out[mode]
        assert w.shape == k.shape and w.dtype == np.float64
    assert len(out["kink"][0]) > 0
    s.close()


@pytest.mark.parametrize("name", _rootset_names())
def test_driver_sweep_vs_reference_root_sets(es_ctx, name):
    """Driver-style sweeps (k x band x mode) of the reference workers executed in the build container
    (tests/golden/roots_*.json): all calls of a sweep go to the GPU as ONE batch per (mode, band length).  Per call:
      * the GPU root list equals, bit for bit, what the oracle state machine + CPU port give (the algorithm whose
        every difference from the reference is explained call by call in tests/test_reference_agreement.py);
      * calls classified "identical" there reproduce the reference's root list, |d omega / omega| <= 1e-10."""
    from tests import refcases
    solver, key = _solvers(es_ctx)[name]
    calls = refcases.load_calls("roots", name)
    cats = AGREEMENT[f"roots:{name}"]
    assert len(cats) == len(calls) >= 6
    by_shape = {}
    for i, c in enumerate(calls):
        by_shape.setdefault((c["fn"], len(c["freq"])), []).append(i)
    n_identical = 0
    for (mode, n), idx in by_shape.items():
        spec = OW.SPECS[(key, mode)]
        got = solver.run_batch(mode, [calls[i]["k"] for i in idx], np.stack([calls[i]["freq"] for i in idx]))
        ev = _port_evaluator(solver, mode)
        for i, mine in zip(idx, got):
            c = calls[i]
            expect, _, _ = OW.run_worker(spec, ev, c["k"], c["freq"])
            assert mine == expect, (name, mode, c["k"])
            if cats[i] == "identical":
                ref = c["roots_w"]
                assert len(mine) == len(ref) and all(abs(a - b) <= 1e-10 * abs(b) for a, b in zip(mine, ref)), \
                    (name, mode, c["k"], mine, ref)
                n_identical += 1
    assert n_identical == cats.count("identical")
    solver.close()


def test_worker_eval_cap_is_reported(es_ctx, monkeypatch):
    """A task that exceeds its evaluation bound is not silently truncated: es_worker_run returns ES_ERR_EVAL_CAP (the
    Python layer raises) -- triggered here by lowering the bound (ES_WORKER_EVAL_CAP, a test aid) below what a deep
    SF-U refinement chain (p_tol = 1e-6) needs."""
    import eigensolver_amd as E
    s = E.SlabUniformFlow(ctx=es_ctx)
    f = np.linspace(0.3, 0.6, 12)[None, :]
    full, nev = s.run_batch("sausage", [1.0], f, return_evals=True)
    assert len(full[0]) >= 1 and int(nev[0]) > 40
    monkeypatch.setenv("ES_WORKER_EVAL_CAP", "20")
    with pytest.raises(E.EsError, match="evaluation cap"):
        s.run_batch("sausage", [1.0], f)
    monkeypatch.delenv("ES_WORKER_EVAL_CAP")
    assert s.run_batch("sausage", [1.0], f) == full
    s.close()


@pytest.mark.parametrize("log_lanes", [0, 1, 3, 6])
def test_speculative_lanes_do_not_change_results(es_ctx, monkeypatch, log_lanes):
    """A task may own 1 ... 64 lanes that evaluate the mid-point tree below a refinement interval ahead of the
    state machine (ES_WORKER_LOG_LANES forces the count): the root lists and the number of evaluations the
    reference would have performed are identical for every lane count, deep chains (p_tol = 1e-6) included."""
    import eigensolver_amd as E
    runs = [(E.SlabUniformFlow(ctx=es_ctx), np.linspace(0.2, 3.3, 24), None),
            (E.CylinderNonUniformFlow(U_i0=0.6, width=1.0, ctx=es_ctx), np.linspace(0.05, 3.9, 20), 40),
            (E.CylinderRotationalFlow(v_twist=0.15, power=1.25, variant="sausage", ctx=es_ctx), np.linspace(0.8, 3.9, 16), 40),
            (E.SlabNonUniformDensity(width=1.5, ctx=es_ctx), np.linspace(0.3, 3.3, 8), 60)]
    for s, ks, n in runs:
        monkeypatch.setenv("ES_WORKER_LOG_LANES", "0")
        ref = s.solve(ks, n) if n is not None else s.solve(ks)
        monkeypatch.setenv("ES_WORKER_LOG_LANES", str(log_lanes))
        out = s.solve(ks, n) if n is not None else s.solve(ks)
        assert sum(len(v[0]) for v in ref.values()) > 0
        for mode in ref:
            assert np.array_equal(ref[mode][0], out[mode][0]) and np.array_equal(ref[mode][1], out[mode][1]), (type(s).__name__, mode)
        # evaluation counts of one batch
        mode = list(ref)[0]
        k = float(ks[len(ks) // 2])
        band = s.bands(k, n)[0] if n is not None else s.bands(k)[0]
        monkeypatch.setenv("ES_WORKER_LOG_LANES", "0")
        r0, e0 = s.run_batch(mode, [k], band[None, :], return_evals=True)
        monkeypatch.setenv("ES_WORKER_LOG_LANES", str(log_lanes))
        r1, e1 = s.run_batch(mode, [k], band[None, :], return_evals=True)
        assert r0 == r1 and np.array_equal(e0, e1)
        s.close()
