"""GPU complex-frequency path (C ABI section 6) against the oracle of oracle/slab_complex.py."""
import numpy as np
import pytest

from oracle.slab_complex import ComplexFlowSlab

pytestmark = pytest.mark.gpu


def oracle_for(solver, mode):
    e = solver.eq
    return ComplexFlowSlab(c_i=e.c_i0, vA_i=e.vA_i0, c_e=e.c_e, vA_e=e.vA_e, rho_i=e.rho_i0, rho_e=e.rho_e, U_i0=e.U_i0,
                           U_e=e.U_e, width=e.width, mode=mode, L_factor=e.L_factor, ic=e.ic, n_nodes=e.n_nodes,
                           variant="sfx" if solver.variant == 0 else "sfg")


@pytest.mark.parametrize("variant", ["sfx", "sfg"])
@pytest.mark.parametrize("width", [0.9, 1e5])
@pytest.mark.parametrize("mode", ["kink", "sausage"])
def test_points_match_oracle(es_ctx, mode, width, variant):
    from eigensolver_amd import SlabComplexFlow
    s = SlabComplexFlow(width=width, variant=variant, ctx=es_ctx)
    rng = np.random.default_rng(5)
    n_ok = 0
    for k in (0.3, 1.1, 2.7):
        w = rng.uniform(-0.5, 3.0, 96) * k / 1.5 + 1j * rng.uniform(-0.4, 0.4, 96)
        D, st, rel = s.eval_points(mode, k, w)
        d, r, so = oracle_for(s, mode).eval_rk4(k, w)
        assert np.array_equal(st.cpu().numpy(), so)
        ok = so == 0
        n_ok += int(ok.sum())
        scale = np.abs(d[ok]) * 100.0 / r[ok]
        assert np.max(np.abs(D.cpu().numpy()[ok] - d[ok]) / scale) < 1e-10
        assert np.max(np.abs(rel.cpu().numpy()[ok] - r[ok]) / r[ok]) < 1e-7
    assert n_ok > 100
    s.close()


def test_grid_layout_and_phase_speed_mode(es_ctx):
    from eigensolver_amd import SlabComplexFlow
    from eigensolver_amd.shooting import W_ABSOLUTE, W_PHASE_SPEED
    s = SlabComplexFlow(width=0.9, ctx=es_ctx)
    k = np.array([0.4, 1.0, 1.9])
    w_re, w_im = np.linspace(0.1, 1.2, 7), np.linspace(-0.2, 0.3, 5)
    D, st, rel = s.eval_grid("kink", k, w_re, w_im, W_PHASE_SPEED)
    assert tuple(D.shape) == (3, 5, 7)
    for r, kk in enumerate(k):
        W = (w_re[None, :] + 1j * w_im[:, None]) * kk
        Dp, sp, relp = s.eval_points("kink", kk, W.ravel())
        assert np.array_equal(st[r].cpu().numpy().ravel(), sp.cpu().numpy())
        a, b = D[r].cpu().numpy().ravel(), Dp.cpu().numpy()
        ok = sp.cpu().numpy() == 0
        assert np.array_equal(a[ok], b[ok])
    Da, sa, _ = s.eval_grid("kink", k[1:2], w_re, w_im, W_ABSOLUTE)            # k = 1: both modes coincide
    assert np.array_equal(Da.cpu().numpy()[sa.cpu().numpy() == 0], D[1:2].cpu().numpy()[st[1:2].cpu().numpy() == 0])
    s.close()


@pytest.mark.parametrize("mode", ["kink", "sausage"])
def test_real_axis_is_the_real_kernel(es_ctx, mode):
    """variant sfg at Im(omega) = 0: |D_c| equals |D| of es_shoot_eval_points (the real path keeps the sign of the
    exterior amplitude, the complex path divides by the amplitude), same leaky mask."""
    from eigensolver_amd import ShootProblem, SlabComplexFlow, equilibrium as q
    eq = q.SlabFlow(U_i0=0.35, width=1.5)
    s = SlabComplexFlow(equilibrium=eq, variant="sfg", ctx=es_ctx)
    gp = ShootProblem(eq, mode, ctx=es_ctx)
    k = np.repeat([0.6, 1.2, 2.4], 40)
    w = k * np.tile(np.linspace(0.05, 2.4, 40), 3)
    Dr, sr, relr = gp.eval_points(k, w, want_rel=True)
    Dc, sc, relc = s.eval_points(mode, k, w + 0j)
    sr, sc = sr.cpu().numpy(), sc.cpu().numpy()
    ok = (sr == 0) & (sc == 0)
    assert ok.sum() > 30 and np.array_equal(sr == 1, sc == 1)
    Dr, Dc, relr = Dr.cpu().numpy()[ok], Dc.cpu().numpy()[ok], relr.cpu().numpy()[ok]
    scale = np.abs(Dr) * 100.0 / relr
    assert np.max(np.abs(np.abs(Dc.real) - np.abs(Dr)) / scale) < 1e-10
    assert np.max(np.abs(Dc.imag) / scale) < 1e-12
    gp.close()
    s.close()


@pytest.mark.parametrize("width,k,w_re,w_im", [(1e5, 0.5, np.linspace(-0.25, 0.5, 16), np.linspace(-0.25, 0.25, 12)),
                                               (0.9, 2.0, np.linspace(-1.0, 5.0, 40), np.linspace(-0.4, 0.4, 17))])
def test_find_roots_matches_oracle(es_ctx, width, k, w_re, w_im):
    """Same candidate cells (winding of D_c around the cell corners) and, where the secant iteration converges, the
    same roots.  (With a sheared flow the real axis inside [k U_min, k U_max] is a branch cut -- the flow continuum --
    and cells next to it yield candidates whose iteration wanders; those are only required to be flagged alike.)"""
    from eigensolver_amd import SlabComplexFlow
    s = SlabComplexFlow(width=width, ctx=es_ctx)
    D, st, rel = s.eval_grid("kink", np.array([k]), w_re, w_im)
    roots, n = s.find_roots("kink", np.array([k]), w_re, w_im, D, st, n_iter=12)
    ro, relo, flo = oracle_for(s, "kink").find_roots(k, w_re, w_im, n_iter=12, tol=4.0)
    assert n == len(ro) and n >= 2
    conv = (flo == 1) & (relo < 1e-2)
    assert conv.sum() >= 1
    assert np.array_equal(roots["flag"].cpu().numpy()[conv], flo[conv])
    assert np.max(np.abs(roots["w"].cpu().numpy()[conv] - ro[conv])) < 1e-8
    s.close()


def test_kelvin_helmholtz_root_known_answer(es_ctx):
    """Uniform super-critical flow (U_i0 = 1.4 vA_i): the unstable kink root of the closed-form dispersion relation,
    omega = 0.31280685 + 0.09428140 i at k = 0.5, is found from the reference's driver grid."""
    from eigensolver_amd import SlabComplexFlow
    s = SlabComplexFlow(width=1e5, ctx=es_ctx)
    out = s.solve([0.3, 0.5], modes=("kink",))
    w, k = out["kink"]
    for kk, want in ((0.3, 0.14293571535248753 + 0.09060650450374286j), (0.5, 0.3128068480162605 + 0.09428139628771133j)):
        sel = w[k == kk]
        assert sel.size and np.min(np.abs(sel - want)) < 1e-7, (kk, sel)
        assert np.min(np.abs(sel - np.conj(want))) < 1e-7                  # the damped partner
    s.close()


def test_worker_signature(es_ctx):
    from eigensolver_amd import SlabComplexFlow

    class Sink:
        def __init__(self):
            self.items = []

        def put(self, x):
            self.items.append(x)

    s = SlabComplexFlow(width=1e5, ctx=es_ctx)
    ws, ks, wsi, ksi = Sink(), Sink(), Sink(), Sink()
    k = 0.5
    freq = np.linspace(0.0 * k, 1.0 * k, 10) + 1j * np.linspace(-0.25, 0.25, 10)       # SF-X:1127
    s.kink(k, ws, ks, wsi, ksi, freq)
    assert len(ws.items) == len(ks.items) == len(wsi.items) == len(ksi.items) == 1
    assert len(ws.items[0]) == len(ks.items[0]) == len(wsi.items[0]) == len(ksi.items[0]) >= 1
    w = np.array(ws.items[0]) + 1j * np.array(wsi.items[0])
    assert np.min(np.abs(w - (0.3128068480162605 + 0.09428139628771133j))) < 1e-7
    s.close()


def test_unsupported_geometry_is_refused(es_ctx):
    from eigensolver_amd import SlabComplexFlow, _lib, equilibrium as q
    s = SlabComplexFlow(equilibrium=q.SlabDensity(width=1.5), ctx=es_ctx)
    with pytest.raises(_lib.EsError):
        s.eval_points("kink", 1.0, np.array([0.5 + 0.1j]))
    s.close()


def test_gpu_reproduces_the_sfx_kink_worker_at_real_frequencies(es_ctx):
    """ES_CX_SFX at Im(omega) = 0 against the numbers the reference's complex worker itself produced there
    (tests/golden/sfx_kink_real_axis.json, see tests/test_oracle_complex.py): D_c = d_ref / V_e(-1) within LSODA's
    tolerance, Im D_c = 0.  (Off the real axis: parity unpinned)"""
    import json
    import os
    from eigensolver_amd import SlabComplexFlow
    from tests.test_oracle_complex import sfx_bound
    tr = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sfx_kink_real_axis.json")))
    n = 0
    for s in tr["sets"]:
        sol = SlabComplexFlow(U_i0=s["U_i0"], width=s["width"], variant="sfx", ctx=es_ctx)
        ev = [e for e in s["evals"] if e["ier"] == 1 and abs(e["ext_value"]) >= 1e-5]
        w = np.array([complex(e["w"], 0.0) for e in ev])
        D, st, rel = (t.cpu().numpy() for t in sol.eval_points("kink", s["k"], w))
        assert np.all(st == 0)
        for e, d, r in zip(ev, D, rel):
            scale = abs(d) * 100.0 / r
            assert abs(d.imag) <= 1e-12 * abs(d)
            assert abs(d.real - e["d"] / e["ext_value"]) / scale < sfx_bound(e["ext_value"]), (s["k"], e["w"], d)
            n += 1
        sol.close()
    assert n >= 32
