"""BASELINE.json configs[4] as named: fp32 bracket + fp64 refine (es_shoot_find_roots_mixed) against the fp64 path
(es_shoot_eval_grid + es_shoot_find_roots): IDENTICAL bracket set and BIT-IDENTICAL root table, identical statuses, and
the fp32 screening error far inside the margin that sends a point back to fp64."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _compare(gp, k, W, n_bisect=30, tol=1e-3):
    D, st, rel = gp.eval_grid(k, W, want_rel=True)
    r64, c64 = gp.find_roots(k, W, D, st, n_bisect=n_bisect, tol_percent=tol)
    rmx, cmx, Dm, stm, stats = gp.find_roots_mixed(k, W, n_bisect=n_bisect, tol_percent=tol)
    D, st, rel, Dm, stm = (t.cpu().numpy() for t in (D, st, rel, Dm, stm))
    assert stats[2] == 0, stats                       # every screened bracket confirmed by fp64 end values
    assert cmx == c64, (cmx, c64)
    for name in ("k", "w", "w_lo", "w_hi", "resid", "row", "flag"):
        a, b = r64[name].cpu().numpy(), rmx[name].cpu().numpy()
        assert np.array_equal(a, b, equal_nan=True), name
    assert not np.any(stm & 0x80)                     # no unsure mark left
    assert np.array_equal(stm, st)
    ok = st == 0
    assert np.array_equal(np.signbit(Dm[ok]), np.signbit(D[ok]))
    n_pts = D.size
    frac_re = stats[0] / n_pts
    # screening error where fp32 vouched for the sign: |D_fp32 - D_fp64| in units of the scale max(|outer|, |inner|)
    # (= |D| 100 / rel, with accept_norm = 0), to be compared with the margin 5e-2 that sends a point back to fp64
    differs = ok & (Dm != D)
    scale = np.abs(D[differs]) * 100.0 / rel[differs]
    err = np.abs(Dm[differs] - D[differs]) / scale if differs.any() else np.zeros(1)
    # what the sign of a vouched-for point depends on: the fp32 error relative to |D| itself (vouched-for means |D| above
    # 5e-2 of the scale the kernel measures its error against)
    err_D = np.abs(Dm[differs] - D[differs]) / np.abs(D[differs]) if differs.any() else np.zeros(1)
    assert err_D.max() < 0.25, err_D.max()
    return c64, frac_re, float(err.max()), int(differs.sum())


def test_config4_rotation_m0_to_10_fp32_bracket_fp64_refine(es_ctx):
    """Cylinder / rotational flow, m = 0..10, the grid of test_configs_gpu.py::test_config4 (twisted family, N = 2000)."""
    from eigensolver_amd import ShootProblem, equilibrium as q
    k = np.linspace(0.25, 4.0, 192)
    W = 0.7 + (np.arange(256) + 0.5) * (1.45 - 0.7) / 256
    total = 0
    worst = 0.0
    for m in range(0, 11):
        mode = "sausage" if m == 0 else "kink"
        eq = q.CylinderRotation(v_twist=0.1, power=1.0, r_axis=0.01 if m == 0 else 0.001)
        gp = ShootProblem(eq, mode, m=m, ctx=es_ctx)
        c, frac, err, nd = _compare(gp, k, W)
        print(f"m={m}: {c} brackets, {100 * frac:.2f} % of the grid re-evaluated in fp64, "
              f"max fp32 error {err:.2e} of the scale on {nd} screened points")
        assert frac < 0.25 and err < 5e-3, (m, frac, err)        # a tenth of the margin
        total += c
        worst = max(worst, err)
        gp.close()
    assert total > 100


@pytest.mark.parametrize("name", ["CF_flow_kink", "CF_flow_sausage", "CDC_w095_kink", "CF_flow_m3", "CR_kink", "CR_sausage"])
def test_mixed_equals_fp64_other_cylinders(es_ctx, name):
    from eigensolver_amd import ShootProblem
    from tests import cases
    eq, mode, m, (lo, hi) = cases.all_cases()[name]
    gp = ShootProblem(eq, mode, m, ctx=es_ctx)
    k = np.linspace(0.05, 3.9, 40)
    nw = 700                                          # ragged: not a multiple of the segment width
    W = lo + (np.arange(nw) + 0.5) * (hi - lo) / nw
    c, frac, err, nd = _compare(gp, k, W, n_bisect=24)
    print(f"{name}: {c} brackets, {100 * frac:.2f} % re-evaluated, max fp32 error {err:.2e} on {nd} points")
    assert c > 0 and frac < 0.25 and err < 5e-3
    gp.close()


@pytest.mark.parametrize("name", ["SD_w15_sausage", "SD_w15_kink", "SFG_flow_kink", "SFG_flow_sausage", "SFU_sausage"])
def test_mixed_equals_fp64_slabs(es_ctx, name):
    """Round 3: the slab families are screened in fp32 as well (density slab: off-diagonal system; flow slab: companion
    form with the reference's D(x), coeff(x) over one denominator) -- identical bracket set, bit-identical root table."""
    from eigensolver_amd import ShootProblem
    from tests import cases
    eq, mode, m, (lo, hi) = cases.all_cases()[name]
    gp = ShootProblem(eq, mode, m, ctx=es_ctx)
    k = np.linspace(0.1, 3.5, 40)
    nw = 700
    W = lo + (np.arange(nw) + 0.5) * (hi - lo) / nw
    c, frac, err, nd = _compare(gp, k, W, n_bisect=24)
    print(f"{name}: {c} brackets, {100 * frac:.2f} % re-evaluated, max fp32 error {err:.2e} on {nd} points")
    # slabs: the error is bounded against the terms of the inner part BEFORE their cancellation (slab_sign - r1), which can
    # exceed max(|outer|, |inner|); _compare holds it to a quarter of |D| at every vouched-for point
    assert c > 0 and frac < 0.4 and err < 5e-2
    gp.close()


def test_mixed_rejects_untracked_slab_and_handles_empty(es_ctx, monkeypatch):
    import eigensolver_amd as E
    from eigensolver_amd import ShootProblem, equilibrium as q
    # a slab whose continuum flag needs per-node sign tracking (here forced) has no fp32 screening
    monkeypatch.setenv("ES_FORCE_SIGN_TRACKING", "1")
    gp = ShootProblem(q.SlabFlow(U_i0=0.35, width=1.5), "kink", ctx=es_ctx)
    monkeypatch.delenv("ES_FORCE_SIGN_TRACKING")
    with pytest.raises(E.EsError, match="unsupported"):
        gp.find_roots_mixed(np.linspace(0.5, 3, 4), np.linspace(1.5, 2.4, 64))
    gp.close()
    gc = ShootProblem(q.CylinderFlow(U_i0=0.6, width=1.0), "kink", ctx=es_ctx)
    r, c, D, st, stats = gc.find_roots_mixed(np.zeros(0), np.linspace(2.8, 4.9, 16))
    assert c == 0 and stats == (0, 0, 0)
    gc.close()


def test_mixed_fuzz_random_cylinder_problems(es_ctx):
    """tools/fuzz_mixed.py in small: 120 random cylinder problems (family, profile parameters, azimuthal order, window,
    grid size) -- identical bracket counts, bit-identical root tables, identical statuses, no ES_ERR_SCREENING.  (1 900
    cases of two other seeds were run the same way on the GPU box: 0 failures, worst fp32 error 1.7e-3 of the scale.)"""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_mixed.py")
    spec = importlib.util.spec_from_file_location("fuzz_mixed", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(120, seed=11) == 0
