"""GPU parity of the shooting path (K3 grid propagator, K4 bracket + bisection, K5 compaction) through the C ABI
against the CPU port (same algorithm, oracle/c/shoot_port.c) and the DOP853 oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import cases  # noqa: E402

CASES = cases.all_cases()
# D agrees with the CPU port to rounding: identical operation order in the propagator; the exterior uses
# exp/log of two different math libraries (<= 1 ulp each).  Tolerance relative to max(|outer|, |inner|).
D_RTOL = 1e-12
ROOT_RTOL = 1e-10          # north star: |d omega / omega| < 1e-10


def _gpu_problem(es_ctx, case):
    from eigensolver_amd import ShootProblem
    eq, mode, m, _ = case
    return ShootProblem(eq, mode, m, ctx=es_ctx)


@pytest.mark.parametrize("name", list(CASES))
def test_grid_vs_port(es_ctx, name):
    case = CASES[name]
    eq, mode, m, _ = case
    gp = _gpu_problem(es_ctx, case)
    port = cases.port_problem(eq, mode, m)
    k, W = cases.sample_kw(case, nk=7, nw=150, seed=3)
    D, st, rel = gp.eval_grid(k, W, want_rel=True)
    D, st, rel = D.cpu().numpy(), st.cpu().numpy(), rel.cpu().numpy()
    Dp, relp, stp = port.eval_grid(k, W, w_mode=1, nthreads=8)
    assert np.array_equal(st, stp), name
    ok = st == 0
    assert ok.sum() > 50, (name, ok.sum())
    scale = np.abs(Dp[ok]) * 100.0 / relp[ok]                 # max(|outer|, |inner|)
    err = np.abs(D[ok] - Dp[ok]) / scale
    assert err.max() < D_RTOL, (name, err.max())
    # determinant signs: every point is compared and reported; a sign may differ only where BOTH values are within the
    # documented GPU-vs-port rounding (D_RTOL of the scale) of zero -- no exclusion zone beyond that bound
    diff = np.signbit(D[ok]) != np.signbit(Dp[ok])
    near0 = np.abs(Dp[ok]) <= 1e-9 * scale
    print(f"{name}: {ok.sum()} points, {near0.sum()} with |D| <= 1e-9 scale, {diff.sum()} sign differences, "
          f"max |dD|/scale {err.max():.2e}")
    assert np.all(np.abs(Dp[ok][diff]) <= D_RTOL * scale[diff]) and np.all(np.abs(D[ok][diff]) <= D_RTOL * scale[diff])
    assert np.allclose(rel[ok], relp[ok], rtol=1e-9, atol=1e-12)
    # flagged lanes carry NaN where the reference skips the point
    assert np.all(np.isnan(D[(st == 1) | (st == 2)]))
    gp.close()


@pytest.mark.parametrize("n_nodes", [2, 3, 129, 130, 20001])
def test_unnormalised_march_node_counts(es_ctx, n_nodes):
    """The marches of the untwisted cylinder carry 3^n z and take the factor back by an exact power of two per LDS
    chunk of 128 steps: node counts of one step, one chunk +- one step and 157 chunks (3^20000 would overflow ten times
    over without the rescaling) against the port, which rescales at the same nodes; and D must not depend on the
    scale: the fine march agrees with the 1001-node march to the RK4 truncation error."""
    from eigensolver_amd import ShootProblem, equilibrium as q
    eq = q.CylinderFlow(U_i0=0.35, width=0.9, n_nodes=n_nodes)
    k = np.array([0.7, 2.3])
    W = np.linspace(0.95, 4.9, 96)
    for mode, m in (("kink", 1), ("sausage", 0)):
        gp = ShootProblem(eq, mode, m, ctx=es_ctx)
        D, st, rel = (t.cpu().numpy() for t in gp.eval_grid(k, W, want_rel=True))
        Dp, relp, stp = cases.port_problem(eq, mode, m).eval_grid(k, W, w_mode=1, nthreads=8)
        assert np.array_equal(st, stp)
        ok = st == 0
        assert ok.sum() > 40 and np.all(np.isfinite(D[ok]))
        scale = np.abs(Dp[ok]) * 100.0 / relp[ok]
        assert (np.abs(D[ok] - Dp[ok]) / scale).max() < D_RTOL
        if n_nodes == 20001:
            eq1 = q.CylinderFlow(U_i0=0.35, width=0.9, n_nodes=1001)
            g1 = ShootProblem(eq1, mode, m, ctx=es_ctx)
            D1, st1, rel1 = (t.cpu().numpy() for t in g1.eval_grid(k, W, want_rel=True))
            both = ok & (st1 == 0)
            sc = np.abs(D1[both]) * 100.0 / rel1[both]
            assert np.median(np.abs(D[both] - D1[both]) / sc) < 1e-7
            g1.close()
        gp.close()


def test_unnormalised_march_axis_target(es_ctx):
    """A non-zero target of the axis condition (bc_const, forced: the untwisted profiles of the reference have none) is
    multiplied at es_problem_create by the factor the unnormalised march leaves on z: GPU against the port, which is
    checked against a normalised march in tests/test_oracle_port.py."""
    from eigensolver_amd import ShootProblem, equilibrium as q

    class Forced(q.CylinderFlow):
        def bc_const(self, axis_bc):
            return 0.37

    eq = Forced(U_i0=0.4, width=0.9, n_nodes=331)
    k = np.array([0.3, 1.7, 3.6])
    W = 0.95 + (np.arange(160) + 0.5) * (4.0 / 160)
    gp = ShootProblem(eq, "kink", 1, ctx=es_ctx)
    D, st, rel = (t.cpu().numpy() for t in gp.eval_grid(k, W, want_rel=True))
    Dp, relp, stp = cases.port_problem(eq, "kink", 1).eval_grid(k, W, w_mode=1, nthreads=8)
    assert np.array_equal(st, stp)
    ok = st == 0
    scale = np.abs(Dp[ok]) * 100.0 / relp[ok]
    assert ok.sum() > 200 and (np.abs(D[ok] - Dp[ok]) / scale).max() < D_RTOL
    gp.close()


@pytest.fixture(scope="module")
def ieee_ctx():
    """Context on the test-only build with -DES_IEEE_DIVISION (lib/libeigensolver_amd_ieee.so): every reciprocal of the
    hot loops is the IEEE quotient, as in the CPU port."""
    from eigensolver_amd import _lib, build
    ctx = _lib.Context(0, lib=_lib.load_variant(build.LIB_IEEE))
    yield ctx
    ctx.close()


def _ulps(a, b):
    """Distance in units in the last place between two finite float64 arrays of the same sign."""
    ia, ib = a.view(np.int64), b.view(np.int64)
    return np.abs(ia - ib)


@pytest.mark.parametrize("name", ["SD_w15_kink", "SD_w15_sausage", "CF_flow_kink", "CF_flow_sausage", "CDC_w095_kink",
                                  "CF_flow_m3"])
def test_ieee_division_build_is_bitwise_the_port(ieee_ctx, es_ctx, name):
    """With IEEE divisions the GPU and the CPU port run the same operations in the same order on the same inputs: the
    only difference left is exp / log of two math libraries in the closed-form EXTERIOR (<= 1 ulp each).
      * density slab (FAM_SLABD): the exterior's exp() only enters through a term below 1e-16 of the value, so D is
        BITWISE the port's at every evaluated point;
      * untwisted cylinder (FAM_CYL0): the Bessel exterior carries exp / log roundings, D within a few ulps of the
        scale, bitwise at most points -- the interior march itself is identical.
    The default build differs from this one only through fast_rcp / qdiv (<= 1 ulp of each reciprocal): asserted below
    as |D_default - D_ieee| <= 1e-12 of the scale and identical statuses."""
    from eigensolver_amd import ShootProblem
    case = CASES[name]
    eq, mode, m, _ = case
    gi = ShootProblem(eq, mode, m, ctx=ieee_ctx)
    gd = ShootProblem(eq, mode, m, ctx=es_ctx)
    port = cases.port_problem(eq, mode, m)
    k, W = cases.sample_kw(case, nk=6, nw=160, seed=17)
    Di, sti, reli = (t.cpu().numpy() for t in gi.eval_grid(k, W, want_rel=True))
    Dd, std = (t.cpu().numpy() for t in gd.eval_grid(k, W))
    Dp, relp, stp = port.eval_grid(k, W, w_mode=1, nthreads=8)
    assert np.array_equal(sti, stp) and np.array_equal(std, stp)
    ev = (stp == 0) | (stp == 3)                      # evaluated points (continuum points are marched too)
    assert ev.sum() > 100
    same = Di[ev] == Dp[ev]
    scale = np.abs(Dp[ev]) * 100.0 / relp[ev]
    if name.startswith("SD_"):
        assert np.all(same), (name, int((~same).sum()), int(ev.sum()))
        assert np.array_equal(reli[ev], relp[ev])
    else:
        err = np.abs(Di[ev] - Dp[ev]) / scale
        print(f"{name}: {int(same.sum())} of {int(ev.sum())} points bitwise equal, max |dD|/scale {err.max():.2e}")
        assert same.mean() > 0.5 and err.max() < 2e-14, (name, same.mean(), err.max())
        assert np.array_equal(np.signbit(Di[ev]), np.signbit(Dp[ev])) or np.all(
            np.abs(Dp[ev][np.signbit(Di[ev]) != np.signbit(Dp[ev])]) < 2e-14 * scale[np.signbit(Di[ev]) != np.signbit(Dp[ev])])
    ok = stp == 0
    sc = np.abs(Dp[ok]) * 100.0 / relp[ok]
    assert np.max(np.abs(Dd[ok] - Di[ok]) / sc) < D_RTOL
    gi.close()
    gd.close()


@pytest.mark.parametrize("name", ["CF_flow_kink", "CR_kink", "SD_w15_kink", "SFG_flow_sausage"])
def test_points_kernel_equals_grid_kernel(es_ctx, name):
    """The per-lane path used by the refinement (scalar-loaded base table) and the LDS-staged grid path run the
    same arithmetic: results must be bit-identical, otherwise brackets and their refinement could disagree."""
    case = CASES[name]
    gp = _gpu_problem(es_ctx, case)
    k, W = cases.sample_kw(case, nk=4, nw=70, seed=5)
    D, st = gp.eval_grid(k, W)
    kk = np.repeat(k, len(W))
    ww = (k[:, None] * W[None, :]).ravel()
    Dp, stp = gp.eval_points(kk, ww)
    D, Dp = D.cpu().numpy().ravel(), Dp.cpu().numpy()
    assert np.array_equal(st.cpu().numpy().ravel(), stp.cpu().numpy())
    both = ~np.isnan(D)
    assert np.array_equal(D[both], Dp[both]) and np.array_equal(np.isnan(D), np.isnan(Dp))
    gp.close()


@pytest.mark.parametrize("name", ["CF_flow_kink", "CF_flow_sausage", "CDC_w095_kink", "SD_w15_sausage",
                                  "SFG_flow_kink"])
def test_grid_vs_truth_oracle(es_ctx, name):
    """Against the adaptive DOP853 restatement of the reference ODEs (independent of the RK4 grid)."""
    case = CASES[name]
    eq, mode, m, _ = case
    gp = _gpu_problem(es_ctx, case)
    truth = cases.truth_problem(eq, mode, m)
    k, W = cases.sample_kw(case, nk=3, nw=8, seed=11)
    D, st, rel = gp.eval_grid(k, W, want_rel=True)
    D, st, rel = D.cpu().numpy(), st.cpu().numpy(), rel.cpu().numpy()
    n = 0
    for i, kk in enumerate(k):
        for j, Wj in enumerate(W):
            d, a, b, s = truth.mismatch(kk, kk * Wj)
            if s != 0 or st[i, j] != 0:
                assert (s == st[i, j]) or 3 in (s, st[i, j])
                continue
            n += 1
            tol = 3e-8 * max(1.0, (1000.0 / eq.n_nodes) ** 4)
            assert abs(D[i, j] - d) <= tol * max(abs(a), abs(b)), (name, kk, Wj, D[i, j], d)
    assert n >= 8
    gp.close()


@pytest.mark.parametrize("name", ["CF_flow_kink", "CF_flow_sausage", "CF_uniform_kink", "CDC_w095_kink", "CR_kink",
                                  "SD_w15_kink", "SFG_flow_kink", "SFU_sausage"])
def test_roots_vs_port(es_ctx, name):
    case = CASES[name]
    eq, mode, m, (lo, hi) = case
    gp = _gpu_problem(es_ctx, case)
    port = cases.port_problem(eq, mode, m)
    k = np.linspace(0.4, 3.9, 24)
    nw = 192
    W = lo + (np.arange(nw) + 0.5) * (hi - lo) / nw
    D, st = gp.eval_grid(k, W)
    roots, cnt = gp.find_roots(k, W, D, st, n_bisect=44, tol_percent=1e-4)
    Dp, relp, stp = port.eval_grid(k, W, w_mode=1, nthreads=8)
    assert np.array_equal(st.cpu().numpy(), stp)
    rp, cntp = port.find_roots(k, W, Dp, stp, w_mode=1, n_bisect=44, tol=1e-4, nthreads=8)
    assert cnt == cntp, (name, cnt, cntp)
    assert cnt > 0, name
    g = {n_: v.cpu().numpy() for n_, v in roots.items()}
    assert np.array_equal(g["row"], rp["row"]) and np.array_equal(g["k"], rp["k"])
    assert np.array_equal(g["flag"], rp["flag"]), name
    acc = g["flag"] == 1
    assert acc.sum() > 0, name
    dw = np.abs(g["w"] - rp["w"]) / np.abs(rp["w"])
    assert dw[acc].max() < ROOT_RTOL, (name, dw[acc].max())
    # accepted roots lie inside their grid bracket, ordered rows-outer / omega-inner
    assert np.all(np.diff(g["row"]) >= 0)
    same = np.diff(g["row"]) == 0
    assert np.all(np.diff(g["w"])[same] > 0)
    # shallow refinements: no section round at all (polish steps inside refine_kernel) and a single round followed by
    # the one-lane polish kernel -- the port's table value for value
    for nb in (0, 3):
        r2, c2 = gp.find_roots(k, W, D, st, n_bisect=nb, tol_percent=1e-4)
        p2, cp2 = port.find_roots(k, W, Dp, stp, w_mode=1, n_bisect=nb, tol=1e-4, nthreads=8)
        assert c2 == cp2 == cnt
        g2 = {n_: v.cpu().numpy() for n_, v in r2.items()}
        assert np.array_equal(g2["flag"], p2["flag"]), (name, nb)
        d2 = np.abs(g2["w"] - p2["w"]) / np.abs(p2["w"])
        assert d2.max() < 1e-9, (name, nb, d2.max())       # unconverged estimates: differences of the 1e-12 D amplified by the secant
        assert np.all((g2["w_lo"] <= g2["w"]) & (g2["w"] <= g2["w_hi"]))
    gp.close()


def test_edge_cases(es_ctx):
    import torch
    case = CASES["CF_flow_kink"]
    gp = _gpu_problem(es_ctx, case)
    # empty inputs
    D, st = gp.eval_grid(np.zeros(0), np.linspace(3, 4, 5))
    assert D.shape == (0, 5)
    D, st = gp.eval_grid([1.0], np.zeros(0))
    assert D.shape == (1, 0)
    # ragged widths (not multiples of the wave size) and a single column
    for nw in (1, 63, 65, 130, 1000, 1025):
        W = np.linspace(2.8, 4.9, nw)
        D, st = gp.eval_grid([0.7, 2.2], W)
        kk = np.repeat([0.7, 2.2], nw)
        Dp, stp = gp.eval_points(kk, (np.array([0.7, 2.2])[:, None] * W[None, :]).ravel())
        a, b = D.cpu().numpy().ravel(), Dp.cpu().numpy()
        assert np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])
    # leaky (m_e < 0) and singular points are flagged, not evaluated
    D, st = gp.eval_grid([1.0], [5.5, 5.0, 0.4975185951049946, 3.0], w_mode=1)
    st = st.cpu().numpy().ravel()
    assert st[0] == 1 and st[3] == 0
    assert st[1] in (1, 2) and st[2] in (1, 2)
    # absolute and per-row frequency modes agree with the phase-speed mode
    k = np.array([0.9, 1.7])
    W = np.linspace(2.8, 4.9, 40)
    D1, _ = gp.eval_grid(k, W, w_mode=1)
    D2, _ = gp.eval_grid(k, k[:, None] * W[None, :], w_mode=2)
    assert torch.equal(torch.nan_to_num(D1), torch.nan_to_num(D2))
    D0, _ = gp.eval_grid([1.7], 1.7 * W, w_mode=0)
    assert torch.equal(torch.nan_to_num(D0[0]), torch.nan_to_num(D1[1]))
    # root table capacity smaller than the number of brackets: count reported, first entries identical
    kk = np.linspace(0.4, 3.9, 40)
    WW = 2.7 + (np.arange(256) + 0.5) * (4.95 - 2.7) / 256
    D, st = gp.eval_grid(kk, WW)
    full, n = gp.find_roots(kk, WW, D, st, n_bisect=30)
    cut, n2 = gp.find_roots(kk, WW, D, st, n_bisect=30, capacity=3)
    assert n2 == n and cut["w"].numel() == 3
    assert torch.equal(cut["w"], full["w"][:3])
    gp.close()


def test_full_size_properties(es_ctx):
    """BASELINE config 4 size (4096 x 4096, Cylinder / non-uniform flow): size-independent properties."""
    import torch
    case = CASES["CF_flow_kink"]
    eq, mode, m, (lo, hi) = case
    gp = _gpu_problem(es_ctx, case)
    n = 4096
    k = np.linspace(0.01, 4.0, n)
    W = lo + (np.arange(n) + 0.5) * (hi - lo) / n
    D, st = gp.eval_grid(k, W)
    roots, cnt = gp.find_roots(k, W, D, st, n_bisect=40, tol_percent=1e-3, capacity=1 << 18)
    assert 0 < cnt < (1 << 18)
    # (a) spot check of the grid against the per-point kernel: bit-identical
    rng = np.random.default_rng(7)
    ii, jj = rng.integers(0, n, 4000), rng.integers(0, n, 4000)
    Dp, stp = gp.eval_points(k[ii], k[ii] * W[jj])
    a = D[torch.as_tensor(ii), torch.as_tensor(jj)].cpu().numpy()
    b = Dp.cpu().numpy()
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])
    # (b) against the CPU port on the same sample
    port = cases.port_problem(eq, mode, m)
    Dc, relc, stc = port.eval_points(k[ii[:600]], k[ii[:600]] * W[jj[:600]], nthreads=8)
    okc = stc == 0
    assert np.array_equal(stp.cpu().numpy()[:600], stc)
    sc = np.abs(Dc[okc]) * 100 / relc[okc]
    assert (np.abs(b[:600][okc] - Dc[okc]) / sc).max() < D_RTOL
    # (c) every accepted root sits in a genuine sign change of D and has a tiny residual
    r = {n_: v.cpu().numpy() for n_, v in roots.items()}
    acc = r["flag"] == 1
    assert acc.sum() > 1000
    assert np.all(r["resid"][acc] < 1e-3)
    assert np.all((r["w_lo"] <= r["w"]) & (r["w"] <= r["w_hi"]))
    assert np.all(r["w_hi"][acc] - r["w_lo"][acc] < 1e-9 * np.abs(r["w"][acc]))
    # (d) ordering and idempotence
    assert np.all(np.diff(r["row"]) >= 0)
    roots2, cnt2 = gp.find_roots(k, W, D, st, n_bisect=40, tol_percent=1e-3, capacity=1 << 18)
    assert cnt2 == cnt and torch.equal(roots2["w"], roots["w"])
    gp.close()


def _stored_tags():
    from tests import stored_sets as S
    return S.PINNED


@pytest.mark.parametrize("tag", _stored_tags())
def test_gpu_accepts_stored_reference_roots(es_ctx, tag):
    """Known answers: the roots the reference authors stored (all 86 pinned Example data/*.pickle files) satisfy the
    GPU determinant at the tolerance of their worker, point for point as the CPU port decides."""
    from eigensolver_amd import ShootProblem
    from tests import stored_sets as S
    eq, tol = S.describe(tag)
    for mode, w, k in S.pairs(tag):
        if len(w) == 0:
            continue
        gp = ShootProblem(eq, mode, ctx=es_ctx)
        D, st, rel = gp.eval_points(k, w, want_rel=True)
        rel = rel.cpu().numpy()
        frac = float(np.mean(rel < tol))
        assert frac >= S.floor_of(tag, mode) - 0.05, (tag, mode, frac)       # committed measured fraction, max drop 0.05
        Dp, relp, stp = cases.port_problem(eq, mode).eval_points(k, w, nthreads=8)
        assert np.array_equal(st.cpu().numpy(), stp)
        decided = np.abs(relp - tol) > 1e-6 * tol                  # points not sitting on the threshold itself
        assert np.array_equal((rel < tol)[decided], (relp < tol)[decided]), (tag, mode)
        gp.close()


@pytest.mark.parametrize("n_nodes", [2, 3, 130, 20001])
def test_node_count_extremes(es_ctx, n_nodes):
    """Node counts from the minimum (one RK4 step) to the reference's dense slab grids (SD-P:89 uses 1e5 nodes):
    chunked LDS staging with ragged last chunks must agree with the CPU port."""
    import dataclasses
    from eigensolver_amd import ShootProblem, equilibrium as q
    eq = dataclasses.replace(q.SlabDensity(width=1.5), n_nodes=n_nodes)
    gp = ShootProblem(eq, "kink", ctx=es_ctx)
    port = cases.port_problem(eq, "kink")
    k = np.array([0.7, 1.9, 3.1])
    W = np.linspace(1.01, 1.24, 70)
    D, st, rel = gp.eval_grid(k, W, want_rel=True)
    Dp, relp, stp = port.eval_grid(k, W, w_mode=1, nthreads=8)
    assert np.array_equal(st.cpu().numpy(), stp)
    ok = stp == 0
    assert ok.sum() > 50
    sc = np.abs(Dp[ok]) * 100.0 / relp[ok]
    assert (np.abs(D.cpu().numpy()[ok] - Dp[ok]) / sc).max() < 1e-11
    gp.close()


def test_continuum_bands_equal_per_node_tracking_gpu(es_ctx, monkeypatch):
    """Same statuses and the same D with the band test (default) and with per-node sign tracking (fallback path for
    profiles whose node intervals do not overlap), on the grid kernel and on the points kernel."""
    from eigensolver_amd import ShootProblem, equilibrium as q
    k = np.linspace(0.05, 4.0, 16)
    W = np.linspace(0.03, 5.0, 2048)
    for eq, mode in [(q.CylinderFlow(U_i0=0.7, width=0.9), "kink"), (q.CylinderDensity(width=0.9), "sausage"),
                     (q.SlabFlow(U_i0=0.35, width=1.5), "kink"), (q.SlabDensity(width=1.5), "sausage")]:
        monkeypatch.delenv("ES_FORCE_SIGN_TRACKING", raising=False)
        gb = ShootProblem(eq, mode, ctx=es_ctx)
        monkeypatch.setenv("ES_FORCE_SIGN_TRACKING", "1")
        gt = ShootProblem(eq, mode, ctx=es_ctx)
        monkeypatch.delenv("ES_FORCE_SIGN_TRACKING")
        Db, stb = gb.eval_grid(k, W)
        Dt, stt = gt.eval_grid(k, W)
        stb, stt = stb.cpu().numpy(), stt.cpu().numpy()
        assert np.array_equal(stb, stt) and int((stb == 3).sum()) > 100
        assert np.array_equal(Db.cpu().numpy()[stb == 0], Dt.cpu().numpy()[stb == 0])
        kk = np.repeat(k, 64)
        ww = kk * np.tile(W[::32], len(k))
        Dpb, spb = gb.eval_points(kk, ww)
        Dpt, spt = gt.eval_points(kk, ww)
        assert np.array_equal(spb.cpu().numpy(), spt.cpu().numpy())
        gb.close()
        gt.close()


@pytest.mark.parametrize("n_nodes", [2, 3, 4, 129, 130, 258, 1000])
def test_node_count_extremes_cylinder(es_ctx, n_nodes):
    """The paired-step march of the wide-row cylinder kernel (two RK4 steps per iteration, a single leading step in
    chunks of odd length) and the one-point-per-lane kernels for step counts around the chunk size (128): both
    against the CPU port, and against each other bit for bit."""
    import dataclasses
    from eigensolver_amd import ShootProblem, equilibrium as q
    eq = dataclasses.replace(q.CylinderFlow(U_i0=0.7, width=0.9), n_nodes=n_nodes)
    gp = ShootProblem(eq, "kink", ctx=es_ctx)
    port = cases.port_problem(eq, "kink")
    k = np.array([0.7, 1.9, 3.1])
    W = 2.7 + (np.arange(2048) + 0.5) * (4.95 - 2.7) / 2048                    # >= 2048 columns: PTS = 4 variant
    D, st, rel = gp.eval_grid(k, W, want_rel=True)
    Dp, relp, stp = port.eval_grid(k, W, w_mode=1, nthreads=8)
    st, D = st.cpu().numpy(), D.cpu().numpy()
    assert np.array_equal(st, stp)
    ok = stp == 0
    assert ok.sum() > 1000
    sc = np.abs(Dp[ok]) * 100.0 / relp[ok]
    assert (np.abs(D[ok] - Dp[ok]) / sc).max() < 1e-11
    kk = np.repeat(k, 64)
    ww = kk * np.tile(W[::32], 3)
    Dq, sq = gp.eval_points(kk, ww)
    assert np.array_equal(Dq.cpu().numpy().reshape(3, 64), D[:, ::32], equal_nan=True)
    gp.close()


@pytest.mark.parametrize("name,w_mode", [("CF_flow_kink", 1), ("CF_flow_sausage", 1), ("CDC_w095_kink", 1), ("CF_flow_m3", 1),
                                         ("SD_w15_kink", 1), ("SFG_flow_kink", 1), ("CR_kink", 1), ("CF_flow_kink", 2),
                                         ("CF_flow_kink", 3)])
def test_skip_continuum_flag(es_ctx, name, w_mode):
    """es_shoot_eval_grid_ex(ES_EVAL_SKIP_CONTINUUM): statuses as without the flag; D and rel bit-identical at every
    point that is not ES_PT_CONTINUUM, NaN at the continuum points; with ES_W_PHASE_SPEED whole columns inside a band
    leave the launch (device-side column compaction) -- rows wider than one omega-segment, ragged sizes, a column
    exactly on a band edge (kept in the launch by the margin test) and the per-row mode (no compaction) included."""
    case = CASES[name]
    eq, mode, m, (lo, hi) = case
    gp = _gpu_problem(es_ctx, case)
    k = np.linspace(0.3, 3.9, 5)
    nw = 1500 if name == "CF_flow_kink" else 333
    if w_mode == 3:
        # rows of >= 2048 points: the 256-thread x 4-point launch shape, whose compacted form is two launches (full
        # segments in 4-wave workgroups, the remainder of every row in one-wave workgroups)
        w_mode, nw = 1, 2600
    if name.startswith("CF_flow"):
        lo = 0.9                          # include the cusp and Alfven bands of the flow profile
    W = np.linspace(lo, hi, nw)
    if name.startswith("CF_flow"):
        # phase speeds exactly on the edges of the Alfven / cusp bands of this profile
        vz = eq.v_z(np.array([-1.0, -1e-3]))
        W[10], W[11] = vz[0] + eq.vA_i0, vz[1] + eq.vA_i0
    if w_mode == 2:
        wq = (k[:, None] * W[None, :]).copy()
        D0, st0, rel0 = gp.eval_grid(k, wq, w_mode=2, want_rel=True)
        D1, st1, rel1 = gp.eval_grid(k, wq, w_mode=2, want_rel=True, skip_continuum=True)
    else:
        D0, st0, rel0 = gp.eval_grid(k, W, want_rel=True)
        D1, st1, rel1 = gp.eval_grid(k, W, want_rel=True, skip_continuum=True)
    D0, st0, rel0, D1, st1, rel1 = (t.cpu().numpy() for t in (D0, st0, rel0, D1, st1, rel1))
    cont = st1 == 3
    # statuses as without the flag, except (documented in the header) that a point inside a band whose full evaluation is
    # non-finite -- a frequency exactly on a singular node -- is reported CONTINUUM, whether or not it was marched
    differ = st0 != st1
    assert np.all((st0[differ] == 2) & (st1[differ] == 3)) and differ.sum() <= 4, (name, np.argwhere(differ)[:5])
    keep = ~cont
    assert np.array_equal(D0[keep], D1[keep], equal_nan=True) and np.array_equal(rel0[keep], rel1[keep], equal_nan=True)
    assert np.all(np.isnan(D1[cont])) and np.all(np.isnan(rel1[cont]))
    if name.startswith(("CF_flow", "CDC", "SD_")):
        assert cont.sum() > 20, (name, cont.sum())           # the window does contain continuum points
    # brackets / roots are unaffected (both ends of a bracket must be ES_PT_OK)
    if w_mode == 1:
        import torch
        r0, c0 = gp.find_roots(k, W, torch.as_tensor(D0, device="cuda"), torch.as_tensor(st0, device="cuda"), n_bisect=20)
        r1, c1 = gp.find_roots(k, W, torch.as_tensor(D1, device="cuda"), torch.as_tensor(st1, device="cuda"), n_bisect=20)
        assert c0 == c1 and np.array_equal(r0["w"].cpu().numpy(), r1["w"].cpu().numpy())
    gp.close()


@pytest.mark.parametrize("name", ["CF_flow_kink", "CR_kink", "SD_w15_kink", "SFG_flow_kink"])
def test_grid_shapes_bit_identical(es_ctx, name, monkeypatch):
    """D, rel and the status of a point do not depend on the launch shape: every shape the family's table holds
    (1, 2, 4 points per lane at its register cap; ES_GRID_SHAPE) gives the same bits, rows wider than one segment and
    ragged row lengths included, and equals the per-point kernel."""
    import ctypes as C
    from eigensolver_amd import _lib
    case = CASES[name]
    gp = _gpu_problem(es_ctx, case)
    k, W = cases.sample_kw(case, nk=3, nw=1111, seed=5)
    pts, wpe, trk = C.c_int(0), C.c_int(0), C.c_int(0)
    ref = None
    monkeypatch.setenv("ES_GRID_ROWS2", "0")                # the one-row shapes (two rows per workgroup: its own test below)
    for nw in (1111, 130):
        wpes = {}
        monkeypatch.delenv("ES_GRID_SHAPE", raising=False)
        # the register cap the table pairs with each points-per-lane count: ask the library with row widths that select it
        for probe_nw in (64, 128, 192, 256, 384, 512, 1024, 4096):
            _lib.check(es_ctx.handle, es_ctx.lib.es_shoot_grid_shape(es_ctx.handle, gp.handle, probe_nw, C.byref(pts), C.byref(wpe), C.byref(trk)))
            if pts.value > 0:                            # negative: the two-rows-per-workgroup shape (its own test below)
                wpes[pts.value] = wpe.value
        assert len(wpes) >= 2, wpes
        out = []
        for p_, w_ in sorted(wpes.items()):
            monkeypatch.setenv("ES_GRID_SHAPE", f"{p_},{w_}")
            D, st, rel = gp.eval_grid(k, W[:nw], want_rel=True)
            out.append((p_, D.cpu().numpy(), st.cpu().numpy(), rel.cpu().numpy()))
        monkeypatch.delenv("ES_GRID_SHAPE", raising=False)
        for p_, D, st, rel in out[1:]:
            assert np.array_equal(D, out[0][1], equal_nan=True), (name, nw, p_)
            assert np.array_equal(st, out[0][2]) and np.array_equal(rel, out[0][3], equal_nan=True), (name, nw, p_)
        Dq, sq = gp.eval_points(np.repeat(k, nw), (k[:, None] * W[None, :nw]).ravel())
        assert np.array_equal(Dq.cpu().numpy().reshape(3, nw), out[0][1], equal_nan=True), (name, nw)
        assert np.array_equal(sq.cpu().numpy().reshape(3, nw), out[0][2])
    gp.close()


@pytest.mark.parametrize("name", ["CF_flow_kink", "SD_w15_kink", "SFG_flow_kink"])
def test_skip_continuum_whole_waves_inside_a_band(es_ctx, name):
    """ES_EVAL_SKIP_CONTINUUM without column compaction (per-row and absolute frequencies): waves and workgroups ALL of
    whose points lie inside a continuum band are not marched -- their points must still be reported ES_PT_CONTINUUM, not
    ES_PT_NONFINITE (the boundary algebra of an unmarched point is 0/0); rows of 300 and 900 points, 1 and 2 points per
    lane, > 256 consecutive band points per row."""
    case = CASES[name]
    eq, mode, m, (lo, hi) = case
    gp = _gpu_problem(es_ctx, case)
    k = np.linspace(0.5, 3.5, 4)
    for nw in (300, 900):
        D0, st0 = gp.eval_grid(k, np.linspace(0.05 if name.startswith("S") else 0.9, hi, 4001))
        st0 = st0.cpu().numpy()
        Wall = np.linspace(0.05 if name.startswith("S") else 0.9, hi, 4001)
        band = np.where((st0 == 3).all(axis=0))[0]
        runs = np.split(band, np.where(np.diff(band) > 1)[0] + 1)          # longest contiguous run of band columns
        band = max(runs, key=len)
        assert band.size > 50, (name, band.size)
        # a window of nw phase speeds whose first three quarters lie inside the band, the rest outside
        inside = np.linspace(Wall[band[0]], Wall[band[-1]], nw * 3 // 4 + 2)[1:-1]
        outside = np.linspace(lo, hi, nw - inside.size)
        W = np.concatenate([inside, outside])
        wq = (k[:, None] * W[None, :]).copy()
        a = [t.cpu().numpy() for t in gp.eval_grid(k, wq, w_mode=2, want_rel=True)]
        b = [t.cpu().numpy() for t in gp.eval_grid(k, wq, w_mode=2, want_rel=True, skip_continuum=True)]
        cont = b[1] == 3
        assert cont[:, :inside.size].mean() > 0.9
        differ = a[1] != b[1]
        assert np.all((a[1][differ] == 2) & (b[1][differ] == 3)) and differ.sum() <= 4, (name, nw, np.argwhere(differ)[:5])
        assert np.array_equal(a[0][~cont], b[0][~cont], equal_nan=True) and np.all(np.isnan(b[0][cont]))
    gp.close()


@pytest.mark.parametrize("name", ["CF_flow_kink", "CR_kink", "SFG_flow_kink"])
def test_find_roots_async_is_find_roots(es_ctx, name):
    """es_shoot_find_roots_async (count in device memory, launches sized for the table capacity, no read-back) writes the
    table of es_shoot_find_roots bit for bit; es_root_table_pack_async packs it as es_root_table_pack does; a capacity
    below the count is reported through the count, the first `capacity` records are the same."""
    import torch
    from eigensolver_amd import distributed as Dm
    case = CASES[name]
    gp = _gpu_problem(es_ctx, case)
    k, W = cases.sample_kw(case, nk=24, nw=200, seed=11)
    D, st = gp.eval_grid(k, W)
    ref, n = gp.find_roots(k, W, D, st, n_bisect=16, tol_percent=1e-3)
    assert n > 8
    for cap in (max(16, 2 * n), n, 5):
        tab = gp.alloc_root_table(cap)
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        full = gp.find_roots_async(k, W, D, st, tab, cnt, n_bisect=16, tol_percent=1e-3)
        torch.cuda.synchronize()
        assert int(cnt.item()) == n
        m_ = min(n, cap)
        for key in ("k", "w", "w_lo", "w_hi", "resid", "row", "flag"):
            assert torch.equal(full[key][:m_], ref[key][:m_]), (name, cap, key)
        rows = torch.arange(len(k), device="cuda") * 3 + 1
        a = Dm.pack_fixed({key: v for key, v in full.items()}, cnt, 2, rows, 64, ctx=es_ctx)
        torch.cuda.synchronize()
        assert int(a[0, 0].item()) == n                   # the header row carries the true count, overflow included
        if cap >= n:
            b = Dm.pack_fixed({key: v for key, v in full.items()}, n, 2, rows, 64, ctx=es_ctx)
            torch.cuda.synchronize()
            assert torch.equal(a, b)
        else:
            assert torch.all(a[1 + cap:] == 0) and torch.equal(a[1:1 + cap, 1], ref["w"][:cap])
    gp.close()


def test_grid_timer_counts_launches(es_ctx):
    """es_context_grid_timer / es_context_grid_time: one event pair per grid-march launch on the context's stream."""
    case = CASES["CF_flow_kink"]
    gp = _gpu_problem(es_ctx, case)
    k, W = cases.sample_kw(case, nk=64, nw=512, seed=2)
    es_ctx.grid_timer(True)
    try:
        es_ctx.grid_time()
        for _ in range(3):
            gp.eval_grid(k, W)
        ms, n = es_ctx.grid_time()
        assert n == 3 and 0.0 < ms < 1e3
        assert es_ctx.grid_time() == (0.0, 0)
    finally:
        es_ctx.grid_timer(False)
    gp.eval_grid(k, W)
    assert es_ctx.grid_time() == (0.0, 0)
    gp.close()


def test_more_tiles_than_one_grid_dimension_holds(es_ctx):
    """One workgroup per tile: beyond 2^22 tiles the launch uses the y dimension of the grid (es_tile_grid).  4 196 304
    rows of 64 frequencies (one tile each: 64 lanes x 1 point) of a 12-node cylinder; the rows cycle through 8
    wavenumbers, so every row -- those past 2^22 in particular -- must carry the bits of the 8-row grid."""
    import torch
    from eigensolver_amd import ShootProblem, equilibrium as q
    gp = ShootProblem(q.CylinderFlow(U_i0=0.6, width=1.0, n_nodes=12), "kink", None, ctx=es_ctx)
    k8 = np.linspace(0.4, 3.6, 8)
    W = 2.7 + (np.arange(64) + 0.5) * (2.2 / 64)
    D8, st8 = gp.eval_grid(k8, W)
    nk = (1 << 22) + 2000
    kk = torch.as_tensor(np.tile(k8, nk // 8 + 1)[:nk].copy(), device="cuda")
    D, st = gp.eval_grid(kk, W)
    assert D.shape == (nk, 64)
    D8n, st8n = D8.cpu().numpy(), st8.cpu().numpy()
    for lo in (0, (1 << 22) - 8, (1 << 22), nk - 2000):      # first rows, the rows around the seam, the tail
        hi = min(lo + 2000, nk)
        idx = np.arange(lo, hi) % 8
        assert np.array_equal(D[lo:hi].cpu().numpy(), D8n[idx], equal_nan=True), lo
        assert np.array_equal(st[lo:hi].cpu().numpy(), st8n[idx]), lo
    assert (st8n == 0).sum() > 100
    del D, st
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name,nk,nw", [("CF_flow_kink", 7, 384), ("CF_flow_kink", 2, 100), ("CF_flow_kink", 5, 512),
                                        ("CF_flow_kink", 64, 129), ("CF_flow_kink", 3, 257), ("SD_w15_kink", 5, 300),
                                        ("SFG_flow_kink", 9, 384), ("SFG_flow_kink", 4, 64)])
def test_two_rows_per_workgroup_bit_identical(es_ctx, monkeypatch, name, nk, nw):
    """Rows of at most 512 frequencies of the untwisted cylinder and the slabs are marched two to a workgroup
    (shoot_grid_kernel_r2); same bits as the one-row shapes (ES_GRID_ROWS2=0), odd row counts and ragged widths included,
    per-row frequencies too."""
    import ctypes as C
    from eigensolver_amd import _lib
    case = CASES[name]
    gp = _gpu_problem(es_ctx, case)
    k, W = cases.sample_kw(case, nk=nk, nw=nw, seed=11)
    pts, wpe, trk = C.c_int(0), C.c_int(0), C.c_int(0)
    monkeypatch.delenv("ES_GRID_ROWS2", raising=False)
    monkeypatch.delenv("ES_GRID_SHAPE", raising=False)
    _lib.check(es_ctx.handle, es_ctx.lib.es_shoot_grid_shape(es_ctx.handle, gp.handle, nw, C.byref(pts), C.byref(wpe), C.byref(trk)))
    assert pts.value == -((nw + 127) // 128), pts.value
    D2, st2, rel2 = gp.eval_grid(k, W, want_rel=True)
    Wrow = (k[:, None] * W[None, :]).copy()
    Dr2, sr2 = gp.eval_grid(k, Wrow, w_mode=2)[:2]
    monkeypatch.setenv("ES_GRID_ROWS2", "0")
    _lib.check(es_ctx.handle, es_ctx.lib.es_shoot_grid_shape(es_ctx.handle, gp.handle, nw, C.byref(pts), C.byref(wpe), C.byref(trk)))
    assert pts.value > 0
    D1, st1, rel1 = gp.eval_grid(k, W, want_rel=True)
    Dr1, sr1 = gp.eval_grid(k, Wrow, w_mode=2)[:2]
    assert np.array_equal(D2.cpu().numpy(), D1.cpu().numpy(), equal_nan=True)
    assert np.array_equal(st2.cpu().numpy(), st1.cpu().numpy()) and np.array_equal(rel2.cpu().numpy(), rel1.cpu().numpy(), equal_nan=True)
    assert np.array_equal(Dr2.cpu().numpy(), Dr1.cpu().numpy(), equal_nan=True) and np.array_equal(sr2.cpu().numpy(), sr1.cpu().numpy())
    assert (st1.cpu().numpy() == 0).sum() > nk * nw // 4
