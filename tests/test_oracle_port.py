"""The C port (oracle/c/shoot_port.c, fixed-grid RK4 as on the GPU) against the DOP853 oracle for every family.
Bounds the discretisation error of the algorithm the product runs."""
import numpy as np
import pytest

from tests import cases


@pytest.mark.parametrize("name", list(cases.all_cases()))
def test_port_vs_truth(name):
    case = cases.all_cases()[name]
    eq, mode, m, _ = case
    port = cases.port_problem(eq, mode, m)
    truth = cases.truth_problem(eq, mode, m)
    k, W = cases.sample_kw(case, nk=3, nw=6, seed=1)
    kk = np.repeat(k, len(W))
    ww = (k[:, None] * W[None, :]).ravel()
    D, rel, st = port.eval_points(kk, ww, nthreads=4)
    n_ok = n_cont = 0
    for i in range(len(kk)):
        d, a, b, s = truth.mismatch(kk[i], ww[i])
        if s in (1, 2):
            assert st[i] == s, (name, kk[i], ww[i], st[i], s)
            continue
        if s == 3 or st[i] == 3:
            # the oracle samples 4001 points, the port the 2N-1 grid points: they may disagree only at the very
            # edge of a continuum band
            n_cont += 1
            continue
        n_ok += 1
        scale = max(abs(a), abs(b))
        # RK4 on the reference's grid vs adaptive DOP853: 4th-order discretisation error
        tol = 3e-8 * max(1.0, (1000.0 / eq.n_nodes) ** 4)
        assert abs(D[i] - d) <= tol * scale, (name, kk[i], ww[i], D[i], d)
    assert n_ok >= 4, (name, n_ok, n_cont)


def test_continuum_bands_equal_per_node_tracking(monkeypatch):
    """Cylinder (no twist) and flow slab: the phase-speed band test (four comparisons per point and term) flags exactly
    the points the per-node sign tracking flags, and leaves D untouched."""
    from eigensolver_amd import equilibrium as q
    photo = dict(c_e=1.5, vA_e=0.5, r_sign=1.0, n_nodes=1000, ic=(1e-8, 1e-8))
    k = np.linspace(0.05, 4.0, 24)
    W = np.linspace(0.03, 5.0, 400)
    n_cont = 0
    for eq, mode in [(q.CylinderFlow(U_i0=0.7, width=0.9), "kink"), (q.CylinderDensity(width=0.9), "sausage"),
                     (q.CylinderDensity(width=1.5, **photo), "kink"), (q.CylinderFlow(U_i0=-0.35, width=3.0), "sausage"),
                     (q.SlabFlow(U_i0=0.35, width=1.5), "kink"), (q.SlabFlow(U_i0=0.9, width=0.9), "sausage"),
                     (q.SlabDensity(width=1.5, n_nodes=1001), "sausage"), (q.SlabDensity(width=0.9), "kink"),
                     (q.SlabFlow(c_i0=2.0 / 3.0, vA_i0=1.0, c_e=0.75, vA_e=0.0, U_i0=0.0, U_e=-0.15, width=float("inf"),
                                 L_factor=7.0), "kink")]:
        monkeypatch.delenv("ES_FORCE_SIGN_TRACKING", raising=False)
        Db, relb, stb = cases.port_problem(eq, mode).eval_grid(k, W, w_mode=1, nthreads=8)
        monkeypatch.setenv("ES_FORCE_SIGN_TRACKING", "1")
        Dt, relt, stt = cases.port_problem(eq, mode).eval_grid(k, W, w_mode=1, nthreads=8)
        assert np.array_equal(stb, stt)
        assert np.array_equal(Db[stb == 0], Dt[stt == 0])
        n_cont += int((stb == 3).sum())
    assert n_cont > 1000


def test_numpy_grid_matches_port():
    """oracle/grid_numpy.py (vectorised NumPy restatement timed by bench.py as the second CPU baseline) against the C
    port on the bench workload in small: identical statuses, |dD| <= 1e-10 of the scale (no fma in NumPy)."""
    import numpy as np
    from eigensolver_amd import equilibrium as q, shooting as s
    from oracle.grid_numpy import CylinderGrid
    from oracle.port import PortProblem
    for eq, mode, m in ((q.CylinderFlow(U_i0=0.7, width=0.9, n_nodes=300), "kink", 1),
                        (q.CylinderFlow(U_i0=0.7, width=0.9, n_nodes=300), "sausage", 0),
                        (q.CylinderDensity(width=0.95, n_nodes=300), "kink", 3)):
        d, prof = s.make_desc(eq, mode, m)
        desc = {f[0]: getattr(d, f[0]) for f in d._fields_}
        port = PortProblem(desc, prof)
        k = np.array([0.05, 0.7, 2.1, 3.9])
        W = 0.9 + (np.arange(200) + 0.5) * (4.1 / 200)
        Dp, relp, stp = port.eval_grid(k, W, w_mode=1, nthreads=2)
        Dn, reln, stn = CylinderGrid(desc, {a: np.asarray(v) for a, v in prof.items()}).eval_grid(k, W)
        assert np.array_equal(stp, stn)
        ok = stp == 0
        assert ok.sum() > 200
        scale = np.abs(Dp[ok]) * 100.0 / relp[ok]
        assert np.max(np.abs(Dp[ok] - Dn[ok]) / scale) < 1e-10


def test_port_unnormalised_march_compensates_axis_target():
    """The port (as the HIP marches of the untwisted cylinder) carries 3^n z with a power-of-two rescale per 128 steps and
    multiplies a non-zero target of the axis condition, bc_const * xi_e, by the same factor.  The reference's own
    profiles never have one for this family (B_phi = 0), so it is forced here and checked against the NumPy
    restatement, whose march keeps z at its true scale: |dD| <= 1e-10 of the scale, for a node count that is not a
    multiple of the chunk."""
    import numpy as np
    from eigensolver_amd import equilibrium as q, shooting as s
    from oracle.grid_numpy import CylinderGrid
    from oracle.port import PortProblem
    eq = q.CylinderFlow(U_i0=0.4, width=0.9, n_nodes=331)
    d, prof = s.make_desc(eq, "kink", 1)
    desc = {f[0]: getattr(d, f[0]) for f in d._fields_}
    k = np.array([0.3, 1.7, 3.6])
    W = 0.95 + (np.arange(160) + 0.5) * (4.0 / 160)
    base = PortProblem(desc, prof).eval_grid(k, W, w_mode=1, nthreads=2)
    desc["bc_const"] = 0.37
    Dp, relp, stp = PortProblem(desc, prof).eval_grid(k, W, w_mode=1, nthreads=2)
    Dn, reln, stn = CylinderGrid(desc, {a: np.asarray(v) for a, v in prof.items()}).eval_grid(k, W)
    ok = (stp == 0) & (stn == 0)
    assert ok.sum() > 200
    scale = np.abs(Dp[ok]) * 100.0 / relp[ok]
    assert np.max(np.abs(Dp[ok] - Dn[ok]) / scale) < 1e-10
    assert np.max(np.abs(Dp[ok] - base[0][ok]) / scale) > 1e-3          # the target does enter D


@pytest.mark.parametrize("which", ["config3", "config1", "config2", "config4"])
def test_port_vs_truth_on_the_config_grids(which):
    """The same independent leg tests/test_full_size_parity_gpu.py runs for the HIP kernels, for the port on the CPU: 192
    random points of the (k, omega) grid of each GPU configuration of BASELINE.json against the adaptive DOP853 oracle (no
    RK4 grid, no code shared with kernel or port), within the discretisation bound at every point that is not within 8
    grid columns of a flagged point (band edges: a coefficient nearly vanishes at a node), 1e-3 there."""
    import bench
    from tests import truth_pool
    if which == "config3":
        eq, mode, m = bench.workload_equilibrium(), "kink", 1
        k, W = bench.workload_grid()
        kind, kw = "CylinderFlow", dict(U_i0=0.7, width=0.9)
    else:
        _, units = bench.workload_units(which)
        _, _, eq, mode, m, k, W = units[min(3, len(units) - 1)]
        kind, kw = {"config1": ("SlabFlow", dict(U_i0=0.35, width=1.5)), "config2": ("CylinderDensity", dict(width=0.95)),
                    "config4": ("CylinderRotation", dict(v_twist=0.1, power=1.0, r_axis=0.001))}[which]
    port = cases.port_problem(eq, mode, m)
    rng = np.random.default_rng(99)
    n = 192
    ii, jj = rng.integers(0, len(k), n), rng.integers(0, len(W), n)
    kk, ww = k[ii], k[ii] * W[jj]
    D, rel, st = port.eval_points(kk, ww, nthreads=4)
    # statuses of the neighbouring grid columns of every sampled point (8 to either side)
    near = np.zeros(n, dtype=bool)
    for t in range(n):
        cols = np.arange(max(0, jj[t] - 8), min(len(W), jj[t] + 9))
        _, _, stn = port.eval_points(np.full(len(cols), k[ii[t]]), k[ii[t]] * W[cols], nthreads=1)
        near[t] = (stn != 0).any()
    tr = truth_pool.evaluate(kind, kw, mode, m, kk, ww, 4)
    d, a, b, s = tr[:, 0], tr[:, 1], tr[:, 2], tr[:, 3].astype(int)
    both = (s == 0) & (st == 0)
    differ = s != st
    assert np.all((s[differ] == 3) | (st[differ] == 3))
    assert both.sum() > 0.5 * n
    tol = 3e-8 * max(1.0, (1000.0 / eq.n_nodes) ** 4)
    err = np.abs(D - d) / np.maximum(np.abs(a), np.abs(b))
    far, edge = both & ~near, both & near
    assert err[far].max() <= tol, (which, err[far].max())
    assert not edge.any() or err[edge].max() <= 1e-3
