"""One parity test per BASELINE.json configuration that runs on the GPU (configs[1], [2], [4]; configs[3] is the
bench workload, covered by test_shoot_gpu.py::test_full_size_properties)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import cases  # noqa: E402


def _sample_vs_port(gp, eq, mode, m, k, W, D, st, n=400, seed=0):
    port = cases.port_problem(eq, mode, m)
    rng = np.random.default_rng(seed)
    ii, jj = rng.integers(0, len(k), n), rng.integers(0, len(W), n)
    Dp, relp, stp = port.eval_points(k[ii], k[ii] * W[jj], nthreads=8)
    a = D[ii, jj]
    assert np.array_equal(st[ii, jj], stp)
    ok = stp == 0
    sc = np.abs(Dp[ok]) * 100.0 / relp[ok]
    assert (np.abs(a[ok] - Dp[ok]) / sc).max() < 1e-12
    return ok.sum()


def test_config1_slab_flow_1024x1024(es_ctx):
    """configs[1]: Slab / non-uniform flow, 1024 x 1024 (k, omega) grid, fp64."""
    from eigensolver_amd import ShootProblem, equilibrium as q
    eq = q.SlabFlow(U_i0=0.35, width=1.5)
    n = 1024
    k = np.linspace(0.05, 3.5, n)
    W = 1.4 + (np.arange(n) + 0.5) * (2.45 - 1.4) / n
    for mode in ("sausage", "kink"):
        gp = ShootProblem(eq, mode, ctx=es_ctx)
        D, st = gp.eval_grid(k, W)
        roots, cnt = gp.find_roots(k, W, D, st, n_bisect=30, tol_percent=1e-3)
        Dn, stn = D.cpu().numpy(), st.cpu().numpy()
        assert _sample_vs_port(gp, eq, mode, None, k, W, Dn, stn) > 100
        r = {a: v.cpu().numpy() for a, v in roots.items()}
        acc = r["flag"] == 1
        assert acc.sum() > 500
        assert np.all(np.diff(r["row"]) >= 0)
        # refined roots against the CPU port's own refinement of the same brackets
        port = cases.port_problem(eq, mode)
        sel = np.where(acc)[0][:: max(1, acc.sum() // 60)]
        for i in sel:
            Dc, relc, stc = port.eval_points([r["k"][i]], [r["w"][i]])
            assert stc[0] == 0 and relc[0] < 1e-3
        gp.close()


def test_config2_cylinder_density_m0_to_4_4096_k(es_ctx):
    """configs[2]: Cylinder / non-uniform density, m = 0..4, 4096 k-points, fp64."""
    from eigensolver_amd import ShootProblem, equilibrium as q
    eq = q.CylinderDensity(width=0.95)
    k = np.linspace(0.01, 4.5, 4096)
    W = 2.05 + (np.arange(384) + 0.5) * (4.95 - 2.05) / 384
    total = 0
    for m in range(0, 5):
        mode = "sausage" if m == 0 else "kink"
        gp = ShootProblem(eq, mode, m=m, ctx=es_ctx)
        D, st = gp.eval_grid(k, W)
        roots, cnt = gp.find_roots(k, W, D, st, n_bisect=30, tol_percent=1e-3)
        _sample_vs_port(gp, eq, mode, m, k, W, D.cpu().numpy(), st.cpu().numpy(), n=250, seed=m)
        acc = int((roots["flag"] == 1).sum())
        total += acc
        gp.close()
    assert total > 4096                 # at least one mode per k summed over the five orders


def test_config4_cylinder_rotation_m0_to_10(es_ctx):
    """configs[4]: Cylinder / rotational flow, m = 0..10 (fp64 throughout; the reference has neither fp32 nor Newton)."""
    from eigensolver_amd import ShootProblem, equilibrium as q
    k = np.linspace(0.25, 4.0, 192)
    W = 0.7 + (np.arange(256) + 0.5) * (1.45 - 0.7) / 256
    n_acc = 0
    for m in range(0, 11):
        mode = "sausage" if m == 0 else "kink"
        eq = q.CylinderRotation(v_twist=0.1, power=1.0, r_axis=0.01 if m == 0 else 0.001)
        gp = ShootProblem(eq, mode, m=m, ctx=es_ctx)
        D, st = gp.eval_grid(k, W)
        roots, cnt = gp.find_roots(k, W, D, st, n_bisect=30, tol_percent=1e-3)
        _sample_vs_port(gp, eq, mode, m, k, W, D.cpu().numpy(), st.cpu().numpy(), n=200, seed=m)
        n_acc += int((roots["flag"] == 1).sum())
        gp.close()
    assert n_acc > 100
