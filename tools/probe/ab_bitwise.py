"""GPU box, same-call A/B of two builds of the library (lib/libeigensolver_amd.so against lib/libeigensolver_amd_prev.so, a
build of the previous commit copied there by hand): D, statuses and root tables of a few problems must be bit-identical."""
import sys, numpy as np
sys.path.insert(0, '.')
from eigensolver_amd import ShootProblem, _lib
from tests.test_shoot_gpu import CASES
out = {}
for tag, path in (("new", "eigensolver_amd/lib/libeigensolver_amd.so"), ("prev", "eigensolver_amd/lib/libeigensolver_amd_prev.so")):
    ctx = _lib.Context(0, lib=_lib.load_variant(path))
    res = []
    for name in sys.argv[1:] or ["CF_flow_kink", "CF_flow_sausage", "CF_flow_m3", "CDC_w095_kink", "CDP_kink", "CF_uniform_kink"]:
        eq, mode, m, (lo, hi) = CASES[name]
        gp = ShootProblem(eq, mode, m, ctx=ctx)
        k = np.linspace(0.2, 4.0, 16); W = lo + (np.arange(2300) + 0.5) * (hi - lo) / 2300
        D, st, rel = gp.eval_grid(k, W, want_rel=True)
        r, c = gp.find_roots(k, W, D, st, n_bisect=16)
        kk = np.repeat(k, 8); ww = kk * np.tile(W[::300][:8], len(k))
        Dq, sq = gp.eval_points(kk, ww)
        res.append((D.cpu().numpy(), st.cpu().numpy(), rel.cpu().numpy(), r["w"].cpu().numpy(), r["flag"].cpu().numpy(), Dq.cpu().numpy()))
        gp.close()
    out[tag] = res
same = all(np.array_equal(a, b, equal_nan=True) for ra, rb in zip(out["new"], out["prev"]) for a, b in zip(ra, rb))
print("bitwise identical D / status / rel / roots / flags / point evaluations:", same, "grid points", sum(r[0].size for r in out["new"]), "roots", sum(r[3].size for r in out["new"]))
