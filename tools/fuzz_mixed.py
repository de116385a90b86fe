"""GPU fuzz of the fp32-screened grid search (es_shoot_find_roots_mixed) against the fp64 path over random cylinder and
(round 3) slab problems: profile family, widths / amplitudes / twist, azimuthal order, mode, (k, omega) windows and grid sizes.  Every
case must give the identical bracket count, a bit-identical root table and identical statuses; prints the worst fp32
error among the points fp32 vouched for and the largest re-evaluated fraction.   python tools/fuzz_mixed.py [n_cases [seed]]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigensolver_amd import ShootProblem, _lib, equilibrium as q  # noqa: E402


def main(n_cases, seed=2024):
    rng = np.random.default_rng(seed)
    ctx = _lib.Context(0)
    worst_err, worst_frac, bad = 0.0, 0.0, 0
    worst_err_D = 0.0
    for c in range(n_cases):
        fam = rng.integers(0, 6)
        m = int(rng.integers(0, 7))
        mode = "sausage" if m == 0 else "kink"
        if fam == 4:
            eq = q.SlabDensity(width=float(rng.choice([0.9, 1.5, 1e5])), n_nodes=int(rng.choice([301, 1001])))
            lo, hi, m = 0.85, 1.3, None
            mode = str(rng.choice(["sausage", "kink"]))
        elif fam == 5:
            eq = q.SlabFlow(U_i0=float(rng.uniform(0.0, 0.5)), width=float(rng.choice([0.9, 1.5, 1e5])))
            lo, hi, m = 1.05, 2.45, None
            mode = str(rng.choice(["sausage", "kink"]))
        elif fam == 0:
            eq = q.CylinderFlow(U_i0=float(rng.uniform(0.0, 0.9)), width=float(rng.choice([0.6, 0.9, 1.5, 1e5])),
                                n_nodes=int(rng.choice([300, 1000])))
            lo, hi = 0.9, 4.95
        elif fam == 1:
            eq = q.CylinderDensity(width=float(rng.choice([0.9, 0.95, 1.5, 3.0])), n_nodes=int(rng.choice([500, 777])))
            lo, hi = 0.9, 4.95
        elif fam == 2:
            eq = q.CylinderDensity(width=float(rng.choice([0.9, 1.5])), c_e=1.5, vA_e=0.5, r_sign=1.0, n_nodes=1000, ic=(1e-8, 1e-8))
            lo, hi = 0.52, 1.48
        else:
            eq = q.CylinderRotation(v_twist=float(rng.choice([0.05, 0.1, 0.15, 0.25])), power=float(rng.choice([0.8, 1.0, 1.25])),
                                    r_axis=0.01 if m == 0 else 0.001, n_nodes=int(rng.choice([1000, 2000])))
            lo, hi = 0.7, 1.45
        nk, nw = int(rng.integers(3, 40)), int(rng.integers(70, 1300))
        k = np.sort(rng.uniform(0.05, 4.2, nk))
        a, b = np.sort(rng.uniform(lo, hi, 2))
        if b - a < 0.05 * (hi - lo):
            a, b = lo, hi
        W = a + (np.arange(nw) + 0.5) * (b - a) / nw
        gp = ShootProblem(eq, mode, m=m, ctx=ctx)
        D, st, rel = gp.eval_grid(k, W, want_rel=True)
        r64, c64 = gp.find_roots(k, W, D, st, n_bisect=20)
        try:
            rmx, cmx, Dm, stm, stats = gp.find_roots_mixed(k, W, n_bisect=20)
        except _lib.EsError as e:
            bad += 1
            print(f"case {c}: {type(eq).__name__} m={m} {mode}: {e}", flush=True)
            gp.close()
            continue
        same = cmx == c64 and all(np.array_equal(r64[n].cpu().numpy(), rmx[n].cpu().numpy(), equal_nan=True) for n in r64)
        same = same and torch.equal(stm, st)
        D, rel, Dm, stn = D.cpu().numpy(), rel.cpu().numpy(), Dm.cpu().numpy(), st.cpu().numpy()
        ok = (stn == 0) & (Dm != D)
        err = float(np.max(np.abs(Dm[ok] - D[ok]) / (np.abs(D[ok]) * 100.0 / rel[ok]))) if ok.any() else 0.0
        err_D = float(np.max(np.abs(Dm[ok] - D[ok]) / np.abs(D[ok]))) if ok.any() else 0.0     # what a vouched-for sign depends on
        worst_err_D = max(worst_err_D, err_D)
        frac = stats[0] / D.size
        worst_err, worst_frac = max(worst_err, err), max(worst_frac, frac)
        if err > 5e-3 and not type(eq).__name__.startswith('Slab'):
            i = np.argmax(np.where(ok, np.abs(Dm - D) / (np.abs(D) * 100.0 / np.where(rel > 0, rel, 1.0)), 0.0))
            ik, iw = np.unravel_index(i, D.shape)
            print(f"case {c}: err {err:.2e} {type(eq).__name__}({getattr(eq, 'width', None)}, U={getattr(eq, 'U_i0', None)}, vt={getattr(eq, 'v_twist', None)}, "
                  f"p={getattr(eq, 'power', None)}, N={eq.n_nodes}) m={m} {mode} at k={k[ik]:.4f} W={W[iw]:.5f} rel64={rel[ik, iw]:.3g} "
                  f"D64={D[ik, iw]:.4g} D32={Dm[ik, iw]:.4g}", flush=True)
        if not same:
            which = [n for n in r64 if not np.array_equal(r64[n].cpu().numpy(), rmx[n].cpu().numpy(), equal_nan=True)]
            nst = int((stm != st).sum())
            print(f"   differing arrays {which}, differing statuses {nst}", flush=True)
            bad += 1
            print(f"case {c}: MISMATCH {type(eq).__name__} m={m} {mode} nk={nk} nw={nw} window=({a:.3f},{b:.3f}) "
                  f"brackets {c64}/{cmx} stats {stats}", flush=True)
        gp.close()
    print(f"{n_cases} cases, {bad} failures, worst fp32 error {worst_err:.2e} of the scale (margin 5e-2), "
          f"largest re-evaluated fraction {worst_frac:.3f}; worst fp32 error relative to |D| at a vouched-for point {worst_err_D:.3f}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 2024))
