"""GPU fuzz of the fp64 grid path against the CPU port (oracle/c/shoot_port.c) over random problems of all four
families: statuses identical, |dD| <= 1e-12 of the scale at every ES_PT_OK point, ES_EVAL_SKIP_CONTINUUM identical
outside the continuum, grid search identical to the port's (bracket rows, flags, roots to 1e-10).
    python tools/fuzz_grid.py [n_cases [seed [n_truth]]]
n_truth > 0: an INDEPENDENT leg for the first n_truth cases -- two evaluated points of each (at least 8 columns away from any
flagged point of their row) against the adaptive DOP853 oracle (oracle/cylinder.py, oracle/slab.py: no RK4 grid, no code
shared with kernel or port), within 4 x the discretisation figure 3e-8 (1000 / N)^4 of the scale that the fixed test cases
meet (random problems reach 1.4 x it: rotation profile v_phi ~ r^0.8, whose derivative is unbounded at the axis); the ratio
to that figure is printed."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eigensolver_amd import ShootProblem, _lib, equilibrium as q  # noqa: E402
from tests import cases  # noqa: E402


def random_problem(rng):
    fam = int(rng.integers(0, 6))
    m = int(rng.integers(0, 5))
    mode = "sausage" if m == 0 else "kink"
    if fam == 0:
        return q.CylinderFlow(U_i0=float(rng.uniform(0.0, 0.9)), width=float(rng.choice([0.6, 0.9, 1.5, 1e5])),
                              n_nodes=int(rng.choice([130, 500, 1000]))), mode, m, (0.9, 4.95)
    if fam == 1:
        return q.CylinderDensity(width=float(rng.choice([0.9, 0.95, 1.5, 3.0])), n_nodes=int(rng.choice([257, 500]))), mode, m, (0.9, 4.95)
    if fam == 2:
        return q.CylinderRotation(v_twist=float(rng.choice([0.05, 0.1, 0.25])), power=float(rng.choice([0.8, 1.0, 1.25])),
                                  r_axis=0.01 if m == 0 else 0.001, n_nodes=int(rng.choice([500, 2000]))), mode, m, (0.7, 1.45)
    mode = str(rng.choice(["sausage", "kink"]))
    if fam == 3:
        return q.SlabDensity(width=float(rng.choice([0.9, 1.5, 1e5])), n_nodes=int(rng.choice([301, 1001]))), mode, None, (0.85, 1.3)
    if fam == 4:
        return q.SlabDensity(width=float(rng.choice([0.9, 1.5])), vA_i0=1.2, vA_e=3.0, c_e=0.4, L_factor=3.0,
                             n_nodes=501), mode, None, (0.75, 1.3)
    return q.SlabFlow(U_i0=float(rng.uniform(0.0, 0.5)), width=float(rng.choice([0.9, 1.5, 1e5]))), mode, None, (-2.45, 2.45)


def main(n_cases, seed=5, n_truth=0):
    import torch
    rng = np.random.default_rng(seed)
    ctx = _lib.Context(0)
    worst, bad = 0.0, 0
    truth_worst, truth_n, truth_bad = 0.0, 0, 0
    for c in range(n_cases):
        eq, mode, m, (lo, hi) = random_problem(rng)
        nk, nw = int(rng.integers(2, 9)), int(rng.integers(40, 400))
        if rng.uniform() < 0.2:               # wide rows: the 4-points-per-lane launch shapes, several segments per row
            nk, nw = int(rng.integers(2, 4)), int(rng.integers(1024, 5200))
        k = np.sort(rng.uniform(0.05, 4.2, nk))
        a, b = np.sort(rng.uniform(lo, hi, 2))
        if b - a < 0.05 * (hi - lo):
            a, b = lo, hi
        W = a + (np.arange(nw) + 0.5) * (b - a) / nw
        W = W[np.abs(W) > 1e-3]
        gp = ShootProblem(eq, mode, m=m, ctx=ctx)
        port = cases.port_problem(eq, mode, m)
        D, st, rel = (t.cpu().numpy() for t in gp.eval_grid(k, W, want_rel=True))
        Ds, ss = (t.cpu().numpy() for t in gp.eval_grid(k, W, skip_continuum=True))
        Dp, relp, stp = port.eval_grid(k, W, w_mode=1, nthreads=8)
        ok = stp == 0
        msg = []
        if not np.array_equal(st, stp):
            msg.append(f"{int((st != stp).sum())} statuses differ")
        if not np.array_equal(ss, st) or not np.array_equal(Ds[st != 3], D[st != 3], equal_nan=True):
            msg.append("skip-continuum output differs")
        if ok.any():
            err = float(np.max(np.abs(D[ok] - Dp[ok]) / (np.abs(Dp[ok]) * 100.0 / relp[ok])))
            worst = max(worst, err)
            if not err < 1e-12:
                msg.append(f"|dD|/scale = {err:.2e}")
        r, cnt = gp.find_roots(k, W, torch.as_tensor(D, device="cuda"), torch.as_tensor(st, device="cuda"), n_bisect=20)
        rp, cntp = port.find_roots(k, W, Dp, stp, w_mode=1, n_bisect=20, tol=1e-3, nthreads=8)
        if cnt != cntp or not np.array_equal(r["row"].cpu().numpy(), rp["row"]) or not np.array_equal(r["flag"].cpu().numpy(), rp["flag"]):
            msg.append(f"bracket tables differ ({cnt} / {cntp})")
        elif cnt:
            acc = rp["flag"] == 1
            dw = np.abs(r["w"].cpu().numpy() - rp["w"]) / np.abs(rp["w"])
            if acc.any() and np.max(dw[acc]) > 1e-10:
                i = int(np.argmax(np.where(acc, dw, 0)))
                msg.append(f"roots differ by more than 1e-10: d={dw[i]:.2e} at k={rp['k'][i]:.4f} w={rp['w'][i]:.6f} "
                           f"[{rp['w_lo'][i]:.6f}, {rp['w_hi'][i]:.6f}] resid gpu {float(r['resid'][i]):.3e} port {rp['resid'][i]:.3e} "
                           f"bracket width gpu {float(r['w_hi'][i] - r['w_lo'][i]):.2e} port {rp['w_hi'][i] - rp['w_lo'][i]:.2e}")
        if c < n_truth and ok.any():
            flagged = st != 0
            cand = [(i, j) for i in range(len(k)) for j in range(8, len(W) - 8)
                    if ok[i, j] and not flagged[i, j - 8:j + 9].any()]
            if cand:
                truth = cases.truth_problem(eq, mode, m)
                tol = 3e-8 * max(1.0, (1000.0 / eq.n_nodes) ** 4)
                for t in rng.choice(len(cand), size=min(2, len(cand)), replace=False):
                    i, j = cand[int(t)]
                    d, a_, b_, s_ = truth.mismatch(k[i], k[i] * W[j])
                    if s_ != 0:
                        continue
                    e = abs(D[i, j] - d) / max(abs(a_), abs(b_))
                    truth_n += 1
                    truth_worst = max(truth_worst, e / tol)
                    if e > 4.0 * tol:
                        truth_bad += 1
                        msg.append(f"DOP853 leg: |dD|/scale {e:.2e} > 4 x {tol:.1e} at k={k[i]:.4f} W={W[j]:.5f}")
        if msg:
            bad += 1
            print(f"case {c}: {type(eq).__name__} {eq} {mode} m={m}: " + "; ".join(msg), flush=True)
        gp.close()
    print(f"{n_cases} cases, {bad} failures, worst |dD|/scale {worst:.2e}")
    if n_truth:
        print(f"DOP853 leg: {truth_n} points of {min(n_truth, n_cases)} cases, {truth_bad} above 4 x the bound, worst error / bound {truth_worst:.3f}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 150, int(sys.argv[2]) if len(sys.argv) > 2 else 5,
                  int(sys.argv[3]) if len(sys.argv) > 3 else 0))
