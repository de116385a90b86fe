"""GPU box: the WHOLE bench grid (4096 x 4096, BASELINE configs[3]) against the CPU port, point by point -- statuses,
D, rel -- and the root table of the grid search against the port's.  Writes gpurun_out/full_size_parity.json.
    python tools/full_size_parity.py [threads]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tests import cases  # noqa: E402


def main():
    nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else bench.host_cores()
    eq = bench.workload_equilibrium()
    k, W = bench.workload_grid()
    port = cases.port_problem(eq, "kink", 1)
    t0 = time.time()
    Dp, relp, stp = port.eval_grid(k, W, w_mode=1, nthreads=nthreads)
    t_port = time.time() - t0
    print(f"port: {Dp.size} points in {t_port:.1f} s on {nthreads} threads", flush=True)
    import torch
    from eigensolver_amd import ShootProblem, _lib
    ctx = _lib.Context(0)
    gp = ShootProblem(eq, "kink", m=1, ctx=ctx)
    kt = torch.as_tensor(k, device="cuda"); Wt = torch.as_tensor(W, device="cuda")
    D, st, rel = gp.eval_grid(kt, Wt, want_rel=True)
    roots, cnt = gp.find_roots(kt, Wt, D, st, n_bisect=bench.N_BISECT, tol_percent=bench.TOL_PERCENT)
    Dg, stg, relg = D.cpu().numpy(), st.cpu().numpy(), rel.cpu().numpy()
    ok = stp == 0
    scale = np.abs(Dp[ok]) * 100.0 / relp[ok]
    err = np.abs(Dg[ok] - Dp[ok]) / scale
    sign_diff = np.signbit(Dg[ok]) != np.signbit(Dp[ok])
    rp, cntp = port.find_roots(k, W, Dp, stp, w_mode=1, n_bisect=bench.N_BISECT, tol=bench.TOL_PERCENT, nthreads=nthreads)
    g = {n: v.cpu().numpy() for n, v in roots.items()}
    same_rows = cnt == cntp and np.array_equal(g["row"], rp["row"]) and np.array_equal(g["flag"], rp["flag"])
    dw = np.abs(g["w"] - rp["w"]) / np.abs(rp["w"]) if same_rows else np.array([np.inf])
    acc = rp["flag"] == 1
    out = {"grid": [len(k), len(W)], "points": int(Dp.size), "port_seconds": t_port, "port_threads": nthreads,
           "statuses_identical": bool(np.array_equal(stg, stp)),
           "status_histogram": np.bincount(stp.ravel(), minlength=4)[:4].tolist(),
           "max_abs_dD_over_scale": float(err.max()), "p999_abs_dD_over_scale": float(np.quantile(err, 0.999)),
           "bitwise_equal_D_fraction": float(np.mean(Dg[ok] == Dp[ok])),
           "sign_differences": int(sign_diff.sum()),
           "max_abs_D_over_scale_at_sign_differences": float((np.abs(Dp[ok][sign_diff]) / scale[sign_diff]).max()) if sign_diff.any() else 0.0,
           "nan_where_flagged": bool(np.all(np.isnan(Dg[(stp == 1) | (stp == 2)]))),
           "brackets": [int(cnt), int(cntp)], "bracket_rows_and_flags_identical": bool(same_rows),
           "accepted_roots": int(acc.sum()), "max_rel_root_difference_accepted": float(dw[acc].max()) if same_rows else None,
           "max_rel_root_difference_all": float(dw.max())}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "full_size_parity.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
