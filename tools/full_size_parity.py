"""GPU box: the WHOLE bench grid (4096 x 4096, BASELINE configs[3]) against the CPU port, point by point -- statuses,
D, rel -- and the root table of the grid search against the port's.  Writes gpurun_out/full_size_parity.json.
    python tools/full_size_parity.py [threads]
    python tools/full_size_parity.py configs12 [threads] -> BASELINE configs[1] and configs[2], fp64 grid + grid search; writes
                                                            gpurun_out/full_size_parity_configs12.json
    python tools/full_size_parity.py config4 [threads]   -> BASELINE configs[4] (rotational flow, m = 0..10, 1024 x 1024 per order,
                                                            N = 2000): the fp64 grid and the MIXED search of every order
                                                            against the port; writes gpurun_out/full_size_parity_config4.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tests import cases  # noqa: E402


def config4(nthreads):
    import torch
    from eigensolver_amd import ShootProblem, _lib, equilibrium as q
    n = 1024
    k = np.linspace(0.25, 4.0, n)
    W = 0.7 + (np.arange(n) + 0.5) * ((1.45 - 0.7) / n)
    kt = torch.as_tensor(k, device="cuda"); Wt = torch.as_tensor(W, device="cuda")
    ctx = _lib.Context(0)
    rows = []
    for m in range(11):
        eq = q.CylinderRotation(v_twist=0.1, power=1.0, r_axis=0.01 if m == 0 else 0.001)
        mode = "sausage" if m == 0 else "kink"
        port = cases.port_problem(eq, mode, m)
        t0 = time.time()
        Dp, relp, stp = port.eval_grid(k, W, w_mode=1, nthreads=nthreads)
        rp, cntp = port.find_roots(k, W, Dp, stp, w_mode=1, n_bisect=bench.N_BISECT, tol=bench.TOL_PERCENT, nthreads=nthreads)
        t_port = time.time() - t0
        gp = ShootProblem(eq, mode, m=m, ctx=ctx)
        D, st = gp.eval_grid(kt, Wt)
        Dg, stg = D.cpu().numpy(), st.cpu().numpy()
        ok = stp == 0
        scale = np.abs(Dp[ok]) * 100.0 / relp[ok]
        err = np.abs(Dg[ok] - Dp[ok]) / scale
        sd = np.signbit(Dg[ok]) != np.signbit(Dp[ok])
        roots, cnt, _, stm, stats = gp.find_roots_mixed(kt, Wt, n_bisect=bench.N_BISECT, tol_percent=bench.TOL_PERCENT)
        g = {a: v.cpu().numpy() for a, v in roots.items()}
        same = cnt == cntp and np.array_equal(g["row"], rp["row"]) and np.array_equal(g["flag"], rp["flag"])
        dw = np.abs(g["w"] - rp["w"]) / np.abs(rp["w"]) if same else np.array([np.inf])
        row = {"m": m, "port_seconds": t_port, "statuses_identical": bool(np.array_equal(stg, stp)),
               "mixed_statuses_identical": bool(np.array_equal(stm.cpu().numpy(), stp)),
               "status_histogram": np.bincount(stp.ravel(), minlength=4)[:4].tolist(),
               "max_abs_dD_over_scale": float(err.max()), "sign_differences": int(sd.sum()),
               "max_abs_D_over_scale_at_sign_differences": float((np.abs(Dp[ok][sd]) / scale[sd]).max()) if sd.any() else 0.0,
               "brackets": [int(cnt), int(cntp)], "mixed_bracket_rows_and_flags_identical": bool(same),
               "accepted_roots": int((rp["flag"] == 1).sum()), "max_rel_root_difference": float(dw.max()),
               "fp64_reevaluations": [int(x) for x in stats]}
        print(json.dumps(row), flush=True)
        rows.append(row)
        gp.close()
    out = {"orders": rows, "all_statuses_identical": all(r["statuses_identical"] and r["mixed_statuses_identical"] for r in rows),
           "all_bracket_tables_identical": all(r["mixed_bracket_rows_and_flags_identical"] for r in rows),
           "max_abs_dD_over_scale": max(r["max_abs_dD_over_scale"] for r in rows),
           "max_rel_root_difference": max(r["max_rel_root_difference"] for r in rows),
           "sign_differences": sum(r["sign_differences"] for r in rows),
           "brackets": sum(r["brackets"][0] for r in rows), "accepted_roots": sum(r["accepted_roots"] for r in rows)}
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "full_size_parity_config4.json"), "w"), indent=1)
    print(json.dumps({a: v for a, v in out.items() if a != "orders"}, indent=1))


def compare(gp, port, k, W, n_bisect, tol, nthreads, label):
    """fp64 grid and grid search of one problem, GPU against the port, every point."""
    import torch
    t0 = time.time()
    Dp, relp, stp = port.eval_grid(k, W, w_mode=1, nthreads=nthreads)
    rp, cntp = port.find_roots(k, W, Dp, stp, w_mode=1, n_bisect=n_bisect, tol=tol, nthreads=nthreads)
    t_port = time.time() - t0
    kt = torch.as_tensor(k, device="cuda"); Wt = torch.as_tensor(W, device="cuda")
    D, st = gp.eval_grid(kt, Wt)
    roots, cnt = gp.find_roots(kt, Wt, D, st, n_bisect=n_bisect, tol_percent=tol)
    Dg, stg = D.cpu().numpy(), st.cpu().numpy()
    ok = stp == 0
    scale = np.abs(Dp[ok]) * 100.0 / relp[ok]
    err = np.abs(Dg[ok] - Dp[ok]) / scale
    sd = np.signbit(Dg[ok]) != np.signbit(Dp[ok])
    g = {a: v.cpu().numpy() for a, v in roots.items()}
    same = cnt == cntp and np.array_equal(g["row"], rp["row"]) and np.array_equal(g["flag"], rp["flag"])
    dw = np.abs(g["w"] - rp["w"]) / np.abs(rp["w"]) if (same and cnt) else np.array([0.0 if same else np.inf])
    row = {"problem": label, "points": int(Dp.size), "port_seconds": t_port, "statuses_identical": bool(np.array_equal(stg, stp)),
           "status_histogram": np.bincount(stp.ravel(), minlength=4)[:4].tolist(),
           "max_abs_dD_over_scale": float(err.max()) if ok.any() else 0.0, "sign_differences": int(sd.sum()),
           "max_abs_D_over_scale_at_sign_differences": float((np.abs(Dp[ok][sd]) / scale[sd]).max()) if sd.any() else 0.0,
           "brackets": [int(cnt), int(cntp)], "bracket_rows_and_flags_identical": bool(same),
           "accepted_roots": int((rp["flag"] == 1).sum()), "max_rel_root_difference": float(dw.max())}
    print(json.dumps(row), flush=True)
    return row


def configs12(nthreads):
    """BASELINE configs[1] (slab / non-uniform flow, 1024 x 1024, both modes) and configs[2] (cylinder / non-uniform density,
    m = 0..4, 4096 k-points x 384 omega) as tests/test_configs_gpu.py defines them, at full size."""
    from eigensolver_amd import ShootProblem, _lib, equilibrium as q
    ctx = _lib.Context(0)
    rows = []
    eq = q.SlabFlow(U_i0=0.35, width=1.5)
    k = np.linspace(0.05, 3.5, 1024)
    W = 1.4 + (np.arange(1024) + 0.5) * (2.45 - 1.4) / 1024
    for mode in ("sausage", "kink"):
        gp = ShootProblem(eq, mode, ctx=ctx)
        rows.append(compare(gp, cases.port_problem(eq, mode), k, W, 30, 1e-3, nthreads, f"configs[1] slab flow {mode}"))
        gp.close()
    eq = q.CylinderDensity(width=0.95)
    k = np.linspace(0.01, 4.5, 4096)
    W = 2.05 + (np.arange(384) + 0.5) * (4.95 - 2.05) / 384
    for m in range(5):
        mode = "sausage" if m == 0 else "kink"
        gp = ShootProblem(eq, mode, m=m, ctx=ctx)
        rows.append(compare(gp, cases.port_problem(eq, mode, m), k, W, 30, 1e-3, nthreads, f"configs[2] cylinder density m={m}"))
        gp.close()
    out = {"problems": rows, "all_statuses_identical": all(r["statuses_identical"] for r in rows),
           "all_bracket_tables_identical": all(r["bracket_rows_and_flags_identical"] for r in rows),
           "points": sum(r["points"] for r in rows), "sign_differences": sum(r["sign_differences"] for r in rows),
           "max_abs_D_over_scale_at_sign_differences": max(r["max_abs_D_over_scale_at_sign_differences"] for r in rows),
           "max_abs_dD_over_scale": max(r["max_abs_dD_over_scale"] for r in rows),
           "max_rel_root_difference": max(r["max_rel_root_difference"] for r in rows),
           "brackets": sum(r["brackets"][0] for r in rows), "accepted_roots": sum(r["accepted_roots"] for r in rows)}
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "full_size_parity_configs12.json"), "w"), indent=1)
    print(json.dumps({a: v for a, v in out.items() if a != "problems"}, indent=1))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "configs12":
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        return configs12(int(sys.argv[2]) if len(sys.argv) > 2 else bench.host_cores())
    if len(sys.argv) > 1 and sys.argv[1] == "config4":
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        return config4(int(sys.argv[2]) if len(sys.argv) > 2 else bench.host_cores())
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
