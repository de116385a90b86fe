#!/usr/bin/env python
"""bench.py -- det(M) evaluations/s and roots/s of the dispersion-relation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N > 1: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[3], the configuration the metric is quoted on): Cylinder / non-uniform
(Gaussian) axial flow, coronal constants of Cylinder_method_flow_testing.py:69-72, ONE 4096 x 4096 (k, omega) grid,
fp64; k = linspace(0.01, 4, 4096), omega = k * W, W half-cell centred in (cT_i0, vA_e).
One "step" = one pass of the hot path over that grid: D(k, omega) at every grid point (HIP propagator), bracket
detection (wave shuffle + ballot), 9-section + secant refinement, ordered root compaction.
The kernels run on torch's current stream of the device (the stream the es_context is created with), so the
torch.cuda.Event pairs around the grid launch time exactly that kernel.

Multi-GPU (default --mode strong): the 4096 k-rows of the ONE grid are tiled across the ranks (strided: rank r owns
rows r, r + N, ... -- the reference's per-k process fan-out, Density_cylinder.py:1142-1153), no data-path collective;
the only exchange is one RCCL all-gather of the fixed-capacity root tables per step.  --mode weak-m is the (k, m)
tiling of round 1: rank r solves azimuthal order m = r + 1 on the full grid.
With --gpus N > 1 and no WORLD_SIZE in the environment this process only starts the N ranks
(python -m torch.distributed.run ...) as a child process and never touches the GPU itself.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NK = NW = 4096
W_LO, W_HI = 0.8944271909999159, 5.0          # (cT_i0, vA_e) of the coronal cylinder
N_BISECT = 16                                 # bracket narrowed by >= 2^16 (6 rounds of 9-section: 9^6 = 5.3e5), then
REFINE_ROUNDS = 6                             # two regula-falsi polish steps -> |d omega/omega| ~ 1e-16 (8 evals per round)
REFINE_POLISH = 2
TOL_PERCENT = 1e-3
EXCHANGE_CAP = 1 << 15                        # records per rank in the fixed-capacity all-gather (6 doubles each)
HBM_PEAK_GBS = 8000.0                         # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VALU_PEAK_TFLOPS = 78.6                  # fp64 vector peak (spec)
# algorithmic traffic per det-eval of the grid kernel: 8 B of D written + amortised inputs; the kernel
# also writes a 1-byte status per point, reported separately
BYTES_PER_EVAL = 8.0 + 16.0 * (1.0 / NW + 1.0 / NK)
BYTES_PER_EVAL_WITH_STATUS = BYTES_PER_EVAL + 1.0
# fp64 operations per det-eval of the grid kernel (FMA = 2, division = 1), see DESIGN.md "kernel K3"
FLOPS_PER_STEP = 2 * 9 + 8 + 32             # 2 coefficient sets (1 add, 3 fma, 2 mul each) + shared reciprocal (1 div, 3 mul, 2 fma) + one adjoint RK4 step (32)


def workload_equilibrium():
    from eigensolver_amd import equilibrium as q
    # Gaussian flow of width 0.9 and amplitude 0.35 vA_i0 (CF:126-135 with the author's commented values)
    return q.CylinderFlow(U_i0=0.7, width=0.9)


def workload_grid():
    k = np.linspace(0.01, 4.0, NK)
    W = W_LO + (np.arange(NW, dtype=np.float64) + 0.5) * ((W_HI - W_LO) / NW)
    return k, W


def measured_traffic_per_launch():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE collected in separate runs of this same command, profiles/README.md) and where they come from."""
    path = os.path.join(ROOT, "profiles", "bench_pmc_hbm_latest.json")
    try:
        d = json.load(open(path))
        f = [v["mean_KB_per_dispatch"] for k, v in d["FETCH_SIZE"].items() if "shoot_grid_kernel" in k][0]
        w = [v["mean_KB_per_dispatch"] for k, v in d["WRITE_SIZE"].items() if "shoot_grid_kernel" in k][0]
        # gfx950: FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads (x2 correction of the guide);
        # the reads of this kernel are 8-B scalar/LDS-staging loads of a 56 KB table, left uncorrected
        return (f + w) * 1024.0, "profiles/bench_pmc_hbm_latest.json (" + str(d.get("round", "?")) + ")"
    except Exception:
        return None, None


def host_cores():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box hands each
    job a share of the host, e.g. 16 of 256 hardware threads)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]              # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())        # cgroup v1
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    env = os.environ.get("ES_BENCH_CPU_CORES")
    if env:
        n = max(1, int(env))
    return n


def cpu_baselines(eq, m, k_np, W_np, target_seconds=10.0):
    """Two CPU legs on ALL host cores, each on a bounded sample of k-rows of the same grid (run before this process
    touches the GPU): (1) the oracle's C port (same algorithm as the HIP kernel, plain C + OpenMP);
    (2) the vectorised NumPy restatement (oracle/grid_numpy.py), one process per k-tile."""
    from oracle.port import PortProblem
    from oracle import grid_numpy
    from eigensolver_amd import shooting as s
    d, p = s.make_desc(eq, "kink", m)
    desc = {f[0]: getattr(d, f[0]) for f in d._fields_}
    port = PortProblem(desc, p)
    cores = host_cores()

    def sample(rate_rows_per_s, lo=8):
        n = int(min(len(k_np), max(lo, target_seconds * rate_rows_per_s)))
        return np.linspace(0, len(k_np) - 1, n).astype(int)

    rows = np.linspace(0, len(k_np) - 1, max(8, cores // 2)).astype(int)       # calibration rows spread over k
    t = time.time()
    port.eval_grid(k_np[rows], W_np, w_mode=1, nthreads=cores)
    rows = sample(len(rows) / (time.time() - t), lo=16)
    t = time.time()
    port.eval_grid(k_np[rows], W_np, w_mode=1, nthreads=cores)
    dt = time.time() - t
    out = {"value": len(rows) * len(W_np) / dt, "unit": "det-evals/s", "cores": cores, "os_cpu_count": os.cpu_count(),
           "kind": "port",
           "sample": f"{len(rows)} of {len(k_np)} k-rows x {len(W_np)} omega (same grid, C port oracle/c/shoot_port.c, "
                     f"OpenMP {cores} threads, {dt:.1f} s)"}
    # NumPy leg: calibrate on one row in-process, then one process per k-tile
    prof = {k_: np.asarray(v) for k_, v in p.items()}
    g = grid_numpy.CylinderGrid(desc, prof)
    t = time.time()
    g.eval_grid(k_np[[len(k_np) // 2]], W_np)
    rows = sample(cores / (time.time() - t), lo=cores)
    dt = grid_numpy.timed_parallel(desc, prof, k_np[rows], W_np, cores)
    out_np = {"value": len(rows) * len(W_np) / dt, "unit": "det-evals/s", "cores": cores, "kind": "port",
              "implementation": "vectorised NumPy restatement (oracle/grid_numpy.py), one process per k-tile",
              "sample": f"{len(rows)} of {len(k_np)} k-rows x {len(W_np)} omega, {cores} processes, {dt:.1f} s"}
    return out, out_np


def launch_children(a):
    """--gpus N without a torch.distributed.run environment: start the N ranks as ONE child process tree and pass its
    output and exit code on.  This parent never initialises the GPU (no HIP call, no exec of itself)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=("strong", "weak-m"), default="strong")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump-roots", default=None, help="write the merged root table of the last step to this .npy")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_children(a))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    eq = workload_equilibrium()
    k_np, W_np = workload_grid()
    cpu = None
    if not a.no_cpu_baseline and world == 1:          # timed on rank 0 at N = 1 only, before the GPU is initialised
        cpu = cpu_baselines(eq, 1, k_np, W_np)

    import torch
    import torch.distributed as dist
    # ES_BENCH_BACKEND=gloo ES_BENCH_SHARE_GPU=1: rehearsal of the N > 1 code path on a one-GPU box (all ranks on
    # cuda:0, collectives through host memory).  The judged runs use the defaults: one rank per GPU over RCCL.
    backend = os.environ.get("ES_BENCH_BACKEND", "nccl")
    if os.environ.get("ES_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)

    from eigensolver_amd import ShootProblem, _lib
    from eigensolver_amd import distributed as D
    dev = torch.device(f"cuda:{local_rank}")
    ctx = _lib.Context(local_rank)
    strong = a.mode == "strong"
    m = 1 if strong else rank + 1                  # weak-m: rank r owns azimuthal order m = r + 1
    prob = ShootProblem(eq, "kink", m=m, ctx=ctx)
    rows_np = D.tile_rows(NK, rank, world, strided=True) if strong else np.arange(NK)
    rows_t = torch.as_tensor(rows_np, device=dev)
    k = torch.as_tensor(k_np[rows_np], dtype=torch.float64, device=dev)
    W = torch.as_tensor(W_np, dtype=torch.float64, device=dev)
    nk_local = int(k.numel())
    table = prob.alloc_root_table(1 << 18)

    def step(ev=None):
        if ev is not None:
            ev[0].record()
        D_, st = prob.eval_grid(k, W)
        if ev is not None:
            ev[1].record()
        roots, nbr = prob.find_roots(k, W, D_, st, n_bisect=N_BISECT, tol_percent=TOL_PERCENT, table=table)
        buf = None
        if world > 1:                                  # the one exchange of the path: a single all-gather
            buf = D.gather_fixed(D.pack_fixed(roots, nbr, m, rows_t, EXCHANGE_CAP), world)
        return roots, nbr, buf, st

    for _ in range(a.warmup):
        step()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        roots, nbr, buf, st = step(events[i])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0

    cdev = dev if backend == "nccl" else torch.device("cpu")
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    hist = torch.bincount(st.reshape(-1).to(torch.int64), minlength=4)[:4].to(torch.float64).to(cdev)
    n_acc = int((roots["flag"] == 1).sum())
    counts = torch.tensor([float(nbr), float(n_acc), float(nk_local * NW)], dtype=torch.float64, device=cdev)
    grid_ms_t = torch.tensor([float(np.mean([e0.elapsed_time(e1) for e0, e1 in events]))], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        dist.all_reduce(hist, op=dist.ReduceOp.SUM)
        dist.all_reduce(grid_ms_t, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    brackets_total, roots_total, grid_points = (int(x) for x in counts.tolist())
    grid_ms = float(grid_ms_t.item())                # slowest rank's average launch of the dominant kernel

    merged = None
    if world > 1:
        merged, per_rank = D.merge_fixed(buf)            # host side, outside the timed region
        assert merged.shape[0] == brackets_total, (merged.shape, brackets_total)
    elif a.dump_roots:
        merged = np.stack([roots["k"].cpu().numpy(), roots["w"].cpu().numpy(), np.full(nbr, float(m)),
                           roots["resid"].cpu().numpy(), roots["flag"].cpu().numpy().astype(np.float64),
                           roots["row"].cpu().numpy().astype(np.float64)], axis=1)

    if rank == 0:
        if a.dump_roots and merged is not None:
            np.save(a.dump_roots, merged)
        evals_per_step = grid_points + brackets_total * (8 * REFINE_ROUNDS + REFINE_POLISH)
        value = evals_per_step * a.steps / dt
        launch_evals = nk_local * NW                     # points one launch of the dominant kernel processes (rank 0)
        achieved = launch_evals * BYTES_PER_EVAL / (grid_ms * 1e-3) / 1e9
        nsteps = eq.n_nodes - 1
        tflops = launch_evals * FLOPS_PER_STEP * nsteps / (grid_ms * 1e-3) / 1e12
        traffic, traffic_src = measured_traffic_per_launch()
        if world > 1:
            traffic, traffic_src = None, None            # the committed PMC passes are N = 1 launches
        tot = float(hist.sum().item())
        frac = {n: float(hist[i].item()) / tot for i, n in enumerate(("ok", "leaky", "nonfinite", "continuum"))}
        par = "single GPU" if world == 1 else (
            f"k-rows of one grid strided over {world} ranks, one RCCL all-gather of the root tables per step" if strong
            else f"m-tiled x{world} (rank r: m = r + 1 on the full grid), one RCCL all-gather of the root tables per step")
        out = {
            "metric": "det(M) evals/sec + roots/sec, 4096x4096 (k,omega) grid",
            "value": value, "unit": "det-evals/s",
            "roots_per_s": roots_total * a.steps / dt,
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "Cylinder / non-uniform (Gaussian) axial flow, coronal, kink (m = 1), ONE 4096x4096 "
                                   "(k,omega) grid, fp64 (BASELINE.json configs[3])" if strong else
                                   "Cylinder / non-uniform (Gaussian) axial flow, coronal, m = rank+1, 4096x4096 "
                                   "(k,omega) grid per GPU, fp64 (BASELINE.json configs[3], (k, m) tiling)",
                       "nk": NK, "nw": NW, "k_rows_per_gpu": nk_local, "interior_nodes": eq.n_nodes,
                       "n_bisect": N_BISECT, "refine_rounds_9section": REFINE_ROUNDS, "refine_polish_steps": REFINE_POLISH,
                       "brackets_per_step": brackets_total, "roots_per_step": roots_total,
                       "gathered_root_records": int(merged.shape[0]) if (world > 1 and merged is not None) else 0,
                       "grid_point_status_fractions": frac,
                       "status_note": "every grid point is marched and counted in `value`; continuum points (Omega^2 "
                                      "crosses omega_A^2(r) or omega_c^2(r) inside the tube) are evaluated as the "
                                      "reference does but never bracketed",
                       "parallelism": par},
            "roofline": {"bound": "hbm", "kernel": "shoot_grid_kernel<FAM_CYL0>", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_eval": BYTES_PER_EVAL,
                         "algorithmic_bytes_per_launch": launch_evals * BYTES_PER_EVAL,
                         "bytes_per_launch_incl_status": launch_evals * BYTES_PER_EVAL_WITH_STATUS,
                         "evals_per_launch": launch_evals, "avg_launch_ms": grid_ms,
                         "note": "fp64-VALU bound, not HBM bound (SURVEY 8d): see valu_fp64"},
            "valu_fp64": {"achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": tflops / FP64_VALU_PEAK_TFLOPS,
                          "flops_per_eval": FLOPS_PER_STEP * nsteps},
        }
        if cpu is not None:
            out["cpu_baseline"], out["cpu_baseline_numpy"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
