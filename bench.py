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

Consecutive steps are software-pipelined over --streams S (default 3; the 512-row tile of an 8-GPU run: 4.11 / 3.07 / 2.88 / 3.03 ms
per step with 1 / 2 / 3 / 4, the whole grid 20.8 ms with 2 or 3) library contexts = HIP streams, one host thread
each: grid launches stay serialised (each has the whole chip), the latency-bound bracket refinement of step i (about
one wave per SIMD) overlaps the grid launch of step i + 1; collectives are issued by the main thread in step order.

Multi-GPU (default --mode strong): the 4096 k-rows of the ONE grid are tiled across the ranks (strided: rank r owns
rows r, r + N, ... -- the reference's per-k process fan-out, Density_cylinder.py:1142-1153), no data-path collective;
the only exchange is one RCCL all-gather of the fixed-capacity root tables per step.  --mode weak-m is the (k, m)
tiling of round 1: rank r solves azimuthal order m = r + 1 on the full grid.
With --gpus N > 1 and no WORLD_SIZE in the environment this process only starts the N ranks
(python -m torch.distributed.run ...) as a child process and never touches the GPU itself.

--workload config1 | config2 | config4: the other GPU configurations of BASELINE.json (never the headline), same JSON
contract: slab / non-uniform flow 1024 x 1024 both modes; cylinder / non-uniform density m = 0..4, 4096 k; cylinder /
rotational flow m = 0..10 with fp32 screening + fp64 refinement.  Their units (modes / azimuthal orders) run on one HIP
stream each; N > 1: every rank owns the k-rows r, r + N, ... of EVERY unit (equal point counts), one all-gather per step.

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
N_BISECT = 16                                 # bracket narrowed by >= 2^16, then two regula-falsi polish steps
REFINE_POLISH = 2                             # -> |d omega/omega| ~ 1e-16


def refine_plan(n_brackets=0):
    """(sections, rounds, distinct determinant evaluations per bracket) of es_shoot_find_roots for N_BISECT: 17-section
    with 16 lanes per bracket (4 rounds: 17^4 >= 2^16), whatever the bracket count (the rule must not depend on the tiling)."""
    sections = 17
    if os.environ.get("ES_REFINE_SECTIONS") in ("5", "9", "17"):    # tuning aid of the library, see DESIGN.md
        sections = int(os.environ["ES_REFINE_SECTIONS"])
    rounds, span = 0, 1.0
    while span < 2.0 ** N_BISECT:
        span *= sections
        rounds += 1
    return sections, rounds, (sections - 1) * rounds + REFINE_POLISH
TOL_PERCENT = 1e-3
# the capacity of the fixed-size all-gather is sized from the data during warm-up (distributed.exchange_capacity)
HBM_PEAK_GBS = 8000.0                         # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VALU_PEAK_TFLOPS = 78.6                  # fp64 vector peak (spec)
# algorithmic traffic per det-eval of the grid kernel: 8 B of D written + amortised inputs; the kernel
# also writes a 1-byte status per point, reported separately
BYTES_PER_EVAL = 8.0 + 16.0 * (1.0 / NW + 1.0 / NK)
BYTES_PER_EVAL_WITH_STATUS = BYTES_PER_EVAL + 1.0
# fp64 operations per det-eval of the grid kernel (FMA = 2, division = 1), see DESIGN.md "kernel K3"
FLOPS_PER_STEP = 2 * 9 + 8 + 26             # 2 coefficient sets (1 add, 3 fma, 2 mul each) + shared reciprocal (1 div, 3 mul, 2 fma) + one adjoint RK4 step in the scaled-coefficient form without the division by 3 (12 fma, 2 add = 26)


def workload_equilibrium():
    from eigensolver_amd import equilibrium as q
    # Gaussian flow of width 0.9 and amplitude 0.35 vA_i0 (CF:126-135 with the author's commented values)
    return q.CylinderFlow(U_i0=0.7, width=0.9)


def workload_grid():
    k = np.linspace(0.01, 4.0, NK)
    W = W_LO + (np.arange(NW, dtype=np.float64) + 0.5) * ((W_HI - W_LO) / NW)
    return k, W


def _pmc_file(workload):
    """Committed rocprofv3 PMC summary of THIS workload's bench command (tools/profile_bench.sh <tag> <workload> +
    tools/summarize_profiles.py): profiles/pmc_latest_<workload>.json; the headline also under its round-1 name."""
    for name in (f"pmc_latest_{workload}.json",) + (("bench_pmc_hbm_latest.json",) if workload == "config3" else ()):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            try:
                return json.load(open(path)), "profiles/" + name
            except Exception:
                pass
    return None, None


def _grid_entry(d, counter, key):
    """Per-dispatch mean of `counter` for the grid-march kernel (fp64 or fp32 screening) in a PMC summary."""
    best = None
    for k, v in (d.get(counter) or {}).items():
        if "shoot_grid" in k and (best is None or v.get("dispatches", 0) > best.get("dispatches", 0)):
            best = v
    return None if best is None else best.get(key)


def measured_traffic_per_launch(workload="config3"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE collected in separate runs of this same command, profiles/README.md) and where they come from."""
    d, src = _pmc_file(workload)
    if d is None:
        return None, None
    f = _grid_entry(d, "FETCH_SIZE", "mean_KB_per_dispatch")
    w = _grid_entry(d, "WRITE_SIZE", "mean_KB_per_dispatch")
    if f is None or w is None:
        return None, None
    # gfx950: FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads (x2 correction of the guide);
    # the reads of this kernel are 8-B scalar/LDS-staging loads of a small L2-resident table, left uncorrected
    return (f + w) * 1024.0, src + " (" + str(d.get("round", "?")) + ")"


def measured_cycles_per_launch(workload="config3"):
    """(GPU cycles, VALU wave-instructions) per launch of the dominant kernel from the committed SQ / GRBM pass
    (GRBM_GUI_ACTIVE is summed over the 8 XCDs; under counter collection every launch runs alone)."""
    d, src = _pmc_file(workload)
    if d is None:
        return None, None, None
    cyc = _grid_entry(d, "GRBM_GUI_ACTIVE", "mean_per_dispatch")
    ins = _grid_entry(d, "SQ_INSTS_VALU", "mean_per_dispatch")
    if cyc is None:
        return None, None, None
    return cyc / 8.0, ins, src + " (" + str(d.get("round", "?")) + ")"


def loop_model(kernel):
    """Instruction counts of the march loop of `kernel` (spelled as rocprofv3 prints it, blanks removed) from the
    disassembly of the shipped code object: profiles/isa_loop_counts.json, written by tools/isa_loop_count.py --write and
    held to the built library by tests/test_codeobj.py."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "isa_loop_counts.json")))
        return t.get(kernel)
    except Exception:
        return None


def valu_views(kernel, launch_evals, nsteps, grid_ms, unshared_ms, workload, pmc_ok=True):
    """`valu_fp64` (executed fp64 flops of the loop's own instruction stream against the fp64 vector peak) and `valu_issue`
    (issue cycles that stream needs against the cycles a launch takes) of one launch of `kernel`."""
    m = loop_model(kernel)
    if m is None:
        return None, None
    p = m["per_point_step"]
    fp64 = p.get("fp64_fma", 0.0) * 2.0 + p.get("fp64", 0.0) + p.get("rcp_f64", 0.0)
    f32 = (p.get("packed_f32_fma", 0.0) * 4.0 + p.get("packed_f32", 0.0) * 2.0 + p.get("f32_fma", 0.0) * 2.0 + p.get("f32", 0.0)
           + p.get("rcp_f32", 0.0))
    flops = fp64 + f32
    tfl = launch_evals * flops * nsteps / (grid_ms * 1e-3) / 1e12
    fp64_view = {"achieved": tfl, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / FP64_VALU_PEAK_TFLOPS,
                 "frac_unshared": (tfl * grid_ms / unshared_ms / FP64_VALU_PEAK_TFLOPS) if unshared_ms else None,
                 "flops_per_point_step": flops, "flops_per_eval": flops * nsteps,
                 "note": "flops EXECUTED by the march loop of " + kernel + " (fma = 2; ISA count, profiles/isa_loop_counts.json)"
                         + ("; fp32 screening march: priced against the fp64 vector peak only for comparison" if f32 else "")}
    issue = None
    cyc, ins, src = measured_cycles_per_launch(workload) if pmc_ok else (None, None, None)
    # instructions behind a uniform forward branch of the loop (the power-of-two renormalisation of the fp32 march, entered
    # on every 16th iteration) are left out of the price: a lower bound of what the launch has to issue
    per_step = m.get("issue_cycles_unconditional", m["issue_cycles_per_wave_point_step"])
    need = per_step * (launch_evals / 64.0) * nsteps / 1024.0
    issue = {"kernel": kernel, "loop_instructions_per_point_step": {k_: round(v, 3) for k_, v in p.items()},
             "issue_cycles_per_wave_point_step": per_step,
             "issue_cycles_per_wave_point_step_with_conditional_block": m["issue_cycles_per_wave_point_step"],
             "issue_cycles_needed_per_launch": need, "cycles_per_launch": cyc,
             "frac": (need / cyc) if cyc else None, "valu_wave_instructions_per_launch": ins, "source": src,
             "frac_at_2p1_ghz_unshared": (need / (unshared_ms * 1e-3 * 2.1e9)) if unshared_ms else None}
    return fp64_view, issue


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


def cpu_baselines(eq, m, k_np, W_np, target_seconds=10.0, mode="kink", numpy_leg=True):
    """Two CPU legs on ALL host cores, each on a bounded sample of k-rows of the same grid (run before this process
    touches the GPU): (1) the oracle's C port (same algorithm as the HIP kernel, plain C + OpenMP);
    (2) the vectorised NumPy restatement (oracle/grid_numpy.py), one process per k-tile."""
    from oracle.port import PortProblem
    from oracle import grid_numpy
    from eigensolver_amd import shooting as s
    d, p = s.make_desc(eq, mode, m)
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
    if not numpy_leg:
        return out, None
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
    ap.add_argument("--skip-continuum", action="store_true",
                    help="time the steps with ES_EVAL_SKIP_CONTINUUM: points inside continuum bands get D = NaN, are not "
                         "marched and are NOT counted in `value`.  Default: every grid point is marched, as the reference "
                         "evaluates every point; the skip mode is then measured after the timed region and reported "
                         "under config.skip_continuum_mode")
    ap.add_argument("--dump-roots", default=None, help="write the merged root table of the last step to this .npy")
    ap.add_argument("--streams", type=int, default=3,
                    help="software pipelining of consecutive steps: S library contexts (HIP streams), each driven by its own "
                         "host thread; the latency-bound refinement of step i overlaps the grid launches of steps i + 1, i + 2")
    ap.add_argument("--no-extra-mode", action="store_true",
                    help="do not measure the other continuum mode after the timed region (profiling runs: keeps the "
                         "per-kernel counters of one mode apart)")
    ap.add_argument("--workload", choices=("config1", "config2", "config3", "config4"), default="config3",
                    help="config3 (default, the headline): BASELINE.json configs[3]; config1: Slab / non-uniform flow, "
                         "1024x1024, both modes; config2: Cylinder / non-uniform density, m = 0..4, 4096 k; config4: Cylinder / "
                         "rotational flow, m = 0..10, fp32 bracket + fp64 refine (second lines, never the headline)")
    ap.add_argument("--precision", choices=("mixed", "f64"), default=None,
                    help="config4: mixed (default: fp32 screening + fp64 refinement) or f64; config1 / config2 are f64")
    ap.add_argument("--share-of", type=int, default=1,
                    help="N = 1 only, a projection aid and never the judged line: this process computes what rank 0 of a "
                         "--gpus E strong-scaling run computes (k-rows 0, E, 2E, ... and the packing of its exchange "
                         "buffer, no collective); the line carries `emulated_share_of` and `projected_whole_job_value`")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_children(a))
    if a.workload != "config3":
        return main_units(a)

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
    strong = a.mode == "strong"
    m = 1 if strong else rank + 1                  # weak-m: rank r owns azimuthal order m = r + 1
    share = a.share_of if (world == 1 and strong and a.share_of > 1) else 1
    rows_np = D.tile_rows(NK, rank, world * share, strided=True) if strong else np.arange(NK)
    rows_t = torch.as_tensor(rows_np, device=dev)
    k = torch.as_tensor(k_np[rows_np], dtype=torch.float64, device=dev)
    W = torch.as_tensor(W_np, dtype=torch.float64, device=dev)
    nk_local = int(k.numel())
    # one lane = (stream, library context, problem, root table, device count); --streams 1: the current stream, inline
    n_lanes = max(1, a.streams)
    lanes = []
    for j in range(n_lanes):
        stream = torch.cuda.current_stream(dev) if n_lanes == 1 else torch.cuda.Stream(device=dev)
        cx = _lib.Context(local_rank, stream=stream)
        pr = ShootProblem(eq, "kink", m=m, ctx=cx)
        lanes.append([stream, cx, pr, None, None])
    ctx, prob = lanes[0][1], lanes[0][2]
    torch.cuda.synchronize()

    skip = bool(a.skip_continuum)
    # sizes from the data, once, before anything is timed: the bracket count of this rank's tile (synchronous call), the
    # table capacity (next power of two >= 2 x count: the asynchronous refinement launches are sized for it) and the
    # capacity of the fixed-size exchange (one all_reduce(MAX) of the counts; round 2 sent 32768 records per rank for ~820)
    with torch.cuda.stream(lanes[0][0]):
        D0, st0 = prob.eval_grid(k, W, skip_continuum=skip)
        _, nbr0 = prob.find_roots(k, W, D0, st0, n_bisect=N_BISECT, tol_percent=TOL_PERCENT, capacity=1 << 18)
        del D0, st0
    table_cap = 1024
    while table_cap < 2 * nbr0:
        table_cap *= 2
    exchange_cap = D.exchange_capacity(nbr0) if (world > 1 or share > 1) else 0
    for lane in lanes:
        with torch.cuda.stream(lane[0]):
            lane[3] = lane[2].alloc_root_table(table_cap)
            lane[4] = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    import threading
    grid_lock = threading.Lock()
    grid_tail = [None]                             # event after the most recent grid launch of any lane
    no_gate = os.environ.get("ES_BENCH_NO_GATE") == "1"     # A/B aid: grid launches of the lanes not ordered

    def compute(j, ev=None):
        """One pass of the hot path over the rank's tile on lane j, everything enqueued and NOTHING read back: grid, bracket
        flags + scan + emit, refinement with the count in device memory (es_shoot_find_roots_async), send buffer of the
        exchange; `done` marks their completion on the lane's stream."""
        stream, cx, pr, tb, cnt = lanes[j]
        with torch.cuda.stream(stream):
            # grid launches of different lanes run one after the other (each has the whole chip, and the events below
            # time one launch); what overlaps with the NEXT step's grid is this step's bracket search and refinement
            with grid_lock:
                if grid_tail[0] is not None and not no_gate:
                    stream.wait_event(grid_tail[0])
                if ev is not None:
                    ev[0].record(stream)
                D_, st = pr.eval_grid(k, W, skip_continuum=skip)
                if ev is not None:
                    ev[1].record(stream)
                tail = torch.cuda.Event()
                tail.record(stream)
                grid_tail[0] = tail
            roots = pr.find_roots_async(k, W, D_, st, tb, cnt, n_bisect=N_BISECT, tol_percent=TOL_PERCENT)
            send = D.pack_fixed(roots, cnt, m, rows_t, exchange_cap, ctx=cx) if exchange_cap else None
            done = torch.cuda.Event()
            done.record(stream)
        return roots, cnt, send, st, done

    def exchange(send, done):
        # the one exchange of the path: a single all-gather, issued by the main thread in step order on every rank
        cur = torch.cuda.current_stream(dev)
        cur.wait_event(done)
        send.record_stream(cur)                        # allocated on the lane's stream, read by the collective on this one
        return D.gather_fixed(send, world)

    def step(ev=None):
        roots, cnt, send, st, done = compute(0, ev)
        buf = exchange(send, done) if world > 1 else None
        return roots, cnt, buf, st

    import queue
    from concurrent.futures import ThreadPoolExecutor
    free_lanes = queue.Queue()
    for j in range(n_lanes):
        free_lanes.put(j)

    def pipelined(ev):
        j = free_lanes.get()
        try:
            return compute(j, ev)
        finally:
            free_lanes.put(j)

    for _ in range(a.warmup):
        for j in range(n_lanes):
            e = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            _, cnt_w, send_w, _, done_w = compute(j, e)
            if world > 1:
                exchange(send_w, done_w)            # also brings the communicator up before the timed region
            torch.cuda.synchronize()
            assert int(cnt_w.item()) <= table_cap, (int(cnt_w.item()), table_cap)
    # the dominant kernel alone on a busy chip: grid launches back to back on one stream, the first one not counted (a
    # launch that follows idle time runs up to 13 % slower while the clocks ramp: tools/probe/time_small_launches.py)
    alone = []
    with torch.cuda.stream(lanes[0][0]):
        for _ in range(4):
            e = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            e[0].record(lanes[0][0])
            prob.eval_grid(k, W, skip_continuum=skip)
            e[1].record(lanes[0][0])
            alone.append(e)
    torch.cuda.synchronize()
    unshared = [e0.elapsed_time(e1) for e0, e1 in alone[1:]]
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    pool = ThreadPoolExecutor(max_workers=n_lanes) if n_lanes > 1 else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if pool is None:
        for i in range(a.steps):
            roots, cnt, buf, st = step(events[i])
    else:
        futs = [pool.submit(pipelined, events[i]) for i in range(a.steps)]
        for f in futs:
            roots, cnt, send, st, done = f.result()
            buf = exchange(send, done) if world > 1 else None
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    nbr = int(cnt.item())                            # the count of the last step, read after the timed region
    if nbr > table_cap:
        raise SystemExit(f"bracket count {nbr} exceeds the table capacity {table_cap} sized during warm-up")
    roots = {key: v[:nbr].clone() for key, v in roots.items()}     # the lane's table is reused by the steps below

    # the other mode of the grid evaluation, a few steps outside the timed region (N = 1 only): reported, never `value`
    other = None
    if world == 1 and share == 1 and not a.no_extra_mode:
        st_main = st
        skip = not skip
        step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_other = max(3, min(10, a.steps))
        for _ in range(n_other):
            r_o, cnt_o, _, st_o = step()
        torch.cuda.synchronize()
        dt_o = (time.perf_counter() - t1) / n_other
        nbr_o = int(cnt_o.item())
        n_eval_o = int((st_o != 3).sum()) if skip else nk_local * NW
        other = {"ms_per_step": dt_o * 1e3, "roots_per_s": int((r_o["flag"][:nbr_o] == 1).sum()) / dt_o,
                 "grid_points_evaluated_per_step": n_eval_o, "brackets_per_step": nbr_o,
                 "det_evals_per_s": (n_eval_o + nbr_o * refine_plan(nbr_o)[2]) / dt_o, "steps": n_other}
        skip = not skip
        st = st_main
    cdev = dev if backend == "nccl" else torch.device("cpu")
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    hist = torch.bincount(st.reshape(-1).to(torch.int64), minlength=4)[:4].to(torch.float64).to(cdev)
    n_acc = int((roots["flag"] == 1).sum())
    # grid points one step really evaluates: all of them, or (default) those outside the continuum bands
    n_eval_local = int((st != 3).sum()) if skip else nk_local * NW
    counts = torch.tensor([float(nbr), float(n_acc), float(n_eval_local), float(nbr * refine_plan(nbr)[2])],
                          dtype=torch.float64, device=cdev)
    grid_ms_t = torch.tensor([float(np.mean([e0.elapsed_time(e1) for e0, e1 in events]))], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        dist.all_reduce(hist, op=dist.ReduceOp.SUM)
        dist.all_reduce(grid_ms_t, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    brackets_total, roots_total, grid_points, refine_evals = (int(x) for x in counts.tolist())
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
        sections, rounds, ev_per_bracket = refine_plan(nbr)          # rank 0's tile (the ranks' counts may differ)
        evals_per_step = grid_points + refine_evals
        value = evals_per_step * a.steps / dt
        launch_evals = n_eval_local                      # points one launch of the dominant kernel evaluates (rank 0)
        achieved = launch_evals * BYTES_PER_EVAL / (grid_ms * 1e-3) / 1e9
        nsteps = eq.n_nodes - 1
        tflops = launch_evals * FLOPS_PER_STEP * nsteps / (grid_ms * 1e-3) / 1e12
        traffic, traffic_src = measured_traffic_per_launch("config3")
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
                       "n_bisect": N_BISECT, "refine_sections": sections, "refine_rounds": rounds,
                       "refine_polish_steps": REFINE_POLISH, "refine_evals_per_bracket": ev_per_bracket,
                       "brackets_per_step": brackets_total, "roots_per_step": roots_total,
                       "gathered_root_records": int(merged.shape[0]) if (world > 1 and merged is not None) else 0,
                       "grid_point_status_fractions": frac,
                       "grid_points_evaluated_per_step": grid_points,
                       "continuum_points": ("ES_EVAL_SKIP_CONTINUUM: points inside a continuum band (Omega^2 crosses "
                                            "omega_A^2(r) or omega_c^2(r) inside the tube; the reference returns integrator "
                                            "noise there and the grid search never brackets them) get D = NaN, are NOT "
                                            "marched and are NOT counted in `value`") if skip else
                                           ("every grid point is marched and counted in `value`, as the reference evaluates "
                                            "every point; the continuum fraction of grid_point_status_fractions is never "
                                            "bracketed"),
                       ("all_points_marched_mode" if skip else "skip_continuum_mode"): other,
                       "pipelined_streams": n_lanes,
                       "parallelism": par},
            "roofline": {"bound": "hbm", "kernel": "shoot_grid_kernel<FAM_CYL0>", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_eval": BYTES_PER_EVAL,
                         "algorithmic_bytes_per_launch": launch_evals * BYTES_PER_EVAL,
                         "bytes_per_launch_incl_status": launch_evals * BYTES_PER_EVAL_WITH_STATUS,
                         "evals_per_launch": launch_evals, "avg_launch_ms": grid_ms,
                         "avg_launch_ms_unshared": (float(np.mean(unshared)) if unshared else None),
                         "launch_note": "avg_launch_ms: HIP events around every grid launch of the timed region; with "
                                        "pipelined streams the previous steps' refinement shares the chip with it. "
                                        "avg_launch_ms_unshared: the same launch alone on a busy chip (back-to-back launches before the timed region)",
                         "note": "fp64-VALU bound, not HBM bound (SURVEY 8d): see valu_fp64"},
            "valu_fp64": {"achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": tflops / FP64_VALU_PEAK_TFLOPS,
                          "frac_unshared": (tflops * grid_ms / float(np.mean(unshared)) / FP64_VALU_PEAK_TFLOPS
                                            if unshared else None),
                          "flops_per_eval": FLOPS_PER_STEP * nsteps},
        }
        # issue-slot view of the same kernel (clock-independent): cycles the loop's instruction stream needs at 4 cycles
        # per fp64 wave-instruction (16 for v_rcp_f64) on 4 SIMDs x 256 CUs -- counted in the disassembly of the shipped
        # code object (profiles/isa_loop_counts.json) -- against the cycles a launch takes (PMC)
        kernel = prob.grid_kernel_name(NW)
        out["roofline"]["kernel"] = kernel
        _, issue = valu_views(kernel, launch_evals, nsteps, grid_ms, float(np.mean(unshared)) if unshared else None,
                              "config3", pmc_ok=(world == 1 and not skip))
        if issue is not None:
            out["valu_issue"] = issue
        out["config"]["exchange_capacity_records"] = exchange_cap
        out["config"]["root_table_capacity"] = table_cap
        if cpu is not None:
            out["cpu_baseline"], out["cpu_baseline_numpy"] = cpu
        if share > 1:
            out["emulated_share_of"] = share
            out["projected_whole_job_value"] = value * share
            out["config"]["parallelism"] = (f"PROJECTION: one GPU computing rank 0's tile of a {share}-GPU strong-scaling run "
                                            f"(every {share}th k-row, exchange buffer packed, no collective)")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def workload_units(name):
    """The units (one problem + one (k, omega) grid each) of BASELINE.json configs[1], [2], [4]; the grids are those of
    tests/test_configs_gpu.py and tools/full_size_parity.py.  Returns (description, n_bisect-independent list of
    (label, unit id, equilibrium, mode, m, k, W))."""
    from eigensolver_amd import equilibrium as q
    if name == "config1":
        # SF-G (flow_multiprocessor_coronal.py:409-480 workers, :758-821 driver): both modes of the Gaussian flow slab
        eq = q.SlabFlow(U_i0=0.35, width=1.5)
        k = np.linspace(0.05, 3.5, 1024)
        W = 1.4 + (np.arange(1024, dtype=np.float64) + 0.5) * (2.45 - 1.4) / 1024
        units = [("sausage", 0, eq, "sausage", None, k, W), ("kink", 1, eq, "kink", None, k, W)]
        desc = ("Slab / non-uniform (Gaussian) flow, coronal, sausage + kink, 1024x1024 (k,omega) grid per mode, fp64 "
                "(BASELINE.json configs[1]); NOT the headline")
    elif name == "config2":
        # CD-C (Density_cylinder.py:546 / :847 workers, :1126-1183 driver): azimuthal orders 0..4 on 4096 wavenumbers
        eq = q.CylinderDensity(width=0.95)
        k = np.linspace(0.01, 4.5, 4096)
        W = 2.05 + (np.arange(384, dtype=np.float64) + 0.5) * (4.95 - 2.05) / 384
        units = [(f"m={m}", m, eq, "sausage" if m == 0 else "kink", m, k, W) for m in range(5)]
        desc = ("Cylinder / non-uniform (Gaussian) density, coronal, m = 0..4, 4096 k x 384 omega per order, fp64 "
                "(BASELINE.json configs[2]); NOT the headline")
    elif name == "config4":
        # CR-KF (Twisted_photospheric_nonlinear_flow_kink_fast.py:454-734): rotational flow, orders 0..10
        k = np.linspace(0.25, 4.0, 1024)
        W = 0.7 + (np.arange(1024, dtype=np.float64) + 0.5) * ((1.45 - 0.7) / 1024)
        units = [(f"m={m}", m, q.CylinderRotation(v_twist=0.1, power=1.0, r_axis=0.01 if m == 0 else 0.001),
                  "sausage" if m == 0 else "kink", m, k, W) for m in range(11)]
        desc = ("Cylinder / rotational flow (v_phi = 0.1 r), photospheric, m = 0..10, 1024x1024 (k,omega) grid per order, "
                "N = 2000 nodes (BASELINE.json configs[4]); NOT the headline")
    else:
        raise ValueError(name)
    return desc, units


def main_units(a):
    """BASELINE.json configs[1], [2], [4] under the same contract as the headline.  One step = every unit (mode /
    azimuthal order) of the workload once: grid march + bracket search + refinement, fp64 (configs[1], [2]) or fp32
    screening march + fp64 re-evaluation of the unsure points and of both bracket ends + fp64 refinement (configs[4],
    es_shoot_find_roots_mixed; --precision f64 runs the fp64 path, bit-identical tables: tests/test_mixed_gpu.py).
    Every unit has its own HIP stream (library context) and host thread: the units are independent problems, and the
    latency-bound refinement launches of one overlap the grid march of another.  N > 1: rank r owns the k-rows r, r + N,
    ... of EVERY unit -- equal point counts on every rank (round 2 dealt whole orders round-robin: 2,2,2,1,1,1,1,1 for
    11 orders on 8 ranks) -- and one all-gather per step carries all units' fixed-capacity tables (capacities sized from
    the data during warm-up)."""
    desc, units = workload_units(a.workload)
    mixed = a.workload == "config4" and a.precision != "f64"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    cpu = None
    if not a.no_cpu_baseline and world == 1:          # rank 0 at N = 1 only, before the GPU is initialised
        u0 = units[len(units) // 2]
        cpu = cpu_baselines(u0[2], u0[4], u0[5], u0[6], mode=u0[3], numpy_leg=False)[0]
        cpu["sample"] = f"unit {u0[0]}: " + cpu["sample"]
    import torch
    import torch.distributed as dist
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
    from concurrent.futures import ThreadPoolExecutor
    dev = torch.device(f"cuda:{local_rank}")
    share = a.share_of if (world == 1 and a.share_of > 1) else 1
    # lanes per unit: consecutive steps of ONE unit are pipelined over L library contexts (streams) as the steps of the
    # headline are over --streams, so that a unit's refinement chain does not hold up its own next march; with many units
    # the other units' marches fill that time anyway
    n_lanes = max(1, -(-6 // len(units)))              # configs[1]: 3, configs[2]: 2, configs[4]: 1
    if os.environ.get("ES_BENCH_UNIT_LANES"):           # A/B aids, never the judged configuration
        n_lanes = max(1, int(os.environ["ES_BENCH_UNIT_LANES"]))
    # ES_BENCH_GATE=1 orders the grid marches of the units one after the other by an event chain (as the steps of the
    # headline are).  Measured on one box, ms per step without / with it: configs[1] 2.72-2.82 / 2.91-2.94, configs[2]
    # 7.37-7.44 / 7.64-7.92, configs[4] 80.0 / 84.2 -- free-running units overlap better than ordered ones; what matters is
    # that there is no barrier between the steps of different units (with one: 3.07, 8.03, 87.5)
    gate = os.environ.get("ES_BENCH_GATE") == "1"
    work, lanes = [], []
    for label, uid, eq, mode, m, k_np, W_np in units:
        rows_np = D.tile_rows(len(k_np), rank, world * share, strided=True)
        unit_lanes = []
        for _ in range(n_lanes):
            stream = torch.cuda.Stream(device=dev)
            cx = _lib.Context(local_rank, stream=stream)
            cx.grid_timer(True)
            prob = ShootProblem(eq, mode, m=m, ctx=cx)
            unit_lanes.append({"label": label, "uid": uid, "eq": eq, "prob": prob, "stream": stream, "ctx": cx,
                               "k": torch.as_tensor(k_np[rows_np], dtype=torch.float64, device=dev),
                               "W": torch.as_tensor(W_np, dtype=torch.float64, device=dev),
                               "rows": torch.as_tensor(rows_np, device=dev), "nk": len(rows_np), "nw": len(W_np)})
        lanes.append(unit_lanes)
        work.append(unit_lanes[0])
    pool = ThreadPoolExecutor(max_workers=len(work) * n_lanes)
    torch.cuda.synchronize()

    def sizing(item):
        """Bracket count of the unit's tile (synchronous fp64 call, once, untimed) -> table and exchange capacities."""
        with torch.cuda.stream(item["stream"]):
            D_, st = item["prob"].eval_grid(item["k"], item["W"])
            _, nbr = item["prob"].find_roots(item["k"], item["W"], D_, st, n_bisect=N_BISECT, tol_percent=TOL_PERCENT,
                                             capacity=1 << 17)
            hist = torch.bincount(st.reshape(-1).to(torch.int64), minlength=4)[:4].cpu().numpy()
        return nbr, hist

    sized = list(pool.map(sizing, work))
    for item, (nbr0, hist) in zip(work, sized):
        cap = 1024
        while cap < 2 * nbr0:
            cap *= 2
        item["xcap"] = D.exchange_capacity(nbr0) if (world > 1 or share > 1) else 0
        item["hist"] = hist
        with torch.cuda.stream(item["stream"]):
            item["table"] = item["prob"].alloc_root_table(cap)
            item["cnt"] = torch.zeros(1, dtype=torch.int32, device=dev)
        item["cap"] = cap
        item["ctx"].grid_time()                       # forget the sizing launches
    for unit_lanes in lanes:
        for ln in unit_lanes[1:]:
            ln["xcap"], ln["hist"], ln["cap"] = unit_lanes[0]["xcap"], unit_lanes[0]["hist"], unit_lanes[0]["cap"]
            with torch.cuda.stream(ln["stream"]):
                ln["table"] = ln["prob"].alloc_root_table(ln["cap"])
                ln["cnt"] = torch.zeros(1, dtype=torch.int32, device=dev)

    import threading
    grid_lock = threading.Lock()
    grid_tail = [None]                             # event after the most recent grid-march launch of any unit

    def one_unit(item):
        """One pass of the hot path over one unit (one lane of it) on its stream: grid march (fp64, or the fp32 screening of
        the mixed search), bracket search, refinement, exchange buffer.  Units and lanes run free: each on its own host thread
        and HIP stream, no barrier between them, so the latency-bound refinement chains of some overlap the throughput-bound
        marches of others."""
        prob, stream = item["prob"], item["stream"]
        with torch.cuda.stream(stream):
            if gate:
                with grid_lock:
                    if grid_tail[0] is not None:
                        stream.wait_event(grid_tail[0])
                    if mixed:
                        D_, st = prob.screen_grid(item["k"], item["W"])
                    else:
                        D_, st = prob.eval_grid(item["k"], item["W"])
                    tail = torch.cuda.Event()
                    tail.record(stream)
                    grid_tail[0] = tail
            elif mixed:
                D_, st = prob.screen_grid(item["k"], item["W"])
            else:
                D_, st = prob.eval_grid(item["k"], item["W"])
            if mixed:
                roots, nbr, _, _, stats = prob.find_roots_screened(item["k"], item["W"], D_, st, n_bisect=N_BISECT,
                                                                   tol_percent=TOL_PERCENT, table=item["table"])
                item["nre"] = stats[0] + stats[1]
                item["count"] = count = nbr
                full = item["table"][0]
            else:
                full = prob.find_roots_async(item["k"], item["W"], D_, st, item["table"], item["cnt"], n_bisect=N_BISECT,
                                             tol_percent=TOL_PERCENT)
                item["nre"] = 0
                count = item["cnt"]
            send = D.pack_fixed(full, count, item["uid"], item["rows"], item["xcap"], ctx=item["ctx"]) if item["xcap"] else None
            done = torch.cuda.Event()
            done.record(stream)
        return send, done

    import queue
    results = queue.Queue()

    def unit_loop(idx, lane, nsteps):
        """Lane `lane` of unit `idx` runs the steps lane, lane + L, ... back to back on its own thread and stream: no barrier
        between the steps of different units (the all-gather of step i waits for the step-i buffers of all units, the units
        themselves go on)."""
        for i in range(lane, nsteps, n_lanes):
            results.put((i, idx) + one_unit(lanes[idx][lane]))

    def run_steps(nsteps):
        futs = [pool.submit(unit_loop, idx, lane, nsteps) for idx in range(len(work)) for lane in range(n_lanes)]
        buf = None
        if world > 1:                                  # the one exchange per step: ONE all-gather of all units' tables
            pending = {}
            cur = torch.cuda.current_stream(dev)
            for i in range(nsteps):
                while len(pending.get(i, {})) < len(work):
                    si, idx, send, done = results.get()
                    pending.setdefault(si, {})[idx] = (send, done)
                sends = []
                for idx in range(len(work)):
                    send, done = pending[i][idx]
                    cur.wait_event(done)
                    send.record_stream(cur)
                    sends.append(send)
                del pending[i]
                buf = D.gather_fixed(D.concat_fixed(sends), world)
        for f in futs:
            f.result()
        while not results.empty():
            results.get()
        return buf

    if a.warmup > 0:
        run_steps(max(a.warmup, n_lanes))              # every lane at least once
        torch.cuda.synchronize()
    for unit_lanes in lanes:
        for ln in unit_lanes:
            ln["ctx"].grid_time()
    # the dominant kernel alone on the chip: grid-march launches of the middle unit back to back (first not counted)
    mid = work[len(work) // 2]
    alone = []
    with torch.cuda.stream(mid["stream"]):
        for _ in range(4):
            e = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            e[0].record(mid["stream"])
            if mixed:
                mid["prob"].screen_grid(mid["k"], mid["W"])
            else:
                mid["prob"].eval_grid(mid["k"], mid["W"])
            e[1].record(mid["stream"])
            alone.append(e)
    torch.cuda.synchronize()
    mid["ctx"].grid_time()
    unshared = [e0.elapsed_time(e1) for e0, e1 in alone[1:]]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    buf = run_steps(a.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    # per-launch duration of the grid-march kernel: HIP events inside the library, on the stream of every launch
    gt = [ln["ctx"].grid_time() for unit_lanes in lanes for ln in unit_lanes]
    grid_ms_local = sum(t for t, _ in gt) / max(1, sum(n for _, n in gt))
    # counts of the last step, read after the timed region
    nbr = nacc = nre = 0
    for item in work:
        c = item["count"] if mixed else int(item["cnt"].item())
        item["count"] = c
        if c > item["cap"]:
            raise SystemExit(f"unit {item['label']}: bracket count {c} exceeds the table capacity {item['cap']}")
        nbr += c
        nacc += int((item["table"][0]["flag"][:c] == 1).sum())
        nre += item["nre"]
    cdev = dev if backend == "nccl" else torch.device("cpu")
    tmax = torch.tensor([dt, grid_ms_local], dtype=torch.float64, device=cdev)
    grid_points_local = sum(item["nk"] * item["nw"] for item in work)
    counts = torch.tensor([float(nbr), float(nacc), float(nre), float(grid_points_local)], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    dt, grid_ms = float(tmax[0].item()), float(tmax[1].item())
    nbr, nacc, nre, grid_points = (int(x) for x in counts.tolist())
    merged = None
    if world > 1:
        merged, _ = D.merge_units(buf, [item["xcap"] for item in work])
        assert merged.shape[0] == nbr, (merged.shape, nbr)
    elif a.dump_roots:
        parts = []
        for item in work:
            c, t = item["count"], item["table"][0]
            parts.append(np.stack([t["k"][:c].cpu().numpy(), t["w"][:c].cpu().numpy(), np.full(c, float(item["uid"])),
                                   t["resid"][:c].cpu().numpy(), t["flag"][:c].cpu().numpy().astype(np.float64),
                                   item["rows"][t["row"][:c].long()].cpu().numpy().astype(np.float64)], axis=1))
        merged = np.concatenate(parts, axis=0)
    if rank == 0:
        if a.dump_roots and merged is not None:
            np.save(a.dump_roots, merged)
        _, rounds, ev_per_bracket = refine_plan()
        evals = grid_points + nre + nbr * ev_per_bracket
        value = evals * a.steps / dt
        launch_evals = mid["nk"] * mid["nw"]
        nw = mid["nw"]
        bytes_per_eval = 8.0 + 16.0 * (1.0 / nw + 1.0 / max(1, mid["nk"]))
        achieved = launch_evals * bytes_per_eval / (grid_ms * 1e-3) / 1e9
        nsteps = mid["eq"].n_nodes - 1
        fam = int(mid["prob"].desc.geometry)
        if mixed:
            kernel = {0: "shoot_grid_f32_kernel<0,4,256,false,4>", 1: "shoot_grid_f32_kernel<1,4,256,true,2>"}[fam]
        else:
            kernel = mid["prob"].grid_kernel_name(nw)
        un = float(np.mean(unshared)) if unshared else None
        fp64_view, issue = valu_views(kernel, launch_evals, nsteps, grid_ms, un, a.workload, pmc_ok=(world == 1))
        traffic, traffic_src = measured_traffic_per_launch(a.workload) if world == 1 else (None, None)
        hist = np.sum([item["hist"] for item in work], axis=0).astype(float)
        frac = {n: hist[i] / hist.sum() for i, n in enumerate(("ok", "leaky", "nonfinite", "continuum"))}
        out = {"metric": f"det(M) evals/sec + roots/sec, {desc.split(' (BASELINE')[0]}",
               "value": value, "unit": "det-evals/s", "roots_per_s": nacc * a.steps / dt,
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f32 screening march + f64 re-evaluation / refinement" if mixed else "f64", "data": "synthetic",
               "config": {"workload": desc, "precision": "mixed" if mixed else "f64", "units": len(work),
                          "grid_per_unit": [len(units[0][5]), len(units[0][6])], "k_rows_per_gpu_per_unit": mid["nk"],
                          "interior_nodes": mid["eq"].n_nodes, "n_bisect": N_BISECT, "refine_sections": 17,
                          "refine_rounds": rounds, "refine_polish_steps": REFINE_POLISH, "pipelined_lanes_per_unit": n_lanes,
                          "grid_points_per_step": grid_points, "fp64_reevaluations_per_step": nre,
                          "brackets_per_step": nbr, "roots_per_step": nacc,
                          "grid_point_status_fractions_rank0": frac,
                          "gathered_root_records": int(merged.shape[0]) if (world > 1 and merged is not None) else 0,
                          "exchange_capacity_records_per_unit": [item["xcap"] for item in work],
                          "parallelism": "single GPU, one HIP stream + host thread per unit and lane, no barrier between the steps of "
                                         "different units" if world == 1 else
                                         f"k-rows of every unit strided over {world} ranks (equal point counts), one "
                                         "RCCL all-gather of all units' root tables per step"},
               "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                            "algorithmic_bytes_per_eval": bytes_per_eval,
                            "algorithmic_bytes_per_launch": launch_evals * bytes_per_eval,
                            "evals_per_launch": launch_evals, "avg_launch_ms": grid_ms, "avg_launch_ms_unshared": un,
                            "launch_note": "avg_launch_ms: HIP events around every launch of the grid-march kernel in the timed "
                                           "region, recorded by the library on the stream of the launch (es_context_grid_timer); "
                                           "the units run free on concurrent streams, so a launch shares the chip with the marches "
                                           "and refinements of the other units; avg_launch_ms_unshared: the same launch of one "
                                           "unit alone, back to back",
                            "note": "fp64-VALU bound, not HBM bound (SURVEY 8d): see valu_fp64 / valu_issue"}}
        if fp64_view is not None:
            out["valu_fp64"] = fp64_view
        if issue is not None:
            out["valu_issue"] = issue
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if share > 1:
            out["emulated_share_of"] = share
            out["projected_whole_job_value"] = value * share
            out["config"]["parallelism"] = (f"PROJECTION: one GPU computing rank 0's tile of a {share}-GPU run (every {share}th "
                                            "k-row of every unit, exchange buffers packed, no collective)")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
