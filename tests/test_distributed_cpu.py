"""N > 1 path on CPU: world_size-2 gloo processes exchange variable-length root tables with the same code the GPU
ranks run over RCCL (eigensolver_amd/distributed.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from eigensolver_amd import distributed as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_roots(rank, n):
    g = torch.Generator().manual_seed(100 + rank)
    return {"k": torch.rand(n, generator=g, dtype=torch.float64) + rank,
            "w": torch.rand(n, generator=g, dtype=torch.float64) * 3,
            "resid": torch.rand(n, generator=g, dtype=torch.float64) * 1e-3,
            "flag": (torch.rand(n, generator=g) > 0.3).to(torch.uint8)}


def _worker(rank, world, port, counts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    roots = _fake_roots(rank, counts[rank])
    table = D.gather_root_tables(roots, m=rank + 1, world=world)
    q.put((rank, table.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_root_tables_gloo_world2():
    world, counts = 2, [7, 0]          # ragged, including an empty table
    for counts in ([7, 0], [3, 11]):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, counts, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = dict(q.get(timeout=120) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        expect = np.concatenate([D.pack_records(_fake_roots(r, counts[r]), r + 1).numpy() for r in range(world)])
        for r in range(world):
            assert got[r].shape == (sum(counts), 5)
            assert np.array_equal(got[r], expect)            # identical, rank-major order on every rank


def test_tiling_covers_every_row_once():
    for n, world in ((4096, 8), (10, 4), (3, 8), (0, 2)):
        for strided in (True, False):
            rows = np.concatenate([D.tile_rows(n, r, world, strided) for r in range(world)]) if n else np.zeros(0)
            assert sorted(rows.tolist()) == list(range(n))
    assert D.tile_modes([0, 1, 2, 3, 4], 1, 2) == [1, 3]
    assert sum((D.tile_modes(list(range(11)), r, 8) for r in range(8)), []) .__len__() == 11


def _worker_modes(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(rank)
    n_s, n_k = (5, 0) if rank == 0 else (2, 7)          # ragged, one empty list
    local = {"sausage": (rng.random(n_s), np.full(n_s, float(rank))), "kink": (rng.random(n_k), np.full(n_k, float(rank)))}
    out = D.gather_mode_results(local)
    q.put((rank, {m: (v[0].tolist(), v[1].tolist()) for m, v in out.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_mode_results_gloo_world2():
    """k-tiled driver run (solvers.solve_distributed): the per-mode root lists of all ranks, rank-major, everywhere."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_modes, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1]
    assert len(got[0]["sausage"][0]) == 7 and len(got[0]["kink"][0]) == 7
    assert got[0]["sausage"][1] == [0.0] * 5 + [1.0] * 2 and got[0]["kink"][1] == [1.0] * 7
    exp0 = np.random.default_rng(0).random(5).tolist()
    assert got[0]["sausage"][0][:5] == exp0


# ---- strong scaling: ONE (k, omega) grid, k-rows tiled over the ranks (bench.py default mode) -----------------------
def _grid_problem():
    """Small version of the bench workload with the oracle's C port as the compute stand-in (no GPU here): the tiling,
    the fixed-capacity exchange and the merge are the code bench.py runs over RCCL."""
    from eigensolver_amd import equilibrium as q, shooting as s
    from oracle.port import PortProblem
    eq = q.CylinderFlow(U_i0=0.7, width=0.9, n_nodes=200)
    d, prof = s.make_desc(eq, "kink", 1)
    port = PortProblem({f[0]: getattr(d, f[0]) for f in d._fields_}, prof)
    k = np.linspace(0.2, 3.9, 21)
    W = 0.9 + (np.arange(64) + 0.5) * (4.1 / 64)
    return port, k, W


def _tile_step(port, k, W, rows):
    Dg, rel, st = port.eval_grid(k[rows], W, w_mode=1, nthreads=1)
    r, cnt = port.find_roots(k[rows], W, Dg, st, w_mode=1, n_bisect=16, tol=1e-3, nthreads=1)
    roots = {n: torch.as_tensor(np.ascontiguousarray(r[n])) for n in ("k", "w", "resid", "flag", "row")}
    return roots, cnt


def _worker_tiled(rank, world, port_no, cap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port_no)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    port, k, W = _grid_problem()
    rows = D.tile_rows(len(k), rank, world, strided=True)
    roots, cnt = _tile_step(port, k, W, rows)
    buf = D.gather_fixed(D.pack_fixed(roots, cnt, 1, torch.as_tensor(rows), cap), world)
    try:
        rec, counts = D.merge_fixed(buf)
        q.put((rank, rec, counts))
    except OverflowError as e:
        q.put((rank, str(e), None))
    dist.barrier()
    dist.destroy_process_group()


def _run_tiled(world, cap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port_no = _free_port()
    procs = [ctx.Process(target=_worker_tiled, args=(r, world, port_no, cap, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, rec, counts = q.get(timeout=300)
        got[r] = (rec, counts)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_k_tiled_grid_root_set_equals_single_rank():
    """The gathered root table of an N-rank k-tiled run is the N = 1 table, record for record (same order: k-rows
    outer, omega inner), on every rank -- world sizes 2 and 3 (ragged tiles: 21 rows)."""
    port, k, W = _grid_problem()
    roots, cnt = _tile_step(port, k, W, np.arange(len(k)))
    single = D.merge_fixed(D.gather_fixed(D.pack_fixed(roots, cnt, 1, torch.arange(len(k)), 256), 1))[0]
    assert single.shape == (cnt, D.N_FIELDS) and cnt > 10
    assert np.all(np.diff(single[:, 5]) >= 0)
    for world in (2, 3):
        got = _run_tiled(world, 256)
        for r in range(world):
            rec, counts = got[r]
            assert sum(counts) == cnt
            assert np.array_equal(rec, single), (world, r)


def test_fixed_exchange_reports_overflow():
    got = _run_tiled(2, 4)                 # capacity far below the number of brackets of a tile
    assert all(isinstance(got[r][0], str) and "capacity" in got[r][0] for r in range(2))


# ---- several units (azimuthal orders / modes) per rank: BASELINE configs[1], [2], [4] in bench.py --------------------
def _unit_problems():
    """Two azimuthal orders of a small rotational-flow cylinder (the configs[4] family), C port as compute stand-in."""
    from eigensolver_amd import equilibrium as q, shooting as s
    from oracle.port import PortProblem
    out = []
    for m in (1, 2):
        eq = q.CylinderRotation(v_twist=0.1, power=1.0, r_axis=0.001, n_nodes=200)
        d, prof = s.make_desc(eq, "kink", m)
        out.append((m, PortProblem({f[0]: getattr(d, f[0]) for f in d._fields_}, prof)))
    k = np.linspace(0.4, 3.8, 14)
    W = 0.7 + (np.arange(48) + 0.5) * (0.75 / 48)
    return out, k, W


def _units_step(rank, world, caps):
    probs, k, W = _unit_problems()
    rows = D.tile_rows(len(k), rank, world, strided=True)
    sends, counts = [], []
    for (m, port), cap in zip(probs, caps):
        roots, cnt = _tile_step(port, k, W, rows)
        sends.append(D.pack_fixed(roots, cnt, m, torch.as_tensor(rows), cap))
        counts.append(cnt)
    return D.concat_fixed(sends), counts


def _worker_units(rank, world, port_no, caps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port_no)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    send, counts = _units_step(rank, world, caps)
    cap_dyn = D.exchange_capacity(max(counts))               # one all_reduce(MAX): identical on every rank
    buf = D.gather_fixed(send, world)
    try:
        rec, per = D.merge_units(buf, caps)
        q.put((rank, rec, per, cap_dyn))
    except OverflowError as e:
        q.put((rank, str(e), None, cap_dyn))
    dist.barrier()
    dist.destroy_process_group()


def _run_units(world, caps):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port_no = _free_port()
    procs = [ctx.Process(target=_worker_units, args=(r, world, port_no, caps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, rec, per, cap_dyn = q.get(timeout=300)
        got[r] = (rec, per, cap_dyn)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_units_tiled_over_ranks_merge_to_the_single_rank_table():
    """(m x k-row) tiling of bench.py --workload config1/2/4: every rank owns rows r, r + N, ... of EVERY unit, the
    units' fixed-capacity tables travel in ONE all-gather, and the merged table is the N = 1 table record for record
    (units outer, rows next, omega inner); the exchange capacity comes from the data and is the same on every rank."""
    caps = [64, 128]
    send, counts = _units_step(0, 1, caps)
    single, per = D.merge_units(D.gather_fixed(send, 1), caps)
    assert per == [counts] and sum(counts) == single.shape[0] and min(counts) > 3
    assert np.all(np.diff(single[:, 2]) >= 0)                     # units in order
    for m in (1.0, 2.0):
        assert np.all(np.diff(single[single[:, 2] == m][:, 5]) >= 0)   # rows in order inside a unit
    for world in (2, 3):
        got = _run_units(world, caps)
        dyn = {got[r][2] for r in range(world)}
        assert len(dyn) == 1 and dyn.pop() >= 64
        for r in range(world):
            rec, per_rank, _ = got[r]
            assert np.array_equal(rec, single), (world, r)
            assert [sum(p[i] for p in per_rank) for i in range(2)] == counts
    got = _run_units(2, [2, 128])                                 # first unit overflows its slot
    assert all(isinstance(got[r][0], str) and "capacity" in got[r][0] for r in range(2))


def test_exchange_capacity_is_a_power_of_two_twice_the_count():
    assert D.exchange_capacity(0) == 64 and D.exchange_capacity(32) == 64 and D.exchange_capacity(33) == 128
    assert D.exchange_capacity(820) == 2048 and D.exchange_capacity(6561) == 16384
