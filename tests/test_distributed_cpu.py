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
