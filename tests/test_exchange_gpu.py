"""The native send-buffer kernel of the multi-GPU exchange (es_root_table_pack) against the torch restatement the CPU /
gloo tests use (eigensolver_amd/distributed.py::pack_fixed), and the single-rank merge."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cap", [64, 4096, 5])
def test_pack_kernel_equals_torch_path(es_ctx, cap):
    import torch
    from eigensolver_amd import ShootProblem, equilibrium as q
    from eigensolver_amd import distributed as D
    gp = ShootProblem(q.CylinderFlow(U_i0=0.6, width=1.0), "kink", ctx=es_ctx)
    k = np.linspace(0.4, 3.9, 12)
    W = 2.7 + (np.arange(160) + 0.5) * (4.95 - 2.7) / 160
    Dg, st = gp.eval_grid(k, W)
    roots, cnt = gp.find_roots(k, W, Dg, st, n_bisect=16)
    assert cnt > 5
    rows = torch.arange(3, 3 + 7 * len(k), 7, device="cuda")            # some strided global row map
    a = D.pack_fixed(roots, cnt, 1, rows, cap, ctx=es_ctx).cpu().numpy()
    b = D.pack_fixed(roots, cnt, 1, rows, cap).cpu().numpy()           # torch path
    assert a.shape == (cap + 1, D.N_FIELDS) and np.array_equal(a, b)
    if cnt <= cap:
        rec, counts = D.merge_fixed(D.gather_fixed(torch.as_tensor(a), 1))
        assert counts == [cnt] and np.array_equal(rec[:, 1], roots["w"].cpu().numpy())
    else:
        with pytest.raises(OverflowError):
            D.merge_fixed(D.gather_fixed(torch.as_tensor(a), 1))
    gp.close()


def test_exchange_over_rccl_one_rank(es_ctx):
    """The exchange as bench.py issues it, over RCCL (backend "nccl"), with the only group a one-GPU box allows: one
    rank.  A buffer packed on a side stream, handed to the collective on the current stream through an event and
    record_stream, all_gather_into_tensor, merge -- the single-rank table back record for record.  (Two ranks on one
    device are refused by RCCL; the N > 1 logic is covered by the gloo tests and tools/rehearse_multi_gpu.sh.)"""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from eigensolver_amd import ShootProblem, _lib, equilibrium as q
    from eigensolver_amd import distributed as D
    if dist.is_initialized():
        pytest.skip("a process group is already up in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    try:
        side = torch.cuda.Stream()
        cx = _lib.Context(0, stream=side)
        gp = ShootProblem(q.CylinderFlow(U_i0=0.6, width=1.0), "kink", ctx=cx)
        k = np.linspace(0.4, 3.9, 12)
        W = 2.7 + (np.arange(160) + 0.5) * (4.95 - 2.7) / 160
        with torch.cuda.stream(side):
            Dg, st = gp.eval_grid(k, W)
            roots, cnt = gp.find_roots(k, W, Dg, st, n_bisect=16)
            rows = torch.arange(len(k), device="cuda")
            send = D.pack_fixed(roots, cnt, 1, rows, 256, ctx=cx)
            done = torch.cuda.Event()
            done.record(side)
        cur = torch.cuda.current_stream()
        cur.wait_event(done)
        send.record_stream(cur)
        out = torch.empty((1 * send.shape[0], send.shape[1]), dtype=send.dtype, device=send.device)
        dist.all_gather_into_tensor(out, send)
        torch.cuda.synchronize()
        assert dist.get_backend() == "nccl"
        rec, counts = D.merge_fixed(out.view(1, send.shape[0], send.shape[1]))
        assert counts == [cnt] and cnt > 5
        assert np.array_equal(rec[:, 1], roots["w"].cpu().numpy()) and np.array_equal(rec[:, 0], roots["k"].cpu().numpy())
        # and through the library's own helper with the group up (world taken from the group)
        rec2, _ = D.merge_fixed(D.gather_fixed(send))
        assert np.array_equal(rec2, rec)
        gp.close()
    finally:
        dist.destroy_process_group()
