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
