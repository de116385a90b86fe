"""Error behaviour of the C ABI on a device: invalid arguments are reported through return codes / EsError with a
message, never by crashing or by silently computing something else."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_invalid_arguments_are_rejected(es_ctx):
    import torch
    from eigensolver_amd import _lib, ShootProblem, equilibrium as q
    lib = es_ctx.lib
    # unknown mode of the analytic kernel
    p = _lib.SlabAnalyticParams(1, 2 / 3, 0, 0.75, 0, -0.15, 2.27, 0.55, 0.0)
    K = torch.ones(4, dtype=torch.float64, device="cuda")
    D = torch.empty(16, dtype=torch.float64, device="cuda")
    assert lib.es_slab_analytic_eval(es_ctx.handle, C.byref(p), 7, _lib.ptr(K), 4, _lib.ptr(K), 4, _lib.ptr(D)) == 1
    assert b"mode" in lib.es_last_error(es_ctx.handle)
    assert lib.es_slab_analytic_eval(es_ctx.handle, C.byref(p), 0, _lib.ptr(K), -1, _lib.ptr(K), 4, _lib.ptr(D)) == 1
    assert lib.es_slab_analytic_eval(es_ctx.handle, C.byref(p), 0, None, 4, _lib.ptr(K), 4, _lib.ptr(D)) == 1
    # problem creation: bad geometry / node count / boundary / missing profiles
    eq = q.CylinderFlow()
    from eigensolver_amd.shooting import make_desc
    d, prof = make_desc(eq, "kink")
    pr = _lib.Profiles()
    keep = {k: np.ascontiguousarray(v) for k, v in prof.items()}
    for name in _lib._PROFILE_FIELDS:
        a = keep.get(name)
        setattr(pr, name, a.ctypes.data if a is not None else None)
    h = C.c_void_p()
    for field, bad in (("geometry", 9), ("n_nodes", 1), ("x_boundary", 0.5), ("axis_bc", 5), ("c1_power", 3)):
        d2 = _lib.ShootDesc.from_buffer_copy(d)
        setattr(d2, field, bad)
        assert lib.es_problem_create(es_ctx.handle, C.byref(d2), C.byref(pr), C.byref(h)) == 1, field
    pr2 = _lib.Profiles()
    assert lib.es_problem_create(es_ctx.handle, C.byref(d), C.byref(pr2), C.byref(h)) == 1       # no profiles
    # twisted profile handed to the un-twisted geometry
    tw = q.CylinderRotation()
    d3, prof3 = make_desc(tw, "kink")
    d3.geometry = 0
    keep3 = {k: np.ascontiguousarray(v) for k, v in prof3.items()}
    pr3 = _lib.Profiles()
    for name in _lib._PROFILE_FIELDS:
        a = keep3.get(name)
        setattr(pr3, name, a.ctypes.data if a is not None else None)
    assert lib.es_problem_create(es_ctx.handle, C.byref(d3), C.byref(pr3), C.byref(h)) == 1
    # evaluation entry points
    gp = ShootProblem(eq, "kink", ctx=es_ctx)
    with pytest.raises(_lib.EsError):
        gp.eval_grid([1.0], [3.0], w_mode=7)
    st = torch.empty(1, dtype=torch.uint8, device="cuda")
    assert lib.es_shoot_eval_points(es_ctx.handle, None, _lib.ptr(K), _lib.ptr(K), 1, _lib.ptr(D), None, _lib.ptr(st)) == 1
    ws = _lib.WorkerSpec(1.0, -1, 10, 0, 0, 0, 0)
    n = torch.zeros(1, dtype=torch.int32, device="cuda")
    assert lib.es_worker_run(es_ctx.handle, gp.handle, C.byref(ws), _lib.ptr(K), 1, _lib.ptr(K), 1, _lib.ptr(D),
                             _lib.ptr(n), 4, None) == 1
    # the context is still usable after errors
    Dg, stg = gp.eval_grid([1.0, 2.0], [3.0, 3.5])
    assert torch.isfinite(Dg).all()
    gp.close()


def test_round2_entry_points_reject_bad_input_and_handle_edges(es_ctx):
    """es_shoot_eval_grid_ex, es_shoot_find_roots_mixed, es_root_table_pack: unknown flags, null pointers, a root
    table that is too small (count still returned, first records valid), empty and one-column grids."""
    import torch
    from eigensolver_amd import _lib, ShootProblem, equilibrium as q
    lib = es_ctx.lib
    eq = q.CylinderFlow(U_i0=0.6, width=1.0)
    gp = ShootProblem(eq, "kink", ctx=es_ctx)
    k = torch.linspace(0.5, 3.5, 6, dtype=torch.float64, device="cuda")
    W = torch.linspace(0.95, 4.9, 97, dtype=torch.float64, device="cuda")
    D = torch.empty((6, 97), dtype=torch.float64, device="cuda")
    st = torch.empty((6, 97), dtype=torch.uint8, device="cuda")
    args = (es_ctx.handle, gp.handle, _lib.ptr(k), 6, _lib.ptr(W), 97, 1)
    assert lib.es_shoot_eval_grid_ex(*args, 8, _lib.ptr(D), None, _lib.ptr(st)) == 1          # unknown flag bit
    assert b"flags" in lib.es_last_error(es_ctx.handle)
    assert lib.es_shoot_eval_grid_ex(*args, 1, None, None, _lib.ptr(st)) == 1                  # null output
    assert lib.es_shoot_eval_grid_ex(*args, 1, _lib.ptr(D), None, _lib.ptr(st)) == 0
    # skip flag on grids narrower than a wave and with a single column / row
    for nk, nw in ((1, 1), (2, 5), (1, 64)):
        Ds, ss = gp.eval_grid(k[:nk], W[:nw], skip_continuum=True)
        D0, s0 = gp.eval_grid(k[:nk], W[:nw])
        assert torch.equal(ss, s0)
        keep = s0 != 3
        assert torch.equal(Ds[keep].nan_to_num(7.0), D0[keep].nan_to_num(7.0))
    # mixed search: capacity below the number of brackets
    full, cnt, _, _, stats = gp.find_roots_mixed(k, W, n_bisect=16)
    assert cnt > 4 and stats[2] == 0
    small, cnt2, _, _, _ = gp.find_roots_mixed(k, W, n_bisect=16, capacity=3)
    assert cnt2 == cnt and small["w"].numel() == 3 and torch.equal(small["w"], full["w"][:3])
    # null table
    n = C.c_int(0)
    assert lib.es_shoot_find_roots_mixed(es_ctx.handle, gp.handle, _lib.ptr(k), 6, _lib.ptr(W), 97, 1, 16, 1e-3,
                                         _lib.ptr(D), _lib.ptr(st), None, C.byref(n), None) == 1
    # pack: count larger than the table it points to
    t, rt = gp.alloc_root_table(4)
    out = torch.empty((9, 6), dtype=torch.float64, device="cuda")
    assert lib.es_root_table_pack(es_ctx.handle, C.byref(rt), 6, 1.0, None, 8, _lib.ptr(out)) == 1
    assert lib.es_root_table_pack(es_ctx.handle, C.byref(rt), 0, 1.0, None, 8, _lib.ptr(out)) == 0
    es_ctx.synchronize()
    assert float(out[0, 0]) == 0.0 and float(out.abs().sum()) == 0.0
    gp.close()


def test_round3_entry_points_reject_bad_arguments(es_ctx, monkeypatch):
    """es_shoot_find_roots_async, es_root_table_pack_async, es_shoot_grid_shape, es_shoot_screen_grid /
    es_shoot_find_roots_screened, es_context_grid_time: null pointers, sizes, families; an unbuilt ES_GRID_SHAPE is an
    error with a message, not a silent fall-back; empty grids are fine."""
    import torch
    from eigensolver_amd import _lib, ShootProblem, equilibrium as q
    lib = es_ctx.lib
    gp = ShootProblem(q.CylinderFlow(U_i0=0.6, width=1.0), "kink", ctx=es_ctx)
    k = torch.linspace(0.5, 3.0, 4, dtype=torch.float64, device="cuda")
    W = torch.linspace(2.8, 4.9, 32, dtype=torch.float64, device="cuda")
    D, st = gp.eval_grid(k, W)
    t, rt = gp.alloc_root_table(64)
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    args = (es_ctx.handle, gp.handle, _lib.ptr(k), 4, _lib.ptr(W), 32, 1, _lib.ptr(D), _lib.ptr(st), 16, 1e-3, C.byref(rt))
    assert lib.es_shoot_find_roots_async(*args, None) == 1                                     # no count word
    assert lib.es_shoot_find_roots_async(*args[:6], 9, *args[7:], _lib.ptr(cnt)) == 1          # w_mode
    assert lib.es_shoot_find_roots_async(None, *args[1:], _lib.ptr(cnt)) == 1
    assert lib.es_shoot_find_roots_async(*args, _lib.ptr(cnt)) == 0
    # empty grid: count zeroed on the device, nothing launched
    cnt.fill_(7)
    assert lib.es_shoot_find_roots_async(es_ctx.handle, gp.handle, _lib.ptr(k), 0, _lib.ptr(W), 32, 1, _lib.ptr(D), _lib.ptr(st),
                                         16, 1e-3, C.byref(rt), _lib.ptr(cnt)) == 0
    torch.cuda.synchronize()
    assert int(cnt.item()) == 0
    out = torch.empty((9, 6), dtype=torch.float64, device="cuda")
    assert lib.es_root_table_pack_async(es_ctx.handle, C.byref(rt), None, 1.0, None, 8, _lib.ptr(out)) == 1
    assert lib.es_root_table_pack_async(es_ctx.handle, C.byref(rt), _lib.ptr(cnt), 1.0, None, 8, _lib.ptr(out)) == 0
    a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
    assert lib.es_shoot_grid_shape(es_ctx.handle, gp.handle, 0, C.byref(a), C.byref(b), C.byref(c)) == 1
    assert lib.es_shoot_grid_shape(es_ctx.handle, gp.handle, 1024, C.byref(a), C.byref(b), C.byref(c)) == 0
    assert a.value in (1, 2, 4) and b.value in (2, 3, 4)
    ms, n = C.c_double(0), C.c_int(0)
    assert lib.es_context_grid_time(es_ctx.handle, None, C.byref(n)) == 1
    # a launch shape outside the build is refused
    monkeypatch.setenv("ES_GRID_SHAPE", "1,2")
    with pytest.raises(_lib.EsError, match="shape"):
        gp.eval_grid(k, W)
    monkeypatch.delenv("ES_GRID_SHAPE")
    gp.eval_grid(k, W)
    # the halves of the mixed search: slabs are unsupported, the screened half needs table and count
    monkeypatch.setenv("ES_FORCE_SIGN_TRACKING", "1")                  # a slab without bands: no fp32 screening
    gs = ShootProblem(q.SlabFlow(U_i0=0.35, width=1.5), "kink", ctx=es_ctx)
    monkeypatch.delenv("ES_FORCE_SIGN_TRACKING")
    with pytest.raises(_lib.EsError, match="unsupported"):
        gs.screen_grid(k, W)
    Ds, ss = gp.screen_grid(k, W)
    nn = C.c_int(0)
    assert lib.es_shoot_find_roots_screened(es_ctx.handle, gp.handle, _lib.ptr(k), 4, _lib.ptr(W), 32, 1, 16, 1e-3, _lib.ptr(Ds),
                                            _lib.ptr(ss), None, C.byref(nn), None) == 1
    r1 = gp.find_roots_screened(k, W, Ds, ss, n_bisect=16, table=(t, rt))
    r2 = gp.find_roots_mixed(k, W, n_bisect=16)
    assert r1[1] == r2[1] and torch.equal(r1[0]["w"], r2[0]["w"])
    gs.close()
    gp.close()
