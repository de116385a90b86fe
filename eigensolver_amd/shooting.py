"""Host side of the shooting path: builds the es_shoot_desc / profile tables from an equilibrium object and calls
the HIP kernels through the C ABI (es_problem_create, es_shoot_eval_grid, es_shoot_eval_points,
es_shoot_find_roots)."""
import ctypes as C

import numpy as np

from . import _lib
from . import equilibrium as eqm

GEOM_CYL, GEOM_CYL_TWIST, GEOM_SLAB_DENSITY, GEOM_SLAB_FLOW = 0, 1, 2, 3
AXIS_KINK, AXIS_SAUSAGE, AXIS_ROTATION_KINK = 0, 1, 2
W_ABSOLUTE, W_PHASE_SPEED, W_PER_ROW = 0, 1, 2
PT_OK, PT_LEAKY, PT_NONFINITE, PT_CONTINUUM = 0, 1, 2, 3


def make_desc(eq, mode, m=None):
    """es_shoot_desc + profile dict for equilibrium `eq` and mode "kink" / "sausage" (azimuthal order m for
    cylinders defaults to the reference's 1 / 0)."""
    d = _lib.ShootDesc()
    d.n_nodes = int(eq.n_nodes)
    d.x_boundary, d.x_end = float(eq.x_boundary), float(eq.x_end)
    d.rho_e, d.vA_e, d.c_e, d.cT_e, d.U_e = eq.rho_e, eq.vA_e, eq.c_e, eq.cT_e, eq.U_e
    d.L_factor = eq.L_factor
    d.ic_value, d.ic_slope = eq.ic
    prof = eq.profiles()
    if isinstance(eq, eqm._CylinderBase):
        d.geometry = GEOM_CYL_TWIST if eq.twisted else GEOM_CYL
        mm = (1 if mode == "kink" else 0) if m is None else int(m)
        d.m = mm
        d.m_ext = mm                      # the reference hard-codes 1 / 0 in the exterior ODE (CF:769, :1065)
        if mode == "sausage":
            d.axis_bc = AXIS_SAUSAGE
        elif eq.twisted:
            d.axis_bc = AXIS_ROTATION_KINK
        else:
            d.axis_bc = AXIS_KINK
        d.c1_power = eq.c1_power
        d.bc_const = eq.bc_const(d.axis_bc)
    elif isinstance(eq, eqm.SlabDensity):
        d.geometry = GEOM_SLAB_DENSITY
        d.slab_mode = 0 if mode == "sausage" else 1
    elif isinstance(eq, eqm.SlabFlow):
        d.geometry = GEOM_SLAB_FLOW
        d.slab_mode = 0 if mode == "sausage" else 1
        d.c_i, d.vA_i, d.rho_i = eq.c_i0, eq.vA_i0, eq.rho_i0
    else:
        raise TypeError(type(eq))
    return d, prof


class ShootProblem:
    """One reference worker configuration resident on the GPU (profile tables in HBM)."""

    def __init__(self, eq, mode, m=None, ctx=None, accept_norm=0):
        """accept_norm = 1: `rel` is normalised by |outer| only (CR-KS:722) instead of max(|outer|, |inner|)."""
        self.ctx = ctx if ctx is not None else _lib.Context()
        self.eq, self.mode = eq, mode
        self.desc, prof = make_desc(eq, mode, m)
        self.desc.accept_norm = int(accept_norm)
        self._prof_np = {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in prof.items()}
        p = _lib.Profiles()
        for name in _lib._PROFILE_FIELDS:
            a = self._prof_np.get(name)
            setattr(p, name, a.ctypes.data if a is not None else None)
        h = C.c_void_p()
        st = self.ctx.lib.es_problem_create(self.ctx.handle, C.byref(self.desc), C.byref(p), C.byref(h))
        _lib.check(self.ctx.handle, st)
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.ctx.lib.es_problem_destroy(self.ctx.handle, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _dev(self, a):
        import torch
        dev = f"cuda:{self.ctx.device}"
        if isinstance(a, torch.Tensor):
            return a.to(device=dev, dtype=torch.float64).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)

    def eval_grid(self, k, w, w_mode=W_PHASE_SPEED, want_rel=False, skip_continuum=False):
        """D[ik, iw], status[ik, iw] (and rel) on the (k, omega) grid; w_mode selects how omega is formed.
        skip_continuum: ES_EVAL_SKIP_CONTINUUM -- points inside a continuum band get D = NaN and are not marched."""
        import torch
        dk, dw = self._dev(k).reshape(-1), self._dev(w)
        nk = dk.numel()
        nw = dw.shape[-1] if w_mode == W_PER_ROW else dw.numel()
        if w_mode == W_PER_ROW:
            assert dw.numel() == nk * nw
        D = torch.empty((nk, nw), dtype=torch.float64, device=dk.device)
        st = torch.empty((nk, nw), dtype=torch.uint8, device=dk.device)
        rel = torch.empty((nk, nw), dtype=torch.float64, device=dk.device) if want_rel else None
        rc = self.ctx.lib.es_shoot_eval_grid_ex(self.ctx.handle, self.handle, _lib.ptr(dk), nk, _lib.ptr(dw), nw,
                                                w_mode, 1 if skip_continuum else 0, _lib.ptr(D),
                                                _lib.ptr(rel) if want_rel else None, _lib.ptr(st))
        _lib.check(self.ctx.handle, rc)
        return (D, st, rel) if want_rel else (D, st)

    def grid_kernel_name(self, nw):
        """The instantiation of the grid kernel es_shoot_eval_grid launches for rows of nw frequencies (es_shoot_grid_shape),
        spelled as tools/codeobj_table.py and rocprofv3 print it."""
        pts, wpe, trk = C.c_int(0), C.c_int(0), C.c_int(0)
        _lib.check(self.ctx.handle, self.ctx.lib.es_shoot_grid_shape(self.ctx.handle, self.handle, int(nw), C.byref(pts),
                                                                     C.byref(wpe), C.byref(trk)))
        if pts.value < 0:                                   # two k-rows per workgroup (es_shoot_grid_shape)
            return f"shoot_grid_kernel_r2<{int(self.desc.geometry)},{-pts.value},{'true' if trk.value else 'false'},{wpe.value}>"
        return f"shoot_grid_kernel<{int(self.desc.geometry)},{pts.value},256,{'true' if trk.value else 'false'},{wpe.value}>"

    def eval_points(self, k, w, want_rel=False):
        import torch
        dk, dw = self._dev(k).reshape(-1), self._dev(w).reshape(-1)
        n = dk.numel()
        assert dw.numel() == n
        D = torch.empty(n, dtype=torch.float64, device=dk.device)
        st = torch.empty(n, dtype=torch.uint8, device=dk.device)
        rel = torch.empty(n, dtype=torch.float64, device=dk.device) if want_rel else None
        rc = self.ctx.lib.es_shoot_eval_points(self.ctx.handle, self.handle, _lib.ptr(dk), _lib.ptr(dw), n,
                                               _lib.ptr(D), _lib.ptr(rel) if want_rel else None, _lib.ptr(st))
        _lib.check(self.ctx.handle, rc)
        return (D, st, rel) if want_rel else (D, st)

    def eigenfunction(self, k, w, n_ext=500):
        """Two-region solution at the (k, omega) pairs: dict of CUDA tensors x_int [N], value_int/flux_int [n, N],
        x_ext/value_ext/flux_ext [n, n_ext] (cylinders: P and xi_r; slabs: Vx and P_T), exterior boundary value +-1."""
        import torch
        dk, dw = self._dev(k).reshape(-1), self._dev(w).reshape(-1)
        n, N = dk.numel(), int(self.desc.n_nodes)
        dev = dk.device
        vi = torch.empty((n, N), dtype=torch.float64, device=dev)
        fi = torch.empty((n, N), dtype=torch.float64, device=dev)
        xe = torch.empty((n, n_ext), dtype=torch.float64, device=dev)
        ve = torch.empty((n, n_ext), dtype=torch.float64, device=dev)
        fe = torch.empty((n, n_ext), dtype=torch.float64, device=dev)
        rc = self.ctx.lib.es_shoot_eigenfunction(self.ctx.handle, self.handle, _lib.ptr(dk), _lib.ptr(dw), n,
                                                 _lib.ptr(vi), _lib.ptr(fi), int(n_ext), _lib.ptr(xe), _lib.ptr(ve),
                                                 _lib.ptr(fe))
        _lib.check(self.ctx.handle, rc)
        x_int = torch.linspace(self.desc.x_boundary, self.desc.x_end, N, dtype=torch.float64, device=dev)
        return dict(x_int=x_int, value_int=vi, flux_int=fi, x_ext=xe, value_ext=ve, flux_ext=fe)

    def alloc_root_table(self, capacity):
        import torch
        dev = f"cuda:{self.ctx.device}"
        t = {n: torch.empty(capacity, dtype=torch.float64, device=dev) for n in ("k", "w", "w_lo", "w_hi", "resid")}
        t["row"] = torch.empty(capacity, dtype=torch.int32, device=dev)
        t["flag"] = torch.empty(capacity, dtype=torch.uint8, device=dev)
        rt = _lib.RootTable(t["k"].data_ptr(), t["w"].data_ptr(), t["w_lo"].data_ptr(), t["w_hi"].data_ptr(),
                            t["resid"].data_ptr(), t["row"].data_ptr(), t["flag"].data_ptr(), capacity)
        return t, rt

    def find_roots(self, k, w, D, status, w_mode=W_PHASE_SPEED, n_bisect=40, tol_percent=1e-3, capacity=None,
                   table=None):
        """Brackets + bisection + classification on the grid evaluated by eval_grid. Returns (dict, count)."""
        dk, dw = self._dev(k).reshape(-1), self._dev(w)
        nk = dk.numel()
        nw = dw.shape[-1] if w_mode == W_PER_ROW else dw.numel()
        cap = int(capacity) if capacity is not None else max(1024, 16 * nk)
        while True:
            t, rt = table if table is not None else self.alloc_root_table(cap)
            n = C.c_int(0)
            rc = self.ctx.lib.es_shoot_find_roots(self.ctx.handle, self.handle, _lib.ptr(dk), nk, _lib.ptr(dw), nw,
                                                  w_mode, _lib.ptr(D), _lib.ptr(status), int(n_bisect),
                                                  float(tol_percent), C.byref(rt), C.byref(n))
            _lib.check(self.ctx.handle, rc, allow_capacity=True)
            if rc == 3 and capacity is None and table is None:
                cap = n.value
                continue
            m = min(n.value, rt.capacity)
            return {key: v[:m] for key, v in t.items()}, n.value

    def find_roots_async(self, k, w, D, status, table, count, w_mode=W_PHASE_SPEED, n_bisect=40, tol_percent=1e-3):
        """es_shoot_find_roots_async: everything enqueued on the context's stream, nothing read back.  `table` is
        (dict, RootTable) from alloc_root_table, `count` an int32 CUDA tensor of one element that receives the bracket
        count (it may exceed the capacity: check when reading it).  Returns the full-capacity dict of the table."""
        dk, dw = self._dev(k).reshape(-1), self._dev(w)
        nk = dk.numel()
        nw = dw.shape[-1] if w_mode == W_PER_ROW else dw.numel()
        t, rt = table
        assert count.is_cuda and count.numel() == 1 and count.element_size() == 4
        rc = self.ctx.lib.es_shoot_find_roots_async(self.ctx.handle, self.handle, _lib.ptr(dk), nk, _lib.ptr(dw), nw,
                                                    w_mode, _lib.ptr(D), _lib.ptr(status), int(n_bisect),
                                                    float(tol_percent), C.byref(rt), _lib.ptr(count))
        _lib.check(self.ctx.handle, rc)
        return t

    def screen_grid(self, k, w, w_mode=W_PHASE_SPEED):
        """Step 1 of the mixed search alone (es_shoot_screen_grid): the fp32 screening march, enqueued.  Returns the
        screened (D, status) for find_roots_screened."""
        import torch
        dk, dw = self._dev(k).reshape(-1), self._dev(w)
        nk = dk.numel()
        nw = dw.shape[-1] if w_mode == W_PER_ROW else dw.numel()
        D = torch.empty((nk, nw), dtype=torch.float64, device=dk.device)
        st = torch.empty((nk, nw), dtype=torch.uint8, device=dk.device)
        rc = self.ctx.lib.es_shoot_screen_grid(self.ctx.handle, self.handle, _lib.ptr(dk), nk, _lib.ptr(dw), nw, w_mode,
                                               _lib.ptr(D), _lib.ptr(st))
        _lib.check(self.ctx.handle, rc)
        return D, st

    def find_roots_screened(self, k, w, D, st, w_mode=W_PHASE_SPEED, n_bisect=40, tol_percent=1e-3, table=None, capacity=None):
        """Steps 2 - 5 of the mixed search on a grid screened by screen_grid; returns what find_roots_mixed returns."""
        dk, dw = self._dev(k).reshape(-1), self._dev(w)
        nk = dk.numel()
        nw = dw.shape[-1] if w_mode == W_PER_ROW else dw.numel()
        t, rt = table if table is not None else self.alloc_root_table(int(capacity) if capacity is not None else max(1024, 16 * nk))
        n = C.c_int(0)
        stats = (C.c_int * 3)()
        rc = self.ctx.lib.es_shoot_find_roots_screened(self.ctx.handle, self.handle, _lib.ptr(dk), nk, _lib.ptr(dw), nw,
                                                       w_mode, int(n_bisect), float(tol_percent), _lib.ptr(D), _lib.ptr(st),
                                                       C.byref(rt), C.byref(n), stats)
        _lib.check(self.ctx.handle, rc, allow_capacity=True)
        m = min(n.value, rt.capacity)
        return {key: v[:m] for key, v in t.items()}, n.value, D, st, tuple(stats)

    def find_roots_mixed(self, k, w, w_mode=W_PHASE_SPEED, n_bisect=40, tol_percent=1e-3, capacity=None, table=None):
        """fp32 screening of the grid + fp64 re-evaluation of every unsure point and of both ends of every bracket +
        fp64 refinement (es_shoot_find_roots_mixed).  Returns (root dict, bracket count, D, status, stats) with
        stats = (fp64 re-evaluations of unsure grid points, of bracket ends, unconfirmed brackets)."""
        import torch
        dk, dw = self._dev(k).reshape(-1), self._dev(w)
        nk = dk.numel()
        nw = dw.shape[-1] if w_mode == W_PER_ROW else dw.numel()
        D = torch.empty((nk, nw), dtype=torch.float64, device=dk.device)
        st = torch.empty((nk, nw), dtype=torch.uint8, device=dk.device)
        cap = int(capacity) if capacity is not None else max(1024, 16 * nk)
        while True:
            t, rt = table if table is not None else self.alloc_root_table(cap)
            n = C.c_int(0)
            stats = (C.c_int * 3)()
            rc = self.ctx.lib.es_shoot_find_roots_mixed(self.ctx.handle, self.handle, _lib.ptr(dk), nk, _lib.ptr(dw), nw,
                                                        w_mode, int(n_bisect), float(tol_percent), _lib.ptr(D),
                                                        _lib.ptr(st), C.byref(rt), C.byref(n), stats)
            _lib.check(self.ctx.handle, rc, allow_capacity=True)
            if rc == 3 and capacity is None and table is None:
                cap = n.value
                continue
            m = min(n.value, rt.capacity)
            return {key: v[:m] for key, v in t.items()}, n.value, D, st, tuple(stats)
