"""Uniform cylinder in closed form (Bessel I/K/J/Y on the GPU, es_cyl_uniform_eval): the benchmark case of the
reference's cylinder scripts (profile width 1e5) without integrating any ODE."""
import ctypes as C

import numpy as np

from . import _lib
from . import equilibrium as eqm


class CylinderUniform:
    def __init__(self, eq=None, mode="kink", m=None, U_i=0.0, ctx=None):
        self.ctx = ctx if ctx is not None else _lib.Context()
        eq = eq if eq is not None else eqm.CylinderFlow()
        mm = (1 if mode == "kink" else 0) if m is None else int(m)
        self.params = _lib.CylUniformParams(eq.c_i0, eq.vA_i0, eq.rho_i0, float(U_i), eq.rho_e, eq.vA_e, eq.c_e,
                                            eq.cT_e, eq.x_boundary, eq.r_axis, eq.L_factor, eq.ic[0], eq.ic[1],
                                            mm, mm, 1 if mode == "sausage" else 0, 0)

    def eval_grid(self, k, w, w_mode=1, want_rel=False):
        import torch
        dev = f"cuda:{self.ctx.device}"
        def to_dev(a):
            if isinstance(a, torch.Tensor):
                return a.to(device=dev, dtype=torch.float64).contiguous()
            return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
        dk = to_dev(k).reshape(-1)
        dw = to_dev(w)
        nk = dk.numel()
        nw = dw.shape[-1] if w_mode == 2 else dw.numel()
        D = torch.empty((nk, nw), dtype=torch.float64, device=dev)
        st = torch.empty((nk, nw), dtype=torch.uint8, device=dev)
        rel = torch.empty((nk, nw), dtype=torch.float64, device=dev) if want_rel else None
        rc = self.ctx.lib.es_cyl_uniform_eval(self.ctx.handle, C.byref(self.params), _lib.ptr(dk), nk, _lib.ptr(dw),
                                              nw, w_mode, _lib.ptr(D), _lib.ptr(rel) if want_rel else None,
                                              _lib.ptr(st))
        _lib.check(self.ctx.handle, rc)
        return (D, st, rel) if want_rel else (D, st)
