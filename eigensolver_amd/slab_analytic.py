"""Host-side mirror of the analytic part of Slab/Non uniform flow/Solver/flow_multiprocessor.py.

The reference keeps this as module-level code: constants (:63-99), m0/me/n0 and the four disp_rel_* functions
(:107-127), grids (:131-152), the scan loops (:166-272) and the pole filter (:284-303).  Here the same
quantities live in one object and the loops run on the GPU through the C ABI (es_slab_analytic_*).
"""
import ctypes as C
import math
import numpy as np

from . import _lib

SAUSAGE, KINK, SAUSAGE_BODY, KINK_BODY = 0, 1, 2, 3
GAMMA = 5.0 / 3.0


class SlabSteadyFlow:
    """Uniform slab with steady flow: closed-form dispersion relations D(W = omega/k, K = k x0)."""

    def __init__(self, vA_i=1.0, c_i=2.0 / 3.0, vA_e=0.0, c_e=0.75, U_i=0.0, U_e=-0.15, ctx=None):
        self.vA_i, self.c_i, self.vA_e, self.c_e, self.U_i, self.U_e = vA_i, c_i, vA_e, c_e, U_i, U_e
        rho_i = 1.0
        rho_e = rho_i * (c_i ** 2 + GAMMA * 0.5 * vA_i ** 2) / (c_e ** 2 + GAMMA * 0.5 * vA_e ** 2)   # :74
        self.R1 = rho_e / rho_i                                                                      # :79
        self.cT_i = math.sqrt(c_i ** 2 / (c_i ** 2 + vA_i ** 2))                                     # :85-86
        self.cT_e = math.sqrt(c_e ** 2 * vA_e ** 2 / vA_i ** 2 * (c_e ** 2 + vA_e ** 2))             # :88-89 (as written)
        self.ctx = ctx
        self.params = _lib.SlabAnalyticParams(vA_i, c_i, vA_e, c_e, U_i, U_e, self.R1, self.cT_i, self.cT_e)

    def _ctx(self):
        if self.ctx is None:
            self.ctx = _lib.Context()
        return self.ctx

    def _dev(self, a):
        import torch
        ctx = self._ctx()
        if isinstance(a, torch.Tensor):
            return a.to(device=f"cuda:{ctx.device}", dtype=torch.float64).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=f"cuda:{ctx.device}")

    def disp_rel(self, mode, W, K):
        """D[iK, iW] = disp_rel_<mode>(W[iW], K[iK]) as a torch.float64 CUDA tensor."""
        import torch
        ctx = self._ctx()
        dK, dW = self._dev(K).reshape(-1), self._dev(W).reshape(-1)
        D = torch.empty((dK.numel(), dW.numel()), dtype=torch.float64, device=dK.device)
        st = ctx.lib.es_slab_analytic_eval(ctx.handle, C.byref(self.params), mode, _lib.ptr(dK), dK.numel(),
                                           _lib.ptr(dW), dW.numel(), _lib.ptr(D))
        _lib.check(ctx.handle, st)
        return D

    def scan(self, mode, K, W, step, capacity=None):
        """Sign-change scan (:166-272). Returns (K_out, W_mid) CUDA tensors in the reference's loop order."""
        import torch
        ctx = self._ctx()
        dK, dW = self._dev(K).reshape(-1), self._dev(W).reshape(-1)
        cap = int(capacity) if capacity is not None else max(1024, 4 * dK.numel())
        while True:
            rK = torch.empty(cap, dtype=torch.float64, device=dK.device)
            rW = torch.empty(cap, dtype=torch.float64, device=dK.device)
            n = C.c_int(0)
            st = ctx.lib.es_slab_analytic_scan(ctx.handle, C.byref(self.params), mode, _lib.ptr(dK), dK.numel(),
                                               _lib.ptr(dW), dW.numel(), float(step), _lib.ptr(rK), _lib.ptr(rW),
                                               cap, C.byref(n))
            _lib.check(ctx.handle, st, allow_capacity=True)
            if st == 3 and capacity is None:
                cap = n.value
                continue
            return rK[:min(n.value, cap)], rW[:min(n.value, cap)], n.value

    def pole_filter(self, mode, rK, rW, thresh=1e-4):
        """:284-303 keep candidates with disp_rel(W, K) < thresh."""
        import torch
        ctx = self._ctx()
        rK, rW = self._dev(rK), self._dev(rW)
        keep = torch.empty(rK.numel(), dtype=torch.uint8, device=rK.device)
        st = ctx.lib.es_slab_analytic_filter(ctx.handle, C.byref(self.params), mode, _lib.ptr(rK), _lib.ptr(rW),
                                             rK.numel(), float(thresh), _lib.ptr(keep))
        _lib.check(ctx.handle, st)
        k = keep.bool()
        return rK[k], rW[k]
