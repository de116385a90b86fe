"""Host-side mirror of the reference's complex-frequency workers (unstable / Kelvin-Helmholtz modes of the flow slab)

    Slab/Non uniform flow/COMPLEX ANALYSIS/flow_multiprocessor_complex_coronal.py
      :348  sausage(wavenumber, sausage_ws, sausage_ks, sausage_ws_imag, sausage_ks_imag, freq)
      :737  kink(wavenumber, kink_ws, kink_ks, kink_ws_imag, kink_ks_imag, freq)
      :1127 driver: freq = linspace(s_i k, s_{i+1} k, 10) + 1j * linspace(-0.25, 0.25, 10)

over the C ABI of include/eigensolver_amd.h section (6).  `freq` is the complex array the reference's driver builds:
its real parts are the Re(omega) samples and its imaginary parts the Im(omega) samples of a rectangular grid.  (The
reference forms `freq[j] + 1j*freq[m]` from that array, which shears the grid; the rectangular reading is the one its
comments and plots describe.)  A root is where the complex mismatch D_c vanishes -- the reference accepts a grid point
whose REAL part of the mismatch is below p_tol = 4 %; here every grid cell around which D_c winds once is refined by
complex secant steps and accepted if 100 |D_c| / max(|outer|, |inner|) < tol.
"""
import ctypes as C
import math
from dataclasses import dataclass

import numpy as np

from . import _lib, equilibrium as eqm
from .shooting import ShootProblem, W_ABSOLUTE, W_PHASE_SPEED

CX_SFX, CX_SFG = 0, 1


@dataclass
class SlabFlowKH(eqm.SlabFlow):
    """Constants of the complex script (SF-X:96-121): densities given, c_e from them (not from pressure balance)."""
    c_i0: float = 1.3
    vA_i0: float = 1.0
    vA_e: float = 0.0
    c_e: float = float("nan")
    rho_i0: float = 9.0
    rho_e_value: float = 5.0
    U_i0: float = 1.4
    U_e: float = 0.0

    def __post_init__(self):
        if math.isnan(self.c_e):
            self.c_e = math.sqrt((self.rho_i0 / self.rho_e_value) * self.c_i0 ** 2 + eqm.GAMMA * 0.5 * self.vA_i0 ** 2)   # SF-X:111

    @property
    def rho_e(self):
        return self.rho_e_value


def _distinct(w, tol=1e-8):
    """A root on a cell edge is found from both neighbouring cells: keep the first of each cluster."""
    keep = []
    for z in w:
        if all(abs(z - y) > tol * max(1.0, abs(z)) for y in keep):
            keep.append(z)
    return np.array(keep, dtype=complex)


class SlabComplexFlow:
    """Complex-frequency flow slab: D_c on (k, Re omega, Im omega) grids, roots, and the reference's worker signature."""
    P_TOL = 4.0                                     # SF-X:301

    def __init__(self, U_i0=1.4, width=1e5, variant="sfx", ctx=None, equilibrium=None, **kw):
        self.eq = equilibrium if equilibrium is not None else SlabFlowKH(U_i0=U_i0, width=width, **kw)
        self.variant = {"sfx": CX_SFX, "sfg": CX_SFG}[variant]
        self.ctx = ctx or _lib.Context()
        self._problems = {}

    def problem(self, mode):
        if mode not in self._problems:
            self._problems[mode] = ShootProblem(self.eq, mode, ctx=self.ctx)
        return self._problems[mode]

    def close(self):
        for p in self._problems.values():
            p.close()
        self._problems = {}

    # ---- evaluation ------------------------------------------------------------------------------------------------
    def eval_grid(self, mode, k, w_re, w_im, w_mode=W_ABSOLUTE):
        """D_c[nk, n_im, n_re] (complex128 tensor), status[nk, n_im, n_re], rel[nk, n_im, n_re]."""
        import torch
        p = self.problem(mode)
        dk, dre, dim = p._dev(k).reshape(-1), p._dev(w_re).reshape(-1), p._dev(w_im).reshape(-1)
        nk, nre, nim = dk.numel(), dre.numel(), dim.numel()
        Dre = torch.empty((nk, nim, nre), dtype=torch.float64, device=dk.device)
        Dim = torch.empty_like(Dre)
        rel = torch.empty_like(Dre)
        st = torch.empty((nk, nim, nre), dtype=torch.uint8, device=dk.device)
        rc = self.ctx.lib.es_complex_eval_grid(self.ctx.handle, p.handle, self.variant, _lib.ptr(dk), nk, _lib.ptr(dre), nre,
                                               _lib.ptr(dim), nim, int(w_mode), _lib.ptr(Dre), _lib.ptr(Dim),
                                               _lib.ptr(rel), _lib.ptr(st))
        _lib.check(self.ctx.handle, rc)
        return torch.complex(Dre, Dim), st, rel

    def eval_points(self, mode, k, w):
        import torch
        p = self.problem(mode)
        w = np.asarray(w, dtype=complex).reshape(-1)
        dk = p._dev(np.broadcast_to(np.asarray(k, dtype=float), w.shape).copy())
        dre, dim = p._dev(w.real.copy()), p._dev(w.imag.copy())
        n = dk.numel()
        Dre = torch.empty(n, dtype=torch.float64, device=dk.device)
        Dim, rel = torch.empty_like(Dre), torch.empty_like(Dre)
        st = torch.empty(n, dtype=torch.uint8, device=dk.device)
        rc = self.ctx.lib.es_complex_eval_points(self.ctx.handle, p.handle, self.variant, _lib.ptr(dk), _lib.ptr(dre),
                                                 _lib.ptr(dim), n, _lib.ptr(Dre), _lib.ptr(Dim), _lib.ptr(rel), _lib.ptr(st))
        _lib.check(self.ctx.handle, rc)
        return torch.complex(Dre, Dim), st, rel

    def find_roots(self, mode, k, w_re, w_im, D, status, w_mode=W_ABSOLUTE, n_iter=12, tol_percent=None, capacity=None):
        """Cells of the (Re, Im) grid around which D_c winds once, refined by complex secant steps.
        Returns ({k, w (complex), resid, row, flag}, count)."""
        import torch
        p = self.problem(mode)
        dk, dre, dim = p._dev(k).reshape(-1), p._dev(w_re).reshape(-1), p._dev(w_im).reshape(-1)
        nk, nre, nim = dk.numel(), dre.numel(), dim.numel()
        Dre, Dim = D.real.contiguous(), D.imag.contiguous()
        tol = self.P_TOL if tol_percent is None else float(tol_percent)
        cap = int(capacity) if capacity is not None else max(256, 4 * nk)
        while True:
            dev = dk.device
            t = {"k": torch.empty(cap, dtype=torch.float64, device=dev), "w_re": torch.empty(cap, dtype=torch.float64, device=dev),
                 "w_im": torch.empty(cap, dtype=torch.float64, device=dev), "resid": torch.empty(cap, dtype=torch.float64, device=dev),
                 "row": torch.empty(cap, dtype=torch.int32, device=dev), "flag": torch.empty(cap, dtype=torch.int32, device=dev)}
            rt = _lib.ComplexRootTable(_lib.ptr(t["k"]), _lib.ptr(t["w_re"]), _lib.ptr(t["w_im"]), _lib.ptr(t["resid"]),
                                       _lib.ptr(t["row"]), _lib.ptr(t["flag"]), cap)
            n = C.c_int(0)
            rc = self.ctx.lib.es_complex_find_roots(self.ctx.handle, p.handle, self.variant, _lib.ptr(dk), nk, _lib.ptr(dre), nre,
                                                    _lib.ptr(dim), nim, int(w_mode), _lib.ptr(Dre), _lib.ptr(Dim),
                                                    _lib.ptr(status), int(n_iter), tol, C.byref(rt), C.byref(n))
            _lib.check(self.ctx.handle, rc, allow_capacity=True)
            if rc == 3 and capacity is None:
                cap = n.value
                continue
            m = min(n.value, cap)
            out = {"k": t["k"][:m], "w": torch.complex(t["w_re"][:m], t["w_im"][:m]), "resid": t["resid"][:m],
                   "row": t["row"][:m], "flag": t["flag"][:m]}
            return out, n.value

    # ---- the reference's worker signature --------------------------------------------------------------------------
    def _worker(self, mode, wavenumber, ws, ks, ws_imag, ks_imag, freq):
        freq = np.asarray(freq, dtype=complex).reshape(-1)
        k = np.array([float(wavenumber)])
        w_re, w_im = np.ascontiguousarray(freq.real), np.ascontiguousarray(freq.imag)
        D, st, rel = self.eval_grid(mode, k, w_re, w_im, W_ABSOLUTE)
        roots, n = self.find_roots(mode, k, w_re, w_im, D, st, W_ABSOLUTE)
        ok = roots["flag"].cpu().numpy() == 1
        w = _distinct(roots["w"].cpu().numpy()[ok])
        kk = [float(wavenumber)] * len(w)
        ks.put(list(kk)); ws.put([float(x) for x in w.real])                       # SF-X:1095-1096
        ks_imag.put(list(kk)); ws_imag.put([float(x) for x in w.imag])             # SF-X:1097-1098

    def sausage(self, wavenumber, sausage_ws, sausage_ks, sausage_ws_imag, sausage_ks_imag, freq):
        self._worker("sausage", wavenumber, sausage_ws, sausage_ks, sausage_ws_imag, sausage_ks_imag, freq)

    def kink(self, wavenumber, kink_ws, kink_ks, kink_ws_imag, kink_ks_imag, freq):
        self._worker("kink", wavenumber, kink_ws, kink_ks, kink_ws_imag, kink_ks_imag, freq)

    def solve(self, wavenumbers, speeds=(-0.5, 0.0, 0.5, 1.0), n_re=10, im_range=(-0.25, 0.25), n_im=10, modes=("kink",),
              n_iter=12, tol_percent=None):
        """The driver block SF-X:1115-1135 in one batch per (mode, band): returns {mode: (omega complex array, k array)}."""
        ks = np.asarray(wavenumbers, dtype=float)
        sp = sorted(speeds)                                                        # SF-X:231-235
        out = {}
        for mode in modes:
            w_all, k_all = [], []
            for i in range(len(sp) - 1):
                W_re = np.linspace(sp[i], sp[i + 1], n_re)
                # Re(omega) scales with k but Im(omega) is absolute in the reference's driver (SF-X:1127), so the
                # (Re, Im) grid differs per k: one call per (k, band) with absolute frequencies.
                for k in ks:
                    w_re, w_im = W_re * k, np.linspace(im_range[0], im_range[1], n_im)
                    D, st, rel = self.eval_grid(mode, np.array([k]), w_re, w_im, W_ABSOLUTE)
                    roots, n = self.find_roots(mode, np.array([k]), w_re, w_im, D, st, W_ABSOLUTE, n_iter=n_iter,
                                               tol_percent=tol_percent)
                    ok = roots["flag"].cpu().numpy() == 1
                    w = _distinct(roots["w"].cpu().numpy()[ok])
                    w_all.append(w)
                    k_all.append(np.full(len(w), k))
            out[mode] = (np.concatenate(w_all) if w_all else np.zeros(0, complex), np.concatenate(k_all) if k_all else np.zeros(0))
        return out
