"""ORACLE (test infrastructure only): ctypes wrapper of oracle/c/libshoot_port.so, the plain-C CPU port of the
fixed-grid shooting evaluation (see oracle/c/shoot_port.c for the reference lines it restates)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "c", "libshoot_port.so")


class ShootDesc(C.Structure):
    """Mirror of es_shoot_desc (include/eigensolver_amd.h)."""
    _fields_ = [("geometry", C.c_int32), ("n_nodes", C.c_int32),
                ("x_boundary", C.c_double), ("x_end", C.c_double),
                ("rho_e", C.c_double), ("vA_e", C.c_double), ("c_e", C.c_double), ("cT_e", C.c_double),
                ("U_e", C.c_double), ("L_factor", C.c_double), ("ic_value", C.c_double), ("ic_slope", C.c_double),
                ("m", C.c_int32), ("m_ext", C.c_int32), ("axis_bc", C.c_int32), ("c1_power", C.c_int32),
                ("bc_const", C.c_double),
                ("slab_mode", C.c_int32), ("accept_norm", C.c_int32),
                ("c_i", C.c_double), ("vA_i", C.c_double), ("rho_i", C.c_double)]


PROFILE_FIELDS = ("r", "rho", "c2", "vA2", "Bz", "Bphi", "vz", "vphi", "rdC3", "U", "dU", "ddU")


class Profiles(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in PROFILE_FIELDS]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        L = C.CDLL(_LIB)
        vp, i, d, l = C.c_void_p, C.c_int, C.c_double, C.c_long
        L.port_create.restype = vp
        L.port_create.argtypes = [C.POINTER(ShootDesc), C.POINTER(Profiles)]
        L.port_destroy.argtypes = [vp]
        L.port_eval.argtypes = [vp, d, d, C.POINTER(d), C.POINTER(d)]
        L.port_eval2.argtypes = [vp, d, d, d, C.POINTER(d), C.POINTER(d)]
        L.port_eval_ext.argtypes = [vp, d, d, d, d, d, C.POINTER(d), C.POINTER(d), C.POINTER(d), C.POINTER(d)]
        L.port_eval_parts.argtypes = [vp, d, d, d, C.POINTER(d), C.POINTER(d), C.POINTER(d), C.POINTER(d)]
        L.port_eval_points.argtypes = [vp, vp, vp, l, vp, vp, vp, i]
        L.port_eval_grid.argtypes = [vp, vp, i, vp, i, i, vp, vp, vp, i]
        L.port_find_roots.restype = l
        L.port_find_roots.argtypes = [vp, vp, i, vp, i, i, vp, vp, i, d, vp, vp, vp, vp, vp, vp, vp, l, i]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class PortProblem:
    """desc_fields: dict with the es_shoot_desc fields; profiles: dict name -> array of length 2N-1."""

    def __init__(self, desc_fields, profiles):
        self.desc = ShootDesc()
        for k, v in desc_fields.items():
            setattr(self.desc, k, v)
        self._prof = {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in profiles.items()}
        p = Profiles()
        for n in PROFILE_FIELDS:
            a = self._prof.get(n)
            setattr(p, n, a.ctypes.data if a is not None else None)
        self.h = lib().port_create(C.byref(self.desc), C.byref(p))

    def __del__(self):
        try:
            lib().port_destroy(self.h)
        except Exception:
            pass

    def eval_one(self, k, w, w_cst=None, ext=None):
        """One evaluation -> (status, d, rel, outer, inner); ext = (value, slope): exterior end state override."""
        D, rel, outer, inner = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        wc = w if w_cst is None else w_cst
        if ext is None:
            st = lib().port_eval_parts(self.h, k, w, wc, C.byref(D), C.byref(rel), C.byref(outer), C.byref(inner))
        else:
            st = lib().port_eval_ext(self.h, k, w, wc, float(ext[0]), float(ext[1]), C.byref(D), C.byref(rel),
                                     C.byref(outer), C.byref(inner))
        return st, D.value, rel.value, outer.value, inner.value

    def eval_points(self, k, w, nthreads=0):
        k = np.ascontiguousarray(k, dtype=np.float64).ravel()
        w = np.ascontiguousarray(w, dtype=np.float64).ravel()
        n = len(k)
        D, rel, st = np.empty(n), np.empty(n), np.empty(n, dtype=np.uint8)
        lib().port_eval_points(self.h, _p(k), _p(w), n, _p(D), _p(rel), _p(st), nthreads)
        return D, rel, st

    def eval_grid(self, k, w, w_mode=1, nthreads=0):
        k = np.ascontiguousarray(k, dtype=np.float64).ravel()
        w = np.ascontiguousarray(w, dtype=np.float64)
        nk = len(k)
        nw = w.shape[-1] if w_mode == 2 else w.size
        D, rel, st = np.empty((nk, nw)), np.empty((nk, nw)), np.empty((nk, nw), dtype=np.uint8)
        lib().port_eval_grid(self.h, _p(k), nk, _p(w), nw, w_mode, _p(D), _p(rel), _p(st), nthreads)
        return D, rel, st

    def find_roots(self, k, w, D, st, w_mode=1, n_bisect=40, tol=1e-3, capacity=1 << 20, nthreads=0):
        k = np.ascontiguousarray(k, dtype=np.float64).ravel()
        w = np.ascontiguousarray(w, dtype=np.float64)
        nk = len(k)
        nw = w.shape[-1] if w_mode == 2 else w.size
        D = np.ascontiguousarray(D, dtype=np.float64)
        st = np.ascontiguousarray(st, dtype=np.uint8)
        o = {n: np.empty(capacity) for n in ("k", "w", "w_lo", "w_hi", "resid")}
        o["row"] = np.empty(capacity, dtype=np.int32)
        o["flag"] = np.empty(capacity, dtype=np.uint8)
        cnt = lib().port_find_roots(self.h, _p(k), nk, _p(w), nw, w_mode, _p(D), _p(st), n_bisect, tol,
                                    _p(o["k"]), _p(o["w"]), _p(o["w_lo"]), _p(o["w_hi"]), _p(o["resid"]),
                                    _p(o["row"]), _p(o["flag"]), capacity, nthreads)
        m = min(cnt, capacity)
        return {n: v[:m] for n, v in o.items()}, cnt
