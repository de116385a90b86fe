"""ctypes binding of libeigensolver_amd.so (the C ABI declared in include/eigensolver_amd.h).

The library is GPU-only.  Loading fails loudly when the shared object is missing -- there is no Python or
CPU fallback for any entry point.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libeigensolver_amd.so")


class SlabAnalyticParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("vA_i", "c_i", "vA_e", "c_e", "mach_i", "mach_e", "R1", "cT_i", "cT_e")]


class EsError(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library (once).  Raises EsError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EsError(f"{LIB_PATH} not found: build it with `python -m eigensolver_amd.build` "
                          "(hipcc, --offload-arch=gfx950). There is no CPU fallback.")
        # Device memory and streams come from torch, and the torch wheel carries its own HIP runtime: it has to be
        # the first one in the process (loading this library before torch leaves two runtimes, and the second one
        # finds no device).
        import torch  # noqa: F401
        _lib = _sig(C.CDLL(LIB_PATH))
    return _lib


def load_variant(path):
    """A second build of the library (tests only: lib/libeigensolver_amd_ieee.so, -DES_IEEE_DIVISION) with the same
    signatures; ctypes keeps the symbols of each handle apart."""
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise EsError(f"{path} not found: build it with `python -m eigensolver_amd.build`")
    return _sig(C.CDLL(path))


def check(ctx, status, allow_capacity=False):
    if status == 0 or (allow_capacity and status == 3):
        return status
    lib = load()
    msg = lib.es_status_string(status).decode()
    detail = lib.es_last_error(ctx).decode() if ctx else ""
    raise EsError(f"libeigensolver_amd: {msg}" + (f" ({detail})" if detail else ""))


class Context:
    """Owns an es_context bound to a HIP device and (optionally) a torch stream."""

    def __init__(self, device=0, stream=None, lib=None):
        import torch
        if not torch.cuda.is_available():
            raise EsError("no HIP device visible: eigensolver_amd has no CPU path")
        self.lib = lib if lib is not None else load()
        self.device = int(device)
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        self.torch_stream = stream
        h = C.c_void_p()
        st = self.lib.es_context_create(self.device, C.c_void_p(stream.cuda_stream), C.byref(h))
        if st != 0:
            raise EsError("es_context_create: " + self.lib.es_status_string(st).decode())
        self.handle = h

    def synchronize(self):
        check(self.handle, self.lib.es_context_synchronize(self.handle))

    def grid_timer(self, enable=True):
        """HIP events around every grid-march launch of this context (es_context_grid_timer)."""
        check(self.handle, self.lib.es_context_grid_timer(self.handle, 1 if enable else 0))

    def grid_time(self):
        """(summed ms, launches) of the grid-march kernels since the last call; synchronises the stream."""
        ms, n = C.c_double(0.0), C.c_int(0)
        check(self.handle, self.lib.es_context_grid_time(self.handle, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if getattr(self, "handle", None):
            self.lib.es_context_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ptr(t):
    """Device pointer of a contiguous torch tensor as c_void_p."""
    assert t.is_contiguous()
    return C.c_void_p(t.data_ptr())


# ---- shooting path structs (include/eigensolver_amd.h section 2) ---------------------------------------------
class ShootDesc(C.Structure):
    _fields_ = [("geometry", C.c_int32), ("n_nodes", C.c_int32),
                ("x_boundary", C.c_double), ("x_end", C.c_double),
                ("rho_e", C.c_double), ("vA_e", C.c_double), ("c_e", C.c_double), ("cT_e", C.c_double),
                ("U_e", C.c_double), ("L_factor", C.c_double), ("ic_value", C.c_double), ("ic_slope", C.c_double),
                ("m", C.c_int32), ("m_ext", C.c_int32), ("axis_bc", C.c_int32), ("c1_power", C.c_int32),
                ("bc_const", C.c_double),
                ("slab_mode", C.c_int32), ("accept_norm", C.c_int32),
                ("c_i", C.c_double), ("vA_i", C.c_double), ("rho_i", C.c_double)]


_PROFILE_FIELDS = ("r", "rho", "c2", "vA2", "Bz", "Bphi", "vz", "vphi", "rdC3", "U", "dU", "ddU")


class Profiles(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _PROFILE_FIELDS]


class RootTable(C.Structure):
    _fields_ = [("d_k", C.c_void_p), ("d_w", C.c_void_p), ("d_w_lo", C.c_void_p), ("d_w_hi", C.c_void_p),
                ("d_resid", C.c_void_p), ("d_row", C.c_void_p), ("d_flag", C.c_void_p), ("capacity", C.c_int32)]


class WorkerSpec(C.Structure):
    _fields_ = [("tol_percent", C.c_double), ("min_len", C.c_int32), ("itt_cap", C.c_int32),
                ("reset_loop_ws_each_iter", C.c_int32), ("break_on_accept", C.c_int32),
                ("stale_ext_const", C.c_int32), ("main_double_append", C.c_int32)]


class CylUniformParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("c_i", "vA_i", "rho_i", "U_i", "rho_e", "vA_e", "c_e", "cT_e",
                                           "r_boundary", "r_axis", "L_factor", "ic_value", "ic_slope")] + \
               [("m", C.c_int32), ("m_ext", C.c_int32), ("axis_bc", C.c_int32), ("reserved", C.c_int32)]


class ComplexRootTable(C.Structure):
    _fields_ = [("d_k", C.c_void_p), ("d_w_re", C.c_void_p), ("d_w_im", C.c_void_p), ("d_resid", C.c_void_p),
                ("d_row", C.c_void_p), ("d_flag", C.c_void_p), ("capacity", C.c_int32)]


def _sig(lib):
    """Argument / result types of every entry point of include/eigensolver_amd.h (sections in header order)."""
    vp, i, d = C.c_void_p, C.c_int, C.c_double
    lib.es_abi_version.restype = i
    lib.es_status_string.restype = C.c_char_p
    lib.es_status_string.argtypes = [i]
    lib.es_context_create.argtypes = [i, vp, C.POINTER(vp)]
    lib.es_context_destroy.argtypes = [vp]
    lib.es_last_error.restype = C.c_char_p
    lib.es_last_error.argtypes = [vp]
    lib.es_context_synchronize.argtypes = [vp]
    lib.es_context_grid_timer.argtypes = [vp, i]
    lib.es_context_grid_time.argtypes = [vp, C.POINTER(d), C.POINTER(i)]
    # (1) closed-form slab
    P = C.POINTER(SlabAnalyticParams)
    lib.es_slab_analytic_eval.argtypes = [vp, P, i, vp, i, vp, i, vp]
    lib.es_slab_analytic_scan.argtypes = [vp, P, i, vp, i, vp, i, d, vp, vp, i, C.POINTER(i)]
    lib.es_slab_analytic_filter.argtypes = [vp, P, i, vp, vp, i, d, vp]
    # (2) shooting determinant, brackets, refinement; (3) worker
    lib.es_problem_create.argtypes = [vp, C.POINTER(ShootDesc), C.POINTER(Profiles), C.POINTER(vp)]
    lib.es_problem_destroy.argtypes = [vp, vp]
    lib.es_shoot_eval_grid.argtypes = [vp, vp, vp, i, vp, i, i, vp, vp, vp]
    lib.es_shoot_eval_grid_ex.argtypes = [vp, vp, vp, i, vp, i, i, i, vp, vp, vp]
    lib.es_shoot_eval_points.argtypes = [vp, vp, vp, vp, i, vp, vp, vp]
    lib.es_shoot_grid_shape.argtypes = [vp, vp, i, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    lib.es_shoot_find_roots.argtypes = [vp, vp, vp, i, vp, i, i, vp, vp, i, d, C.POINTER(RootTable), C.POINTER(i)]
    lib.es_shoot_find_roots_mixed.argtypes = [vp, vp, vp, i, vp, i, i, i, d, vp, vp, C.POINTER(RootTable), C.POINTER(i),
                                              C.POINTER(i)]
    lib.es_shoot_screen_grid.argtypes = [vp, vp, vp, i, vp, i, i, vp, vp]
    lib.es_shoot_find_roots_screened.argtypes = [vp, vp, vp, i, vp, i, i, i, d, vp, vp, C.POINTER(RootTable), C.POINTER(i),
                                                 C.POINTER(i)]
    lib.es_shoot_find_roots_async.argtypes = [vp, vp, vp, i, vp, i, i, vp, vp, i, d, C.POINTER(RootTable), vp]
    lib.es_root_table_pack.argtypes = [vp, C.POINTER(RootTable), i, d, vp, i, vp]
    lib.es_root_table_pack_async.argtypes = [vp, C.POINTER(RootTable), vp, d, vp, i, vp]
    lib.es_worker_run.argtypes = [vp, vp, C.POINTER(WorkerSpec), vp, i, vp, i, vp, vp, i, vp]
    # (4) closed-form uniform cylinder; (5) eigenfunctions
    lib.es_cyl_uniform_eval.argtypes = [vp, C.POINTER(CylUniformParams), vp, i, vp, i, i, vp, vp, vp]
    lib.es_shoot_eigenfunction.argtypes = [vp, vp, vp, vp, i, vp, vp, i, vp, vp, vp]
    # (6) complex frequencies
    lib.es_complex_eval_grid.argtypes = [vp, vp, i, vp, i, vp, i, vp, i, i, vp, vp, vp, vp]
    lib.es_complex_eval_points.argtypes = [vp, vp, i, vp, vp, vp, i, vp, vp, vp, vp]
    lib.es_complex_find_roots.argtypes = [vp, vp, i, vp, i, vp, i, vp, i, i, vp, vp, vp, i, d,
                                          C.POINTER(ComplexRootTable), C.POINTER(i)]
    return lib
