"""Per-geometry solver objects that keep the reference's worker call signature

        sausage(wavenumber, sausage_ws, sausage_ks, freq)      kink(wavenumber, kink_ws, kink_ks, freq)

(Slab / Cylinder x non-uniform density / non-uniform flow / rotational flow) and run them on the GPU through the
C ABI (es_worker_run).  One object replaces one reference script: its equilibrium block (module globals), the
checked-in tolerances / recursion caps of its workers, its frequency-band builder and its `__main__` driver
(fan-out over (k, band, mode), fan-in of the root lists, pickle layout [w_sausage, k_sausage, w_kink, k_kink]).

    reference file                                                      class
    Slab/Non uniform density/Photospheric/.../multiprocessor_Inhomogeneous_method.py      SlabNonUniformDensity
    Slab/Non uniform density/Coronal/.../multiprocessor_Inhomogeneous_method_coronal.py   SlabNonUniformDensity(coronal=True)
    Slab/Non uniform flow/Solver/flow_multiprocessor.py                                    SlabUniformFlow
    Slab/Non uniform flow/Solver/flow_multiprocessor_coronal.py                            SlabNonUniformFlow
    Cylinder/Non-uniform density/Coronal/solvers/Density_cylinder.py                      CylinderNonUniformDensity
    Cylinder/Non-uniform density/Photospheric/Solvers/Density_cylinder_photospheric.py    CylinderNonUniformDensity(photospheric=True)
    Cylinder/Non-uniform flow/Coronal/solvers/Cylinder_method_flow_testing.py             CylinderNonUniformFlow
    Cylinder/Rotational flow/Photospheric/Solvers/Twisted_photospheric_*.py               CylinderRotationalFlow
"""
import ctypes as C

import numpy as np

from . import _lib
from . import equilibrium as eqm
from .shooting import ShootProblem


class _WorkerSolver:
    """Common machinery: problems per mode, batched worker runs, the reference-signature entry points."""

    # (tol_percent, min_len, itt_cap, reset_loop_ws_each_iter, break_on_accept, accept_norm[, stale_ext_const
    #  [, main_double_append]]) per mode
    WORKER = {}
    modes = ("sausage", "kink")

    def __init__(self, eq, ctx=None):
        self.eq = eq
        self._ctx = ctx                 # created on first GPU use: speeds() / bands() / eq need no device
        self._problems = {}

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.Context()
        return self._ctx

    def close(self):
        for p in self._problems.values():
            p.close()
        self._problems = {}

    def problem(self, mode):
        if mode not in self._problems:
            # accept_norm is part of the problem (it defines `rel`)
            self._problems[mode] = ShootProblem(self.eq, mode, ctx=self.ctx, accept_norm=int(self.WORKER[mode][5]))
        return self._problems[mode]

    def worker_spec(self, mode, tol=None):
        t = self.WORKER[mode]
        stale = int(t[6]) if len(t) > 6 else 0
        dbl = int(t[7]) if len(t) > 7 else 0
        return _lib.WorkerSpec(float(t[0] if tol is None else tol), int(t[1]), int(t[2]), int(t[3]), int(t[4]), stale, dbl)

    def run_batch(self, mode, wavenumbers, freqs, tol=None, max_roots=None, return_evals=False):
        """Many worker calls at once: wavenumbers[t], freqs[t, :] -> list of root lists (one per task)."""
        import torch
        prob = self.problem(mode)
        dev = f"cuda:{self.ctx.device}"
        k = torch.as_tensor(np.ascontiguousarray(wavenumbers, dtype=np.float64).reshape(-1), device=dev)
        f = torch.as_tensor(np.ascontiguousarray(freqs, dtype=np.float64), device=dev)
        nt = k.numel()
        nf = f.shape[-1] if f.ndim > 1 else (f.numel() // max(nt, 1))
        f = f.reshape(nt, nf).contiguous()
        spec = self.worker_spec(mode, tol)
        cap = int(max_roots) if max_roots else max(8, 2 * nf)
        while True:
            roots = torch.empty((nt, cap), dtype=torch.float64, device=dev)
            n = torch.zeros(nt, dtype=torch.int32, device=dev)
            nev = torch.zeros(nt, dtype=torch.int32, device=dev)
            rc = self.ctx.lib.es_worker_run(self.ctx.handle, prob.handle, C.byref(spec), _lib.ptr(k), nt,
                                            _lib.ptr(f), nf, _lib.ptr(roots), _lib.ptr(n), cap, _lib.ptr(nev))
            _lib.check(self.ctx.handle, rc, allow_capacity=True)
            n_h = n.cpu().numpy()
            if rc == 3 and max_roots is None:
                cap = int(n_h.max())
                continue
            break
        r_h = roots.cpu().numpy()
        out = [r_h[t, :min(n_h[t], cap)].tolist() for t in range(nt)]
        return (out, nev.cpu().numpy()) if return_evals else out

    # ---- the reference's worker signature --------------------------------------------------------------------
    def _worker(self, mode, wavenumber, ws_sink, ks_sink, freq):
        roots = self.run_batch(mode, [float(wavenumber)], np.asarray(freq, dtype=np.float64)[None, :])[0]
        ks_sink.put([float(wavenumber)] * len(roots))      # the reference puts the k list first (CD-C:823-824)
        ws_sink.put(list(roots))

    def sausage(self, wavenumber, sausage_ws, sausage_ks, freq):
        self._worker("sausage", wavenumber, sausage_ws, sausage_ks, freq)

    def kink(self, wavenumber, kink_ws, kink_ks, freq):
        self._worker("kink", wavenumber, kink_ws, kink_ks, freq)

    # ---- the reference's driver block ---------------------------------------------------------------------------
    def speeds(self):
        raise NotImplementedError

    def bands(self, k, n_per_band):
        """a9: linspace(speeds[i]*k, speeds[i+1]*k, n) for consecutive sorted characteristic speeds (CD-C:1142-1145)."""
        sp = sorted(self.speeds())
        return [np.linspace(sp[i] * k, sp[i + 1] * k, n_per_band) for i in range(len(sp) - 1)]

    def solve(self, wavenumbers, n_per_band, modes=None):
        """The `__main__` block (e.g. CD-C:1129-1183): every (k, band, mode) task, results flattened into the
        pickle layout [w_sausage, k_sausage, w_kink, k_kink] (only the modes the script defines)."""
        out = {}
        tasks_k, tasks_f = [], []
        for k in wavenumbers:
            for b in self.bands(float(k), n_per_band):
                tasks_k.append(float(k))
                tasks_f.append(b)
        for mode in (modes or self.modes):
            roots = self.run_batch(mode, tasks_k, np.stack(tasks_f)) if tasks_k else []
            w = [x for r in roots for x in r]
            kk = [tasks_k[t] for t, r in enumerate(roots) for _ in r]
            out[mode] = (np.array(w), np.array(kk))
        return out


def solve_distributed(solver, wavenumbers, n_per_band=None, modes=None, strided=True):
    """The reference's driver block k-tiled over the ranks of the default process group (one process per GPU): every
    rank solves its strided tile of the wavenumbers (no data-path collective), then one all-gather of the root lists
    over RCCL.  Returns the same dict as `solver.solve` on every rank (rank-major order)."""
    import torch.distributed as dist
    from .distributed import gather_mode_results, tile_rows
    ks = np.asarray(wavenumbers, dtype=np.float64)
    if dist.is_initialized() and dist.get_world_size() > 1:
        mine = ks[tile_rows(len(ks), dist.get_rank(), dist.get_world_size(), strided)]
    else:
        mine = ks
    local = solver.solve(mine, n_per_band, modes) if n_per_band is not None else solver.solve(mine, modes=modes)
    return gather_mode_results(local)


class CylinderNonUniformDensity(_WorkerSolver):
    """Density_cylinder.py (coronal, CD-C) / Density_cylinder_photospheric.py (CD-P)."""

    def __init__(self, width=0.95, photospheric=False, ctx=None, **kw):
        if photospheric:
            eq = eqm.CylinderDensity(width=width, c_e=1.5, vA_e=0.5, r_sign=1.0, n_nodes=1000, ic=(1e-8, 1e-8), **kw)
            self.WORKER = {"kink": (1.0, 2, 300, 0, 0, 0), "sausage": (1.0, 2, 300, 0, 0, 0)}     # CD-P:525, :561
        else:
            eq = eqm.CylinderDensity(width=width, **kw)
            self.WORKER = {"kink": (1.0, 2, 150, 0, 0, 0), "sausage": (1.0, 2, 150, 0, 0, 0)}     # CD-C:522, :558
        self.photospheric = photospheric
        super().__init__(eq, ctx)

    def speeds(self):
        e = self.eq
        if self.photospheric:
            return [e.c_i0, e.cT_i0, (e.c_i0 + e.cT_i0) / 2.0, 0.675, 0.8, 0.7]                    # CD-P:227
        # CD-C:225, with the reference's missing comma `cT_e -c_e`
        return [e.c_i0, e.c_e, e.vA_i0, e.vA_e, e.cT_i0, e.cT_e - e.c_e, -e.c_i0, -e.vA_i0, -e.vA_e, -e.cT_i0, -e.cT_e]


class CylinderNonUniformFlow(_WorkerSolver):
    """Cylinder_method_flow_testing.py (CF)."""
    WORKER = {"kink": (6.0, 2, 250, 0, 0, 0), "sausage": (6.0, 2, 250, 0, 0, 0)}                  # CF:530, :566

    def __init__(self, U_i0=0.0, width=1e5, ctx=None, **kw):
        super().__init__(eqm.CylinderFlow(U_i0=U_i0, width=width, **kw), ctx)

    def speeds(self):
        e = self.eq
        return [e.c_i0, e.vA_i0, e.vA_e, e.cT_i0, e.c_kink]                                       # CF:231


class CylinderRotationalFlow(_WorkerSolver):
    """Twisted_photospheric_nonlinear_flow_kink_{fast,slow}.py / Twisted_photospheric_flow_sausage{,_slow}.py."""

    def __init__(self, v_twist=0.25, power=0.8, variant="kink_fast", ctx=None, **kw):
        spec = {"kink_fast": ("kink", (2.5, 2, 500, 0, 1, 0), 1e-3),        # CR-KF:435, :464, :722
                "kink_slow": ("kink", (3.0, 2, 500, 0, 1, 1), 1e-3),        # CR-KS:441, :468, :722
                # CR-SF:419, :475, r_ax 0.01 (:157), stale xi_e_const (:558), all_ws appended twice (:684, :726)
                "sausage": ("sausage", (1.5, 2, 250, 0, 0, 0, 1, 1), 1e-2),
                "sausage_slow": ("sausage", (4.5, 2, 250, 0, 0, 0, 1, 1), 1e-2)}[variant]           # CR-SS:423, :479
        self.modes = (spec[0],)
        self.WORKER = {spec[0]: spec[1]}
        self.variant = variant
        super().__init__(eqm.CylinderRotation(v_twist=v_twist, power=power, r_axis=spec[2], **kw), ctx)

    def speeds(self):
        e = self.eq
        return {"kink_fast": [e.c_kink, 1.35, 1.4], "kink_slow": [e.c_i0, e.c_kink, 1.1, 1.2],     # CR-KF:227, CR-KS:229
                "sausage": [e.c_e, e.c_kink, 1.4],                                                 # CR-SF:224
                "sausage_slow": [1.0, 0.98, 0.96, 0.94, 0.92, 0.9, 0.88]}[self.variant]            # CR-SS:232


class SlabNonUniformDensity(_WorkerSolver):
    """multiprocessor_Inhomogeneous_method.py (photospheric, SD-P) / ..._coronal.py (SD-C)."""

    def __init__(self, width=1e5, coronal=False, n_nodes=2001, ctx=None, **kw):
        if coronal:
            eq = eqm.SlabDensity(width=width, vA_i0=1.2, vA_e=3.0, c_e=0.4, L_factor=3.0, n_nodes=n_nodes, **kw)  # SD-C:72-75
            t = 1.0                                                                                # SD-C:378
        else:
            eq = eqm.SlabDensity(width=width, n_nodes=n_nodes, **kw)
            t = 3.0                                                                                # SD-P:275
        self.WORKER = {"sausage": (t, 1, 100, 1, 0, 0), "kink": (t, 1, 100, 0, 0, 0)}
        self.coronal = coronal
        super().__init__(eq, ctx)

    def speeds(self):
        e = self.eq
        if self.coronal:
            # SD-C:202: speeds = [1.0, cT_bound, 1.2, 1.3, 0.9], cT_bound = tube speed of the profile at x = -1 (SD-C:182-185)
            pr = eqm.SlabDensity(**{**e.__dict__, "n_nodes": 2}).profiles()        # points -1, 0, +1
            c2b, vA2b = float(pr["c2"][0]), float(pr["vA2"][0])
            cT_bound = float(np.sqrt(c2b * vA2b / (c2b + vA2b)))
            return [1.0, cT_bound, 1.2, 1.3, 0.9]
        return [e.c_i0, e.cT_i0]                                                                  # SD-P:171 (zoom speeds, uniform case)


class SlabNonUniformFlow(_WorkerSolver):
    """flow_multiprocessor_coronal.py (SF-G): Gaussian flow."""
    WORKER = {"sausage": (1.0, 1, 100, 1, 0, 0), "kink": (1.0, 1, 100, 0, 0, 0)}                   # SF-G:250, :297

    def __init__(self, U_i0=0.9, width=1e5, ctx=None, **kw):
        super().__init__(eqm.SlabFlow(U_i0=U_i0, width=width, **kw), ctx)

    def speeds(self):
        e = self.eq
        return [-e.vA_e, 0.0, e.c_i0, e.c_e, e.vA_i0, e.vA_e, e.cT_i0, e.cT_e]                     # SF-G:180


class SlabUniformFlow(_WorkerSolver):
    """flow_multiprocessor.py (SF-U): uniform steady flow, shooting workers (the analytic part is SlabSteadyFlow)."""
    WORKER = {"sausage": (1e-6, 1, 200, 1, 0, 0), "kink": (1e-6, 1, 200, 0, 0, 0)}                 # SF-U:419, :459

    def __init__(self, ctx=None, **kw):
        p = dict(c_i0=2.0 / 3.0, vA_i0=1.0, c_e=0.75, vA_e=0.0, U_i0=0.0, U_e=-0.15, width=float("inf"),
                 L_factor=7.0, n_nodes=500)
        p.update(kw)
        super().__init__(eqm.SlabFlow(**p), ctx)

    def bands(self, k, n_per_band=None):
        e = self.eq
        return [np.logspace(0.001, 0.55, 80) - 1, np.linspace(e.cT_i0 * k, (e.c_e + e.U_e) * k, 100)]   # SF-U:813, :838

    def solve(self, wavenumbers, n_per_band=None, modes=None):
        out = {}
        for mode in (modes or self.modes):
            w_all, k_all = [], []
            for bi in range(2):
                ks = [float(k) for k in wavenumbers]
                fr = np.stack([self.bands(k)[bi] for k in ks])
                for t, r in enumerate(self.run_batch(mode, ks, fr)):
                    w_all += r
                    k_all += [ks[t]] * len(r)
            out[mode] = (np.array(w_all), np.array(k_all))
        return out
