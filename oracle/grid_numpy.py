"""ORACLE (test infrastructure only -- never imported by the product path).

Vectorised NumPy restatement of the (k, omega)-grid evaluation of the boundary determinant for the UNTWISTED cylinder
(reference: Cylinder/Non-uniform flow/Coronal/solvers/Cylinder_method_flow_testing.py:694-804 per point,
Cylinder/Non-uniform density/Coronal/solvers/Density_cylinder.py:694-804): for one k-row all omega are marched
together as NumPy arrays -- the "fp64 NumPy path on the host cores" SURVEY.md section 8d(ii) asks to be timed beside
the GPU (bench.py, second cpu_baseline entry, one process per k-tile).

Same discretisation as oracle/c/shoot_port.c and the HIP kernel (one classical RK4 step per interval of the reference's
interior grid, adjoint single-row march, shooting condition by superposition, closed-form exterior), written
independently with scipy's Bessel functions for the exterior; NumPy has no fused multiply-add, so values agree with
the C port to rounding (tests/test_oracle_port.py::test_numpy_grid_matches_port), not bit for bit.
"""
import numpy as np
from scipy import special

ST_OK, ST_LEAKY, ST_NONFINITE, ST_CONTINUUM = 0, 1, 2, 3


class CylinderGrid:
    """desc: dict with the es_shoot_desc fields (include/eigensolver_amd.h); prof: the 2N-1 profile samples."""

    def __init__(self, desc, prof):
        assert desc["geometry"] == 0, "untwisted cylinder only"
        self.d = dict(desc)
        r, rho, c2, Bz = (np.asarray(prof[n], dtype=np.float64) for n in ("r", "rho", "c2", "Bz"))
        vz = np.asarray(prof.get("vz", np.zeros_like(r)), dtype=np.float64)
        bA = Bz / np.sqrt(rho)                       # (k B_z)/sqrt(rho) per unit k, CF:581
        S = c2 + bA * bA
        self.vz, self.bA2, self.q = vz, bA * bA, c2 / S
        self.a1 = rho / r                            # a12 = rho (Om^2 - wA^2) / r
        self.B = r / (rho * S)
        self.e1, self.e2 = 1.0 / (r * rho), r / rho
        self.N = int(desc["n_nodes"])
        self.h = (desc["x_end"] - desc["x_boundary"]) / (self.N - 1)
        # continuum bands in phase speed: node j inside iff |W - vz_j| < a_j
        self.bands = []
        for a in (np.abs(bA), np.abs(bA) * np.sqrt(self.q)):
            lo, hi = vz - a, vz + a
            self.bands.append((lo.min(), lo.max(), hi.min(), hi.max()))

    def _coef(self, j, k, w):
        """a12, a21 of node j for all omega (arrays)."""
        Om = w - k * self.vz[j]
        Om2 = Om * Om
        wA2 = k * k * self.bA2[j]
        wc2 = wA2 * self.q[j]
        t1, t2 = Om2 - wA2, Om2 - wc2
        g = self.d["m"] ** 2 * self.e1[j] + k * k * self.e2[j]
        with np.errstate(all="ignore"):
            a12 = self.a1[j] * t1
            a21 = (g * t2 - self.B[j] * (Om2 * Om2)) / (t1 * t2)        # -r C2 / D
        return a12, a21

    def _exterior(self, k, w):
        d = self.d
        k2, w2 = k * k, w * w
        vAe2, ce2, cTe2 = d["vA_e"] ** 2, d["c_e"] ** 2, d["cT_e"] ** 2
        with np.errstate(all="ignore"):
            m_e = (k2 * vAe2 - w2) * (k2 * ce2 - w2) / ((vAe2 + ce2) * (k2 * cTe2 - w2))
            cst = -1.0 / (d["rho_e"] * (k2 * vAe2 - w2))
            st = np.where(m_e < 0, ST_LEAKY, np.where((m_e > 0) & np.isfinite(m_e) & np.isfinite(cst), ST_OK, ST_NONFINITE))
            mu = np.sqrt(np.where(st == ST_OK, m_e, 1.0))
            sgn = -1.0 if d["x_boundary"] < 0 else 1.0
            n = d["m_ext"]
            xR, xb = mu * (d["L_factor"] * 2.0 * np.pi / k), mu
            Kb, Kb1, KR, KR1 = special.kve(n, xb), special.kve(n + 1, xb), special.kve(n, xR), special.kve(n + 1, xR)
            Ib, Ib1, IR, IR1 = special.ive(n, xb), special.ive(n + 1, xb), special.ive(n, xR), special.ive(n + 1, xR)
            dKb, dKR = -Kb1 + (n / xb) * Kb, -KR1 + (n / xR) * KR
            dIb, dIR = Ib1 + (n / xb) * Ib, IR1 + (n / xR) * IR
            g = d["ic_slope"] / (sgn * mu)
            a_s, b_s = -(d["ic_value"] * dKR - g * KR), -(g * IR - d["ic_value"] * dIR)
            E2 = np.exp(-2.0 * (xR - xb))
            P = b_s * Kb + E2 * a_s * Ib
            dP = sgn * mu * (b_s * dKb + E2 * a_s * dIb)
            nrm = np.abs(P)
            yb, dyb = P / nrm, dP / nrm
        st = np.where((st == ST_OK) & ~(np.isfinite(yb) & np.isfinite(dyb)), ST_NONFINITE, st)
        return st, cst, yb, dyb

    def eval_row(self, k, w):
        """D, rel, status for one wavenumber and an array of omega."""
        d = self.d
        w = np.asarray(w, dtype=np.float64)
        st, cst, yb, dyb = self._exterior(k, w)
        h, N = self.h, self.N
        # adjoint march of the row (p, q) from the axis end back to the boundary (see es_shoot_device.hpp)
        b1, a1 = self._coef(2 * (N - 1), k, w)
        if d["axis_bc"] == 1:
            p, q = np.zeros_like(w), b1.copy()       # P'(r_ax) = a12 Xi = 0
        else:
            p, q = np.ones_like(w), np.zeros_like(w)
        b0, a0 = b1, a1
        with np.errstate(all="ignore"):
            for j in range(N - 2, -1, -1):
                bm, am = self._coef(2 * j + 1, k, w)
                b1, a1 = self._coef(2 * j, k, w)
                k1p, k1q = a0 * q, b0 * p
                tp, tq = p + 0.5 * h * k1p, q + 0.5 * h * k1q
                k2p, k2q = am * tq, bm * tp
                tp, tq = p + 0.5 * h * k2p, q + 0.5 * h * k2q
                k3p, k3q = am * tq, bm * tp
                tp, tq = p + h * k3p, q + h * k3q
                k4p, k4q = a1 * tq, b1 * tp
                p = p + (h / 6.0) * (k1p + k4p) + (h / 3.0) * (k2p + k3p)
                q = q + (h / 6.0) * (k1q + k4q) + (h / 3.0) * (k2q + k3q)
                b0, a0 = b1, a1
            xi_e = cst * dyb
            if d["axis_bc"] == 0:
                Xb = (d["bc_const"] * xi_e - p * yb) / q
            else:
                Xb = -(p * yb) / q
            xi_i = Xb / d["x_boundary"]
            D = xi_e - xi_i
            rel = np.abs(D) * 100.0 / np.maximum(np.abs(xi_e), np.abs(xi_i))
        W = w / k
        crossed = np.zeros(w.shape, dtype=bool)
        for lo_min, lo_max, hi_min, hi_max in self.bands:
            crossed |= ((W > lo_min) & (W < hi_max)) & ~((W > lo_max) & (W < hi_min))
        out_st = st.astype(np.uint8)
        ok = st == ST_OK
        D = np.where(ok, D, np.nan)
        rel = np.where(ok, rel, np.nan)
        out_st[ok & ~np.isfinite(D)] = ST_NONFINITE
        out_st[ok & np.isfinite(D) & crossed] = ST_CONTINUUM
        return D, rel, out_st

    def eval_grid(self, k, W, phase_speed=True):
        k = np.asarray(k, dtype=np.float64)
        W = np.asarray(W, dtype=np.float64)
        D = np.empty((len(k), len(W)))
        rel = np.empty_like(D)
        st = np.empty(D.shape, dtype=np.uint8)
        for i, kk in enumerate(k):
            D[i], rel[i], st[i] = self.eval_row(kk, kk * W if phase_speed else W)
        return D, rel, st


def _tile(args):
    desc, prof, k, W = args
    import time
    t = time.perf_counter()
    CylinderGrid(desc, prof).eval_grid(k, W)
    return time.perf_counter() - t


def timed_parallel(desc, prof, k, W, procs):
    """Wall time of eval_grid over the rows `k`, one process per k-tile (bench.py's NumPy baseline)."""
    import multiprocessing as mp
    import time
    tiles = [(desc, prof, k[i::procs], W) for i in range(procs) if len(k[i::procs])]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(len(tiles)) as pool:
        pool.map(_tile, tiles)
    return time.perf_counter() - t0
