"""ORACLE (test infrastructure only -- never imported by the product path).

Complex-frequency (unstable / Kelvin-Helmholtz modes) variant of the flow-slab mismatch, SURVEY.md section 8f row 3:
    Slab/Non uniform flow/COMPLEX ANALYSIS/flow_multiprocessor_complex_coronal.py (SF-X)
      :122-127, :150-185   constants, Gaussian flow U(x) and its derivatives
      :369-371             m_e, p_e_const                    (same expressions as SF-G with complex omega)
      :375-404             m0, D, coeff, P_Ti, add_P_Ti      (D is the corrected form of :382, NOT SF-G:421)
      :419-421, :440-446   exterior / interior ODEs
      :424-426, :455-456   interior boundary value, total pressures
      :1127                driver grid: Re(omega) over a phase-speed band x Im(omega) in [-0.25, 0.25]

PARITY UNPINNED against the reference's outputs: SF-X does not run under the installed NumPy (`np.linspace(.., 500.)`),
mixes real and imaginary parts inconsistently (`omega[k] + 1j*omega[n]` on an already complex array, real-part-only
shooting in the main loop, complex shooting in locate_*; SF-X:422-426, :455-456, :553-577) and ships no stored
output.  What IS pinned:
  * the coefficient functions above, against the values the reference's own lambdified functions give
    (tests/golden/complex_coefficients.json, captured by tools/gen_golden_complex.py);
  * the whole evaluation in the uniform-flow limit, against the closed-form complex dispersion function
    (`closed_form_uniform`, the standard slab relation the reference's SF-U:117-127 states for real W);
  * the machinery at Im(omega) = 0 with variant "sfg" against the real flow-slab oracle (oracle/slab.py), which is
    pinned to traces of the reference.

Definition used here (the consistent reading of SF-X): everything complex,
    V_e'' = m_e V_e,  far-field values ic, decaying branch;  y = V_e'(-1) / V_e(-1)
    outer = p_e_const * y                                         (total pressure outside, per unit V_e(-1))
    Vx(-1) = (omega - k U(-1)) / (omega - k U_e)                  (continuity of the displacement)
    Vx'' = -D Vx' - coeff Vx  on [-1, 1],  Vx(+1) = -/+ Vx(-1)  (sausage / kink), slope by superposition
    inner = P_Ti(-1) * (Vx'(-1) - add(-1) Vx(-1))
    D_c = outer - inner,  rel = 100 |D_c| / max(|outer|, |inner|)
Two evaluators of the interior: `eval_rk4` (the algorithm the HIP kernel runs: adjoint RK4 on the reference's ix
grid, one step per interval, mid-point coefficient sets) and `eval_truth` (DOP853 at rtol 1e-12).
"""
import numpy as np
from scipy.integrate import solve_ivp

ST_OK, ST_LEAKY, ST_NONFINITE = 0, 1, 2


class ComplexFlowSlab:
    def __init__(self, c_i=1.3, vA_i=1.0, c_e=None, vA_e=0.0, rho_i=9.0, rho_e=5.0, U_i0=1.4, U_e=0.0, width=1e5,
                 x0=0.0, mode="kink", L_factor=3.0, ic=(1e-8, 1e-15), n_nodes=500, variant="sfx"):
        gamma = 5.0 / 3.0
        self.c_i, self.vA_i, self.vA_e, self.rho_i, self.rho_e = c_i, vA_i, vA_e, rho_i, rho_e
        self.c_e = np.sqrt((rho_i / rho_e) * c_i ** 2 + gamma * 0.5 * vA_i ** 2) if c_e is None else c_e   # SF-X:111
        self.U_i0, self.U_e, self.width, self.x0 = U_i0, U_e, width, x0
        self.mode, self.L_factor, self.ic, self.n_nodes, self.variant = mode, float(L_factor), ic, int(n_nodes), variant
        self.cT_i2 = c_i ** 2 * vA_i ** 2 / (c_i ** 2 + vA_i ** 2)
        self.cT_e2 = self.c_e ** 2 * vA_e ** 2 / (self.c_e ** 2 + vA_e ** 2)

    # ---- profile (SF-X:166-167 and sympy derivatives :190-197) -------------------------------------------------
    def gauss(self, x):
        return np.exp(-(np.asarray(x, dtype=float) - self.x0) ** 2 / self.width ** 2)

    def U(self, x):
        return self.U_e + (self.U_i0 - self.U_e) * self.gauss(x)

    def dU(self, x):
        x = np.asarray(x, dtype=float)
        return (self.U_i0 - self.U_e) * self.gauss(x) * (-2.0 * (x - self.x0) / self.width ** 2)

    def ddU(self, x):
        x = np.asarray(x, dtype=float)
        return (self.U_i0 - self.U_e) * self.gauss(x) * (4.0 * (x - self.x0) ** 2 / self.width ** 4 - 2.0 / self.width ** 2)

    # ---- coefficient functions ---------------------------------------------------------------------------------
    def exterior_constants(self, k, w):
        Oe = w - k * self.U_e
        S = self.vA_e ** 2 + self.c_e ** 2
        m_e = ((k ** 2 * self.vA_e ** 2 - Oe ** 2) * (k ** 2 * self.c_e ** 2 - Oe ** 2)) / (S * (k ** 2 * self.cT_e2 - Oe ** 2))   # SF-X:369
        p_e = self.rho_e * S * (k ** 2 * self.cT_e2 - Oe ** 2) / (Oe * (k ** 2 * self.c_e ** 2 - Oe ** 2))                        # SF-X:371
        return m_e, p_e

    def interior_coefficients(self, x, k, w):
        """(m0, D, coeff, P_Ti, add) at x for complex omega; x scalar or array, w scalar or array (broadcast)."""
        Om = w - k * self.U(x)
        Om2 = Om * Om
        S = self.c_i ** 2 + self.vA_i ** 2
        kc2, kvA2, kcT2 = k ** 2 * self.c_i ** 2, k ** 2 * self.vA_i ** 2, k ** 2 * self.cT_i2
        m0 = (kc2 - Om2) * (kvA2 - Om2) / (S * (kcT2 - Om2))                                     # SF-X:375
        dU, ddU = self.dU(x), self.ddU(x)
        if self.variant == "sfx":
            D = 2.0 * k * dU * (Om2 / (Om2 - kc2) - kcT2 / (Om2 - kcT2)) / Om                    # SF-X:382
        else:                                                                                    # SF-G:421 as written
            D = 2.0 * k * dU * ((Om2 - kcT2) + k ** 4 * self.cT_i2 * self.c_i ** 2 / (S * (Om2 - kcT2))) / (Om * (Om2 - kc2))
        coeff = k * ddU / Om + k * dU * D / Om - m0                                              # SF-X:389
        P_Ti = self.rho_i * S * (kcT2 - Om2) / (Om * (kc2 - Om2))                                # SF-X:395
        add = -(k * dU) / Om if self.variant == "sfx" else 0.0 * Om                              # SF-X:401
        return m0, D, coeff, P_Ti, add

    # ---- exterior in closed form ---------------------------------------------------------------------------------
    def exterior(self, k, w):
        """(status, outer = p_e V'/V at the boundary, m_e).  Points with Re(m_e) < 0 are skipped by the reference."""
        w = np.asarray(w, dtype=complex)
        m_e, p_e = self.exterior_constants(k, w)
        with np.errstate(all="ignore"):
            mu = np.sqrt(m_e)                               # principal branch, Re(mu) >= 0: decays towards -infinity
            R = self.L_factor * 2.0 * np.pi / k
            E2 = np.exp(-2.0 * mu * (R - 1.0))
            gp, gm = self.ic[0] + self.ic[1] / mu, self.ic[0] - self.ic[1] / mu
            y = mu * (gp - E2 * gm) / (gp + E2 * gm)
            outer = p_e * y
        st = np.where(m_e.real < 0.0, ST_LEAKY, ST_OK)
        st = np.where((st == ST_OK) & ~np.isfinite(outer), ST_NONFINITE, st)
        return st, outer, m_e

    def _finish(self, k, w, st, outer, r1, r2):
        sigma = -1.0 if self.mode == "sausage" else 1.0
        xb = -1.0
        Vb = (w - k * self.U(xb)) / (w - k * self.U_e)                                           # SF-X:422
        sv = (sigma - r1) * Vb / r2
        _, _, _, PTi, add = self.interior_coefficients(xb, k, w)
        inner = PTi * (sv - add * Vb)                                                            # SF-X:455
        d = outer - inner
        with np.errstate(all="ignore"):
            rel = np.abs(d) * 100.0 / np.maximum(np.abs(outer), np.abs(inner))
        bad = (st == ST_OK) & ~np.isfinite(d)
        st = np.where(bad, ST_NONFINITE, st)
        d = np.where(st == ST_OK, d, np.nan + 0j)
        rel = np.where(st == ST_OK, rel, np.nan)
        return d, rel, st.astype(np.uint8)

    # ---- the GPU algorithm in NumPy: adjoint RK4 of the row (T11, T12), vectorised over the points ----------------
    def eval_rk4(self, k, w):
        w = np.atleast_1d(np.asarray(w, dtype=complex))
        st, outer, _ = self.exterior(k, w)
        N = self.n_nodes
        x = np.linspace(-1.0, 1.0, 2 * N - 1)             # nodes and mid-points
        h = 2.0 / (N - 1)

        def A(i):                                         # u' = v, v' = -D v - coeff u
            _, D, cf, _, _ = self.interior_coefficients(x[i], k, w)
            return -cf, -D                                # a21, a22 (a11 = 0, a12 = 1)

        p = np.ones_like(w)                               # functional Vx(+1) = (1, 0) . (u, v)
        q = np.zeros_like(w)

        def rhs(a, pp, qq):                               # A^T z
            a21, a22 = a
            return a21 * qq, pp + a22 * qq

        with np.errstate(all="ignore"):
            B0 = A(2 * (N - 1))
            for j in range(N - 2, -1, -1):
                Bm, B1 = A(2 * j + 1), A(2 * j)
                k1p, k1q = rhs(B0, p, q)
                k2p, k2q = rhs(Bm, p + 0.5 * h * k1p, q + 0.5 * h * k1q)
                k3p, k3q = rhs(Bm, p + 0.5 * h * k2p, q + 0.5 * h * k2q)
                k4p, k4q = rhs(B1, p + h * k3p, q + h * k3q)
                p = p + h / 6.0 * (k1p + k4p) + h / 3.0 * (k2p + k3p)
                q = q + h / 6.0 * (k1q + k4q) + h / 3.0 * (k2q + k3q)
                B0 = B1
        return self._finish(k, w, st, outer, p, q)

    # ---- truth: DOP853 on both columns of the transfer matrix ------------------------------------------------------
    def eval_truth(self, k, w, rtol=1e-12):
        w = np.atleast_1d(np.asarray(w, dtype=complex))
        st, outer, _ = self.exterior(k, w)
        r1 = np.full(w.shape, np.nan + 0j)
        r2 = np.full(w.shape, np.nan + 0j)
        for i, wi in enumerate(w):
            if st[i] != ST_OK:
                continue

            def f(x, y, wi=wi):
                _, D, cf, _, _ = self.interior_coefficients(x, k, wi)
                return [y[1], -D * y[1] - cf * y[0], y[3], -D * y[3] - cf * y[2]]

            sol = solve_ivp(f, (-1.0, 1.0), np.array([1, 0, 0, 1], dtype=complex), method="DOP853", rtol=rtol, atol=1e-30)
            r1[i], r2[i] = sol.y[0, -1], sol.y[2, -1]
        return self._finish(k, w, st, outer, r1, r2)

    # ---- uniform flow: closed form -----------------------------------------------------------------------------------
    def closed_form_uniform(self, k, w):
        """D_c when U(x) = U_i0 on the whole slab (width -> infinity): Vx'' = m0 Vx, sinh / cosh interior."""
        w = np.atleast_1d(np.asarray(w, dtype=complex))
        st, outer, _ = self.exterior(k, w)
        Om = w - k * self.U_i0
        S = self.c_i ** 2 + self.vA_i ** 2
        kc2, kvA2, kcT2 = k ** 2 * self.c_i ** 2, k ** 2 * self.vA_i ** 2, k ** 2 * self.cT_i2
        with np.errstate(all="ignore"):
            m = np.sqrt((kc2 - Om ** 2) * (kvA2 - Om ** 2) / (S * (kcT2 - Om ** 2)))
            Vb = Om / (w - k * self.U_e)
            slope = -Vb * m / np.tanh(m) if self.mode == "sausage" else -Vb * m * np.tanh(m)     # Vx'(-1)
            PTi = self.rho_i * S * (kcT2 - Om ** 2) / (Om * (kc2 - Om ** 2))
            inner = PTi * slope
            d = outer - inner
            rel = np.abs(d) * 100.0 / np.maximum(np.abs(outer), np.abs(inner))
        d = np.where(st == ST_OK, d, np.nan + 0j)
        return d, np.where(st == ST_OK, rel, np.nan), st.astype(np.uint8)

    # ---- root search on a (Re, Im) grid: the algorithm of es_complex_find_roots ------------------------------------
    @staticmethod
    def _quadrant(z):
        return np.where(z.real >= 0, np.where(z.imag >= 0, 0, 3), np.where(z.imag >= 0, 1, 2))

    def find_roots(self, k, w_re, w_im, n_iter=12, tol=4.0, evaluator=None):
        """Cells of the rectangular grid w_re x w_im around whose corners D_c winds once (quadrant count; an
        ambiguous half-turn edge also qualifies), refined by complex secant steps from the cell centre.
        Returns (roots complex array, rel array, flag array) ordered by (i_im, i_re)."""
        ev = evaluator or self.eval_rk4
        w_re, w_im = np.asarray(w_re, float), np.asarray(w_im, float)
        W = (w_re[None, :] + 1j * w_im[:, None])
        d, rel, st = ev(k, W.ravel())
        d, st = d.reshape(W.shape), st.reshape(W.shape)
        q = self._quadrant(d)

        def turns(a, b):
            t = (b - a) & 3
            return np.where(t == 0, 0, np.where(t == 1, 1, np.where(t == 3, -1, 8)))

        q00, q10, q11, q01 = q[:-1, :-1], q[:-1, 1:], q[1:, 1:], q[1:, :-1]
        total = turns(q00, q10) + turns(q10, q11) + turns(q11, q01) + turns(q01, q00)
        okc = (st[:-1, :-1] == 0) & (st[:-1, 1:] == 0) & (st[1:, 1:] == 0) & (st[1:, :-1] == 0)
        cand = okc & ((total == 4) | (total == -4) | (total >= 6))
        roots, rels, flags = [], [], []
        for im, ire in zip(*np.nonzero(cand)):
            a, b = W[im, ire], W[im + 1, ire + 1]
            centre, half = 0.5 * (a + b), 0.5 * (b - a)
            w0, w1 = centre, centre + 0.5 * half
            f0, f1 = ev(k, [w0])[0][0], ev(k, [w1])[0][0]
            for _ in range(n_iter):
                df = f1 - f0
                with np.errstate(all="ignore"):
                    w2 = w1 - f1 * ((w1 - w0) / df)
                if not np.isfinite(w2) or df == 0:
                    w2 = w1
                w0, f0, w1 = w1, f1, w2
                f1 = ev(k, [w1])[0][0]
                if not np.isfinite(f1):
                    w1, f1 = w0, f0
            dd, rr, ss = ev(k, [w1])
            ok = ss[0] == 0 and rr[0] < tol and abs(w1 - centre) ** 2 <= 16.0 * abs(half) ** 2
            roots.append(w1); rels.append(rr[0]); flags.append(1 if ok else 0)
        return np.array(roots, dtype=complex), np.array(rels), np.array(flags, dtype=int)
