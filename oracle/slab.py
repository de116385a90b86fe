"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's slab path:

 (A) analytic dispersion functions + sign-change scan of
     Slab/Non uniform flow/Solver/flow_multiprocessor.py (SF-U) :107-127 (functions), :131-152 (grids),
     :166-272 (scan loops), :284-303 (body-mode pole filter).  numpy calls are the reference's own
     (np.sqrt / np.tanh / np.tan on fp64), so values agree with the reference bit for bit.

 (B) shooting evaluation of the total-pressure mismatch of the slab workers
       SF-U :446-642 / :645-807   uniform flow,       Vx'' = m_i Vx                      (SF-U:566-567)
       SF-G :284-536 / :538-755   Gaussian flow,      Vx'' = -D Vx' - coeff Vx           (SF-G:460-461)
       SD-P :307-547 / :550-775   Gaussian density,   Vx'' = (-F'/F) Vx' + m0 Vx         (SD-P:479-480)
       SD-C                       same, coronal constants
     restated as in oracle/cylinder.py: exterior in closed form (exponentials) with the reference's far-field
     initial values, interior by DOP853 at rtol 1e-12, symmetry condition Vx(+1) = -/+ Vx(-1) imposed exactly
     by superposition (it is linear in the unknown slope) instead of fsolve.
     For the density slab the equation is integrated in flux form (F Vx')' = F m0 Vx, which is the same ODE.
     For the flow slab the reference's D(x) (SF-G:421-422) is *not* the logarithmic derivative of a flux
     function (its k^4 cT^2 c^2 term lacks a factor c^2 compared with the textbook equation), so that ODE is
     integrated exactly as written, with U' and U'' of the Gaussian in closed form.

Pinned by tests/golden/slab_analytic.npz and tests/golden/trace_S*.json (tools/gen_golden.py).
"""
import math
import numpy as np
from scipy.integrate import solve_ivp

GAMMA = 5.0 / 3.0
ST_OK, ST_LEAKY, ST_NONFINITE, ST_CONTINUUM = 0, 1, 2, 3

SAUSAGE, KINK, SAUSAGE_BODY, KINK_BODY = 0, 1, 2, 3


# =====================================================================================================
# (A) analytic slab dispersion relations, SF-U:63-127
# =====================================================================================================
class SlabAnalytic:
    def __init__(self, vA_i=1.0, c_i=2.0 / 3.0, vA_e=0.0, c_e=0.75, U_i=0.0, U_e=-0.15):
        self.vA_i, self.c_i, self.vA_e, self.c_e = vA_i, c_i, vA_e, c_e
        self.mach_i, self.mach_e = U_i, U_e                      # SF-U:97-98 (not divided by vA_i)
        rho_i = 1.0
        rho_e = rho_i * (c_i ** 2 + GAMMA * 0.5 * vA_i ** 2) / (c_e ** 2 + GAMMA * 0.5 * vA_e ** 2)   # SF-U:74
        self.R1 = rho_e / rho_i                                  # SF-U:79
        self.cT_i = np.sqrt(c_i ** 2 / (c_i ** 2 + vA_i ** 2))   # SF-U:85-86 (normalised form)
        # SF-U:88-89, operator precedence exactly as written: c_e^2 vA_e^2 / vA_i^2 * (c_e^2 + vA_e^2)
        self.cT_e = np.sqrt(c_e ** 2 * vA_e ** 2 / vA_i ** 2 * (c_e ** 2 + vA_e ** 2))

    def m0(self, W):
        c_i, vA_i, mi = self.c_i, self.vA_i, self.mach_i
        with np.errstate(all="ignore"):
            return np.sqrt((c_i ** 2 - (W - mi) ** 2) * (vA_i ** 2 - (W - mi) ** 2)
                           / ((c_i ** 2 + vA_i ** 2) * (self.cT_i ** 2 - (W - mi) ** 2)))

    def me(self, W):
        c_e, vA_e, me_ = self.c_e, self.vA_e, self.mach_e
        with np.errstate(all="ignore"):
            return np.sqrt((c_e ** 2 - (W - me_) ** 2) * (vA_e ** 2 - (W - me_) ** 2)
                           / ((c_e ** 2 + vA_e ** 2) * (self.cT_e ** 2 - (W - me_) ** 2)))

    def n0(self, W):
        c_i, vA_i, mi = self.c_i, self.vA_i, self.mach_i
        with np.errstate(all="ignore"):
            return np.sqrt(abs((c_i ** 2 - (W - mi) ** 2) * (vA_i ** 2 - (W - mi) ** 2)
                               / ((c_i ** 2 + vA_i ** 2) * (self.cT_i ** 2 - (W - mi) ** 2))))

    def disp(self, mode, W, K):
        """disp_rel_{sausage,kink,sausage_body,kink_body}(W, K), SF-U:117-127."""
        W = np.asarray(W, dtype=float)
        K = np.asarray(K, dtype=float)
        R1, vA_e, vA_i, mi, me_ = self.R1, self.vA_e, self.vA_i, self.mach_i, self.mach_e
        with np.errstate(all="ignore"):
            if mode == SAUSAGE:
                return R1 * (vA_e ** 2 - (W - me_) ** 2) * self.m0(W) * np.tanh(K * self.m0(W)) \
                    / (self.me(W) * (vA_i ** 2 - (W - mi) ** 2)) + 1
            if mode == KINK:
                return R1 * (vA_e ** 2 - (W - me_) ** 2) * self.m0(W) \
                    / (np.tanh(K * self.m0(W)) * self.me(W) * (vA_i ** 2 - (W - mi) ** 2)) + 1
            if mode == SAUSAGE_BODY:
                return R1 * (vA_e ** 2 - (W - me_) ** 2) * self.n0(W) * np.tan(K * self.n0(W)) \
                    / (self.me(W) * (vA_i ** 2 - (W - mi) ** 2)) - 1
            if mode == KINK_BODY:
                return R1 * (vA_e ** 2 - (W - me_) ** 2) * self.n0(W) \
                    / (np.tan(K * self.n0(W)) * self.me(W) * (vA_i ** 2 - (W - mi) ** 2)) + 1
        raise ValueError(mode)

    def scan(self, mode, K_values, W_values, step):
        """Sign-change scan SF-U:166-272: for each K, each V in W_values: f(V,K)*f(V+step,K) < 0 -> (K, (V+V+step)/2).

        Returns (K_out, W_mid) in the reference's loop order (K outer, V inner)."""
        K_values = np.asarray(K_values, dtype=float)
        V1 = np.asarray(W_values, dtype=float)
        V2 = V1 + step
        ks, ws = [], []
        for x in K_values:
            with np.errstate(all="ignore"):
                prod = self.disp(mode, V1, x) * self.disp(mode, V2, x)
            idx = np.nonzero(prod < 0)[0]
            for i in idx:
                ks.append(float(x))
                ws.append(float((V1[i] + V2[i]) / 2))
        return np.array(ks), np.array(ws)

    def pole_filter(self, mode, K_out, W_mid, thresh=1e-4):
        """SF-U:284-303: keep body-mode candidates with disp_rel(W_mid, K) < 1e-4 (one-sided, as written)."""
        with np.errstate(all="ignore"):
            v = np.array([self.disp(mode, w, k) for w, k in zip(W_mid, K_out)])
        keep = v < thresh
        return K_out[keep], W_mid[keep]


# =====================================================================================================
# (B) slab shooting workers
# =====================================================================================================
class SlabEquilibrium:
    """kind = "uniform_flow" (SF-U:63-99, 406-410), "flow" (SF-G:63-126), "density" (SD-P:68-162)."""

    def __init__(self, kind, c_i0=1.0, vA_i0=1.0, c_e=0.75, vA_e=0.0, rho_i0=1.0, width=1e5, x0=0.0,
                 U_i0=0.0, U_e=0.0):
        self.kind = kind
        self.c_i0, self.vA_i0, self.c_e, self.vA_e, self.rho_i0 = c_i0, vA_i0, c_e, vA_e, rho_i0
        self.width, self.x0, self.U_i0, self.U_e = width, x0, U_i0, U_e
        self.rho_e = rho_i0 * (c_i0 ** 2 + GAMMA * 0.5 * vA_i0 ** 2) / (c_e ** 2 + GAMMA * 0.5 * vA_e ** 2)
        den = c_e ** 2 + vA_e ** 2
        self.cT_e = math.sqrt(c_e ** 2 * vA_e ** 2 / den)           # SF-U:410, SD-P:162
        self.cT_i0 = math.sqrt(c_i0 ** 2 * vA_i0 ** 2 / (c_i0 ** 2 + vA_i0 ** 2))

    def gauss(self, x):
        return np.exp(-(np.asarray(x, dtype=float) - self.x0) ** 2 / self.width ** 2)

    def rho(self, x):
        if self.kind == "density":
            return self.rho_e + (self.rho_i0 - self.rho_e) * self.gauss(x)          # SD-P:102-103
        return np.full_like(np.asarray(x, dtype=float), self.rho_i0)

    def vA2(self, x):
        if self.kind == "density":
            return self.vA_i0 ** 2 * self.rho_i0 / self.rho(x)                      # SD-P:147-148
        return np.full_like(np.asarray(x, dtype=float), self.vA_i0 ** 2)

    def c2(self, x):
        if self.kind == "density":
            return self.rho_e * (self.c_e ** 2 + 0.5 * GAMMA * self.vA_e ** 2) / self.rho(x) \
                - 0.5 * GAMMA * self.vA2(x)                                         # SD-P:154-155
        return np.full_like(np.asarray(x, dtype=float), self.c_i0 ** 2)

    def U(self, x):
        if self.kind == "flow":
            return self.U_e + (self.U_i0 - self.U_e) * self.gauss(x)                # SF-G:124-126
        return np.full_like(np.asarray(x, dtype=float), self.U_i0)

    def dU(self, x):
        if self.kind == "flow":
            x = np.asarray(x, dtype=float)
            return (self.U_i0 - self.U_e) * self.gauss(x) * (-2.0 * (x - self.x0) / self.width ** 2)
        return np.zeros_like(np.asarray(x, dtype=float))

    def ddU(self, x):
        if self.kind == "flow":
            x = np.asarray(x, dtype=float)
            g = self.gauss(x)
            return (self.U_i0 - self.U_e) * g * (4.0 * (x - self.x0) ** 2 / self.width ** 4 - 2.0 / self.width ** 2)
        return np.zeros_like(np.asarray(x, dtype=float))


class SlabProblem:
    def __init__(self, eq, mode, L_factor=7.0, ic=(1e-8, 1e-15)):
        self.eq = eq
        self.mode = mode              # "sausage": Vx(+1) = -Vx(-1);  "kink": Vx(+1) = +Vx(-1)
        self.L_factor = float(L_factor)
        self.ic = (float(ic[0]), float(ic[1]))

    def exterior(self, k, w):
        """(m_e, p_e_const, V_b, dV_b), exterior solution scaled to |V_b| = 1 (amplitude sign kept)."""
        eq = self.eq
        k, w = np.float64(k), np.float64(w)          # IEEE semantics (inf/nan, no ZeroDivisionError), as numpy scalars in the reference
        with np.errstate(all="ignore"):
            Oe = w - k * eq.U_e
            k2 = k * k
            m_e = ((k2 * eq.vA_e ** 2 - Oe ** 2) * (k2 * eq.c_e ** 2 - Oe ** 2)) / \
                  ((eq.vA_e ** 2 + eq.c_e ** 2) * (k2 * eq.cT_e ** 2 - Oe ** 2))                       # SF-U:542
            p_e = eq.rho_e * (eq.vA_e ** 2 + eq.c_e ** 2) * (k2 * eq.cT_e ** 2 - Oe ** 2) / \
                (Oe * (k2 * eq.c_e ** 2 - Oe ** 2))                                                   # SF-U:545
        if not np.isfinite(m_e) or not (m_e >= 0.0):
            return m_e, p_e, float("nan"), float("nan")
        mu = math.sqrt(m_e)
        if mu == 0.0:
            return m_e, p_e, float("nan"), float("nan")
        R = self.L_factor * 2.0 * math.pi / k
        ic0, ic1 = self.ic
        E2 = math.exp(-2.0 * mu * (R - 1.0))
        gp, gm = ic0 + ic1 / mu, ic0 - ic1 / mu
        V = gp + E2 * gm
        dV = mu * (gp - E2 * gm)
        n = abs(V)
        return m_e, p_e, V / n, dV / n

    # interior ------------------------------------------------------------------------------------------
    def _coef(self, x, k, w):
        eq = self.eq
        rho, c2, vA2 = eq.rho(x), eq.c2(x), eq.vA2(x)
        S = c2 + vA2
        cT2 = c2 * vA2 / S
        Om = w - k * eq.U(x)
        k2 = k * k
        m0 = ((k2 * c2 - Om ** 2) * (k2 * vA2 - Om ** 2)) / (S * (k2 * cT2 - Om ** 2))       # SD-P:344 / SF-G:416
        F = rho * S * (k2 * cT2 - Om ** 2) / (k2 * c2 - Om ** 2)                              # SD-P:330
        return rho, c2, vA2, S, cT2, Om, m0, F

    def _rhs(self, x, y, k, w):
        xa = np.array([x])
        rho, c2, vA2, S, cT2, Om, m0, F = (v[0] for v in self._coef(xa, k, w))
        eq = self.eq
        out = np.empty_like(y)
        if eq.kind == "density":
            # flux form: y = (V, F V')
            out[0::2] = y[1::2] / F
            out[1::2] = F * m0 * y[0::2]
        else:
            k2 = k * k
            dU, ddU = eq.dU(xa)[0], eq.ddU(xa)[0]
            t = Om ** 2 - k2 * cT2
            Dref = 2.0 * k * dU * (t + (k2 * k2 * cT2 * c2) / (S * t)) / (Om * (Om ** 2 - k2 * c2))   # SF-G:421
            coeff = k * ddU / Om + k * dU * Dref / Om - m0                                           # SF-G:427
            out[0::2] = y[1::2]
            out[1::2] = -Dref * y[1::2] - coeff * y[0::2]
        return out

    def continuum(self, k, w, n=4001):
        x = np.linspace(-1.0, 1.0, n)
        rho, c2, vA2, S, cT2, Om, m0, F = self._coef(x, k, w)
        k2 = k * k
        terms = [k2 * cT2 - Om ** 2, k2 * c2 - Om ** 2, k2 * vA2 - Om ** 2, Om]
        return bool(any(np.any(np.sign(t) != np.sign(t[0])) for t in terms))

    def mismatch(self, k, w, rtol=1e-12, ext_override=None):
        """(d, P_e, P_i, status): left_P_solution[-1] - inside_P_solution[0], normalised to |Vx_e(-1)| = 1."""
        eq = self.eq
        m_e, p_e, Vb_e, dVb_e = self.exterior(k, w)
        if ext_override is not None and np.isfinite(m_e) and m_e >= 0:
            a = abs(ext_override[0])
            Vb_e, dVb_e = ext_override[0] / a, ext_override[1] / a
        if np.isfinite(m_e) and m_e < 0.0:
            return float("nan"), float("nan"), float("nan"), ST_LEAKY
        if not np.isfinite(Vb_e):
            return float("nan"), float("nan"), float("nan"), ST_NONFINITE
        if self.continuum(k, w):
            return float("nan"), float("nan"), float("nan"), ST_CONTINUUM
        st = ST_OK
        with np.errstate(all="ignore"):
            xb = np.array([-1.0])
            rho, c2, vA2, S, cT2, Om, m0, F = (v[0] for v in self._coef(xb, k, w))
            Oe = w - k * eq.U_e
            Vb = Vb_e * Om / Oe if eq.kind != "density" else Vb_e           # SF-U:558 ; SD-P:473
            P_left = p_e * dVb_e                                            # SF-U:559
            y0 = np.array([1.0, 0.0, 0.0, 1.0])
            sol = solve_ivp(self._rhs, (-1.0, 1.0), y0, method="DOP853", rtol=rtol, atol=1e-300, args=(k, w))
            y = sol.y[:, -1]
            T11, T12 = y[0], y[2]
            sgn = -1.0 if self.mode == "sausage" else 1.0                   # V(+1) = sgn * V(-1)
            s = (sgn - T11) * Vb / T12                                      # second state component at x=-1
            P_Ti = F / Om                                                   # SD-P:346 (Om = w), SF-G:433
            if eq.kind == "density":
                P_in = P_Ti * (s / F)                                       # s = F V'
            else:
                P_in = P_Ti * s
            d = P_left - P_in
        if not np.isfinite(d):
            st = ST_NONFINITE
        return d, P_left, P_in, st
