"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's cylinder shooting evaluation of the boundary "determinant"
D(k, omega; m) = xi_e(boundary) - xi_i(boundary)  (radial-displacement mismatch), following

  * Cylinder/Non-uniform density/Coronal/solvers/Density_cylinder.py            (CD-C)  :546-824, :847-1124
  * Cylinder/Non-uniform density/Photospheric/Solvers/Density_cylinder_photospheric.py (CD-P)
  * Cylinder/Non-uniform flow/Coronal/solvers/Cylinder_method_flow_testing.py   (CF)    :554-839, :855-1133
  * Cylinder/Rotational flow/Photospheric/Solvers/Twisted_photospheric_*.py     (CR-*)  e.g. CR-KF:454-734

What the reference does per (k, omega) (CF:694-804 is representative):
  1. exterior: m_e and xi_e_const (CF:699-702); if m_e < 0 the point is skipped (CF:760);
     P'' = -P'/r + (m_e + m^2/r^2) P integrated by LSODA from the far field r = -/+ 3*2pi/k with
     P0 = [1e-8, 1e-8] (or [1e-8, 1e-15]) to the boundary |r| = 1 (CF:765-777);
  2. interior: the Hain-Luest / SGH coefficient set D, Q, T, C1, C2, C3 (CF:577-626), F = r D / C3,
     g = -(r C1/C3)' - r (C2 - C1^2/C3)/D, ODE (F P')' = g P on ix = linspace(-/+1, -/+r_ax, N) with
     P(boundary) = P_e(boundary) and the unknown slope found with fsolve so that the axis condition holds
     (kink: P(r_ax) = const * xi_e, sausage: P'(r_ax) = 0) (CF:782-794);
  3. xi_i = (C1 P + D P')/C3 at the boundary (CF:798), mismatch d = xi_e - xi_i (CF:803).

Restatement used here (mathematically identical, no symbolic differentiation needed):
  * the second-order ODE together with xi = (C1 P + D P')/C3 is equivalent to the first-order SGH pair
        D P'  = C3 xi - C1 P,        D (r xi)' = C1 (r xi) - r C2 P,
    (differentiate F P' + (r C1/C3) P = r xi and substitute: gives exactly the reference's g);
  * the exterior ODE is the modified Bessel equation: P = a I_m(mu |r|) + b K_m(mu |r|), mu = sqrt(m_e), with
    (a, b) fixed by the reference's far-field initial values -- closed form via scipy.special.ive / kve;
  * the axis condition is linear in the unknown boundary slope, so it is imposed exactly by superposition of two
    interior solves instead of fsolve (fsolve's answer when it converges).
The interior integration uses scipy's DOP853 at rtol = 1e-12 ("truth" for the fixed-grid RK4 port in
oracle/c/ and for the HIP kernels).

Pinned against the reference itself: tests/golden/trace_*.json hold amplitude-normalised mismatches produced by
executing the reference workers in the build container (tools/gen_golden.py); tests/test_oracle_golden.py
checks this module against them.  LSODA's tolerance (1.5e-8) bounds that comparison, see SURVEY section 8c.
"""
import math
import numpy as np
from scipy import special
from scipy.integrate import solve_ivp

GAMMA = 5.0 / 3.0

# status codes shared by oracle and product (include/eigensolver_amd.h)
ST_OK, ST_LEAKY, ST_NONFINITE, ST_CONTINUUM = 0, 1, 2, 3


class CylinderEquilibrium:
    """Closed-form restatement of the sympy equilibrium blocks (layer L0 of each cylinder script).

    kind = "density" : Gaussian density  rho_e + (rho_i0 - rho_e) exp(-(r-r0)^2/dr^2)          (CD-C:135-136)
                       B_i = B_0, vA_i = (B_i + B_phi)/sqrt(rho)                                 (CD-C:188-200)
           "flow"    : constant density, Gaussian axial flow U_e + (U_i0 - U_e) exp(-(r-r0)^2/dr^2) (CF:134-135)
           "rotation": constant density, v_phi = v_twist r^power, P_i = rho v_twist^2 r^(2p)/(2p) + P_0,
                       c_i = sqrt(gamma P_i / rho)                                               (CR-KF:176-189)
    c_i for "density"/"flow": sqrt(rho_e (c_e^2 + gamma/2 vA_e^2)/rho - gamma/2 vA_i^2)           (CD-C:210-211, CF:207-208)
    """

    def __init__(self, kind, c_i0=1.0, vA_i0=2.0, c_e=0.5, vA_e=5.0, rho_i0=1.0, width=1e5, r0=0.0,
                 U_i0=0.0, U_e=0.0, v_twist=0.0, power=1.0, B_twist=0.0):
        self.kind = kind
        self.c_i0, self.vA_i0, self.c_e, self.vA_e, self.rho_i0 = c_i0, vA_i0, c_e, vA_e, rho_i0
        self.width, self.r0, self.U_i0, self.U_e = width, r0, U_i0, U_e
        self.v_twist, self.power, self.B_twist = v_twist, power, B_twist
        # CD-C:77-80
        self.rho_e = rho_i0 * (c_i0 ** 2 + GAMMA * 0.5 * vA_i0 ** 2) / (c_e ** 2 + GAMMA * 0.5 * vA_e ** 2)
        self.cT_e = math.sqrt(c_e ** 2 * vA_e ** 2 / (c_e ** 2 + vA_e ** 2))          # CD-C:75
        self.cT_i0 = math.sqrt(c_i0 ** 2 * vA_i0 ** 2 / (c_i0 ** 2 + vA_i0 ** 2))     # CD-C:74
        self.B_0 = vA_i0 * math.sqrt(rho_i0)                                          # CD-C:103
        self.P_0 = c_i0 ** 2 * rho_i0 / GAMMA                                         # CD-C:97
        self.c_kink = math.sqrt((rho_i0 * vA_i0 ** 2 + self.rho_e * vA_e ** 2) / (rho_i0 + self.rho_e))

    # --- profile functions of r (signed radial coordinate, as the reference evaluates them) ---------------
    def rho(self, r):
        r = np.asarray(r, dtype=float)
        if self.kind == "density":
            return self.rho_e + (self.rho_i0 - self.rho_e) * np.exp(-(r - self.r0) ** 2 / self.width ** 2)
        return np.full_like(r, self.rho_i0)

    def v_z(self, r):
        r = np.asarray(r, dtype=float)
        if self.kind == "flow":
            return self.U_e + (self.U_i0 - self.U_e) * np.exp(-(r - self.r0) ** 2 / self.width ** 2)
        return np.zeros_like(r)

    def v_phi(self, r):
        r = np.asarray(r, dtype=float)
        if self.kind == "rotation":
            return self.v_twist * r ** self.power
        return np.zeros_like(r)

    def B_phi(self, r):
        r = np.asarray(r, dtype=float)
        return self.B_twist * r if self.B_twist != 0.0 else np.zeros_like(r)

    def B_z(self, r):
        # CF:185-186  B_0 sqrt(1 - 2 B_phi^2/B_0^2); CD-C:199-200 B_0
        return self.B_0 * np.sqrt(1.0 - 2.0 * self.B_phi(r) ** 2 / self.B_0 ** 2)

    def vA(self, r):
        # (B_i + B_iphi)/sqrt(rho) exactly as written (CF:173-174, CD-C:188-189)
        return (self.B_z(r) + self.B_phi(r)) / np.sqrt(self.rho(r))

    def c2(self, r):
        r = np.asarray(r, dtype=float)
        if self.kind == "rotation":
            P_i = self.rho(r) * self.v_twist ** 2 * (r ** (2.0 * self.power) / (2.0 * self.power)) + self.P_0
            return P_i * GAMMA / self.rho(r)
        return self.rho_e * (self.c_e ** 2 + 0.5 * GAMMA * self.vA_e ** 2) / self.rho(r) - 0.5 * GAMMA * self.vA(r) ** 2

    def r_dC3diff(self, r):
        """r d/dr [ (B_phi/r)^2 - rho (v_phi/r)^2 ]   (C3_diff, CF:610-611) -- analytic for the profiles above."""
        r = np.asarray(r, dtype=float)
        out = np.zeros_like(r)
        if self.kind == "rotation":
            # rho const: -rho v_twist^2 d/dr r^(2p-2) * r = -rho v_twist^2 (2p-2) r^(2p-2)
            out = out - self.rho(r) * self.v_twist ** 2 * (2.0 * self.power - 2.0) * r ** (2.0 * self.power - 2.0)
        # B_phi = B_twist r -> (B_phi/r)^2 constant -> derivative 0
        return out


class CylinderProblem:
    """One reference worker configuration (everything the reference keeps in module globals)."""

    def __init__(self, eq, m, r_sign=-1.0, r_axis=1e-3, L_factor=3.0, ic=(1e-8, 1e-8), c1_power=2,
                 axis_bc="kink", m_ext=None):
        self.eq = eq
        self.m = float(m)
        self.m_ext = float(m if m_ext is None else m_ext)   # order hard-coded in the exterior ODE (CF:769, CD-C:1063)
        self.r_sign = float(r_sign)        # -1: ix = linspace(-1, -r_ax) (CD/CF); +1: linspace(1, r_ax) (CR)
        self.r_axis = float(r_axis)
        self.L_factor = float(L_factor)    # far field at |r| = L_factor * 2 pi / k
        self.ic = (float(ic[0]), float(ic[1]))
        self.c1_power = int(c1_power)      # C1 = Q*Omega (CD-C:590) or Q*Omega^2 (CF:598, CR-KF:493)
        self.axis_bc = axis_bc             # "kink": P(r_ax) = c * xi_e ; "sausage": P'(r_ax) = 0 ; "rotation_kink"

    # ---- exterior (CF:699-702, 765-777) ----------------------------------------------------------------
    def exterior(self, k, w):
        """Return (m_e, xi_e_const, P_b, dP_b) with the exterior solution scaled so that |P_b| = 1.

        sign(P_b) is the sign of the reference's LSODA amplitude at the boundary (positive initial values)."""
        eq = self.eq
        k, w = np.float64(k), np.float64(w)          # IEEE semantics (inf/nan, no ZeroDivisionError), as numpy scalars in the reference
        with np.errstate(all="ignore"):
            k2 = k * k
            w2 = w * w
            m_e = ((k2 * eq.vA_e ** 2 - w2) * (k2 * eq.c_e ** 2 - w2)) / \
                  ((eq.vA_e ** 2 + eq.c_e ** 2) * (k2 * eq.cT_e ** 2 - w2))
            xi_e_const = -1.0 / (eq.rho_e * (k2 * eq.vA_e ** 2 - w2))
        if not np.isfinite(m_e) or not (m_e >= 0.0):
            return m_e, xi_e_const, float("nan"), float("nan")
        mu = math.sqrt(m_e)
        sgn = self.r_sign
        R = self.L_factor * 2.0 * math.pi / k
        xR, xb = mu * R, mu * 1.0
        n = self.m_ext
        if mu == 0.0:
            return m_e, xi_e_const, float("nan"), float("nan")
        # scaled Bessel functions: ive = e^-x I, kve = e^x K ; derivatives from recurrences
        def I_pair(x):
            i0, i1 = special.ive(n, x), special.ive(n + 1, x)
            return i0, i1 + (n / x) * i0            # e^-x I_n, e^-x I_n'
        def K_pair(x):
            k0, k1 = special.kve(n, x), special.kve(n + 1, x)
            return k0, -k1 + (n / x) * k0           # e^x K_n, e^x K_n'
        IR, dIR = I_pair(xR)
        KR, dKR = K_pair(xR)
        Ib, dIb = I_pair(xb)
        Kb, dKb = K_pair(xb)
        ic0, ic1 = self.ic
        g = ic1 / (sgn * mu)                         # d/d|x| of P at the far point, per unit mu
        # a = -xR (ic0 K' - g K),  b = -xR (g I - ic0 I')   [Wronskian I K' - I' K = -1/x]
        a_s = -(ic0 * dKR - g * KR)                  # a / (xR e^{xR})  (scaled)
        b_s = -(g * IR - ic0 * dIR)                  # b / (xR e^{-xR})
        E2 = math.exp(-2.0 * (xR - xb))
        P = b_s * Kb + E2 * a_s * Ib                 # common factor xR e^{xR - xb} dropped (positive)
        dP = sgn * mu * (b_s * dKb + E2 * a_s * dIb)
        nrm = abs(P)
        return m_e, xi_e_const, P / nrm, dP / nrm

    # ---- interior coefficient set (CF:577-626) -----------------------------------------------------------
    def coefficients(self, r, k, w):
        eq, m = self.eq, self.m
        rho, vz, vphi, Bphi, Bz = eq.rho(r), eq.v_z(r), eq.v_phi(r), eq.B_phi(r), eq.B_z(r)
        c2, vA = eq.c2(r), eq.vA(r)
        S = c2 + vA ** 2
        Om = w - m * vphi / r - k * vz                                  # shift_freq (CF:578)
        wA = m * Bphi / r + (k * Bz) / np.sqrt(rho)                      # alfven_freq, precedence as written (CF:581)
        wc = wA * np.sqrt(c2) / np.sqrt(S)                               # cusp_freq (CF:584)
        t1 = Om ** 2 - wA ** 2
        t2 = Om ** 2 - wc ** 2
        D = rho * S * t1 * t2                                            # CF:587
        kb = m * Bphi / r + k * Bz
        Q = -t1 * rho * vphi ** 2 / r + 2.0 * Om ** 2 * Bphi ** 2 / r + 2.0 * Om * Bphi * vphi * kb / r   # CF:592
        T = kb * Bphi + rho * vphi * Om                                  # CF:595
        C1 = Q * Om ** self.c1_power - 2.0 * m * S * t2 * T / r ** 2     # CF:598 / CD-C:590
        C2 = Om ** 4 - S * (m ** 2 / r ** 2 + k ** 2) * t2               # CF:603
        C3 = D * (rho * t1 + eq.r_dC3diff(r)) + Q ** 2 - 4.0 * S * t2 * T ** 2 / r ** 2   # CF:609-614
        return D, C1, C2, C3, t1, t2

    def _rhs(self, r, y, k, w):
        D, C1, C2, C3, _, _ = self.coefficients(np.array([r]), k, w)
        D, C1, C2, C3 = D[0], C1[0], C2[0], C3[0]
        P, X = y[0::2], y[1::2]                  # two columns interleaved
        dP = (C3 / (r * D)) * X - (C1 / D) * P
        dX = (C1 / D) * X - (r * C2 / D) * P
        out = np.empty_like(y)
        out[0::2], out[1::2] = dP, dX
        return out

    def transfer(self, k, w, rtol=1e-12):
        """Transfer matrix of (P, Xi = r xi) from the boundary r_b = r_sign to the axis point r_sign*r_axis."""
        rb, ra = self.r_sign * 1.0, self.r_sign * self.r_axis
        y0 = np.array([1.0, 0.0, 0.0, 1.0])     # columns (P,Xi)=(1,0) and (0,1), interleaved as P1,X1,P2,X2
        sol = solve_ivp(self._rhs, (rb, ra), y0, method="DOP853", rtol=rtol, atol=1e-300, args=(k, w))
        y = sol.y[:, -1]
        return np.array([[y[0], y[2]], [y[1], y[3]]])   # [[T11,T12],[T21,T22]]

    def continuum(self, k, w, n=4001):
        """True if Omega^2 - omega_A^2(r) or Omega^2 - omega_c^2(r) changes sign inside the interior domain."""
        r = np.linspace(self.r_sign, self.r_sign * self.r_axis, n)
        _, _, _, _, t1, t2 = self.coefficients(r, k, w)
        return bool(np.any(np.sign(t1) != np.sign(t1[0])) or np.any(np.sign(t2) != np.sign(t2[0])))

    # ---- the determinant -----------------------------------------------------------------------------------
    def mismatch(self, k, w, rtol=1e-12, ext_override=None):
        """Return (d, xi_e, xi_i, status): amplitude-normalised so that |P_e(boundary)| = 1 (sign kept).

        ext_override = (P_b, dP_b): use this exterior end state (e.g. the reference's own LSODA result from a golden
        trace) instead of the closed form -- isolates the interior part of the comparison."""
        m_e, xi_c, Pb, dPb = self.exterior(k, w)
        if ext_override is not None and np.isfinite(m_e) and m_e >= 0:
            a = abs(ext_override[0])
            Pb, dPb = ext_override[0] / a, ext_override[1] / a
        if np.isfinite(m_e) and m_e < 0.0:
            return float("nan"), float("nan"), float("nan"), ST_LEAKY
        if not np.isfinite(Pb):
            return float("nan"), float("nan"), float("nan"), ST_NONFINITE
        xi_e = xi_c * dPb                                          # left_xi_solution[-1] (CF:775)
        rb, ra = self.r_sign, self.r_sign * self.r_axis
        if self.continuum(k, w):
            # a coefficient is singular inside the domain: the adaptive integrator cannot cross it; the port and the
            # HIP kernel step over it on the fixed grid and flag the lane (their D there is not compared)
            return float("nan"), xi_e, float("nan"), ST_CONTINUUM
        st = ST_OK
        with np.errstate(all="ignore"):
            T = self.transfer(k, w, rtol)
            eq = self.eq
            if self.axis_bc == "kink":
                # P(r_ax) = B_phi(-1)^2 * xi_e   (CF:795;  zero target in CD-C:787)
                target = float(eq.B_phi(np.array([rb]))[0]) ** 2 * xi_e
                Xb = (target - T[0, 0] * Pb) / T[0, 1]
            elif self.axis_bc == "rotation_kink":
                # P(r_ax) + (B_phi(1)^2 - rho(1) v_phi(1)^2) xi_e = 0   (CR-KF:695-698)
                one = np.array([1.0])
                cst = float(eq.B_phi(one)[0]) ** 2 - float(eq.rho(one)[0]) * float(eq.v_phi(one)[0]) ** 2
                Xb = (-cst * xi_e - T[0, 0] * Pb) / T[0, 1]
            elif self.axis_bc == "sausage":
                # P'(r_ax) = 0 with P' = C3/(r D) Xi - C1/D P at r_ax   (CD-C:1082-1085)
                D, C1, C2, C3, _, _ = self.coefficients(np.array([ra]), k, w)
                al, be = C3[0] / (ra * D[0]), -C1[0] / D[0]
                # al*(T21 Pb + T22 Xb) + be*(T11 Pb + T12 Xb) = 0
                Xb = -(al * T[1, 0] + be * T[0, 0]) * Pb / (al * T[1, 1] + be * T[0, 1])
            else:
                raise ValueError(self.axis_bc)
            xi_i = Xb / rb                                           # inside_xi_solution[0] (CF:798)
            d = xi_e - xi_i
        if not np.isfinite(d):
            st = ST_NONFINITE
        return d, xi_e, xi_i, st


def eigenfunction(prob, k, w, n_nodes, n_ext=500, rtol=1e-12):
    """ORACLE: the two-region solution at (k, omega) the reference's analysis scripts plot
    (Cylinder/Non-uniform flow/Coronal/Eigenfunctions/analysis_cylinder_flow_coronal.py:813-924): interior P and
    xi_r on linspace(r_b, r_ax, n_nodes) by DOP853 from the boundary state fixed by the axis condition, exterior on
    linspace(-/+ L 2pi/k, -/+1, n_ext) in closed form; scaled so that |P_e(r_b)| = 1 (sign of the reference's amplitude).
    Returns dict(r_int, P_int, xi_int, r_ext, P_ext, xi_ext)."""
    eq = prob.eq
    m_e, xi_c, Pb, dPb = prob.exterior(k, w)
    xi_e = xi_c * dPb
    rb, ra = prob.r_sign, prob.r_sign * prob.r_axis
    T = prob.transfer(k, w, rtol)
    if prob.axis_bc == "kink":
        target = float(eq.B_phi(np.array([rb]))[0]) ** 2 * xi_e
        Xb = (target - T[0, 0] * Pb) / T[0, 1]
    elif prob.axis_bc == "rotation_kink":
        one = np.array([1.0])
        cst = float(eq.B_phi(one)[0]) ** 2 - float(eq.rho(one)[0]) * float(eq.v_phi(one)[0]) ** 2
        Xb = (-cst * xi_e - T[0, 0] * Pb) / T[0, 1]
    else:
        D, C1, C2, C3, _, _ = prob.coefficients(np.array([ra]), k, w)
        al, be = C3[0] / (ra * D[0]), -C1[0] / D[0]
        Xb = -(al * T[1, 0] + be * T[0, 0]) * Pb / (al * T[1, 1] + be * T[0, 1])
    r_int = np.linspace(rb, ra, n_nodes)
    sol = solve_ivp(prob._rhs, (rb, ra), np.array([Pb, Xb]), method="DOP853", rtol=rtol, atol=1e-300,
                    t_eval=r_int, args=(k, w))
    P_int, X_int = sol.y[0], sol.y[1]
    # exterior
    mu = math.sqrt(m_e)
    sgn = prob.r_sign
    R = prob.L_factor * 2.0 * math.pi / k
    r_ext = np.linspace(sgn * R, sgn * 1.0, n_ext)
    n = prob.m_ext
    xR, xb = mu * R, mu
    x = mu * np.abs(r_ext)
    ic0, ic1 = prob.ic
    g = ic1 / (sgn * mu)
    KR, KR1 = special.kve(n, xR), special.kve(n + 1, xR)
    IR, IR1 = special.ive(n, xR), special.ive(n + 1, xR)
    dKR, dIR = -KR1 + (n / xR) * KR, IR1 + (n / xR) * IR
    a_s, b_s = -(ic0 * dKR - g * KR), -(g * IR - ic0 * dIR)
    Kx, Kx1, Ix, Ix1 = special.kve(n, x), special.kve(n + 1, x), special.ive(n, x), special.ive(n + 1, x)
    dKx, dIx = -Kx1 + (n / x) * Kx, Ix1 + (n / x) * Ix
    Kb, Ib = special.kve(n, xb), special.ive(n, xb)
    den = abs(b_s * Kb + math.exp(-2 * (xR - xb)) * a_s * Ib)
    dec = np.exp(-(x - xb))
    E2x = np.exp(-2 * (xR - x))
    P_ext = dec * (b_s * Kx + E2x * a_s * Ix) / den
    dP_ext = sgn * mu * dec * (b_s * dKx + E2x * a_s * dIx) / den
    return dict(r_int=r_int, P_int=P_int, xi_int=X_int / r_int, r_ext=r_ext, P_ext=P_ext, xi_ext=xi_c * dP_ext)


def uniform_closed_form(eq, k, w, m, r_sign=-1.0, r_axis=1e-3, L_factor=3.0, ic=(1e-8, 1e-8), axis_bc="kink",
                        U_i=0.0):
    """ORACLE: the determinant of the UNIFORM cylinder (profile width -> infinity, the reference's benchmark case) in
    closed form with scipy's Bessel functions: interior I_m/K_m (m_i > 0) or J_m/Y_m (m_i < 0) of sqrt(|m_i|) |r| with
    the second-kind admixture fixed by the axis condition at r_axis, exterior as CylinderProblem.exterior.
    Returns (d, xi_e, xi_i, status) in the normalisation of CylinderProblem.mismatch.
    Follows SURVEY.md section 8a ("uniform, untwisted, static limit") and appendix A.4."""
    prob = CylinderProblem(eq, m, r_sign=r_sign, r_axis=r_axis, L_factor=L_factor, ic=ic, axis_bc=axis_bc)
    m_e, xi_c, Pb, dPb = prob.exterior(k, w)
    if np.isfinite(m_e) and m_e < 0.0:
        return float("nan"), float("nan"), float("nan"), ST_LEAKY
    if not np.isfinite(Pb):
        return float("nan"), float("nan"), float("nan"), ST_NONFINITE
    c2, vA2, rho = eq.c_i0 ** 2, eq.vA_i0 ** 2, eq.rho_i0
    S = c2 + vA2
    cT2 = c2 * vA2 / S
    k2 = k * k
    Om = w - k * U_i
    Om2 = Om * Om
    with np.errstate(all="ignore"):
        m_i = (k2 * vA2 - Om2) * (k2 * c2 - Om2) / (S * (k2 * cT2 - Om2))
    n = float(m)
    if m_i > 0:
        kap = math.sqrt(m_i)
        xa, xb = kap * r_axis, kap

        def f1(x):   # e^-x I, e^-x I'
            a, b = special.ive(n, x), special.ive(n + 1, x)
            return a, b + (n / x) * a

        def f2(x):   # e^x K, e^x K'
            a, b = special.kve(n, x), special.kve(n + 1, x)
            return a, -b + (n / x) * a
        Ia, dIa = f1(xa); Ka, dKa = f2(xa); Ib, dIb = f1(xb); Kb, dKb = f2(xb)
        E2 = math.exp(-2.0 * (xb - xa))
        g = -(dIa / dKa) if axis_bc == "sausage" else -(Ia / Ka)
        ld = kap * (dIb + E2 * g * dKb) / (Ib + E2 * g * Kb)
    elif m_i < 0:
        kap = math.sqrt(-m_i)
        xa, xb = kap * r_axis, kap

        def fj(x):
            a, b = special.jv(n, x), special.jv(n + 1, x)
            return a, -b + (n / x) * a

        def fy(x):
            a, b = special.yv(n, x), special.yv(n + 1, x)
            return a, -b + (n / x) * a
        Ja, dJa = fj(xa); Ya, dYa = fy(xa); Jb, dJb = fj(xb); Yb, dYb = fy(xb)
        g = -(dJa / dYa) if axis_bc == "sausage" else -(Ja / Ya)
        ld = kap * (dJb + g * dYb) / (Jb + g * Yb)
    else:
        return float("nan"), float("nan"), float("nan"), ST_NONFINITE
    xi_e = xi_c * dPb
    xi_i = r_sign * ld * Pb / (rho * (Om2 - k2 * vA2))
    d = xi_e - xi_i
    return d, xi_e, xi_i, (ST_OK if np.isfinite(d) else ST_NONFINITE)
