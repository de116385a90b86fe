"""Equilibrium / profile definitions of the reference solvers in closed form (layer L0 of every script).

The reference builds these with sympy + lambdify at import time and edits the source to change parameters
("config / flags").  Here each geometry is a small parameter object that can sample its
profiles on the 2N-1 points (nodes and midpoints) the HIP propagator consumes.

  CylinderDensity   Cylinder/Non-uniform density/{Coronal,Photospheric}: Density_cylinder.py:69-221 (CD-C),
                    Density_cylinder_photospheric.py (CD-P)
  CylinderFlow      Cylinder/Non-uniform flow/Coronal: Cylinder_method_flow_testing.py:69-221 (CF)
  CylinderRotation  Cylinder/Rotational flow/Photospheric: Twisted_photospheric_*.py:110-221 (CR-*)
  SlabDensity       Slab/Non uniform density: multiprocessor_Inhomogeneous_method*.py:68-162 (SD-P, SD-C)
  SlabFlow          Slab/Non uniform flow: flow_multiprocessor.py:63-99,406-410 (SF-U uniform),
                    flow_multiprocessor_coronal.py:63-152 (SF-G Gaussian)
"""
import math
from dataclasses import dataclass, field

import numpy as np

GAMMA = 5.0 / 3.0


def sample_points(x_boundary, x_end, n_nodes):
    """x_j, j = 0..2N-2: the reference's interior grid linspace(x_boundary, x_end, N) plus its midpoints."""
    return np.linspace(x_boundary, x_end, 2 * n_nodes - 1)


@dataclass
class _CylinderBase:
    c_i0: float = 1.0
    vA_i0: float = 2.0
    c_e: float = 0.5
    vA_e: float = 5.0          # coronal defaults (CD-C:69-72); photospheric: vA_e = 0.5, c_e = 1.5 (CD-P:70-72)
    rho_i0: float = 1.0
    r_sign: float = -1.0       # CD-C / CF integrate on negative r, CD-P / CR-* on positive r
    r_axis: float = 1e-3       # |last node| of ix (CD-C:120); 0.01 in CR-SF:157
    n_nodes: int = 500         # len(ix)
    L_factor: float = 3.0      # lx = linspace(-/+ 3*2pi/k, -/+1, 500)
    ic: tuple = (1e-8, 1e-8)   # exterior initial values P0
    c1_power: int = 2          # C1 = Q*Omega^2 (CF:598, CR-KF:493); 1 in CD-C:590 / CD-P
    U_e: float = 0.0

    @property
    def rho_e(self):
        return self.rho_i0 * (self.c_i0 ** 2 + GAMMA * 0.5 * self.vA_i0 ** 2) / (self.c_e ** 2 + GAMMA * 0.5 * self.vA_e ** 2)

    @property
    def cT_e(self):
        return math.sqrt(self.c_e ** 2 * self.vA_e ** 2 / (self.c_e ** 2 + self.vA_e ** 2))

    @property
    def cT_i0(self):
        return math.sqrt(self.c_i0 ** 2 * self.vA_i0 ** 2 / (self.c_i0 ** 2 + self.vA_i0 ** 2))

    @property
    def c_kink(self):
        return math.sqrt((self.rho_i0 * self.vA_i0 ** 2 + self.rho_e * self.vA_e ** 2) / (self.rho_i0 + self.rho_e))

    @property
    def B_0(self):
        return self.vA_i0 * math.sqrt(self.rho_i0)

    @property
    def x_boundary(self):
        return self.r_sign * 1.0

    @property
    def x_end(self):
        return self.r_sign * self.r_axis

    # defaults for the un-twisted, static cylinder
    def rho(self, r):
        return np.full_like(r, self.rho_i0)

    def v_z(self, r):
        return np.zeros_like(r)

    def v_phi(self, r):
        return np.zeros_like(r)

    def B_phi(self, r):
        return np.zeros_like(r)

    def B_z(self, r):
        return self.B_0 * np.sqrt(1.0 - 2.0 * self.B_phi(r) ** 2 / self.B_0 ** 2)      # CF:185-186

    def vA(self, r):
        return (self.B_z(r) + self.B_phi(r)) / np.sqrt(self.rho(r))                     # CF:173-174, as written

    def c2(self, r):
        return self.rho_e * (self.c_e ** 2 + 0.5 * GAMMA * self.vA_e ** 2) / self.rho(r) - 0.5 * GAMMA * self.vA(r) ** 2

    def rdC3(self, r):
        return np.zeros_like(r)

    twisted = False

    def bc_const(self, axis_bc):
        return 0.0

    def profiles(self):
        r = sample_points(self.x_boundary, self.x_end, self.n_nodes)
        return dict(r=r, rho=self.rho(r), c2=self.c2(r), Bz=self.B_z(r), Bphi=self.B_phi(r), vz=self.v_z(r),
                    vphi=self.v_phi(r), rdC3=self.rdC3(r))


@dataclass
class CylinderDensity(_CylinderBase):
    """Gaussian density rho_e + (rho_i0 - rho_e) exp(-(r-r0)^2/dr^2), B_i = B_0 (CD-C:135-136, 199-200)."""
    width: float = 0.95
    r0: float = 0.0
    c1_power: int = 1
    ic: tuple = (1e-8, 1e-15)      # CD-C:768 ; CD-P uses (1e-8, 1e-8)

    def rho(self, r):
        return self.rho_e + (self.rho_i0 - self.rho_e) * np.exp(-(r - self.r0) ** 2 / self.width ** 2)

    def B_z(self, r):
        return np.full_like(r, self.B_0)


@dataclass
class CylinderFlow(_CylinderBase):
    """Constant density, Gaussian axial flow v_z = U_e + (U_i0 - U_e) exp(-(r-r0)^2/dr^2) (CF:134-135)."""
    width: float = 1e5
    r0: float = 0.0
    U_i0: float = 0.0
    n_nodes: int = 1000            # CF:120

    def v_z(self, r):
        return self.U_e + (self.U_i0 - self.U_e) * np.exp(-(r - self.r0) ** 2 / self.width ** 2)


@dataclass
class CylinderRotation(_CylinderBase):
    """v_phi = v_twist r^power, P_i = rho v_twist^2 r^(2p)/(2p) + P_0, c_i = sqrt(gamma P_i/rho) (CR-KF:176-189)."""
    v_twist: float = 0.25
    power: float = 0.8
    c_e: float = 1.5
    vA_e: float = 0.5
    r_sign: float = 1.0
    n_nodes: int = 2000            # CR-KF:157
    twisted = True

    def v_phi(self, r):
        return self.v_twist * r ** self.power

    def c2(self, r):
        P_0 = self.c_i0 ** 2 * self.rho_i0 / GAMMA                                       # CR-KF:128
        P_i = self.rho(r) * self.v_twist ** 2 * (r ** (2.0 * self.power) / (2.0 * self.power)) + P_0
        return P_i * GAMMA / self.rho(r)

    def rdC3(self, r):
        # r d/dr[-rho v_twist^2 r^(2p-2)]  (B_phi = 0, rho constant)  -- C3_diff, CR-KF:512-513
        return -self.rho(r) * self.v_twist ** 2 * (2.0 * self.power - 2.0) * r ** (2.0 * self.power - 2.0)

    def bc_const(self, axis_bc):
        one = np.array([1.0])
        return float(self.B_phi(one)[0] ** 2 - self.rho(one)[0] * self.v_phi(one)[0] ** 2)   # CR-KF:696


@dataclass
class _SlabBase:
    c_i0: float = 1.0
    vA_i0: float = 1.9
    c_e: float = 1.3
    vA_e: float = 0.8              # photospheric density slab (SD-P:70-74)
    rho_i0: float = 1.0
    U_e: float = 0.0
    n_nodes: int = 501
    L_factor: float = 7.0
    ic: tuple = (1e-8, 1e-8)
    x_boundary: float = -1.0
    x_end: float = 1.0

    @property
    def rho_e(self):
        return self.rho_i0 * (self.c_i0 ** 2 + GAMMA * 0.5 * self.vA_i0 ** 2) / (self.c_e ** 2 + GAMMA * 0.5 * self.vA_e ** 2)

    @property
    def cT_e(self):
        return math.sqrt(self.c_e ** 2 * self.vA_e ** 2 / (self.c_e ** 2 + self.vA_e ** 2))

    @property
    def cT_i0(self):
        return math.sqrt(self.c_i0 ** 2 * self.vA_i0 ** 2 / (self.c_i0 ** 2 + self.vA_i0 ** 2))


@dataclass
class SlabDensity(_SlabBase):
    """Gaussian density, vA_i = vA_i0 sqrt(rho_i0/rho), c_i from pressure balance (SD-P:102-155)."""
    width: float = 1e5
    x0: float = 0.0

    def profiles(self):
        x = sample_points(self.x_boundary, self.x_end, self.n_nodes)
        rho = self.rho_e + (self.rho_i0 - self.rho_e) * np.exp(-(x - self.x0) ** 2 / self.width ** 2)
        vA2 = self.vA_i0 ** 2 * self.rho_i0 / rho
        c2 = self.rho_e * (self.c_e ** 2 + 0.5 * GAMMA * self.vA_e ** 2) / rho - 0.5 * GAMMA * vA2
        return dict(rho=rho, c2=c2, vA2=vA2)


@dataclass
class SlabFlow(_SlabBase):
    """Uniform slab with flow U(x) = U_e + (U_i0 - U_e) exp(-(x-x0)^2/dx^2) (SF-G:124-126); width=inf: SF-U."""
    c_i0: float = 0.3
    vA_i0: float = 1.0
    c_e: float = 0.2
    vA_e: float = 2.5              # coronal (SF-G:63-67)
    U_i0: float = 0.9
    width: float = 1e5
    x0: float = 0.0
    n_nodes: int = 500
    L_factor: float = 3.0
    ic: tuple = (1e-8, 1e-15)

    def profiles(self):
        x = sample_points(self.x_boundary, self.x_end, self.n_nodes)
        if math.isinf(self.width):
            return dict(U=np.full_like(x, self.U_i0), dU=np.zeros_like(x), ddU=np.zeros_like(x))
        g = np.exp(-(x - self.x0) ** 2 / self.width ** 2)
        amp = self.U_i0 - self.U_e
        U = self.U_e + amp * g
        dU = amp * g * (-2.0 * (x - self.x0) / self.width ** 2)
        ddU = amp * g * (4.0 * (x - self.x0) ** 2 / self.width ** 4 - 2.0 / self.width ** 2)
        return dict(U=U, dU=dU, ddU=ddU)
