// K2: uniform cylinder in closed form -- the determinant D(k, omega; m) from Bessel functions, one grid point per lane.
//
// Restates, for the uniform limit the reference uses as its benchmark case (profile width 1e5, CF:126, CD-C:125 with
// dr = 1e5), what the workers compute numerically (CF:694-804):
//   interior  P'' + P'/r - (m_i + m^2/r^2) P = 0 ,  m_i = (k^2 vA^2 - Om^2)(k^2 c^2 - Om^2)/((c^2+vA^2)(k^2 cT^2 - Om^2)),
//             Om = omega - k U_i        (the general coefficient set of CF:577-626 reduces to this)
//             -> P = f1(kappa |r|) + beta f2(kappa |r|),  (f1, f2) = (I_m, K_m) for m_i > 0, (J_m, Y_m) for m_i < 0,
//             beta from the axis condition at r_axis: P = 0 (kink, CF:795 with B_phi = 0) or P' = 0 (sausage, CF:1092)
//   xi_i = P' / (rho_i (Om^2 - k^2 vA^2))   (xi = (C1 P + D P')/C3 with C1 = 0, C3 = D rho (Om^2 - wA^2), CF:798)
//   exterior and mismatch exactly as in the shooting path (exterior_cylinder).
#include "es_common.hpp"
#include "es_shoot_device.hpp"

namespace {

struct UniDev {
  double c2, vA2, rho_i, U_i, S, cT2;
  double r_sign, r_axis;
  int axis_bc;
};

__device__ __forceinline__ double pick_w_u(const double* __restrict__ wv, int w_mode, double k, int row, int nw, int iw) {
  if (w_mode == ES_W_PHASE_SPEED) return k * wv[iw];
  if (w_mode == ES_W_PER_ROW) return wv[(size_t)row * nw + iw];
  return wv[iw];
}

// d ln P / d|r| at |r| = 1 of the interior solution that satisfies the axis condition at r_axis.
__device__ __forceinline__ double interior_logder(const UniDev& U, int m, double m_i, bool& singular) {
  const double dm = (double)m;
  singular = false;
  if (m_i > 0.0) {
    const double kap = sqrt(m_i), xa = kap * U.r_axis, xb = kap;
    double Ib, Ib1, Kb, Kb1, Ia, Ia1, Ka, Ka1;
    esb::ke_pair(m, xb, Kb, Kb1);  esb::ie_pair_from_k(m, xb, Kb, Kb1, Ib, Ib1);
    esb::ke_pair(m, xa, Ka, Ka1);  esb::ie_pair_from_k(m, xa, Ka, Ka1, Ia, Ia1);
    const double dIb = Ib1 + (dm / xb) * Ib, dKb = -Kb1 + (dm / xb) * Kb;      // scaled derivatives
    const double dIa = Ia1 + (dm / xa) * Ia, dKa = -Ka1 + (dm / xa) * Ka;
    // P = I + beta K ; in scaled form beta K(xb)/I(xb) carries exp(-2 (xb - xa))
    const double E2 = exp(-2.0 * (xb - xa));
    const double g = (U.axis_bc == ES_AXIS_SAUSAGE) ? -(dIa / dKa) : -(Ia / Ka);
    const double num = dIb + E2 * g * dKb;
    const double den = Ib + E2 * g * Kb;
    return kap * num / den;
  } else if (m_i < 0.0) {
    const double kap = sqrt(-m_i), xa = kap * U.r_axis, xb = kap;
    double Jb, Jb1, Yb, Yb1, Ja, Ja1, Ya, Ya1;
    esb::jy_pair(m, xb, Jb, Jb1, Yb, Yb1);
    esb::jy_pair(m, xa, Ja, Ja1, Ya, Ya1);
    const double dJb = -Jb1 + (dm / xb) * Jb, dYb = -Yb1 + (dm / xb) * Yb;
    const double dJa = -Ja1 + (dm / xa) * Ja, dYa = -Ya1 + (dm / xa) * Ya;
    const double g = (U.axis_bc == ES_AXIS_SAUSAGE) ? -(dJa / dYa) : -(Ja / Ya);
    return kap * (dJb + g * dYb) / (Jb + g * Yb);
  }
  singular = true;
  return NAN;
}

__global__ __launch_bounds__(256) void cyl_uniform_kernel(ShootDev P, UniDev U, const double* __restrict__ kv, int nk,
                                                          const double* __restrict__ wv, int nw, int w_mode,
                                                          double* __restrict__ Dout, double* __restrict__ relout,
                                                          uint8_t* __restrict__ stout) {
  const int iw = blockIdx.x * 256 + threadIdx.x;
  for (int row = blockIdx.y; row < nk; row += gridDim.y) {
    if (iw >= nw) continue;
    const double k = kv[row];
    const double w = pick_w_u(wv, w_mode, k, row, nw, iw);
    const Exterior X = exterior_cylinder(P, k, w, w);
    const double k2 = k * k;
    const double Om = w - k * U.U_i;
    const double Om2 = Om * Om;
    const double m_i = ((k2 * U.vA2 - Om2) * (k2 * U.c2 - Om2)) / (U.S * (k2 * U.cT2 - Om2));
    uint8_t st = (uint8_t)X.status;
    double D = NAN, rel = NAN;
    if (X.status == ES_PT_OK) {
      bool sing;
      const double ld = interior_logder(U, P.m, m_i, sing);          // d ln P / d|r|
      const double Pb = X.yb;
      const double dPdr = U.r_sign * ld * Pb;                        // dP/dr in the signed coordinate
      const double xi_i = dPdr / (U.rho_i * (Om2 - k2 * U.vA2));
      const double xi_e = X.cst * X.dyb;
      D = xi_e - xi_i;
      rel = fabs(D) * 100.0 / fmax(fabs(xi_e), fabs(xi_i));
      if (sing || !isfinite(D)) { st = ES_PT_NONFINITE; }
    }
    const size_t o = (size_t)row * nw + iw;
    Dout[o] = D;
    stout[o] = st;
    if (relout) relout[o] = rel;
  }
}

}  // namespace

extern "C" int es_cyl_uniform_eval(es_context* ctx, const es_cyl_uniform_params* p, const double* d_k, int nk,
                                   const double* d_w, int nw, int w_mode, double* d_D, double* d_rel,
                                   uint8_t* d_status) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  ES_REQUIRE(ctx, p != nullptr, "null params");
  ES_REQUIRE(ctx, nk >= 0 && nw >= 0, "negative size");
  ES_REQUIRE(ctx, w_mode >= 0 && w_mode <= 2, "w_mode");
  ES_REQUIRE(ctx, p->r_boundary == -1.0 || p->r_boundary == 1.0, "r_boundary must be -1 or +1");
  ES_REQUIRE(ctx, p->r_axis > 0.0 && p->r_axis < 1.0, "r_axis");
  ES_REQUIRE(ctx, p->m >= 0 && p->m <= 64 && p->m_ext >= 0 && p->m_ext <= 64, "m");
  ES_REQUIRE(ctx, p->axis_bc == ES_AXIS_KINK || p->axis_bc == ES_AXIS_SAUSAGE, "axis_bc");
  if (nk == 0 || nw == 0) return ES_SUCCESS;
  ES_REQUIRE(ctx, d_k && d_w && d_D && d_status, "null pointer");
  ES_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ShootDev S;
  memset(&S, 0, sizeof(S));
  S.family = FAM_CYL0;
  S.xb = p->r_boundary;
  S.rho_e = p->rho_e; S.vAe2 = p->vA_e * p->vA_e; S.ce2 = p->c_e * p->c_e; S.cTe2 = p->cT_e * p->cT_e;
  S.Se = S.vAe2 + S.ce2;
  S.R_factor = p->L_factor * 2.0 * 3.14159265358979323846;
  S.ic0 = p->ic_value; S.ic1 = p->ic_slope;
  S.m = p->m; S.m_ext = p->m_ext; S.axis_bc = p->axis_bc;
  UniDev U;
  U.c2 = p->c_i * p->c_i; U.vA2 = p->vA_i * p->vA_i; U.rho_i = p->rho_i; U.U_i = p->U_i;
  U.S = U.c2 + U.vA2; U.cT2 = U.c2 * U.vA2 / U.S;
  U.r_sign = p->r_boundary; U.r_axis = p->r_axis; U.axis_bc = p->axis_bc;
  dim3 grid((nw + 255) / 256, nk < 65535 ? nk : 65535), block(256);
  hipLaunchKernelGGL(cyl_uniform_kernel, grid, block, 0, ctx->stream, S, U, d_k, nk, d_w, nw, w_mode, d_D, d_rel, d_status);
  ES_HIP_CHECK(ctx, hipGetLastError());
  return ES_SUCCESS;
}
