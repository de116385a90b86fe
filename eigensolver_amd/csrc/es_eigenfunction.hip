// Eigenfunctions at given (k, omega): what the reference's analysis scripts recompute at a chosen root
// (analysis_cylinder_flow_coronal.py:813-924).  One (k, omega) pair per lane for the interior (adjoint march to get
// the boundary flux from the far-end condition, then a forward RK4 march that writes the state at every node);
// one (pair, exterior point) per lane for the closed-form exterior.
#include "es_shoot_shared.hpp"

namespace {
using namespace es_shoot_shared;

// forward RK4 step of one state vector (u, v):  y' = A y
template <bool DIAG>
__device__ __forceinline__ void rk4_step_forward(double& u, double& v, const Coef& A0, const Coef& Am, const Coef& A1,
                                                 double h, double h2, double h6, double h3) {
#define ES_RHS_F(A, uu, vv, ku, kv)                                              \
  if (DIAG) { ku = fma(A.a11, uu, A.a12 * vv); kv = fma(A.a22, vv, A.a21 * uu); } \
  else      { ku = A.a12 * vv;                 kv = A.a21 * uu; }
  double k1u, k1v, k2u, k2v, k3u, k3v, k4u, k4v, tu, tv;
  ES_RHS_F(A0, u, v, k1u, k1v);
  tu = fma(h2, k1u, u); tv = fma(h2, k1v, v);
  ES_RHS_F(Am, tu, tv, k2u, k2v);
  tu = fma(h2, k2u, u); tv = fma(h2, k2v, v);
  ES_RHS_F(Am, tu, tv, k3u, k3v);
  tu = fma(h, k3u, u); tv = fma(h, k3v, v);
  ES_RHS_F(A1, tu, tv, k4u, k4v);
  u = fma(h6, k1u + k4u, fma(h3, k2u + k3u, u));
  v = fma(h6, k1v + k4v, fma(h3, k2v + k3v, v));
#undef ES_RHS_F
}

template <int FAM>
__global__ __launch_bounds__(64) void eigen_interior_kernel(ShootDev P, const double* __restrict__ kv,
                                                            const double* __restrict__ wv, int n,
                                                            double* __restrict__ val, double* __restrict__ flux) {
  constexpr int NE = FamTraits<FAM>::NE;
  constexpr int NB = FamTraits<FAM>::NB;
  constexpr bool DIAG = FamTraits<FAM>::DIAG;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = i < n;
  const double k = in ? kv[i] : 1.0;
  const double w = in ? wv[i] : 1.0;
  const KScal s = make_kscal(P, k);
  const int nsteps = P.n_nodes - 1;
  const double h = P.h, h2 = 0.5 * P.h, h6 = P.h / 6.0, h3 = P.h / 3.0;
  SignTrack trk;
  double b[NB], e[NE], e2[NE];
  const ExteriorLite X = exterior_lite(P, k, w, w);
  // (1) adjoint march: the row of the transfer matrix picked by the far-end condition -> boundary state (u_b, v_b)
  load_base<FAM>(P, 2 * nsteps, b);
  make_entry<FAM, fam_scaled<FAM>()>(b, s, e);
  Coef B0;
  coefficients<FAM>(e, P, s, w, B0, trk);
  double zp, zq;
  adjoint_start(P, B0, zp, zq);
  for (int j = nsteps - 1; j >= 0; --j) {
    Coef Bm, B1;
    load_base<FAM>(P, 2 * j + 1, b);
    make_entry<FAM, fam_scaled<FAM>()>(b, s, e);
    load_base<FAM>(P, 2 * j, b);
    make_entry<FAM, fam_scaled<FAM>()>(b, s, e2);
    coefficients2<FAM>(e, e2, P, s, w, Bm, B1, trk);
    adjoint_step_normalised<FAM>(zp, zq, B0, Bm, B1, h, h2, h6, h3);      // z at its true scale
    B0 = B1;
  }
  // boundary state from the same algebra as the determinant
  double ub, vb, flux_scale;     // flux = v * flux_scale(node) ; see below
  if (FAM == FAM_CYL0 || FAM == FAM_CYLT) {
    ub = X.yb;
    const double xi_e = X.outer;
    if (P.axis_bc == ES_AXIS_KINK) vb = (P.bc_const_raw * xi_e - zp * ub) / zq;
    else if (P.axis_bc == ES_AXIS_ROTATION_KINK) vb = (-(P.bc_const_raw * xi_e) - zp * ub) / zq;
    else vb = -(zp * ub) / zq;
    flux_scale = 1.0;            // xi = Xi / r, applied per node
  } else if (FAM == FAM_SLABD) {
    ub = X.yb;
    vb = (P.slab_sign - zp) * ub / zq;       // v = F Vx' ; P_T = v / w
    flux_scale = 1.0 / w;
  } else {
    const double Omb = w - e2[0];
    ub = X.yb * Omb / X.Oe;
    vb = (P.slab_sign - zp) * ub / zq;       // v = Vx' ; P_T = P_Ti(x) v
    flux_scale = 1.0;
  }
  // (2) forward march from the boundary, writing every node
  double u = ub, v = vb;
  load_base<FAM>(P, 0, b);
  make_entry<FAM>(b, s, e);
  Coef A0;
  coefficients<FAM>(e, P, s, w, A0, trk);
  auto store = [&](int node, const double* en) {
    if (!in) return;
    const size_t o = (size_t)i * P.n_nodes + node;
    val[o] = u;
    if (FAM == FAM_CYL0 || FAM == FAM_CYLT) {
      const double x = P.xb + (double)node * P.h;
      flux[o] = v / x;                                            // xi_r = Xi / r
    } else if (FAM == FAM_SLABD) {
      flux[o] = v * flux_scale;
    } else {
      const double Om = w - en[0];
      const double Om2 = Om * Om;
      flux[o] = P.rho_i * P.S_i * (s.kcT2 - Om2) / (Om * (s.kc2 - Om2)) * v;    // P_Ti(x) Vx', SF-G:433
    }
  };
  store(0, e);
  for (int j = 0; j < nsteps; ++j) {
    Coef Am, A1;
    load_base<FAM>(P, 2 * j + 1, b);
    make_entry<FAM>(b, s, e);
    load_base<FAM>(P, 2 * j + 2, b);
    make_entry<FAM>(b, s, e2);
    coefficients2<FAM>(e, e2, P, s, w, Am, A1, trk);
    rk4_step_forward<DIAG>(u, v, A0, Am, A1, h, h2, h6, h3);
    A0 = A1;
    store(j + 1, e2);
  }
}

// exterior: cylinders P = a I_m + b K_m, slabs Vx = a e^{mu x} + b e^{-mu x}; relative to the boundary value (+-1)
__global__ __launch_bounds__(256) void eigen_exterior_kernel(ShootDev P, const double* __restrict__ kv,
                                                             const double* __restrict__ wv, int n, int n_ext,
                                                             double* __restrict__ xs, double* __restrict__ val,
                                                             double* __restrict__ flux) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)n * n_ext) return;
  const int i = (int)(t / n_ext), j = (int)(t - (long)i * n_ext);
  const double k = kv[i], w = wv[i];
  const bool cyl = (P.family == FAM_CYL0 || P.family == FAM_CYLT);
  const Exterior X = cyl ? exterior_cylinder(P, k, w, w) : exterior_slab(P, k, w);
  const double sgn = (P.xb < 0.0) ? -1.0 : 1.0;
  const double R = P.R_factor / k;
  // np.linspace(sgn*R, sgn*1, n_ext)[j]
  const double step = (sgn * 1.0 - sgn * R) / (double)(n_ext - 1);
  const double x = (j == n_ext - 1) ? sgn * 1.0 : sgn * R + (double)j * step;
  double v = NAN, f = NAN;
  if (X.status == ES_PT_OK) {
    const double mu = sqrt(X.m_e);
    const double ax = fabs(x);
    if (cyl) {
      const int m = P.m_ext;
      const double dn = (double)m;
      const double xR = mu * R, xb = mu, xx = mu * ax;
      double KR, KR1, Kb, Kb1, Kx, Kx1, IR, IR1, Ib, Ib1, Ix, Ix1;
      esb::ke_pair(m, xR, KR, KR1); esb::ke_pair(m, xb, Kb, Kb1); esb::ke_pair(m, xx, Kx, Kx1);
      const double dKR = -KR1 + (dn / xR) * KR, dKx = -Kx1 + (dn / xx) * Kx;
      const double g = P.ic1 / (sgn * mu);
      double a_s = 0.0, b_s;
      double Ibv = 0.0, Ixv = 0.0, dIxv = 0.0;
      if (xR - xb < 40.0) {
        esb::ie_pair(m, xR, IR, IR1); esb::ie_pair(m, xb, Ib, Ib1); esb::ie_pair(m, xx, Ix, Ix1);
        const double dIR = IR1 + (dn / xR) * IR;
        a_s = -(P.ic0 * dKR - g * KR);
        b_s = -(g * IR - P.ic0 * dIR);
        Ibv = Ib; Ixv = Ix; dIxv = Ix1 + (dn / xx) * Ix;
      } else {
        const double rI = 1.0 - 0.5 / xR - (4.0 * dn * dn - 1.0) / (8.0 * xR * xR);
        b_s = -(g - P.ic0 * rI);
      }
      const double E2b = exp(-2.0 * (xR - xb)), E2x = exp(-2.0 * (xR - xx));
      const double den = fabs(b_s * Kb + E2b * a_s * Ibv);
      const double dec = exp(-(xx - xb));
      v = dec * (b_s * Kx + E2x * a_s * Ixv) / den;
      const double dv = sgn * mu * dec * (b_s * dKx + E2x * a_s * dIxv) / den;
      f = X.cst * dv;                                          // xi_e = xi_e_const * P'
    } else {
      const double E2x = exp(-2.0 * mu * (R - ax)), E2b = exp(-2.0 * mu * (R - 1.0));
      const double gp = P.ic0 + P.ic1 / mu, gm = P.ic0 - P.ic1 / mu;
      const double den = fabs(gp + E2b * gm);
      const double dec = exp(-mu * (ax - 1.0));
      v = dec * (gp + E2x * gm) / den;
      const double dv = mu * dec * (gp - E2x * gm) / den;
      f = X.cst * dv;                                          // left_P = p_e_const * Vx'
    }
  }
  xs[t] = x;
  val[t] = v;
  flux[t] = f;
}

template <int FAM>
int launch_interior(es_context* ctx, const es_problem* prob, const double* d_k, const double* d_w, int n, double* v,
                    double* f) {
  hipLaunchKernelGGL((eigen_interior_kernel<FAM>), dim3((n + 63) / 64), dim3(64), 0, ctx->stream, prob->dev, d_k, d_w,
                     n, v, f);
  ES_HIP_CHECK(ctx, hipGetLastError());
  return ES_SUCCESS;
}

}  // namespace

extern "C" int es_shoot_eigenfunction(es_context* ctx, const es_problem* prob, const double* d_k, const double* d_w,
                                      int n, double* d_int_value, double* d_int_flux, int n_ext, double* d_ext_x,
                                      double* d_ext_value, double* d_ext_flux) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  ES_REQUIRE(ctx, prob != nullptr, "null problem");
  ES_REQUIRE(ctx, n >= 0 && n_ext >= 0, "negative size");
  ES_REQUIRE(ctx, n_ext == 0 || n_ext >= 2, "n_ext must be 0 or >= 2");
  if (n == 0) return ES_SUCCESS;
  ES_REQUIRE(ctx, d_k && d_w && d_int_value && d_int_flux, "null pointer");
  ES_REQUIRE(ctx, n_ext == 0 || (d_ext_x && d_ext_value && d_ext_flux), "null exterior arrays");
  ES_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int rc;
  switch (prob->dev.family) {
    case FAM_CYL0: rc = launch_interior<FAM_CYL0>(ctx, prob, d_k, d_w, n, d_int_value, d_int_flux); break;
    case FAM_CYLT: rc = launch_interior<FAM_CYLT>(ctx, prob, d_k, d_w, n, d_int_value, d_int_flux); break;
    case FAM_SLABD: rc = launch_interior<FAM_SLABD>(ctx, prob, d_k, d_w, n, d_int_value, d_int_flux); break;
    case FAM_SLABF: rc = launch_interior<FAM_SLABF>(ctx, prob, d_k, d_w, n, d_int_value, d_int_flux); break;
    default: rc = ES_ERR_UNSUPPORTED;
  }
  if (rc) return rc;
  if (n_ext > 0) {
    const long tot = (long)n * n_ext;
    hipLaunchKernelGGL(eigen_exterior_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream,
                       prob->dev, d_k, d_w, n, n_ext, d_ext_x, d_ext_value, d_ext_flux);
    ES_HIP_CHECK(ctx, hipGetLastError());
  }
  return ES_SUCCESS;
}
