// K6: complex-frequency determinant of the flow slab, winding-number cell detection and complex secant refinement.
// See include/eigensolver_amd.h section (6) for the reference lines this replaces and oracle/slab_complex.py for the
// CPU restatement (same algorithm: closed-form exterior, adjoint RK4 on the reference's ix grid with mid-point
// coefficient sets, far-end condition by superposition).
//
// One (k, omega) point per lane.  The profile table (U, U', U'' at nodes and mid-points) is staged in LDS chunk by
// chunk as in shoot_point; every lane forms its own complex coefficients.  This path is a handful of complex
// divisions per node and is not on the benchmark path: written for clarity, IEEE divisions throughout.
#include "es_shoot_shared.hpp"

namespace {
using namespace es_shoot_shared;

struct cx {
  double re, im;
};
__device__ __forceinline__ cx mk(double a, double b = 0.0) { return cx{a, b}; }
__device__ __forceinline__ cx operator+(cx a, cx b) { return cx{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cx operator-(cx a, cx b) { return cx{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cx operator-(cx a) { return cx{-a.re, -a.im}; }
__device__ __forceinline__ cx operator*(cx a, cx b) { return cx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cx operator*(double s, cx a) { return cx{s * a.re, s * a.im}; }
__device__ __forceinline__ cx operator/(cx a, cx b) {
  const double n = b.re * b.re + b.im * b.im;
  return cx{(a.re * b.re + a.im * b.im) / n, (a.im * b.re - a.re * b.im) / n};
}
__device__ __forceinline__ cx operator-(double s, cx a) { return cx{s - a.re, -a.im}; }
__device__ __forceinline__ cx operator+(double s, cx a) { return cx{s + a.re, a.im}; }
__device__ __forceinline__ double cabs2(cx a) { return a.re * a.re + a.im * a.im; }
__device__ __forceinline__ double cabs_(cx a) { return hypot(a.re, a.im); }
__device__ __forceinline__ bool cfinite(cx a) { return isfinite(a.re) && isfinite(a.im); }
// principal square root (Re >= 0)
__device__ __forceinline__ cx csqrt_(cx z) {
  const double r = hypot(z.re, z.im);
  if (r == 0.0) return cx{0.0, 0.0};
  double a = sqrt(0.5 * (r + fabs(z.re)));
  double b = 0.5 * z.im / a;
  if (z.re >= 0.0) return cx{a, b};
  return cx{fabs(b), copysign(a, z.im)};
}
__device__ __forceinline__ cx cexp_(cx z) {
  const double e = exp(z.re);
  double s, c;
  sincos(z.im, &s, &c);
  return cx{e * c, e * s};
}

// interior coefficients at one node: u' = v, v' = a21 u + a22 v  with  a21 = -coeff, a22 = -D
struct CxCoef { cx a21, a22; };

struct CxConsts {
  double k, k2, kc2, kvA2, kcT2, k4c, S_i;
  int variant;
};

__device__ __forceinline__ void cx_terms(const CxConsts& C, cx w, double U, double dU, cx& Om, cx& Om2, cx& m0, cx& D) {
  Om = w - mk(C.k * U);
  Om2 = Om * Om;
  const cx n1 = C.kc2 - Om2, n3 = C.kvA2 - Om2, nT = C.kcT2 - Om2;
  m0 = (n1 * n3) / (C.S_i * nT);                                                     // SF-X:375
  const double kdU2 = 2.0 * C.k * dU;
  if (C.variant == ES_CX_SFX) {
    D = kdU2 * ((Om2 / (Om2 - mk(C.kc2)) - mk(C.kcT2) / (Om2 - mk(C.kcT2))) / Om);   // SF-X:382
  } else {
    const cx t = Om2 - mk(C.kcT2);
    D = kdU2 * ((t + mk(C.k4c) / (C.S_i * t)) / (Om * (Om2 - mk(C.kc2))));          // SF-G:421 as written
  }
}

__device__ __forceinline__ CxCoef cx_coef(const CxConsts& C, cx w, double U, double dU, double ddU) {
  cx Om, Om2, m0, D;
  cx_terms(C, w, U, dU, Om, Om2, m0, D);
  const cx coeff = mk(C.k * ddU) / Om + (C.k * dU) * (D / Om) - m0;                   // SF-X:389
  return CxCoef{-coeff, -D};
}

// rhs of the transposed system: A^T z with A = [[0, 1], [a21, a22]]
__device__ __forceinline__ void cx_rhs(const CxCoef& A, cx p, cx q, cx& kp, cx& kq) {
  kp = A.a21 * q;
  kq = p + A.a22 * q;
}

__device__ __forceinline__ void cx_shoot_point(const ShootDev& P, int variant, double k, cx w, cx& D, double& rel,
                                               uint8_t& st, double* __restrict__ sb) {
  constexpr int NB = 3;
  CxConsts C;
  C.k = k; C.k2 = k * k;
  C.kc2 = C.k2 * P.c2_i; C.kvA2 = C.k2 * P.vA2_i; C.kcT2 = C.k2 * P.cT2_i;
  C.k4c = C.k2 * C.k2 * P.cT2_i * P.c2_i;
  C.S_i = P.S_i; C.variant = variant;
  // exterior (closed form, decaying branch), SF-X:369-371, :419-426
  const cx Oe = w - mk(k * P.U_e);
  const cx Oe2 = Oe * Oe;
  const cx m_e = ((C.k2 * P.vAe2 - Oe2) * (C.k2 * P.ce2 - Oe2)) / (P.Se * (C.k2 * P.cTe2 - Oe2));
  const cx p_e = (P.rho_e * P.Se) * ((C.k2 * P.cTe2 - Oe2) / (Oe * (C.k2 * P.ce2 - Oe2)));
  int status = ES_PT_OK;
  if (m_e.re < 0.0) status = ES_PT_LEAKY;                                             // `if m_e.real < 0: pass`
  const cx mu = csqrt_(m_e);
  const double R = P.R_factor / k;
  const cx E2 = cexp_((-2.0 * (R - 1.0)) * mu);
  const cx gq = mk(P.ic1) / mu;
  const cx gp = P.ic0 + gq, gm = P.ic0 - gq;
  const cx y = mu * ((gp - E2 * gm) / (gp + E2 * gm));
  const cx outer = p_e * y;
  if (status == ES_PT_OK && !cfinite(outer)) status = ES_PT_NONFINITE;

  // adjoint march of the functional Vx(+1) = (1, 0) . (u, v) from x = +1 back to x = -1
  const int nsteps = P.n_nodes - 1;
  const double h = P.h, h2 = 0.5 * P.h, h6 = P.h / 6.0, h3 = P.h / 3.0;
  cx zp = mk(1.0), zq = mk(0.0);
  CxCoef B0{mk(0.0), mk(0.0)};
  double Ub = 0.0, dUb = 0.0;
  const int nchunks = (nsteps + CH - 1) / CH;
  for (int c = nchunks - 1; c >= 0; --c) {
    const int c0 = c * CH;
    const int nst = (nsteps - c0 < CH) ? (nsteps - c0) : CH;
    __syncthreads();
#pragma unroll
    for (int f = 0; f < NB; ++f)
      for (int i = threadIdx.x; i < 2 * nst + 1; i += blockDim.x) sb[i * NB + f] = P.base[(size_t)f * P.npts + 2 * c0 + i];
    __syncthreads();
    if (c == nchunks - 1) B0 = cx_coef(C, w, sb[2 * nst * NB + SF_U], sb[2 * nst * NB + SF_DU], sb[2 * nst * NB + SF_DDU]);
    for (int j = nst - 1; j >= 0; --j) {
      const double* bm = sb + (2 * j + 1) * NB;
      const double* b1 = sb + (2 * j) * NB;
      const CxCoef Bm = cx_coef(C, w, bm[SF_U], bm[SF_DU], bm[SF_DDU]);
      const CxCoef B1 = cx_coef(C, w, b1[SF_U], b1[SF_DU], b1[SF_DDU]);
      cx k1p, k1q, k2p, k2q, k3p, k3q, k4p, k4q;
      cx_rhs(B0, zp, zq, k1p, k1q);
      cx_rhs(Bm, zp + h2 * k1p, zq + h2 * k1q, k2p, k2q);
      cx_rhs(Bm, zp + h2 * k2p, zq + h2 * k2q, k3p, k3q);
      cx_rhs(B1, zp + h * k3p, zq + h * k3q, k4p, k4q);
      zp = zp + h6 * (k1p + k4p) + h3 * (k2p + k3p);
      zq = zq + h6 * (k1q + k4q) + h3 * (k2q + k3q);
      B0 = B1;
      if (c == 0 && j == 0) { Ub = b1[SF_U]; dUb = b1[SF_DU]; }
    }
  }
  // boundary: continuity of the displacement, symmetry condition by superposition, total pressures (SF-X:422, :455)
  cx Omb, Omb2, m0b, Db;
  cx_terms(C, w, Ub, dUb, Omb, Omb2, m0b, Db);
  const cx Vb = Omb / Oe;
  const cx sv = ((P.slab_sign - zp) * Vb) / zq;
  const cx PTi = (P.rho_i * P.S_i) * ((C.kcT2 - Omb2) / (Omb * (C.kc2 - Omb2)));      // SF-X:395
  const cx add = (variant == ES_CX_SFX) ? -(mk(k * dUb) / Omb) : mk(0.0);             // SF-X:401
  const cx inner = PTi * (sv - add * Vb);
  const cx d = outer - inner;
  st = (uint8_t)status;
  D = d;
  rel = cabs_(d) * 100.0 / fmax(cabs_(outer), cabs_(inner));
  if (status != ES_PT_OK) { D = cx{NAN, NAN}; rel = NAN; return; }
  if (!cfinite(d)) { st = ES_PT_NONFINITE; D = cx{NAN, NAN}; rel = NAN; }
}

__device__ __forceinline__ cx cx_pick_w(int w_mode, double k, double wre, double wim) {
  return (w_mode == ES_W_PHASE_SPEED) ? cx{k * wre, k * wim} : cx{wre, wim};
}

// grid (n_re > 0: index -> (row, i_im, i_re)) or point list (n_re == 0)
__global__ __launch_bounds__(256) void cx_eval_kernel(ShootDev P, int variant, const double* __restrict__ kv,
                                                      const double* __restrict__ wre, const double* __restrict__ wim,
                                                      long n, int n_re, int n_im, int w_mode,
                                                      double* __restrict__ Dre, double* __restrict__ Dim,
                                                      double* __restrict__ relout, uint8_t* __restrict__ stout) {
  __shared__ double sb[3 * (2 * CH + 1)];
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = i < n;
  double k = 1.0;
  cx w = cx{1.0, 0.1};
  if (in) {
    if (n_re > 0) {
      const long row = i / ((long)n_re * n_im);
      const long r = i - row * (long)n_re * n_im;
      const int iim = (int)(r / n_re), ire = (int)(r - (long)iim * n_re);
      k = kv[row];
      w = cx_pick_w(w_mode, k, wre[ire], wim[iim]);
    } else {
      k = kv[i];
      w = cx{wre[i], wim[i]};
    }
  }
  cx D; double rel; uint8_t st;
  cx_shoot_point(P, variant, k, w, D, rel, st, sb);
  if (in) {
    Dre[i] = D.re;
    Dim[i] = D.im;
    stout[i] = st;
    if (relout) relout[i] = rel;
  }
}

// ---- cells with a zero of D_c inside: winding number of D around the four corners ------------------------------
__device__ __forceinline__ int quadrant(double re, double im) { return (re >= 0.0) ? (im >= 0.0 ? 0 : 3) : (im >= 0.0 ? 1 : 2); }
// change of quadrant between consecutive corners as a signed quarter-turn count; +-2 is ambiguous -> reported as 8
__device__ __forceinline__ int quarter_turns(int qa, int qb) {
  const int d = (qb - qa) & 3;
  return d == 0 ? 0 : (d == 1 ? 1 : (d == 3 ? -1 : 8));
}

__global__ __launch_bounds__(256) void cx_flag_kernel(const double* __restrict__ Dre, const double* __restrict__ Dim,
                                                      const uint8_t* __restrict__ st, int n_re, int n_im, long cells,
                                                      uint64_t* __restrict__ masks, int* __restrict__ block_counts) {
  __shared__ int wave_cnt[4];
  const long c = (long)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  bool flag = false;
  if (c < cells) {
    const long per_row = (long)n_re * n_im;
    const long r = c % per_row;
    const int iim = (int)(r / n_re), ire = (int)(r - (long)iim * n_re);
    if (ire < n_re - 1 && iim < n_im - 1) {
      const long c00 = c, c10 = c + 1, c11 = c + n_re + 1, c01 = c + n_re;      // counter-clockwise in (re, im)
      if (st[c00] == ES_PT_OK && st[c10] == ES_PT_OK && st[c11] == ES_PT_OK && st[c01] == ES_PT_OK) {
        const int q0 = quadrant(Dre[c00], Dim[c00]), q1 = quadrant(Dre[c10], Dim[c10]);
        const int q2 = quadrant(Dre[c11], Dim[c11]), q3 = quadrant(Dre[c01], Dim[c01]);
        const int t0 = quarter_turns(q0, q1), t1 = quarter_turns(q1, q2), t2 = quarter_turns(q2, q3), t3 = quarter_turns(q3, q0);
        const int total = t0 + t1 + t2 + t3;
        // +-4: one full turn (a simple zero / pole inside); an ambiguous half-turn edge: refine and let the secant decide
        flag = (total == 4 || total == -4 || total >= 6);
      }
    }
  }
  const uint64_t m = __ballot(flag);
  if (lane == 0) {
    masks[c >> 6] = m;
    wave_cnt[threadIdx.x >> 6] = __popcll(m);
  }
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

__global__ __launch_bounds__(256) void cx_emit_kernel(const double* __restrict__ kv, const double* __restrict__ wre,
                                                      const double* __restrict__ wim, int n_re, int n_im, int w_mode,
                                                      long cells, const uint64_t* __restrict__ masks,
                                                      const int* __restrict__ block_off, es_complex_root_table tab,
                                                      double* __restrict__ half_re, double* __restrict__ half_im) {
  const long c = (long)blockIdx.x * 256 + threadIdx.x;
  if (c >= cells) return;
  if (!((masks[c >> 6] >> (c & 63)) & 1ull)) return;
  const int pos = es_cell_rank(masks, block_off, c);
  if (pos >= tab.capacity) return;
  const long per_row = (long)n_re * n_im;
  const long row = c / per_row;
  const long r = c - row * per_row;
  const int iim = (int)(r / n_re), ire = (int)(r - (long)iim * n_re);
  const double k = kv[row];
  const cx a = cx_pick_w(w_mode, k, wre[ire], wim[iim]);
  const cx b = cx_pick_w(w_mode, k, wre[ire + 1], wim[iim + 1]);
  tab.d_k[pos] = k;
  tab.d_row[pos] = (int32_t)row;
  tab.d_w_re[pos] = 0.5 * (a.re + b.re);                   // cell centre: start of the refinement
  tab.d_w_im[pos] = 0.5 * (a.im + b.im);
  half_re[pos] = 0.5 * (b.re - a.re);
  half_im[pos] = 0.5 * (b.im - a.im);
}

// complex secant iteration, one candidate per lane; uniform trip count (all lanes evaluate together)
__global__ __launch_bounds__(64) void cx_refine_kernel(ShootDev P, int variant, es_complex_root_table tab,
                                                       const double* __restrict__ half_re,
                                                       const double* __restrict__ half_im, int n, int n_iter,
                                                       double tol_percent) {
  __shared__ double sb[3 * (2 * CH + 1)];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = i < n;
  const double k = in ? tab.d_k[i] : 1.0;
  const cx centre = in ? cx{tab.d_w_re[i], tab.d_w_im[i]} : cx{1.0, 0.1};
  const cx half = in ? cx{half_re[i], half_im[i]} : cx{0.1, 0.1};
  cx w0 = centre, w1 = centre + cx{0.5 * half.re, 0.5 * half.im};
  cx f0, f1; double rel0, rel1; uint8_t s0, s1;
  cx_shoot_point(P, variant, k, w0, f0, rel0, s0, sb);
  cx_shoot_point(P, variant, k, w1, f1, rel1, s1, sb);
  for (int it = 0; it < n_iter; ++it) {
    const cx df = f1 - f0;
    cx w2 = w1 - f1 * ((w1 - w0) / df);
    if (!cfinite(w2) || cabs2(df) == 0.0) w2 = w1;          // converged (f1 == f0) or broken: stay
    w0 = w1; f0 = f1;
    w1 = w2;
    cx_shoot_point(P, variant, k, w1, f1, rel1, s1, sb);
    if (!cfinite(f1)) { w1 = w0; f1 = f0; }                 // stepped onto a leaky / singular point: back off
  }
  // rel of the final iterate
  cx_shoot_point(P, variant, k, w1, f1, rel1, s1, sb);
  if (in) {
    const double dist2 = cabs2(w1 - centre), diag2 = 4.0 * cabs2(half);
    tab.d_w_re[i] = w1.re;
    tab.d_w_im[i] = w1.im;
    tab.d_resid[i] = rel1;
    tab.d_flag[i] = (s1 == ES_PT_OK && rel1 < tol_percent && dist2 <= 4.0 * diag2) ? 1 : 0;
  }
}

int check_cx(es_context* ctx, const es_problem* prob, int variant) {
  ES_REQUIRE(ctx, prob != nullptr, "null problem");
  ES_REQUIRE(ctx, variant == ES_CX_SFX || variant == ES_CX_SFG, "variant");
  if (prob->dev.family != FAM_SLABF) {
    ctx->last_error = "complex frequencies are implemented for ES_GEOM_SLAB_FLOW problems only";
    return ES_ERR_UNSUPPORTED;
  }
  return ES_SUCCESS;
}

}  // namespace

extern "C" int es_complex_eval_grid(es_context* ctx, const es_problem* prob, int variant, const double* d_k, int nk,
                                    const double* d_w_re, int n_re, const double* d_w_im, int n_im, int w_mode,
                                    double* d_D_re, double* d_D_im, double* d_rel, uint8_t* d_status) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  int rc = check_cx(ctx, prob, variant);
  if (rc) return rc;
  ES_REQUIRE(ctx, nk >= 0 && n_re >= 0 && n_im >= 0, "negative size");
  ES_REQUIRE(ctx, w_mode == ES_W_ABSOLUTE || w_mode == ES_W_PHASE_SPEED, "w_mode");
  const long n = (long)nk * n_re * n_im;
  if (n == 0) return ES_SUCCESS;
  ES_REQUIRE(ctx, d_k && d_w_re && d_w_im && d_D_re && d_D_im && d_status, "null pointer");
  ES_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(cx_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, prob->dev, variant,
                     d_k, d_w_re, d_w_im, n, n_re, n_im, w_mode, d_D_re, d_D_im, d_rel, d_status);
  ES_HIP_CHECK(ctx, hipGetLastError());
  return ES_SUCCESS;
}

extern "C" int es_complex_eval_points(es_context* ctx, const es_problem* prob, int variant, const double* d_k,
                                      const double* d_w_re, const double* d_w_im, int n, double* d_D_re,
                                      double* d_D_im, double* d_rel, uint8_t* d_status) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  int rc = check_cx(ctx, prob, variant);
  if (rc) return rc;
  ES_REQUIRE(ctx, n >= 0, "negative size");
  if (n == 0) return ES_SUCCESS;
  ES_REQUIRE(ctx, d_k && d_w_re && d_w_im && d_D_re && d_D_im && d_status, "null pointer");
  ES_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(cx_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, prob->dev, variant,
                     d_k, d_w_re, d_w_im, (long)n, 0, 0, ES_W_ABSOLUTE, d_D_re, d_D_im, d_rel, d_status);
  ES_HIP_CHECK(ctx, hipGetLastError());
  return ES_SUCCESS;
}

extern "C" int es_complex_find_roots(es_context* ctx, const es_problem* prob, int variant, const double* d_k, int nk,
                                     const double* d_w_re, int n_re, const double* d_w_im, int n_im, int w_mode,
                                     const double* d_D_re, const double* d_D_im, const uint8_t* d_status, int n_iter,
                                     double tol_percent, es_complex_root_table* table, int* out_count) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  int rc = check_cx(ctx, prob, variant);
  if (rc) return rc;
  ES_REQUIRE(ctx, table && out_count, "null pointer");
  ES_REQUIRE(ctx, nk >= 0 && n_re >= 0 && n_im >= 0 && n_iter >= 0 && n_iter <= 1000 && table->capacity >= 0, "size");
  ES_REQUIRE(ctx, w_mode == ES_W_ABSOLUTE || w_mode == ES_W_PHASE_SPEED, "w_mode");
  *out_count = 0;
  const long cells = (long)nk * n_re * n_im;
  if (cells == 0) return ES_SUCCESS;
  ES_REQUIRE(ctx, d_k && d_w_re && d_w_im && d_D_re && d_D_im && d_status, "null pointer");
  ES_REQUIRE(ctx, table->capacity == 0 || (table->d_k && table->d_w_re && table->d_w_im && table->d_resid &&
                                           table->d_row && table->d_flag), "null root table arrays");
  ES_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  rc = es_ensure_scan_scratch(ctx, (size_t)cells);
  if (rc) return rc;
  const int nblocks = (int)((cells + 255) / 256);
  hipLaunchKernelGGL(cx_flag_kernel, dim3(nblocks), dim3(256), 0, ctx->stream, d_D_re, d_D_im, d_status, n_re, n_im,
                     cells, ctx->d_masks, ctx->d_block_counts);
  ES_HIP_CHECK(ctx, hipGetLastError());
  int total = 0;
  rc = es_scan_block_counts(ctx, nblocks, &total);
  if (rc) return rc;
  *out_count = total;
  const int n = total < table->capacity ? total : table->capacity;
  if (n > 0) {
    rc = es_ensure_scratch(ctx, 2 * (size_t)n * sizeof(double));
    if (rc) return rc;
    double* half = (double*)ctx->d_scratch;
    hipLaunchKernelGGL(cx_emit_kernel, dim3(nblocks), dim3(256), 0, ctx->stream, d_k, d_w_re, d_w_im, n_re, n_im,
                       w_mode, cells, ctx->d_masks, ctx->d_block_counts, *table, half, half + n);
    hipLaunchKernelGGL(cx_refine_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, prob->dev, variant, *table,
                       half, half + n, n, n_iter, tol_percent);
    ES_HIP_CHECK(ctx, hipGetLastError());
    ES_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  return (total > table->capacity) ? ES_ERR_CAPACITY : ES_SUCCESS;
}
