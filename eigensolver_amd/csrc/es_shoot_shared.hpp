// Shared between es_shoot.hip and es_worker.hip: problem handle and the per-lane determinant evaluation.
#pragma once
#include "es_common.hpp"
#include "es_shoot_device.hpp"

struct es_problem {
  ShootDev dev;
  double* d_base = nullptr;
  es_shoot_desc desc;
};

namespace es_shoot_shared {

constexpr int ES_REFINE_POLISH = 2;   // regula-falsi steps after the 9-section rounds (0: report the bracket midpoint)
constexpr int CH = 128;   // RK4 steps per LDS chunk: (2*CH+1) * NE * 8 B of LDS (12.3 KiB for NE = 6)

template <int FAM>
__device__ __forceinline__ void load_base(const ShootDev& P, int pt, double* b) {
#pragma unroll
  for (int f = 0; f < FamTraits<FAM>::NB; ++f) b[f] = P.base[(size_t)f * P.npts + pt];
}

__device__ __forceinline__ double pick_w(const double* __restrict__ wv, int w_mode, double k, int row, int nw, int iw) {
  if (w_mode == ES_W_PHASE_SPEED) return k * wv[iw];
  if (w_mode == ES_W_PER_ROW) return wv[(size_t)row * nw + iw];
  return wv[iw];
}

// What the boundary algebra needs from the exterior: outer = cst * dyb (xi_e resp. P_left), the normalised boundary
// value yb (+-1), the Doppler-shifted exterior frequency (flow slab) and the status.
struct ExteriorLite {
  double outer, yb, Oe;
  int status;
};

// Evaluated BEFORE the RK4 march (few live registers), inlined: the Bessel series / continued fraction need ~100
// VGPRs of their own, which then overlap with nothing; only the three results are carried through the loop.
__device__ __forceinline__ ExteriorLite exterior_lite(const ShootDev& P, double k, double w, double w_cst) {
  const Exterior X = (P.family == FAM_CYL0 || P.family == FAM_CYLT) ? exterior_cylinder(P, k, w, w_cst) : exterior_slab(P, k, w);
  ExteriorLite L;
  L.outer = X.cst * X.dyb;
  L.yb = X.yb;
  L.Oe = X.Oe;
  L.status = X.status;
  return L;
}

__device__ __forceinline__ void finish_point(const ShootDev& P, const Mismatch& M, const ExteriorLite& X, bool crossed,
                                             double& D, double& rel, uint8_t& st) {
  st = (uint8_t)X.status;
  D = M.d;
  const double sc = P.accept_norm ? fabs(M.outer) : fmax(fabs(M.outer), fabs(M.inner));
  rel = fabs(M.d) * 100.0 / sc;                       // CF:817
  if (X.status != ES_PT_OK) { D = NAN; rel = NAN; return; }
  if (!isfinite(D)) { st = ES_PT_NONFINITE; return; }
  if (crossed) st = ES_PT_CONTINUUM;
}

// One (k, omega) pair per lane.  Base-table indices are wave-uniform -> scalar loads.
template <int FAM, bool TRACK>
__device__ __forceinline__ void shoot_point_impl(const ShootDev& P, double k, double w, double w_cst, double& D,
                                                 double& rel, uint8_t& st) {
  constexpr int NE = FamTraits<FAM>::NE;
  constexpr int NB = FamTraits<FAM>::NB;
  constexpr bool DIAG = FamTraits<FAM>::DIAG;
  const KScal s = make_kscal(P, k);
  const int nsteps = P.n_nodes - 1;
  const double h = P.h, h2 = 0.5 * P.h, h6 = P.h / 6.0, h3 = P.h / 3.0;
  SignTrack trk;
  double b[NB], e[NE], e2[NE];
  const ExteriorLite X = exterior_lite(P, k, w, w_cst);
  // adjoint march from the last node back to the boundary (same arithmetic as the grid kernel)
  load_base<FAM>(P, 2 * nsteps, b);
  make_entry<FAM>(b, s, e);
  Coef B0;
  coefficients<FAM, TRACK>(e, P, s, w, B0, trk);
  double zp, zq;
  adjoint_start(P, B0, zp, zq);
  for (int j = nsteps - 1; j >= 0; --j) {
    Coef Bm, B1;
    load_base<FAM>(P, 2 * j + 1, b);
    make_entry<FAM>(b, s, e);
    load_base<FAM>(P, 2 * j, b);
    make_entry<FAM>(b, s, e2);
    coefficients2<FAM, TRACK>(e, e2, P, s, w, Bm, B1, trk);
    rk4_step_adjoint<DIAG>(zp, zq, B0, Bm, B1, h, h2, h6, h3);
    B0 = B1;
  }
  const Mismatch M = boundary_algebra<FAM>(P, s, w, X, zp, zq, e2);
  finish_point(P, M, X, TRACK ? trk.crossed() : band_crossed(P, k, w), D, rel, st);
}

template <int FAM>
__device__ __forceinline__ void shoot_point(const ShootDev& P, double k, double w, double w_cst, double& D,
                                            double& rel, uint8_t& st) {
  if (FAM == FAM_CYL0 && P.use_bands) shoot_point_impl<FAM, FAM != FAM_CYL0>(P, k, w, w_cst, D, rel, st);   // wave-uniform branch
  else shoot_point_impl<FAM, true>(P, k, w, w_cst, D, rel, st);
}

}  // namespace es_shoot_shared
