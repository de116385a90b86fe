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

// Deliberately NOT inlined: the Bessel series / continued fraction need ~100 VGPRs of their own; as a real call
// they stay out of the register allocation of the RK4 loop (the call sits after the loop, once per point).
static __device__ __noinline__ Exterior exterior_any(const ShootDev& P, double k, double w) {
  return (P.family == FAM_CYL0 || P.family == FAM_CYLT) ? exterior_cylinder(P, k, w) : exterior_slab(P, k, w);
}

__device__ __forceinline__ void finish_point(const ShootDev& P, const Mismatch& M, const Exterior& X, bool crossed,
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
template <int FAM>
__device__ __forceinline__ void shoot_point(const ShootDev& P, double k, double w, double& D, double& rel,
                                            uint8_t& st) {
  constexpr int NE = FamTraits<FAM>::NE;
  constexpr int NB = FamTraits<FAM>::NB;
  constexpr bool DIAG = FamTraits<FAM>::DIAG;
  const KScal s = make_kscal(P, k);
  const int nsteps = P.n_nodes - 1;
  const double h = P.h, h2 = 0.5 * P.h, h6 = P.h / 6.0, h3 = P.h / 3.0;
  SignTrack trk;
  double b[NB], e[NE], e2[NE];
  // adjoint march from the last node back to the boundary (same arithmetic as the grid kernel)
  load_base<FAM>(P, 2 * nsteps, b);
  make_entry<FAM>(b, s, e);
  Coef B0;
  coefficients<FAM>(e, P, s, w, B0, trk);
  double zp, zq;
  adjoint_start(P, B0, zp, zq);
  for (int j = nsteps - 1; j >= 0; --j) {
    Coef Bm, B1;
    load_base<FAM>(P, 2 * j + 1, b);
    make_entry<FAM>(b, s, e);
    load_base<FAM>(P, 2 * j, b);
    make_entry<FAM>(b, s, e2);
    coefficients2<FAM>(e, e2, P, s, w, Bm, B1, trk);
    rk4_step_adjoint<DIAG>(zp, zq, B0, Bm, B1, h, h2, h6, h3);
    B0 = B1;
  }
  const Exterior X = exterior_any(P, k, w);
  const Mismatch M = boundary_algebra<FAM>(P, s, w, X, zp, zq, e2);
  finish_point(P, M, X, trk.crossed(), D, rel, st);
}

}  // namespace es_shoot_shared
