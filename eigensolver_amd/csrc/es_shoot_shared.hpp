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
// 1 if chunk c_ of n_ stages the entry of its far node (index 2 nst): only the chunk that starts the march reads it -- the others
// get its coefficients in B0 from the chunk before -- and 2 nst entries are ONE pass of 256 threads (four of a 64-lane wave
// building 4 x 64) where 2 nst + 1 took one more pass for a single entry.  -DES_STAGE_FAR_NODE_ALWAYS: A/B build.
#if defined(ES_STAGE_FAR_NODE_ALWAYS)
#define ES_FAR_NODE(c_, n_) 1
#else
#define ES_FAR_NODE(c_, n_) (((c_) == (n_) - 1) ? 1 : 0)
#endif
constexpr int CH = 128;   // RK4 steps per LDS chunk: (2*CH+1) * NE * 8 B of LDS (14.4 KiB for NE = 7)

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

// One (k, omega) pair per lane, unrelated k per lane.  Must be called by ALL threads of the workgroup together (it
// contains barriers): the k-independent base table is staged chunk by chunk (CH RK4 steps) in the LDS buffer `sb`
// (FamTraits<FAM>::NB * (2 CH + 1) doubles, point-major) with coalesced loads, and every lane then reads the same
// LDS address per node (broadcast) and forms its own node entries.  Same arithmetic, in the same order, as the grid
// kernel.  (Reading the table with wave-uniform scalar loads instead left these latency-bound kernels -- one wave
// per SIMD or fewer -- waiting ~400 ns per RK4 step.)
template <int FAM, bool TRACK, bool BANDS = !TRACK, int C1P = 0>
__device__ __forceinline__ void shoot_point_impl(const ShootDev& P, double k, double w, double w_cst, double& D,
                                                 double& rel, uint8_t& st, double* __restrict__ sb) {
  constexpr int NE = FamTraits<FAM>::NE;
  constexpr int NB = FamTraits<FAM>::NB;
  const KScal s = make_kscal(P, k);
  const int nsteps = P.n_nodes - 1;
  const double h = P.h, h2 = 0.5 * P.h, h6 = P.h / 6.0, h3 = P.h / 3.0;
  SignTrack trk;
  double b[NB], e[NE], e2[NE];
  const ExteriorLite X = exterior_lite(P, k, w, w_cst);
  Coef B0;
  double zp = 0.0, zq = 0.0;
  // adjoint march from the last node back to the boundary, chunk by chunk
  const int nchunks = (nsteps + CH - 1) / CH;
  for (int c = nchunks - 1; c >= 0; --c) {
    const int c0 = c * CH;
    const int nst = (nsteps - c0 < CH) ? (nsteps - c0) : CH;
    __syncthreads();                                   // previous chunk fully consumed
#pragma unroll
    for (int f = 0; f < NB; ++f)
      for (int i = threadIdx.x; i < 2 * nst + ES_FAR_NODE(c, nchunks); i += blockDim.x)      // far node: first chunk only
        sb[i * NB + f] = P.base[(size_t)f * P.npts + 2 * c0 + i];
    __syncthreads();
    if (c == nchunks - 1) {                            // last node: start vector of the march
#pragma unroll
      for (int f = 0; f < NB; ++f) b[f] = sb[2 * nst * NB + f];
      make_entry<FAM, fam_scaled<FAM>()>(b, s, e);
      coefficients<FAM, TRACK, C1P>(e, P, s, w, B0, trk);
      adjoint_start(P, B0, zp, zq);
    }
    // The coefficients of a step (of a pair of steps: fam_rcp4) are formed one iteration AHEAD of the RK4 stages that use
    // them: they depend on the node and on omega only, not on the marched row, so a lone wave -- these kernels run one or
    // two per SIMD -- has the entries, the division and the products of the next step to issue while the dependent fma
    // chain of this step's stages waits for its results.  Same operations on the same values as the grid kernel.
#define ES_ENTRY(NODE, E)                                                                      \
    {                                                                                          \
      _Pragma("unroll") for (int f = 0; f < NB; ++f) b[f] = sb[(NODE) * NB + f];               \
      make_entry<FAM, fam_scaled<FAM>()>(b, s, E);                                             \
    }
    int j = nst - 1;
    if (fam_rcp4<FAM>()) {
      if (!(j & 1)) {                                  // even top step (odd number of steps): alone
        Coef Bm, B1;
        ES_ENTRY(2 * j + 1, e)
        ES_ENTRY(2 * j, e2)
        coefficients2<FAM, TRACK, C1P>(e, e2, P, s, w, Bm, B1, trk);
        adjoint_step<FAM>(zp, zq, B0, Bm, B1, h, h2, h6, h3);
        B0 = B1;
        --j;
      }
      if (j >= 1) {                                    // steps j and j - 1 with one division (coefficients4)
        double e1[NE], em2[NE];
        Coef Cm, C1, Cm2, C2;
        ES_ENTRY(2 * j + 1, e)
        ES_ENTRY(2 * j, e1)
        ES_ENTRY(2 * j - 1, em2)
        ES_ENTRY(2 * j - 2, e2)
        coefficients4<FAM, TRACK>(e, e1, em2, e2, P, s, w, Cm, C1, Cm2, C2, trk);
        for (; j >= 3; j -= 2) {
          Coef Nm, N1, Nm2, N2;
          ES_ENTRY(2 * j - 3, e)
          ES_ENTRY(2 * j - 4, e1)
          ES_ENTRY(2 * j - 5, em2)
          ES_ENTRY(2 * j - 6, e2)
          coefficients4<FAM, TRACK>(e, e1, em2, e2, P, s, w, Nm, N1, Nm2, N2, trk);
          adjoint_step<FAM>(zp, zq, B0, Cm, C1, h, h2, h6, h3);
          adjoint_step<FAM>(zp, zq, C1, Cm2, C2, h, h2, h6, h3);
          B0 = C2;
          Cm = Nm; C1 = N1; Cm2 = Nm2; C2 = N2;
        }
        adjoint_step<FAM>(zp, zq, B0, Cm, C1, h, h2, h6, h3);
        adjoint_step<FAM>(zp, zq, C1, Cm2, C2, h, h2, h6, h3);
        B0 = C2;
      }
    } else {
      Coef Cm, C1;
      ES_ENTRY(2 * j + 1, e)
      ES_ENTRY(2 * j, e2)
      coefficients2<FAM, TRACK, C1P>(e, e2, P, s, w, Cm, C1, trk);
      for (; j >= 1; --j) {
        Coef Nm, N1;
        ES_ENTRY(2 * j - 1, e)
        ES_ENTRY(2 * j - 2, e2)
        coefficients2<FAM, TRACK, C1P>(e, e2, P, s, w, Nm, N1, trk);
        adjoint_step<FAM>(zp, zq, B0, Cm, C1, h, h2, h6, h3);
        B0 = C1;
        Cm = Nm; C1 = N1;
      }
      adjoint_step<FAM>(zp, zq, B0, Cm, C1, h, h2, h6, h3);
      B0 = C1;
    }
#undef ES_ENTRY
    adjoint_rescale<FAM>(zp, zq, nsteps - c0 - nst, nsteps - c0);   // steps marched before / after this chunk
  }
  const Mismatch M = boundary_algebra<FAM>(P, s, w, X, zp, zq, e2);
  finish_point(P, M, X, TRACK ? trk.crossed() : (BANDS ? band_crossed(P, k, w) : false), D, rel, st);
}

// The 64 lanes of a WAVE fall into NG groups of 64 / NG lanes with the SAME k each (the section points of one bracket):
// the node entries -- what of a one-point march depends on (node, k) only, 40 % of its instructions -- are formed once
// per group and node by the wave itself into its own LDS table `tbl` ([NG][2 CHR + 1][NE] doubles per chunk of CHR
// steps) and read back by the lanes of the group (one address per group: broadcast).  No workgroup barrier: a wave only
// reads what it wrote (LDS operations of a wave complete in order; the wave barrier keeps the compiler from moving
// them across).  The k of every group is fetched by shuffles up front.  The entries are
// make_entry's of the per-lane form and the arithmetic after them is shoot_point_impl's: results bit-identical.
template <int FAM, bool TRACK, int CHR, int NG>
__device__ __forceinline__ void shoot_point_wavegroup_impl(const ShootDev& P, double k, double w, double& D, double& rel,
                                                           uint8_t& st, double* __restrict__ tbl) {
  constexpr int NE = FamTraits<FAM>::NE;
  constexpr int NB = FamTraits<FAM>::NB;
  constexpr int NODES = 2 * CHR + 1;
  constexpr int GL = 64 / NG;                          // lanes per group
  const int lane = threadIdx.x & 63;
  const int grp = lane / GL;
  const KScal s = make_kscal(P, k);
  const int nsteps = P.n_nodes - 1;
  const double h = P.h, h2 = 0.5 * P.h, h6 = P.h / 6.0, h3 = P.h / 3.0;
  SignTrack trk;
  double e[NE], e2[NE];
  const ExteriorLite X = exterior_lite(P, k, w, w);
  Coef B0;
  double zp = 0.0, zq = 0.0;
  const double* mine = tbl + (size_t)grp * NODES * NE;
  double kgs[NG];                                      // the k of every group, fetched while all lanes are active
#pragma unroll
  for (int g2 = 0; g2 < NG; ++g2) kgs[g2] = __shfl(k, g2 * GL);
  const int nchunks = (nsteps + CHR - 1) / CHR;
  for (int c = nchunks - 1; c >= 0; --c) {
    const int c0 = c * CHR;
    const int nst = (nsteps - c0 < CHR) ? (nsteps - c0) : CHR;
    const int nn = 2 * nst + ES_FAR_NODE(c, nchunks);          // the far node's entries: only the chunk that starts the march reads them
    __builtin_amdgcn_wave_barrier();                   // previous chunk consumed by every lane of this wave
    for (int idx = lane; idx < NG * nn; idx += 64) {
      const int g2 = idx / nn, i = idx - g2 * nn;
      double kg = kgs[0];
#pragma unroll
      for (int q = 1; q < NG; ++q) kg = (g2 == q) ? kgs[q] : kg;
      const KScal sg = make_kscal(P, kg);
      double b[NB], eg[NE];
#pragma unroll
      for (int f = 0; f < NB; ++f) b[f] = P.base[(size_t)f * P.npts + 2 * c0 + i];
      make_entry<FAM, fam_scaled<FAM>()>(b, sg, eg);
#pragma unroll
      for (int f = 0; f < NE; ++f) tbl[((size_t)g2 * NODES + i) * NE + f] = eg[f];
    }
    __builtin_amdgcn_wave_barrier();
    if (c == nchunks - 1) {                            // last node: start vector of the march
#pragma unroll
      for (int f = 0; f < NE; ++f) e[f] = mine[2 * nst * NE + f];
      coefficients<FAM, TRACK>(e, P, s, w, B0, trk);
      adjoint_start(P, B0, zp, zq);
    }
    int j = nst - 1;
    if (!fam_rcp4<FAM>() || !(j & 1)) {
      // one step at a time; with fam_rcp4() only an even top step (odd number of steps) is taken this way
      for (; j >= 0 && (!fam_rcp4<FAM>() || !(j & 1)); --j) {
        Coef Bm, B1;
#pragma unroll
        for (int f = 0; f < NE; ++f) { e[f] = mine[(2 * j + 1) * NE + f]; e2[f] = mine[2 * j * NE + f]; }
        coefficients2<FAM, TRACK>(e, e2, P, s, w, Bm, B1, trk);
        adjoint_step<FAM>(zp, zq, B0, Bm, B1, h, h2, h6, h3);
        B0 = B1;
      }
    }
    if (fam_rcp4<FAM>() && j >= 1) {
      // steps j and j - 1 with one division (coefficients4, as shoot_point_impl: same values), and the coefficients of a
      // pair formed one iteration AHEAD of the two RK4 steps that use them: they depend on the node and on omega only, not
      // on the marched row, so a lone wave -- a refinement launch has one or two per SIMD -- has the division of the next
      // pair to issue while the dependent fma chain of this pair's RK4 stages waits for its results
      double e1[NE], em2[NE];
      Coef Cm, C1, Cm2, C2;
#pragma unroll
      for (int f = 0; f < NE; ++f) {
        e[f] = mine[(2 * j + 1) * NE + f]; e1[f] = mine[2 * j * NE + f];
        em2[f] = mine[(2 * j - 1) * NE + f]; e2[f] = mine[(2 * j - 2) * NE + f];
      }
      coefficients4<FAM, TRACK>(e, e1, em2, e2, P, s, w, Cm, C1, Cm2, C2, trk);
      for (; j >= 3; j -= 2) {
        Coef Nm, N1, Nm2, N2;
#pragma unroll
        for (int f = 0; f < NE; ++f) {
          e[f] = mine[(2 * j - 3) * NE + f]; e1[f] = mine[(2 * j - 4) * NE + f];
          em2[f] = mine[(2 * j - 5) * NE + f]; e2[f] = mine[(2 * j - 6) * NE + f];
        }
        coefficients4<FAM, TRACK>(e, e1, em2, e2, P, s, w, Nm, N1, Nm2, N2, trk);
        adjoint_step<FAM>(zp, zq, B0, Cm, C1, h, h2, h6, h3);
        adjoint_step<FAM>(zp, zq, C1, Cm2, C2, h, h2, h6, h3);
        B0 = C2;
        Cm = Nm; C1 = N1; Cm2 = Nm2; C2 = N2;
      }
      adjoint_step<FAM>(zp, zq, B0, Cm, C1, h, h2, h6, h3);
      adjoint_step<FAM>(zp, zq, C1, Cm2, C2, h, h2, h6, h3);
      B0 = C2;
    }
    adjoint_rescale<FAM>(zp, zq, nsteps - c0 - nst, nsteps - c0);
  }
  const Mismatch M = boundary_algebra<FAM>(P, s, w, X, zp, zq, e2);
  finish_point(P, M, X, TRACK ? trk.crossed() : band_crossed(P, k, w), D, rel, st);
}

template <int FAM, int CHR, int NG>
__device__ __forceinline__ void shoot_point_wavegroup(const ShootDev& P, double k, double w, double& D, double& rel,
                                                      uint8_t& st, double* __restrict__ tbl) {
  if (fam_has_bands<FAM>() && P.use_bands) shoot_point_wavegroup_impl<FAM, !fam_has_bands<FAM>(), CHR, NG>(P, k, w, D, rel, st, tbl);
  else shoot_point_wavegroup_impl<FAM, true, CHR, NG>(P, k, w, D, rel, st, tbl);
}

// STATUS = false: the caller only uses D (section rounds of the refinement: the sign of D steers them, the status of a
// section point is never looked at -- the classification comes from the last polish evaluation): no per-node sign
// tracking (12 integer instructions per point-step for the twisted family), ES_PT_CONTINUUM is then never reported;
// D, rel and the other statuses are the same bits.
template <int FAM, bool STATUS = true>
__device__ __forceinline__ void shoot_point(const ShootDev& P, double k, double w, double w_cst, double& D,
                                            double& rel, uint8_t& st, double* __restrict__ sb) {
  if (FAM == FAM_CYLT) {                               // uniform branches: c1_power as a constant of the march (coef_pre)
    if (!STATUS) {
      if (P.c1_power == 2) shoot_point_impl<FAM, false, false, 2>(P, k, w, w_cst, D, rel, st, sb);
      else shoot_point_impl<FAM, false, false, 1>(P, k, w, w_cst, D, rel, st, sb);
    } else {
      if (P.c1_power == 2) shoot_point_impl<FAM, true, false, 2>(P, k, w, w_cst, D, rel, st, sb);
      else shoot_point_impl<FAM, true, false, 1>(P, k, w, w_cst, D, rel, st, sb);
    }
  } else if (!STATUS) {
    shoot_point_impl<FAM, false, false>(P, k, w, w_cst, D, rel, st, sb);
  } else if (fam_has_bands<FAM>() && P.use_bands) shoot_point_impl<FAM, !fam_has_bands<FAM>()>(P, k, w, w_cst, D, rel, st, sb);   // uniform branch
  else shoot_point_impl<FAM, true>(P, k, w, w_cst, D, rel, st, sb);
}

// LDS buffer every kernel that calls shoot_point declares
#define ES_POINT_LDS(FAM) __shared__ double es_point_lds[FamTraits<FAM>::NB * (2 * es_shoot_shared::CH + 1)]

}  // namespace es_shoot_shared
