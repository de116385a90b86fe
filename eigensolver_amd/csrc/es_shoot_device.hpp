// Device-side pieces of the shooting evaluation of the boundary determinant D(k, omega):
//   * per-family coefficient sets of the interior first-order system  y' = A(x; k, omega) y,  y = (u, v)
//   * the fixed-grid RK4 propagation of ONE row of the 2x2 interior transfer matrix (adjoint march from the far end
//     of the interior back to the boundary, see rk4_step_adjoint)
//   * the closed-form exterior solution with the reference's far-field initial values
//   * the boundary algebra (axis / symmetry condition by superposition, mismatch)
//
// Families (reference rows a3-a7 of SURVEY.md section 8):
//   FAM_CYL0   cylinder without twist/rotation (CD-C, CD-P, CF):   u = P, v = Xi = r*xi
//                  P'  = rho (Om^2 - wA^2)/r * Xi ,   Xi' = -r C2/D * P
//   FAM_CYLT   cylinder, general Hain-Luest/SGH set with v_phi, B_phi (CR-*):
//                  P'  = -C1/D P + C3/(r D) Xi ,      Xi' = -r C2/D P + C1/D Xi
//   FAM_SLABD  slab, non-uniform density (SD-P, SD-C), flux form:  u = Vx, v = F Vx'
//                  u'  = v/F ,                        v'  = F m0 u = rho (k^2 vA^2 - w^2) u
//   FAM_SLABF  slab with flow (SF-U uniform, SF-G Gaussian), the reference's equation: u = Vx, v = Vx'
//                  u'  = v ,                          v'  = -D v - coeff u      (D, coeff over ONE common denominator)
// All arithmetic is written out operation by operation (the translation unit is compiled with
// -ffp-contract=off; fused multiply-adds appear only as explicit fma() calls) and is mirrored line by line by the
// CPU port in oracle/c/shoot_port.c, so that the two agree to the last bit wherever no libm call is involved.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "es_bessel.hpp"
#include "../../include/eigensolver_amd.h"

enum { FAM_CYL0 = 0, FAM_CYLT = 1, FAM_SLABD = 2, FAM_SLABF = 3 };

// Plain-data description of one problem, passed by value to the kernels.
struct ShootDev {
  int family;
  int n_nodes;        // interior nodes N (RK4 steps = N-1); base tables hold 2N-1 points (nodes + midpoints)
  int npts;
  double xb;          // first node (boundary): -1 or +1
  double h;           // signed RK4 step
  const double* base; // [NB][npts] in HBM
  // exterior medium
  double rho_e, vAe2, ce2, cTe2, Se, U_e;
  double R_factor;    // far field at |x| = R_factor / k   (L * 2 pi)
  double ic0, ic1;    // far-field initial values of the reference's exterior solve
  // cylinder
  int m, m_ext, axis_bc, c1_power;
  double bc_const;     // as the fp64 marches need it: for the unnormalised family times the scale their z carries (adjoint_scale)
  double bc_const_raw; // as given (fp32 march, eigenfunction kernel: z at its true scale)
  // slab
  double slab_sign;   // -1 sausage: Vx(+1) = -Vx(-1);  +1 kink
  double c2_i, vA2_i, S_i, cT2_i, rho_i;   // uniform interior speeds of the flow slab
  int accept_norm;    // 0: rel uses max(|outer|,|inner|); 1: |outer| only (CR-KS:722)
  // FAM_CYL0 / FAM_SLABF / FAM_SLABD: continuum bands in phase speed, [term][min lo, max lo, min hi, max hi]
  //   cylinder: term 0 Alfven, 1 cusp;  flow slab: 0 sound, 1 tube, 2 Alfven, 3 Doppler-shifted frequency (lo = -inf);
  //   density slab: 0 sound, 1 tube, 2 Alfven (centred on W = 0)
  int use_bands, n_bands;
  double band[4][4];
};

template <int FAM> struct FamTraits;
// families whose continuum flag can come from phase-speed bands (band_crossed) instead of per-node sign tracking
template <int FAM> constexpr bool fam_has_bands() { return FAM == FAM_CYL0 || FAM == FAM_SLABF || FAM == FAM_SLABD; }
// SHAPE of A for the adjoint march: 0 off-diagonal (a11 = a22 = 0), 1 full, 2 companion (a11 = 0, a12 = 1)
template <> struct FamTraits<FAM_CYL0> { static constexpr int NB = 7, NE = 7, SHAPE = 0; static constexpr bool DIAG = false; };
template <> struct FamTraits<FAM_CYLT> { static constexpr int NB = 20, NE = 16, SHAPE = 1; static constexpr bool DIAG = true; };
template <> struct FamTraits<FAM_SLABD> { static constexpr int NB = 3, NE = 5, SHAPE = 0; static constexpr bool DIAG = false; };
template <> struct FamTraits<FAM_SLABF> { static constexpr int NB = 3, NE = 4, SHAPE = 2; static constexpr bool DIAG = true; };

// families whose adjoint marches run in the scaled-coefficient form
template <int FAM> constexpr bool fam_scaled() { return FAM == FAM_CYL0; }

// base-field indices
enum { C0_VZ = 0, C0_BA, C0_Q, C0_A1, C0_B1, C0_E1, C0_E2 };
// twisted cylinder: everything of the node entries that does not depend on k is formed once, on the host, for the
// problem's m (es_problem_create, operation for operation what make_entry did per lane and node before: same values)
enum { CT_MB = 0 /* m B_phi/r */, CT_BZ, CT_BA, CT_MV /* m v_phi/r */, CT_VZ, CT_Q, CT_E3 /* rho S */, CT_RHO, CT_E5, CT_E6,
       CT_C7 /* 2 B_phi v_phi */, CT_INVR, CT_BPHI, CT_E9, CT_E10, CT_S, CT_M2R2 /* m^2/r^2 */, CT_RDC3, CT_E13, CT_R };
enum { SD_RHO = 0, SD_C2, SD_VA2 };
enum { SF_U = 0, SF_DU, SF_DDU };

struct Coef { double a11, a12, a21, a22; };

// per-(k, m) scalars that do not depend on the node
struct KScal {
  double k, k2, m, m2;
  double kc2, kvA2, kcT2, k4c;   // flow slab only
  double h2;                     // half RK4 step (scaled coefficient form of the untwisted cylinder, make_entry<.., true>)
};

__device__ __forceinline__ KScal make_kscal(const ShootDev& P, double k) {
  KScal s;
  s.k = k; s.k2 = k * k; s.m = (double)P.m; s.m2 = s.m * s.m;
  s.kc2 = s.k2 * P.c2_i; s.kvA2 = s.k2 * P.vA2_i; s.kcT2 = s.k2 * P.cT2_i;
  s.k4c = s.k2 * s.k2 * P.cT2_i * P.c2_i;
  s.h2 = 0.5 * P.h;
  return s;
}

// ---- node entries: everything that depends on (node, k, m) but not on omega ---------------------------------
// SCALED (FAM_CYL0 adjoint marches): the entries that become a12 and a21 carry the factor h/2, so that coef_pre /
// coef_finish deliver (h/2) a12 and (h/2) a21 directly and the RK4 step needs no separate products with the step size
// (rk4_step_adjoint_scaled0).
template <int FAM, bool SCALED = false>
__device__ __forceinline__ void make_entry(const double* b, const KScal& s, double* e) {
  if (FAM == FAM_CYL0) {
    const double wA = s.k * b[C0_BA];
    const double wA2 = wA * wA;
    e[0] = s.k * b[C0_VZ];                     // Doppler shift k v_z
    e[1] = wA2;                                // omega_A^2
    e[2] = wA2 * b[C0_Q];                      // omega_c^2 = omega_A^2 c^2/(c^2+vA^2)
    e[3] = b[C0_A1];                           // rho / r
    // a21 = -r C2/(rho S t1 t2) with -r C2/(rho S) = g t2 - B (t2 + omega_c^2)^2, B = r/(rho S),
    // g = (m^2/r^2 + k^2) r/rho, t1 = t2 + omega_c^2 - omega_A^2.  Dividing the quadratic by t1 t2:
    //     a21 = -B + (lambda t2 + c0) / (t1 t2),   lambda = g - B (omega_A^2 + omega_c^2),   c0 = -B omega_c^4
    // so the numerator left over the denominator is LINEAR in t2 (one fma) and -B joins in the fma with 1/den.
    const double Bq = b[C0_B1];
    const double g = s.m2 * b[C0_E1] + s.k2 * b[C0_E2];
    e[4] = -Bq;
    e[5] = g - Bq * (e[1] + e[2]);
    e[6] = -(Bq * (e[2] * e[2]));
    if (SCALED) { e[3] *= s.h2; e[4] *= s.h2; e[5] *= s.h2; e[6] *= s.h2; }
  } else if (FAM == FAM_CYLT) {
    const double kb = fma(s.k, b[CT_BZ], b[CT_MB]);            // m B_phi/r + k B_z
    const double wA = fma(s.k, b[CT_BA], b[CT_MB]);            // as written in the reference (CF:581)
    const double wA2 = wA * wA;
    e[0] = fma(s.k, b[CT_VZ], b[CT_MV]);                       // shift: m v_phi/r + k v_z
    e[1] = wA2;
    e[2] = wA2 * b[CT_Q];
    e[3] = b[CT_E3];                                           // rho S
    e[4] = b[CT_RHO];
    e[5] = b[CT_E5];                                           // q1 = rho v_phi^2 / r
    e[6] = b[CT_E6];                                           // q2 = 2 B_phi^2 / r
    e[7] = b[CT_C7] * kb * b[CT_INVR];                         // q3 = 2 B_phi v_phi kb / r
    e[8] = kb * b[CT_BPHI];                                    // tt1
    e[9] = b[CT_E9];                                           // tt2 = rho v_phi
    e[10] = b[CT_E10];                                         // c1c = 2 m S / r^2
    e[11] = b[CT_S] * (b[CT_M2R2] + s.k2);                     // c2c = S (m^2/r^2 + k^2)
    e[12] = b[CT_RDC3];                                        // r d/dr[(B_phi/r)^2 - rho (v_phi/r)^2]
    e[13] = b[CT_E13];                                         // c3b = 4 S / r^2
    e[14] = b[CT_R];
    e[15] = b[CT_INVR];
  } else if (FAM == FAM_SLABD) {
    const double rho = b[SD_RHO], c2 = b[SD_C2], vA2 = b[SD_VA2];
    const double S = c2 + vA2;
    const double cT2 = c2 * vA2 / S;
    e[0] = s.k2 * c2;
    e[1] = s.k2 * cT2;
    e[2] = s.k2 * vA2;
    e[3] = rho * S;
    e[4] = rho;
  } else {
    e[0] = s.k * b[SF_U];
    e[1] = s.k * b[SF_DU];
    e[2] = s.k * b[SF_DDU];
    e[3] = 2.0 * e[1];
  }
}

// sign tracking for the continuum / singular-point flag: the IEEE sign bits (high dwords) of every watched term are
// OR-ed and AND-ed over all nodes with 32-bit integer ops (0.6 of the issue cost of an fp64 op on gfx950, and cheaper
// than v_cmp_lt_f64 + scalar mask accumulation, which was measured: tools/probe/valu_probe.hip); a term whose OR
// has the sign bit set while its AND has not took both signs, i.e. crossed zero inside the domain.
struct SignTrack {
  int any_or[4] = {0, 0, 0, 0};
  int any_and[4] = {-1, -1, -1, -1};
  __device__ __forceinline__ void add(int i, double t) {
    const int hi = __double2hiint(t);
    any_or[i] |= hi;
    any_and[i] &= hi;
  }
  __device__ __forceinline__ bool crossed() const {
    // sign bit set in the OR (some node negative) and clear in the AND (some node non-negative); unused terms: 0
    return (((any_or[0] & ~any_and[0]) | (any_or[1] & ~any_and[1]) | (any_or[2] & ~any_and[2]) |
             (any_or[3] & ~any_and[3])) < 0);
  }
};

// FAM_CYL0 / FAM_SLABF / FAM_SLABD without per-node tracking (ShootDev::use_bands): omega_A^2 = k^2 bA^2 and the Doppler shift k v_z scale
// with k, so  t1_j < 0  <=>  |W - vz_j| < |bA_j|  with W = omega/k, an interval (lo_j, hi_j) that does not depend on
// k.  When consecutive intervals overlap (checked on the host at problem creation) their union is
// (min lo, max hi) and  "negative at some node but not at all nodes"  <=>  W in (min lo, max hi) and not
// (max lo < W < min hi): four comparisons per point instead of four integer ops per node.  Same for the cusp term
// with |bA_j| sqrt(q_j), for the flow slab with |W - U_j| < c_i, c_Ti, vA_i and W < U_j (lo = -infinity), and for the
// density slab with |W| < c_j, c_Tj, vA_j.
__device__ __forceinline__ bool band_crossed(const ShootDev& P, double k, double w) {
  const double W = w / k;
  bool c = false;
  for (int t = 0; t < P.n_bands; ++t) {
    const bool some = (W > P.band[t][0]) && (W < P.band[t][3]);
    const bool all = (W > P.band[t][1]) && (W < P.band[t][2]);
    c = c || (some && !all);
  }
  return c;
}

// Coefficient matrix A(x; k, omega) of one node in two parts: everything except ONE reciprocal.  The entries marked
// "/den" are numerators; coef_finish() multiplies them with 1/den.  The caller computes the reciprocals of the
// mid-point and end-point denominators of a step with a single reciprocal (fast_rcp): inv = 1/(den_m * den_1),
// 1/den_m = den_1 * inv, 1/den_1 = den_m * inv  (one reciprocal per RK4 step instead of two).
struct CoefPre { double n11, n12, n21, n22, den; };

// C1P: c1_power of the twisted family as a compile-time constant (1 or 2), 0 = read it from the problem at run time.  The
// one-point kernels branch once per evaluation (shoot_point) and march with the constant: the run-time select between Om and
// Om^2 is two v_cndmask_b32 per node, four of the 113 instructions of a point-step there; same value either way.
template <int FAM, bool TRACK = true, int C1P = 0>
__device__ __forceinline__ void coef_pre(const double* e, const ShootDev& P, const KScal& s, double w, CoefPre& C,
                                         SignTrack& st) {
  if (FAM == FAM_CYL0) {
    const double Om = w - e[0];
    const double t1 = fma(Om, Om, -e[1]);                // Om^2 - omega_A^2
    const double t2 = fma(Om, Om, -e[2]);                // Om^2 - omega_c^2
    if (TRACK) { st.add(0, t1); st.add(1, t2); }
    C.n11 = 0.0;
    C.n12 = e[3] * t1;                                   // rho (Om^2 - wA^2) / r
    C.n21 = fma(e[5], t2, e[6]);                         // lambda t2 + c0              /den
    C.n22 = e[4];                                        // -B, added after the division (coef_finish)
    C.den = t1 * t2;
  } else if (FAM == FAM_CYLT) {
    // 23 fp64 instructions per node (33 until round 3, when every product-sum below was a multiply and an add): the sums
    // are explicit fused multiply-adds, grouped as in the fp32 screening kernel (coef_pre_f32); t2 T is formed once
    const double Om = w - e[0];
    const double Om2 = Om * Om;
    const double t1 = Om2 - e[1];
    const double t2 = Om2 - e[2];
    if (TRACK) { st.add(0, t1); st.add(1, t2); }
    const double D = e[3] * t1 * t2;
    const double Q = fma(Om, e[7], fma(Om2, e[6], -(t1 * e[5])));
    const double T = fma(e[9], Om, e[8]);
    const double OmP = (C1P == 2 || (C1P == 0 && P.c1_power == 2)) ? Om2 : Om;
    const double t2T = t2 * T;
    const double C1 = fma(Q, OmP, -(e[10] * t2T));
    const double C2 = fma(Om2, Om2, -(e[11] * t2));
    const double C3 = fma(D, fma(e[4], t1, e[12]), fma(Q, Q, -((e[13] * t2T) * T)));
    if (TRACK) { st.add(2, C3 * D); }   // F = r D / C3 changes sign where C3 does
    C.n11 = -C1;                                         // all four entries /den
    C.n22 = C1;
    C.n12 = C3 * e[15];
    C.n21 = -(e[14] * C2);
    C.den = D;
  } else if (FAM == FAM_SLABD) {
    const double w2 = w * w;
    const double n1 = e[0] - w2;                         // k^2 c^2 - w^2
    const double n2 = e[1] - w2;                         // k^2 cT^2 - w^2
    const double n3 = e[2] - w2;                         // k^2 vA^2 - w^2
    if (TRACK) { st.add(0, n1); st.add(1, n2); st.add(2, n3); }
    C.n11 = 0.0; C.n22 = 0.0;
    C.n12 = n1;                                          // 1/F = n1 / (rho S n2)      /den
    C.n21 = e[4] * n3;                                   // F m0
    C.den = e[3] * n2;
  } else {
    // m0, D and coeff of SF-G:416-427 over the common denominator den = S t Om^2 n1 (t = Om^2 - k^2 cT^2,
    // n1 = k^2 c^2 - Om^2, n3 = k^2 vA^2 - Om^2, G = S t^2 + k^4 cT^2 c^2):
    //   m0 = -n1 n3 / (S t),   D = -2 k U' G Om / den,   coeff = (k U'' S t Om n1 - 2 (k U')^2 G + n1^2 n3 Om^2) / den
    const double Om = w - e[0];
    const double Om2 = Om * Om;
    const double t = Om2 - s.kcT2;
    const double n1 = s.kc2 - Om2;
    const double n3 = s.kvA2 - Om2;
    if (TRACK) { st.add(0, n1); st.add(1, t); st.add(2, n3); st.add(3, Om); }
    const double St = P.S_i * t;
    const double G = fma(St, t, s.k4c);
    const double X = Om * n1;
    const double OX = Om * X;                            // n1 Om^2: shared by the numerator of coeff and the denominator
    const double g2 = e[3] * G;                          // 2 k U' G
    C.n11 = 0.0;
    C.n12 = 1.0;
    C.n22 = g2 * Om;                                     // -D                          /den
    C.n21 = -fma(e[2], St * X, fma(-g2, e[1], (n1 * n3) * OX));         // -coeff       /den
    C.den = St * OX;
  }
}

template <int FAM>
__device__ __forceinline__ void coef_finish(const CoefPre& C, double inv, Coef& A) {
  if (FAM == FAM_CYL0) {
    A.a11 = 0.0; A.a22 = 0.0; A.a12 = C.n12; A.a21 = fma(C.n21, inv, C.n22);
  } else if (FAM == FAM_CYLT) {
    // n11 = -n22: a11 = (-C1) inv = -(C1 inv) bit for bit -- one product, the negation rides on the source modifier of
    // the fma that uses it
    A.a22 = C.n22 * inv; A.a11 = -A.a22; A.a12 = C.n12 * inv; A.a21 = C.n21 * inv;
  } else if (FAM == FAM_SLABD) {
    A.a11 = 0.0; A.a22 = 0.0; A.a12 = C.n12 * inv; A.a21 = C.n21;
  } else {
    A.a11 = 0.0; A.a12 = 1.0; A.a21 = C.n21 * inv; A.a22 = C.n22 * inv;
  }
}

// single node (first node of a traversal): its own division
template <int FAM, bool TRACK = true, int C1P = 0>
__device__ __forceinline__ void coefficients(const double* e, const ShootDev& P, const KScal& s, double w,
                                             Coef& A, SignTrack& st) {
  CoefPre C;
  coef_pre<FAM, TRACK, C1P>(e, P, s, w, C, st);
  coef_finish<FAM>(C, 1.0 / C.den, A);
}

// Reciprocal for the hot loop: hardware seed r0 (v_rcp_f64, relative error e ~ 2^-23 at worst) and ONE third-order
// correction  r = r0 (1 + e + e^2),  e = 1 - x r0  (exact to fma rounding):  1/x = r0/(1 - e), so the truncation error
// is e^3 < 2^-69 and the result is the correctly rounded r0 + r0 (e + e^2) up to the last bit -- within 1 ulp of the
// IEEE quotient 1.0/x (measured, tools/probe/rcp_probe.hip: seed error 2^-24.4, result identical to the IEEE quotient
// for all of 4.2e6 random arguments; one Newton step alone leaves 2.2e-15), at 4 instructions instead of the
// 11 of the IEEE division sequence (v_div_scale x2, v_rcp, 5 fma, v_div_fmas, v_div_fixup) or 5 with two Newton
// steps.  Zero / non-finite denominators give non-finite results either way (flagged lanes).
__device__ __forceinline__ double fast_rcp(double x) {
#if defined(ES_IEEE_DIVISION)
  return 1.0 / x;
#else
  const double r0 = __builtin_amdgcn_rcp(x);
  const double e = fma(-x, r0, 1.0);
  const double t = fma(e, e, e);
  return fma(r0, t, r0);
#endif
}

// two nodes of one RK4 step (mid-point, end-point) with one division
template <int FAM, bool TRACK = true, int C1P = 0>
__device__ __forceinline__ void coefficients2(const double* em, const double* e1, const ShootDev& P,
                                              const KScal& s, double w, Coef& Am, Coef& A1, SignTrack& st) {
  CoefPre Cm, C1;
  coef_pre<FAM, TRACK, C1P>(em, P, s, w, Cm, st);
  coef_pre<FAM, TRACK, C1P>(e1, P, s, w, C1, st);
  const double inv = fast_rcp(Cm.den * C1.den);
  coef_finish<FAM>(Cm, C1.den * inv, Am);
  coef_finish<FAM>(C1, Cm.den * inv, A1);
}

// four nodes of TWO consecutive RK4 steps (mid-point and end-point of each) with one division: the product tree
//   d12 = den_m den_1, d34 = den_m' den_1', inv = 1 / (d12 d34), 1/d12 = d34 inv, 1/d34 = d12 inv, 1/den_m = den_1 (1/d12), ...
// costs 9 multiplications and one reciprocal (v_rcp_f64, quarter rate, + its 3-fma correction) per two steps where two
// coefficients2 cost 6 and two: half a reciprocal less per point-step (152.5 -> 144.5 issue cycles for the untwisted
// cylinder; same-box A/B of the headline 20.5 -> 19.8 ms per step).  Every fp64 march of a family with fam_rcp4() pairs
// its steps the same way -- step j with step j - 1 for every ODD j, counted from the boundary (j = 0); an even top step
// (odd number of steps) is taken alone -- so that the grid kernels, the point kernels (chunks of CH or CHR steps: both
// even) and the CPU port produce the same bits.  The four denominators are products of two watched terms each; their
// product stays far inside the fp64 range (|t| <= ~1e4 in the reference's units, >= 1e-300 only at a flagged point).
// Not the twisted cylinder: its point kernels form 16 entries per node and lane, four nodes at once do not fit 256 registers.
#if defined(ES_NO_RCP4)                                // A/B build (timing only: the CPU port pairs the steps)
template <int FAM> constexpr bool fam_rcp4() { return false; }
#else
template <int FAM> constexpr bool fam_rcp4() { return FAM != FAM_CYLT; }
#endif

template <int FAM, bool TRACK = true>
__device__ __forceinline__ void coefficients4(const double* em, const double* e1, const double* em2, const double* e12,
                                              const ShootDev& P, const KScal& s, double w, Coef& Am, Coef& A1, Coef& Am2,
                                              Coef& A12, SignTrack& st) {
  CoefPre Cm, C1, Cn, C2;
  coef_pre<FAM, TRACK>(em, P, s, w, Cm, st);
  coef_pre<FAM, TRACK>(e1, P, s, w, C1, st);
  coef_pre<FAM, TRACK>(em2, P, s, w, Cn, st);
  coef_pre<FAM, TRACK>(e12, P, s, w, C2, st);
  const double d12 = Cm.den * C1.den, d34 = Cn.den * C2.den;
  const double inv = fast_rcp(d12 * d34);
  const double i12 = d34 * inv, i34 = d12 * inv;
  coef_finish<FAM>(Cm, C1.den * i12, Am);
  coef_finish<FAM>(C1, Cm.den * i12, A1);
  coef_finish<FAM>(Cn, C2.den * i34, Am2);
  coef_finish<FAM>(C2, Cn.den * i34, A12);
}

// ---- one RK4 step of the ADJOINT (row-vector) propagation -----------------------------------------------------
// Only one row of the interior transfer matrix T is needed (the axis / symmetry condition is one linear functional
// of the state at the far end).  With M the RK4 step matrix from node j to j+1, M^T is the RK4 step of the
// transposed system taken through the stages in reverse order (A_{j+1}^T, A_{j+1/2}^T, A_j^T), same h.  So the
// row  r = r_end^T T  is obtained by ONE vector z = (p, q) marched from the far end back to the boundary:
//      z <- M_j^T z ,   rhs(z) = A^T z = (a11 p + a21 q, a12 p + a22 q).
template <int SHAPE>
__device__ __forceinline__ void rk4_step_adjoint(double& p, double& q, const Coef& B0, const Coef& Bm, const Coef& B1,
                                                 double h, double h2, double h6, double h3) {
#define ES_RHS_T(A, pp, qq, kp, kq)                                                          \
  if (SHAPE == 1)      { kp = fma(-A.a22, pp, A.a21 * qq); kq = fma(A.a22, qq, A.a12 * pp); } /* a11 = -a22 */ \
  else if (SHAPE == 2) { kp = A.a21 * qq;                 kq = fma(A.a22, qq, pp); }         \
  else                 { kp = A.a21 * qq;                 kq = A.a12 * pp; }
  double k1p, k1q, k2p, k2q, k3p, k3q, k4p, k4q, tp, tq;
  ES_RHS_T(B0, p, q, k1p, k1q);
  tp = fma(h2, k1p, p); tq = fma(h2, k1q, q);
  ES_RHS_T(Bm, tp, tq, k2p, k2q);
  tp = fma(h2, k2p, p); tq = fma(h2, k2q, q);
  ES_RHS_T(Bm, tp, tq, k3p, k3q);
  tp = fma(h, k3p, p); tq = fma(h, k3q, q);
  ES_RHS_T(B1, tp, tq, k4p, k4q);
  p = fma(h6, k1p + k4p, fma(h3, k2p + k3p, p));         // p + h/6 (k1 + k4) + h/3 (k2 + k3)
  q = fma(h6, k1q + k4q, fma(h3, k2q + k3q, q));
#undef ES_RHS_T
}

// The same RK4 step for an off-diagonal A whose entries arrive pre-multiplied by h/2 (A_ = (h/2) a21, B_ = (h/2) a12):
// with t1 = z + A0^T z, t2 = z + Am^T t1, t3 = z + 2 Am^T t2 (the three stage arguments) the classical combination
// z + h/6 (k1 + 2 k2 + 2 k3 + k4) equals (t1 + 2 t2 + t3 - z + A1^T t3) / 3 -- 18 instructions instead of 22, and no
// separate products with the step size.  Mathematically the step of rk4_step_adjoint<0>; rounding differs.
__device__ __forceinline__ void rk4_step_adjoint_scaled0(double& p, double& q, const Coef& B0, const Coef& Bm,
                                                         const Coef& B1) {
  const double tp1 = fma(B0.a21, q, p),   tq1 = fma(B0.a12, p, q);
  const double tp2 = fma(Bm.a21, tq1, p), tq2 = fma(Bm.a12, tp1, q);
  const double am2 = Bm.a21 + Bm.a21,     bm2 = Bm.a12 + Bm.a12;
  const double tp3 = fma(am2, tq2, p),    tq3 = fma(bm2, tp2, q);
  const double sp = fma(2.0, tp2, tp1 + tp3) - p;
  const double sq = fma(2.0, tq2, tq1 + tq3) - q;
  constexpr double third = 1.0 / 3.0;
  p = fma(B1.a21, tq3, sp) * third;
  q = fma(B1.a12, tp3, sq) * third;
}

// The determinant needs the row z only up to a common factor (the far-end condition of the untwisted cylinder is
// homogeneous: D depends on z_p / z_q; a non-zero target is multiplied by the same factor, see adjoint_scale), so the
// marches of that family drop the division by 3: 3 z' = t1 + 2 t2 + t3 - z + A1^T t3 -- 14 instructions per step
// instead of 18.  z then grows by 3 per step and is brought back by an exact power of two at the end of every LDS
// chunk, chosen from the number of steps marched so far so that the accumulated factor 3^s 2^-floor(s log2 3) stays in
// [1, 2) whatever the node count: adjoint_rescale(s_before, s_after) = 2^-(floor(s_after log2 3) - floor(s_before log2 3)).
__device__ __forceinline__ void rk4_step_adjoint_scaled0_x3(double& p, double& q, const Coef& B0, const Coef& Bm,
                                                            const Coef& B1) {
  const double tp1 = fma(B0.a21, q, p),   tq1 = fma(B0.a12, p, q);
  const double tp2 = fma(Bm.a21, tq1, p), tq2 = fma(Bm.a12, tp1, q);
  const double am2 = Bm.a21 + Bm.a21,     bm2 = Bm.a12 + Bm.a12;
  const double tp3 = fma(am2, tq2, p),    tq3 = fma(bm2, tp2, q);
  // t1 + 2 t2 + (t3 - z) with t3 - z = 2 Am^T t2 taken as the product it is: two fmas per component instead of add, fma, sub
  const double sp = fma(am2, tq2, fma(2.0, tp2, tp1));
  const double sq = fma(bm2, tp2, fma(2.0, tq2, tq1));
  p = fma(B1.a21, tq3, sp);
  q = fma(B1.a12, tp3, sq);
}

// families whose fp64 marches carry z times a known factor (rk4_step_adjoint_scaled0_x3)
template <int FAM> constexpr bool fam_unnormalised() { return FAM == FAM_CYL0; }

__host__ __device__ inline int adjoint_rescale_exp(int s_before, int s_after) {
  return (int)((double)s_before * 1.5849625007211561) - (int)((double)s_after * 1.5849625007211561);
}

template <int FAM>
__device__ __forceinline__ void adjoint_rescale(double& p, double& q, int s_before, int s_after) {
  if (fam_unnormalised<FAM>()) {
    const int ex = adjoint_rescale_exp(s_before, s_after);
    p = ldexp(p, ex);
    q = ldexp(q, ex);
  }
}

template <int FAM>
__device__ __forceinline__ void adjoint_step(double& p, double& q, const Coef& B0, const Coef& Bm, const Coef& B1,
                                             double h, double h2, double h6, double h3) {
  if (fam_unnormalised<FAM>()) rk4_step_adjoint_scaled0_x3(p, q, B0, Bm, B1);
  else if (fam_scaled<FAM>()) rk4_step_adjoint_scaled0(p, q, B0, Bm, B1);
  else rk4_step_adjoint<FamTraits<FAM>::SHAPE>(p, q, B0, Bm, B1, h, h2, h6, h3);
}

// the same step with z kept at its true scale (eigenfunction kernel)
template <int FAM>
__device__ __forceinline__ void adjoint_step_normalised(double& p, double& q, const Coef& B0, const Coef& Bm, const Coef& B1,
                                                        double h, double h2, double h6, double h3) {
  if (fam_scaled<FAM>()) rk4_step_adjoint_scaled0(p, q, B0, Bm, B1);
  else rk4_step_adjoint<FamTraits<FAM>::SHAPE>(p, q, B0, Bm, B1, h, h2, h6, h3);
}

// start vector of the adjoint march: the functional the far-end condition applies to (u, v)
__device__ __forceinline__ void adjoint_start(const ShootDev& P, const Coef& A_last, double& p, double& q) {
  if ((P.family == FAM_CYL0 || P.family == FAM_CYLT) && P.axis_bc == ES_AXIS_SAUSAGE) {
    p = A_last.a11; q = A_last.a12;        // P'(r_ax) = a11 P + a12 Xi = 0   (CD-C:1082-1085)
  } else {
    p = 1.0; q = 0.0;                      // P(r_ax) = target (kink) ; Vx(+1) = -/+ Vx(-1) (slabs)
  }
}

// ---- exterior ------------------------------------------------------------------------------------------------
struct Exterior {
  double m_e;       // reference's m_e
  double cst;       // xi_e_const (cylinder) or p_e_const (slab)
  double yb, dyb;   // exterior solution and its derivative at the boundary, scaled to |yb| = 1 (sign kept)
  double Oe;        // Doppler-shifted exterior frequency (slab)
  int status;       // ES_PT_OK / LEAKY / NONFINITE
};

// w_cst: frequency at which xi_e_const is evaluated (= w everywhere except inside CR-SF's locate_sausage, which reads
// the enclosing loop's stale value)
__device__ __forceinline__ Exterior exterior_cylinder(const ShootDev& P, double k, double w, double w_cst) {
  Exterior X;
  const double k2 = k * k, w2 = w * w;
  X.Oe = w;
  X.m_e = ((k2 * P.vAe2 - w2) * (k2 * P.ce2 - w2)) / (P.Se * (k2 * P.cTe2 - w2));     // CF:699
  X.cst = -1.0 / (P.rho_e * (k2 * P.vAe2 - w_cst * w_cst));                          // CF:702
  X.yb = X.dyb = NAN;
  if (X.m_e < 0.0) { X.status = ES_PT_LEAKY; return X; }
  if (!(X.m_e > 0.0) || !isfinite(X.m_e) || !isfinite(X.cst)) { X.status = ES_PT_NONFINITE; return X; }
  X.status = ES_PT_OK;
  const double mu = sqrt(X.m_e);
  const double sgn = (P.xb < 0.0) ? -1.0 : 1.0;
  const double xR = mu * (P.R_factor / k), xb = mu;
  const int n = P.m_ext;
  double Kb, Kb1, KR, KR1;
  esb::ke_pair(n, xb, Kb, Kb1);
  esb::ke_pair(n, xR, KR, KR1);
  const double dn = (double)n;
  const double dKb = -Kb1 + (dn / xb) * Kb;          // e^x K_n'(x)
  const double dKR = -KR1 + (dn / xR) * KR;
  const double g = P.ic1 / (sgn * mu);
  // P = a I + b K with (a, b) from the far-field values; common positive factor xR e^{xR - xb} dropped
  double Pv, dPv;
  const double gap = xR - xb;
  if (gap < 40.0) {
    double Ib, Ib1, IR, IR1;
    esb::ie_pair_from_k(n, xb, Kb, Kb1, Ib, Ib1);
    esb::ie_pair_from_k(n, xR, KR, KR1, IR, IR1);
    const double dIb = Ib1 + (dn / xb) * Ib;         // e^-x I_n'(x)
    const double dIR = IR1 + (dn / xR) * IR;
    const double a_s = -(P.ic0 * dKR - g * KR);
    const double b_s = -(g * IR - P.ic0 * dIR);
    const double E2 = exp(-2.0 * gap);
    Pv = b_s * Kb + E2 * a_s * Ib;
    dPv = sgn * mu * (b_s * dKb + E2 * a_s * dIb);
  } else {
    // I-admixture below 1e-34: b ~ (ic0 - g) up to a positive factor (I' ~ I at large argument is NOT assumed:
    // the sign of b only needs g I - ic0 I', evaluated with the leading asymptotic ratio I'/I = 1 - (1/2x) ...)
    double IR, IR1;
    // ratio I_n'/I_n at xR from the uniform large-argument expansion is not needed to machine precision here:
    // only sign(b) enters (the K-part is normalised away).  Use I'/I = 1 - 1/(2x) - (4n^2-1)/(8x^2).
    const double rI = 1.0 - 0.5 / xR - (4.0 * dn * dn - 1.0) / (8.0 * xR * xR);
    (void)IR; (void)IR1;
    const double b_s = -(g - P.ic0 * rI);
    Pv = b_s * Kb;
    dPv = sgn * mu * (b_s * dKb);
  }
  const double nrm = fabs(Pv);
  X.yb = Pv / nrm;
  X.dyb = dPv / nrm;
  if (!isfinite(X.yb) || !isfinite(X.dyb)) X.status = ES_PT_NONFINITE;
  return X;
}

__device__ __forceinline__ Exterior exterior_slab(const ShootDev& P, double k, double w) {
  Exterior X;
  const double k2 = k * k;
  const double Oe = w - k * P.U_e;
  const double Oe2 = Oe * Oe;
  X.Oe = Oe;
  X.m_e = ((k2 * P.vAe2 - Oe2) * (k2 * P.ce2 - Oe2)) / (P.Se * (k2 * P.cTe2 - Oe2));          // SF-U:542
  X.cst = P.rho_e * P.Se * (k2 * P.cTe2 - Oe2) / (Oe * (k2 * P.ce2 - Oe2));                   // SF-U:545
  X.yb = X.dyb = NAN;
  if (X.m_e < 0.0) { X.status = ES_PT_LEAKY; return X; }
  if (!(X.m_e > 0.0) || !isfinite(X.m_e) || !isfinite(X.cst)) { X.status = ES_PT_NONFINITE; return X; }
  X.status = ES_PT_OK;
  const double mu = sqrt(X.m_e);
  const double R = P.R_factor / k;
  const double E2 = exp(-2.0 * mu * (R - 1.0));
  const double gp = P.ic0 + P.ic1 / mu, gm = P.ic0 - P.ic1 / mu;
  const double V = gp + E2 * gm;
  const double dV = mu * (gp - E2 * gm);
  const double nrm = fabs(V);
  X.yb = V / nrm;
  X.dyb = dV / nrm;
  if (!isfinite(X.yb) || !isfinite(X.dyb)) X.status = ES_PT_NONFINITE;
  return X;
}

// ---- boundary algebra ------------------------------------------------------------------------------------------
// Inputs: the row r = (r1, r2) of the transfer matrix selected by the far-end condition (r . (u_b, v_b) = target),
// the node entries of the first node and the exterior (outer = cst * dyb, yb, Oe).  Output: mismatch d = outer - inner.
struct Mismatch { double d, outer, inner; };

template <int FAM, class Ext>
__device__ __forceinline__ Mismatch boundary_algebra(const ShootDev& P, const KScal& s, double w, const Ext& X,
                                                     double r1, double r2, const double* e_first) {
  Mismatch M;
  if (FAM == FAM_CYL0 || FAM == FAM_CYLT) {
    const double Pb = X.yb;
    const double xi_e = X.outer;                                        // left_xi_solution[-1], CF:775
    double Xb;
    if (P.axis_bc == ES_AXIS_KINK) {
      Xb = (P.bc_const * xi_e - r1 * Pb) / r2;                          // P(r_ax) = B_phi(-1)^2 xi_e, CF:795
    } else if (P.axis_bc == ES_AXIS_ROTATION_KINK) {
      Xb = (-(P.bc_const * xi_e) - r1 * Pb) / r2;                       // CR-KF:695-698
    } else {
      Xb = -(r1 * Pb) / r2;                                             // P'(r_ax) = 0, CD-C:1082-1085
    }
    const double xi_i = Xb / P.xb;                                      // inside_xi_solution[0], CF:798
    M.outer = xi_e; M.inner = xi_i; M.d = xi_e - xi_i;
  } else {
    const double P_left = X.outer;                                      // left_P_solution[-1], SF-U:559
    if (FAM == FAM_SLABD) {
      const double Vb = X.yb;                                           // SD-P:473
      const double sv = (P.slab_sign - r1) * Vb / r2;                   // v(-1) = F Vx'(-1)
      M.inner = sv / w;                                                 // P_Ti Vx' = (F/w)(sv/F), SD-P:346,495
    } else {
      const double Omb = w - e_first[0];                                // w - k U_i(-1)
      const double Vb = X.yb * Omb / X.Oe;                              // SF-U:558
      const double sv = (P.slab_sign - r1) * Vb / r2;                   // Vx'(-1)
      const double Omb2 = Omb * Omb;
      const double PTi = P.rho_i * P.S_i * (s.kcT2 - Omb2) / (Omb * (s.kc2 - Omb2));   // SF-G:433
      M.inner = PTi * sv;
    }
    M.outer = P_left;
    M.d = P_left - M.inner;
  }
  return M;
}
