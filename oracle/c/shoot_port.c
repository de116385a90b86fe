/* ORACLE (test infrastructure only; never linked into or called by the product library).
 *
 * Plain-C CPU port of the shooting evaluation of the boundary determinant D(k, omega): the same algorithm the HIP
 * kernels run (closed-form exterior with the reference's far-field initial values, fixed-grid RK4 propagation of
 * the interior transfer matrix on the reference's `ix` grid, axis / symmetry condition by superposition,
 * mismatch), written independently in C, operation order fixed (compile with -ffp-contract=off; fused
 * multiply-adds only through fma()).  It restates, per (k, omega), what the reference workers compute:
 *   Cylinder/Non-uniform flow/Coronal/solvers/Cylinder_method_flow_testing.py:694-804, :991-1111   (CF)
 *   Cylinder/Non-uniform density/Coronal/solvers/Density_cylinder.py:694-804, :990-1104            (CD-C)
 *   Cylinder/Rotational flow/Photospheric/Solvers/Twisted_photospheric_nonlinear_flow_kink_fast.py:601-712 (CR-KF)
 *   Slab/Non uniform density/Photospheric/Solvers/multiprocessor_Inhomogeneous_method.py:421-501    (SD-P)
 *   Slab/Non uniform flow/Solver/flow_multiprocessor_coronal.py:400-480, flow_multiprocessor.py:533-586 (SF-G, SF-U)
 * and the bracket / 3-point-linspace bisection of e.g. CF:823-829 run to convergence on a (k, omega) grid.
 *
 * Pinned by tests/test_oracle_port.py against oracle/cylinder.py / oracle/slab.py (DOP853 "truth", themselves
 * pinned to traces of the reference) and used (a) as the bit-level comparator of the HIP kernels, (b) as the
 * timed CPU baseline ("port") of bench.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../../include/eigensolver_amd.h"

#define PORT_CH 128                                     /* es_shoot_shared::CH */
static int port_rescale_exp(int s_before, int s_after) {   /* adjoint_rescale_exp: steps marched before / after the chunk */
  return (int)((double)s_before * 1.5849625007211561) - (int)((double)s_after * 1.5849625007211561);
}
#define PORT_CHEB_N 25
#define PI 3.14159265358979323846264338327950288
#define EULER 0.57721566490153286060651209008240243

/* ---- scaled modified Bessel functions (same algorithms as the device header, written independently) -------- */
static void ke01(double x, double* k0, double* k1) {
  if (x <= 2.0) {
    double t = 0.25 * x * x, lg = log(0.5 * x) + EULER;
    double term0 = 1.0, i0 = 1.0, s0 = 0.0, hk = 0.0, term1 = 1.0, i1 = 1.0, s1 = 1.0;
    for (int k = 1; k < 40; ++k) {
      double kk = (double)k;
      term0 = term0 * t / (kk * kk);
      hk += 1.0 / kk;
      i0 += term0;
      s0 += hk * term0;
      term1 = term1 * t / (kk * (kk + 1.0));
      i1 += term1;
      s1 += (hk + hk + 1.0 / (kk + 1.0)) * term1;
      if (term0 < 1e-18 * i0) break;
    }
    double ex = exp(x);
    *k0 = ex * (-lg * i0 + s0);
    *k1 = ex * (1.0 / x + lg * (0.5 * x) * i1 - 0.25 * x * s1);
  } else {
    /* Chebyshev series of sqrt(x) e^x K_{0,1}(x) in y = 4/x - 1 (the tables of csrc/es_bessel.hpp), Clenshaw's recurrence in
       the operation order of the HIP code (whose qdiv is the IEEE quotient in the ES_IEEE_DIVISION build) */
    static const double c0[PORT_CHEB_N] = {
    1.2201515410329777, -0.0314481013119645, 0.0015698838857300533, -0.00012849549581627802, 
    1.39498137188765e-05, -1.8317555227191195e-06, 2.766813639445015e-07, -4.660489897687948e-08, 
    8.574034017414225e-09, -1.6975345093890614e-09, 3.5773972814003283e-10, -7.957489244477396e-11, 
    1.8559491149549264e-11, -4.514597883374519e-12, 1.1403405882073441e-12, -2.9800969231481784e-13, 
    8.032890775068375e-14, -2.2275133267462965e-14, 6.340076476276646e-15, -1.848593377920907e-15, 
    5.5120559994043335e-16, -1.6782311257549006e-16, 5.2103917776435543e-17, -1.6475805939842632e-17, 
    5.3004337711773354e-18 };
    static const double c1[PORT_CHEB_N] = {
    1.3603130952422213, 0.10392373657681724, -0.002857816859622779, 0.00019521551847135162, 
    -1.936197974166083e-05, 2.406484947837217e-06, -3.5019606030878126e-07, 5.7410841254500495e-08, 
    -1.0345762465678097e-08, 2.0150497551970347e-09, -4.1903547593419254e-10, 9.218315187605315e-11, 
    -2.129967838427791e-11, 5.139639673482343e-12, -1.2891739609498229e-12, 3.348419666052243e-13, 
    -8.976705182010146e-14, 2.4771544242195988e-14, -7.0198370892147685e-15, 2.038703166239861e-15, 
    -6.057047270643018e-16, 1.8380935752430455e-16, -5.689462849193648e-17, 1.7940510478863572e-17, 
    -5.7567444820733025e-18 };
    const double y = 4.0 / x - 1.0, y2 = y + y;
    double p1 = 0.0, p2 = 0.0, q1 = 0.0, q2 = 0.0;
    for (int j = PORT_CHEB_N - 1; j >= 1; --j) {
      const double pn = fma(y2, p1, c0[j] - p2), qn = fma(y2, q1, c1[j] - q2);
      p2 = p1; p1 = pn;
      q2 = q1; q1 = qn;
    }
    const double rs = 1.0 / sqrt(x);
    *k0 = fma(y, p1, c0[0] - p2) * rs;
    *k1 = fma(y, q1, c1[0] - q2) * rs;
  }
}
static void ke_pair(int n, double x, double* kn, double* kn1) {
  double a, b;
  ke01(x, &a, &b);
  double tox = 2.0 / x;
  for (int j = 1; j <= n; ++j) { double c = a + (double)j * tox * b; a = b; b = c; }
  *kn = a; *kn1 = b;
}
static void ie_pair(int n, double x, double* in_, double* in1) {
  double t = 0.25 * x * x, hx = 0.5 * x, pre = 1.0;
  for (int j = 1; j <= n; ++j) pre *= hx / (double)j;
  double ta = 1.0, sa = 1.0, tb = 1.0, sb = 1.0;
  for (int k = 1; k < 400; ++k) {
    double kk = (double)k;
    ta = ta * t / (kk * (kk + (double)n));
    tb = tb * t / (kk * (kk + (double)n + 1.0));
    sa += ta; sb += tb;
    if (ta < 1e-18 * sa) break;
  }
  double ex = exp(-x);
  *in_ = ex * pre * sa;
  *in1 = ex * pre * (hx / ((double)n + 1.0)) * sb;
}

/* I_n, I_{n+1} (scaled) from known K_n, K_{n+1}: ratio by Miller's backward recurrence, size from the Wronskian
   I_n K_{n+1} + I_{n+1} K_n = 1/x */
static void ie_pair_from_k(int n, double x, double kn, double kn1, double* in_, double* in1) {
  if (x < 0.5) { ie_pair(n, x, in_, in1); return; }
  const int M = n + 10 + (int)sqrt(40.0 * x);
  const double tox = 2.0 / x;
  double ip = 0.0, ic = 1e-200, ktox = (double)M * tox;
  for (int k = M; k > n; --k) {
    double im = fma(ktox, ic, ip);
    ip = ic; ic = im; ktox -= tox;
  }
  double f = ip / ic;
  *in_ = 1.0 / (x * fma(f, kn, kn1));
  *in1 = f * *in_;
}

/* ---- problem ------------------------------------------------------------------------------------------------ */
typedef struct {
  int family, n_nodes, npts, nb;
  double xb, h;
  double* base;
  double rho_e, vAe2, ce2, cTe2, Se, U_e, R_factor, ic0, ic1;
  int m, m_ext, axis_bc, c1_power;
  double bc_const, slab_sign, c2_i, vA2_i, S_i, cT2_i, rho_i;
  int accept_norm;
  int use_bands, n_bands; /* families 0 and 3: continuum flag from phase-speed bands instead of per-node sign tracking */
  double band[4][4];      /* [term][min lo, max lo, min hi, max hi]: cylinder Alfven, cusp; flow slab sound, tube, Alfven, Om */
} port_problem;

static const int NB_OF[4] = {7, 11, 3, 3};

port_problem* port_create(const es_shoot_desc* d, const es_profiles* pr) {
  port_problem* P = (port_problem*)calloc(1, sizeof(port_problem));
  int N = d->n_nodes, npts = 2 * N - 1, nb = NB_OF[d->geometry];
  P->family = d->geometry; P->n_nodes = N; P->npts = npts; P->nb = nb;
  P->base = (double*)malloc(sizeof(double) * (size_t)nb * npts);
#define B(f, i) P->base[(size_t)(f) * npts + (i)]
  for (int i = 0; i < npts; ++i) {
    if (d->geometry == ES_GEOM_CYLINDER || d->geometry == ES_GEOM_CYLINDER_TWIST) {
      double r = pr->r[i], rho = pr->rho[i], c2 = pr->c2[i], Bz = pr->Bz[i];
      double Bphi = pr->Bphi ? pr->Bphi[i] : 0.0, vz = pr->vz ? pr->vz[i] : 0.0, vphi = pr->vphi ? pr->vphi[i] : 0.0;
      double sr = sqrt(rho), bA = Bz / sr, vA = (Bz + Bphi) / sr, S = c2 + vA * vA, q = c2 / S;
      if (d->geometry == ES_GEOM_CYLINDER) {
        B(0, i) = vz; B(1, i) = bA; B(2, i) = q; B(3, i) = rho / r; B(4, i) = r / (rho * S);
        B(5, i) = 1.0 / (r * rho); B(6, i) = r / rho;
      } else {
        B(0, i) = r; B(1, i) = 1.0 / r; B(2, i) = rho; B(3, i) = S; B(4, i) = q; B(5, i) = bA; B(6, i) = Bz;
        B(7, i) = Bphi / r; B(8, i) = vphi / r; B(9, i) = vz; B(10, i) = pr->rdC3 ? pr->rdC3[i] : 0.0;
      }
    } else if (d->geometry == ES_GEOM_SLAB_DENSITY) {
      B(0, i) = pr->rho[i]; B(1, i) = pr->c2[i]; B(2, i) = pr->vA2[i];
    } else {
      B(0, i) = pr->U[i]; B(1, i) = pr->dU ? pr->dU[i] : 0.0; B(2, i) = pr->ddU ? pr->ddU[i] : 0.0;
    }
  }
#undef B
  P->xb = d->x_boundary;
  P->h = (d->x_end - d->x_boundary) / (double)(N - 1);
  P->rho_e = d->rho_e; P->vAe2 = d->vA_e * d->vA_e; P->ce2 = d->c_e * d->c_e; P->cTe2 = d->cT_e * d->cT_e;
  P->Se = P->vAe2 + P->ce2; P->U_e = d->U_e;
  P->R_factor = d->L_factor * 2.0 * 3.14159265358979323846;
  P->ic0 = d->ic_value; P->ic1 = d->ic_slope;
  P->m = d->m; P->m_ext = d->m_ext; P->axis_bc = d->axis_bc; P->c1_power = d->c1_power; P->bc_const = d->bc_const;
  if (d->geometry == ES_GEOM_CYLINDER && d->bc_const != 0.0) {   /* the factor the unnormalised march leaves on z (es_problem_create) */
    double c = 1.0;
    const int nsteps = N - 1;
    for (int ch = (nsteps + PORT_CH - 1) / PORT_CH - 1; ch >= 0; --ch) {
      const int c0 = ch * PORT_CH, nst = (nsteps - c0 < PORT_CH) ? (nsteps - c0) : PORT_CH;
      for (int i = 0; i < nst; ++i) c *= 3.0;
      c = ldexp(c, port_rescale_exp(nsteps - c0 - nst, nsteps - c0));
    }
    P->bc_const = d->bc_const * c;
  }
  P->slab_sign = (d->slab_mode == ES_SLAB_MODE_SAUSAGE) ? -1.0 : 1.0;
  P->c2_i = d->c_i * d->c_i; P->vA2_i = d->vA_i * d->vA_i; P->S_i = P->c2_i + P->vA2_i;
  P->cT2_i = (P->S_i > 0.0) ? P->c2_i * P->vA2_i / P->S_i : 0.0;
  P->rho_i = d->rho_i;
  P->accept_norm = d->accept_norm;
  if (d->geometry == ES_GEOM_CYLINDER || d->geometry == ES_GEOM_SLAB_FLOW || d->geometry == ES_GEOM_SLAB_DENSITY) {
    /* node j is inside band t of phase speed W iff centre_j - a_j < W < centre_j + a_j (cylinder: centre v_z, a = |bA|,
       |bA| sqrt(q); flow slab: centre U, a = c_i, cT_i, vA_i, and the half line W < U_j; density slab: centre 0,
       a = c_j, cT_j, vA_j); if consecutive node intervals overlap, "inside at some node but not at all" is a test
       against four numbers */
    const int cyl = (d->geometry == ES_GEOM_CYLINDER), flow = (d->geometry == ES_GEOM_SLAB_FLOW);
    P->use_bands = getenv("ES_FORCE_SIGN_TRACKING") ? 0 : 1;
    P->n_bands = cyl ? 2 : (flow ? 4 : 3);
    const double slab_a[3] = {sqrt(P->c2_i), sqrt(P->cT2_i), sqrt(P->vA2_i)};
    for (int t = 0; t < P->n_bands; ++t) {
      double lo_min = INFINITY, lo_max = -INFINITY, hi_min = INFINITY, hi_max = -INFINITY, lo_prev = 0.0, hi_prev = 0.0;
      const int half_line = (flow && t == 3);
      int never = !cyl;
      for (int i = 0; i < npts; ++i) {
        double centre, a;
        if (cyl) {
          centre = P->base[i];
          a = fabs(P->base[(size_t)1 * npts + i]);
          if (t == 1) a *= sqrt(P->base[(size_t)2 * npts + i]);
        } else if (flow) {
          centre = P->base[i];
          a = half_line ? 0.0 : slab_a[t];
        } else {
          double c2 = P->base[(size_t)1 * npts + i], vA2 = P->base[(size_t)2 * npts + i];
          centre = 0.0;
          a = sqrt(t == 0 ? c2 : (t == 1 ? c2 * vA2 / (c2 + vA2) : vA2));
        }
        if (a > 0.0 || half_line) never = 0;
        double lo = half_line ? -INFINITY : centre - a, hi = centre + a;
        if (!isfinite(hi) || (cyl && !(a > 0.0))) P->use_bands = 0;
        if (i > 0 && !half_line && a > 0.0 && !(lo < hi_prev && lo_prev < hi)) P->use_bands = 0;
        lo_min = fmin(lo_min, lo); lo_max = fmax(lo_max, lo); hi_min = fmin(hi_min, hi); hi_max = fmax(hi_max, hi);
        lo_prev = lo; hi_prev = hi;
      }
      P->band[t][0] = lo_min; P->band[t][1] = lo_max; P->band[t][2] = hi_min; P->band[t][3] = hi_max;
      if (never) { P->band[t][0] = INFINITY; P->band[t][3] = -INFINITY; }
    }
  }
  return P;
}
static int band_crossed(const port_problem* P, double k, double w) {
  const double W = w / k;
  int c = 0;
  for (int t = 0; t < P->n_bands; ++t) {
    const int some = (W > P->band[t][0]) && (W < P->band[t][3]);
    const int all = (W > P->band[t][1]) && (W < P->band[t][2]);
    c |= (some && !all);
  }
  return c;
}
void port_destroy(port_problem* P) { if (P) { free(P->base); free(P); } }

typedef struct { double k, k2, m, m2, kc2, kvA2, kcT2, k4c; } kscal;
typedef struct { double a11, a12, a21, a22; } coef;
/* per watched term: "negative at some node" / "negative at every node" (same bookkeeping as the HIP SignTrack,
   which keeps the two flags of all 64 lanes of a wave in scalar lane masks) */
typedef struct { int some_neg[4], all_neg[4]; } strack;
static inline void st_add(strack* s, int i, double t) {
  const int neg = t < 0.0;
  s->some_neg[i] |= neg; s->all_neg[i] &= neg;
}
static inline int st_crossed(const strack* s) {
  return ((s->some_neg[0] & ~s->all_neg[0]) | (s->some_neg[1] & ~s->all_neg[1]) | (s->some_neg[2] & ~s->all_neg[2]) |
          (s->some_neg[3] & ~s->all_neg[3])) & 1;
}

static void make_entry(const port_problem* P, int pt, const kscal* s, double* e) {
  double b[11];
  for (int f = 0; f < P->nb; ++f) b[f] = P->base[(size_t)f * P->npts + pt];
  switch (P->family) {
    case 0: {
      double wA = s->k * b[1], wA2 = wA * wA;
      double g = s->m2 * b[5] + s->k2 * b[6];
      e[0] = s->k * b[0]; e[1] = wA2; e[2] = wA2 * b[2]; e[3] = b[3];
      /* a21 = -r C2/(rho S t1 t2) = -B + (lambda t2 + c0)/(t1 t2), lambda = g - B (wA^2 + wc^2), c0 = -B wc^4 */
      e[4] = -b[4]; e[5] = g - b[4] * (e[1] + e[2]); e[6] = -(b[4] * (e[2] * e[2]));
      /* scaled-coefficient form of the untwisted cylinder (make_entry<FAM_CYL0, true> of the HIP code): the entries
       * that become a12 and a21 carry h/2 */
      { const double h2 = 0.5 * P->h; e[3] *= h2; e[4] *= h2; e[5] *= h2; e[6] *= h2; }
    } break;
    case 1: {
      double r = b[0], invr = b[1], rho = b[2], S = b[3];
      double Bphi = b[7] * r, vphi = b[8] * r;
      double kb = fma(s->k, b[6], s->m * b[7]);
      double wA = fma(s->k, b[5], s->m * b[7]), wA2 = wA * wA, invr2 = invr * invr;
      e[0] = fma(s->k, b[9], s->m * b[8]); e[1] = wA2; e[2] = wA2 * b[4]; e[3] = rho * S; e[4] = rho;
      e[5] = rho * vphi * vphi * invr; e[6] = 2.0 * Bphi * Bphi * invr; e[7] = 2.0 * Bphi * vphi * kb * invr;
      e[8] = kb * Bphi; e[9] = rho * vphi; e[10] = 2.0 * s->m * S * invr2; e[11] = S * (s->m2 * invr2 + s->k2);
      e[12] = b[10]; e[13] = 4.0 * S * invr2; e[14] = r; e[15] = invr;
    } break;
    case 2: {
      double rho = b[0], c2 = b[1], vA2 = b[2], S = c2 + vA2, cT2 = c2 * vA2 / S;
      e[0] = s->k2 * c2; e[1] = s->k2 * cT2; e[2] = s->k2 * vA2; e[3] = rho * S; e[4] = rho;
    } break;
    default:
      e[0] = s->k * b[0]; e[1] = s->k * b[1]; e[2] = s->k * b[2]; e[3] = 2.0 * e[1];
  }
}

/* numerators and the one denominator of a node's coefficient matrix (see coef_finish) */
typedef struct { double n11, n12, n21, n22, den; } coefpre;

static void coef_pre(const port_problem* P, const double* e, const kscal* s, double w, coefpre* C, strack* st) {
  switch (P->family) {
    case 0: {
      double Om = w - e[0], t1 = fma(Om, Om, -e[1]), t2 = fma(Om, Om, -e[2]);
      if (st) { st_add(st, 0, t1); st_add(st, 1, t2); }
      C->n11 = 0.0;
      C->n12 = e[3] * t1;
      C->n21 = fma(e[5], t2, e[6]);
      C->n22 = e[4];
      C->den = t1 * t2;
    } break;
    case 1: {
      double Om = w - e[0], Om2 = Om * Om, t1 = Om2 - e[1], t2 = Om2 - e[2];
      st_add(st, 0, t1); st_add(st, 1, t2);
      /* product-sums as explicit fused multiply-adds, grouped as in the HIP coef_pre<FAM_CYLT> (round 3) */
      double D = e[3] * t1 * t2;
      double Q = fma(Om, e[7], fma(Om2, e[6], -(t1 * e[5])));
      double T = fma(e[9], Om, e[8]);
      double OmP = (P->c1_power == 2) ? Om2 : Om;
      double t2T = t2 * T;
      double C1 = fma(Q, OmP, -(e[10] * t2T));
      double C2 = fma(Om2, Om2, -(e[11] * t2));
      double C3 = fma(D, fma(e[4], t1, e[12]), fma(Q, Q, -((e[13] * t2T) * T)));
      st_add(st, 2, C3 * D);
      C->n11 = -C1; C->n22 = C1; C->n12 = C3 * e[15]; C->n21 = -(e[14] * C2); C->den = D;
    } break;
    case 2: {
      double w2 = w * w, n1 = e[0] - w2, n2 = e[1] - w2, n3 = e[2] - w2;
      if (st) { st_add(st, 0, n1); st_add(st, 1, n2); st_add(st, 2, n3); }
      C->n11 = 0.0; C->n22 = 0.0; C->n12 = n1; C->n21 = e[4] * n3; C->den = e[3] * n2;
    } break;
    default: {
      double Om = w - e[0], Om2 = Om * Om, t = Om2 - s->kcT2, n1 = s->kc2 - Om2, n3 = s->kvA2 - Om2;
      if (st) { st_add(st, 0, n1); st_add(st, 1, t); st_add(st, 2, n3); st_add(st, 3, Om); }
      /* m0, D, coeff of SF-G:416-427 over the common denominator S t Om^2 n1 (G = S t^2 + k^4 cT^2 c^2):
         D = -2 k U' G Om / den,  coeff = (k U'' S t Om n1 - 2 (k U')^2 G + n1^2 n3 Om^2) / den */
      double St = P->S_i * t, G = fma(St, t, s->k4c), X = Om * n1, OX = Om * X, g2 = e[3] * G;
      C->n11 = 0.0; C->n12 = 1.0;
      C->n22 = g2 * Om;
      C->n21 = -fma(e[2], St * X, fma(-g2, e[1], (n1 * n3) * OX));
      C->den = St * OX;
    }
  }
}

static void coef_finish(const port_problem* P, const coefpre* C, double inv, coef* A) {
  switch (P->family) {
    case 0: A->a11 = 0.0; A->a22 = 0.0; A->a12 = C->n12; A->a21 = fma(C->n21, inv, C->n22); break;
    case 1: A->a11 = C->n11 * inv; A->a22 = C->n22 * inv; A->a12 = C->n12 * inv; A->a21 = C->n21 * inv; break;
    case 2: A->a11 = 0.0; A->a22 = 0.0; A->a12 = C->n12 * inv; A->a21 = C->n21; break;
    default: A->a11 = 0.0; A->a12 = 1.0; A->a21 = C->n21 * inv; A->a22 = C->n22 * inv;
  }
}

static void coefficients(const port_problem* P, const double* e, const kscal* s, double w, coef* A, strack* st) {
  coefpre C;
  coef_pre(P, e, s, w, &C, st);
  coef_finish(P, &C, 1.0 / C.den, A);
}

/* mid-point and end-point of one step with a single division */
static void coefficients2(const port_problem* P, const double* em, const double* e1, const kscal* s, double w,
                          coef* Am, coef* A1, strack* st) {
  coefpre Cm, C1;
  coef_pre(P, em, s, w, &Cm, st);
  coef_pre(P, e1, s, w, &C1, st);
  double inv = 1.0 / (Cm.den * C1.den);
  coef_finish(P, &Cm, C1.den * inv, Am);
  coef_finish(P, &C1, Cm.den * inv, A1);
}

/* mid-point and end-point of TWO consecutive steps with a single division (coefficients4 of the HIP header: the product
   tree d12 = den_m den_1, d34 = den_m' den_1', inv = 1 / (d12 d34), 1/d12 = d34 inv, 1/d34 = d12 inv) */
static void coefficients4(const port_problem* P, const double* em, const double* e1, const double* em2, const double* e12,
                          const kscal* s, double w, coef* Am, coef* A1, coef* Am2, coef* A12, strack* st) {
  coefpre Cm, C1, Cn, C2;
  coef_pre(P, em, s, w, &Cm, st);
  coef_pre(P, e1, s, w, &C1, st);
  coef_pre(P, em2, s, w, &Cn, st);
  coef_pre(P, e12, s, w, &C2, st);
  const double d12 = Cm.den * C1.den, d34 = Cn.den * C2.den;
  const double inv = 1.0 / (d12 * d34);
  const double i12 = d34 * inv, i34 = d12 * inv;
  coef_finish(P, &Cm, C1.den * i12, Am);
  coef_finish(P, &C1, Cm.den * i12, A1);
  coef_finish(P, &Cn, C2.den * i34, Am2);
  coef_finish(P, &C2, Cn.den * i34, A12);
}

/* adjoint right-hand side A^T z and one RK4 step of the row-vector march (see rk4_step_adjoint in the HIP header) */
static inline void rhs_t(int diag, const coef* A, double p, double q, double* kp, double* kq) {
  if (diag) { *kp = fma(A->a11, p, A->a21 * q); *kq = fma(A->a22, q, A->a12 * p); }
  else { *kp = A->a21 * q; *kq = A->a12 * p; }
}
/* rk4_step_adjoint_scaled0 of the HIP code: off-diagonal A pre-multiplied by h/2 */
static void rk4_adjoint_scaled0(double* p, double* q, const coef* B0, const coef* Bm, const coef* B1) {
  const double tp1 = fma(B0->a21, *q, *p), tq1 = fma(B0->a12, *p, *q);
  const double tp2 = fma(Bm->a21, tq1, *p), tq2 = fma(Bm->a12, tp1, *q);
  const double am2 = Bm->a21 + Bm->a21, bm2 = Bm->a12 + Bm->a12;
  const double tp3 = fma(am2, tq2, *p), tq3 = fma(bm2, tp2, *q);
  /* t1 + 2 t2 + (t3 - z), t3 - z = 2 Am^T t2 taken as the product (as the HIP step) */
  const double sp = fma(am2, tq2, fma(2.0, tp2, tp1));
  const double sq = fma(bm2, tp2, fma(2.0, tq2, tq1));
  /* rk4_step_adjoint_scaled0_x3: no division by 3 (D depends on z_p / z_q only); the factor 3 per step is taken back by
     an exact power of two at the end of every chunk of PORT_CH steps, as in the HIP kernels */
  *p = fma(B1->a21, tq3, sp);
  *q = fma(B1->a12, tp3, sq);
}

static void rk4_adjoint(int diag, double* p, double* q, const coef* B0, const coef* Bm, const coef* B1, double h,
                        double h2, double h6, double h3) {
  double k1p, k1q, k2p, k2q, k3p, k3q, k4p, k4q, tp, tq;
  rhs_t(diag, B0, *p, *q, &k1p, &k1q);
  tp = fma(h2, k1p, *p); tq = fma(h2, k1q, *q);
  rhs_t(diag, Bm, tp, tq, &k2p, &k2q);
  tp = fma(h2, k2p, *p); tq = fma(h2, k2q, *q);
  rhs_t(diag, Bm, tp, tq, &k3p, &k3q);
  tp = fma(h, k3p, *p); tq = fma(h, k3q, *q);
  rhs_t(diag, B1, tp, tq, &k4p, &k4q);
  *p = fma(h6, k1p + k4p, fma(h3, k2p + k3p, *p));
  *q = fma(h6, k1q + k4q, fma(h3, k2q + k3q, *q));
}

typedef struct { double m_e, cst, yb, dyb, Oe; int status; } exterior;

static exterior ext_cyl(const port_problem* P, double k, double w, double w_cst) {
  exterior X;
  double k2 = k * k, w2 = w * w;
  X.Oe = w;
  X.m_e = ((k2 * P->vAe2 - w2) * (k2 * P->ce2 - w2)) / (P->Se * (k2 * P->cTe2 - w2));
  X.cst = -1.0 / (P->rho_e * (k2 * P->vAe2 - w_cst * w_cst));
  X.yb = X.dyb = NAN;
  if (X.m_e < 0.0) { X.status = ES_PT_LEAKY; return X; }
  if (!(X.m_e > 0.0) || !isfinite(X.m_e) || !isfinite(X.cst)) { X.status = ES_PT_NONFINITE; return X; }
  X.status = ES_PT_OK;
  double mu = sqrt(X.m_e), sgn = (P->xb < 0.0) ? -1.0 : 1.0;
  double xR = mu * (P->R_factor / k), xb = mu;
  int n = P->m_ext;
  double Kb, Kb1, KR, KR1, dn = (double)n;
  ke_pair(n, xb, &Kb, &Kb1);
  ke_pair(n, xR, &KR, &KR1);
  double dKb = -Kb1 + (dn / xb) * Kb, dKR = -KR1 + (dn / xR) * KR;
  double g = P->ic1 / (sgn * mu), Pv, dPv, gap = xR - xb;
  if (gap < 40.0) {
    double Ib, Ib1, IR, IR1;
    ie_pair_from_k(n, xb, Kb, Kb1, &Ib, &Ib1);
    ie_pair_from_k(n, xR, KR, KR1, &IR, &IR1);
    double dIb = Ib1 + (dn / xb) * Ib, dIR = IR1 + (dn / xR) * IR;
    double a_s = -(P->ic0 * dKR - g * KR), b_s = -(g * IR - P->ic0 * dIR), E2 = exp(-2.0 * gap);
    Pv = b_s * Kb + E2 * a_s * Ib;
    dPv = sgn * mu * (b_s * dKb + E2 * a_s * dIb);
  } else {
    double rI = 1.0 - 0.5 / xR - (4.0 * dn * dn - 1.0) / (8.0 * xR * xR);
    double b_s = -(g - P->ic0 * rI);
    Pv = b_s * Kb;
    dPv = sgn * mu * (b_s * dKb);
  }
  double nrm = fabs(Pv);
  X.yb = Pv / nrm; X.dyb = dPv / nrm;
  if (!isfinite(X.yb) || !isfinite(X.dyb)) X.status = ES_PT_NONFINITE;
  return X;
}

static exterior ext_slab(const port_problem* P, double k, double w) {
  exterior X;
  double k2 = k * k, Oe = w - k * P->U_e, Oe2 = Oe * Oe;
  X.Oe = Oe;
  X.m_e = ((k2 * P->vAe2 - Oe2) * (k2 * P->ce2 - Oe2)) / (P->Se * (k2 * P->cTe2 - Oe2));
  X.cst = P->rho_e * P->Se * (k2 * P->cTe2 - Oe2) / (Oe * (k2 * P->ce2 - Oe2));
  X.yb = X.dyb = NAN;
  if (X.m_e < 0.0) { X.status = ES_PT_LEAKY; return X; }
  if (!(X.m_e > 0.0) || !isfinite(X.m_e) || !isfinite(X.cst)) { X.status = ES_PT_NONFINITE; return X; }
  X.status = ES_PT_OK;
  double mu = sqrt(X.m_e), R = P->R_factor / k, E2 = exp(-2.0 * mu * (R - 1.0));
  double gp = P->ic0 + P->ic1 / mu, gm = P->ic0 - P->ic1 / mu;
  double V = gp + E2 * gm, dV = mu * (gp - E2 * gm), nrm = fabs(V);
  X.yb = V / nrm; X.dyb = dV / nrm;
  if (!isfinite(X.yb) || !isfinite(X.dyb)) X.status = ES_PT_NONFINITE;
  return X;
}

/* One determinant evaluation.  Returns status; *D, *rel as the product defines them.  w_cst: frequency at which the
 * exterior constant xi_e_const is taken (= w except inside CR-SF's locate_sausage, see oracle/workers.py). */
int port_eval2(const port_problem* P, double k, double w, double w_cst, double* D, double* rel);
int port_eval(const port_problem* P, double k, double w, double* D, double* rel) { return port_eval2(P, k, w, w, D, rel); }
static int port_eval_core(const port_problem* P, double k, double w, double w_cst, const double* ext_override,
                          double* D, double* rel, double* outer_out, double* inner_out);
int port_eval2(const port_problem* P, double k, double w, double w_cst, double* D, double* rel) {
  return port_eval_core(P, k, w, w_cst, NULL, D, rel, NULL, NULL);
}
/* The same evaluation with the exterior END STATE (value, slope at the boundary) supplied by the caller instead of
 * the closed form -- used by tests/test_reference_agreement.py to feed the reference's own LSODA exterior end state
 * (golden fixtures) to this interior, which separates the reference's exterior error from its interior noise.
 * Also returns outer / inner (xi_e, xi_i resp. P_e, P_i, normalised by |value|). */
int port_eval_ext(const port_problem* P, double k, double w, double w_cst, double ext_value, double ext_slope,
                  double* D, double* rel, double* outer, double* inner) {
  const double ov[2] = {ext_value, ext_slope};
  return port_eval_core(P, k, w, w_cst, ov, D, rel, outer, inner);
}
int port_eval_parts(const port_problem* P, double k, double w, double w_cst, double* D, double* rel, double* outer,
                    double* inner) {
  return port_eval_core(P, k, w, w_cst, NULL, D, rel, outer, inner);
}
static int port_eval_core(const port_problem* P, double k, double w, double w_cst, const double* ext_override,
                          double* D, double* rel, double* outer_out, double* inner_out) {
  kscal s;
  s.k = k; s.k2 = k * k; s.m = (double)P->m; s.m2 = s.m * s.m;
  s.kc2 = s.k2 * P->c2_i; s.kvA2 = s.k2 * P->vA2_i; s.kcT2 = s.k2 * P->cT2_i;
  s.k4c = s.k2 * s.k2 * P->cT2_i * P->c2_i;
  const int diag = (P->family == 1 || P->family == 3);
  const int nsteps = P->n_nodes - 1;
  const double h = P->h, h2 = 0.5 * P->h, h6 = P->h / 6.0, h3 = P->h / 3.0;
  double e[16], e2[16], zp, zq;
  strack trk = {{0, 0, 0, 0}, {-1, -1, -1, -1}};
  coef B0, Bm, B1;
  /* adjoint march: one row of the transfer matrix, from the last node back to the boundary */
  make_entry(P, 2 * nsteps, &s, e);
  strack* tp = (P->family != 1 && P->use_bands) ? NULL : &trk;
  coefficients(P, e, &s, w, &B0, tp);
  if (P->family <= 1 && P->axis_bc == ES_AXIS_SAUSAGE) { zp = B0.a11; zq = B0.a12; } else { zp = 1.0; zq = 0.0; }
  for (int j = nsteps - 1; j >= 0; --j) {
    if (P->family != 1 && (j & 1)) {
      /* fam_rcp4 of the HIP header (every family but the twisted cylinder): step j with step j - 1 for every odd j, one
         division for both (an even top step alone) */
      double e1[16], em2[16];
      coef Bm2, B2;
      make_entry(P, 2 * j + 1, &s, e);
      make_entry(P, 2 * j, &s, e1);
      make_entry(P, 2 * j - 1, &s, em2);
      make_entry(P, 2 * j - 2, &s, e2);
      coefficients4(P, e, e1, em2, e2, &s, w, &Bm, &B1, &Bm2, &B2, tp);
      if (P->family == 0) {
        rk4_adjoint_scaled0(&zp, &zq, &B0, &Bm, &B1);
        rk4_adjoint_scaled0(&zp, &zq, &B1, &Bm2, &B2);
      } else {
        rk4_adjoint(diag, &zp, &zq, &B0, &Bm, &B1, h, h2, h6, h3);
        rk4_adjoint(diag, &zp, &zq, &B1, &Bm2, &B2, h, h2, h6, h3);
      }
      B0 = B2;
      --j;                                             /* two steps taken: j is now the even step of the pair */
    } else {
    make_entry(P, 2 * j + 1, &s, e);
    make_entry(P, 2 * j, &s, e2);
    coefficients2(P, e, e2, &s, w, &Bm, &B1, tp);
    if (P->family == 0) rk4_adjoint_scaled0(&zp, &zq, &B0, &Bm, &B1);
    else rk4_adjoint(diag, &zp, &zq, &B0, &Bm, &B1, h, h2, h6, h3);
    B0 = B1;
    }
    if (P->family == 0 && j % PORT_CH == 0) {          /* end of an LDS chunk of the HIP march: adjoint_rescale */
      const int nst = (nsteps - j < PORT_CH) ? (nsteps - j) : PORT_CH;
      const int ex = port_rescale_exp(nsteps - j - nst, nsteps - j);
      zp = ldexp(zp, ex); zq = ldexp(zq, ex);
    }
  }
  exterior X = (P->family <= 1) ? ext_cyl(P, k, w, w_cst) : ext_slab(P, k, w);
  if (ext_override && X.status == ES_PT_OK) {
    const double nrm = fabs(ext_override[0]);
    X.yb = ext_override[0] / nrm; X.dyb = ext_override[1] / nrm;
  }
  double outer, inner;
  if (P->family <= 1) {
    double Pb = X.yb, xi_e = X.cst * X.dyb, Xb;
    if (P->axis_bc == ES_AXIS_KINK) Xb = (P->bc_const * xi_e - zp * Pb) / zq;
    else if (P->axis_bc == ES_AXIS_ROTATION_KINK) Xb = (-(P->bc_const * xi_e) - zp * Pb) / zq;
    else Xb = -(zp * Pb) / zq;
    outer = xi_e; inner = Xb / P->xb;
  } else {
    double P_left = X.cst * X.dyb;
    if (P->family == 2) {
      double sv = (P->slab_sign - zp) * X.yb / zq;
      inner = sv / w;
    } else {
      double Omb = w - e2[0], Vb = X.yb * Omb / X.Oe, sv = (P->slab_sign - zp) * Vb / zq, Omb2 = Omb * Omb;
      double PTi = P->rho_i * P->S_i * (s.kcT2 - Omb2) / (Omb * (s.kc2 - Omb2));
      inner = PTi * sv;
    }
    outer = P_left;
  }
  double d = outer - inner;
  int st = X.status;
  double sc = P->accept_norm ? fabs(outer) : fmax(fabs(outer), fabs(inner));
  *D = d;
  *rel = fabs(d) * 100.0 / sc;
  if (outer_out) *outer_out = outer;
  if (inner_out) *inner_out = inner;
  if (X.status != ES_PT_OK) { *D = NAN; *rel = NAN; return st; }
  if (!isfinite(d)) return ES_PT_NONFINITE;
  if (tp ? st_crossed(&trk) : band_crossed(P, k, w)) st = ES_PT_CONTINUUM;
  return st;
}

static double pick_w(const double* wv, int w_mode, double k, long row, int nw, int iw) {
  if (w_mode == ES_W_PHASE_SPEED) return k * wv[iw];
  if (w_mode == ES_W_PER_ROW) return wv[row * nw + iw];
  return wv[iw];
}

void port_eval_points(const port_problem* P, const double* k, const double* w, long n, double* D, double* rel,
                      uint8_t* st, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 64)
  for (long i = 0; i < n; ++i) {
    double d, r;
    st[i] = (uint8_t)port_eval(P, k[i], w[i], &d, &r);
    D[i] = d;
    if (rel) rel[i] = r;
  }
}

void port_eval_grid(const port_problem* P, const double* k, int nk, const double* w, int nw, int w_mode, double* D,
                    double* rel, uint8_t* st, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
  for (long c = 0; c < (long)nk * nw; ++c) {
    long row = c / nw; int iw = (int)(c - row * nw);
    double d, r;
    st[c] = (uint8_t)port_eval(P, k[row], pick_w(w, w_mode, k[row], row, nw, iw), &d, &r);
    D[c] = d;
    if (rel) rel[c] = r;
  }
}

/* Grid search: brackets (rows outer, omega inner), n_bisect bisection steps, classification. Returns the count. */
long port_find_roots(const port_problem* P, const double* k, int nk, const double* w, int nw, int w_mode,
                     const double* D, const uint8_t* st, int n_bisect, double tol, double* out_k, double* out_w,
                     double* out_lo, double* out_hi, double* out_resid, int32_t* out_row, uint8_t* out_flag,
                     long capacity, int nthreads) {
  long count = 0;
  long* cells = (long*)malloc(sizeof(long) * (size_t)(capacity > 0 ? capacity : 1));
  for (long row = 0; row < nk; ++row)
    for (int j = 0; j + 1 < nw; ++j) {
      long c = row * nw + j;
      if (st[c] == ES_PT_OK && st[c + 1] == ES_PT_OK && D[c] * D[c + 1] < 0.0) {
        if (count < capacity) cells[count] = c;
        ++count;
      }
    }
  long n = count < capacity ? count : capacity;
  int rounds = 0;
  const int polish = 2;
  /* (LANES+1)-section as the HIP refine_kernel: the fixed rule kRefineSections = 17 (it must not depend on the bracket
     count of the call, or a tiled grid would be refined differently from the whole one) */
  int sections = 17;
  { const char* ev = getenv("ES_REFINE_SECTIONS"); const int v = ev ? atoi(ev) : 0; if (v == 5 || v == 9 || v == 17) sections = v; }
  for (double span = 1.0, need = ldexp(1.0, n_bisect < 1000 ? n_bisect : 1000); span < need; span *= (double)sections) ++rounds;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 4)
  for (long i = 0; i < n; ++i) {
    long c = cells[i], row = c / nw; int j = (int)(c - row * nw);
    double kk = k[row], lo = pick_w(w, w_mode, kk, row, nw, j), hi = pick_w(w, w_mode, kk, row, nw, j + 1);
    double flo = D[c], fhi = D[c + 1], d = NAN, r = NAN;
    int s = ES_PT_NONFINITE;
    /* multi-section rounds, as the HIP refine_kernel: points lo + (hi-lo)*(j+1)/sections, first sign change from the left */
    const int L = sections - 1;
    for (int it = 0; it < rounds; ++it) {
      double x[64], dv[64];
      int first = L;
      for (int j = 0; j < L; ++j) {
        x[j] = lo + (hi - lo) * ((double)(j + 1) / (double)sections);
        port_eval(P, kk, x[j], &dv[j], &r);
      }
      for (int j = 0; j < L; ++j) if (dv[j] * flo < 0.0) { first = j; break; }
      double nlo = lo, nflo = flo;
      if (first > 0) { nlo = x[first - 1]; nflo = (dv[first - 1] == dv[first - 1]) ? dv[first - 1] : flo; }
      if (first < L) { hi = x[first]; fhi = dv[first]; }
      lo = nlo; flo = nflo;
    }
    /* regula-falsi polish (ES_REFINE_POLISH = 2 steps in the HIP kernel) */
    double root = lo + (hi - lo) * 0.5;
    if (polish == 0) s = port_eval(P, kk, root, &d, &r);
    for (int it = 0; it < polish; ++it) {
      double x = lo - flo * (hi - lo) / (fhi - flo);
      /* as the HIP refine_kernel: a secant point on / outside an end keeps the end with the smaller |f| */
      if (!(x > lo && x < hi)) x = (x == x) ? ((fabs(flo) <= fabs(fhi)) ? lo : hi) : lo + (hi - lo) * 0.5;
      s = port_eval(P, kk, x, &d, &r);
      root = x;
      if (d * flo < 0.0) { hi = x; fhi = d; } else if (d == d) { lo = x; flo = d; }
    }
    out_k[i] = kk; out_w[i] = root; out_lo[i] = lo; out_hi[i] = hi; out_resid[i] = r; out_row[i] = (int32_t)row;
    out_flag[i] = (s == ES_PT_OK && r < tol) ? 1 : 0;
  }
  free(cells);
  return count;
}
