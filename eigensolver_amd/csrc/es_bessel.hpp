// fp64 modified Bessel functions of integer order for the exterior (uniform-medium) solution of the
// cylinder workers:  P_e(r) = a I_m(mu |r|) + b K_m(mu |r|)  (the closed form of the reference's LSODA solve of
// P'' = -P'/r + (m_e + m^2/r^2) P, e.g. Cylinder_method_flow_testing.py:773-777).
//
// Exponentially scaled forms are used throughout:  Ke_n(x) = e^x K_n(x),  Ie_n(x) = e^-x I_n(x).
//   * K_0, K_1:  x <= 2  ascending series (A&S 9.6.12-13 form with harmonic numbers);
//                x >  2  Chebyshev series of sqrt(x) e^x K_{0,1}(x) in 4/x - 1 (Steed's CF2 until round 2);
//                then the upward recurrence K_{n+1} = K_{n-1} + (2n/x) K_n (stable).
//   * I_n, I_{n+1}: ascending power series (all terms positive, no cancellation); only called for x < ~50
//                where the I-admixture of the far-field initial values is not below rounding.
// Accuracy (tests/test_hostmath.py, against scipy.special.kve / ive): <= 4e-16 relative on x in [1e-6, 700].
#pragma once
#include <math.h>

#ifndef ES_HD
#if defined(__HIPCC__)
#define ES_HD __host__ __device__ __forceinline__
#else
#define ES_HD static inline
#endif
#endif

namespace esb {

// a / b for the series and continued-fraction loops.  On the device: v_rcp_f64 seed + one third-order correction
// (r0 (1 + e + e^2), e = 1 - b r0; within 1 ulp of 1/b) and one multiply, 5 instructions instead of the 11 of the
// IEEE division sequence; the quotient is within ~1.5 ulp.  On the host (tests/hostmath, CPU builds): plain division.
ES_HD double qdiv(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ES_IEEE_DIVISION)
  const double r0 = __builtin_amdgcn_rcp(b);
  const double e = __builtin_fma(-b, r0, 1.0);
  const double t = __builtin_fma(e, e, e);
  return a * __builtin_fma(r0, t, r0);
#else
  return a / b;
#endif
}

// Reciprocals of the x-independent divisors of the loops below -- 1/k (series) and 1/i, 1/a_i with a_i = -(i - 1/2)^2
// (CF2) -- from constant memory on the device: the loop index is wave-uniform, so they arrive by scalar loads, and a
// multiply replaces a reciprocal sequence (v_rcp_f64 is quarter rate: 16 issue cycles + 4 instructions each).  1/k is the
// correctly rounded quotient the host computes; beyond the table, and in the ES_IEEE_DIVISION build, the quotient itself.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ES_IEEE_DIVISION)
#define ES_BESSEL_TABLES 1
#define ES_R(k) (1.0 / (double)(k))
#define ES_R8(k) ES_R(k), ES_R((k) + 1), ES_R((k) + 2), ES_R((k) + 3), ES_R((k) + 4), ES_R((k) + 5), ES_R((k) + 6), ES_R((k) + 7)
#define ES_A(i) (1.0 / (-(((double)(i)) - 0.5) * (((double)(i)) - 0.5)))
#define ES_A8(i) ES_A(i), ES_A((i) + 1), ES_A((i) + 2), ES_A((i) + 3), ES_A((i) + 4), ES_A((i) + 5), ES_A((i) + 6), ES_A((i) + 7)
constexpr int kRcpTable = 64;
static __constant__ double kRcpInt[kRcpTable] = {0.0, ES_R(1), ES_R(2), ES_R(3), ES_R(4), ES_R(5), ES_R(6), ES_R(7), ES_R8(8), ES_R8(16),
                                                 ES_R8(24), ES_R8(32), ES_R8(40), ES_R8(48), ES_R8(56)};
static __constant__ double kRcpCF2a[kRcpTable] = {0.0, ES_A(1), ES_A(2), ES_A(3), ES_A(4), ES_A(5), ES_A(6), ES_A(7), ES_A8(8), ES_A8(16),
                                                  ES_A8(24), ES_A8(32), ES_A8(40), ES_A8(48), ES_A8(56)};
#undef ES_R
#undef ES_R8
#undef ES_A
#undef ES_A8
#endif

// a / k for a small positive integer k (wave-uniform)
ES_HD double qdiv_int(double a, int k) {
#if defined(ES_BESSEL_TABLES)
  if (k < kRcpTable) return a * kRcpInt[k];
#endif
  return qdiv(a, (double)k);
}
// a / a_i, a_i = -(i - 1/2)^2 (the second CF2 coefficient sequence; `ai` is its value, used beyond the table)
ES_HD double qdiv_cf2a(double a, int i, double ai) {
#if defined(ES_BESSEL_TABLES)
  if (i < kRcpTable) return a * kRcpCF2a[i];
#endif
  return qdiv(a, ai);
}

constexpr double kEulerGamma = 0.57721566490153286060651209008240243;
constexpr double kPi = 3.14159265358979323846264338327950288;

// sqrt(x) e^x K_0(x) and sqrt(x) e^x K_1(x) on x >= 2 as Chebyshev series in y = 4/x - 1 (25 terms: the 26th is below 2e-18;
// coefficients by interpolation at 96 Chebyshev nodes in 50-digit arithmetic, mpmath).  Evaluated by Clenshaw's recurrence
// the pair is within 2.8e-16 of the true values on [2, 700] (tests/test_hostmath.py against scipy.special.kve) at about
// 110 instructions, where Steed's CF2 needed 10 ... 40 iterations of 25 instructions and a reciprocal each.
#define ES_CHEB_K0 { \
    1.2201515410329777, -0.0314481013119645, 0.0015698838857300533, -0.00012849549581627802, \
    1.39498137188765e-05, -1.8317555227191195e-06, 2.766813639445015e-07, -4.660489897687948e-08, \
    8.574034017414225e-09, -1.6975345093890614e-09, 3.5773972814003283e-10, -7.957489244477396e-11, \
    1.8559491149549264e-11, -4.514597883374519e-12, 1.1403405882073441e-12, -2.9800969231481784e-13, \
    8.032890775068375e-14, -2.2275133267462965e-14, 6.340076476276646e-15, -1.848593377920907e-15, \
    5.5120559994043335e-16, -1.6782311257549006e-16, 5.2103917776435543e-17, -1.6475805939842632e-17, \
    5.3004337711773354e-18 }
#define ES_CHEB_K1 { \
    1.3603130952422213, 0.10392373657681724, -0.002857816859622779, 0.00019521551847135162, \
    -1.936197974166083e-05, 2.406484947837217e-06, -3.5019606030878126e-07, 5.7410841254500495e-08, \
    -1.0345762465678097e-08, 2.0150497551970347e-09, -4.1903547593419254e-10, 9.218315187605315e-11, \
    -2.129967838427791e-11, 5.139639673482343e-12, -1.2891739609498229e-12, 3.348419666052243e-13, \
    -8.976705182010146e-14, 2.4771544242195988e-14, -7.0198370892147685e-15, 2.038703166239861e-15, \
    -6.057047270643018e-16, 1.8380935752430455e-16, -5.689462849193648e-17, 1.7940510478863572e-17, \
    -5.7567444820733025e-18 }
constexpr int kChebN = 25;
#if defined(__HIP_DEVICE_COMPILE__)
static __constant__ double kChebK0[kChebN] = ES_CHEB_K0;     // wave-uniform index: scalar loads
static __constant__ double kChebK1[kChebN] = ES_CHEB_K1;
#else
static const double kChebK0[kChebN] = ES_CHEB_K0;
static const double kChebK1[kChebN] = ES_CHEB_K1;
#endif

// scaled K_0, K_1 (e^x K)
ES_HD void ke01(double x, double& k0, double& k1) {
  if (x <= 2.0) {
    const double t = 0.25 * x * x;
    const double lg = log(0.5 * x) + kEulerGamma;
    // I0 = sum t^k/(k!)^2 ; S0 = sum_{k>=1} H_k t^k/(k!)^2
    // I1/(x/2) = sum t^k/(k!(k+1)!) ; S1 = sum (H_k + H_{k+1}) t^k/(k!(k+1)!)
    double term0 = 1.0, i0 = 1.0, s0 = 0.0, hk = 0.0;
    double term1 = 1.0, i1 = 1.0, s1 = 1.0;      // k = 0: H_0 + H_1 = 1
    for (int k = 1; k < 40; ++k) {
      const double rk = qdiv_int(1.0, k), rk1 = qdiv_int(1.0, k + 1);
      term0 = term0 * t * (rk * rk);
      hk += rk;
      i0 += term0;
      s0 += hk * term0;
      term1 = term1 * t * (rk * rk1);
      i1 += term1;
      s1 += (hk + hk + rk1) * term1;
      if (term0 < 1e-18 * i0) break;
    }
    const double ex = exp(x);
    k0 = ex * (-lg * i0 + s0);
    k1 = ex * (qdiv(1.0, x) + lg * (0.5 * x) * i1 - 0.25 * x * s1);
  } else {
    // Chebyshev series of sqrt(x) e^x K_{0,1}(x) in y = 4/x - 1, Clenshaw: b_j = 2 y b_{j+1} - b_{j+2} + c_j
    const double y = qdiv(4.0, x) - 1.0, y2 = y + y;
    double p1 = 0.0, p2 = 0.0, q1 = 0.0, q2 = 0.0;
#pragma unroll
    for (int j = kChebN - 1; j >= 1; --j) {
      const double pn = fma(y2, p1, kChebK0[j] - p2);
      const double qn = fma(y2, q1, kChebK1[j] - q2);
      p2 = p1; p1 = pn;
      q2 = q1; q1 = qn;
    }
    const double rs = qdiv(1.0, sqrt(x));
    k0 = fma(y, p1, kChebK0[0] - p2) * rs;
    k1 = fma(y, q1, kChebK1[0] - q2) * rs;
  }
}

// scaled K_n, K_{n+1} for integer n >= 0
ES_HD void ke_pair(int n, double x, double& kn, double& knp1) {
  double a, b;
  ke01(x, a, b);
  const double tox = qdiv(2.0, x);
  for (int j = 1; j <= n; ++j) {      // (a, b) = (K_{j-1}, K_j) -> (K_j, K_{j+1})
    const double c = a + (double)j * tox * b;
    a = b;
    b = c;
  }
  kn = a;
  knp1 = b;
}

// scaled I_n, I_{n+1} (e^-x I) by the ascending series; intended for x <~ 60
ES_HD void ie_pair(int n, double x, double& in_, double& inp1) {
  const double t = 0.25 * x * x;
  const double hx = 0.5 * x;
  // prefactor (x/2)^n / n!
  double pre = 1.0;
  for (int j = 1; j <= n; ++j) pre *= hx / (double)j;
  double term_a = 1.0, sum_a = 1.0;      // order n:   sum t^k / (k! (n+1)_k)
  double term_b = 1.0, sum_b = 1.0;      // order n+1: sum t^k / (k! (n+2)_k)
  for (int k = 1; k < 400; ++k) {
    const double kk = (double)k;
    term_a = term_a * t / (kk * (kk + (double)n));
    term_b = term_b * t / (kk * (kk + (double)n + 1.0));
    sum_a += term_a;
    sum_b += term_b;
    if (term_a < 1e-18 * sum_a) break;
  }
  const double ex = exp(-x);
  in_ = ex * pre * sum_a;
  inp1 = ex * pre * (hx / ((double)n + 1.0)) * sum_b;
}

// scaled I_n, I_{n+1} when the scaled K_n, K_{n+1} at the same argument are already known (they always are in the
// exterior solution): the ratio f = I_{n+1}/I_n comes from Miller's backward recurrence
// I_{k-1} = (2k/x) I_k + I_{k+1} started at M = n + 10 + sqrt(40 x) (I is the minimal solution, so the arbitrary
// start is forgotten; M is 6+ orders beyond what 3e-16 needs on x in [0.5, 700], tests/test_hostmath.py) -- one fma
// per order, no division -- and the normalisation from the Wronskian  I_n K_{n+1} + I_{n+1} K_n = 1/x  (unchanged
// by the e^{-x}, e^{x} scalings).  ~40 orders at x = 18 instead of 45 series terms with two divisions each.
ES_HD void ie_pair_from_k(int n, double x, double kn, double knp1, double& in_, double& inp1) {
  if (x < 0.5) { ie_pair(n, x, in_, inp1); return; }        // a handful of series terms; also keeps (2k/x)^M finite
  const int M = n + 10 + (int)sqrt(40.0 * x);
  const double tox = 2.0 / x;
  double ip = 0.0, ic = 1e-200;                              // I_{M+1}, I_M up to a common factor
  double ktox = (double)M * tox;
  for (int k = M; k > n; --k) {                              // (ip, ic) = (I_{k+1}, I_k) -> (I_k, I_{k-1})
    const double im = fma(ktox, ic, ip);
    ip = ic;
    ic = im;
    ktox -= tox;
  }
  const double f = ip / ic;                                  // I_{n+1} / I_n
  in_ = 1.0 / (x * fma(f, kn, knp1));
  inp1 = f * in_;
}

// J_n, J_{n+1}, Y_n, Y_{n+1} for integer n >= 0, x > 0 (body modes of the uniform cylinder: m_i < 0).
// One Miller backward recurrence J_{k-1} = (2k/x) J_k - J_{k+1} from an even start index M > x gives all J_k up to a
// common factor, fixed by 1 = J_0 + 2 sum_{k>=1} J_{2k}; Y_0 and Y_1 follow from the Neumann series
//   (pi/2) Y_0 = (ln(x/2)+gamma) J_0 + 2 sum_{k>=1} (-1)^{k+1} J_{2k}/k                       (A&S 9.1.88)
//   (pi/2) Y_1 = -J_0/x + (ln(x/2)+gamma-1) J_1 - sum_{k>=1} (-1)^k (2k+1) J_{2k+1}/(k(k+1))  (A&S 9.1.89, n = 1)
// accumulated during the same recurrence, and Y_n by the (stable) upward recurrence.
ES_HD void jy_pair(int n, double x, double& jn, double& jn1, double& yn, double& yn1) {
  int M = (int)(1.15 * x) + 36 + n;
  M += (M & 1);                                   // even
  const double tox = 2.0 / x;
  double jp = 0.0, jc = 1e-280;                   // J_{M+1}, J_M (arbitrary scale)
  double norm = 0.0, sy0 = 0.0, sy1 = 0.0;        // sum J_{2k}; sum (-1)^{k+1} J_{2k}/k; sum (-1)^k (2k+1) J_{2k+1}/(k(k+1))
  double rn = 0.0, rn1 = 0.0;
  for (int k = M; k >= 1; --k) {                  // jc = J_k on entry
    if (k == n) rn = jc;
    if (k == n + 1) rn1 = jc;
    const int h = k >> 1;
    const double sgn = (h & 1) ? -1.0 : 1.0;      // (-1)^h
    if ((k & 1) == 0) {                           // k = 2h, h >= 1
      norm += jc;
      sy0 -= sgn * jc / (double)h;                // (-1)^{h+1} J_{2h}/h
    } else if (k >= 3) {                          // k = 2h+1, h >= 1
      sy1 += sgn * (double)k * jc / ((double)h * (double)(h + 1));
    }
    const double jm = (double)k * tox * jc - jp;  // J_{k-1}
    jp = jc;
    jc = jm;
    if (fabs(jc) > 1e250) {                       // rescale everything accumulated so far
      const double sc = 1e-250;
      jc *= sc; jp *= sc; norm *= sc; sy0 *= sc; sy1 *= sc; rn *= sc; rn1 *= sc;
    }
  }
  // now jc = J_0, jp = J_1 (unnormalised)
  if (n == 0) rn = jc;
  if (n + 1 == 1) rn1 = jp;
  const double inv = 1.0 / (jc + 2.0 * norm);
  const double j0 = jc * inv, j1 = jp * inv;
  const double lg = log(0.5 * x) + kEulerGamma;
  const double y0 = (2.0 / kPi) * (lg * j0 + 2.0 * sy0 * inv);
  const double y1 = (2.0 / kPi) * (-j0 / x + (lg - 1.0) * j1 - sy1 * inv);
  jn = rn * inv;
  jn1 = rn1 * inv;
  double a = y0, b = y1;                          // (Y_{j-1}, Y_j) -> (Y_j, Y_{j+1})
  for (int j = 1; j <= n; ++j) {
    const double c = (double)j * tox * b - a;
    a = b;
    b = c;
  }
  yn = a;
  yn1 = b;
}

}  // namespace esb
