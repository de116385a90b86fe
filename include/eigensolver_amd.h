/* eigensolver_amd.h -- C ABI of the MI355X (gfx950) dispersion-relation hot path.
 *
 * The reference (samuelskirvin/EIGENSOLVER) has no FFI: its operator boundary is the per-geometry worker
 *     sausage(wavenumber, sausage_ws, sausage_ks, freq) / kink(wavenumber, kink_ws, kink_ks, freq)
 * (e.g. Cylinder/Non-uniform flow/Coronal/solvers/Cylinder_method_flow_testing.py:554, :855) plus the analytic
 * scan at module level of Slab/Non uniform flow/Solver/flow_multiprocessor.py:107-303.  Everything those
 * functions capture from module globals is passed here as explicit POD structs / arrays.
 *
 * Conventions
 *  - every function returns an int status (ES_SUCCESS == 0); no exceptions cross the boundary;
 *  - pointers named d_* are DEVICE pointers (HBM), h_* are host pointers; all floating point is IEEE fp64;
 *  - all work is enqueued on the hipStream_t given at context creation (passed as void*); functions that
 *    return counts through host pointers synchronise that stream, the pure *_async entry points do not;
 *  - the library is GPU-only: there is no CPU fallback behind any entry point.
 */
#ifndef EIGENSOLVER_AMD_H
#define EIGENSOLVER_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ES_ABI_VERSION 1

/* ---- status codes of the library calls -------------------------------------------------------------- */
enum {
  ES_SUCCESS = 0,
  ES_ERR_INVALID_ARG = 1,
  ES_ERR_HIP = 2,          /* a HIP runtime call failed: see es_last_error()           */
  ES_ERR_CAPACITY = 3,     /* caller-provided output buffer too small (count is still returned) */
  ES_ERR_NO_DEVICE = 4,
  ES_ERR_UNSUPPORTED = 5
};

/* ---- per-point status written next to D(k, omega) ------------------------------------------------------
 * The reference silently skips m_e < 0 ("leaky", e.g. Cylinder_method_flow_testing.py:760) and lets inf/nan
 * propagate; here every lane reports why it has no usable determinant. */
enum {
  ES_PT_OK = 0,
  ES_PT_LEAKY = 1,        /* m_e < 0: reference skips the point                                  */
  ES_PT_NONFINITE = 2,    /* evaluated by the reference but the result is inf/nan (singular speed) */
  ES_PT_CONTINUUM = 3     /* Omega^2 crosses omega_A^2(r) or omega_c^2(r) inside the domain        */
};

typedef struct es_context es_context;   /* opaque: device, stream, scratch workspace */
typedef struct es_problem es_problem;   /* opaque: one worker configuration + its profile tables in HBM */

/* ---- context ------------------------------------------------------------------------------------------ */
int es_abi_version(void);
const char* es_status_string(int status);
/* device: HIP device ordinal; stream: hipStream_t (NULL = default stream of that device). */
int es_context_create(int device, void* stream, es_context** out);
int es_context_destroy(es_context* ctx);
const char* es_last_error(const es_context* ctx);
int es_context_synchronize(es_context* ctx);

/* ========================================================================================================
 * (1) Analytic slab dispersion relations with steady flow and their sign-change scan.
 *     Replaces flow_multiprocessor.py:107-127 (m0, me, n0, disp_rel_*) and :166-272 (scan loops),
 *     :284-303 (one-sided pole filter).
 * ====================================================================================================== */
typedef struct es_slab_analytic_params {
  double vA_i, c_i, vA_e, c_e;   /* flow_multiprocessor.py:63-66                                   */
  double mach_i, mach_e;         /* :97-98  (U_i, U_e, not divided by vA_i)                         */
  double R1;                     /* :79     rho_e / rho_i                                           */
  double cT_i, cT_e;             /* :85-89  the *normalised* tube speeds exactly as the script computes them */
} es_slab_analytic_params;

enum { ES_SLAB_SAUSAGE = 0, ES_SLAB_KINK = 1, ES_SLAB_SAUSAGE_BODY = 2, ES_SLAB_KINK_BODY = 3 };

/* D[iK * nW + iW] = disp_rel_<mode>(W[iW], K[iK]).  One grid point per lane. */
int es_slab_analytic_eval(es_context* ctx, const es_slab_analytic_params* p, int mode,
                          const double* d_K, int nK, const double* d_W, int nW, double* d_D);

/* Scan loops :166-272: for every K (outer) and V in W (inner): f(V,K) * f(V+step,K) < 0  ->  root (K, (V+V+step)/2).
 * Roots are written in the reference's loop order.  *h_count receives the number found (may exceed capacity,
 * then ES_ERR_CAPACITY is returned and only `capacity` roots are written). */
int es_slab_analytic_scan(es_context* ctx, const es_slab_analytic_params* p, int mode,
                          const double* d_K, int nK, const double* d_W, int nW, double step,
                          double* d_rootK, double* d_rootW, int capacity, int* h_count);

/* Pole filter :284-303: keep[i] = disp_rel_<mode>(rootW[i], rootK[i]) < thresh (one-sided, as written). */
int es_slab_analytic_filter(es_context* ctx, const es_slab_analytic_params* p, int mode,
                            const double* d_rootK, const double* d_rootW, int n, double thresh,
                            uint8_t* d_keep);

#ifdef __cplusplus
}
#endif
#endif /* EIGENSOLVER_AMD_H */
