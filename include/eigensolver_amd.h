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
  ES_ERR_UNSUPPORTED = 5,
  ES_ERR_EVAL_CAP = 6,     /* es_worker_run: a task exceeded its evaluation bound; its root list is incomplete   */
  ES_ERR_SCREENING = 7     /* es_shoot_find_roots_mixed: an fp32-screened bracket was not confirmed in fp64      */
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
/* Measurement aid (bench.py `roofline`): with the timer on, every launch of a grid-march kernel (the fp64 kernel of
 * es_shoot_eval_grid[_ex], the fp32 screening kernel of es_shoot_find_roots_mixed) is bracketed by HIP events on the
 * context's stream.  es_context_grid_time synchronises the stream, returns the summed elapsed time and the number of
 * launches since the last call and forgets them.  The reference has no counterpart (it times whole runs with
 * time.time(), e.g. Density_cylinder.py:1129). */
int es_context_grid_timer(es_context* ctx, int enable);
int es_context_grid_time(es_context* ctx, double* h_total_ms, int* h_launches);
/* sizeof() of the ABI structs as this library was compiled, for binding self-checks:
 * which = 0 es_slab_analytic_params, 1 es_shoot_desc, 2 es_profiles, 3 es_root_table, 4 es_worker_spec,
 * 5 es_cyl_uniform_params; -1 for an unknown index. */
int es_abi_sizeof(int which);

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


/* ========================================================================================================
 * (2) Shooting evaluation of the boundary determinant D(k, omega) for non-uniform interiors.
 *     Replaces, per (k, omega), the body of the reference workers (coefficients -> exterior ODE -> interior
 *     shoot -> mismatch), e.g. Cylinder_method_flow_testing.py:694-804 (kink) / :991-1111 (sausage),
 *     multiprocessor_Inhomogeneous_method.py:421-501, flow_multiprocessor_coronal.py:400-480,
 *     Twisted_photospheric_nonlinear_flow_kink_fast.py:601-712.
 *     D is the reference's mismatch (xi_e - xi_i for cylinders, P_e - P_i for slabs) divided by the exterior
 *     amplitude |P_e(boundary)| resp. |Vx_e(boundary)|, sign of the reference's amplitude kept.
 * ====================================================================================================== */
enum { ES_GEOM_CYLINDER = 0, ES_GEOM_CYLINDER_TWIST = 1, ES_GEOM_SLAB_DENSITY = 2, ES_GEOM_SLAB_FLOW = 3 };
enum { ES_AXIS_KINK = 0, ES_AXIS_SAUSAGE = 1, ES_AXIS_ROTATION_KINK = 2 };
enum { ES_SLAB_MODE_SAUSAGE = 0, ES_SLAB_MODE_KINK = 1 };
enum { ES_W_ABSOLUTE = 0,      /* omega = w[iw]                       (one frequency vector for all k)     */
       ES_W_PHASE_SPEED = 1,   /* omega = k * w[iw]                   (the reference's bands speeds*k)      */
       ES_W_PER_ROW = 2 };     /* omega = w[ik * nw + iw]             (one frequency array per task)        */

/* Everything a reference worker captures from module globals. */
typedef struct es_shoot_desc {
  int32_t geometry;        /* ES_GEOM_*                                                                   */
  int32_t n_nodes;         /* interior nodes N: the reference's `ix` grid (linspace(x_b, x_end, N)); the
                              propagator takes one RK4 step per interval                                    */
  double x_boundary;       /* first node: -1 (CD-C, CF, slabs) or +1 (CD-P, CR-*)                          */
  double x_end;            /* last node: -/+ r_axis for cylinders (0.001 / 0.01), +1 for slabs             */
  /* exterior medium */
  double rho_e, vA_e, c_e, cT_e, U_e;
  double L_factor;         /* far field at |x| = L_factor * 2 pi / k   (3 or 7)                             */
  double ic_value, ic_slope; /* reference's P0 / V0 = [1e-8, 1e-8] or [1e-8, 1e-15]                          */
  /* cylinder */
  int32_t m;               /* azimuthal order used in the interior coefficient set                          */
  int32_t m_ext;           /* order hard-coded in the reference's exterior ODE (1 kink, 0 sausage)          */
  int32_t axis_bc;         /* ES_AXIS_*                                                                     */
  int32_t c1_power;        /* C1 = Q*Omega (1: CD-C:590) or Q*Omega^2 (2: CF:598, CR-KF:493)                */
  double bc_const;         /* kink: B_phi(x_b)^2 ; rotation: B_phi(1)^2 - rho(1) v_phi(1)^2                 */
  /* slab */
  int32_t slab_mode;       /* ES_SLAB_MODE_*                                                                */
  int32_t accept_norm;     /* 0: rel = 100|d|/max(|outer|,|inner|) (all workers) ; 1: 100|d|/|outer| (CR-KS:722)   */
  double c_i, vA_i, rho_i; /* uniform interior speeds of the flow slab (SF-U / SF-G)                        */
} es_shoot_desc;

/* Profile samples on the 2N-1 points x_j = x_boundary + j*(x_end - x_boundary)/(2N-2)  (nodes and midpoints),
 * host pointers, each of length 2N-1; unused ones may be NULL:
 *   cylinder : r (the x_j themselves), rho, c2 (= c_i^2), Bz, Bphi, vz, vphi, rdC3 (= r d/dr[(Bphi/r)^2 - rho (vphi/r)^2])
 *   slab dens: rho, c2, vA2
 *   slab flow: U, dU, ddU                                                                                   */
typedef struct es_profiles {
  const double* r; const double* rho; const double* c2; const double* vA2;
  const double* Bz; const double* Bphi; const double* vz; const double* vphi; const double* rdC3;
  const double* U; const double* dU; const double* ddU;
} es_profiles;

int es_problem_create(es_context* ctx, const es_shoot_desc* desc, const es_profiles* h_profiles, es_problem** out);
int es_problem_destroy(es_context* ctx, es_problem* prob);

/* D[ik*nw + iw], status[ik*nw + iw] (ES_PT_*), optional rel[ik*nw + iw] = 100*|d|/max(|outer|,|inner|)
 * (the reference's acceptance measure, e.g. Cylinder_method_flow_testing.py:817).  One grid point per lane,
 * radial-profile coefficients staged in LDS per k-row. */
int es_shoot_eval_grid(es_context* ctx, const es_problem* prob, const double* d_k, int nk,
                       const double* d_w, int nw, int w_mode,
                       double* d_D, double* d_rel /* may be NULL */, uint8_t* d_status);

/* es_shoot_eval_grid with options.  flags = 0 is es_shoot_eval_grid.
 * ES_EVAL_SKIP_CONTINUUM: points whose status is ES_PT_CONTINUUM get D = rel = NaN instead of the value a march
 * through the singular layer gives (the reference integrates through it with LSODA and returns integrator noise,
 * which the grid search never brackets -- es_shoot_find_roots requires both ends ES_PT_OK).  Where the flag can be
 * decided before the march (families with connected continuum bands in phase speed) such points
 * are not marched at all; with ES_W_PHASE_SPEED whole omega-columns inside a band are removed from the launch
 * (ordered compaction of the live columns on the device, no host synchronisation).  D and rel are identical to
 * flags = 0 at every point whose status is not ES_PT_CONTINUUM; statuses are identical except that a continuum point
 * whose march would have overflowed (ES_PT_NONFINITE with flags = 0) is reported ES_PT_CONTINUUM. */
enum { ES_EVAL_SKIP_CONTINUUM = 1 };
int es_shoot_eval_grid_ex(es_context* ctx, const es_problem* prob, const double* d_k, int nk,
                          const double* d_w, int nw, int w_mode, int flags,
                          double* d_D, double* d_rel /* may be NULL */, uint8_t* d_status);

/* Measurement aid: the launch shape es_shoot_eval_grid selects for rows of nw frequencies of this problem -- points per
 * lane, waves per SIMD of the register cap, per-node sign tracking on / off -- i.e. the instantiation
 * shoot_grid_kernel<family, pts, 256, track, wpe> that bench.py prices against profiles/isa_loop_counts.json.  A NEGATIVE
 * *h_pts = -p names the shape with two k-rows per workgroup, shoot_grid_kernel_r2<family, p, track, wpe> (rows of at most
 * 512 frequencies of the untwisted cylinder and the slabs, at least two rows). */
int es_shoot_grid_shape(es_context* ctx, const es_problem* prob, int nw, int* h_pts, int* h_wpe, int* h_track);

/* The same determinant at n arbitrary (k, omega) pairs (one pair per lane, no shared k). */
int es_shoot_eval_points(es_context* ctx, const es_problem* prob, const double* d_k, const double* d_w, int n,
                         double* d_D, double* d_rel /* may be NULL */, uint8_t* d_status);

/* Root table of the grid search (structure of arrays, caller allocated, `capacity` entries each). */
typedef struct es_root_table {
  double* d_k;        /* wavenumber of the row                                       */
  double* d_w;        /* refined omega                                               */
  double* d_w_lo;     /* bracket [w_lo, w_hi] from the grid                          */
  double* d_w_hi;
  double* d_resid;    /* rel = 100 |d| / max(|outer|, |inner|) at the refined omega  */
  int32_t* d_row;     /* row index ik                                                */
  uint8_t* d_flag;    /* 1 = accepted root (resid < tol), 0 = sign change at a pole / continuum edge */
  int32_t capacity;
} es_root_table;

/* Grid search: brackets = sign changes of D between omega-neighbours of the same row with both ends ES_PT_OK
 * (wavefront shuffle + ballot, ordered compaction: rows outer, omega inner); each bracket is narrowed at least
 * as far as `n_bisect` bisection steps would (the reference's 3-point linspace refinement, e.g. :823-829, run to
 * convergence; executed as rounds of 17-section with 16 lanes per bracket -- a fixed rule, so that the table of a grid
 * does not depend on how the grid is tiled over calls or GPUs -- as many as shrink the bracket by 2^n_bisect, always keeping the sign change nearest to the lower end), then polished in fp64 by two regula-falsi steps (the secant through the
 * bracket ends: the Newton-type refinement of the north star, without a derivative of D) and classified with the
 * reference's acceptance rule rel < tol_percent at the last secant point, which is the root reported;
 * [w_lo, w_hi] is the final bracket around it.
 * d_D / d_status must hold the output of es_shoot_eval_grid for the same inputs. */
int es_shoot_find_roots(es_context* ctx, const es_problem* prob, const double* d_k, int nk,
                        const double* d_w, int nw, int w_mode, const double* d_D, const uint8_t* d_status,
                        int n_bisect, double tol_percent, es_root_table* table, int* h_count);

/* es_shoot_find_roots without any host synchronisation (pipelined callers, k-tiles of a multi-GPU run: a 512-row tile
 * is 3 ms of GPU work, a read-back in the middle of it is a tenth of that).  The bracket count is written to the
 * caller's device word d_count (it may exceed table->capacity: then only `capacity` brackets were written and refined --
 * the caller checks when it reads the count, as es_shoot_find_roots does for it).  The refinement launches are sized for
 * table->capacity and take the count from device memory, so size the table for the data (about twice the expected
 * count), not for the worst case.  Same table, bit for bit, as es_shoot_find_roots. */
int es_shoot_find_roots_async(es_context* ctx, const es_problem* prob, const double* d_k, int nk,
                              const double* d_w, int nw, int w_mode, const double* d_D, const uint8_t* d_status,
                              int n_bisect, double tol_percent, es_root_table* table, int32_t* d_count);

/* Mixed-precision grid search (BASELINE.json configs[4]: "fp32 bracket + fp64 refine"; the reference itself is fp64
 * throughout), for the cylinder families and (round 3) for the slab families whose continuum flag comes from phase-speed bands
 * (every profile of the reference; a slab profile whose node intervals do not overlap needs per-node sign tracking, which
 * exists in fp64 only: ES_ERR_UNSUPPORTED).  What is GUARANTEED: every bracket
 * it reports is an fp64 bracket (both ends re-evaluated in fp64, ES_ERR_SCREENING otherwise) and is refined exactly as
 * es_shoot_find_roots refines it.  What is EMPIRICAL: that no fp64 bracket is missed -- a sign change between two points
 * fp32 judged "sure" (|D| > 5e-2 of the scale, every watched coefficient term more than 1e-3 away from zero) would go
 * unnoticed; the thresholds are supported by measurement, not by an error bound (largest fp32 error among vouched-for points
 * 4.5e-3 of the scale over 3 800 random problems, tools/fuzz_mixed.py; every point of configs[4] in
 * tests/test_full_size_parity_gpu.py and tools/full_size_parity.py): in all of them the bracket set is identical and the root
 * table bit-identical to es_shoot_eval_grid + es_shoot_find_roots.
 *   1. the (k, omega) grid is marched in fp32 (exterior and boundary algebra in fp64); points at which fp32 cannot vouch
 *      for the sign of D or for the status (|D| < 5e-2 of max(|outer|, |inner|), a pole of D nearby, a coefficient within
 *      1e-3 of a singular point at some node, non-finite result) are marked and re-evaluated in fp64;
 *   2. brackets are detected on the merged array, BOTH ends of every bracket are re-evaluated in fp64 (the determinant
 *      signs at bracket endpoints are fp64 signs) -- a bracket these values do not confirm makes the call return
 *      ES_ERR_SCREENING;
 *   3. refinement and classification in fp64 exactly as es_shoot_find_roots.
 * d_D / d_status (nk x nw, caller allocated) receive the screening result: fp64 values at the re-evaluated points, fp32-
 * accurate values elsewhere (not defined at ES_PT_CONTINUUM points of the band families).
 * h_stats (optional, 3 ints): fp64 re-evaluations of unsure grid points, of bracket ends, unconfirmed brackets. */
int es_shoot_find_roots_mixed(es_context* ctx, const es_problem* prob, const double* d_k, int nk,
                              const double* d_w, int nw, int w_mode, int n_bisect, double tol_percent,
                              double* d_D, uint8_t* d_status, es_root_table* table, int* h_count, int* h_stats);

/* The two halves of es_shoot_find_roots_mixed as separate calls (same result when called one after the other on the same
 * arrays): es_shoot_screen_grid enqueues step 1, the fp32 screening march (nothing read back); es_shoot_find_roots_screened
 * runs steps 2 - 5 on the screened d_D / d_status.  A caller that runs several problems on several streams can then order
 * the throughput-bound screening launches one after the other and let the latency-bound remainder of one problem run under
 * the screening of the next (bench.py --workload config4). */
int es_shoot_screen_grid(es_context* ctx, const es_problem* prob, const double* d_k, int nk, const double* d_w, int nw,
                         int w_mode, double* d_D, uint8_t* d_status);
int es_shoot_find_roots_screened(es_context* ctx, const es_problem* prob, const double* d_k, int nk,
                                 const double* d_w, int nw, int w_mode, int n_bisect, double tol_percent,
                                 double* d_D, uint8_t* d_status, es_root_table* table, int* h_count, int* h_stats);

/* Send buffer of the multi-GPU exchange (one all-gather of fixed-capacity buffers per step):
 * d_out is (cap + 1) x 6 doubles, row 0 = (count, 0, ...), rows 1 .. min(count, cap) = (k, omega, m, resid, flag,
 * global row) of the first records of `table`, the rest zero.  d_rows_global[local row] maps the rows of a k-tile to
 * the rows of the whole grid (NULL: identity).  Replaces the reference's positional pairing of two Queues
 * (Density_cylinder.py:1155-1168).  Asynchronous on the context's stream. */
int es_root_table_pack(es_context* ctx, const es_root_table* table, int count, double m,
                       const int64_t* d_rows_global, int cap, double* d_out);
/* The same with the count in device memory (the d_count of es_shoot_find_roots_async). */
int es_root_table_pack_async(es_context* ctx, const es_root_table* table, const int32_t* d_count, double m,
                             const int64_t* d_rows_global, int cap, double* d_out);

/* ========================================================================================================
 * (3) The reference worker itself: kink(wavenumber, kink_ws, kink_ks, freq) / sausage(...) for a batch of
 *     (wavenumber, freq[]) tasks -- main loop over freq, acceptance test, sign-change detection against the
 *     previously evaluated point, recursive 3-point refinement locate_*() with all of the reference's
 *     bookkeeping (e.g. Cylinder_method_flow_testing.py:554-839; quirks listed in DESIGN.md "worker semantics").
 * ====================================================================================================== */
typedef struct es_worker_spec {
  double tol_percent;              /* xi_tol / p_tol / P_tol                                               */
  int32_t min_len;                 /* refinement needs len(ws) > min_len: 1 slabs (SF-U:518), 2 cylinders (CD-C:680) */
  int32_t itt_cap;                 /* `if itt_num > cap: break`  (100 ... 500)                              */
  int32_t reset_loop_ws_each_iter; /* slab sausage workers clear loop_ws at every main iteration (SF-U:536) */
  int32_t break_on_accept;         /* CR kink workers: `break` after the first accepted grid point (CR-KF:722) */
  int32_t stale_ext_const;         /* CR sausage workers: locate_sausage() uses the enclosing loop's xi_e_const, i.e. the
                                      value at the grid frequency that opened the bracket (CR-SF:558 vs :617)            */
  int32_t main_double_append;      /* CR sausage workers append freq[j] to all_ws twice per main-loop evaluation (CR-SF:684 and
                                      :726): len(all_ws) > 2 holds after two evaluations and the refinement interval
                                      linspace(all_ws[-2], all_ws[-1], 3) is the degenerate [w, w, w]                     */
} es_worker_spec;

/* Task t: wavenumber d_k[t], frequencies d_freq[t*nfreq .. t*nfreq+nfreq).  Roots of task t are written to
 * d_roots[t*max_roots ...] in the order the reference appends them; d_nroots[t] is their number (may exceed
 * max_roots, then ES_ERR_CAPACITY is returned).  Every task is bounded by 3 (itt_cap + 2)(nfreq + 1) evaluations (the
 * reference bounds the recursion by itt_cap only); a task that reaches the bound stops, is marked by a NEGATIVE
 * d_nevals[t] and makes the call return ES_ERR_EVAL_CAP -- its root list is incomplete, never silently truncated.
 * d_nevals[t] (optional) counts the determinant evaluations the
 * reference worker performs for the task (the library itself evaluates fewer points -- it does not re-evaluate the
 * end points of a refinement interval -- and, with several lanes per task, some it never uses). */
int es_worker_run(es_context* ctx, const es_problem* prob, const es_worker_spec* spec,
                  const double* d_k, int ntasks, const double* d_freq, int nfreq,
                  double* d_roots, int32_t* d_nroots, int max_roots, int32_t* d_nevals /* may be NULL */);

/* ========================================================================================================
 * (4) Uniform cylinder in closed form (the limit the reference uses as its benchmark case, profile width 1e5,
 *     e.g. Cylinder_method_flow_testing.py:126): the interior ODE is then Bessel's equation, so the same
 *     determinant follows from I_m/K_m (m_i > 0) or J_m/Y_m (m_i < 0, body modes) of sqrt(|m_i|) r with the
 *     K_m/Y_m admixture fixed by the reference's axis condition at r_axis, matched to the exterior K_m/I_m
 *     solution.  No ODE is integrated.  Same normalisation, status codes and rel as es_shoot_eval_grid.
 * ====================================================================================================== */
typedef struct es_cyl_uniform_params {
  double c_i, vA_i, rho_i, U_i;          /* uniform interior: sound speed, Alfven speed, density, axial flow */
  double rho_e, vA_e, c_e, cT_e;         /* exterior                                                        */
  double r_boundary;                     /* -1 or +1 (sign convention of the reference file)                */
  double r_axis;                         /* |r| of the inner end of the reference's ix grid (0.001 / 0.01)   */
  double L_factor, ic_value, ic_slope;   /* far field of the exterior solve                                  */
  int32_t m, m_ext, axis_bc;             /* ES_AXIS_KINK (P(r_ax) = 0) or ES_AXIS_SAUSAGE (P'(r_ax) = 0)      */
  int32_t reserved;
} es_cyl_uniform_params;

int es_cyl_uniform_eval(es_context* ctx, const es_cyl_uniform_params* p, const double* d_k, int nk,
                        const double* d_w, int nw, int w_mode,
                        double* d_D, double* d_rel /* may be NULL */, uint8_t* d_status);

/* ========================================================================================================
 * (5) Eigenfunctions at given (k, omega) -- the two-region solve the reference's analysis scripts repeat at a
 *     chosen root to plot P_T(r) and xi_r(r) (Cylinder/Non-uniform flow/Coronal/Eigenfunctions/
 *     analysis_cylinder_flow_coronal.py:813-924): interior on the problem's node grid (linspace(x_boundary,
 *     x_end, N)), exterior on linspace(-/+ L*2pi/k, -/+1, n_ext) in closed form.
 *     Cylinders: value = P, flux = xi_r (= xi_e_const P' outside, (C1 P + D P')/C3 inside); slabs: value = Vx,
 *     flux = total pressure P_T.  Both regions are scaled so that the exterior value at the boundary is +-1 (sign
 *     of the reference's amplitude); the reference's plot normalisation (division by max|exterior|) is a host-side
 *     step on these arrays.  Layout: [i * N + j] / [i * n_ext + j] for pair i, node j (node 0 = boundary for the
 *     interior arrays; exterior arrays run from the far field to the boundary).
 * ====================================================================================================== */
int es_shoot_eigenfunction(es_context* ctx, const es_problem* prob, const double* d_k, const double* d_w, int n,
                           double* d_int_value, double* d_int_flux,            /* n x N      */
                           int n_ext, double* d_ext_x, double* d_ext_value, double* d_ext_flux /* n x n_ext */);

/* ======================================================================================================
 * (6) Complex frequencies (unstable / Kelvin-Helmholtz modes of the flow slab) -- SURVEY 8f row 3.
 *     Replaces the determinant evaluation and the (Re omega, Im omega) scan of
 *       Slab/Non uniform flow/COMPLEX ANALYSIS/flow_multiprocessor_complex_coronal.py
 *         :348 / :737   sausage / kink(wavenumber, ws, ks, ws_imag, ks_imag, freq)   (6-argument workers)
 *         :369-404      m_e, p_e_const, m0, D, coeff, P_Ti, add_P_Ti with omega = omega_r + i omega_i
 *         :419-456      exterior / interior ODEs, boundary value, total pressures
 *         :1127         driver grid: Re(omega) over a phase-speed band x Im(omega) over [-0.25, 0.25]
 *     in its consistent reading (everything complex; the reference mixes real and imaginary parts, see
 *     DESIGN.md): D_c(k, omega) = p_e V_e'/V_e - P_Ti (Vx' - add Vx) at x = -1, the far-end condition
 *     Vx(+1) = -/+ Vx(-1) imposed by superposition, rel = 100 |D_c| / max(|outer|, |inner|).  Points with
 *     Re(m_e) < 0 are ES_PT_LEAKY (`if m_e.real < 0: pass`, SF-X:405).  Only for ES_GEOM_SLAB_FLOW problems.
 *     variant: ES_CX_SFX = the complex script's formulas (D of SF-X:382, P_T with the U' term of SF-X:401, :455);
 *              ES_CX_SFG = the real script's (D of flow_multiprocessor_coronal.py:421, no U' term): at
 *              Im(omega) = 0 this is es_shoot_eval_* up to the sign normalisation of the exterior amplitude.
 *     Grid layout: [(row * n_im + i_im) * n_re + i_re]; w_mode ES_W_ABSOLUTE: omega = w_re + i w_im;
 *     ES_W_PHASE_SPEED: omega = k (w_re + i w_im).
 *     es_complex_find_roots: a grid cell holds a root if D_c winds once around 0 along its four corners (quadrant
 *     count; cells with a non-finite corner are skipped); each such cell is refined by `n_iter` complex secant
 *     steps from the cell centre; flag = 1 if the final rel < tol_percent and the iterate stayed within two cell
 *     diagonals of the centre.  Ordered by (row, i_im, i_re).
 * ====================================================================================================== */
enum { ES_CX_SFX = 0, ES_CX_SFG = 1 };

int es_complex_eval_grid(es_context* ctx, const es_problem* prob, int variant, const double* d_k, int nk,
                         const double* d_w_re, int n_re, const double* d_w_im, int n_im, int w_mode,
                         double* d_D_re, double* d_D_im, double* d_rel /* may be NULL */, uint8_t* d_status);

int es_complex_eval_points(es_context* ctx, const es_problem* prob, int variant, const double* d_k,
                           const double* d_w_re, const double* d_w_im, int n, double* d_D_re, double* d_D_im,
                           double* d_rel /* may be NULL */, uint8_t* d_status);

typedef struct es_complex_root_table {
  double* d_k;
  double* d_w_re;      /* refined root */
  double* d_w_im;
  double* d_resid;     /* rel (percent) at the refined root */
  int32_t* d_row;
  int32_t* d_flag;     /* 1 accepted, 0 not converged / left the cell neighbourhood (e.g. a pole) */
  int32_t capacity;
} es_complex_root_table;

int es_complex_find_roots(es_context* ctx, const es_problem* prob, int variant, const double* d_k, int nk,
                          const double* d_w_re, int n_re, const double* d_w_im, int n_im, int w_mode,
                          const double* d_D_re, const double* d_D_im, const uint8_t* d_status, int n_iter,
                          double tol_percent, es_complex_root_table* table, int* out_count);

#ifdef __cplusplus
}
#endif
#endif /* EIGENSOLVER_AMD_H */
