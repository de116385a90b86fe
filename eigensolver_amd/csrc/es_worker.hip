// The reference worker  kink(wavenumber, kink_ws, kink_ks, freq) / sausage(...)  as a device-resident state machine,
// one task (wavenumber, freq[]) per lane.
//
// Restates the main loop and the recursive locate_*() of the workers, e.g.
//   Cylinder_method_flow_testing.py:556-694 (locate_kink), :702-829 (main loop)   [cylinders: len(ws) > 2]
//   flow_multiprocessor.py:452-526, :533-625                                       [slabs:     len(ws) > 1]
//   Twisted_photospheric_nonlinear_flow_kink_fast.py:456-596, :601-734             [break after first accepted point]
// with the module-global history lists (xi_diff_check, xi_diff_loop_check, all_ws, loop_ws) kept as per-task state
// for ONE worker call, exactly as a freshly forked reference process sees them.
//
// Phase A (es_shoot_eval_grid, called by es_worker_run): D, rel and status at every freq[j] of every task, fully
// parallel (tasks x nfreq lanes).  Phase B (worker_kernel): the sequential logic; main-loop points are table
// look-ups, only the refinement points are evaluated.  The recursion of locate_*() becomes an explicit stack of
// frames {omega[3], kk, itt}; after a child returns the parent continues its loop with the re-bound omega and the
// incremented itt, as the Python code does.  All lanes of a wave call the determinant evaluation together (lanes
// without a pending refinement point evaluate a dummy frequency), so the expensive part stays convergent.
//
// Phase B is a chain of dependent evaluations per task (refinement level after level), so a run with few tasks is
// bound by the latency of one evaluation.  A task therefore owns L = 2^l lanes (l chosen on the host from the task
// count): when the state machine needs the mid-point of an interval, the L lanes evaluate the whole binary tree of
// mid-points l levels deep below that interval (lane i the node with heap index i + 1, formed with the same
// lo + (hi - lo) * 0.5 the state machine uses, hence bit-identical frequencies) and keep the results; the next
// l - 1 levels of the descent are then table look-ups (wave ballot on the frequency).  The L lanes of a task run the
// state machine redundantly on identical state; results and decisions are exactly those of one lane per task.
#include "es_shoot_shared.hpp"
#include <cstdlib>
#include <vector>

namespace {
using namespace es_shoot_shared;

// One activation of locate_*(): omega[3], loop index kk, itt_num.  The reference re-evaluates the two end points of
// every refinement interval although it has just evaluated them (same inputs, same result); their mismatch, acceptance
// measure and status are carried in the frame instead (cached = 1), so that only the mid-point costs an evaluation.
struct Frame {
  double w0, w1, w2;
  double d0, rel0, d2, rel2;
  int kk, itt, cached;
  uint8_t st0, st2;
};

struct WorkerArgs {
  double tol;
  int min_len, itt_cap, reset_loop_ws_each_iter, break_on_accept, stale_ext_const, main_double_append;
  int ntasks, nfreq, max_roots, stack_depth;
  int log_lanes;                  // lanes per task = 1 << log_lanes (1 ... 64)
  const double* k;
  const double* freq;
  const double* D;
  const double* rel;
  const uint8_t* st;
  Frame* stack;
  double* roots;
  int32_t* nroots;
  int32_t* nevals;
  long eval_cap;                  // hard bound on the evaluations of one task
  int* abort_flag;                // set to 1 by any task that hit eval_cap (its root list is incomplete)
};

template <int FAM>
__global__ __launch_bounds__(64) void worker_kernel(ShootDev P, WorkerArgs a) {
  ES_POINT_LDS(FAM);
  const int logL = a.log_lanes, L = 1 << logL;
  const int grp = (int)threadIdx.x >> logL, sub = (int)threadIdx.x & (L - 1);
  const int t = blockIdx.x * (64 >> logL) + grp;                  // all L lanes of a group serve task t
  const bool live = t < a.ntasks;
  // this lane's entry of the group's speculation table
  double tw = 0.0, td = 0.0, trel = 0.0;
  int tst = 0;
  bool tvalid = false;
  // table look-up: is the frequency w held by a lane of this group?  (group-uniform control flow: the lanes of a
  // group are always active together, so the ballot sees the whole group)
  auto lookup = [&](double w, double& d, double& rel, uint8_t& st) -> bool {
    const unsigned long long m = __ballot(tvalid && tw == w);
    const unsigned long long gm = (L == 64) ? m : ((m >> (grp << logL)) & ((1ull << L) - 1ull));
    if (gm == 0ull) return false;
    const int src = (grp << logL) + (__ffsll((long long)gm) - 1);
    d = __shfl(td, src);
    rel = __shfl(trel, src);
    st = (uint8_t)__shfl(tst, src);
    return true;
  };
  const double k = live ? a.k[t] : 1.0;
  Frame* stk = a.stack + (size_t)(live ? t : 0) * a.stack_depth;
  // per-task history (what the reference keeps in module-global lists)
  int j = 0, sp = 0, nroots = 0, nevals = 0;
  double main_prev = 0.0, loop_prev = 0.0;           // *_diff_check[-1], *_diff_loop_check[-1] (start [0])
  int all_len = 0, loop_len = 0;                     // len(all_ws), len(loop_ws)
  double all_m1 = 0.0, all_m2 = 0.0, loop_m1 = 0.0, loop_m2 = 0.0;   // [-1], [-2]
  // mismatch / measure / status at those points (for the end-point cache of the next refinement interval)
  double all_d1 = 0.0, all_d2 = 0.0, all_r1 = 0.0, all_r2 = 0.0, loop_d1 = 0.0, loop_d2 = 0.0, loop_r1 = 0.0, loop_r2 = 0.0;
  uint8_t all_s1 = 0, all_s2 = 0, loop_s1 = 0, loop_s2 = 0;
  double w_stale = 1.0;                              // grid frequency that opened the current refinement (CR-SF:617)
  bool done = !live, aborted = false;
  // hard bound on the work of one task (every loop iteration below consumes one evaluation or pops a frame); a task
  // that reaches it is reported (negative nevals, ES_ERR_EVAL_CAP from es_worker_run), never silently truncated
  const long eval_cap = a.eval_cap;

  auto emit = [&](double w) {
    if (nroots < a.max_roots) a.roots[(size_t)t * a.max_roots + nroots] = w;
    ++nroots;
  };

  // one point of the loop inside locate_*() (CF:556-694), evaluated or taken from the frame's end-point cache; the
  // frame's kk has already been advanced
  auto loop_point = [&](double w, double d, double rel, uint8_t st) {
    Frame& f = stk[sp - 1];
    if ((long)nevals > eval_cap) { sp = 0; done = true; aborted = true; return; }
    if (st == ES_PT_LEAKY) return;                              // `if m_e < 0: pass`
    ++nevals;
    loop_m2 = loop_m1; loop_m1 = w; ++loop_len;                 // loop_ws.append(omega[k])
    loop_d2 = loop_d1; loop_d1 = d; loop_r2 = loop_r1; loop_r1 = rel; loop_s2 = loop_s1; loop_s1 = st;
    const double sign = d * loop_prev;                          // CF:678
    loop_prev = d;
    if (rel < a.tol) {                                          // CF:681-686
      emit(w);
      loop_len = 0;
      --sp;                                                     // break
    } else if (sign < 0.0 && loop_len > a.min_len) {            // CF:688-694
      const double lo = loop_m2, hi = loop_m1;
      f.w0 = lo; f.w1 = lo + (hi - lo) * 0.5; f.w2 = hi;        // omega re-bound in the caller's frame
      f.d0 = loop_d2; f.rel0 = loop_r2; f.st0 = loop_s2;        // both ends were evaluated inside locate_*()
      f.d2 = loop_d1; f.rel2 = loop_r1; f.st2 = loop_s1;
      f.cached = 1;
      f.itt += 1;
      loop_len = 0;
      if (sp < a.stack_depth) {
        Frame c = f;
        c.kk = 0;
        stk[sp] = c;
        ++sp;
      }
    }
  };

  for (;;) {
    // ---- advance without evaluating until a refinement point is needed --------------------------------------
    bool need = false, spec = false;
    double w_eval = 1.0, root_lo = 0.0, root_hi = 0.0;
    while (!done && !need) {
      if (sp == 0) {                                              // main loop over freq (CF:702)
        if (j >= a.nfreq) { done = true; break; }
        if (a.reset_loop_ws_each_iter) loop_len = 0;              // SF-U:536
        const size_t o = (size_t)t * a.nfreq + j;
        const double w = a.freq[o];
        const uint8_t st = a.st[o];
        const double d = a.D[o], rel = a.rel[o];
        ++j;
        if (st == ES_PT_LEAKY) continue;                          // `if m_e < 0: pass`
        ++nevals;
        all_m2 = all_m1; all_m1 = w; ++all_len;                   // all_ws.append(freq[j])
        all_d2 = all_d1; all_d1 = d; all_r2 = all_r1; all_r1 = rel; all_s2 = all_s1; all_s1 = st;
        if (a.main_double_append) {                               // ... and once more (CR-SF:684 and :726)
          all_m2 = w; ++all_len;
          all_d2 = d; all_r2 = rel; all_s2 = st;
        }
        const double sign = d * main_prev;                        // sign_check.append(d * check[-2])
        main_prev = d;
        if (rel < a.tol) {                                        // CF:817
          emit(w);
          all_len = 0;
          if (a.break_on_accept) done = true;                     // CR-KF:722
        } else if (sign < 0.0 && all_len > a.min_len) {           // CF:822-829
          // np.linspace(all_ws[-2], all_ws[-1], 3); the end points keep their main-loop values unless the worker
          // evaluates refinement points with a different exterior constant (CR-SF:617)
          Frame f{all_m2, all_m2 + (all_m1 - all_m2) * 0.5, all_m1, all_d2, all_r2, all_d1, all_r1, 0, 0,
                  a.stale_ext_const ? 0 : 1, all_s2, all_s1};
          all_len = 0;
          w_stale = w;
          if (a.stale_ext_const) tvalid = false;                  // table entries depend on w_stale
          stk[0] = f;
          sp = 1;
        }
      } else {                                                    // inside locate_*(): top frame
        Frame& f = stk[sp - 1];
        if (f.kk >= 3 || f.itt > a.itt_cap) { --sp; continue; }   // loop exhausted / `if itt_num > cap: break`
        if (f.cached && f.kk != 1) {                              // end point: values known, no evaluation
          const bool first = (f.kk == 0);
          ++f.kk;
          loop_point(first ? f.w0 : f.w2, first ? f.d0 : f.d2, first ? f.rel0 : f.rel2, first ? f.st0 : f.st2);
          continue;
        }
        w_eval = (f.kk == 0) ? f.w0 : (f.kk == 1 ? f.w1 : f.w2);
        {
          double dl, rl; uint8_t sl;
          if (lookup(w_eval, dl, rl, sl)) {                       // evaluated speculatively in an earlier round
            ++f.kk;
            loop_point(w_eval, dl, rl, sl);
            continue;
          }
        }
        spec = (f.kk == 1) && (L > 1);                            // mid-point: evaluate the tree below (w0, w2)
        root_lo = f.w0; root_hi = f.w2;
        need = true;
      }
    }
    if (!__any(need)) break;                        // workgroup = one wave: the exit is workgroup-uniform
    // ---- one determinant evaluation per lane (dummy for lanes that do not need one) ---------------------------
    double w_lane = w_eval;
    if (need && spec) {
      // node with heap index i = sub + 1 of the mid-point tree below (root_lo, root_hi); index L (last lane) is
      // outside the tree and repeats the root
      const int i = (sub + 1 < L) ? (sub + 1) : 1;
      const int level = 31 - __clz(i);
      double lo = root_lo, hi = root_hi;
      for (int b = level - 1; b >= 0; --b) {
        const double mid = lo + (hi - lo) * 0.5;
        if ((i >> b) & 1) lo = mid; else hi = mid;
      }
      w_lane = lo + (hi - lo) * 0.5;
    }
    double d, rel; uint8_t st;
    shoot_point<FAM>(P, k, w_lane, a.stale_ext_const ? w_stale : w_lane, d, rel, st, es_point_lds);
    if (need) {
      tw = w_lane; td = d; trel = rel; tst = st; tvalid = true;  // (re)build this group's table
      // the point the state machine asked for: the root of the tree (lane 0 of the group) or every lane's own
      const int src = (grp << logL);
      const double d0 = __shfl(d, src), r0 = __shfl(rel, src);
      const uint8_t s0 = (uint8_t)__shfl((int)st, src);
      ++stk[sp - 1].kk;
      loop_point(w_eval, d0, r0, s0);
    }
  }
  if (live) {
    a.nroots[t] = nroots;
    if (a.nevals) a.nevals[t] = aborted ? -nevals : nevals;
    if (aborted && sub == 0) atomicOr(a.abort_flag, 1);
  }
}

template <int FAM>
int launch_worker(es_context* ctx, const es_problem* prob, const WorkerArgs& a) {
  const int tasks_per_wg = 64 >> a.log_lanes;
  hipLaunchKernelGGL((worker_kernel<FAM>), dim3((a.ntasks + tasks_per_wg - 1) / tasks_per_wg), dim3(64), 0, ctx->stream,
                     prob->dev, a);
  ES_HIP_CHECK(ctx, hipGetLastError());
  return ES_SUCCESS;
}

}  // namespace

extern "C" int es_worker_run(es_context* ctx, const es_problem* prob, const es_worker_spec* spec, const double* d_k,
                             int ntasks, const double* d_freq, int nfreq, double* d_roots, int32_t* d_nroots,
                             int max_roots, int32_t* d_nevals) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  ES_REQUIRE(ctx, prob && spec, "null pointer");
  ES_REQUIRE(ctx, ntasks >= 0 && nfreq >= 0 && max_roots >= 0, "negative size");
  ES_REQUIRE(ctx, spec->min_len >= 0 && spec->itt_cap >= 0 && spec->itt_cap <= 100000, "worker spec");
  if (ntasks == 0) return ES_SUCCESS;
  ES_REQUIRE(ctx, d_k && d_nroots && (nfreq == 0 || d_freq) && (max_roots == 0 || d_roots), "null pointer");
  ES_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t cells = (size_t)ntasks * (size_t)(nfreq > 0 ? nfreq : 1);
  const int depth = spec->itt_cap + 3;
  // scratch of one call, carved out of the context's buffer: main-loop table (D, rel, status) + frame stacks
  auto align = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t bD = align(cells * sizeof(double)), bS = align(cells), bF = align((size_t)ntasks * depth * sizeof(Frame));
  int rc = es_ensure_scratch(ctx, 2 * bD + bS + bF);
  if (rc) return rc;
  char* base = (char*)ctx->d_scratch;
  double* D = (double*)base;
  double* rel = (double*)(base + bD);
  Frame* stack = (Frame*)(base + 2 * bD);
  uint8_t* st = (uint8_t*)(base + 2 * bD + bF);
  if (rc == ES_SUCCESS && nfreq > 0)
    rc = es_shoot_eval_grid(ctx, prob, d_k, ntasks, d_freq, nfreq, ES_W_PER_ROW, D, rel, st);
  if (rc == ES_SUCCESS) {
    WorkerArgs a;
    a.tol = spec->tol_percent; a.min_len = spec->min_len; a.itt_cap = spec->itt_cap;
    a.reset_loop_ws_each_iter = spec->reset_loop_ws_each_iter; a.break_on_accept = spec->break_on_accept;
    a.stale_ext_const = spec->stale_ext_const;
    a.main_double_append = spec->main_double_append;
    a.ntasks = ntasks; a.nfreq = nfreq; a.max_roots = max_roots; a.stack_depth = depth;
    // lanes per task: as many as keep the launch within ~2 waves per SIMD (1024 SIMDs x 64 lanes x 2)
    a.log_lanes = 6;
    while (a.log_lanes > 0 && ((long)ntasks << a.log_lanes) > 131072L) --a.log_lanes;
    if (const char* ev = getenv("ES_WORKER_LOG_LANES")) { const int v = atoi(ev); if (v >= 0 && v <= 6) a.log_lanes = v; }
    a.k = d_k; a.freq = d_freq; a.D = D; a.rel = rel; a.st = st; a.stack = stack;
    a.roots = d_roots; a.nroots = d_nroots; a.nevals = d_nevals;
    a.eval_cap = 3L * (spec->itt_cap + 2) * ((long)nfreq + 1);
    if (const char* ev = getenv("ES_WORKER_EVAL_CAP")) { const long v = atol(ev); if (v > 0) a.eval_cap = v; }   // test aid
    a.abort_flag = ctx->d_total;
    if (hipMemsetAsync(ctx->d_total, 0, sizeof(int), ctx->stream) != hipSuccess) rc = ES_ERR_HIP;
    if (rc == ES_SUCCESS)
    switch (prob->dev.family) {
      case FAM_CYL0: rc = launch_worker<FAM_CYL0>(ctx, prob, a); break;
      case FAM_CYLT: rc = launch_worker<FAM_CYLT>(ctx, prob, a); break;
      case FAM_SLABD: rc = launch_worker<FAM_SLABD>(ctx, prob, a); break;
      case FAM_SLABF: rc = launch_worker<FAM_SLABF>(ctx, prob, a); break;
      default: rc = ES_ERR_UNSUPPORTED;
    }
  }
  if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == ES_SUCCESS) {
    ctx->last_error = "worker kernel failed";
    rc = ES_ERR_HIP;
  }
  if (rc != ES_SUCCESS) return rc;
  ES_HIP_CHECK(ctx, hipMemcpy(ctx->h_total, ctx->d_total, sizeof(int), hipMemcpyDeviceToHost));
  if (*ctx->h_total != 0) {
    ctx->last_error = "a worker task exceeded its evaluation cap (root list incomplete; d_nevals < 0 marks the task)";
    return ES_ERR_EVAL_CAP;
  }
  // capacity check on the host (counts are small)
  std::vector<int32_t> h((size_t)ntasks);
  ES_HIP_CHECK(ctx, hipMemcpy(h.data(), d_nroots, (size_t)ntasks * sizeof(int32_t), hipMemcpyDeviceToHost));
  for (int t = 0; t < ntasks; ++t)
    if (h[t] > max_roots) return ES_ERR_CAPACITY;
  return ES_SUCCESS;
}
