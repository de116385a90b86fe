// K3/K4/K5: shooting evaluation of D(k, omega) on a (k, omega) grid, bracket detection, 17-section + secant refinement
// and ordered root compaction.  See es_shoot_device.hpp for the arithmetic and include/eigensolver_amd.h for the
// reference lines each entry point replaces.
//
// Kernel layout (DESIGN.md section "kernels"):
//  * shoot_grid_kernel<FAM, PTS, MAXT, TRACK, WPE>: one workgroup per tile = (k-row, omega-segment of T * PTS points), omega
//    along the lanes (PTS points per lane, strided by the workgroup size so the 8-byte D stores of a wave are one
//    contiguous 512 B segment).  k is workgroup uniform, so everything that depends on (node, k, m) but not on omega is
//    computed ONCE per tile into an LDS table, chunk by chunk (CH RK4 steps per chunk); every lane then reads the same
//    LDS address (broadcast).  Launch shapes per family: pick_shape(); the launch has exactly one workgroup per tile
//    (es_tile_grid).  Families with fam_rcp4() take two RK4 steps per division (coefficients4).
//  * shoot_points_kernel<FAM>: one (k, omega) pair per lane with unrelated k: the k-independent base table is
//    staged in LDS chunk by chunk (es_shoot_shared.hpp: shoot_point), node entries are formed per lane.
//  * bracket_flag_kernel: sign change against the omega-neighbour through __shfl_down (lane 63 reads the halo
//    element), ballot masks + per-block counts; bracket_emit_kernel writes the ordered bracket list.
//  * refine_kernel<FAM, LANES, CHR, SECTIONS_ONLY, ONE>: LANES = 16 lanes per bracket, 17-section rounds steered by a wave
//    ballot (uniform trip count -> no divergence); refine_polish_kernel: two secant steps with one lane per bracket and
//    the final classification with the reference's acceptance measure.  Both cylinder families launch one round / one step
//    per launch (ONE); the bracket count may stay on the device (es_shoot_find_roots_async).
//  * shoot_grid_f32_kernel: fp32 screening march of es_shoot_find_roots_mixed.
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <type_traits>

#include "es_shoot_shared.hpp"

namespace {
using namespace es_shoot_shared;

// ------------------------------------------------------------------------------------------------------------
// Options of one grid launch: skip = ES_EVAL_SKIP_CONTINUUM; cols != nullptr: only the omega-columns cols[1 .. cols[0]]
// are evaluated (live-column list built on the device by column_classify_kernel), the others were filled beforehand.
// part (compacted launches only): 0 = every segment; 1 = the full segments of a row; 2 = the columns behind the last
// full segment of width main_span, in segments of this launch's own (narrower) workgroups.  A partly filled segment in
// a 4-wave workgroup costs as much as a full one: its live wave shares a SIMD with two waves of other workgroups, and
// those workgroups are tied by their barriers to the pace of that SIMD, so the idle wave slots buy nothing; given to
// one-wave workgroups of a second launch, the remainder costs what its waves cost.
struct GridOpts {
  int skip;
  const int* cols;
  int part;
  int main_span;
};

// ONE tile per workgroup, the launch has as many workgroups as tiles (es_tile_grid: the y dimension only counts beyond
// 2^22 tiles).  The kernels are written as a loop over tiles that runs once: round 2 and most of round 3 launched at most
// 2^22 workgroups and let each stride over the tiles -- a loop the compiler hoisted every tile-invariant value out of
// (64-bit literals of the Bessel polynomials, kernel arguments, step sizes), and carried them in registers through the
// march: 172 -> 103 VGPRs for the fp32 kernel of the untwisted cylinder, 286 -> 94 SGPRs spilled to lanes and no scratch
// left in the twisted one once the loop was gone.
constexpr long ES_TILE_GRID_X = 1L << 22;
__device__ __forceinline__ long es_tile_index() { return (long)blockIdx.y * ES_TILE_GRID_X + blockIdx.x; }
inline dim3 es_tile_grid(long tiles) {
  return tiles <= ES_TILE_GRID_X ? dim3((unsigned)tiles) : dim3((unsigned)ES_TILE_GRID_X, (unsigned)((tiles + ES_TILE_GRID_X - 1) / ES_TILE_GRID_X));
}

// ROWS = 2: a workgroup holds TWO k-rows, waves 0, 1 the first and waves 2, 3 the second (128 lanes x PTS points each; its
// own LDS table per row).  A workgroup should have four waves, one per SIMD of its CU: the waves of a workgroup are tied to
// each other by the barriers around every LDS chunk, so a CU filled with three-wave workgroups runs at the pace of the SIMD
// that got four waves while another got two -- measured on 4096 rows, N = 2001, ns per point-step relative to rows of 1024
// frequencies (4 points x 256 lanes): 768 (4 x 192 lanes) 1.17, 384 (2 x 192) 1.17, 512 (4 x 128) 1.39, 256 (4 x 64) 1.83
// (tools/probe/time_row_width.py).  Rows of at most 128 x PTS frequencies therefore go two to a workgroup.
template <int FAM, int PTS, int MAXT, bool TRACK, int WPE = 0>
__global__ __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu(WPE ? WPE : 1, WPE ? WPE : 8)))
void shoot_grid_kernel(ShootDev P, const double* __restrict__ kv, int nk,
                                                          const double* __restrict__ wv, int nw, int w_mode,
                                                          double* __restrict__ Dout, double* __restrict__ relout,
                                                          uint8_t* __restrict__ stout, GridOpts opts) {
  constexpr int ROWS = 1;
#include "es_shoot_grid_body.hpp"
}

// two k-rows per workgroup (ROWS = 2): 256 threads, never compacted launches
template <int FAM, int PTS, bool TRACK, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void shoot_grid_kernel_r2(ShootDev P, const double* __restrict__ kv, int nk,
                                                          const double* __restrict__ wv, int nw, int w_mode,
                                                          double* __restrict__ Dout, double* __restrict__ relout,
                                                          uint8_t* __restrict__ stout, GridOpts opts) {
  constexpr int ROWS = 2;
  constexpr int MAXT = 256;
#include "es_shoot_grid_body.hpp"
}

// ---- ES_EVAL_SKIP_CONTINUUM with ES_W_PHASE_SPEED: whole omega-columns inside a continuum band ----------------------
// band_crossed() tests W' = (k W)/k, which is W up to the two roundings of the product and the quotient (|W' - W| <= 2
// ulp).  A column is removed from the launch only if W is inside a band by more than that for every k:
//     W - m > min lo  and  W + m < max hi  and  (W + m <= max lo  or  W - m >= min hi),      m = 4 ulp(W);
// columns within the margin of a band edge stay in the launch and are flagged point by point as before.
__device__ __forceinline__ bool column_dead(const ShootDev& P, double W) {
  const double m = 4.0 * 2.220446049250313e-16 * fabs(W);
  bool c = false;
  for (int t = 0; t < P.n_bands; ++t) {
    const bool some = (W - m > P.band[t][0]) && (W + m < P.band[t][3]);
    const bool notall = (W + m <= P.band[t][1]) || (W - m >= P.band[t][2]);
    c = c || (some && notall);
  }
  return c;
}

// One workgroup: dead flag per column + ordered list of the live columns (cols[0] = count, cols[1..] = indices).
__global__ __launch_bounds__(1024) void column_classify_kernel(ShootDev P, const double* __restrict__ Wv, int nw,
                                                               int* __restrict__ cols, uint8_t* __restrict__ dead) {
  __shared__ int wave_cnt[16];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int j0 = 0; j0 < nw; j0 += 1024) {
    const int j = j0 + tid;
    const bool in = j < nw;
    const bool dd = in && column_dead(P, Wv[j]);
    if (in) dead[j] = dd ? 1 : 0;
    const bool live = in && !dd;
    const uint64_t m = __ballot(live);
    if (lane == 0) wave_cnt[wid] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int v = 0; v < wid; ++v) off += wave_cnt[v];
    const uint64_t lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (live) cols[1 + off + __popcll(m & lower)] = j;
    __syncthreads();
    if (tid == 0) { int t = 0; for (int v = 0; v < 16; ++v) t += wave_cnt[v]; base += t; }
    __syncthreads();
  }
  if (tid == 0) cols[0] = base;
}

// D = rel = NaN, status = CONTINUUM at every point of a dead column (status of a leaky / singular exterior takes
// precedence, exactly as finish_point() orders them: the exterior is evaluated here, it is cheap next to a march).
__global__ __launch_bounds__(256) void fill_dead_columns_kernel(ShootDev P, const double* __restrict__ kv, int nk,
                                                                const double* __restrict__ wv, int nw,
                                                                const uint8_t* __restrict__ dead,
                                                                double* __restrict__ Dout, double* __restrict__ relout,
                                                                uint8_t* __restrict__ stout) {
  const long cells = (long)nk * nw;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (long)gridDim.x * blockDim.x) {
    const long row = c / nw;
    const int j = (int)(c - row * nw);
    if (!dead[j]) continue;
    const double k = kv[row];
    const double w = k * wv[j];
    // exterior status from m_e and the exterior constant alone (the Bessel evaluation is not needed for it)
    const double k2 = k * k;
    const double Oe = (P.family == FAM_CYL0 || P.family == FAM_CYLT) ? w : (w - k * P.U_e);
    const double Oe2 = Oe * Oe;
    const double m_e = ((k2 * P.vAe2 - Oe2) * (k2 * P.ce2 - Oe2)) / (P.Se * (k2 * P.cTe2 - Oe2));
    const double cst = (P.family == FAM_CYL0 || P.family == FAM_CYLT)
                           ? -1.0 / (P.rho_e * (k2 * P.vAe2 - w * w))
                           : P.rho_e * P.Se * (k2 * P.cTe2 - Oe2) / (Oe * (k2 * P.ce2 - Oe2));
    uint8_t st = ES_PT_CONTINUUM;
    if (m_e < 0.0) st = ES_PT_LEAKY;
    else if (!(m_e > 0.0) || !isfinite(m_e) || !isfinite(cst)) st = ES_PT_NONFINITE;
    Dout[c] = NAN;
    if (relout) relout[c] = NAN;
    stout[c] = st;
  }
}

template <int FAM>
__global__ __launch_bounds__(256) void shoot_points_kernel(ShootDev P, const double* __restrict__ kv,
                                                           const double* __restrict__ wv, int n,
                                                           double* __restrict__ Dout, double* __restrict__ relout,
                                                           uint8_t* __restrict__ stout) {
  ES_POINT_LDS(FAM);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = i < n;
  const double k = in ? kv[i] : 1.0;
  const double w = in ? wv[i] : 1.0;
  double D, rel; uint8_t st;
  shoot_point<FAM>(P, k, w, w, D, rel, st, es_point_lds);
  if (in) {
    Dout[i] = D;
    stout[i] = st;
    if (relout) relout[i] = rel;
  }
}

// ---- brackets ------------------------------------------------------------------------------------------------
// cell c = row*nw + j ; bracket if j < nw-1, both ends ES_PT_OK and D[c]*D[c+1] < 0 (sign product as in CF:806).
__global__ __launch_bounds__(256) void bracket_flag_kernel(const double* __restrict__ D,
                                                           const uint8_t* __restrict__ st, int nw, long cells,
                                                           uint64_t* __restrict__ masks,
                                                           int* __restrict__ block_counts) {
  __shared__ int wave_cnt[4];
  const long c = (long)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  double d0 = 0.0; int ok0 = 0;
  if (c < cells) { d0 = D[c]; ok0 = (st[c] == ES_PT_OK); }
  // neighbour in omega: lane+1 of the same wave, or the halo element for lane 63
  double d1 = __shfl_down(d0, 1);
  int ok1 = __shfl_down(ok0, 1);
  if (lane == 63) {
    if (c + 1 < cells) { d1 = D[c + 1]; ok1 = (st[c + 1] == ES_PT_OK); } else { d1 = 0.0; ok1 = 0; }
  }
  bool flag = false;
  if (c < cells) {
    const long row = c / nw;
    const int j = (int)(c - row * nw);
    flag = (j < nw - 1) && ok0 && ok1 && (d0 * d1 < 0.0);
  }
  const uint64_t m = __ballot(flag);
  if (lane == 0) {
    masks[c >> 6] = m;
    wave_cnt[threadIdx.x >> 6] = __popcll(m);
  }
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

__global__ __launch_bounds__(256) void bracket_emit_kernel(const double* __restrict__ kv,
                                                           const double* __restrict__ wv, int nw, int w_mode,
                                                           long cells, const double* __restrict__ D,
                                                           const uint64_t* __restrict__ masks,
                                                           const int* __restrict__ block_off, es_root_table tab,
                                                           double* __restrict__ d_lo_sign, double* __restrict__ d_hi_sign) {
  const long c = (long)blockIdx.x * 256 + threadIdx.x;
  if (c >= cells) return;
  const uint64_t m = masks[c >> 6];
  if (!((m >> (c & 63)) & 1ull)) return;
  const int pos = es_cell_rank(masks, block_off, c);
  if (pos >= tab.capacity) return;
  const long row = c / nw;
  const int j = (int)(c - row * nw);
  const double k = kv[row];
  tab.d_k[pos] = k;
  tab.d_row[pos] = (int32_t)row;
  tab.d_w_lo[pos] = pick_w(wv, w_mode, k, (int)row, nw, j);
  tab.d_w_hi[pos] = pick_w(wv, w_mode, k, (int)row, nw, j + 1);
  d_lo_sign[pos] = D[c];
  d_hi_sign[pos] = D[c + 1];
}

// Bracket refinement by (LANES+1)-section: LANES lanes share one bracket (64/LANES brackets per wave).  Per round the
// lanes evaluate D at the LANES interior points lo + (hi-lo)*(j+1)/(LANES+1); a ballot collects "sign differs from
// D(lo)" and the first set bit picks the sub-interval that keeps the sign change next to lo (the one a scan from lo
// would find, as the reference's left-to-right 3-point refinement does).  n_rounds = ceil(n_bisect ln2 / ln(LANES+1))
// rounds shrink the bracket at least as much as n_bisect bisections would; uniform trip count, no divergence.
// The kernel is bound by the number of SEQUENTIAL marches of one point per lane, not by throughput, as long as the
// launch stays below a few waves per SIMD: LANES = 16 (17-section, 4 rounds for n_bisect = 16) up to 32768 brackets,
// The section rule is FIXED (17-section, 16 lanes per bracket: 4 rounds for n_bisect = 16): it must not depend on the bracket
// count of the call, or a grid tiled over N ranks -- fewer brackets per call -- would be refined by another rule than
// the same grid on one rank and the merged root table would no longer be the N = 1 table bit for bit (round 2 switched
// to 9-section above 32768 brackets per call; the device-side count of es_shoot_find_roots_async could not even know).
// ES_REFINE_SECTIONS = 5 / 9 / 17 in the environment selects another rule for the whole process (tuning aid, honoured by
// the port); the port mirrors the default.
constexpr int kRefineSections = 17;
constexpr int kSharedMin = 0;      // brackets from which the wave-shared node entries are used (untwisted cylinder): always

// Workgroups of 4 waves share ONE LDS staging of the (k-independent) base table: 3.6 KB of LDS per wave instead of 14.4,
// so occupancy is no longer LDS-bound and the compiler aims for the register footprint of the point kernel -- which is
// what lets refinement waves fit beside the grid kernel's when consecutive steps are pipelined over two streams.
constexpr int REFINE_WAVES = 4;

// WPE = 3 (<= 168 VGPRs: 165 for the untwisted cylinder, no spill; two grid waves of 168 and one refinement wave share a
// SIMD's 512 registers) except for the twisted family, which needs 234 and would spill (WPE = 2).
// CHR > 0: the node entries of a bracket are formed once per chunk of CHR steps by the wave into its own LDS table and
// shared by the LANES lanes of the bracket (shoot_point_wavegroup); CHR = 0: every lane forms its own (shoot_point).
// SECTIONS_ONLY (n_polish < 0, what launch_refine uses whenever there is a section round): no status is needed from the
// evaluations (shoot_point<FAM, false>).
template <int FAM, int LANES, int CHR = 0, bool SECTIONS_ONLY = false, bool ONE = false,
          int WPE = ((FAM == FAM_CYLT || !ONE) ? 2 : 3)>
__global__ __launch_bounds__(64 * REFINE_WAVES) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void refine_kernel(ShootDev P, es_root_table tab, double* d_lo, double* d_hi, const int* __restrict__ d_n, int n_max,
                   int n_rounds, int n_polish, double tol_percent) {                      // d_lo / d_hi alias table columns
  // bracket count: from device memory (es_shoot_find_roots_async: the host never reads it; the launch is sized for
  // n_max = the table capacity and the workgroups beyond the count return at once) or n_max itself
  const int n = d_n ? (*d_n < n_max ? *d_n : n_max) : n_max;
  if ((int)(blockIdx.x * REFINE_WAVES * (64 / LANES)) >= n) return;                       // workgroup-uniform
  constexpr int GROUPS = 64 / LANES;
  constexpr int WTBL = GROUPS * (2 * (CHR > 0 ? CHR : 1) + 1) * FamTraits<FAM>::NE;      // doubles per wave
  __shared__ double es_point_lds[CHR > 0 ? REFINE_WAVES * WTBL : FamTraits<FAM>::NB * (2 * es_shoot_shared::CH + 1)];
  const int lane = threadIdx.x & 63;
  const int g = lane / LANES, j = lane % LANES;
  const int i = (blockIdx.x * REFINE_WAVES + ((int)threadIdx.x >> 6)) * GROUPS + g;
  const bool in = i < n;
  const double k = in ? tab.d_k[i] : 1.0;
  double lo = in ? tab.d_w_lo[i] : 1.0;
  double hi = in ? tab.d_w_hi[i] : 2.0;
  double flo = in ? d_lo[i] : 1.0;
  double fhi = in ? d_hi[i] : -1.0;
  const double frac = (double)(j + 1) / (double)(LANES + 1);
  double D = 0.0, rel = 0.0; uint8_t st = 0;
  double root = lo;
  // ONE evaluation site for the section rounds and the polish steps (the determinant evaluation is inlined: a single
  // copy keeps the kernel at the register footprint of the point kernel plus the bracket state, so that refinement
  // waves fit beside the grid kernel's when consecutive steps are pipelined over two streams)
  const int n_final = (n_polish > 0) ? n_polish : (n_polish == 0 ? 1 : 0);
  // ONE: a single section round per launch (the host launches the kernel n_rounds times; lo / hi / D(lo) / D(hi) travel
  // through the table columns, the same doubles).  A loop over rounds around the inlined determinant evaluation lets the
  // compiler hoist every round-invariant value of that code -- 64-bit literals of the Bessel polynomials, kernel
  // arguments -- out of it and carry them through the march (see es_tile_index).
  const int n_it = ONE ? 1 : n_rounds + n_final;
  for (int it = 0; it < n_it; ++it) {
    const bool section = ONE ? true : it < n_rounds;
    double x;
    if (section) {
      x = lo + (hi - lo) * frac;
    } else if (n_polish > 0) {
      // Newton-type polish in fp64: regula-falsi (secant through the bracket ends), every lane of the group the same.
      // A secant point that rounding puts ON or just outside an end means that end is the root to the last bit (|f|
      // there is rounding noise): stay at the end with the smaller |f| -- bisecting at that stage would throw the
      // root half a bracket away.  Only a NaN estimate falls back to the mid-point.
      x = lo - flo * (hi - lo) / (fhi - flo);
      if (!(x > lo && x < hi)) x = (x == x) ? ((fabs(flo) <= fabs(fhi)) ? lo : hi) : lo + (hi - lo) * 0.5;
    } else {
      x = lo + (hi - lo) * 0.5;                        // n_polish = 0: report the bracket mid-point
    }
    if (CHR > 0) shoot_point_wavegroup<FAM, (CHR > 0 ? CHR : 1), GROUPS>(P, k, x, D, rel, st, es_point_lds + ((int)threadIdx.x >> 6) * WTBL);
    else if (SECTIONS_ONLY) shoot_point<FAM, false>(P, k, x, x, D, rel, st, es_point_lds);
    else shoot_point<FAM>(P, k, x, x, D, rel, st, es_point_lds);
    if (section) {
      const bool diff = (D * flo < 0.0);               // NaN products compare false, as in the reference
      const unsigned long long bal = __ballot(diff);
      const unsigned long long bits = (LANES == 64) ? bal : ((bal >> (LANES * g)) & ((1ull << (LANES & 63)) - 1ull));
      const int first = bits ? (__ffsll((long long)bits) - 1) : LANES;     // first point whose sign differs from D(lo)
      const int src_hi = g * LANES + (first < LANES ? first : LANES - 1);
      const int src_lo = g * LANES + (first > 0 ? first - 1 : 0);
      const double x_hi = __shfl(x, src_hi), d_hi_new = __shfl(D, src_hi);
      const double x_lo = __shfl(x, src_lo), d_lo_new = __shfl(D, src_lo);
      if (first < LANES) { hi = x_hi; fhi = d_hi_new; }
      if (first > 0) { lo = x_lo; flo = (d_lo_new == d_lo_new) ? d_lo_new : flo; }
    } else {
      root = x;
      if (n_polish > 0) {
        if (D * flo < 0.0) { hi = x; fhi = D; } else if (D == D) { lo = x; flo = D; }
      }
    }
  }
  if (in && j == 0) {
    tab.d_w_lo[i] = lo;
    tab.d_w_hi[i] = hi;
    if (n_polish < 0) {                                // section rounds only: D at the ends goes on to refine_polish_kernel
      d_lo[i] = flo;
      d_hi[i] = fhi;
    } else {
      tab.d_w[i] = root;
      tab.d_resid[i] = rel;
      tab.d_flag[i] = (st == ES_PT_OK && rel < tol_percent) ? 1 : 0;
    }
  }
}

// The polish steps of refine_kernel with ONE lane per bracket (same arithmetic, bit for bit): in refine_kernel all LANES
// lanes of a bracket evaluate the same secant point, so two of its six marches do a sixteenth of the work they cost.
// ONE: a single polish step per launch (see refine_kernel), `n_polish` = 1 on the last of them, 0 before: the bracket and D
// at its ends go back to the columns they came from until the last step writes root, residual and flag.
template <int FAM, bool ONE = false, int WPE = ((FAM == FAM_CYLT || !ONE) ? 2 : 3)>
__global__ __launch_bounds__(64 * REFINE_WAVES) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void refine_polish_kernel(ShootDev P, es_root_table tab, double* d_lo, double* d_hi, const int* __restrict__ d_n,
                          int n_max, int n_polish, double tol_percent) {   // d_lo / d_hi alias table columns (no restrict)
  ES_POINT_LDS(FAM);
  const int n = d_n ? (*d_n < n_max ? *d_n : n_max) : n_max;
  if ((int)(blockIdx.x * (64 * REFINE_WAVES)) >= n) return;                               // workgroup-uniform
  const int i = blockIdx.x * (64 * REFINE_WAVES) + (int)threadIdx.x;
  const bool in = i < n;
  const double k = in ? tab.d_k[i] : 1.0;
  double lo = in ? tab.d_w_lo[i] : 1.0;
  double hi = in ? tab.d_w_hi[i] : 2.0;
  double flo = in ? d_lo[i] : 1.0;
  double fhi = in ? d_hi[i] : -1.0;
  double D = 0.0, rel = 0.0; uint8_t st = 0;
  double root = lo;
  const int n_it = ONE ? 1 : n_polish;
  for (int it = 0; it < n_it; ++it) {
    double x = lo - flo * (hi - lo) / (fhi - flo);
    if (!(x > lo && x < hi)) x = (x == x) ? ((fabs(flo) <= fabs(fhi)) ? lo : hi) : lo + (hi - lo) * 0.5;
    shoot_point<FAM>(P, k, x, x, D, rel, st, es_point_lds);
    root = x;
    if (D * flo < 0.0) { hi = x; fhi = D; } else if (D == D) { lo = x; flo = D; }
  }
  if (in) {
    tab.d_w_lo[i] = lo;
    tab.d_w_hi[i] = hi;
    if (ONE && n_polish == 0) {                        // not the last step: state for the next launch
      d_lo[i] = flo;
      d_hi[i] = fhi;
    } else {
      tab.d_w[i] = root;
      tab.d_resid[i] = rel;
      tab.d_flag[i] = (st == ES_PT_OK && rel < tol_percent) ? 1 : 0;
    }
  }
}

// =====================================================================================================================
// fp32 screening of the (k, omega) grid (BASELINE.json configs[4]: "fp32 bracket + fp64 refine").
//
// The bracket search only needs the SIGN of D at the grid points and the per-point status.  shoot_grid_f32_kernel
// marches the interior in fp32 (node entries formed in fp64 per row and rounded once into the LDS table; exterior and
// boundary algebra stay fp64) and marks every point at which fp32 cannot vouch for sign or status as UNSURE
// (status bit 0x80):
//   * a watched term of the coefficient set (Om^2 - omega_A^2, Om^2 - omega_c^2; C3 for the twisted family) comes
//     within F32_TAU_NODE (1e-3) of zero relative to the size of its parts at some node (near-singular coefficient: large
//     relative error in fp32, and the sign tracking that decides ES_PT_CONTINUUM is not reliable);
//   * |D| < F32_TAU_D * max(|outer|, |inner|)  (a root is close: the sign is within the fp32 error of the march);
//   * |outer| < F32_TAU_POLE * |inner|          (a pole of D is close: the sign of 1/r2 is within that error);
//   * a non-finite fp32 result.
// es_shoot_find_roots_mixed() re-evaluates the unsure points in fp64 (shoot_points_kernel, the arithmetic of the fp64
// grid kernel bit for bit), detects the brackets on the merged array, re-evaluates BOTH ENDS of every bracket in fp64
// (so the refinement starts from exactly the numbers the fp64 path has) and refines in fp64.
constexpr float F32_TAU_NODE = 1e-3f;
constexpr double F32_TAU_D = 5e-2;
constexpr double F32_TAU_POLE = 5e-2;
constexpr uint8_t F32_UNSURE = 0x80;

// All fp32 arithmetic of the march is written on PAIRS of points (clang ext-vector float2 -> v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32: two points per instruction); node entries are wave-uniform scalars broadcast to both
// halves by the packed instructions' operand selects.
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f v2(float x) { return (v2f){x, x}; }
__device__ __forceinline__ v2f vfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f vabs(v2f a) { return __builtin_elementwise_abs(a); }
__device__ __forceinline__ v2f vmin(v2f a, v2f b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ v2f vmax(v2f a, v2f b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ v2f vrcp(v2f a) { return (v2f){__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)}; }

struct CoefF { v2f a11, a12, a21, a22; };
struct CoefPreF { v2f n11, n12, n21, n22, den; };

// What the screening pass knows about the watched terms of the coefficient set:
//   band families (status known exactly from W = omega/k): per point the minimum over the nodes of |t1 t2| = |den| / (rho S)
//     -- how close Om^2 came to omega_A^2 or omega_c^2 (1 instruction per point and node);
//   tracked families: t1 = (omega - e0)^2 - omega_A^2 is negative at node j exactly for omega inside the interval
//     (e0_j - |omega_A,j|, e0_j + |omega_A,j|), t2 likewise with omega_c: the workgroup accumulates, while it stages the
//     node entries of its row, the extremes of the interval ends over all nodes -- min / max of the lower ends, min / max of
//     the upper ends, max of 1 / half-width (RowBands) -- and judges every point against these ten numbers AFTER the march
//     (round 2 kept running minima and maxima of t1 and t2 per point and node: 56 of the 560 VALU instructions of a loop
//     iteration).  With S = tau (omega^2 + omega_A^2(boundary)) and delta = S max_j(1 / a_j): all t_j > S if omega is more
//     than delta outside every interval, all t_j < -S if it is more than delta inside every interval; anything else -- the
//     continuum points and a margin around the band edges -- goes to fp64, which decides with the per-node tracking of
//     the fp64 kernels.  Twisted family in addition, per point: the extremes of C3 over the nodes (the fp64 kernel watches the sign of C3 D; D keeps its sign at a vouched-for point) and the minimum of
//     |C3| - tau |D (rho t1 + r d/dr[..])| (how close C3 came to zero relative to its leading part).
template <bool TRACK>
struct ScreenF {
  v2f mn = {3.0e38f, 3.0e38f};                          // !TRACK: min |t1 t2|
  v2f c3m = {3.0e38f, 3.0e38f};
  v2f c3lo = {3.0e38f, 3.0e38f}, c3hi = {-3.0e38f, -3.0e38f};   // extremes of C3 over the nodes (twisted family)
};

// extremes over the nodes of a row of the intervals in which t1 (index 0) and t2 (index 1) are negative; fp32, rounded
// OUTWARDS (a bound that is off by an ulp must err on the side of "unsure")
struct RowBands {
  float lo_min[2] = {3.0e38f, 3.0e38f}, lo_max[2] = {-3.0e38f, -3.0e38f};
  float hi_min[2] = {3.0e38f, 3.0e38f}, hi_max[2] = {-3.0e38f, -3.0e38f};
  float inv_a[2] = {0.0f, 0.0f};
  __device__ __forceinline__ static float down(double x) { const float f = (float)x; return f - fabsf(f) * 2.4e-7f - 1e-37f; }
  __device__ __forceinline__ static float up(double x) { const float f = (float)x; return f + fabsf(f) * 2.4e-7f + 1e-37f; }
  __device__ __forceinline__ void add(int t, double centre, double a) {
    const double lo = centre - a, hi = centre + a;
    lo_min[t] = fminf(lo_min[t], down(lo)); lo_max[t] = fmaxf(lo_max[t], up(lo));
    hi_min[t] = fminf(hi_min[t], down(hi)); hi_max[t] = fmaxf(hi_max[t], up(hi));
    inv_a[t] = fmaxf(inv_a[t], (a > 0.0) ? up(1.0 / a) : 3.0e38f);
  }
  // all lanes of the workgroup end up with the extremes over the whole workgroup; `red` = 10 * (T / 64) floats of LDS
  __device__ __forceinline__ void reduce(float* red, int T) {
    float v[10] = {lo_min[0], lo_min[1], hi_min[0], hi_min[1], -lo_max[0], -lo_max[1], -hi_max[0], -hi_max[1], -inv_a[0], -inv_a[1]};
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v[i] = fminf(v[i], __shfl_xor(v[i], off));
    const int wave = threadIdx.x >> 6, nwaves = T >> 6;
    __syncthreads();                                   // the LDS table of the last chunk is no longer read
    if ((threadIdx.x & 63) == 0)
#pragma unroll
      for (int i = 0; i < 10; ++i) red[wave * 10 + i] = v[i];
    __syncthreads();
    for (int wv = 0; wv < nwaves; ++wv)
#pragma unroll
      for (int i = 0; i < 10; ++i) v[i] = fminf(v[i], red[wv * 10 + i]);
    lo_min[0] = v[0]; lo_min[1] = v[1]; hi_min[0] = v[2]; hi_min[1] = v[3];
    lo_max[0] = -v[4]; lo_max[1] = -v[5]; hi_max[0] = -v[6]; hi_max[1] = -v[7]; inv_a[0] = -v[8]; inv_a[1] = -v[9];
  }
  // 0: every t_j of term t is certainly > S;  1: certainly < -S;  -1: cannot tell.
  // t = (|Om| - a)(|Om| + a): outside an interval by d, t > d (d + 2 a), i.e. > S once d >= sqrt(S) or d >= S / (2 a);
  // inside by d, -t > d a, i.e. > S once d >= S / a.
  __device__ __forceinline__ int sign_of(int t, double w, double S) const {
    const double d_in = S * (double)inv_a[t];
    const double d_out = fmin(sqrt(S), 0.5 * d_in);
    if (w < (double)lo_min[t] - d_out || w > (double)hi_max[t] + d_out) return 0;
    if (w > (double)lo_max[t] + d_in && w < (double)hi_min[t] - d_in) return 1;
    return -1;
  }
  // the same for ONE node (interval centre e0, half-width a, both fp64): +1 / -1 certain sign of t there, 0 cannot tell
  __device__ __forceinline__ static int node_sign(double w, double S, double e0, double a) {
    const double d = fabs(w - e0) - a;
    if (d >= fmin(sqrt(S), a > 0.0 ? 0.5 * S / a : INFINITY) * (1.0 + 1e-6)) return 1;
    if (a > 0.0 && -d >= (S / a) * (1.0 + 1e-6)) return -1;
    return 0;
  }
};

// per-row scalars of the flow slab in fp32 (KScal: k^2 c^2, k^2 vA^2, k^2 cT^2, k^4 cT^2 c^2; S_i)
struct SlabScalF { float kc2, kvA2, kcT2, k4c, S; };

// C1P = c1_power of the twisted family as a template parameter: a runtime select between Om and Om^2 costs two
// v_cndmask_b32 per pair and node (the kernel branches once per chunk instead).
template <int FAM, bool TRACK, int C1P = 1>
__device__ __forceinline__ void coef_pre_f32(const float* e, v2f w, CoefPreF& C, ScreenF<TRACK>& sc,
                                             const SlabScalF& ss) {
  if (FAM == FAM_SLABD) {
    // density slab (coef_pre<FAM_SLABD>): u' = v / F, v' = F m0 u; watched terms n1, n2, n3 (band family: the smallest
    // of their magnitudes over the nodes says how close a coefficient came to a singular point)
    const v2f w2 = w * w;
    const v2f n1 = v2(e[0]) - w2, n2 = v2(e[1]) - w2, n3 = v2(e[2]) - w2;
    sc.mn = vmin(sc.mn, vmin(vmin(vabs(n1), vabs(n2)), vabs(n3)));
    C.n11 = v2(0.0f); C.n22 = v2(0.0f);
    C.n12 = n1;
    C.n21 = v2(e[4]) * n3;
    C.den = v2(e[3]) * n2;
    return;
  }
  if (FAM == FAM_SLABF) {
    // flow slab (coef_pre<FAM_SLABF>): m0, D, coeff of SF-G:416-427 over the common denominator S t Om^2 n1; watched terms
    // n1, t, n3 and Om (as Om^2, on the scale of the others)
    const v2f Om = w - v2(e[0]);
    const v2f Om2 = Om * Om;
    const v2f t = Om2 - v2(ss.kcT2), n1 = v2(ss.kc2) - Om2, n3 = v2(ss.kvA2) - Om2;
    sc.mn = vmin(sc.mn, vmin(vmin(vabs(n1), vabs(t)), vmin(vabs(n3), Om2)));
    const v2f St = v2(ss.S) * t;
    const v2f G = vfma(St, t, v2(ss.k4c));
    const v2f X = Om * n1;
    const v2f OX = Om * X;
    const v2f g2 = v2(e[3]) * G;
    C.n11 = v2(0.0f);
    C.n12 = v2(1.0f);
    C.n22 = g2 * Om;
    C.n21 = -vfma(v2(e[2]), St * X, vfma(-g2, v2(e[1]), (n1 * n3) * OX));
    C.den = St * OX;
    return;
  }
  const v2f Om = w - v2(e[0]);
  const v2f Om2 = Om * Om;
  const v2f t1 = Om2 - v2(e[1]);
  const v2f t2 = Om2 - v2(e[2]);
  const v2f t12 = t1 * t2;
  if (!TRACK) sc.mn = vmin(sc.mn, vabs(t12));
  if (FAM == FAM_CYL0) {
    C.n11 = v2(0.0f);
    C.n12 = v2(e[3]) * t1;
    C.n21 = vfma(v2(e[5]), t2, v2(e[6]));
    C.n22 = v2(e[4]);
    C.den = t12;
  } else {
    const v2f D = v2(e[3]) * t12;
    const v2f Q = vfma(Om, v2(e[7]), vfma(Om2, v2(e[6]), -(t1 * v2(e[5]))));
    const v2f T = vfma(v2(e[9]), Om, v2(e[8]));
    const v2f OmP = (C1P == 2) ? Om2 : Om;
    const v2f t2T = t2 * T;
    const v2f C1 = vfma(Q, OmP, -(v2(e[10]) * t2T));
    const v2f C2 = vfma(Om2, Om2, -(v2(e[11]) * t2));
    const v2f c3a = D * vfma(v2(e[4]), t1, v2(e[12]));
    const v2f C3 = c3a + vfma(Q, Q, -(v2(e[13]) * t2T * T));
    // third watched term of the fp64 kernel: the sign of F = r D / C3, i.e. of C3 D.  A point is only vouched for when t1
    // and t2 keep one sign over all nodes (RowBands), and e[3] = rho S > 0: D = e[3] t1 t2 then keeps its sign and C3 D
    // changes sign exactly where C3 does -- C3 itself is tracked (no product per node).
    // (its extremes over the nodes: with the two nodes of a step one v_min3_f32 and one v_max3_f32 per point, where OR-ing
    // and AND-ing the sign bits took four integer instructions)
    sc.c3lo.x = fminf(sc.c3lo.x, C3.x); sc.c3hi.x = fmaxf(sc.c3hi.x, C3.x);
    sc.c3lo.y = fminf(sc.c3lo.y, C3.y); sc.c3hi.y = fmaxf(sc.c3hi.y, C3.y);
    // |C3| - tau |c3a| per component with the source modifiers of the unpacked v_fma_f32 (the packed form has no |x|
    // modifier: four v_and_b32 and a v_pk_fma_f32 per pair and node)
    sc.c3m.x = fminf(sc.c3m.x, fmaf(-F32_TAU_NODE, fabsf(c3a.x), fabsf(C3.x)));
    sc.c3m.y = fminf(sc.c3m.y, fmaf(-F32_TAU_NODE, fabsf(c3a.y), fabsf(C3.y)));
    C.n11 = v2(0.0f);                                   // a11 = -a22: not formed (rk4_step_adjoint_f32 negates a22)
    C.n22 = C1;
    C.n12 = C3 * v2(e[15]);
    C.n21 = -(v2(e[14]) * C2);
    C.den = D;
  }
}

template <int FAM>
__device__ __forceinline__ void coef_finish_f32(const CoefPreF& C, v2f inv, CoefF& A) {
  if (FAM == FAM_CYL0) {
    A.a11 = v2(0.0f); A.a22 = v2(0.0f); A.a12 = C.n12; A.a21 = vfma(C.n21, inv, C.n22);
  } else if (FAM == FAM_SLABD) {
    A.a11 = v2(0.0f); A.a22 = v2(0.0f); A.a12 = C.n12 * inv; A.a21 = C.n21;
  } else if (FAM == FAM_SLABF) {
    A.a11 = v2(0.0f); A.a12 = v2(1.0f); A.a21 = C.n21 * inv; A.a22 = C.n22 * inv;
  } else {
    A.a11 = v2(0.0f); A.a22 = C.n22 * inv; A.a12 = C.n12 * inv; A.a21 = C.n21 * inv;       // a11 = -a22
  }
}

template <int FAM>
__device__ __forceinline__ void rk4_step_adjoint_f32(v2f& p, v2f& q, const CoefF& B0, const CoefF& Bm, const CoefF& B1,
                                                     float h, float h2, float h6, float h3) {
#define ES_RHS_TF(A, pp, qq, kp, kq)                                                            \
  if (FAM == FAM_CYLT)       { kp = vfma(-A.a22, pp, A.a21 * qq); kq = vfma(A.a22, qq, A.a12 * pp); } \
  else if (FAM == FAM_SLABF) { kp = A.a21 * qq;                  kq = vfma(A.a22, qq, pp); }         \
  else                       { kp = A.a21 * qq;                  kq = A.a12 * pp; }
  v2f k1p, k1q, k2p, k2q, k3p, k3q, k4p, k4q, tp, tq;
  ES_RHS_TF(B0, p, q, k1p, k1q);
  tp = vfma(v2(h2), k1p, p); tq = vfma(v2(h2), k1q, q);
  ES_RHS_TF(Bm, tp, tq, k2p, k2q);
  tp = vfma(v2(h2), k2p, p); tq = vfma(v2(h2), k2q, q);
  ES_RHS_TF(Bm, tp, tq, k3p, k3q);
  tp = vfma(v2(h), k3p, p); tq = vfma(v2(h), k3q, q);
  ES_RHS_TF(B1, tp, tq, k4p, k4q);
  p = vfma(v2(h6), k1p + k4p, vfma(v2(h3), k2p + k3p, p));
  q = vfma(v2(h6), k1q + k4q, vfma(v2(h3), k2q + k3q, q));
#undef ES_RHS_TF
}

// One RK4 step of the fp32 march for the NP pairs of a lane.  `node` = the fp32 entries of node 2J in the LDS table,
// those of the mid-point 2J + 1 follow NES floats later (node-major table: every entry of a step sits at a constant
// offset from ONE uniform address -- ds_read_b128 with immediate offsets; the field-major table of round 2 needed a
// base register of its own per field: 16 v_mov_b32 + 25 s_add_i32 per loop iteration of the twisted family).
template <int FAM, int NP, bool TRACK, int C1P>
__device__ __forceinline__ void f32_step(const float* node, const v2f* wf, v2f* zp, v2f* zq, const CoefF* BIN, CoefF* BOUT,
                                         ScreenF<TRACK>* scr, const SlabScalF& ss, float h, float h2, float h6, float h3) {
  constexpr int NE = FamTraits<FAM>::NE;
  constexpr int NES = (NE + 3) & ~3;
  float em[NE], e1[NE];
#pragma unroll
  for (int f = 0; f < NE; ++f) { e1[f] = node[f]; em[f] = node[NES + f]; }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    CoefPreF Cm, C1;
    coef_pre_f32<FAM, TRACK, C1P>(em, wf[p], Cm, scr[p], ss);
    coef_pre_f32<FAM, TRACK, C1P>(e1, wf[p], C1, scr[p], ss);
    const v2f inv = vrcp(Cm.den * C1.den);
    CoefF Bm;
    coef_finish_f32<FAM>(Cm, C1.den * inv, Bm);
    coef_finish_f32<FAM>(C1, Cm.den * inv, BOUT[p]);
    rk4_step_adjoint_f32<FAM>(zp[p], zq[p], BIN[p], Bm, BOUT[p], h, h2, h6, h3);
  }
}

// The steps of one LDS chunk (nst of them, from the far end of the chunk towards the boundary).  The adjoint march is
// renormalised: fp32 has 8 bits of exponent, an evanescent interior grows like e^{kappa (1 - r_ax)} (1e30 and more at
// large k).  Both components are scaled by a power of two whenever they leave [2^-40, 2^40]; the boundary algebra only
// uses the ratio r1 : r2 and the (rescaled) target of the axis condition.
template <int FAM, int NP, bool TRACK, int C1P>
__device__ __forceinline__ void f32_march_chunk(const float* lds, int nst, bool first, bool sausage_axis, const v2f* wf,
                                                v2f* zp, v2f* zq, int* zexp, CoefF* B0, CoefF* B1, ScreenF<TRACK>* scr,
                                                const SlabScalF& ss, float h, float h2, float h6, float h3) {
  constexpr int NE = FamTraits<FAM>::NE;
  constexpr int NES = (NE + 3) & ~3;
  if (first) {                                         // far end of the march: coefficients of the last node, start values
    float eL[NE];
#pragma unroll
    for (int f = 0; f < NE; ++f) eL[f] = lds[2 * nst * NES + f];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      CoefPreF C;
      coef_pre_f32<FAM, TRACK, C1P>(eL, wf[p], C, scr[p], ss);
      coef_finish_f32<FAM>(C, v2(1.0f) / C.den, B0[p]);
      if ((FAM == FAM_CYL0 || FAM == FAM_CYLT) && sausage_axis) {
        zp[p] = (FAM == FAM_CYLT) ? -B0[p].a22 : B0[p].a11;
        zq[p] = B0[p].a12;
      } else { zp[p] = v2(1.0f); zq[p] = v2(0.0f); }
    }
  }
  // steps in pairs with the roles of B0 / B1 swapped (no coefficient copies); CH is even, so only the last chunk of an
  // odd march has a single leading step
  int j = nst - 1;
  if (nst & 1) {
    f32_step<FAM, NP, TRACK, C1P>(lds + 2 * j * NES, wf, zp, zq, B0, B1, scr, ss, h, h2, h6, h3);
#pragma unroll
    for (int p = 0; p < NP; ++p) B0[p] = B1[p];
    --j;
  }
  for (; j >= 1; j -= 2) {
    f32_step<FAM, NP, TRACK, C1P>(lds + 2 * j * NES, wf, zp, zq, B0, B1, scr, ss, h, h2, h6, h3);
    f32_step<FAM, NP, TRACK, C1P>(lds + 2 * (j - 1) * NES, wf, zp, zq, B1, B0, scr, ss, h, h2, h6, h3);
    if (((j - 1) & 31) == 0) {                         // renormalise every 32 steps
#pragma unroll
      for (int p = 0; p < 2 * NP; ++p) {
        float a = (p & 1) ? zp[p >> 1].y : zp[p >> 1].x;
        float b = (p & 1) ? zq[p >> 1].y : zq[p >> 1].x;
        const float mag = fmaxf(fabsf(a), fabsf(b));
        if (mag > 1.0995116e12f || (mag < 9.094947e-13f && mag > 0.0f)) {       // outside [2^-40, 2^40]
          int ex;
          (void)frexpf(mag, &ex);
          a = ldexpf(a, -ex);
          b = ldexpf(b, -ex);
          zexp[p] += ex;
          if (p & 1) { zp[p >> 1].y = a; zq[p >> 1].y = b; } else { zp[p >> 1].x = a; zq[p >> 1].x = b; }
        }
      }
    }
  }
}

// The closed-form exterior (fp64 Bessel code, ~100 VGPRs) is evaluated AFTER the march, when the loop state is dead;
// before the march only the sign of m_e is needed to know which lanes have something to march.
template <int FAM, int PTS, int MAXT, bool TRACK, int WPE>
__global__ __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void shoot_grid_f32_kernel(ShootDev P, const double* __restrict__ kv, int nk, const double* __restrict__ wv, int nw,
                           int w_mode, double* __restrict__ Dout, uint8_t* __restrict__ stout) {
  static_assert(PTS % 2 == 0, "points are processed in pairs");
  constexpr int NP = PTS / 2;
  constexpr int NE = FamTraits<FAM>::NE;
  constexpr int NES = (NE + 3) & ~3;                   // floats per node in the LDS table (16-byte rows)
  __shared__ __align__(16) float lds[NES * (2 * CH + 1)];
  const int T = blockDim.x;
  const int nsteps = P.n_nodes - 1;
  const float h = (float)P.h, h2 = (float)(0.5 * P.h), h6 = (float)(P.h / 6.0), h3 = (float)(P.h / 3.0);
  const int nseg = (nw + T * PTS - 1) / (T * PTS);
  const long ntiles = (long)nk * nseg;
  for (long tile = es_tile_index(), once = 1; once && tile < ntiles; once = 0) {
    const int seg = (int)(tile / nk);                  // segment-major, as in shoot_grid_kernel (XCD balance)
    const int row = (int)(tile - (long)seg * nk);
    const int w0 = seg * T * PTS;
    const double k = kv[row];
    const KScal s = make_kscal(P, k);
    const SlabScalF ss = {(float)s.kc2, (float)s.kvA2, (float)s.kcT2, (float)s.k4c, (float)P.S_i};
    v2f wf[NP], zp[NP], zq[NP];
    int zexp[PTS];                                     // accumulated power-of-two scaling of (zp, zq), per point
    CoefF B0[NP], B1[NP];
    ScreenF<TRACK> scr[NP];
    bool lane_live = false;
#pragma unroll
    for (int p = 0; p < PTS; ++p) {
      const int iw = w0 + p * T + (int)threadIdx.x;
      const double w = (iw < nw) ? pick_w(wv, w_mode, k, row, nw, iw) : 1.0;
      if (p & 1) wf[p >> 1].y = (float)w; else wf[p >> 1].x = (float)w;
      // m_e > 0 (evanescent exterior) and, for the band families, not inside a continuum band: worth a march
      const double Oe = (FAM == FAM_SLABD || FAM == FAM_SLABF) ? (w - k * P.U_e) : w;
      const double k2 = k * k, w2 = Oe * Oe;
      const double m_e = ((k2 * P.vAe2 - w2) * (k2 * P.ce2 - w2)) / (P.Se * (k2 * P.cTe2 - w2));
      const bool dead = !TRACK && band_crossed(P, k, w);
      lane_live = lane_live || (iw < nw && m_e > 0.0 && !dead);
      zexp[p] = 0;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) { zp[p] = v2(0.0f); zq[p] = v2(0.0f); }
    const bool wave_live = __any(lane_live);
    const bool wg_live = __syncthreads_or(wave_live ? 1 : 0) != 0;
    const int nchunks = wg_live ? (nsteps + CH - 1) / CH : 0;
    const bool sausage_axis = P.axis_bc == ES_AXIS_SAUSAGE;
    RowBands bands;
    for (int c = nchunks - 1; c >= 0; --c) {
      const int c0 = c * CH;
      const int nst = (nsteps - c0 < CH) ? (nsteps - c0) : CH;
      __syncthreads();
      for (int i = threadIdx.x; i < 2 * nst + ES_FAR_NODE(c, nchunks); i += T) {   // far node: first chunk only (es_shoot_grid_body.hpp)
        double b[FamTraits<FAM>::NB], e[NE];
        load_base<FAM>(P, 2 * c0 + i, b);
        make_entry<FAM>(b, s, e);                      // fp64, rounded to fp32 once
        float ef32[NES];
#pragma unroll
        for (int f = 0; f < NES; ++f) ef32[f] = (f < NE) ? (float)e[f] : 0.0f;
#pragma unroll
        for (int f = 0; f < NES; f += 4)
          *reinterpret_cast<float4*>(&lds[i * NES + f]) = make_float4(ef32[f], ef32[f + 1], ef32[f + 2], ef32[f + 3]);
        if (TRACK) {                                   // where t1 / t2 are negative at this node: omega within a of e0
          bands.add(0, e[0], sqrt(e[1]));
          bands.add(1, e[0], sqrt(e[2]));
        }
      }
      __syncthreads();
      if (!wave_live) continue;
      if (FAM == FAM_CYLT && P.c1_power == 2)
        f32_march_chunk<FAM, NP, TRACK, 2>(lds, nst, c == nchunks - 1, sausage_axis, wf, zp, zq, zexp, B0, B1, scr, ss, h, h2, h6, h3);
      else
        f32_march_chunk<FAM, NP, TRACK, 1>(lds, nst, c == nchunks - 1, sausage_axis, wf, zp, zq, zexp, B0, B1, scr, ss, h, h2, h6, h3);
    }
    if (TRACK && wg_live) bands.reduce(lds, T);        // workgroup-uniform condition (the reduction has barriers)
    // the exterior code below needs products of k the prologue has already formed (k^2 vA_e^2, ...): reused, they would
    // sit in registers across the whole march -- and did not fit (five spilled doubles per lane, 10 MB of scratch writes
    // per launch).  An opaque copy of k makes the compiler form them again here.
    double kx = k;
    asm volatile("" : "+v"(kx));
    double bf[FamTraits<FAM>::NB], ef[NE];
    load_base<FAM>(P, 0, bf);
    make_entry<FAM>(bf, s, ef);
    double emid[3] = {0.0, 0.0, 0.0}, elast[3] = {0.0, 0.0, 0.0};     // e0, omega_A^2, omega_c^2 of two more sampled nodes
    if (TRACK) {
      double et[NE];
      load_base<FAM>(P, P.npts / 2, bf);
      make_entry<FAM>(bf, s, et);
      emid[0] = et[0]; emid[1] = et[1]; emid[2] = et[2];
      load_base<FAM>(P, P.npts - 1, bf);
      make_entry<FAM>(bf, s, et);
      elast[0] = et[0]; elast[1] = et[1]; elast[2] = et[2];
      load_base<FAM>(P, 0, bf);
    }
#pragma unroll
    for (int p = 0; p < PTS; ++p) {
      const int iw = w0 + p * T + (int)threadIdx.x;
      if (iw >= nw) continue;
      const double w = pick_w(wv, w_mode, kx, row, nw, iw);
      const ExteriorLite X = exterior_lite(P, kx, w, w);
      const bool hi_half = (p & 1);
      const ScreenF<TRACK>& sc = scr[p >> 1];
      const float zpp = hi_half ? zp[p >> 1].y : zp[p >> 1].x;
      const float zqq = hi_half ? zq[p >> 1].y : zq[p >> 1].x;
      // r = (r1, r2) * 2^zexp: the common factor multiplies the homogeneous part of the axis condition; its target
      // (bc_const * xi_e, non-zero only for the twisted kink condition) is divided by it instead
      ShootDev Pl = P;
      Pl.bc_const = P.bc_const_raw * ldexp(1.0, -zexp[p]);
      Pl.slab_sign = P.slab_sign * ldexp(1.0, -zexp[p]);          // slabs: (slab_sign - r1) / r2 with r = (r1, r2) 2^zexp
      const Mismatch M = boundary_algebra<FAM>(Pl, s, w, X, (double)zpp, (double)zqq, ef);
      // watched terms: certain sign / certain crossing / unsure (see ScreenF)
      const float S = F32_TAU_NODE * (float)(w * w + ef[1]);
      bool crossed, node_unsure;
      if (!TRACK) {                                     // band families: the status is exact (fp64 test on W = omega/k)
        crossed = band_crossed(P, kx, w);
        const float mnp = hi_half ? sc.mn.y : sc.mn.x;
        if (FAM == FAM_SLABD || FAM == FAM_SLABF) {
          // slabs: the smallest magnitude of a watched term over the nodes against tau x the scale of these terms
          // (omega^2 + k^2 c^2; the flow slab's omega is the Doppler-shifted one at the boundary)
          const double Omb = (FAM == FAM_SLABF) ? (w - ef[0]) : w;
          const float sz = (float)(Omb * Omb + ((FAM == FAM_SLABF) ? s.kc2 : ef[0]));
          node_unsure = !crossed && !(mnp > F32_TAU_NODE * sz);
        } else {
        // an evaluated point with a coefficient close to a singular point: |t1 t2| below tau (omega^2 + omega_A^2)^2 at
        // some node, i.e. ONE of the two factors within tau of zero relative to its size
        const float sz = (float)(w * w + ef[1]);
        node_unsure = !crossed && !(mnp > F32_TAU_NODE * sz * sz);
        }
      } else {
        const float c3m = hi_half ? sc.c3m.y : sc.c3m.x;
        const float c3lo = hi_half ? sc.c3lo.y : sc.c3lo.x, c3hi = hi_half ? sc.c3hi.y : sc.c3hi.x;
        // t1, t2: certainly of one sign at every node (judged from the row's interval extremes); or certainly of BOTH signs
        // inside the domain, seen at three sampled nodes (boundary, middle, far end: one of them certainly negative, another
        // certainly positive => ES_PT_CONTINUUM whatever happens in between -- most of a continuum band of a monotone
        // profile); anything else is the margin fp64 decides
        const bool sure12 = bands.sign_of(0, w, (double)S) >= 0 && bands.sign_of(1, w, (double)S) >= 0;
        bool cross12 = false;
        if (!sure12) {
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int s0 = RowBands::node_sign(w, (double)S, ef[0], sqrt(ef[1 + t]));
            const int s1 = RowBands::node_sign(w, (double)S, emid[0], sqrt(emid[1 + t]));
            const int s2 = RowBands::node_sign(w, (double)S, elast[0], sqrt(elast[1 + t]));
            cross12 = cross12 || ((s0 < 0 || s1 < 0 || s2 < 0) && (s0 > 0 || s1 > 0 || s2 > 0));
          }
        }
        if (cross12) {
          crossed = true; node_unsure = false;
        } else if (sure12 && (FAM != FAM_CYLT || c3m >= 0.0f)) {
          crossed = (FAM == FAM_CYLT) && (c3lo < 0.0f) && (c3hi >= 0.0f);   // C3 takes both signs (c3m >= 0: never within tau of 0)
          node_unsure = false;
        } else {
          crossed = false; node_unsure = true;
        }
      }
      double D, rel; uint8_t st;
      finish_point(P, M, X, crossed, D, rel, st);
      if (!TRACK && crossed && X.status == ES_PT_OK) { st = ES_PT_CONTINUUM; D = NAN; }   // not marched: no D to judge
      bool unsure = false;
      if (X.status == ES_PT_OK && crossed) {
        unsure = TRACK && !isfinite(M.d);               // fp64 reports ES_PT_NONFINITE before ES_PT_CONTINUUM: let it decide
      } else if (X.status == ES_PT_OK) {
        double scale = fmax(fabs(M.outer), fabs(M.inner));
        if (FAM == FAM_SLABD || FAM == FAM_SLABF) {
          // slabs: the inner term is (slab_sign - r1) x ..., and r1 -- an element of the transfer matrix of an oscillating
          // solution -- comes close to +-1: the fp32 error of r1 is then measured against the terms BEFORE the cancellation.
          // The same algebra with the sign of slab_sign chosen so that nothing cancels gives that magnitude.
          ShootDev Pb = Pl;
          Pb.slab_sign = (zpp >= 0.0f) ? -fabs(Pl.slab_sign) : fabs(Pl.slab_sign);
          const Mismatch Mb = boundary_algebra<FAM>(Pb, s, w, X, (double)zpp, (double)zqq, ef);
          scale = fmax(scale, fabs(Mb.inner));
        }
        unsure = node_unsure || !isfinite(M.d) || !(fabs(M.d) > F32_TAU_D * scale) ||
                 !(fabs(M.outer) > F32_TAU_POLE * fabs(M.inner));
      }
      const size_t o = (size_t)row * nw + iw;
      Dout[o] = D;
      stout[o] = unsure ? (uint8_t)(st | F32_UNSURE) : st;
    }
  }
}

// cells with the UNSURE bit -> ballot masks + block counts (same layout as bracket_flag_kernel)
__global__ __launch_bounds__(256) void unsure_flag_kernel(const uint8_t* __restrict__ st, long cells,
                                                          uint64_t* __restrict__ masks, int* __restrict__ block_counts) {
  __shared__ int wave_cnt[4];
  const long c = (long)blockIdx.x * 256 + threadIdx.x;
  const bool flag = (c < cells) && (st[c] & F32_UNSURE);
  const uint64_t m = __ballot(flag);
  if ((threadIdx.x & 63) == 0) {
    masks[c >> 6] = m;
    wave_cnt[threadIdx.x >> 6] = __popcll(m);
  }
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// ordered list of the flagged cells: (k, omega) pairs for the fp64 point kernel and the cell index to scatter back to
__global__ __launch_bounds__(256) void unsure_gather_kernel(const double* __restrict__ kv, const double* __restrict__ wv,
                                                            int nw, int w_mode, long cells,
                                                            const uint64_t* __restrict__ masks,
                                                            const int* __restrict__ block_off, double* __restrict__ pk,
                                                            double* __restrict__ pw, long* __restrict__ pcell) {
  const long c = (long)blockIdx.x * 256 + threadIdx.x;
  if (c >= cells) return;
  if (!((masks[c >> 6] >> (c & 63)) & 1ull)) return;
  const int pos = es_cell_rank(masks, block_off, c);
  const long row = c / nw;
  const int j = (int)(c - row * nw);
  const double k = kv[row];
  pk[pos] = k;
  pw[pos] = pick_w(wv, w_mode, k, (int)row, nw, j);
  pcell[pos] = c;
}

__global__ __launch_bounds__(256) void scatter_points_kernel(const long* __restrict__ pcell, const double* __restrict__ pD,
                                                             const uint8_t* __restrict__ pst, int n,
                                                             double* __restrict__ D, uint8_t* __restrict__ st) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  D[pcell[i]] = pD[i];
  st[pcell[i]] = pst[i];
}

// both ends of every bracket as (k, omega) pairs: [0, n) lower ends, [n, 2n) upper ends
__global__ __launch_bounds__(256) void bracket_ends_kernel(es_root_table tab, int n, double* __restrict__ pk,
                                                           double* __restrict__ pw) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  pk[i] = tab.d_k[i]; pk[n + i] = tab.d_k[i];
  pw[i] = tab.d_w_lo[i]; pw[n + i] = tab.d_w_hi[i];
}

// fp64 values at the bracket ends replace the screening values the refinement starts from; a bracket whose fp64 ends do
// not change sign (or are not both ES_PT_OK) would be a failure of the screening bound: counted, never hidden
__global__ __launch_bounds__(256) void bracket_ends_store_kernel(const double* __restrict__ pD, const uint8_t* __restrict__ pst,
                                                                 int n, double* __restrict__ d_lo, double* __restrict__ d_hi,
                                                                 int* __restrict__ violations) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double a = pD[i], b = pD[n + i];
  d_lo[i] = a;
  d_hi[i] = b;
  if (!(pst[i] == ES_PT_OK && pst[n + i] == ES_PT_OK && a * b < 0.0)) atomicAdd(violations, 1);
}

// ---- host side -------------------------------------------------------------------------------------------------
int check_problem(es_context* ctx, const es_problem* prob) {
  ES_REQUIRE(ctx, prob != nullptr, "null problem");
  return ES_SUCCESS;
}

// Launch shape of the grid kernel = (PTS points per lane, WPE waves per SIMD the register cap allows).  Every shape runs
// workgroups of at most 256 threads (one wave per SIMD of a CU) over tiles (k-row, omega-segment of T * PTS points) and
// parks the exterior results in LDS during the march, so the register cap 512 / WPE is the march loop's alone:
//   WPE = 2 -> 256 VGPRs, 3 -> 168, 4 -> 128.
// Round 2 picked between 1024-, 512- and 256-thread shapes by row width only; the 1024-thread ones (128 VGPRs by launch
// bounds) spilled 12 - 89 VGPRs inside the march loop of every family but the untwisted cylinder with one point per lane --
// exactly the shapes BASELINE configs[1] and configs[4] (1024 columns) selected.  Now every family has its own table of
// spill-free shapes (tools/codeobj_table.py prints registers / spills / LDS of each built instantiation;
// tests/test_codeobj.py fails on a spill in a shape the table selects) and the cost of a point in each of them, measured
// on the GPU (tools/probe/time_grid_shapes.py -> profiles/r3_grid_shapes.json); pick_shape() minimises
// padded points x cost per point for the row width at hand.  ES_GRID_SHAPE="pts,wpe" in the environment overrides it
// (tuning aid).  D does not depend on the shape: a point's arithmetic is the same in all of them
// (tests/test_shoot_gpu.py::test_grid_shapes_bit_identical).
struct GridShape { int pts, wpe; };

// Per family, indexed by points per lane (1, 2, 4): the register cap paired with it and the relative cost of one point,
// measured on the grids of the BASELINE configs (profiles/r3_grid_shapes.json: min of three launches, MI355X; all nine
// (points, cap) combinations per family bit-identical).  Reading of that table: with the spills gone the shapes of one
// family lie within 15 % of each other -- every one of them runs at 0.85 - 1.0 of the issue bound of its own instruction
// stream (tools/isa_loop_count.py) -- so the choice is about padding (a row of 384 frequencies fills 2 points x 192 lanes
// exactly, 4 points x 128 lanes waste a quarter) and about not spilling; four points per lane share the LDS reads and
// the loop overhead best.  WPE = 4 is used where the kernel fits 128 registers without a spill.
template <int FAM> struct ShapeTable;
// (untwisted cylinder: three points per lane are built for rows of 513 - 768 frequencies = 3 x 256 lanes, a four-wave workgroup
// where 4 x 192 lanes make a three-wave one.  For a row of 384 frequencies, configs[2], 3 x 128 lanes was measured too: 1.28 ms per
// launch against 1.14 ms for 2 x 192: its two-wave workgroups hold 39 KB of LDS each, four fit a CU, two waves per SIMD;
// measured again at the end of round 3 with a 128-thread instantiation -- 27 KB of LDS, 145 registers, two steps per division --:
// 1.12 - 1.14 ms against 1.08 - 1.10 ms for 2 x 192, configs[2] 7.18 - 7.25 against 7.24 ms per step; what rows of 384 needed was a
// four-wave workgroup: two rows each, shoot_grid_kernel_r2, 0.99 ms.  The other families keep wpe[3] = 0.)
// (measured again after the kernels became one tile per workgroup -- es_tile_index -- which freed 40 - 90 registers per
// shape: profiles/r3e_grid_shapes.json, ms per launch, best register cap per point count)
//   untwisted cylinder 1024 x 4096: 4 pts 5.03 (wpe 3), 2 pts 5.14, 1 pt 5.46; 4096 x 384: 2 pts x 192 lanes 1.27 (wpe 4; 1.32
//   at wpe 3), 4 pts x 128 lanes 1.80
template <> struct ShapeTable<FAM_CYL0>  { static constexpr int wpe[5] = {0, 4, 3, 3, 3}; static constexpr double cost[5] = {0, 1.09, 1.025, 1.01, 1.0}; };
// twisted cylinder 1024 x 1024, N = 2000: 4 pts 6.60 (200 registers, two workgroups per CU), 2 pts 6.70, 1 pt 6.85 (7.8 - 8.2
// for every shape while the tile loop was there)
template <> struct ShapeTable<FAM_CYLT>  { static constexpr int wpe[5] = {0, 3, 2, 0, 2}; static constexpr double cost[5] = {0, 1.04, 1.015, 0, 1.0}; };
// density slab 1024 x 1024: 4 pts 1.45 (wpe 3), 2 pts 1.62 (wpe 4), 1 pt 1.75 (wpe 4)
template <> struct ShapeTable<FAM_SLABD> { static constexpr int wpe[5] = {0, 4, 4, 0, 3}; static constexpr double cost[5] = {0, 1.20, 1.12, 0, 1.0}; };
// flow slab 1024 x 1024: 4 pts 1.28 (wpe 3), 2 pts 1.25 (wpe 4), 1 pt 1.33 (wpe 2): within the noise of one another
template <> struct ShapeTable<FAM_SLABF> { static constexpr int wpe[5] = {0, 2, 4, 0, 3}; static constexpr double cost[5] = {0, 1.04, 1.0, 0, 1.0}; };

inline int shape_threads(int nw, int pts) {
  int T = ((nw + pts - 1) / pts + 63) / 64 * 64;
  if (T < 64) T = 64;
  if (T > 256) T = 256;
  return T;
}

template <int FAM>
GridShape pick_shape(int nw, bool track) {
  GridShape best{4, 2};
  double best_cost = 1e300;
  for (int pts : {4, 3, 2, 1}) {
    if (ShapeTable<FAM>::wpe[pts] == 0) continue;      // not a shape of this family
    const int T = shape_threads(nw, pts);
    const long span = (long)T * pts;
    const long padded = (nw + span - 1) / span * span;
    // workgroups of fewer than four waves do not load the four SIMDs of a CU evenly and their waves wait for each other at
    // the chunk barriers: ns per point-step against four-wave workgroups, measured (tools/probe/time_row_width.py)
    static const double wave_count_cost[5] = {0.0, 1.83, 1.39, 1.17, 1.0};
    const double c = (double)padded * ShapeTable<FAM>::cost[pts] * wave_count_cost[T / 64];
    if (c < best_cost) { best_cost = c; best = GridShape{pts, ShapeTable<FAM>::wpe[pts]}; }
  }
  // per-node sign tracking of a band family (profiles whose continuum intervals do not overlap): 6 - 12 more registers
  // per point; the 256-register shapes hold them without a spill
  if (track && fam_has_bands<FAM>()) best.wpe = 2;
  if (const char* ev = getenv("ES_GRID_SHAPE")) {
    int p = 0, w = 0;
    if (sscanf(ev, "%d,%d", &p, &w) == 2 && p >= 1 && p <= 4 && w >= 2 && w <= 4) best = GridShape{p, w};
  }
  return best;
}

// which (PTS, WPE, TRACK) instantiations are built: the shapes the tables select, the tracking fall-backs at WPE = 2, and
// (-DES_ALL_GRID_SHAPES, the measuring build of tools/probe/time_grid_shapes.py) everything ES_GRID_SHAPE can name
template <int FAM, int PTS, int WPE, bool TRACK>
constexpr bool shape_built() {
  if (PTS == 3 && ShapeTable<FAM>::wpe[3] == 0) return false;
#if defined(ES_ALL_GRID_SHAPES)
  return true;
#else
  if (TRACK && fam_has_bands<FAM>()) return WPE == 2;
  if (FAM == FAM_CYL0 && PTS == 4 && WPE == 2) return true;   // A/B aid for the headline shape (ES_GRID_SHAPE=4,2: 175 VGPRs, no spill)
  return ShapeTable<FAM>::wpe[PTS] == WPE;
#endif
}

// rows of at most 512 frequencies of the band families (untwisted cylinder, slabs) go two to a four-wave workgroup
// (shoot_grid_kernel_r2: 128 lanes x ceil(nw / 128) points per row) unless the problem needs per-node sign tracking; ES_GRID_ROWS2=0 and ES_GRID_SHAPE select
// the one-row shapes (A/B aids)
template <int FAM>
bool rows2_shape(int nw, bool track) {
  if (!fam_has_bands<FAM>() || track || nw > 512) return false;
  const char* r2 = getenv("ES_GRID_ROWS2");
  return !(r2 && r2[0] == '0') && !getenv("ES_GRID_SHAPE");
}

template <int FAM>
int launch_grid(es_context* ctx, const es_problem* prob, const double* d_k, int nk, const double* d_w, int nw,
                int w_mode, double* d_D, double* d_rel, uint8_t* d_status, int flags = 0) {
  GridOpts opts;
  opts.skip = (flags & ES_EVAL_SKIP_CONTINUUM) ? 1 : 0;
  opts.cols = nullptr;
  opts.part = 0;
  opts.main_span = 0;
  if (opts.skip && w_mode == ES_W_PHASE_SPEED && fam_has_bands<FAM>() && prob->dev.use_bands) {
    // live-column list on the device, dead columns filled at once; the grid launch below is sized for all nw
    // columns (the count stays on the device), workgroups beyond the live ones return immediately
    if ((size_t)nw + 1 > ctx->cols_cap) {
      ES_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      if (ctx->d_cols) ES_HIP_CHECK(ctx, hipFree(ctx->d_cols));
      if (ctx->d_coldead) ES_HIP_CHECK(ctx, hipFree(ctx->d_coldead));
      ctx->d_cols = nullptr; ctx->d_coldead = nullptr; ctx->cols_cap = 0;
      ES_HIP_CHECK(ctx, hipMalloc(&ctx->d_cols, ((size_t)nw + 1) * sizeof(int)));
      ES_HIP_CHECK(ctx, hipMalloc(&ctx->d_coldead, (size_t)nw));
      ctx->cols_cap = (size_t)nw + 1;
    }
    hipLaunchKernelGGL(column_classify_kernel, dim3(1), dim3(1024), 0, ctx->stream, prob->dev, d_w, nw, ctx->d_cols,
                       ctx->d_coldead);
    ES_HIP_CHECK(ctx, hipGetLastError());
    const long cells = (long)nk * nw;
    const int fb = (int)((cells + 255) / 256 < 8192 ? (cells + 255) / 256 : 8192);
    hipLaunchKernelGGL(fill_dead_columns_kernel, dim3(fb), dim3(256), 0, ctx->stream, prob->dev, d_k, nk, d_w, nw,
                       ctx->d_coldead, d_D, d_rel, d_status);
    ES_HIP_CHECK(ctx, hipGetLastError());
    opts.cols = ctx->d_cols;
  }
  // families with connected continuum bands: no per-node sign tracking (band_crossed)
  const bool track = !(fam_has_bands<FAM>() && prob->dev.use_bands);
  GridShape shape = pick_shape<FAM>(nw, track);
  const int T = shape_threads(nw, shape.pts);
  const long tiles = (long)nk * ((nw + (long)T * shape.pts - 1) / ((long)T * shape.pts));
  const dim3 grid = es_tile_grid(tiles);
  bool launched = false;
  es_timer_begin(ctx);
  if constexpr (fam_has_bands<FAM>()) {
    if (rows2_shape<FAM>(nw, track) && !opts.cols && nk >= 2) {
      const int pts2 = (nw + 127) / 128;
      const dim3 grid2 = es_tile_grid(((long)nk + 1) / 2);
#define ES_R2(P_)                                                                                                   \
      if (pts2 == P_) hipLaunchKernelGGL((shoot_grid_kernel_r2<FAM, P_, false, 3>), grid2, dim3(256), 0, ctx->stream, \
                                         prob->dev, d_k, nk, d_w, nw, w_mode, d_D, d_rel, d_status, opts);
      ES_R2(1) ES_R2(2) ES_R2(3) ES_R2(4)
#undef ES_R2
      launched = true;
    }
  }
  auto one = [&](auto pts_c, auto wpe_c, auto track_c) {
    constexpr int PTS = decltype(pts_c)::value, WPE = decltype(wpe_c)::value;
    constexpr bool TRACK = decltype(track_c)::value;
    if constexpr (shape_built<FAM, PTS, WPE, TRACK>() && (TRACK || fam_has_bands<FAM>())) {
      if (launched || shape.pts != PTS || shape.wpe != WPE || track != TRACK) return;
      launched = true;
      if constexpr (FAM == FAM_CYL0 && PTS == 4 && !TRACK) {
        if (opts.cols && T > 64) {
          // compacted launch: full segments in 4-wave workgroups, the remainder of each row in one-wave workgroups
          opts.part = 1;
          hipLaunchKernelGGL((shoot_grid_kernel<FAM, PTS, 256, TRACK, WPE>), grid, dim3(T), 0, ctx->stream,
                             prob->dev, d_k, nk, d_w, nw, w_mode, d_D, d_rel, d_status, opts);
          opts.part = 2;
          opts.main_span = 4 * T;
          const long tiles2 = (long)nk * (T / 64);
          hipLaunchKernelGGL((shoot_grid_kernel<FAM, 4, 64, false, 2>), es_tile_grid(tiles2),
                             dim3(64), 0, ctx->stream, prob->dev, d_k, nk, d_w, nw, w_mode, d_D, d_rel, d_status, opts);
          return;
        }
      }
      hipLaunchKernelGGL((shoot_grid_kernel<FAM, PTS, 256, TRACK, WPE>), grid, dim3(T), 0, ctx->stream, prob->dev,
                         d_k, nk, d_w, nw, w_mode, d_D, d_rel, d_status, opts);
    }
  };
  using std::integral_constant;
#define ES_SHAPE(P_, W_)                                                                                            \
  one(integral_constant<int, P_>{}, integral_constant<int, W_>{}, integral_constant<bool, false>{});                \
  one(integral_constant<int, P_>{}, integral_constant<int, W_>{}, integral_constant<bool, true>{});
  ES_SHAPE(4, 2) ES_SHAPE(4, 3) ES_SHAPE(4, 4) ES_SHAPE(3, 2) ES_SHAPE(3, 3) ES_SHAPE(2, 2) ES_SHAPE(2, 3) ES_SHAPE(2, 4) ES_SHAPE(1, 2) ES_SHAPE(1, 3) ES_SHAPE(1, 4)
#undef ES_SHAPE
  es_timer_end(ctx);
  if (!launched) {
    ctx->last_error = "grid launch shape not built (ES_GRID_SHAPE names a shape outside this build)";
    return ES_ERR_UNSUPPORTED;
  }
  ES_HIP_CHECK(ctx, hipGetLastError());
  return ES_SUCCESS;
}

template <int FAM>
int launch_points(es_context* ctx, const es_problem* prob, const double* d_k, const double* d_w, int n,
                  double* d_D, double* d_rel, uint8_t* d_status) {
  hipLaunchKernelGGL((shoot_points_kernel<FAM>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, prob->dev, d_k,
                     d_w, n, d_D, d_rel, d_status);
  ES_HIP_CHECK(ctx, hipGetLastError());
  return ES_SUCCESS;
}

// d_n: bracket count in device memory (nullptr: n_max IS the count); n_max: launch bound (count known on the host, or the
// table capacity); n_hint: what the count is expected to be (selects between variants that give identical results)
template <int FAM>
int launch_refine(es_context* ctx, const es_problem* prob, const es_root_table& tab, double* d_lo, double* d_hi,
                  const int* d_n, int n_max, int n_hint, int n_bisect, double tol) {
  // (LANES+1)-section rounds equivalent to n_bisect halvings: (LANES+1)^R >= 2^n_bisect
  int sections = kRefineSections;
  if (const char* ev = getenv("ES_REFINE_SECTIONS")) {            // tuning aid, honoured by the port as well: 5, 9 or 17
    const int v = atoi(ev);
    if (v == 5 || v == 9 || v == 17) sections = v;
  }
  int rounds = 0;
  for (double span = 1.0, need = ldexp(1.0, n_bisect < 1000 ? n_bisect : 1000); span < need; span *= (double)sections) ++rounds;
  // section rounds with LANES lanes per bracket, then the polish steps with one lane per bracket (d_lo / d_hi carry D at
  // the ends of the narrowed bracket from one kernel to the other)
  const int np = (ES_REFINE_POLISH > 0 && rounds > 0) ? -1 : ES_REFINE_POLISH;
  // 17-section of the untwisted cylinder with np < 0 (sections only): node entries shared inside the wave, 32 steps per
  // chunk (14.6 KB of LDS per wave).  Bit-identical to the per-lane entries, so the choice could follow the bracket count
  // (ES_REFINE_SHARED_MIN); measured in round 3 with the count on the device (same box, tile of an E-GPU run, ms per
  // step shared / per-lane: E = 1 22.05 / 22.29, E = 2 11.35 / 11.28, E = 4 5.85 / 5.98, E = 8 3.19 / 3.29): no
  // crossover worth a rule, the shared form is the default at every count.  The twisted family (16 entries per node:
  // 17 KB per wave at 16 steps per chunk) lost 7 % on configs[4] in round 2 and, measured again with the cheaper round-3
  // coefficient set (67.5 KB of LDS per workgroup, 181 VGPRs), 11 % (86.2 against 77.4 ms per step): per-lane entries stay.
  constexpr int CHR = (FAM == FAM_CYL0) ? 32 : 0;
  int shared_min = kSharedMin;
  if (const char* ev = getenv("ES_REFINE_SHARED_MIN")) shared_min = atoi(ev);
  const bool shared_entries = CHR > 0 && np < 0 && n_hint >= shared_min && !getenv("ES_REFINE_PRIVATE_ENTRIES");
  auto blocks = [&](int per_wg) { return dim3((n_max + per_wg - 1) / per_wg); };
  // One section round / polish step per launch for the twisted family (refine_kernel: 255 -> 194 registers; configs[4]
  // 74.2 -> 72.6 ms per step on one box).  Measured for the others too, registers 167 -> 102: no gain on configs[2] and [3]
  // (7.58 / 7.58, 20.81 / 20.82 ms), a loss on configs[1] (2.92 -> 3.05 ms: six short launches per search instead of two);
  // a register cap of 168 for the twisted kernels (three waves per SIMD, 14 - 18 spilled values) made no difference
  // (72.4 / 72.6).  ES_REFINE_ROUNDS_IN_KERNEL=1: the single launch (A/B aid).
  constexpr bool ONE_ROUND_FAM = (FAM == FAM_CYLT || FAM == FAM_CYL0);
  const bool one_round = ONE_ROUND_FAM && !getenv("ES_REFINE_ROUNDS_IN_KERNEL");
#define ES_REFINE(LANES_, CHR_, SO_, PER_WG_)                                                                           \
  if (SO_ && one_round) {                                                                                               \
    if constexpr (ONE_ROUND_FAM && SO_) {                                                                               \
      for (int r_ = 0; r_ < rounds; ++r_)                                                                               \
        hipLaunchKernelGGL((refine_kernel<FAM, LANES_, CHR_, true, true>), blocks(PER_WG_), dim3(64 * REFINE_WAVES), 0, \
                           ctx->stream, prob->dev, tab, d_lo, d_hi, d_n, n_max, 1, np, tol);                            \
    }                                                                                                                   \
  } else                                                                                                                \
    hipLaunchKernelGGL((refine_kernel<FAM, LANES_, CHR_, SO_, false>), blocks(PER_WG_), dim3(64 * REFINE_WAVES), 0,     \
                       ctx->stream, prob->dev, tab, d_lo, d_hi, d_n, n_max, rounds, np, tol)
  if (sections == 17 && shared_entries) ES_REFINE(16, CHR, true, 4 * REFINE_WAVES);       // shared entries imply np < 0
  else if (np < 0) {
    if (sections == 17) ES_REFINE(16, 0, true, 4 * REFINE_WAVES);
    else if (sections == 9) ES_REFINE(8, 0, true, 8 * REFINE_WAVES);
    else ES_REFINE(4, 0, true, 16 * REFINE_WAVES);
  } else {
    if (sections == 17) ES_REFINE(16, 0, false, 4 * REFINE_WAVES);
    else if (sections == 9) ES_REFINE(8, 0, false, 8 * REFINE_WAVES);
    else ES_REFINE(4, 0, false, 16 * REFINE_WAVES);
  }
#undef ES_REFINE
  ES_HIP_CHECK(ctx, hipGetLastError());
  if (np < 0) {
    if (one_round) {
      if constexpr (ONE_ROUND_FAM) {
        for (int p_ = 0; p_ < ES_REFINE_POLISH; ++p_)
          hipLaunchKernelGGL((refine_polish_kernel<FAM, true>), blocks(64 * REFINE_WAVES), dim3(64 * REFINE_WAVES), 0, ctx->stream,
                             prob->dev, tab, d_lo, d_hi, d_n, n_max, p_ == ES_REFINE_POLISH - 1 ? 1 : 0, tol);
      }
    } else
      hipLaunchKernelGGL((refine_polish_kernel<FAM, false>), blocks(64 * REFINE_WAVES), dim3(64 * REFINE_WAVES), 0, ctx->stream,
                         prob->dev, tab, d_lo, d_hi, d_n, n_max, ES_REFINE_POLISH, tol);
    ES_HIP_CHECK(ctx, hipGetLastError());
  }
  return ES_SUCCESS;
}

int dispatch_refine(es_context* ctx, const es_problem* prob, const es_root_table& tab, const int* d_n, int n_max, int n_hint,
                    int n_bisect, double tol) {
  // the d_w / d_resid columns double as scratch for D at the two bracket ends until refinement overwrites them
  switch (prob->dev.family) {
    case FAM_CYL0: return launch_refine<FAM_CYL0>(ctx, prob, tab, tab.d_w, tab.d_resid, d_n, n_max, n_hint, n_bisect, tol);
    case FAM_CYLT: return launch_refine<FAM_CYLT>(ctx, prob, tab, tab.d_w, tab.d_resid, d_n, n_max, n_hint, n_bisect, tol);
    case FAM_SLABD: return launch_refine<FAM_SLABD>(ctx, prob, tab, tab.d_w, tab.d_resid, d_n, n_max, n_hint, n_bisect, tol);
    case FAM_SLABF: return launch_refine<FAM_SLABF>(ctx, prob, tab, tab.d_w, tab.d_resid, d_n, n_max, n_hint, n_bisect, tol);
    default: return ES_ERR_UNSUPPORTED;
  }
}

#define ES_DISPATCH_FAMILY(fam, CALL)                      \
  switch (fam) {                                           \
    case FAM_CYL0: return CALL(FAM_CYL0);                  \
    case FAM_CYLT: return CALL(FAM_CYLT);                  \
    case FAM_SLABD: return CALL(FAM_SLABD);                \
    case FAM_SLABF: return CALL(FAM_SLABF);                \
    default: return ES_ERR_UNSUPPORTED;                    \
  }

}  // namespace

// ---- problem creation: pack the k-independent base fields (host, fp64) and upload them -----------------------
extern "C" int es_problem_create(es_context* ctx, const es_shoot_desc* d, const es_profiles* pr, es_problem** out) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  ES_REQUIRE(ctx, d && pr && out, "null pointer");
  *out = nullptr;
  ES_REQUIRE(ctx, d->n_nodes >= 2 && d->n_nodes <= (1 << 24), "n_nodes out of range");
  ES_REQUIRE(ctx, d->geometry >= 0 && d->geometry <= 3, "geometry");
  ES_REQUIRE(ctx, d->x_boundary == -1.0 || d->x_boundary == 1.0, "x_boundary must be -1 or +1");
  const int N = d->n_nodes, npts = 2 * N - 1;
  int nb = 0;
  switch (d->geometry) {
    case ES_GEOM_CYLINDER: nb = FamTraits<FAM_CYL0>::NB; break;
    case ES_GEOM_CYLINDER_TWIST: nb = FamTraits<FAM_CYLT>::NB; break;
    case ES_GEOM_SLAB_DENSITY: nb = FamTraits<FAM_SLABD>::NB; break;
    default: nb = FamTraits<FAM_SLABF>::NB; break;
  }
  std::vector<double> base((size_t)nb * npts);
  auto B = [&](int f, int i) -> double& { return base[(size_t)f * npts + i]; };
  if (d->geometry == ES_GEOM_CYLINDER || d->geometry == ES_GEOM_CYLINDER_TWIST) {
    ES_REQUIRE(ctx, pr->r && pr->rho && pr->c2 && pr->Bz, "cylinder profiles r, rho, c2, Bz required");
    ES_REQUIRE(ctx, d->axis_bc >= 0 && d->axis_bc <= 2, "axis_bc");
    ES_REQUIRE(ctx, d->c1_power == 1 || d->c1_power == 2, "c1_power");
    ES_REQUIRE(ctx, d->m >= 0 && d->m_ext >= 0 && d->m_ext <= 64, "m");
    for (int i = 0; i < npts; ++i) {
      const double r = pr->r[i], rho = pr->rho[i], c2 = pr->c2[i], Bz = pr->Bz[i];
      const double Bphi = pr->Bphi ? pr->Bphi[i] : 0.0;
      const double vz = pr->vz ? pr->vz[i] : 0.0;
      const double vphi = pr->vphi ? pr->vphi[i] : 0.0;
      const double sr = sqrt(rho);
      const double bA = Bz / sr;                    // (k B_z)/sqrt(rho) per unit k, CF:581
      const double vA = (Bz + Bphi) / sr;           // vA_i as written, CF:173-174
      const double S = c2 + vA * vA;
      const double q = c2 / S;
      if (d->geometry == ES_GEOM_CYLINDER) {
        ES_REQUIRE(ctx, Bphi == 0.0 && vphi == 0.0, "ES_GEOM_CYLINDER needs Bphi = vphi = 0 (use CYLINDER_TWIST)");
        B(C0_VZ, i) = vz;
        B(C0_BA, i) = bA;
        B(C0_Q, i) = q;
        B(C0_A1, i) = rho / r;
        B(C0_B1, i) = r / (rho * S);
        B(C0_E1, i) = 1.0 / (r * rho);
        B(C0_E2, i) = r / rho;
      } else {
        // the k-independent parts of the node entries, in the operation order make_entry used per lane and node
        const double dm = (double)d->m, dm2 = dm * dm;
        const double invr = 1.0 / r, invr2 = invr * invr;
        const double bphr = Bphi / r, vphr = vphi / r;
        const double Bphi_n = bphr * r, vphi_n = vphr * r;   // as the device formed them from the stored B_phi/r, v_phi/r
        B(CT_MB, i) = dm * bphr;
        B(CT_BZ, i) = Bz;
        B(CT_BA, i) = bA;
        B(CT_MV, i) = dm * vphr;
        B(CT_VZ, i) = vz;
        B(CT_Q, i) = q;
        B(CT_E3, i) = rho * S;
        B(CT_RHO, i) = rho;
        B(CT_E5, i) = rho * vphi_n * vphi_n * invr;
        B(CT_E6, i) = 2.0 * Bphi_n * Bphi_n * invr;
        B(CT_C7, i) = 2.0 * Bphi_n * vphi_n;
        B(CT_INVR, i) = invr;
        B(CT_BPHI, i) = Bphi_n;
        B(CT_E9, i) = rho * vphi_n;
        B(CT_E10, i) = 2.0 * dm * S * invr2;
        B(CT_S, i) = S;
        B(CT_M2R2, i) = dm2 * invr2;
        B(CT_RDC3, i) = pr->rdC3 ? pr->rdC3[i] : 0.0;
        B(CT_E13, i) = 4.0 * S * invr2;
        B(CT_R, i) = r;
      }
    }
  } else if (d->geometry == ES_GEOM_SLAB_DENSITY) {
    ES_REQUIRE(ctx, pr->rho && pr->c2 && pr->vA2, "slab density profiles rho, c2, vA2 required");
    for (int i = 0; i < npts; ++i) {
      B(SD_RHO, i) = pr->rho[i];
      B(SD_C2, i) = pr->c2[i];
      B(SD_VA2, i) = pr->vA2[i];
    }
  } else {
    ES_REQUIRE(ctx, pr->U, "slab flow profile U required");
    for (int i = 0; i < npts; ++i) {
      B(SF_U, i) = pr->U[i];
      B(SF_DU, i) = pr->dU ? pr->dU[i] : 0.0;
      B(SF_DDU, i) = pr->ddU ? pr->ddU[i] : 0.0;
    }
  }
  es_problem* p = new es_problem();
  p->desc = *d;
  ShootDev& S = p->dev;
  memset(&S, 0, sizeof(S));
  S.family = d->geometry;
  S.n_nodes = N;
  S.npts = npts;
  S.xb = d->x_boundary;
  S.h = (d->x_end - d->x_boundary) / (double)(N - 1);
  S.rho_e = d->rho_e;
  S.vAe2 = d->vA_e * d->vA_e;
  S.ce2 = d->c_e * d->c_e;
  S.cTe2 = d->cT_e * d->cT_e;
  S.Se = S.vAe2 + S.ce2;
  S.U_e = d->U_e;
  S.R_factor = d->L_factor * 2.0 * 3.14159265358979323846;
  S.ic0 = d->ic_value;
  S.ic1 = d->ic_slope;
  S.m = d->m; S.m_ext = d->m_ext; S.axis_bc = d->axis_bc; S.c1_power = d->c1_power;
  S.bc_const = S.bc_const_raw = d->bc_const;
  if (d->geometry == FAM_CYL0 && d->bc_const != 0.0) {
    // the fp64 marches of this family deliver z times adjoint_scale (a factor 3 per step, an exact power of two back per
    // LDS chunk): a non-zero target of the axis condition carries the same factor
    double c = 1.0;
    const int nsteps = N - 1;
    for (int ch = (nsteps + es_shoot_shared::CH - 1) / es_shoot_shared::CH - 1; ch >= 0; --ch) {
      const int c0 = ch * es_shoot_shared::CH;
      const int nst = (nsteps - c0 < es_shoot_shared::CH) ? (nsteps - c0) : es_shoot_shared::CH;
      for (int i = 0; i < nst; ++i) c *= 3.0;
      c = ldexp(c, adjoint_rescale_exp(nsteps - c0 - nst, nsteps - c0));
    }
    S.bc_const = d->bc_const * c;
  }
  S.slab_sign = (d->slab_mode == ES_SLAB_MODE_SAUSAGE) ? -1.0 : 1.0;
  S.c2_i = d->c_i * d->c_i;
  S.vA2_i = d->vA_i * d->vA_i;
  S.S_i = S.c2_i + S.vA2_i;
  S.cT2_i = (S.S_i > 0.0) ? S.c2_i * S.vA2_i / S.S_i : 0.0;
  S.rho_i = d->rho_i;
  S.accept_norm = d->accept_norm;
  if (d->geometry == ES_GEOM_CYLINDER || d->geometry == ES_GEOM_SLAB_FLOW || d->geometry == ES_GEOM_SLAB_DENSITY) {
    // continuum bands in phase speed (band_crossed): node j is inside band t iff centre_j - a_j < W < centre_j + a_j
    //   cylinder:     centre = v_z, a = |bA| (Alfven), |bA| sqrt(q) (cusp)
    //   flow slab:    centre = U,   a = c_i, cT_i, vA_i (uniform) and the half line W < U_j (sign of Om)
    //   density slab: centre = 0,   a = c_j, cT_j, vA_j
    const bool cyl = (d->geometry == ES_GEOM_CYLINDER), flow = (d->geometry == ES_GEOM_SLAB_FLOW);
    S.use_bands = 1;
    S.n_bands = cyl ? 2 : (flow ? 4 : 3);
    const double slab_a[3] = {sqrt(S.c2_i), sqrt(S.cT2_i), sqrt(S.vA2_i)};
    for (int t = 0; t < S.n_bands; ++t) {
      double lo_min = INFINITY, lo_max = -INFINITY, hi_min = INFINITY, hi_max = -INFINITY, lo_prev = 0.0, hi_prev = 0.0;
      const bool half_line = (flow && t == 3);
      bool never = !cyl;                                   // slab term whose speed is zero at every node
      for (int i = 0; i < npts; ++i) {
        double centre, a;
        if (cyl) {
          centre = B(C0_VZ, i);
          a = fabs(B(C0_BA, i)) * (t == 0 ? 1.0 : sqrt(B(C0_Q, i)));
        } else if (flow) {
          centre = B(SF_U, i);
          a = half_line ? 0.0 : slab_a[t];
        } else {
          const double c2 = B(SD_C2, i), vA2 = B(SD_VA2, i);
          centre = 0.0;
          a = sqrt(t == 0 ? c2 : (t == 1 ? c2 * vA2 / (c2 + vA2) : vA2));
        }
        if (a > 0.0 || half_line) never = false;
        const double lo = half_line ? -INFINITY : centre - a, hi = centre + a;
        if (!std::isfinite(hi) || (cyl && !(a > 0.0))) S.use_bands = 0;              // empty / undefined interval
        if (i > 0 && !half_line && a > 0.0 && !(lo < hi_prev && lo_prev < hi)) S.use_bands = 0;   // disjoint neighbours
        lo_min = fmin(lo_min, lo); lo_max = fmax(lo_max, lo);
        hi_min = fmin(hi_min, hi); hi_max = fmax(hi_max, hi);
        lo_prev = lo; hi_prev = hi;
      }
      S.band[t][0] = lo_min; S.band[t][1] = lo_max; S.band[t][2] = hi_min; S.band[t][3] = hi_max;
      // a speed of zero (e.g. cT = 0 without field): the term never changes sign -> a band that nothing satisfies
      if (never) { S.band[t][0] = INFINITY; S.band[t][3] = -INFINITY; }
    }
    if (getenv("ES_FORCE_SIGN_TRACKING")) S.use_bands = 0;
  }
  if (hipSetDevice(ctx->device) != hipSuccess ||
      hipMalloc(&p->d_base, base.size() * sizeof(double)) != hipSuccess) {
    ctx->last_error = "hipMalloc(base table) failed";
    delete p;
    return ES_ERR_HIP;
  }
  if (hipMemcpyAsync(p->d_base, base.data(), base.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) {
    ctx->last_error = "upload of base table failed";
    (void)hipFree(p->d_base);
    delete p;
    return ES_ERR_HIP;
  }
  S.base = p->d_base;
  *out = p;
  return ES_SUCCESS;
}

extern "C" int es_problem_destroy(es_context* ctx, es_problem* prob) {
  if (!prob) return ES_SUCCESS;
  if (ctx) (void)hipSetDevice(ctx->device);
  if (prob->d_base) (void)hipFree(prob->d_base);
  delete prob;
  return ES_SUCCESS;
}

extern "C" int es_shoot_eval_grid_ex(es_context* ctx, const es_problem* prob, const double* d_k, int nk,
                                     const double* d_w, int nw, int w_mode, int flags, double* d_D, double* d_rel,
                                     uint8_t* d_status) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  int rc = check_problem(ctx, prob);
  if (rc) return rc;
  ES_REQUIRE(ctx, nk >= 0 && nw >= 0, "negative size");
  ES_REQUIRE(ctx, w_mode >= 0 && w_mode <= 2, "w_mode");
  ES_REQUIRE(ctx, (flags & ~ES_EVAL_SKIP_CONTINUUM) == 0, "unknown flags");
  if (nk == 0 || nw == 0) return ES_SUCCESS;
  ES_REQUIRE(ctx, d_k && d_w && d_D && d_status, "null pointer");
  ES_HIP_CHECK(ctx, hipSetDevice(ctx->device));
#define CALL_GRID(F) launch_grid<F>(ctx, prob, d_k, nk, d_w, nw, w_mode, d_D, d_rel, d_status, flags)
  ES_DISPATCH_FAMILY(prob->dev.family, CALL_GRID)
#undef CALL_GRID
}

extern "C" int es_shoot_grid_shape(es_context* ctx, const es_problem* prob, int nw, int* h_pts, int* h_wpe, int* h_track) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  int rc = check_problem(ctx, prob);
  if (rc) return rc;
  ES_REQUIRE(ctx, h_pts && h_wpe && h_track && nw > 0, "shape query arguments");
  GridShape g{0, 0};
  bool track = true;
  switch (prob->dev.family) {
    case FAM_CYL0: track = !prob->dev.use_bands; g = pick_shape<FAM_CYL0>(nw, track); break;
    case FAM_CYLT: g = pick_shape<FAM_CYLT>(nw, true); break;
    case FAM_SLABD: track = !prob->dev.use_bands; g = pick_shape<FAM_SLABD>(nw, track); break;
    case FAM_SLABF: track = !prob->dev.use_bands; g = pick_shape<FAM_SLABF>(nw, track); break;
    default: return ES_ERR_UNSUPPORTED;
  }
  *h_pts = g.pts; *h_wpe = g.wpe; *h_track = track ? 1 : 0;
  if (prob->dev.family != FAM_CYLT && rows2_shape<FAM_CYL0>(nw, track)) { *h_pts = -((nw + 127) / 128); *h_wpe = 3; }
  return ES_SUCCESS;
}

extern "C" int es_shoot_eval_grid(es_context* ctx, const es_problem* prob, const double* d_k, int nk,
                                  const double* d_w, int nw, int w_mode, double* d_D, double* d_rel,
                                  uint8_t* d_status) {
  return es_shoot_eval_grid_ex(ctx, prob, d_k, nk, d_w, nw, w_mode, 0, d_D, d_rel, d_status);
}

extern "C" int es_shoot_eval_points(es_context* ctx, const es_problem* prob, const double* d_k, const double* d_w,
                                    int n, double* d_D, double* d_rel, uint8_t* d_status) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  int rc = check_problem(ctx, prob);
  if (rc) return rc;
  ES_REQUIRE(ctx, n >= 0, "negative size");
  if (n == 0) return ES_SUCCESS;
  ES_REQUIRE(ctx, d_k && d_w && d_D && d_status, "null pointer");
  ES_HIP_CHECK(ctx, hipSetDevice(ctx->device));
#define CALL_PTS(F) launch_points<F>(ctx, prob, d_k, d_w, n, d_D, d_rel, d_status)
  ES_DISPATCH_FAMILY(prob->dev.family, CALL_PTS)
#undef CALL_PTS
}

namespace {
// flag + scan + emit + refine, everything enqueued, no host synchronisation: the bracket count stays in ctx->d_total
int find_roots_enqueue(es_context* ctx, const es_problem* prob, const double* d_k, int nk, const double* d_w, int nw,
                       int w_mode, const double* d_D, const uint8_t* d_status, int n_bisect, double tol_percent,
                       const es_root_table* table) {
  const long cells = (long)nk * nw;
  int rc = es_ensure_scan_scratch(ctx, (size_t)cells);
  if (rc) return rc;
  const int nblocks = (int)((cells + 255) / 256);
  hipLaunchKernelGGL(bracket_flag_kernel, dim3(nblocks), dim3(256), 0, ctx->stream, d_D, d_status, nw, cells,
                     ctx->d_masks, ctx->d_block_counts);
  ES_HIP_CHECK(ctx, hipGetLastError());
  rc = es_scan_block_counts_async(ctx, nblocks);
  if (rc) return rc;
  if (table->capacity > 0) {
    hipLaunchKernelGGL(bracket_emit_kernel, dim3(nblocks), dim3(256), 0, ctx->stream, d_k, d_w, nw, w_mode, cells,
                       d_D, ctx->d_masks, ctx->d_block_counts, *table, table->d_w, table->d_resid);
    ES_HIP_CHECK(ctx, hipGetLastError());
  }
  return ES_SUCCESS;
}

int check_find_roots_args(es_context* ctx, const es_problem* prob, int nk, int nw, int w_mode, int n_bisect,
                          const es_root_table* table) {
  int rc = check_problem(ctx, prob);
  if (rc) return rc;
  ES_REQUIRE(ctx, table, "null pointer");
  ES_REQUIRE(ctx, nk >= 0 && nw >= 0 && n_bisect >= 0 && table->capacity >= 0, "negative size");
  ES_REQUIRE(ctx, w_mode >= 0 && w_mode <= 2, "w_mode");
  ES_REQUIRE(ctx, table->capacity == 0 || (table->d_k && table->d_w && table->d_w_lo && table->d_w_hi &&
                                           table->d_resid && table->d_row && table->d_flag),
             "null root table arrays");
  return ES_SUCCESS;
}
}  // namespace

extern "C" int es_shoot_find_roots(es_context* ctx, const es_problem* prob, const double* d_k, int nk,
                                   const double* d_w, int nw, int w_mode, const double* d_D,
                                   const uint8_t* d_status, int n_bisect, double tol_percent, es_root_table* table,
                                   int* h_count) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  ES_REQUIRE(ctx, h_count, "null pointer");
  int rc = check_find_roots_args(ctx, prob, nk, nw, w_mode, n_bisect, table);
  if (rc) return rc;
  *h_count = 0;
  if ((long)nk * nw == 0) return ES_SUCCESS;
  ES_REQUIRE(ctx, d_k && d_w && d_D && d_status, "null pointer");
  ES_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  rc = find_roots_enqueue(ctx, prob, d_k, nk, d_w, nw, w_mode, d_D, d_status, n_bisect, tol_percent, table);
  if (rc) return rc;
  // the one read-back of this entry point: the count it returns through a host pointer also sizes the refinement launch
  ES_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_total, ctx->d_total, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ES_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  const int total = *ctx->h_total;
  *h_count = total;
  const int n = total < table->capacity ? total : table->capacity;
  if (n > 0) {
    rc = dispatch_refine(ctx, prob, *table, nullptr, n, n, n_bisect, tol_percent);
    if (rc) return rc;
  }
  return total > table->capacity ? ES_ERR_CAPACITY : ES_SUCCESS;
}

extern "C" int es_shoot_find_roots_async(es_context* ctx, const es_problem* prob, const double* d_k, int nk,
                                         const double* d_w, int nw, int w_mode, const double* d_D,
                                         const uint8_t* d_status, int n_bisect, double tol_percent,
                                         es_root_table* table, int32_t* d_count) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  ES_REQUIRE(ctx, d_count, "null pointer");
  int rc = check_find_roots_args(ctx, prob, nk, nw, w_mode, n_bisect, table);
  if (rc) return rc;
  ES_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if ((long)nk * nw == 0) {
    ES_HIP_CHECK(ctx, hipMemsetAsync(d_count, 0, sizeof(int32_t), ctx->stream));
    return ES_SUCCESS;
  }
  ES_REQUIRE(ctx, d_k && d_w && d_D && d_status, "null pointer");
  rc = find_roots_enqueue(ctx, prob, d_k, nk, d_w, nw, w_mode, d_D, d_status, n_bisect, tol_percent, table);
  if (rc) return rc;
  ES_HIP_CHECK(ctx, hipMemcpyAsync(d_count, ctx->d_total, sizeof(int32_t), hipMemcpyDeviceToDevice, ctx->stream));
  if (table->capacity > 0) {
    // refinement sized for the table capacity, count taken from the caller's device word (ctx->d_total is reused by the
    // next call on this context); the expected count for the variant choice: half the capacity (callers size the
    // table at about twice the brackets they expect)
    rc = dispatch_refine(ctx, prob, *table, d_count, table->capacity, table->capacity / 2, n_bisect, tol_percent);
    if (rc) return rc;
  }
  return ES_SUCCESS;
}


// ---- fp32 screening + fp64 refinement (configs[4]) ---------------------------------------------------------------------
namespace {
template <int FAM>
int launch_grid_f32(es_context* ctx, const es_problem* prob, const double* d_k, int nk, const double* d_w, int nw,
                    int w_mode, double* d_D, uint8_t* d_status) {
  if constexpr (FAM == FAM_CYL0 || FAM == FAM_CYLT) {
    es_timer_begin(ctx);
    constexpr int PTS = 4;
    int T = ((nw + PTS - 1) / PTS + 63) / 64 * 64;
    if (T < 64) T = 64;
    if (T > 256) T = 256;
    const long tiles = (long)nk * ((nw + T * PTS - 1) / (T * PTS));
    const dim3 grid = es_tile_grid(tiles);
    const bool bands = fam_has_bands<FAM>() && prob->dev.use_bands;
    // register caps: 128 VGPRs (4 waves per SIMD) for the untwisted family, 168 (3 waves) for the twisted one
    if constexpr (FAM == FAM_CYL0) {
      int v0 = 0;
      if (const char* ev = getenv("ES_F32_VARIANT")) v0 = atoi(ev);             // tuning aid
      if (bands && v0 == 1)
        hipLaunchKernelGGL((shoot_grid_f32_kernel<FAM, PTS, 256, false, 3>), grid, dim3(T), 0, ctx->stream,
                           prob->dev, d_k, nk, d_w, nw, w_mode, d_D, d_status);
      else if (bands && v0 == 2)
        hipLaunchKernelGGL((shoot_grid_f32_kernel<FAM, PTS, 256, false, 2>), grid, dim3(T), 0, ctx->stream,
                           prob->dev, d_k, nk, d_w, nw, w_mode, d_D, d_status);
      else if (bands && v0 == 3) {
        int T8 = ((nw + 7) / 8 + 63) / 64 * 64;
        if (T8 < 64) T8 = 64;
        if (T8 > 256) T8 = 256;
        const long t8 = (long)nk * ((nw + T8 * 8 - 1) / (T8 * 8));
        hipLaunchKernelGGL((shoot_grid_f32_kernel<FAM, 8, 256, false, 2>), es_tile_grid(t8),
                           dim3(T8), 0, ctx->stream, prob->dev, d_k, nk, d_w, nw, w_mode, d_D, d_status);
      } else if (bands)
        hipLaunchKernelGGL((shoot_grid_f32_kernel<FAM, PTS, 256, false, 4>), grid, dim3(T), 0, ctx->stream,
                           prob->dev, d_k, nk, d_w, nw, w_mode, d_D, d_status);
      else
        hipLaunchKernelGGL((shoot_grid_f32_kernel<FAM, PTS, 256, true, 4>), grid, dim3(T), 0, ctx->stream,
                           prob->dev, d_k, nk, d_w, nw, w_mode, d_D, d_status);
    } else {
      // measured on configs[4] (1024^2, N = 2000), packed fp32: 4 points per lane at 2 waves per SIMD (no spills) 4.5 ms;
      // the same capped at 168 registers (3 waves, spills) 5.6 ms; 2 points per lane at 4 waves 4.7 ms; fp64 9.4 ms
      int variant = 1;
      if (const char* ev = getenv("ES_F32_VARIANT")) variant = atoi(ev);        // tuning aid
      if (variant == 1) {
        hipLaunchKernelGGL((shoot_grid_f32_kernel<FAM, PTS, 256, true, 2>), grid, dim3(T), 0, ctx->stream,
                           prob->dev, d_k, nk, d_w, nw, w_mode, d_D, d_status);
      } else if (variant == 2) {
        int T2 = ((nw + 1) / 2 + 63) / 64 * 64;
        if (T2 < 64) T2 = 64;
        if (T2 > 256) T2 = 256;
        const long tiles2 = (long)nk * ((nw + T2 * 2 - 1) / (T2 * 2));
        hipLaunchKernelGGL((shoot_grid_f32_kernel<FAM, 2, 256, true, 4>), es_tile_grid(tiles2),
                           dim3(T2), 0, ctx->stream, prob->dev, d_k, nk, d_w, nw, w_mode, d_D, d_status);
      } else {
        hipLaunchKernelGGL((shoot_grid_f32_kernel<FAM, PTS, 256, true, 3>), grid, dim3(T), 0, ctx->stream,
                           prob->dev, d_k, nk, d_w, nw, w_mode, d_D, d_status);
      }
    }
    es_timer_end(ctx);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_SUCCESS;
  } else {
    // slab families: band problems only (their status is exact from the phase speed; a profile whose node intervals do not
    // overlap falls back to per-node tracking in fp64, which the fp32 pass has no counterpart of)
    if (!prob->dev.use_bands) {
      ctx->last_error = "fp32 screening of a slab needs connected continuum bands (per-node sign tracking: fp64 only)";
      return ES_ERR_UNSUPPORTED;
    }
    es_timer_begin(ctx);
    constexpr int PTS = 4;
    int T = ((nw + PTS - 1) / PTS + 63) / 64 * 64;
    if (T < 64) T = 64;
    if (T > 256) T = 256;
    const long tiles = (long)nk * ((nw + T * PTS - 1) / (T * PTS));
    const dim3 grid = es_tile_grid(tiles);
    hipLaunchKernelGGL((shoot_grid_f32_kernel<FAM, PTS, 256, false, 3>), grid, dim3(T), 0, ctx->stream, prob->dev, d_k,
                       nk, d_w, nw, w_mode, d_D, d_status);
    es_timer_end(ctx);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_SUCCESS;
  }
}

template <int FAM>
int points_into(es_context* ctx, const es_problem* prob, const double* pk, const double* pw, int n, double* pD,
                uint8_t* pst) {
  return launch_points<FAM>(ctx, prob, pk, pw, n, pD, nullptr, pst);
}
}  // namespace

namespace {
int check_mixed_args(es_context* ctx, const es_problem* prob, int nk, int nw, int w_mode) {
  int rc = check_problem(ctx, prob);
  if (rc) return rc;
  ES_REQUIRE(ctx, nk >= 0 && nw >= 0, "negative size");
  ES_REQUIRE(ctx, w_mode >= 0 && w_mode <= 2, "w_mode");
  const int fam = prob->dev.family;
  if (fam < FAM_CYL0 || fam > FAM_SLABF) return ES_ERR_UNSUPPORTED;
  if ((fam == FAM_SLABD || fam == FAM_SLABF) && !prob->dev.use_bands) {
    ctx->last_error = "fp32 screening of a slab needs connected continuum bands (per-node sign tracking: fp64 only)";
    return ES_ERR_UNSUPPORTED;
  }
  return ES_SUCCESS;
}

int launch_grid_f32_any(es_context* ctx, const es_problem* prob, const double* d_k, int nk, const double* d_w, int nw, int w_mode,
                        double* d_D, uint8_t* d_status) {
  switch (prob->dev.family) {
    case FAM_CYL0: return launch_grid_f32<FAM_CYL0>(ctx, prob, d_k, nk, d_w, nw, w_mode, d_D, d_status);
    case FAM_CYLT: return launch_grid_f32<FAM_CYLT>(ctx, prob, d_k, nk, d_w, nw, w_mode, d_D, d_status);
    case FAM_SLABD: return launch_grid_f32<FAM_SLABD>(ctx, prob, d_k, nk, d_w, nw, w_mode, d_D, d_status);
    default: return launch_grid_f32<FAM_SLABF>(ctx, prob, d_k, nk, d_w, nw, w_mode, d_D, d_status);
  }
}

int points_any(es_context* ctx, const es_problem* prob, const double* pk, const double* pw, int n, double* pD, uint8_t* pst) {
  switch (prob->dev.family) {
    case FAM_CYL0: return points_into<FAM_CYL0>(ctx, prob, pk, pw, n, pD, pst);
    case FAM_CYLT: return points_into<FAM_CYLT>(ctx, prob, pk, pw, n, pD, pst);
    case FAM_SLABD: return points_into<FAM_SLABD>(ctx, prob, pk, pw, n, pD, pst);
    default: return points_into<FAM_SLABF>(ctx, prob, pk, pw, n, pD, pst);
  }
}
}  // namespace

// Step 1 of the mixed search alone: the fp32 screening march, enqueued (no read-back).
extern "C" int es_shoot_screen_grid(es_context* ctx, const es_problem* prob, const double* d_k, int nk, const double* d_w,
                                    int nw, int w_mode, double* d_D, uint8_t* d_status) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  int rc = check_mixed_args(ctx, prob, nk, nw, w_mode);
  if (rc) return rc;
  if ((long)nk * nw == 0) return ES_SUCCESS;
  ES_REQUIRE(ctx, d_k && d_w && d_D && d_status, "null pointer");
  ES_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  return launch_grid_f32_any(ctx, prob, d_k, nk, d_w, nw, w_mode, d_D, d_status);
}

extern "C" int es_shoot_find_roots_mixed(es_context* ctx, const es_problem* prob, const double* d_k, int nk,
                                         const double* d_w, int nw, int w_mode, int n_bisect, double tol_percent,
                                         double* d_D, uint8_t* d_status, es_root_table* table, int* h_count,
                                         int* h_stats) {
  int rc = es_shoot_screen_grid(ctx, prob, d_k, nk, d_w, nw, w_mode, d_D, d_status);
  if (rc) return rc;
  return es_shoot_find_roots_screened(ctx, prob, d_k, nk, d_w, nw, w_mode, n_bisect, tol_percent, d_D, d_status, table,
                                      h_count, h_stats);
}

// Steps 2 - 5 of the mixed search on a grid screened by es_shoot_screen_grid.
extern "C" int es_shoot_find_roots_screened(es_context* ctx, const es_problem* prob, const double* d_k, int nk,
                                            const double* d_w, int nw, int w_mode, int n_bisect, double tol_percent,
                                            double* d_D, uint8_t* d_status, es_root_table* table, int* h_count,
                                            int* h_stats) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  int rc = check_mixed_args(ctx, prob, nk, nw, w_mode);
  if (rc) return rc;
  ES_REQUIRE(ctx, table && h_count, "null pointer");
  ES_REQUIRE(ctx, n_bisect >= 0 && table->capacity >= 0, "negative size");
  *h_count = 0;
  if (h_stats) h_stats[0] = h_stats[1] = h_stats[2] = 0;
  const long cells = (long)nk * nw;
  if (cells == 0) return ES_SUCCESS;
  ES_REQUIRE(ctx, d_k && d_w && d_D && d_status, "null pointer");
  ES_REQUIRE(ctx, table->capacity == 0 || (table->d_k && table->d_w && table->d_w_lo && table->d_w_hi &&
                                           table->d_resid && table->d_row && table->d_flag),
             "null root table arrays");
  ES_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  // 2. unsure points -> fp64
  rc = es_ensure_scan_scratch(ctx, (size_t)cells);
  if (rc) return rc;
  const int nblocks = (int)((cells + 255) / 256);
  hipLaunchKernelGGL(unsure_flag_kernel, dim3(nblocks), dim3(256), 0, ctx->stream, d_status, cells, ctx->d_masks,
                     ctx->d_block_counts);
  ES_HIP_CHECK(ctx, hipGetLastError());
  int n_unsure = 0;
  rc = es_scan_block_counts(ctx, nblocks, &n_unsure);
  if (rc) return rc;
  auto align = [](size_t b) { return (b + 255) & ~(size_t)255; };
  auto carve = [&](size_t n, double*& pk, double*& pw, double*& pD, long*& pcell, uint8_t*& pst) -> int {
    const size_t bd = align(n * sizeof(double));
    int r = es_ensure_scratch(ctx, 4 * bd + align(n));
    if (r) return r;
    char* b = (char*)ctx->d_scratch;
    pk = (double*)b; pw = (double*)(b + bd); pD = (double*)(b + 2 * bd); pcell = (long*)(b + 3 * bd);
    pst = (uint8_t*)(b + 4 * bd);
    return ES_SUCCESS;
  };
  double *pk, *pw, *pD; long* pcell; uint8_t* pst;
  if (n_unsure > 0) {
    rc = carve((size_t)n_unsure, pk, pw, pD, pcell, pst);
    if (rc) return rc;
    hipLaunchKernelGGL(unsure_gather_kernel, dim3(nblocks), dim3(256), 0, ctx->stream, d_k, d_w, nw, w_mode, cells,
                       ctx->d_masks, ctx->d_block_counts, pk, pw, pcell);
    ES_HIP_CHECK(ctx, hipGetLastError());
    rc = points_any(ctx, prob, pk, pw, n_unsure, pD, pst);
    if (rc) return rc;
    hipLaunchKernelGGL(scatter_points_kernel, dim3((n_unsure + 255) / 256), dim3(256), 0, ctx->stream, pcell, pD, pst,
                       n_unsure, d_D, d_status);
    ES_HIP_CHECK(ctx, hipGetLastError());
  }
  // 3. brackets on the merged array (signs and statuses are now those of the fp64 path)
  hipLaunchKernelGGL(bracket_flag_kernel, dim3(nblocks), dim3(256), 0, ctx->stream, d_D, d_status, nw, cells,
                     ctx->d_masks, ctx->d_block_counts);
  ES_HIP_CHECK(ctx, hipGetLastError());
  int total = 0;
  rc = es_scan_block_counts(ctx, nblocks, &total);
  if (rc) return rc;
  *h_count = total;
  const int n = total < table->capacity ? total : table->capacity;
  int violations = 0;
  if (n > 0) {
    hipLaunchKernelGGL(bracket_emit_kernel, dim3(nblocks), dim3(256), 0, ctx->stream, d_k, d_w, nw, w_mode, cells,
                       d_D, ctx->d_masks, ctx->d_block_counts, *table, table->d_w, table->d_resid);
    ES_HIP_CHECK(ctx, hipGetLastError());
    // 4. both ends of every bracket in fp64: the refinement starts from the numbers of the fp64 path
    rc = carve((size_t)2 * n, pk, pw, pD, pcell, pst);
    if (rc) return rc;
    hipLaunchKernelGGL(bracket_ends_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, *table, n, pk, pw);
    ES_HIP_CHECK(ctx, hipGetLastError());
    rc = points_any(ctx, prob, pk, pw, 2 * n, pD, pst);
    if (rc) return rc;
    ES_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_total, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(bracket_ends_store_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, pD, pst, n,
                       table->d_w, table->d_resid, ctx->d_total);
    ES_HIP_CHECK(ctx, hipGetLastError());
    ES_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_total, ctx->d_total, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    // 5. fp64 refinement
    rc = dispatch_refine(ctx, prob, *table, nullptr, n, n, n_bisect, tol_percent);
    if (rc) return rc;
    ES_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    violations = *ctx->h_total;
  }
  if (h_stats) { h_stats[0] = n_unsure; h_stats[1] = 2 * n; h_stats[2] = violations; }
  if (violations != 0) {
    ctx->last_error = "fp32 screening: a bracket was not confirmed by the fp64 values at its ends";
    return ES_ERR_SCREENING;
  }
  return total > table->capacity ? ES_ERR_CAPACITY : ES_SUCCESS;
}


// ---- exchange record packing (multi-GPU root-table gather) ---------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void pack_records_kernel(es_root_table tab, int count_host, const int* __restrict__ d_count,
                                                           double m, const int64_t* __restrict__ rows_global, int cap,
                                                           double* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;          // row of the send buffer: 0 = header, 1.. = records
  if (i > cap) return;
  const int count = d_count ? *d_count : count_host;
  double* o = out + (size_t)i * 6;
  if (i == 0) {
    o[0] = (double)count; o[1] = o[2] = o[3] = o[4] = o[5] = 0.0;
    return;
  }
  const int r = i - 1;
  int n = count < cap ? count : cap;
  if (n > tab.capacity) n = tab.capacity;
  if (r < n) {
    const int row = tab.d_row[r];
    o[0] = tab.d_k[r]; o[1] = tab.d_w[r]; o[2] = m; o[3] = tab.d_resid[r]; o[4] = (double)tab.d_flag[r];
    o[5] = rows_global ? (double)rows_global[row] : (double)row;
  } else {
    o[0] = o[1] = o[2] = o[3] = o[4] = o[5] = 0.0;
  }
}
}  // namespace

extern "C" int es_root_table_pack(es_context* ctx, const es_root_table* table, int count, double m,
                                  const int64_t* d_rows_global, int cap, double* d_out) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  ES_REQUIRE(ctx, table && d_out && count >= 0 && cap >= 0, "pack arguments");
  ES_REQUIRE(ctx, cap == 0 || count == 0 || (table->d_k && table->d_w && table->d_resid && table->d_row && table->d_flag),
             "null root table arrays");
  ES_REQUIRE(ctx, (count < cap ? count : cap) <= table->capacity, "count exceeds the table");
  ES_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(pack_records_kernel, dim3((cap + 1 + 255) / 256), dim3(256), 0, ctx->stream, *table, count, nullptr, m,
                     d_rows_global, cap, d_out);
  ES_HIP_CHECK(ctx, hipGetLastError());
  return ES_SUCCESS;
}

extern "C" int es_root_table_pack_async(es_context* ctx, const es_root_table* table, const int32_t* d_count, double m,
                                        const int64_t* d_rows_global, int cap, double* d_out) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  ES_REQUIRE(ctx, table && d_out && d_count && cap >= 0, "pack arguments");
  ES_REQUIRE(ctx, cap == 0 || (table->d_k && table->d_w && table->d_resid && table->d_row && table->d_flag),
             "null root table arrays");
  ES_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(pack_records_kernel, dim3((cap + 1 + 255) / 256), dim3(256), 0, ctx->stream, *table, 0, d_count, m,
                     d_rows_global, cap, d_out);
  ES_HIP_CHECK(ctx, hipGetLastError());
  return ES_SUCCESS;
}
