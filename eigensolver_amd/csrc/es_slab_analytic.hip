// K1: closed-form slab dispersion relations with steady flow, one (K, W) grid point per lane, and the
// sign-change scan with ordered root compaction.
//
// Restates Slab/Non uniform flow/Solver/flow_multiprocessor.py:107-127 (m0, me, n0, disp_rel_*),
// :166-272 (scan: f(V)*f(V+step) < 0 -> root at (V + V+step)/2) and :284-303 (pole filter).
//
// Arithmetic follows the reference expression order operation by operation; this file is compiled with
// -ffp-contract=off so that no product-sum is fused where NumPy does not fuse.  sqrt / division are IEEE;
// tanh / tan come from the device math library and agree with NumPy's to a few ulp (tests state the bound).
//
// Roofline note: ~60 fp64 ops + 2 sqrt + 1 tanh/tan per evaluation against 8 B written per grid
// point -> fp64-VALU bound, not HBM bound; layout is W fastest (coalesced 512 B store per wave, K is
// wave-uniform so the K load is a scalar load).
#include "es_common.hpp"

namespace {

struct SlabP {
  double vA_i, c_i, vA_e, c_e, mach_i, mach_e, R1, cT_i, cT_e;
};

__device__ __forceinline__ double sq(double x) { return x * x; }

// m0(W), flow_multiprocessor.py:107-108
__device__ __forceinline__ double m0_arg(const SlabP& p, double W) {
  const double wi2 = sq(W - p.mach_i);
  return (sq(p.c_i) - wi2) * (sq(p.vA_i) - wi2) / ((sq(p.c_i) + sq(p.vA_i)) * (sq(p.cT_i) - wi2));
}
// me(W), :110-111
__device__ __forceinline__ double me_arg(const SlabP& p, double W) {
  const double we2 = sq(W - p.mach_e);
  return (sq(p.c_e) - we2) * (sq(p.vA_e) - we2) / ((sq(p.c_e) + sq(p.vA_e)) * (sq(p.cT_e) - we2));
}

template <int MODE>
__device__ __forceinline__ double disp_rel(const SlabP& p, double W, double K) {
  const double me = sqrt(me_arg(p, W));
  const double num = p.R1 * (sq(p.vA_e) - sq(W - p.mach_e));
  const double di = sq(p.vA_i) - sq(W - p.mach_i);
  if (MODE == ES_SLAB_SAUSAGE) {          // :117-118  R1*(..)*m0*tanh(K*m0)/(me*(..)) + 1
    const double m0 = sqrt(m0_arg(p, W));
    return num * m0 * tanh(K * m0) / (me * di) + 1.0;
  } else if (MODE == ES_SLAB_KINK) {      // :120-121  R1*(..)*m0/(tanh(K*m0)*me*(..)) + 1
    const double m0 = sqrt(m0_arg(p, W));
    return num * m0 / (tanh(K * m0) * me * di) + 1.0;
  } else if (MODE == ES_SLAB_SAUSAGE_BODY) {  // :123-124  R1*(..)*n0*tan(K*n0)/(me*(..)) - 1
    const double n0 = sqrt(fabs(m0_arg(p, W)));
    return num * n0 * tan(K * n0) / (me * di) - 1.0;
  } else {                                // :126-127  R1*(..)*n0/(tan(K*n0)*me*(..)) + 1
    const double n0 = sqrt(fabs(m0_arg(p, W)));
    return num * n0 / (tan(K * n0) * me * di) + 1.0;
  }
}

__device__ __forceinline__ double disp_rel_dyn(int mode, const SlabP& p, double W, double K) {
  switch (mode) {
    case ES_SLAB_SAUSAGE: return disp_rel<ES_SLAB_SAUSAGE>(p, W, K);
    case ES_SLAB_KINK: return disp_rel<ES_SLAB_KINK>(p, W, K);
    case ES_SLAB_SAUSAGE_BODY: return disp_rel<ES_SLAB_SAUSAGE_BODY>(p, W, K);
    default: return disp_rel<ES_SLAB_KINK_BODY>(p, W, K);
  }
}

// Everything except the tanh / tan factor depends on W only (m0, me, the density/speed prefactors): it is computed
// once per lane and reused for K_TILE rows, so a grid point costs one multiply, one tanh/tan, the products and the
// final division -- the same operations in the same order as disp_rel<MODE>() (bit-identical results).
//   sausage: ((num*m0) * tanh(K*m0)) / (me*di) + 1        kink: (num*m0) / ((tanh(K*m0)*me)*di) + 1
template <int MODE>
struct SlabColumn {
  double root;     // m0 (surface modes) or n0 (body modes)
  double A;        // num * root
  double me, di;
  __device__ __forceinline__ void init(const SlabP& p, double W) {
    me = sqrt(me_arg(p, W));
    const double num = p.R1 * (sq(p.vA_e) - sq(W - p.mach_e));
    di = sq(p.vA_i) - sq(W - p.mach_i);
    const double arg = m0_arg(p, W);
    root = (MODE == ES_SLAB_SAUSAGE || MODE == ES_SLAB_KINK) ? sqrt(arg) : sqrt(fabs(arg));
    A = num * root;
  }
  __device__ __forceinline__ double eval(double K) const {
    if (MODE == ES_SLAB_SAUSAGE) return A * tanh(K * root) / (me * di) + 1.0;
    if (MODE == ES_SLAB_KINK) return A / (tanh(K * root) * me * di) + 1.0;
    if (MODE == ES_SLAB_SAUSAGE_BODY) return A * tan(K * root) / (me * di) - 1.0;
    return A / (tan(K * root) * me * di) + 1.0;
  }
};

constexpr int K_TILE = 16;

// grid: x = ceil(nW / 256) blocks along W, y = tiles of K_TILE rows.  W on the lanes (coalesced 512 B stores per wave),
// K is workgroup-uniform (scalar load).
template <int MODE>
__global__ __launch_bounds__(256) void slab_eval_kernel(SlabP p, const double* __restrict__ Kv, int nK,
                                                        const double* __restrict__ Wv, int nW,
                                                        double* __restrict__ D) {
  const int iW = blockIdx.x * 256 + threadIdx.x;
  if (iW >= nW) return;
  SlabColumn<MODE> col;
  col.init(p, Wv[iW]);
  for (int tile = blockIdx.y; tile * K_TILE < nK; tile += gridDim.y) {
    const int k0 = tile * K_TILE;
    const int k1 = (k0 + K_TILE < nK) ? k0 + K_TILE : nK;
    for (int iK = k0; iK < k1; ++iK) D[(size_t)iK * nW + iW] = col.eval(Kv[iK]);
  }
}

// Scan: cell c = iK*nW + iW; flag = f(V)*f(V+step) < 0, exactly the product test of :174-177.
// Flags leave the kernel as one 64-bit ballot per wave plus a per-block (256 cells) count.
template <int MODE>
__global__ __launch_bounds__(256) void slab_scan_flag_kernel(SlabP p, const double* __restrict__ Kv,
                                                             const double* __restrict__ Wv, int nW, long cells,
                                                             double step, uint64_t* __restrict__ masks,
                                                             int* __restrict__ block_counts) {
  __shared__ int wave_cnt[4];
  const long c = (long)blockIdx.x * 256 + threadIdx.x;
  bool flag = false;
  if (c < cells) {
    const long iK = c / nW;
    const int iW = (int)(c - iK * nW);
    const double K = Kv[iK];
    const double V1 = Wv[iW];
    const double V2 = V1 + step;
    const double prod = disp_rel<MODE>(p, V1, K) * disp_rel<MODE>(p, V2, K);
    flag = prod < 0.0;            // NaN products compare false, as in the reference
  }
  const uint64_t m = __ballot(flag);
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    masks[c >> 6] = m;
    wave_cnt[wid] = __popcll(m);
  }
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

__global__ __launch_bounds__(256) void slab_scan_emit_kernel(const double* __restrict__ Kv,
                                                             const double* __restrict__ Wv, int nW, long cells,
                                                             double step, const uint64_t* __restrict__ masks,
                                                             const int* __restrict__ block_off,
                                                             double* __restrict__ rootK, double* __restrict__ rootW,
                                                             int capacity) {
  const long c = (long)blockIdx.x * 256 + threadIdx.x;
  if (c >= cells) return;
  const uint64_t m = masks[c >> 6];
  if (!((m >> (c & 63)) & 1ull)) return;
  const int pos = es_cell_rank(masks, block_off, c);
  if (pos >= capacity) return;
  const long iK = c / nW;
  const int iW = (int)(c - iK * nW);
  const double V1 = Wv[iW];
  const double V2 = V1 + step;
  rootK[pos] = Kv[iK];
  rootW[pos] = (V1 + V2) / 2;      // :180 midpoint_V = (V_1 + V_2) / 2
}

__global__ __launch_bounds__(256) void slab_filter_kernel(SlabP p, int mode, const double* __restrict__ rootK,
                                                          const double* __restrict__ rootW, int n, double thresh,
                                                          uint8_t* __restrict__ keep) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double v = disp_rel_dyn(mode, p, rootW[i], rootK[i]);
  keep[i] = (v < thresh) ? 1 : 0;    // :290 `if test_body_kink_sol < 0.0001` (one-sided)
}

SlabP to_dev(const es_slab_analytic_params* p) {
  SlabP s{p->vA_i, p->c_i, p->vA_e, p->c_e, p->mach_i, p->mach_e, p->R1, p->cT_i, p->cT_e};
  return s;
}

}  // namespace

extern "C" int es_slab_analytic_eval(es_context* ctx, const es_slab_analytic_params* p, int mode,
                                     const double* d_K, int nK, const double* d_W, int nW, double* d_D) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  ES_REQUIRE(ctx, nK >= 0 && nW >= 0, "negative size");
  ES_REQUIRE(ctx, p && (nK == 0 || nW == 0 || (d_K && d_W && d_D)), "null pointer");
  ES_REQUIRE(ctx, mode >= 0 && mode <= 3, "mode");
  if (nK == 0 || nW == 0) return ES_SUCCESS;
  ES_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const SlabP sp = to_dev(p);
  const int ntiles = (nK + K_TILE - 1) / K_TILE;
  dim3 grid((nW + 255) / 256, ntiles < 65535 ? ntiles : 65535), block(256);
  switch (mode) {
    case ES_SLAB_SAUSAGE: hipLaunchKernelGGL(slab_eval_kernel<ES_SLAB_SAUSAGE>, grid, block, 0, ctx->stream, sp, d_K, nK, d_W, nW, d_D); break;
    case ES_SLAB_KINK: hipLaunchKernelGGL(slab_eval_kernel<ES_SLAB_KINK>, grid, block, 0, ctx->stream, sp, d_K, nK, d_W, nW, d_D); break;
    case ES_SLAB_SAUSAGE_BODY: hipLaunchKernelGGL(slab_eval_kernel<ES_SLAB_SAUSAGE_BODY>, grid, block, 0, ctx->stream, sp, d_K, nK, d_W, nW, d_D); break;
    default: hipLaunchKernelGGL(slab_eval_kernel<ES_SLAB_KINK_BODY>, grid, block, 0, ctx->stream, sp, d_K, nK, d_W, nW, d_D); break;
  }
  ES_HIP_CHECK(ctx, hipGetLastError());
  return ES_SUCCESS;
}

extern "C" int es_slab_analytic_scan(es_context* ctx, const es_slab_analytic_params* p, int mode,
                                     const double* d_K, int nK, const double* d_W, int nW, double step,
                                     double* d_rootK, double* d_rootW, int capacity, int* h_count) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  ES_REQUIRE(ctx, nK >= 0 && nW >= 0 && capacity >= 0, "negative size");
  ES_REQUIRE(ctx, p && h_count && (nK == 0 || nW == 0 || (d_K && d_W)), "null pointer");
  ES_REQUIRE(ctx, capacity == 0 || (d_rootK && d_rootW), "null output with capacity > 0");
  ES_REQUIRE(ctx, mode >= 0 && mode <= 3, "mode");
  *h_count = 0;
  const long cells = (long)nK * nW;
  if (cells == 0) return ES_SUCCESS;
  ES_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int rc = es_ensure_scan_scratch(ctx, (size_t)cells);
  if (rc != ES_SUCCESS) return rc;
  const SlabP sp = to_dev(p);
  const int nblocks = (int)((cells + 255) / 256);
  dim3 grid(nblocks), block(256);
  switch (mode) {
    case ES_SLAB_SAUSAGE: hipLaunchKernelGGL(slab_scan_flag_kernel<ES_SLAB_SAUSAGE>, grid, block, 0, ctx->stream, sp, d_K, d_W, nW, cells, step, ctx->d_masks, ctx->d_block_counts); break;
    case ES_SLAB_KINK: hipLaunchKernelGGL(slab_scan_flag_kernel<ES_SLAB_KINK>, grid, block, 0, ctx->stream, sp, d_K, d_W, nW, cells, step, ctx->d_masks, ctx->d_block_counts); break;
    case ES_SLAB_SAUSAGE_BODY: hipLaunchKernelGGL(slab_scan_flag_kernel<ES_SLAB_SAUSAGE_BODY>, grid, block, 0, ctx->stream, sp, d_K, d_W, nW, cells, step, ctx->d_masks, ctx->d_block_counts); break;
    default: hipLaunchKernelGGL(slab_scan_flag_kernel<ES_SLAB_KINK_BODY>, grid, block, 0, ctx->stream, sp, d_K, d_W, nW, cells, step, ctx->d_masks, ctx->d_block_counts); break;
  }
  ES_HIP_CHECK(ctx, hipGetLastError());
  int total = 0;
  rc = es_scan_block_counts(ctx, nblocks, &total);
  if (rc != ES_SUCCESS) return rc;
  *h_count = total;
  if (total > 0 && capacity > 0) {
    hipLaunchKernelGGL(slab_scan_emit_kernel, grid, block, 0, ctx->stream, d_K, d_W, nW, cells, step,
                       ctx->d_masks, ctx->d_block_counts, d_rootK, d_rootW, capacity);
    ES_HIP_CHECK(ctx, hipGetLastError());
  }
  return total > capacity ? ES_ERR_CAPACITY : ES_SUCCESS;
}

extern "C" int es_slab_analytic_filter(es_context* ctx, const es_slab_analytic_params* p, int mode,
                                       const double* d_rootK, const double* d_rootW, int n, double thresh,
                                       uint8_t* d_keep) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  ES_REQUIRE(ctx, p && (n == 0 || (d_rootK && d_rootW && d_keep)), "null pointer");
  ES_REQUIRE(ctx, n >= 0, "negative size");
  ES_REQUIRE(ctx, mode >= 0 && mode <= 3, "mode");
  if (n == 0) return ES_SUCCESS;
  ES_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(slab_filter_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, to_dev(p), mode,
                     d_rootK, d_rootW, n, thresh, d_keep);
  ES_HIP_CHECK(ctx, hipGetLastError());
  return ES_SUCCESS;
}
