// Context, error strings and the ordered-compaction scan shared by the scan / bracket kernels.
#include "es_common.hpp"

extern "C" int es_abi_version(void) { return ES_ABI_VERSION; }

extern "C" int es_abi_sizeof(int which) {
  switch (which) {
    case 0: return (int)sizeof(es_slab_analytic_params);
    case 1: return (int)sizeof(es_shoot_desc);
    case 2: return (int)sizeof(es_profiles);
    case 3: return (int)sizeof(es_root_table);
    case 4: return (int)sizeof(es_worker_spec);
    case 5: return (int)sizeof(es_cyl_uniform_params);
    case 6: return (int)sizeof(es_complex_root_table);
    default: return -1;
  }
}

extern "C" const char* es_status_string(int s) {
  switch (s) {
    case ES_SUCCESS: return "success";
    case ES_ERR_INVALID_ARG: return "invalid argument";
    case ES_ERR_HIP: return "HIP runtime error";
    case ES_ERR_CAPACITY: return "output capacity too small";
    case ES_ERR_NO_DEVICE: return "no HIP device";
    case ES_ERR_UNSUPPORTED: return "unsupported configuration";
    case ES_ERR_EVAL_CAP: return "worker task exceeded its evaluation cap";
    case ES_ERR_SCREENING: return "fp32-screened bracket not confirmed in fp64";
    default: return "unknown status";
  }
}

extern "C" int es_context_create(int device, void* stream, es_context** out) {
  if (!out) return ES_ERR_INVALID_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return ES_ERR_NO_DEVICE;
  if (device < 0 || device >= n) return ES_ERR_INVALID_ARG;
  es_context* ctx = new es_context();
  ctx->device = device;
  ctx->stream = (hipStream_t)stream;
  if (hipSetDevice(device) != hipSuccess) { delete ctx; return ES_ERR_HIP; }
  if (hipMalloc(&ctx->d_total, sizeof(int)) != hipSuccess) { delete ctx; return ES_ERR_HIP; }
  if (hipHostMalloc(&ctx->h_total, sizeof(int)) != hipSuccess) { (void)hipFree(ctx->d_total); delete ctx; return ES_ERR_HIP; }
  *out = ctx;
  return ES_SUCCESS;
}

extern "C" int es_context_destroy(es_context* ctx) {
  if (!ctx) return ES_SUCCESS;
  (void)hipSetDevice(ctx->device);
  if (ctx->d_masks) (void)hipFree(ctx->d_masks);
  if (ctx->d_block_counts) (void)hipFree(ctx->d_block_counts);
  if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
  if (ctx->d_cols) (void)hipFree(ctx->d_cols);
  if (ctx->d_coldead) (void)hipFree(ctx->d_coldead);
  if (ctx->d_total) (void)hipFree(ctx->d_total);
  if (ctx->h_total) (void)hipHostFree(ctx->h_total);
  for (auto& e : ctx->timer_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  delete ctx;
  return ES_SUCCESS;
}

void es_timer_begin(es_context* ctx) {
  if (!ctx->timer_on) return;
  hipEvent_t a = nullptr, b = nullptr;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
  (void)hipEventRecord(a, ctx->stream);
  ctx->timer_events.emplace_back(a, b);
}

void es_timer_end(es_context* ctx) {
  if (!ctx->timer_on || ctx->timer_events.empty()) return;
  (void)hipEventRecord(ctx->timer_events.back().second, ctx->stream);
}

extern "C" int es_context_grid_timer(es_context* ctx, int enable) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  ctx->timer_on = enable != 0;
  return ES_SUCCESS;
}

extern "C" int es_context_grid_time(es_context* ctx, double* h_total_ms, int* h_launches) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  ES_REQUIRE(ctx, h_total_ms && h_launches, "null pointer");
  ES_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ES_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  double total = 0.0;
  int n = 0;
  for (auto& e : ctx->timer_events) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) { total += (double)ms; ++n; }
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  ctx->timer_events.clear();
  *h_total_ms = total;
  *h_launches = n;
  return ES_SUCCESS;
}

extern "C" const char* es_last_error(const es_context* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

extern "C" int es_context_synchronize(es_context* ctx) {
  if (!ctx) return ES_ERR_INVALID_ARG;
  ES_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return ES_SUCCESS;
}

int es_ensure_scratch(es_context* ctx, size_t bytes) {
  if (bytes <= ctx->scratch_cap) return ES_SUCCESS;
  ES_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (ctx->d_scratch) {
    ES_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));        // earlier calls may still read the old buffer
    ES_HIP_CHECK(ctx, hipFree(ctx->d_scratch));
    ctx->d_scratch = nullptr;
    ctx->scratch_cap = 0;
  }
  const size_t cap = bytes + bytes / 4;                          // head room: batches of a sweep grow slowly
  ES_HIP_CHECK(ctx, hipMalloc(&ctx->d_scratch, cap));
  ctx->scratch_cap = cap;
  return ES_SUCCESS;
}

int es_ensure_scan_scratch(es_context* ctx, size_t cells) {
  const size_t nmask = (cells + 63) / 64 + 4;
  const size_t nblk = (cells + 255) / 256 + 1;
  ES_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (nmask > ctx->masks_cap) {
    if (ctx->d_masks) ES_HIP_CHECK(ctx, hipFree(ctx->d_masks));
    ES_HIP_CHECK(ctx, hipMalloc(&ctx->d_masks, nmask * sizeof(uint64_t)));
    ctx->masks_cap = nmask;
  }
  if (nblk > ctx->blocks_cap) {
    if (ctx->d_block_counts) ES_HIP_CHECK(ctx, hipFree(ctx->d_block_counts));
    ES_HIP_CHECK(ctx, hipMalloc(&ctx->d_block_counts, nblk * sizeof(int)));
    ctx->blocks_cap = nblk;
  }
  return ES_SUCCESS;
}

// Single-workgroup exclusive scan over the per-block counts (<= a few 1e5 entries; latency-bound, off the hot path).
__global__ __launch_bounds__(1024) void es_block_scan_kernel(int* __restrict__ counts, int n, int* __restrict__ total) {
  __shared__ int wave_sums[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    const int v = (i < n) ? counts[i] : 0;
    int x = v;                                   // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int y = __shfl_up(x, off);
      if (lane >= off) x += y;
    }
    if (lane == 63) wave_sums[wid] = x;
    __syncthreads();
    if (wid == 0) {
      int s = (lane < 16) ? wave_sums[lane] : 0;
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) {
        const int y = __shfl_up(s, off);
        if (lane >= off) s += y;
      }
      if (lane < 16) wave_sums[lane] = s;        // inclusive over waves
    }
    __syncthreads();
    const int wave_excl = (wid == 0) ? 0 : wave_sums[wid - 1];
    const int c = carry;
    if (i < n) counts[i] = c + wave_excl + x - v;   // exclusive offset
    __syncthreads();
    if (tid == 1023) carry = c + wave_sums[15];
    __syncthreads();
  }
  if (tid == 0) *total = carry;
}

int es_scan_block_counts_async(es_context* ctx, int nblocks) {
  hipLaunchKernelGGL(es_block_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_block_counts, nblocks, ctx->d_total);
  ES_HIP_CHECK(ctx, hipGetLastError());
  return ES_SUCCESS;
}

int es_scan_block_counts(es_context* ctx, int nblocks, int* h_total_out) {
  int rc = es_scan_block_counts_async(ctx, nblocks);
  if (rc) return rc;
  ES_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_total, ctx->d_total, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ES_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  *h_total_out = *ctx->h_total;
  return ES_SUCCESS;
}
