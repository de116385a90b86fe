// Internal declarations shared by the translation units of libeigensolver_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>
#include "../../include/eigensolver_amd.h"

struct es_context {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string last_error;
  // scratch for bracket compaction (grown on demand, never shrunk)
  uint64_t* d_masks = nullptr;     size_t masks_cap = 0;      // one ballot mask per 64 cells
  int* d_block_counts = nullptr;   size_t blocks_cap = 0;     // per-256-cell block counts / exclusive offsets
  int* d_total = nullptr;                                     // device-side total count
  int* h_total = nullptr;                                     // pinned host mirror
  // general call scratch (worker tables and frame stacks, refinement start data): grown on demand, never shrunk;
  // calls on one context are serialised by its stream, so one buffer suffices
  void* d_scratch = nullptr;       size_t scratch_cap = 0;
  // live-column list of es_shoot_eval_grid_ex(ES_EVAL_SKIP_CONTINUUM): [0] = count, [1..] = column indices; per-column
  // dead flags behind it
  int* d_cols = nullptr;           size_t cols_cap = 0;
  uint8_t* d_coldead = nullptr;
  // es_context_grid_timer: event pairs around the launches of the dominant (grid march) kernels on this stream
  bool timer_on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> timer_events;
};

// Bracket a launch of a grid-march kernel with HIP events on the context's stream (no-ops unless the timer is on).
void es_timer_begin(es_context* ctx);
void es_timer_end(es_context* ctx);

#define ES_HIP_CHECK(ctx, expr)                                                                 \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess) {                                                                     \
      (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);                    \
      return ES_ERR_HIP;                                                                        \
    }                                                                                           \
  } while (0)

#define ES_REQUIRE(ctx, cond, msg)                                                              \
  do {                                                                                          \
    if (!(cond)) {                                                                              \
      if (ctx) (ctx)->last_error = std::string("invalid argument: ") + (msg);                   \
      return ES_ERR_INVALID_ARG;                                                                \
    }                                                                                           \
  } while (0)

// Grow ctx->d_scratch to at least `bytes` (synchronises the stream before freeing the old buffer).
int es_ensure_scratch(es_context* ctx, size_t bytes);

// Grow the compaction scratch so that `cells` cells fit.
int es_ensure_scan_scratch(es_context* ctx, size_t cells);

// Ordered compaction of flagged cells (flags live as 64-bit wave ballots in ctx->d_masks):
//   step 1 (done by the caller's flag kernel): masks[c/64], block_counts[c/256]
//   step 2: exclusive scan of block_counts -> offsets, total
// Returns the total through ctx->h_total after a stream sync.
int es_scan_block_counts(es_context* ctx, int nblocks, int* h_total_out);
// The scan alone, enqueued: offsets in ctx->d_block_counts, total in ctx->d_total, nothing read back.
int es_scan_block_counts_async(es_context* ctx, int nblocks);

// Position of a flagged cell inside the ordered output, from the masks and the scanned block offsets.
__device__ __forceinline__ int es_cell_rank(const uint64_t* __restrict__ masks, const int* __restrict__ block_off,
                                            long cell) {
  const long blk = cell >> 8;
  const int wave_in_blk = (int)((cell >> 6) & 3);
  const int lane = (int)(cell & 63);
  int pos = block_off[blk];
  const uint64_t* m = masks + (blk << 2);
  for (int w = 0; w < wave_in_blk; ++w) pos += __popcll(m[w]);
  const uint64_t lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  pos += __popcll(m[wave_in_blk] & lower);
  return pos;
}
