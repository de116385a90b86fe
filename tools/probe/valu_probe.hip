// Issue-rate probe for the fp64 VALU instructions of the shooting march (gfx950).  For each instruction class one
// kernel runs NCHAIN independent dependency chains per lane so that the pipeline latency is covered, with W waves
// per SIMD; the cycles per instruction per wave follow from s_memtime / clock64 deltas.
//   hipcc -O3 --offload-arch=gfx950 -o tools/probe/valu_probe tools/probe/valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITER = 4096;
constexpr int NCH = 8;

enum { OP_FMA = 0, OP_MUL, OP_ADD, OP_RCP, OP_CMP, OP_MIX, OP_FMA_DEP, OP_I32OR, OP_COUNT };
static const char* names[] = {"v_fma_f64 (8 chains)", "v_mul_f64 (8 chains)", "v_add_f64 (8 chains)", "v_rcp_f64 (8 chains)",
                              "v_cmp_lt_f64 + s_or_b64", "fma,mul,add,fma mix", "v_fma_f64 (1 chain, dependent)", "v_or_b32 (8 chains)"};

template <int OP>
__global__ void probe(double* out, long long* cycles, double seed) {
  double a[NCH];
  int ia[NCH];
  unsigned long long mask = 0;
#pragma unroll
  for (int c = 0; c < NCH; ++c) { a[c] = seed + 1e-3 * (threadIdx.x + c); ia[c] = threadIdx.x + c; }
  const double m = 1.0 + 1e-9 * seed, b = 1e-12 * seed;
  const long long t0 = clock64();
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (OP == OP_FMA) a[c] = __builtin_fma(a[c], m, b);
      else if (OP == OP_MUL) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[c]) : "v"(m));
      else if (OP == OP_ADD) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[c]) : "v"(b));
      else if (OP == OP_RCP) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[c]));
      else if (OP == OP_CMP) { mask |= __ballot(a[c] < (double)it); }
      else if (OP == OP_MIX) {
        if ((c & 3) == 0) a[c] = __builtin_fma(a[c], m, b);
        else if ((c & 3) == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[c]) : "v"(m));
        else if ((c & 3) == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[c]) : "v"(b));
        else a[c] = __builtin_fma(a[c], m, b);
      }
      else if (OP == OP_FMA_DEP) a[0] = __builtin_fma(a[0], m, b);
      else if (OP == OP_I32OR) asm volatile("v_or_b32 %0, %0, %1" : "+v"(ia[c]) : "v"(it));
    }
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int c = 0; c < NCH; ++c) s += a[c] + ia[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (double)(mask & 1);
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(int waves_per_simd, double* d_out, long long* d_cyc, int ncu) {
  const int threads = 64 * 4 * waves_per_simd;       // one workgroup per CU, waves spread over the 4 SIMDs
  const int blocks = ncu;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, 1.0);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, 1.0);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> cyc(blocks);
  CHECK(hipMemcpy(cyc.data(), d_cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost));
  double avg = 0; for (auto c : cyc) avg += (double)c; avg /= blocks;
  const double inst_per_wave = (double)ITER * NCH;
  // clock64() counts at a fixed 100 MHz reference on gfx9; convert with the wall time instead
  const double wall_cyc_at = ms * 1e-3;                      // seconds
  printf("%-34s waves/SIMD %d : %8.3f ms  -> %.2f ns per instr per wave-slot (x%d waves sharing a SIMD: %.2f ns per instr)  [clock64 delta %.0f]\n",
         names[OP], waves_per_simd, ms, wall_cyc_at * 1e9 / inst_per_wave, waves_per_simd,
         wall_cyc_at * 1e9 / inst_per_wave / waves_per_simd, avg);
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("device %s, %d CUs, clockRate %d kHz\n", prop.name, ncu, prop.clockRate);
  double* d_out; long long* d_cyc;
  CHECK(hipMalloc(&d_out, sizeof(double) * ncu * 1024));
  CHECK(hipMalloc(&d_cyc, sizeof(long long) * ncu));
  for (int w : {1, 2, 4}) {
    run<OP_FMA>(w, d_out, d_cyc, ncu);
    run<OP_MUL>(w, d_out, d_cyc, ncu);
    run<OP_ADD>(w, d_out, d_cyc, ncu);
    run<OP_RCP>(w, d_out, d_cyc, ncu);
    run<OP_CMP>(w, d_out, d_cyc, ncu);
    run<OP_MIX>(w, d_out, d_cyc, ncu);
    run<OP_FMA_DEP>(w, d_out, d_cyc, ncu);
    run<OP_I32OR>(w, d_out, d_cyc, ncu);
  }
  return 0;
}
