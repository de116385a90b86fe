// Accuracy of the v_rcp_f64 seed and of the corrected reciprocals built on it (gfx950), against the IEEE quotient.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o tools/probe/rcp_probe tools/probe/rcp_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

__global__ void k(const double* x, double* seed, double* one, double* third, double* ieee, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  const double r0 = __builtin_amdgcn_rcp(v);
  const double e = __builtin_fma(-v, r0, 1.0);
  seed[i] = r0;
  one[i] = __builtin_fma(r0, e, r0);                       // one Newton step
  third[i] = __builtin_fma(r0, __builtin_fma(e, e, e), r0); // r0 (1 + e + e^2)
  ieee[i] = 1.0 / v;
}

int main() {
  const int n = 1 << 22;
  std::vector<double> x(n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    const double m = 1.0 + (double)rand() / RAND_MAX + (double)rand() / RAND_MAX / RAND_MAX;   // mantissa in [1, 2)
    x[i] = ldexp(m, (rand() % 200) - 100) * ((rand() & 1) ? 1.0 : -1.0);
  }
  double *dx, *d[4];
  hipMalloc(&dx, n * 8); hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  for (auto& p : d) hipMalloc(&p, n * 8);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d[0], d[1], d[2], d[3], n);
  std::vector<double> h[4];
  for (int j = 0; j < 4; ++j) { h[j].resize(n); hipMemcpy(h[j].data(), d[j], n * 8, hipMemcpyDeviceToHost); }
  const char* names[3] = {"v_rcp_f64 seed", "seed + one Newton step", "seed (1 + e + e^2)"};
  for (int j = 0; j < 3; ++j) {
    double worst = 0; long exact = 0;
    for (int i = 0; i < n; ++i) {
      const double ref = (double)(1.0L / (long double)x[i]);
      const double err = fabs((h[j][i] - ref) / ref);
      if (err > worst) worst = err;
      if (h[j][i] == h[3][i]) ++exact;
    }
    printf("%-24s max relative error %.3e = 2^%.1f ; identical to the IEEE quotient in %.4f %% of %d arguments\n", names[j], worst,
           log2(worst > 0 ? worst : 1e-300), 100.0 * exact / n, n);
  }
  return 0;
}
