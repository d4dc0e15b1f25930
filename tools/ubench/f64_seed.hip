// Accuracy of v_rcp_f64 / v_rsq_f64 seeds and of the refinement schemes of the fp64 throughput build (MI355X).
// Diagnostic tool: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off f64_seed.hip -o f64_seed
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(const double* x, double* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double a = x[i];
  double r0 = __builtin_amdgcn_rcp(a);
  double e = __builtin_fma(-a, r0, 1.0);
  double r2 = __builtin_fma(e, r0, r0);                  // two Newton rounds (shipped in round 2)
  e = __builtin_fma(-a, r2, 1.0);
  r2 = __builtin_fma(e, r2, r2);
  double e1 = __builtin_fma(-a, r0, 1.0);                // one cubic round: r (1 + e + e^2)
  double r1 = __builtin_fma(r0, __builtin_fma(e1, e1, e1), r0);
  double y0 = __builtin_amdgcn_rsq(a);
  double f = __builtin_fma(-(a * y0), y0, 1.0);
  double y1 = __builtin_fma(y0 * f, __builtin_fma(0.375, f, 0.5), y0);   // one cubic round
  double g = __builtin_fma(-(a * y1), y1, 1.0);
  double y2 = __builtin_fma(y1 * g, __builtin_fma(0.375, g, 0.5), y1);   // two (shipped in round 2)
  o[0 * n + i] = r0; o[1 * n + i] = r1; o[2 * n + i] = r2; o[3 * n + i] = y0; o[4 * n + i] = y1; o[5 * n + i] = y2;
}
int main() {
  const int n = 1 << 20;
  double* hx = (double*)malloc(n * sizeof(double)), *ho = (double*)malloc(6 * n * sizeof(double));
  srand(1);
  for (int i = 0; i < n; ++i) { double u = rand() / (double)RAND_MAX; hx[i] = ldexp(1.0 + u, (rand() % 80) - 40); }
  double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * n * 8);
  hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(ho, dout, 6 * n * 8, hipMemcpyDeviceToHost);
  const char* names[6] = {"v_rcp_f64 seed", "rcp: one cubic round", "rcp: two Newton rounds", "v_rsq_f64 seed", "rsq: one cubic round", "rsq: two cubic rounds"};
  for (int c = 0; c < 6; ++c) {
    double worst = 0;
    for (int i = 0; i < n; ++i) {
      long double ref = c < 3 ? 1.0L / (long double)hx[i] : 1.0L / sqrtl((long double)hx[i]);
      double err = fabs((double)(((long double)ho[c * n + i] - ref) / ref));
      if (err > worst) worst = err;
    }
    printf("%-26s max relative error %.3e (%.2f ulp of 2^-53)\n", names[c], worst, worst / 1.1102230246251565e-16);
  }
  return 0;
}
