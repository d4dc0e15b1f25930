// Micro-benchmark: the shader clock the chip actually runs at under fp64 vector load (round 4: how much of the distance
// to the 78.65 TFLOP/s fp64 vector peak - quoted at the 2.4 GHz peak engine clock - is the clock itself?).
// s_memtime counts shader-clock cycles, s_memrealtime a constant 100 MHz: ratio x 100 MHz = clock.  Also prints the wall
// time (HIP events) and the fp64 FMA rate reached.   hipcc --offload-arch=gfx950 -O3 clock.hip -o clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define REP8(x) x x x x x x x x

__global__ __launch_bounds__(64) void k(double* out, unsigned long long* clk, int iters, double seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  double b = 1.0000001, c = 0.0001;
  unsigned long long t0, r0, t1, r1;
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                      "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
  }
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
  double* out; unsigned long long* clk;
  CHECK(hipMalloc(&out, 8192 * 64 * sizeof(double)));
  CHECK(hipMalloc(&clk, 2 * sizeof(unsigned long long)));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int waves_per_simd[] = {0, 1, 2, 4};     // 0 = ONE wave on the whole chip
  for (int w : waves_per_simd) {
    const int blocks = w ? 1024 * w : 1;
    const int iters = 400000 / (w ? w : 1);
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, clk, iters, 1.0);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    }
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CHECK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    const double clock_ghz = (double)h[0] / (double)h[1] * 0.1;
    const double fma = (double)blocks * 64.0 * iters * 64.0;          // fp64 FMAs executed
    printf("%s: %d workgroups of one wave, %8.2f ms, shader clock %.3f GHz (s_memtime / s_memrealtime), %.1f TFLOP/s fp64 (2 flops per FMA), "
           "%.2f cycles per v_fma_f64 and SIMD\n", w ? "waves per SIMD" : "one wave on the chip", blocks, ms, clock_ghz, 2.0 * fma / (ms * 1e-3) / 1e12,
           w ? (double)h[0] / ((double)iters * 64.0 * w) : (double)h[0] / ((double)iters * 64.0));
  }
  return 0;
}
