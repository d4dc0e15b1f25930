// Micro-benchmark: does a wave64 VALU instruction cost less when the upper (or lower) 32 lanes of EXEC are off?
// If the second 32-lane pass is skipped, packing a wave's surviving trajectories into one half pays.
// Diagnostic tool: hipcc --offload-arch=gfx950 -O3 exec_half.hip -o exec_half
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define REP8(x) x x x x x x x x

template <int F64>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed, int lo, int hi) {
  const int lane = threadIdx.x & 63;
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
  float b = 1.0001f, c = 0.0001f;
  double db = 1.0001, dc = 0.0001;
  if (lane >= lo && lane < hi) {   // EXEC = lanes [lo, hi) for the whole loop
    for (int i = 0; i < iters; ++i) {
      if (!F64) {
        REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                          "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
      } else {
        REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                          "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                          : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(dc));)
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3);
}

template <int F64>
void run(const char* name, float* d, int cus, int lo, int hi) {
  const int iters = 20000;
  for (int w = 1; w <= 3; ++w) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<F64>, dim3(cus * w), dim3(256), 0, 0, d, 1000, 1.0f, lo, hi);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<F64>, dim3(cus * w), dim3(256), 0, 0, d, iters, 1.0f, lo, hi);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double inst_per_wave = (double)iters * 64;
    printf("%-10s lanes [%2d,%2d)  waves/SIMD=%d  %.3f ms  %.2f cyc/instr/wave  %.2f cyc/instr/SIMD\n", name, lo, hi, w, ms,
           ms * 1e-3 * 2.4e9 / inst_per_wave, ms * 1e-3 * 2.4e9 / (inst_per_wave * w));
  }
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  printf("device %s, %d CUs\n", p.name, cus);
  float* d; CHECK(hipMalloc(&d, (size_t)cus * 4 * 256 * sizeof(float)));
  const int ranges[][2] = {{0, 64}, {0, 48}, {0, 32}, {32, 64}, {0, 16}, {0, 1}, {16, 48}};
  for (auto& r : ranges) run<0>("v_fma_f32", d, cus, r[0], r[1]);
  for (auto& r : ranges) run<1>("v_fma_f64", d, cus, r[0], r[1]);
  return 0;
}
