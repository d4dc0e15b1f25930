// Micro-benchmark: VALU issue cost on MI355X per instruction class, for 1..4 waves per SIMD.
// Diagnostic tool (not part of the product): hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = 1.0001f, c = 0.0001f;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {  // dependent v_fma_f32 chain
      REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));)
    } else if (KIND == 1) {  // 8 independent chains
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    } else if (KIND == 2) {  // 2 independent chains
      REP8(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                        "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3"
                        : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));)
    } else if (KIND == 3) {  // v_cmp + v_cndmask pairs (dependent through vcc), 32 pairs
      REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %2\n v_cndmask_b32 %0, %1, %0, vcc\n v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %1, %0, %1, vcc\n"
                        "v_cmp_gt_f32 vcc, %0, %2\n v_cndmask_b32 %0, %1, %0, vcc\n v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %1, %0, %1, vcc"
                        : "+v"(a0), "+v"(a1) : "v"(c) : "vcc");)
    } else if (KIND == 4) {  // dependent v_rcp_f32
      REP64(asm volatile("v_rcp_f32 %0, %0" : "+v"(a0));)
    } else if (KIND == 5) {  // independent v_mov
      REP8(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (KIND == 6) {  // packed fma, 4 independent pairs
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                        "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
      a0 = p0.x + p0.y; a1 = p1.x + p1.y; a2 = p2.x + p2.y; a3 = p3.x + p3.y; a4 = a5 = a6 = a7 = 0;
    } else if (KIND == 7) {  // dependent chain mixing fma with an SGPR-source select (v_cndmask e64)
      REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_fma_f32 %0, %0, %1, %2\n"
                        "v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_fma_f32 %0, %0, %1, %2\n v_mul_f32 %0, %0, %1"
                        : "+v"(a0) : "v"(b), "v"(c));)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
void run(const char* name, float* d, int cus) {
  const int iters = 20000;
  for (int w = 1; w <= 4; ++w) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(cus * w), dim3(256), 0, 0, d, 1000, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(cus * w), dim3(256), 0, 0, d, iters, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double inst_per_wave = (double)iters * 64;
    // cycles (at 2.4 GHz nominal) per instruction per SIMD = elapsed / (waves per SIMD * instr per wave)
    printf("%-34s waves/SIMD=%d  %.3f ms  %.2f cyc/instr/wave  %.2f cyc/instr/SIMD\n", name, w, ms,
           ms * 1e-3 * 2.4e9 / inst_per_wave, ms * 1e-3 * 2.4e9 / (inst_per_wave * w));
  }
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
  float* d; CHECK(hipMalloc(&d, (size_t)cus * 4 * 256 * sizeof(float)));
  run<0>("dependent v_fma_f32", d, cus);
  run<2>("2 independent v_fma_f32 chains", d, cus);
  run<1>("8 independent v_fma_f32 chains", d, cus);
  run<7>("dependent fma/mul/add mix", d, cus);
  run<3>("v_cmp + v_cndmask (dependent)", d, cus);
  run<4>("dependent v_rcp_f32", d, cus);
  run<5>("independent v_mov_b32", d, cus);
  run<6>("4 independent v_pk_fma_f32", d, cus);
  return 0;
}
