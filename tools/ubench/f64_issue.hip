// Micro-benchmark: fp64 VALU issue cost on MI355X for dependent / independent chains, scalar-operand forms and
// interleaved scalar moves, at 1..3 waves per SIMD.  Diagnostic tool (not part of the product):
//   hipcc --offload-arch=gfx950 -O3 f64_issue.hip -o f64_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(64) void k(double* out, int iters, double seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  double b = 1.0000001, c = 0.0001;
  double sb = 1.0000001, sc = 0.0001;
  asm volatile("" : "+s"(sb), "+s"(sc));
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {  // dependent chain
      REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));)
    } else if (KIND == 1) {  // 2 chains
      REP8(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                        "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3"
                        : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));)
    } else if (KIND == 2) {  // 4 chains
      REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                        "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
    } else if (KIND == 3) {  // dependent chain, scalar addend (Horner with the coefficient in an SGPR pair)
      REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "s"(sc));)
    } else if (KIND == 4) {  // dependent chain, scalar addend, two s_mov between (the shipped Horner form)
      REP64(asm volatile("s_mov_b32 s20, 0x3166d0f9\n s_mov_b32 s21, 0x3d781619\n v_fma_f64 %0, %0, %1, s[20:21]" : "+v"(a0) : "v"(b) : "s20", "s21");)
    } else if (KIND == 5) {  // 2 chains of v_mul_f64
      REP8(asm volatile("v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2\n v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2\n"
                        "v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2\n v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2"
                        : "+v"(a0), "+v"(a1) : "v"(b));)
    } else if (KIND == 6) {  // 2 chains of v_add_f64
      REP8(asm volatile("v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2\n v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2\n"
                        "v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2\n v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2"
                        : "+v"(a0), "+v"(a1) : "v"(c));)
    } else if (KIND == 7) {  // 64-bit select: v_cmp_gt_f64 + 2 v_cndmask_b32 (3 instructions per select)
      float f0 = (float)a2, f1 = (float)a3;
      REP8(asm volatile("v_cmp_gt_f64 vcc, %2, %3\n v_cndmask_b32 %0, %1, %0, vcc\n v_cndmask_b32 %1, %0, %1, vcc\n v_cmp_gt_f64 vcc, %3, %2\n"
                        "v_cndmask_b32 %0, %1, %0, vcc\n v_cndmask_b32 %1, %0, %1, vcc\n v_cmp_gt_f64 vcc, %2, %3\n v_cndmask_b32 %0, %1, %0, vcc"
                        : "+v"(f0), "+v"(f1) : "v"(a0), "v"(c) : "vcc");)
      a2 = f0; a3 = f1;
    } else if (KIND == 8) {  // dependent v_rcp_f64
      REP64(asm volatile("v_rcp_f64 %0, %0" : "+v"(a0));)
    } else if (KIND == 9) {  // v_max_f64, 2 chains
      REP8(asm volatile("v_max_f64 %0, %0, %2\n v_max_f64 %1, %1, %2\n v_max_f64 %0, %0, %2\n v_max_f64 %1, %1, %2\n"
                        "v_max_f64 %0, %0, %2\n v_max_f64 %1, %1, %2\n v_max_f64 %0, %0, %2\n v_max_f64 %1, %1, %2"
                        : "+v"(a0), "+v"(a1) : "v"(c));)
    } else if (KIND == 10) {  // v_mov_b64, 2 chains
      REP8(asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %0\n v_mov_b64 %0, %1\n v_mov_b64 %1, %0\n"
                        "v_mov_b64 %0, %1\n v_mov_b64 %1, %0\n v_mov_b64 %0, %1\n v_mov_b64 %1, %0"
                        : "+v"(a0), "+v"(a1));)
    } else if (KIND == 11) {  // fma dependent chain alternating with an independent ds_read_b64 (LDS beside VALU)
      REP8(asm volatile("v_fma_f64 %0, %0, %2, %3\n ds_read_b64 %1, %4\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n"
                        "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n s_waitcnt lgkmcnt(0)"
                        : "+v"(a0), "=v"(a1) : "v"(b), "v"(c), "v"(threadIdx.x * 8) : "memory");)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <int KIND>
void run(const char* name, double* d, int cus, int per_rep) {
  const int iters = 6000;
  for (int w = 1; w <= 3; ++w) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(cus * 4 * w), dim3(64), 0, 0, d, 500, 1.0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(cus * 4 * w), dim3(64), 0, 0, d, iters, 1.0);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double inst_per_wave = (double)iters * per_rep;
    printf("%-46s waves/SIMD=%d  %8.3f ms  %6.2f cyc/instr/wave  %6.2f cyc/instr/SIMD\n", name, w, ms,
           ms * 1e-3 * 2.4e9 / inst_per_wave, ms * 1e-3 * 2.4e9 / (inst_per_wave * w));
  }
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz (cycles below are nominal 2.4 GHz cycles per VALU instruction)\n", p.name, cus, p.clockRate);
  double* d; CHECK(hipMalloc(&d, (size_t)cus * 4 * 3 * 64 * sizeof(double)));
  run<0>("dependent v_fma_f64", d, cus, 64);
  run<1>("2 independent v_fma_f64 chains", d, cus, 64);
  run<2>("4 independent v_fma_f64 chains", d, cus, 64);
  run<3>("dependent v_fma_f64, SGPR addend", d, cus, 64);
  run<4>("dependent v_fma_f64, SGPR addend + 2 s_mov each", d, cus, 64);
  run<5>("2 chains v_mul_f64", d, cus, 64);
  run<6>("2 chains v_add_f64", d, cus, 64);
  run<7>("v_cmp_gt_f64 + v_cndmask_b32 pairs", d, cus, 64);
  run<8>("dependent v_rcp_f64", d, cus, 64);
  run<9>("2 chains v_max_f64", d, cus, 64);
  run<10>("v_mov_b64 ping-pong", d, cus, 64);
  run<11>("dependent fma x8 + 1 ds_read_b64 (per fma)", d, cus, 64);
  return 0;
}
