// Micro-benchmark: what an fp64 compare and the select behind it cost on MI355X, alone and with a second / third wave on
// the SIMD (round 4: profiles/r3_ubench_f64_issue.txt showed cmp + cndmask pairs at 15.9 cycles per instruction that a second
// wave does not hide).  Diagnostic tool:   hipcc --offload-arch=gfx950 -O3 cmp_issue.hip -o cmp_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define REP8(x) x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(64) void k(double* out, int iters, double seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  double b = 1.0000001, c = 0.0001;
  float f0 = (float)a2, f1 = (float)a3;
  int i0 = threadIdx.x, i1 = 7;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {         // v_cmp_gt_f64 alone, results to four SGPR pairs in turn (nobody reads them)
      REP8(asm volatile("v_cmp_gt_f64 s[20:21], %0, %1\n v_cmp_gt_f64 s[22:23], %1, %0\n v_cmp_gt_f64 s[24:25], %0, %2\n v_cmp_gt_f64 s[26:27], %2, %0\n"
                        "v_cmp_gt_f64 s[20:21], %0, %1\n v_cmp_gt_f64 s[22:23], %1, %0\n v_cmp_gt_f64 s[24:25], %0, %2\n v_cmp_gt_f64 s[26:27], %2, %0"
                        :: "v"(a0), "v"(a1), "v"(c) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
    } else if (KIND == 1) {  // v_cmp_gt_f32 alone
      REP8(asm volatile("v_cmp_gt_f32 s[20:21], %0, %1\n v_cmp_gt_f32 s[22:23], %1, %0\n v_cmp_gt_f32 s[24:25], %0, %1\n v_cmp_gt_f32 s[26:27], %1, %0\n"
                        "v_cmp_gt_f32 s[20:21], %0, %1\n v_cmp_gt_f32 s[22:23], %1, %0\n v_cmp_gt_f32 s[24:25], %0, %1\n v_cmp_gt_f32 s[26:27], %1, %0"
                        :: "v"(f0), "v"(f1) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
    } else if (KIND == 2) {  // v_cmp_gt_i32 alone
      REP8(asm volatile("v_cmp_gt_i32 s[20:21], %0, %1\n v_cmp_gt_i32 s[22:23], %1, %0\n v_cmp_gt_i32 s[24:25], %0, %1\n v_cmp_gt_i32 s[26:27], %1, %0\n"
                        "v_cmp_gt_i32 s[20:21], %0, %1\n v_cmp_gt_i32 s[22:23], %1, %0\n v_cmp_gt_i32 s[24:25], %0, %1\n v_cmp_gt_i32 s[26:27], %1, %0"
                        :: "v"(i0), "v"(i1) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
    } else if (KIND == 3) {  // v_cndmask_b32 alone, mask in a fixed SGPR pair, two chains
      REP8(asm volatile("v_cndmask_b32 %0, %1, %0, s[20:21]\n v_cndmask_b32 %1, %0, %1, s[20:21]\n v_cndmask_b32 %0, %1, %0, s[20:21]\n v_cndmask_b32 %1, %0, %1, s[20:21]\n"
                        "v_cndmask_b32 %0, %1, %0, s[20:21]\n v_cndmask_b32 %1, %0, %1, s[20:21]\n v_cndmask_b32 %0, %1, %0, s[20:21]\n v_cndmask_b32 %1, %0, %1, s[20:21]"
                        : "+v"(f0), "+v"(f1) :: "s20", "s21");)
    } else if (KIND == 4) {  // cmp -> cndmask -> cndmask through VCC, the select of a double: 1 + 2, back to back
      REP8(asm volatile("v_cmp_gt_f64 vcc, %2, %3\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_cmp_gt_f64 vcc, %3, %2\n"
                        "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_nop\n v_nop"
                        : "+v"(f0), "+v"(f1) : "v"(a0), "v"(c) : "vcc");)
    } else if (KIND == 5) {  // the same select with four independent v_fma_f64 between the compare and its selects
      REP8(asm volatile("v_cmp_gt_f64 vcc, %6, %7\n v_fma_f64 %2, %2, %7, %7\n v_fma_f64 %3, %3, %7, %7\n v_fma_f64 %4, %4, %7, %7\n v_fma_f64 %5, %5, %7, %7\n"
                        "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_nop"
                        : "+v"(f0), "+v"(f1), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b) : "v"(a0), "v"(c) : "vcc");)
    } else if (KIND == 6) {  // v_cmp_class_f64 alone
      REP8(asm volatile("v_cmp_class_f64 s[20:21], %0, %1\n v_cmp_class_f64 s[22:23], %0, %1\n v_cmp_class_f64 s[24:25], %0, %1\n v_cmp_class_f64 s[26:27], %0, %1\n"
                        "v_cmp_class_f64 s[20:21], %0, %1\n v_cmp_class_f64 s[22:23], %0, %1\n v_cmp_class_f64 s[24:25], %0, %1\n v_cmp_class_f64 s[26:27], %0, %1"
                        :: "v"(a0), "v"(i1) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
    } else if (KIND == 7) {  // four v_fma_f64 chains with one v_cmp_gt_f64 (unread) per four: does the compare cost more than its slot?
      REP8(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_cmp_gt_f64 s[20:21], %0, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n"
                        "v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_cmp_gt_f64 s[22:23], %1, %4"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "s20", "s21", "s22", "s23");)
    } else if (KIND == 8) {  // the select of a double through an SGPR pair (e64 forms), 1 + 2
      REP8(asm volatile("v_cmp_gt_f64 s[20:21], %2, %3\n v_cndmask_b32 %0, %0, %1, s[20:21]\n v_cndmask_b32 %1, %1, %0, s[20:21]\n v_cmp_gt_f64 s[22:23], %3, %2\n"
                        "v_cndmask_b32 %0, %0, %1, s[22:23]\n v_cndmask_b32 %1, %1, %0, s[22:23]\n v_nop\n v_nop"
                        : "+v"(f0), "+v"(f1) : "v"(a0), "v"(c) : "s20", "s21", "s22", "s23");)
    } else if (KIND == 9) {  // v_nop only (the issue floor of the harness)
      REP8(asm volatile("v_nop\n v_nop\n v_nop\n v_nop\n v_nop\n v_nop\n v_nop\n v_nop");)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + f0 + f1 + i0;
}

template <int KIND>
void run(const char* name, double* d, int cus) {
  const int iters = 6000, per_rep = 64;
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
    printf("%-58s waves/SIMD=%d  %8.3f ms  %6.2f cyc/instr/wave  %6.2f cyc/instr/SIMD\n", name, w, ms,
           ms * 1e-3 * 2.4e9 / inst_per_wave, ms * 1e-3 * 2.4e9 / (inst_per_wave * w));
  }
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  printf("device %s, %d CUs (nominal 2.4 GHz cycles per instruction; 64 instructions per repetition, v_nop included where present)\n", p.name, cus);
  double* d; CHECK(hipMalloc(&d, (size_t)cus * 4 * 3 * 64 * sizeof(double)));
  run<9>("v_nop", d, cus);
  run<0>("v_cmp_gt_f64 -> SGPR pair, unread", d, cus);
  run<1>("v_cmp_gt_f32 -> SGPR pair, unread", d, cus);
  run<2>("v_cmp_gt_i32 -> SGPR pair, unread", d, cus);
  run<6>("v_cmp_class_f64 -> SGPR pair, unread", d, cus);
  run<3>("v_cndmask_b32, mask in a fixed SGPR pair, 2 chains", d, cus);
  run<4>("cmp_f64 vcc + 2 cndmask, twice, + 2 v_nop", d, cus);
  run<8>("cmp_f64 s[..] + 2 cndmask, twice, + 2 v_nop", d, cus);
  run<5>("cmp_f64 vcc, 4 fma_f64, 2 cndmask, v_nop", d, cus);
  run<7>("6 fma_f64 (4 chains) + 2 unread cmp_f64", d, cus);
  return 0;
}
