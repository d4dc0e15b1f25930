// Micro-benchmark (round 4): does the compiler interleave TWO independent trajectories carried by one lane?
// The fp64 throughput kernel keeps two waves per SIMD (256 registers each) and reaches 11.2 k cycles per RK4 step and SIMD where a
// perfect interleave of the two instruction streams would need 7.9 k (profiles/r4_lone_wave_latency.txt).  Alternative: ONE wave
// per SIMD with 512 registers whose every lane carries two samples, written as a 2-vector type - the two chains are then
// adjacent in program order.  This times a dependent log2 -> exp2 chain (the shape of the atmosphere evaluation) both ways at the
// same number of elements per SIMD.     hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast pair_ilp.hip -o pair_ilp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct d2 { double a, b; };
__device__ __forceinline__ d2 operator+(d2 x, d2 y) { return {x.a + y.a, x.b + y.b}; }
__device__ __forceinline__ d2 operator-(d2 x, d2 y) { return {x.a - y.a, x.b - y.b}; }
__device__ __forceinline__ d2 operator*(d2 x, d2 y) { return {x.a * y.a, x.b * y.b}; }
__device__ __forceinline__ d2 operator+(d2 x, double y) { return {x.a + y, x.b + y}; }
__device__ __forceinline__ d2 operator-(d2 x, double y) { return {x.a - y, x.b - y}; }
__device__ __forceinline__ d2 operator*(d2 x, double y) { return {x.a * y, x.b * y}; }
__device__ __forceinline__ double vfma(double x, double y, double z) { return __builtin_fma(x, y, z); }
__device__ __forceinline__ d2 vfma(d2 x, d2 y, d2 z) { return {__builtin_fma(x.a, y.a, z.a), __builtin_fma(x.b, y.b, z.b)}; }
__device__ __forceinline__ d2 vfma(d2 x, d2 y, double z) { return {__builtin_fma(x.a, y.a, z), __builtin_fma(x.b, y.b, z)}; }
__device__ __forceinline__ double vrcp0(double x) { return __builtin_amdgcn_rcp(x); }
__device__ __forceinline__ d2 vrcp0(d2 x) { return {__builtin_amdgcn_rcp(x.a), __builtin_amdgcn_rcp(x.b)}; }
__device__ __forceinline__ double neg(double x) { return -x; }
__device__ __forceinline__ d2 neg(d2 x) { return {-x.a, -x.b}; }
__device__ __forceinline__ double splat(double, double v) { return v; }
__device__ __forceinline__ d2 splat(d2, double v) { return {v, v}; }
__device__ __forceinline__ double sum(double x) { return x; }
__device__ __forceinline__ double sum(d2 x) { return x.a + x.b; }

template <typename T>
__device__ __forceinline__ T rcp(T x) {     // seed + one cubic round, as the kernel's m_rcp
  const T r = vrcp0(x);
  const T e = vfma(neg(x), r, 1.0);
  return vfma(r * e, e + 1.0, r);           // r (1 + e + e^2)
}

template <typename T>
__device__ __forceinline__ T chain(T m) {   // m in [0.71, 1.41]: log2 m by the atanh series, then 2^(0.3 log2 m) by Taylor: ~45 dependent operations
  const T s = (m - 1.0) * rcp(m + 1.0);
  const T z = s * s;
  T q = splat(z, 0.12545174268599682);
  q = vfma(q, z, 0.1373995277037108); q = vfma(q, z, 0.15186263588304877); q = vfma(q, z, 0.16973048304767038);
  q = vfma(q, z, 0.19236121412069308); q = vfma(q, z, 0.2219552470623382); q = vfma(q, z, 0.26231074652821787);
  q = vfma(q, z, 0.32059091242337740); q = vfma(q, z, 0.41218831597291380); q = vfma(q, z, 0.57706364236207930);
  q = vfma(q, z, 0.96177273726013220); q = vfma(q, z, 2.88539008177792680);
  const T f = (s * q) * 0.3;
  T p = splat(f, 1.3691488853904128e-12);
  p = vfma(p, f, 2.5678435993488206e-11); p = vfma(p, f, 4.4455382718708116e-10); p = vfma(p, f, 7.054911620801123e-09);
  p = vfma(p, f, 1.01780860092397e-07); p = vfma(p, f, 1.321548679014431e-06); p = vfma(p, f, 1.5252733804059841e-05);
  p = vfma(p, f, 0.0001540353039338161); p = vfma(p, f, 0.0013333558146428443); p = vfma(p, f, 0.009618129107628477);
  p = vfma(p, f, 0.05550410866482158); p = vfma(p, f, 0.24022650695910072); p = vfma(p, f, 0.6931471805599453);
  p = vfma(p, f, 1.0);
  return p * 0.9 + 0.1;                      // back into [0.7, 1.4]
}

template <typename T, int MINW>
__global__ __launch_bounds__(64, MINW) void k(double* out, int iters, double seed) {
  T x = splat(T(), 0.8 + 0.001 * threadIdx.x + seed);
  if (sizeof(T) == 16) ((double*)&x)[1] += 0.05;
  for (int i = 0; i < iters; ++i) x = chain(x);
  out[blockIdx.x * 64 + threadIdx.x] = sum(x);
}

template <typename T, int MINW>
void run(const char* name, double* d, int blocks, int elems_per_lane) {
  const int iters = 20000;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<T, MINW>), dim3(blocks), dim3(64), 0, 0, d, 200, 0.0);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<T, MINW>), dim3(blocks), dim3(64), 0, 0, d, iters, 0.0);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double chains_per_simd = (double)blocks / 1024.0 * elems_per_lane;       // 256 CUs x 4 SIMDs
  printf("%-64s %8.3f ms   %7.1f nominal cycles per chain evaluation and SIMD (%.0f evaluations side by side)\n", name, ms,
         ms * 1e-3 * 2.4e9 / (iters * chains_per_simd), chains_per_simd);
}

int main() {
  double* d; CHECK(hipMalloc(&d, 4096 * 64 * sizeof(double)));
  run<double, 1>("one wave per SIMD, one sample per lane", d, 1024, 1);
  run<double, 2>("two waves per SIMD, one sample per lane (the shipped layout)", d, 2048, 1);
  run<d2, 1>("one wave per SIMD, two samples per lane", d, 1024, 2);
  run<d2, 2>("two waves per SIMD, two samples per lane", d, 2048, 2);
  run<double, 2>("four waves per SIMD, one sample per lane", d, 4096, 1);
  return 0;
}
