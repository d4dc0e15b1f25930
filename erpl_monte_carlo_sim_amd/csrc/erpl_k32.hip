// fp32 throughput instantiation (BASELINE configs 3-5): algebraic shortcuts + hardware
// transcendental instructions; time stays fp64.
#define ERPL_REAL float
#define ERPL_FAITHFUL 0
#ifndef ERPL_FAST_F32
#define ERPL_FAST_F32 1
#endif
// RK4 stages fully unrolled: -8 % time vs the rolled loop (no loop-carried register moves,
// cross-stage scheduling); the fp64 gate keeps the rolled loop (code size, compile time).
#ifndef ERPL_STAGE_UNROLL
#define ERPL_STAGE_UNROLL 4
#endif
// two resident waves per SIMD for the uncapped build: at most 256 registers (left to itself the allocator took a
// 257th with the wind prefetch in and halved the occupancy)
#define ERPL_FLIGHT_MIN_WAVES 2
#define ERPL_SUFFIX f32
#define ERPL_CAT_(a, b) a##b
#define ERPL_CAT(a, b) ERPL_CAT_(a, b)
#define ERPL_LAUNCH_NAME erpl_launch_f32
#include "erpl_kernels.inc"
