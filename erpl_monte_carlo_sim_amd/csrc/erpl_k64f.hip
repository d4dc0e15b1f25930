// fp64 throughput instantiation (ERPL_PREC_F64_FAST): the short formulation of the fp32 RHS
// (one reciprocal per denominator, interval-record atmosphere, no trig of atan2) carried in double
// with FMA contraction.  MI355X runs fp64 vector FMAs at half the fp32 rate, so this build keeps
// fp64 parity with the reference on the chaotic samples at a fraction of the cost
// of the reference-order gate kernel (erpl_k64.hip).  Round 3: two waves per SIMD (<= 256 registers
// per lane: lane state, wind interval and table records in LDS - ERPL_TWO_WAVE in erpl_kernels.inc).
// Round 4: lanes whose speed passes 1e6 m/s are handed to the reference-order kernel (ERPL_HANDOFF): which
// intermediate of a blow-up's last steps turns inf and which NaN decides how the reference's flight ends, and only
// the reference's own operation order reproduces that.
#define ERPL_REAL double
#define ERPL_FAITHFUL 0
#define ERPL_FAST_F32 0
#define ERPL_FAST_F64 1
#ifndef ERPL_TWO_WAVE
#define ERPL_TWO_WAVE 1
#endif
#ifndef ERPL_FLIGHT_MIN_WAVES
#define ERPL_FLIGHT_MIN_WAVES (ERPL_TWO_WAVE ? 2 : 1)
#endif
#ifndef ERPL_STAGE_UNROLL
#define ERPL_STAGE_UNROLL 4
#endif
#define ERPL_SUFFIX f64f
#define ERPL_CAT_(a, b) a##b
#define ERPL_CAT(a, b) ERPL_CAT_(a, b)
#define ERPL_LAUNCH_NAME erpl_launch_f64f
#include "erpl_kernels.inc"
