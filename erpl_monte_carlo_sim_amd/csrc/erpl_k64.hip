// fp64 correctness-gate instantiation (BASELINE config 2): reference operation order, IEEE
// division/sqrt, libm-grade transcendentals, FMA contraction off (see Makefile).
#define ERPL_REAL double
#define ERPL_FAITHFUL 1
#define ERPL_FAST_F32 0
#define ERPL_SUFFIX f64
#define ERPL_CAT_(a, b) a##b
#define ERPL_CAT(a, b) ERPL_CAT_(a, b)
#define ERPL_LAUNCH_NAME erpl_launch_f64
#include "erpl_kernels.inc"
