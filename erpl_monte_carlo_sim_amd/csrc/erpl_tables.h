// erpl_tables.h — device-side constant tables shared by the host API (erpl_api.cpp) and the
// kernel translation units.  Everything here is derived on the HOST in fp64 from erpl_config
// with the reference's own expressions (so the values are bit-identical to what the Python
// reference recomputes on every call), then staged to LDS / scalar registers by the kernels.
#pragma once
#include <stdint.h>

#include "../../include/erpl_mc.h"

#define ERPL_MAX_UNION_KNOTS (2 * ERPL_MAX_MACH_KNOTS)
#define ERPL_MACH_REC 8         // x0a cd0_y0 cd0_s cda_y0 cda_s x0b cp_y0 cp_s
#define ERPL_ATM_LAYERS 5
#define ERPL_ATM_REC 12          // aT bT Tlo Thi invTref eL href eH eM base (2 pad)
#define ERPL_RES_R 20
#define ERPL_RES_D 5
#define ERPL_RES_I 6
#define ERPL_MAX_PHASES 2048
#define ERPL_EXT_Q 4            // words per cursor array of the hand-over queue (phases 0..3 of the sweep launch)
#define ERPL_COAST_TABLE 2048   // rail-iteration counts covered by the NaN fast-forward table

// Scalar constants, uniform over the batch.  X-macro so the fp64 master copy can be converted to
// the kernel's working precision field by field.
#define ERPL_SCALARS(X)                                                                        \
  X(dq2)            /* (diameter/4)**2                      rocket.py:122 */                    \
  X(cg_dry)         /* center_of_mass_dry                   rocket.py:31  */                    \
  X(prop_cg)        /* center_of_mass_dry - 0.5             rocket.py:116 */                    \
  X(third)          /* propellant_length**2 / 12 = 4/12     rocket.py:123 */                    \
  X(Ixx_dry) X(Iyy_dry)                                                                          \
  X(ref_area) X(ref_diam) X(cp_location)                                                         \
  X(AR)             /* 2 s^2 / fin_area                     rocket.py:177 */                    \
  X(two_pi_AR)      /* 2*pi*AR                              rocket.py:180 */                    \
  X(cos_sweep) X(cos_sweep_c) /* cos(sweep), max(cos(sweep),1e-6)  rocket.py:179-180 */         \
  X(AR_over_cos)    /* AR / cos_sweep_c (fast path only) */                                      \
  X(stall_angle) X(max_angle) X(inv_stall_span) /* radians(15), radians(45), 1/(max-stall) */    \
  X(chute_area) X(chute_cd) X(chute_alt) X(power_off)                                            \
  X(P0) X(T0) X(lapse) X(Rg) X(g0) X(h_tropo) X(h_strat) X(T_strat)                             \
  X(tropo_exp)      /* g/(R*lapse)                          environment.py:33 */                \
  X(p11) X(p20) X(p25) /* layer base pressures              environment.py:38-75 */             \
  X(grad_exp)       /* g/(R*0.0028)                         environment.py:81 */                \
  X(dt_rail) X(dt_flight) X(half_dt) X(dt_sixth) X(max_time) X(rail_length)                     \
  X(pitch_damping) X(yaw_damping)                                                                \
  /* host-folded constants used only by the fast (non-faithful) RHS */                           \
  X(inv_T0) X(inv_Ts) X(inv_Rg)                                                                  \
  X(k_iso)          /* -g/(R*T_strat) * log2(e)  : isothermal layers as exp2 */                  \
  X(k_meso)         /* g*log2(e)/R               : mesosphere scale-height term */               \
  X(two_pi_AR_cos)  /* 2*pi*AR*cos(sweep) */                                                     \
  X(area_diam)      /* reference_area * reference_diameter */                                    \
  X(chute_k)        /* 0.5 * parachute_cd * parachute_area */                                    \
  X(AR_over_cos2)   /* (AR / cos_sweep_c)^2 */                                                   \
  X(q_of_PM2)       /* 0.5 * (1.4 * 287.053) / R : q_dynamic = q_of_PM2 * P * Mach^2 */

template <typename R>
struct ErplScalars {
#define X(name) R name;
  ERPL_SCALARS(X)
#undef X
};

// Master tables in fp64 (device global memory); kernels convert while staging into LDS.
struct ErplTables {
  ErplScalars<double> s64;
  ErplScalars<float> s32;
  double dt_rail, dt_flight, max_time;  // fp64 copies for the time accumulation
  int32_t motor_kind, n_curve, n_union, n_coast;
  double curve_t[ERPL_MAX_CURVE_KNOTS], curve_f[ERPL_MAX_CURVE_KNOTS];
  double union_knots[ERPL_MAX_UNION_KNOTS];
  double mach_rec[(ERPL_MAX_UNION_KNOTS + 1) * ERPL_MACH_REC];
  // Atmosphere layers of environment.py:26-103 as one formula (fast path, LDS-staged):
  //   T = clamp(bT + aT*h, Tlo, Thi);  P = base * 2^( eL*log2(T*invTref) + (h-href)*(eH + eM/T) )
  double atm_rec[ERPL_ATM_LAYERS * ERPL_ATM_REC];
  // NaN fast-forward: a trajectory whose position is all-NaN can no longer trip any event
  // (every comparison is false), so it runs `while t < max_time: t += dt` to the end.  Its final
  // time and step count depend only on the number of rail iterations (t is an accumulated sum),
  // so the host tabulates them once per config.
  double coast_t[ERPL_COAST_TABLE];
  int32_t coast_steps[ERPL_COAST_TABLE];
};

// Arguments of both kernels (passed by value).
struct ErplKArgs {
  int64_t n;
  int32_t k_wind, flags;
  const double* ic;
  const double* rocket;
  const double* motor;
  const double* alt_grid;
  const void* wind;
  double* summary;
  int32_t* status;
  // Resume queue (ping-pong): lane records written by the rail kernel (phase 0) and by every flight
  // launch for the lanes that reached their step-chunk limit; the next launch pops them densely.
  void* res_r[2];              // [ERPL_RES_R][res_cap] working precision: y[14], apogee, first_apogee,
                               //   max_speed2, max_coast, cx, cy
  double* res_d[2];            // [ERPL_RES_D][res_cap]: t, t_rail, apogee_t, first_apogee_t, latch_t
  int32_t* res_i[2];           // [ERPL_RES_I][res_cap]: id, steps, nrail, mode|flags, traj_len, ready (the phase
                               //   that may pop the record, once it is published to running adopters)
  int64_t res_cap;
  unsigned long long* qcnt;    // [ERPL_MAX_PHASES + 2] records available to phase p (phase 0: n)
  unsigned long long* qhead;   // [ERPL_MAX_PHASES + 2] pop cursor of phase p
  // Hand-over queue (fp64 throughput build -> reference-order kernel, ERPL_HANDOFF in erpl_kernels.inc): records of
  // the lanes that left the RK4 loop at an unphysical speed, same layout and capacity as one resume-queue buffer.
  // The sweep launch of the reference-order kernel pops them as ITS phase 1: res_*[1] = ext_*, qcnt = ext_q
  // (ext_cnt = &ext_q[1]), qhead = ext_q + ERPL_EXT_Q.  NULL / unused in the other builds.
  void* ext_r;
  double* ext_d;
  int32_t* ext_i;
  unsigned long long* ext_q;   // [2 * ERPL_EXT_Q] behind qhead in the same allocation (zeroed by the rail kernel)
  unsigned long long* ext_cnt;
  int32_t phase;               // index of this flight launch
  int32_t chunk_steps;         // RK4 steps a lane may take per launch (<= 0: unlimited, one launch)
  int32_t waves_per_simd;      // fp32 flight-kernel build to launch: 2 (256 VGPRs) or 3 (168 VGPRs, spills)
  int32_t adopt_spin;          // polls an adopting lane waits for a claimed record's ready word (< 0: none - test knob)
  int32_t adopt_lanes;         // > 0: a wave left with at most this many flying lanes once the queue is empty
                               //   parks them in the next phase's queue, where waves that still fly more pick
                               //   them up into their idle lanes (the launcher clears it for the last phase)
  // trajectory capture
  int64_t n_traj, traj_stride, traj_cap;
  const int64_t* traj_ids;
  double* traj;
  int64_t* traj_len;
  const ErplTables* tables;
  unsigned long long* counters;  // [0] queue head, [1] total steps, [2] wave iterations
  int32_t refill_threshold;
  int32_t n_union, n_curve, motor_kind, n_coast;   // uniform table sizes (also in *tables)
  double dt_rail, dt_flight, max_time;             // fp64 time constants
};

// launchers implemented in erpl_k64.hip / erpl_k32.hip
extern "C++" {
// ev: NULL or three hipEvent_t recorded before the rail kernel, between the kernels and after the
// flight kernel, on `stream`.
// scalars: host pointer to ErplScalars<double> / ErplScalars<float>, passed to the kernels BY VALUE
// (kernel-argument segment -> scalar registers; a pointer into global memory would be re-read
// through the vector memory path on every use because the kernels also store to global memory).
// n_phases flight launches follow the rail launch (1 when a.chunk_steps <= 0).
// tail_stream (or NULL): the launches behind the first go to that stream, after main_done (a hipEvent_t recorded on
// `stream` behind the first launch).
int erpl_launch_f64(const ErplKArgs& a, const void* scalars, int block, int max_blocks, int n_phases, void* stream, void** ev,
                    void* tail_stream, void* main_done);
int erpl_launch_f32(const ErplKArgs& a, const void* scalars, int block, int max_blocks, int n_phases, void* stream, void** ev,
                    void* tail_stream, void* main_done);
int erpl_launch_f64f(const ErplKArgs& a, const void* scalars, int block, int max_blocks, int n_phases, void* stream, void** ev,
                    void* tail_stream, void* main_done);
// The sweep of the fp64 throughput build's hand-over queue by the reference-order flight kernel (no rail launch): `a` is
// the batch's argument block with the hand-over queue mapped as phase 1 (see ErplKArgs::ext_r).  Compiled for <= 256
// registers (two waves per SIMD, spills to scratch: it makes a few steps per record) so that its workgroups start
// beside the throughput build's waves instead of waiting for an empty SIMD.
int erpl_launch_f64_sweep(const ErplKArgs& a, const void* scalars, int block, int max_blocks, void* stream);
// known-answer evaluation of one device function per lane (erpl_mc_debug_eval); in / out are [rows][m]
int erpl_launch_debug_f64(const ErplKArgs& a, const void* scalars, int what, int64_t m, const double* in, double* out, void* stream);
int erpl_launch_debug_f32(const ErplKArgs& a, const void* scalars, int what, int64_t m, const double* in, double* out, void* stream);
int erpl_launch_debug_f64f(const ErplKArgs& a, const void* scalars, int what, int64_t m, const double* in, double* out, void* stream);
// extraction of the per-step diagnostic histories (fp64 only): a.traj = records, a.traj_cap = m,
// a.n_traj = sample index, a.summary = out [m][ERPL_DIAG_DIM]
int erpl_launch_extract_f64(const ErplKArgs& a, const void* scalars, double time_offset, void* stream);
}
