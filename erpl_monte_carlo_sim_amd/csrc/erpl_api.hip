// erpl_api.hip — host side of the C ABI declared in include/erpl_mc.h.
// Derives the device tables from erpl_config with the reference's own expressions (glibc libm,
// the same pow/exp CPython uses), owns the per-GPU workspace and enqueues the two kernels.
// There is deliberately no CPU execution path in this library.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <thread>
#include <vector>

#include "erpl_tables.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                     \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) return fail(ERPL_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

bool finite_all(const double* v, int n) {
  for (int i = 0; i < n; ++i) if (!std::isfinite(v[i])) return false;
  return true;
}
bool increasing(const double* v, int n) {
  for (int i = 1; i < n; ++i) if (!(v[i] > v[i - 1])) return false;
  return true;
}

// np.interp interval record of table (xp, fp, n) for abscissae in [u, next union knot)
void interval_record(const double* xp, const double* fp, int n, bool below_all, double u,
                     double& x0, double& y0, double& s) {
  if (below_all || u < xp[0]) { x0 = xp[0]; y0 = fp[0]; s = 0.0; return; }
  if (u >= xp[n - 1]) { x0 = xp[n - 1]; y0 = fp[n - 1]; s = 0.0; return; }
  int j = 0;
  while (j + 1 < n && xp[j + 1] <= u) ++j;
  x0 = xp[j]; y0 = fp[j];
  s = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);  // numpy arr_interp slope
}

template <typename R>
void convert_scalars(const ErplScalars<double>& a, ErplScalars<R>& b) {
#define X(name) b.name = (R)a.name;
  ERPL_SCALARS(X)
#undef X
}

int build_tables(const erpl_config& c, ErplTables& T) {
  memset(&T, 0, sizeof(T));
  if (c.n_cd < 1 || c.n_cd > ERPL_MAX_MACH_KNOTS || c.n_cp < 1 || c.n_cp > ERPL_MAX_MACH_KNOTS)
    return fail(ERPL_ERR_INVALID, "Mach table sizes out of range (n_cd=%d n_cp=%d)", c.n_cd, c.n_cp);
  if (!finite_all(c.cd_mach, c.n_cd) || !finite_all(c.cd0, c.n_cd) || !finite_all(c.cda, c.n_cd) ||
      !finite_all(c.cp_mach, c.n_cp) || !finite_all(c.cp_shift, c.n_cp))
    return fail(ERPL_ERR_INVALID, "non-finite aerodynamic table entry");
  if (!increasing(c.cd_mach, c.n_cd) || !increasing(c.cp_mach, c.n_cp))
    return fail(ERPL_ERR_INVALID, "Mach knots must be strictly increasing");
  if (c.motor_kind != ERPL_MOTOR_LIQUID && c.motor_kind != ERPL_MOTOR_SOLID)
    return fail(ERPL_ERR_INVALID, "unknown motor_kind %d", c.motor_kind);
  if (c.motor_kind == ERPL_MOTOR_SOLID) {
    if (c.n_curve < 1 || c.n_curve > ERPL_MAX_CURVE_KNOTS)
      return fail(ERPL_ERR_INVALID, "thrust curve size out of range (%d)", c.n_curve);
    if (!finite_all(c.curve_time, c.n_curve) || !finite_all(c.curve_thrust, c.n_curve) ||
        !increasing(c.curve_time, c.n_curve))
      return fail(ERPL_ERR_INVALID, "thrust curve must be finite with increasing time");
  }
  if (!(c.dt_initial > 0) || !std::isfinite(c.dt_initial) || !std::isfinite(c.max_time))
    return fail(ERPL_ERR_INVALID, "dt_initial must be positive and finite");
  // the kernels count steps in int32 and advance time by `t += dt`: a horizon of more than 2^30 steps
  // overflows the counter, and long before that dt drops below ulp(t) and the loop stops advancing
  // (the reference would spin for ever there too) - refuse it instead of hanging the GPU
  if (c.max_time > 0 && c.max_time / ((0.005 < c.dt_initial) ? 0.005 : c.dt_initial) > 1073741824.0)
    return fail(ERPL_ERR_INVALID, "max_time / dt exceeds 2^30 steps");

  ErplScalars<double>& s = T.s64;
  s.dq2 = pow(c.diameter / 4, 2.0);                    // rocket.py:122
  s.cg_dry = c.center_of_mass_dry;
  s.prop_cg = c.center_of_mass_dry - 0.5;              // rocket.py:116
  s.third = 4.0 / 12;                                  // rocket.py:121-123
  s.Ixx_dry = c.Ixx_dry; s.Iyy_dry = c.Iyy_dry;
  s.ref_area = c.reference_area; s.ref_diam = c.reference_diameter; s.cp_location = c.cp_location;
  const double fin_area = 0.5 * (c.fin_root_chord + c.fin_tip_chord) * c.fin_span;  // rocket.py:176
  s.AR = (fin_area > 0) ? 2 * pow(c.fin_span, 2.0) / fin_area : 0.0;                // rocket.py:177
  s.two_pi_AR = 2 * M_PI * s.AR;                                                   // rocket.py:180
  s.cos_sweep = cos(c.fin_sweep_angle);
  s.cos_sweep_c = (1e-6 > s.cos_sweep) ? 1e-6 : s.cos_sweep;                        // rocket.py:179
  s.AR_over_cos = s.AR / s.cos_sweep_c;
  s.stall_angle = 15.0 * (M_PI / 180.0);                                           // rocket.py:167-168
  s.max_angle = 45.0 * (M_PI / 180.0);
  s.inv_stall_span = 1.0 / (s.max_angle - s.stall_angle);
  s.chute_area = c.parachute_area; s.chute_cd = c.parachute_cd;
  s.chute_alt = c.parachute_deployment_altitude; s.power_off = c.power_off_drag_factor;
  s.P0 = c.sea_level_pressure; s.T0 = c.sea_level_temperature; s.lapse = c.temperature_lapse_rate;
  s.Rg = c.gas_constant; s.g0 = c.gravity; s.h_tropo = c.troposphere_height;
  s.h_strat = c.stratosphere_height; s.T_strat = c.stratosphere_temp;
  s.tropo_exp = c.gravity / (c.gas_constant * c.temperature_lapse_rate);           // environment.py:33
  s.p11 = c.sea_level_pressure * pow(c.stratosphere_temp / c.sea_level_temperature, s.tropo_exp);
  s.p20 = s.p11 * exp(-c.gravity * (c.stratosphere_height - c.troposphere_height) /
                      (c.gas_constant * c.stratosphere_temp));                     // environment.py:56-62
  s.p25 = s.p20 * exp(-c.gravity * 5000.0 / (c.gas_constant * c.stratosphere_temp)); // :72-75
  s.grad_exp = c.gravity / (c.gas_constant * 0.0028);                              // :81
  s.dt_rail = c.dt_initial;
  s.dt_flight = (0.005 < c.dt_initial) ? 0.005 : c.dt_initial;                     // simulator.py:209
  s.half_dt = 0.5 * s.dt_flight;
  s.dt_sixth = s.dt_flight / 6.0;
  s.max_time = c.max_time; s.rail_length = c.rail_length;
  s.pitch_damping = c.pitch_damping; s.yaw_damping = c.yaw_damping;
  s.inv_T0 = 1.0 / s.T0; s.inv_Ts = 1.0 / s.T_strat; s.inv_Rg = 1.0 / s.Rg;
  s.k_iso = -s.g0 / (s.Rg * s.T_strat) * M_LOG2E;
  s.k_meso = s.g0 * M_LOG2E / s.Rg;
  s.two_pi_AR_cos = s.two_pi_AR * s.cos_sweep;
  s.area_diam = s.ref_area * s.ref_diam;
  s.AR_over_cos2 = s.AR_over_cos * s.AR_over_cos;
  s.q_of_PM2 = 0.5 * (1.4 * 287.053) / s.Rg;
  s.chute_k = 0.5 * s.chute_cd * s.chute_area;
  convert_scalars(T.s64, T.s32);
  T.dt_rail = s.dt_rail; T.dt_flight = s.dt_flight; T.max_time = s.max_time;
  T.motor_kind = c.motor_kind;
  T.n_curve = (c.motor_kind == ERPL_MOTOR_SOLID) ? c.n_curve : 0;
  for (int i = 0; i < T.n_curve; ++i) { T.curve_t[i] = c.curve_time[i]; T.curve_f[i] = c.curve_thrust[i]; }

  // union of the Cd and CP-shift Mach knots and the per-interval np.interp records
  std::vector<double> u(c.cd_mach, c.cd_mach + c.n_cd);
  u.insert(u.end(), c.cp_mach, c.cp_mach + c.n_cp);
  std::sort(u.begin(), u.end());
  u.erase(std::unique(u.begin(), u.end()), u.end());
  T.n_union = (int)u.size();
  for (int i = 0; i < T.n_union; ++i) T.union_knots[i] = u[i];
  for (int i = 0; i <= T.n_union; ++i) {
    double* r = &T.mach_rec[i * ERPL_MACH_REC];
    const bool below = (i == 0);
    const double left = below ? u[0] : u[i - 1];
    double x0, y0, sl;
    interval_record(c.cd_mach, c.cd0, c.n_cd, below, left, x0, y0, sl);
    r[0] = x0; r[1] = y0; r[2] = sl;
    interval_record(c.cd_mach, c.cda, c.n_cd, below, left, x0, y0, sl);
    r[3] = y0; r[4] = sl;
    interval_record(c.cp_mach, c.cp_shift, c.n_cp, below, left, x0, y0, sl);
    r[5] = x0; r[6] = y0; r[7] = sl;
  }

  {  // atmosphere layer records for the fast path (see erpl_tables.h)
    const double inf = INFINITY;
    auto rec = [&](int k, double aT, double bT, double Tlo, double Thi, double invTref, double eL,
                   double href, double eH, double eM, double base) {
      double* r = &T.atm_rec[k * ERPL_ATM_REC];
      r[0] = aT; r[1] = bT; r[2] = Tlo; r[3] = Thi; r[4] = invTref; r[5] = eL;
      r[6] = href; r[7] = eH; r[8] = eM; r[9] = base; r[10] = 0; r[11] = 0;
    };
    rec(0, -s.lapse, s.T0, -inf, inf, s.inv_T0, s.tropo_exp, 0.0, 0.0, 0.0, s.P0);
    rec(1, 0.0, s.T_strat, -inf, inf, s.inv_Ts, 0.0, s.h_tropo, s.k_iso, 0.0, s.p11);
    rec(2, 0.001, s.T_strat - 0.001 * s.h_strat, -inf, 228.65, s.inv_Ts, 0.0, s.h_strat, s.k_iso, 0.0, s.p20);
    rec(3, 0.001, s.T_strat - 0.001 * s.h_strat, -inf, 228.65, s.inv_Ts, s.grad_exp, s.h_strat, 0.0, 0.0, s.p25);
    rec(4, -0.0028, 228.65 + 0.0028 * 32000.0, 180.0, inf, 1.0 / 228.65, 0.0, 32000.0, 0.0, -s.k_meso, 868.02);
  }

  // NaN fast-forward table (see erpl_tables.h): final t and step count of
  // `t = sum_r dt_rail; while t < max_time: t += dt_flight` per rail-iteration count r.
  T.n_coast = 0;
  const double est = (c.max_time > 0) ? c.max_time / s.dt_flight : 0;
  if (est < 2.0e6) {
    std::vector<double> t0(ERPL_COAST_TABLE);
    double t = 0.0;
    for (int r = 0; r < ERPL_COAST_TABLE; ++r) { t0[r] = t; t += s.dt_rail; }
    const int nthr = 8;
    std::vector<std::thread> pool;
    const double dtf = s.dt_flight, tmax = c.max_time;
    for (int w = 0; w < nthr; ++w) {
      pool.emplace_back([&, w]() {
        for (int r = w; r < ERPL_COAST_TABLE; r += nthr) {
          double tt = t0[r];
          int32_t steps = 0;
          while (tt < tmax) { tt += dtf; ++steps; }
          T.coast_t[r] = tt;
          T.coast_steps[r] = steps;
        }
      });
    }
    for (auto& th : pool) th.join();
    T.n_coast = ERPL_COAST_TABLE;
  }
  return ERPL_OK;
}

}  // namespace

// One workspace of the context: resume queues, queue cursors and counters of ONE batch in flight.
// Hardware queues the HIP runtime multiplexes this process's streams onto: GPU_MAX_HW_QUEUES, read by the
// runtime when it initialises (default 4).  Streams that share a queue run their kernels one after the
// other, so more batches in flight than queues is slower than three (measured at 131 072 samples, fp32:
// depth 4 / 4 queues 16.9 ms per batch, depth 4 / 8 queues 11.8 ms).  The library cannot ask the runtime;
// it trusts the environment variable the host set before the first HIP call (the Python package does).
static int hw_queues_env() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("GPU_MAX_HW_QUEUES");
    const int q = e ? atoi(e) : 0;
    v = q > 0 ? q : 4;
  }
  return v;
}

// Slot 0 serves erpl_mc_run_batch (on the caller's stream); slots 0..depth-1 serve erpl_mc_submit_batch
// round-robin, each on its own internal stream.  Whoever uses a slot first waits (on the device) for
// the slot's previous batch and records `done` behind its own kernels.
#define ERPL_TICKET_RING 256
struct ErplSlot {
  void* res_r[2] = {nullptr, nullptr};      // resume-queue records (see erpl_tables.h)
  double* res_d[2] = {nullptr, nullptr};
  int32_t* res_i[2] = {nullptr, nullptr};
  unsigned long long* d_queue = nullptr;    // qcnt[ERPL_MAX_PHASES + 2], qhead[...], then the hand-over queue's two cursor arrays
  void* ext_r = nullptr;                    // hand-over queue records (fp64 throughput build -> reference-order kernel),
  double* ext_d = nullptr;                  //   allocated with the first ERPL_PREC_F64_FAST batch of the set
  int32_t* ext_i = nullptr;
  int64_t ext_cap = 0;
  unsigned long long* d_counters = nullptr; // 16 words
  int64_t cap = 0;
  hipEvent_t done = nullptr;                // everything of the slot's latest batch has run
  hipEvent_t main_done = nullptr;           // its main flight launch has (the sweep stream waits for this)
  bool used = false;                        // `done` has been recorded at least once
  int64_t ticket = 0;                       // last batch submitted through this slot
  unsigned long long* own_counters = nullptr; // pinned: d_counters[0..3] of the slot's latest erpl_mc_run_batch
  unsigned long long* h_counters = nullptr; // pinned host copy of d_counters[0..3] of the slot's latest batch: `own_counters`,
                                            // or the ticket record of the batch (erpl_mc_submit_batch)
  bool latest_is_run = false;               // the slot's latest batch came through erpl_mc_run_batch (no ticket record)
  int64_t last_n = 0, seq = 0;              // its size and its position in the order of all batches of the context
};

struct erpl_ctx {
  int device = 0;
  int n_cu = 256;
  bool has_cfg = false;
  ErplTables h_tables;            // host copy (scalars are passed to the kernels by value)
  ErplTables* d_tables = nullptr;
  // Two workspaces ("sets") per lane of erpl_mc_submit_batch, used alternately: slot[lane] and
  // slot[ERPL_MAX_OVERLAP + lane].  A batch with lane adoption runs its rail and main flight launch on the lane's
  // main stream and its sweep launches (the few long trajectories nobody adopted) on the lane's sweep stream, so
  // the lane's NEXT batch - on the other set - starts when the main launch is over and overlaps the sweeps:
  // batches of equal length submitted together run in step, and without this the tails of a whole round of
  // them met on an otherwise empty GPU before the next round could start.
  ErplSlot slot[2 * ERPL_MAX_OVERLAP];
  hipStream_t lane_stream[ERPL_MAX_OVERLAP] = {};
  hipStream_t lane_sweep[ERPL_MAX_OVERLAP] = {};
  hipEvent_t lane_in_ready[ERPL_MAX_OVERLAP] = {};
  unsigned lane_uses[ERPL_MAX_OVERLAP] = {};   // batches the lane has taken: parity picks the set
  int depth = 3;                  // slots erpl_mc_submit_batch cycles through (erpl_mc_create: 8 with enough hardware queues)
  int64_t submitted = 0;          // tickets handed out
  // One record per ticket: its own completion event and its own pinned copy of the batch's counters, so
  // that erpl_mc_check_batch(T) waits for T alone and reports T's own lost records however often T's workspace has
  // been reused since.  A ring of the last ERPL_TICKET_RING tickets; a record that leaves the ring unreported is
  // latched in `recycled_incomplete`.
  hipEvent_t ring_done[ERPL_TICKET_RING] = {};
  int64_t ring_ticket[ERPL_TICKET_RING] = {};
  unsigned long long* ring_counters = nullptr;   // pinned [ERPL_TICKET_RING][4]
  int64_t acked = 0;                // tickets <= acked have been reported by a blocking check of ALL batches
  int64_t recycled_incomplete = 0;  // first incomplete ticket that left the ring before such a check saw it
  int last_slot = 0;              // slot of the most recent batch (erpl_mc_last_stats)
  int64_t reserve_n = 0;          // erpl_mc_reserve request, applied to a slot when it is first used
  int adopt_spin = 1 << 22;       // polls of an adopting lane for a claimed record's ready word (erpl_mc_set_adopt_spin)
  int adopt = -1;                 // lane adoption: flying lanes at or below which a wave hands its lanes over; 0 = off; < 0 = by batch
  int chunk = -1;                 // steps per launch between compactions; 0 = one launch; < 0 = by the batches seen so far
  int short_depth = 4;            // erpl_mc_set_short_flight_overlap
  double seen_mean_steps = 0.0;   // physics RK4 steps per trajectory of the most recent COMPLETED batch
  int64_t seen_seq = 0, batches = 0;
  int waves = 0;   // 0 = choose by batch size
  // one wave per workgroup: a finished wave frees its slot for the next batch at once (measured 2-5 %
  // over 256-thread workgroups, alone and overlapped); refill as soon as a lane is idle (best: 1..4)
  int block = 64, max_blocks = 0, refill = 1;
  bool profiling = false;
  long long profiled_runs = 0;
  hipEvent_t ev[3 * ERPL_PROFILE_RING] = {};
};

namespace {

void slot_free_workspace(ErplSlot& s) {
  for (int k = 0; k < 2; ++k) {
    (void)hipFree(s.res_r[k]); (void)hipFree(s.res_d[k]); (void)hipFree(s.res_i[k]);
    s.res_r[k] = nullptr; s.res_d[k] = nullptr; s.res_i[k] = nullptr;
  }
  (void)hipFree(s.ext_r); (void)hipFree(s.ext_d); (void)hipFree(s.ext_i);
  s.ext_r = nullptr; s.ext_d = nullptr; s.ext_i = nullptr; s.ext_cap = 0;
  s.cap = 0;
}

// The hand-over queue of the set (one more record buffer), for batches of the fp64 throughput build.
int slot_reserve_handoff(ErplSlot& s) {
  if (s.ext_cap >= s.cap) return ERPL_OK;
  if (s.used) HIP_TRY(hipEventSynchronize(s.done));
  (void)hipFree(s.ext_r); (void)hipFree(s.ext_d); (void)hipFree(s.ext_i);
  s.ext_r = nullptr; s.ext_d = nullptr; s.ext_i = nullptr; s.ext_cap = 0;
  const size_t rows = (size_t)s.cap;
  HIP_TRY(hipMalloc(&s.ext_r, rows * ERPL_RES_R * sizeof(double)));
  HIP_TRY(hipMalloc((void**)&s.ext_d, rows * ERPL_RES_D * sizeof(double)));
  HIP_TRY(hipMalloc((void**)&s.ext_i, rows * ERPL_RES_I * sizeof(int32_t)));
  s.ext_cap = s.cap;
  return ERPL_OK;
}

// Grows the slot's workspace to n samples.  The slot's previous batch may still be using the old one.
int slot_reserve(ErplSlot& s, int64_t n) {
  if (n <= s.cap) return ERPL_OK;
  if (s.used) HIP_TRY(hipEventSynchronize(s.done));
  slot_free_workspace(s);
  const size_t rows = (size_t)n;
  for (int k = 0; k < 2; ++k) {
    HIP_TRY(hipMalloc(&s.res_r[k], rows * ERPL_RES_R * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&s.res_d[k], rows * ERPL_RES_D * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&s.res_i[k], rows * ERPL_RES_I * sizeof(int32_t)));
  }
  s.cap = n;
  return ERPL_OK;
}

int slot_init(ErplSlot& s) {
  if (s.d_queue) return ERPL_OK;
  HIP_TRY(hipMalloc((void**)&s.d_counters, 16 * sizeof(unsigned long long)));
  HIP_TRY(hipMalloc((void**)&s.d_queue, (2 * (ERPL_MAX_PHASES + 2) + 2 * ERPL_EXT_Q) * sizeof(unsigned long long)));
  HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&s.main_done, hipEventDisableTiming));
  HIP_TRY(hipHostMalloc((void**)&s.own_counters, 4 * sizeof(unsigned long long), hipHostMallocDefault));
  memset(s.own_counters, 0, 4 * sizeof(unsigned long long));
  s.h_counters = s.own_counters;
  return ERPL_OK;
}

void slot_destroy(ErplSlot& s) {
  slot_free_workspace(s);
  (void)hipFree(s.d_counters); (void)hipFree(s.d_queue);
  if (s.done) (void)hipEventDestroy(s.done);
  if (s.main_done) (void)hipEventDestroy(s.main_done);
  if (s.own_counters) (void)hipHostFree(s.own_counters);
  s = ErplSlot();
}

int check_batch(const erpl_ctx* c, const erpl_batch* b, const erpl_out* o) {
  if (!c || !b || !o) return fail(ERPL_ERR_INVALID, "NULL argument");
  if (!c->has_cfg) return fail(ERPL_ERR_CONFIG, "erpl_mc_set_config has not been called");
  if (b->n < 0) return fail(ERPL_ERR_INVALID, "negative batch size");
  if (b->n == 0) return ERPL_OK;
  if (b->n > 2147483647LL) return fail(ERPL_ERR_INVALID, "at most 2^31 - 1 samples per batch");
  if (b->precision != ERPL_PREC_F64 && b->precision != ERPL_PREC_F32 && b->precision != ERPL_PREC_F64_FAST)
    return fail(ERPL_ERR_INVALID, "unknown precision %d", b->precision);
  if (b->k_wind < 0 || b->k_wind > ERPL_MAX_WIND_KNOTS)
    return fail(ERPL_ERR_INVALID, "k_wind %d out of range 0..%d", b->k_wind, ERPL_MAX_WIND_KNOTS);
  if (!b->ic || !b->rocket || !b->motor) return fail(ERPL_ERR_INVALID, "NULL input buffer");
  if (b->k_wind > 0 && (!b->alt_grid || !b->wind)) return fail(ERPL_ERR_INVALID, "k_wind > 0 but no wind buffers");
  if (!o->summary || !o->status) return fail(ERPL_ERR_INVALID, "NULL output buffer");
  if (o->n_traj < 0 || (o->n_traj > 0 && (!o->traj_ids || !o->traj || !o->traj_len || o->traj_cap < 1 || o->traj_stride < 1)))
    return fail(ERPL_ERR_INVALID, "inconsistent trajectory-capture arguments");
  return ERPL_OK;
}

// Kernel arguments shared by every entry point that launches device code for a batch.
void fill_common_args(const erpl_ctx* c, const erpl_batch* b, ErplKArgs& a) {
  memset(&a, 0, sizeof(a));
  a.n = b->n; a.k_wind = b->k_wind; a.flags = b->flags;
  a.ic = b->ic; a.rocket = b->rocket; a.motor = b->motor; a.alt_grid = b->alt_grid; a.wind = b->wind;
  a.tables = c->d_tables;
  const ErplTables& T = c->h_tables;
  a.n_union = T.n_union; a.n_curve = T.n_curve; a.motor_kind = T.motor_kind; a.n_coast = T.n_coast;
  a.dt_rail = T.dt_rail; a.dt_flight = T.dt_flight; a.max_time = T.max_time;
}

// Trajectory length is not known in advance: the scheduling choices that depend on it (step chunks, how many batches of
// short flights start side by side) follow the batches this context has already FINISHED - their device step counter is
// copied to pinned memory behind every batch.
constexpr double kLongFlightSteps = 8192.0;
void note_finished_batches(erpl_ctx* c) {
  for (int i = 0; i < 2 * ERPL_MAX_OVERLAP; ++i) {
    ErplSlot& q = c->slot[i];
    if (q.used && q.seq > c->seen_seq && q.last_n > 0 && hipEventQuery(q.done) == hipSuccess) {
      c->seen_mean_steps = (double)q.h_counters[1] / (double)q.last_n;
      c->seen_seq = q.seq;
    }
  }
}

// Rail + flight kernels of one batch through the lane's next set, on stream `st`; `sweep` (or NULL) = the stream
// the launches behind the main one go to when the batch runs with lane adoption.
int enqueue_batch(erpl_ctx* c, int lane, const erpl_batch* b, const erpl_out* o, hipStream_t st, int in_flight,
                  hipStream_t sweep, int64_t ticket, unsigned long long* ring_slot = nullptr, hipEvent_t ring_done = nullptr) {
  // two workspaces per lane only where the lane's next batch may start beside the sweeps of its previous one (a sweep
  // stream exists); erpl_mc_run_batch and lanes without lane adoption stay on their first set
  const int si = lane + ((sweep && (c->lane_uses[lane] & 1u)) ? ERPL_MAX_OVERLAP : 0);
  ErplSlot& s = c->slot[si];
  int rc = slot_init(s);
  if (rc != ERPL_OK) return rc;
  rc = slot_reserve(s, (b->n > c->reserve_n) ? b->n : c->reserve_n);
  if (rc != ERPL_OK) return rc;
  // the slot's previous batch (possibly on another stream) must have drained its queues
  if (s.used) HIP_TRY(hipStreamWaitEvent(st, s.done, 0));
  // (queue cursors and counters are zeroed by the rail kernel itself)
  ErplKArgs a;
  fill_common_args(c, b, a);
  a.summary = o->summary; a.status = o->status;
  for (int k = 0; k < 2; ++k) { a.res_r[k] = s.res_r[k]; a.res_d[k] = s.res_d[k]; a.res_i[k] = s.res_i[k]; }
  a.res_cap = s.cap;
  a.qcnt = s.d_queue; a.qhead = s.d_queue + (ERPL_MAX_PHASES + 2);
  if (b->precision == ERPL_PREC_F64_FAST) {
    rc = slot_reserve_handoff(s);
    if (rc != ERPL_OK) return rc;
    a.ext_r = s.ext_r; a.ext_d = s.ext_d; a.ext_i = s.ext_i;
  }
  a.ext_q = s.d_queue + 2 * (ERPL_MAX_PHASES + 2);
  a.ext_cnt = a.ext_q + 1;
  a.n_traj = o->n_traj; a.traj_stride = o->traj_stride; a.traj_cap = o->traj_cap;
  a.traj_ids = o->traj_ids; a.traj = o->traj; a.traj_len = o->traj_len;
  a.counters = s.d_counters;
  a.refill_threshold = c->refill;
  const ErplTables& T = c->h_tables;
  const int max_blocks = c->max_blocks > 0 ? c->max_blocks : c->n_cu * 8 * (256 / c->block);
  // the three-wave build pays once three resident waves per SIMD stay busy: a batch that refills them a
  // few times over (measured +4..12 % from 3 rounds up; between 1 and 3 rounds the rounding of "rounds"
  // decides), or several batches in flight sharing the SIMDs (131 072 samples x 3 deep: +13 %)
  const bool dense = b->n >= (int64_t)c->n_cu * 4 * 64 * 3 * 3 ||
                     (in_flight >= 2 && b->n * in_flight >= (int64_t)c->n_cu * 4 * 64 * 3);
  a.waves_per_simd = c->waves ? c->waves : (dense ? 3 : 2);
  // step-chunked launches with compaction in between (erpl_mc_set_chunk); every lane ends within
  // ceil(max_time / dt) + 1 steps, so that many steps' worth of chunks drains the queue
  int n_phases = 1;
  a.chunk_steps = 0;
  // Automatic step chunks (erpl_mc_set_chunk < 0, the default): compaction between step-chunked launches pays when
  // trajectories are long AND something else fills the GPU at every chunk barrier - i.e. for overlapped batches
  // of long flights (measured three deep at 131 072 samples: 15 k-step flights -20 %, 42 k-step flights -17 %,
  // 2.5 k-step flights +3 %).  Trajectory length is not known in advance, so the choice follows the batches this
  // context has already finished: their device step counter is copied to pinned memory behind every batch.
  // Results do not depend on the choice (bitwise).
  int chunk_steps = c->chunk;
  if (chunk_steps < 0) {
    note_finished_batches(c);
    chunk_steps = (in_flight >= 2 && o->n_traj == 0 && c->seen_mean_steps >= kLongFlightSteps) ? 2048 : 0;
  }
  if (chunk_steps > 0 && T.max_time > 0) {
    const double max_steps = ceil(T.max_time / T.dt_flight) + 2.0;
    double chunk = (double)chunk_steps;
    if (ceil(max_steps / chunk) + 1.0 > (double)ERPL_MAX_PHASES) chunk = ceil(max_steps / (double)(ERPL_MAX_PHASES - 2));
    a.chunk_steps = (int)chunk;
    n_phases = (int)ceil(max_steps / chunk) + 1;
  }
  // Lane adoption (erpl_mc_set_adopt): two sweep launches behind the main one fly out what no running wave
  // adopted - the first parks its own thin waves once more, the last one never parks.
  // Automatic (erpl_mc_set_adopt < 0, the default): every batch handed over with erpl_mc_submit_batch while the lanes
  // have hardware queues of their own (two streams each: 2 x depth + 2 with the caller's).  The sweeps run on the
  // lane's second stream, so the next batch of the lane follows the main launch at once and the few long
  // trajectories of a batch finish beside it (131 072 samples, fp32: 32.4 -> 20.8 ms one deep, 16.5 -> 10.8 two
  // deep, 11.3 -> 9.0 three deep, 10.6 -> 9.0 eight deep; fp64 throughput build 44.7 -> 38.1 three deep, 37.7 ->
  // 35.9 eight deep; the gate kernel 129 -> 116 three deep).  Without the queues the second stream of a lane
  // lands on another lane's queue and the hand-overs cost more than they save (four queues, three deep: 11.4 ->
  // 25.5 ms): off.  erpl_mc_run_batch runs on the caller's one stream, where a batch is bound by its own longest
  // trajectory and the hand-overs only lengthen that (32.5 -> 36.1 ms): off.  Step chunks already re-pack every
  // lane, and chunk-parked records would be adopted straight back (measured 8x slower): exclusive.
  int adopt = c->adopt;
  // (limit: fp32 12 / 16 / 24 / 32 / 48 -> 9.30 / 9.05 / 9.00 / 8.93 / 8.97 ms; the one-wave-per-SIMD fp64 builds like it
  // higher - 24 / 40 / 48 / 56 -> 35.7 / 34.8 / 35.3 / 35.3 ms eight deep, 40.3 / 38.4 / 38.2 / 40.9 three deep)
  if (adopt < 0) adopt = (sweep && hw_queues_env() >= 2 * in_flight + 2) ? (b->precision == ERPL_PREC_F32 ? 24 : 40) : 0;
  a.adopt_lanes = (o->n_traj == 0 && a.chunk_steps == 0) ? adopt : 0;
  a.adopt_spin = c->adopt_spin;
  if (a.adopt_lanes > 0 && n_phases < 3) n_phases = 3;
  void** ev = c->profiling ? (void**)&c->ev[3 * (c->profiled_runs % ERPL_PROFILE_RING)] : nullptr;
  // with lane adoption the launches behind the main one hold the batch's few longest trajectories: they go to the
  // lane's sweep stream, and the lane's next batch (other set) follows the main launch at once
  hipStream_t tail = (sweep && a.adopt_lanes > 0) ? sweep : nullptr;
  int lrc;
  if (b->precision == ERPL_PREC_F64) lrc = erpl_launch_f64(a, &T.s64, c->block, max_blocks, n_phases, st, ev, tail, s.main_done);
  else if (b->precision == ERPL_PREC_F64_FAST) lrc = erpl_launch_f64f(a, &T.s64, c->block, max_blocks, n_phases, st, ev, tail, s.main_done);
  else lrc = erpl_launch_f32(a, &T.s32, c->block, max_blocks, n_phases, st, ev, tail, s.main_done);
  if (c->profiling && lrc == 0) c->profiled_runs++;
  if (lrc != 0) return fail(ERPL_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)lrc));
  hipStream_t last = tail ? tail : st;
  s.h_counters = ring_slot ? ring_slot : s.own_counters;
  HIP_TRY(hipMemcpyAsync(s.h_counters, s.d_counters, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, last));
  HIP_TRY(hipEventRecord(s.done, last));
  if (ring_done) HIP_TRY(hipEventRecord(ring_done, last));
  s.used = true;
  s.latest_is_run = ticket <= 0;
  if (ticket > 0) s.ticket = ticket;   // (an erpl_mc_run_batch on this set leaves the ticket: its `done` is later and covers it)
  s.last_n = b->n;
  s.seq = ++c->batches;
  c->last_slot = si;
  c->lane_uses[lane]++;
  return ERPL_OK;
}

int wait_all_host(erpl_ctx* c) {
  for (int i = 0; i < 2 * ERPL_MAX_OVERLAP; ++i)
    if (c->slot[i].used) HIP_TRY(hipEventSynchronize(c->slot[i].done));
  return ERPL_OK;
}

}  // namespace

extern "C" {

int erpl_mc_abi_version(void) { return ERPL_MC_ABI_VERSION; }
const char* erpl_mc_last_error(void) { return g_err; }

int erpl_mc_create(int device, erpl_ctx** out) {
  if (!out) return fail(ERPL_ERR_INVALID, "out is NULL");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(ERPL_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= count) return fail(ERPL_ERR_INVALID, "device %d out of range (%d)", device, count);
  HIP_TRY(hipSetDevice(device));
  erpl_ctx* c = new erpl_ctx();
  c->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
  // two streams per lane (main, sweep) + the caller's stream and one more of its own
  c->depth = (hw_queues_env() >= 2 * ERPL_MAX_OVERLAP + 2) ? ERPL_MAX_OVERLAP : ((hw_queues_env() >= 12) ? (hw_queues_env() - 2) / 2 : 3);
  hipError_t e = hipMalloc((void**)&c->d_tables, sizeof(ErplTables));
  for (int i = 0; i < 3 * ERPL_PROFILE_RING && e == hipSuccess; ++i) e = hipEventCreate(&c->ev[i]);
  if (e != hipSuccess) { (void)erpl_mc_destroy(c); return fail(ERPL_ERR_HIP, "hipMalloc/hipEventCreate: %s", hipGetErrorString(e)); }
  if (slot_init(c->slot[0]) != ERPL_OK) { (void)erpl_mc_destroy(c); return ERPL_ERR_HIP; }
  *out = c;
  return ERPL_OK;
}

int erpl_mc_destroy(erpl_ctx* c) {
  if (!c) return ERPL_OK;
  (void)hipSetDevice(c->device);
  (void)wait_all_host(c);
  (void)hipFree(c->d_tables);
  for (int i = 0; i < 2 * ERPL_MAX_OVERLAP; ++i) slot_destroy(c->slot[i]);
  for (int i = 0; i < ERPL_MAX_OVERLAP; ++i) {
    if (c->lane_stream[i]) (void)hipStreamDestroy(c->lane_stream[i]);
    if (c->lane_sweep[i]) (void)hipStreamDestroy(c->lane_sweep[i]);
    if (c->lane_in_ready[i]) (void)hipEventDestroy(c->lane_in_ready[i]);
  }
  for (int i = 0; i < 3 * ERPL_PROFILE_RING; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  for (int i = 0; i < ERPL_TICKET_RING; ++i) if (c->ring_done[i]) (void)hipEventDestroy(c->ring_done[i]);
  if (c->ring_counters) (void)hipHostFree(c->ring_counters);
  delete c;
  return ERPL_OK;
}

int erpl_mc_set_config(erpl_ctx* c, const erpl_config* cfg) {
  if (!c || !cfg) return fail(ERPL_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());   // a batch still in flight on some stream reads the tables being replaced
  ErplTables& T = c->h_tables;
  c->has_cfg = false;
  int rc = build_tables(*cfg, T);
  if (rc != ERPL_OK) return rc;
  HIP_TRY(hipMemcpy(c->d_tables, &T, sizeof(T), hipMemcpyHostToDevice));
  c->has_cfg = true;
  return ERPL_OK;
}

int erpl_mc_reserve(erpl_ctx* c, int64_t n) {
  if (!c || n < 0) return fail(ERPL_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  if (n > c->reserve_n) c->reserve_n = n;
  // both workspaces of every lane in use now (erpl_mc_run_batch on lane 0 stays allocation-free, hence
  // graph-capturable, and no erpl_mc_submit_batch allocates in the middle of a run); lanes beyond the current depth
  // that have been used before grow too, fresh ones take the size on first use
  // (the second workspace of a lane is only ever used with lane adoption on: without it, it is not allocated -
  // a workspace costs 448 bytes per sample)
  const bool may_adopt = c->adopt > 0 || (c->adopt < 0 && hw_queues_env() >= 2 * c->depth + 2);
  for (int i = 0; i < 2 * ERPL_MAX_OVERLAP; ++i) {
    if (i % ERPL_MAX_OVERLAP >= c->depth && !c->slot[i].d_queue) continue;
    if (i >= ERPL_MAX_OVERLAP && !may_adopt && !c->slot[i].d_queue) continue;
    int rc = slot_init(c->slot[i]);
    if (rc == ERPL_OK) rc = slot_reserve(c->slot[i], c->reserve_n);
    if (rc != ERPL_OK) return rc;
  }
  return ERPL_OK;
}

int erpl_mc_set_chunk(erpl_ctx* c, int chunk_steps) {
  if (!c) return fail(ERPL_ERR_INVALID, "NULL ctx");
  c->chunk = chunk_steps < 0 ? -1 : chunk_steps;
  return ERPL_OK;
}

int erpl_mc_set_waves_per_simd(erpl_ctx* c, int waves) {
  if (!c) return fail(ERPL_ERR_INVALID, "NULL ctx");
  if (waves != 0 && waves != 2 && waves != 3) return fail(ERPL_ERR_INVALID, "waves per SIMD must be 0 (auto), 2 or 3");
  c->waves = waves;
  return ERPL_OK;
}

int erpl_mc_set_adopt_spin(erpl_ctx* c, int polls) {
  if (!c) return fail(ERPL_ERR_INVALID, "null context");
  c->adopt_spin = polls;
  return ERPL_OK;
}

int erpl_mc_set_adopt(erpl_ctx* c, int lanes) {
  if (!c) return fail(ERPL_ERR_INVALID, "null context");
  if (lanes > 63) return fail(ERPL_ERR_INVALID, "adopt lanes must be at most 63");
  c->adopt = lanes < 0 ? -1 : lanes;
  return ERPL_OK;
}

int erpl_mc_set_launch(erpl_ctx* c, int block_threads, int max_blocks, int refill_threshold) {
  if (!c) return fail(ERPL_ERR_INVALID, "NULL ctx");
  if (block_threads != 64 && block_threads != 128 && block_threads != 256)
    return fail(ERPL_ERR_INVALID, "block_threads must be 64, 128 or 256");
  if (refill_threshold < 1 || refill_threshold > 64) return fail(ERPL_ERR_INVALID, "refill_threshold must be 1..64");
  c->block = block_threads;
  c->max_blocks = max_blocks < 0 ? 0 : max_blocks;
  c->refill = refill_threshold;
  return ERPL_OK;
}

int erpl_mc_run_batch(erpl_ctx* c, const erpl_batch* b, const erpl_out* o, void* stream) {
  int rc = check_batch(c, b, o);
  if (rc != ERPL_OK || b->n == 0) return rc;
  HIP_TRY(hipSetDevice(c->device));
  return enqueue_batch(c, 0, b, o, (hipStream_t)stream, 1, nullptr, 0);
}

int erpl_mc_get_overlap(erpl_ctx* c) { return c ? c->depth : 0; }

int erpl_mc_set_overlap(erpl_ctx* c, int depth) {
  if (!c) return fail(ERPL_ERR_INVALID, "NULL ctx");
  if (depth < 1 || depth > ERPL_MAX_OVERLAP) return fail(ERPL_ERR_INVALID, "overlap depth must be 1..%d", ERPL_MAX_OVERLAP);
  HIP_TRY(hipSetDevice(c->device));
  int rc = wait_all_host(c);
  if (rc != ERPL_OK) return rc;
  c->depth = depth;
  return ERPL_OK;
}

int erpl_mc_set_short_flight_overlap(erpl_ctx* c, int depth) {
  if (!c) return fail(ERPL_ERR_INVALID, "NULL ctx");
  if (depth < 0 || depth > ERPL_MAX_OVERLAP) return fail(ERPL_ERR_INVALID, "short-flight overlap must be 0..%d", ERPL_MAX_OVERLAP);
  c->short_depth = depth;
  return ERPL_OK;
}

int erpl_mc_submit_batch(erpl_ctx* c, const erpl_batch* b, const erpl_out* o, void* stream, int64_t* ticket) {
  int rc = check_batch(c, b, o);
  if (rc != ERPL_OK) return rc;
  if (ticket) *ticket = c->submitted;   // an empty batch is complete as soon as its predecessors are
  if (b->n == 0) return ERPL_OK;
  HIP_TRY(hipSetDevice(c->device));
  // Short flights (erpl_mc_set_short_flight_overlap) go round fewer lanes: fewer streams busy at a time
  int depth = c->depth;
  if (c->short_depth > 0 && c->short_depth < depth) {
    note_finished_batches(c);
    if (c->seen_mean_steps > 0.0 && c->seen_mean_steps < kLongFlightSteps) depth = c->short_depth;
  }
  const int lane = (int)(c->submitted % depth);
  if (!c->lane_stream[lane]) HIP_TRY(hipStreamCreateWithFlags(&c->lane_stream[lane], hipStreamNonBlocking));
  // the sweep stream only where lane adoption can come on: a stream takes a hardware queue, and with the HIP default
  // of four a second one per lane would push the main streams onto shared queues
  const bool may_adopt = c->adopt > 0 || (c->adopt < 0 && hw_queues_env() >= 2 * c->depth + 2);
  if (may_adopt && !c->lane_sweep[lane]) HIP_TRY(hipStreamCreateWithFlags(&c->lane_sweep[lane], hipStreamNonBlocking));
  if (!c->lane_in_ready[lane]) HIP_TRY(hipEventCreateWithFlags(&c->lane_in_ready[lane], hipEventDisableTiming));
  // inputs written on the caller's stream so far are visible to the batch
  HIP_TRY(hipEventRecord(c->lane_in_ready[lane], (hipStream_t)stream));
  HIP_TRY(hipStreamWaitEvent(c->lane_stream[lane], c->lane_in_ready[lane], 0));
  // the ticket's own record: completion event + pinned counters (recycled ERPL_TICKET_RING tickets later)
  const int64_t t_new = c->submitted + 1;
  const int ri = (int)(t_new % ERPL_TICKET_RING);
  if (!c->ring_counters) {
    HIP_TRY(hipHostMalloc((void**)&c->ring_counters, ERPL_TICKET_RING * 4 * sizeof(unsigned long long), hipHostMallocDefault));
    memset(c->ring_counters, 0, ERPL_TICKET_RING * 4 * sizeof(unsigned long long));
  }
  if (!c->ring_done[ri]) HIP_TRY(hipEventCreateWithFlags(&c->ring_done[ri], hipEventDisableTiming));
  if (c->ring_ticket[ri] > 0) {   // the record of ticket t_new - ERPL_TICKET_RING leaves the ring: keep what nobody has been told yet
    HIP_TRY(hipEventSynchronize(c->ring_done[ri]));
    if (c->ring_counters[4 * ri + 3] != 0ull && c->ring_ticket[ri] > c->acked && c->recycled_incomplete == 0)
      c->recycled_incomplete = c->ring_ticket[ri];
  }
  for (int i = 0; i < 2 * ERPL_MAX_OVERLAP; ++i)   // (a set idle since that ticket must not read the record's next life)
    if (c->slot[i].h_counters == &c->ring_counters[4 * ri]) c->slot[i].h_counters = c->slot[i].own_counters;
  memset(&c->ring_counters[4 * ri], 0, 4 * sizeof(unsigned long long));
  c->ring_ticket[ri] = 0;
  rc = enqueue_batch(c, lane, b, o, c->lane_stream[lane], depth, may_adopt ? c->lane_sweep[lane] : nullptr, t_new,
                     &c->ring_counters[4 * ri], c->ring_done[ri]);
  if (rc != ERPL_OK) return rc;
  c->ring_ticket[ri] = t_new;
  ++c->submitted;
  if (ticket) *ticket = c->submitted;
  return ERPL_OK;
}

namespace {
// the record of ticket t, or -1 once it has left the ring (then the batch has finished: recycling waited for it)
int ring_index(const erpl_ctx* c, int64_t t) {
  const int ri = (int)(t % ERPL_TICKET_RING);
  return (t > 0 && c->ring_ticket[ri] == t) ? ri : -1;
}
int report_incomplete(int64_t t, unsigned long long lost) {
  return fail(ERPL_ERR_INCOMPLETE, "lane hand-over timed out in batch %lld: %llu record(s) lost, their samples carry ERPL_ST_INCOMPLETE",
              (long long)t, lost);
}
// Blocking check of EVERY batch handed to the context so far; a failure is reported once: tickets up to the last one
// are acknowledged afterwards (erpl_mc_check_batch(T) keeps answering for T itself while T's record is in the ring).
int check_all(erpl_ctx* c) {
  const int rc = wait_all_host(c);
  if (rc != ERPL_OK) return rc;
  int64_t bad = c->recycled_incomplete;
  unsigned long long lost = 0ull;
  for (int i = 0; i < ERPL_TICKET_RING; ++i) {
    const int64_t t = c->ring_ticket[i];
    if (t > c->acked && c->ring_counters[4 * i + 3] != 0ull && (bad == 0 || t < bad)) { bad = t; lost = c->ring_counters[4 * i + 3]; }
  }
  c->acked = c->submitted;
  c->recycled_incomplete = 0;
  if (bad > 0) return report_incomplete(bad, lost);
  for (int i = 0; i < 2 * ERPL_MAX_OVERLAP; ++i)   // erpl_mc_run_batch batches carry no ticket: the set's own copy
    if (c->slot[i].used && c->slot[i].latest_is_run && c->slot[i].own_counters[3] != 0ull)
      return report_incomplete(0, c->slot[i].own_counters[3]);
  return ERPL_OK;
}
}  // namespace

int erpl_mc_wait_batch(erpl_ctx* c, int64_t ticket, void* stream) {
  if (!c) return fail(ERPL_ERR_INVALID, "NULL ctx");
  if (ticket > c->submitted) return fail(ERPL_ERR_INVALID, "ticket %lld has not been handed out", (long long)ticket);
  HIP_TRY(hipSetDevice(c->device));
  if (ticket > 0) {
    // the ticket's own event (a ticket that has left the ring has finished: nothing to order behind)
    const int ri = ring_index(c, ticket);
    if (ri >= 0) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, c->ring_done[ri], 0));
  } else if (ticket < 0) {
    // every set's latest batch; a set is reused only behind its previous batch, so this covers all of them
    // (erpl_mc_run_batch's batches are ordered by the caller's own stream)
    for (int i = 0; i < 2 * ERPL_MAX_OVERLAP; ++i)
      if (c->slot[i].used && !c->slot[i].latest_is_run) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, c->slot[i].done, 0));
  }
  // the host does not block here, so only batches that have ALREADY finished can be reported (their counters sit in
  // pinned memory behind their event); erpl_mc_check_batch / erpl_mc_synchronize are the blocking checks
  if (c->recycled_incomplete > 0) return report_incomplete(c->recycled_incomplete, 0ull);
  for (int i = 0; i < ERPL_TICKET_RING; ++i) {
    const int64_t t = c->ring_ticket[i];
    if (t > c->acked && c->ring_counters[4 * i + 3] != 0ull && hipEventQuery(c->ring_done[i]) == hipSuccess)
      return report_incomplete(t, c->ring_counters[4 * i + 3]);
  }
  return ERPL_OK;
}

int erpl_mc_check_batch(erpl_ctx* c, int64_t ticket) {
  if (!c) return fail(ERPL_ERR_INVALID, "NULL ctx");
  if (ticket > c->submitted) return fail(ERPL_ERR_INVALID, "ticket %lld has not been handed out", (long long)ticket);
  HIP_TRY(hipSetDevice(c->device));
  if (ticket <= 0) return check_all(c);
  const int ri = ring_index(c, ticket);
  if (ri < 0) {   // older than the ring: finished long ago; what is left of it is the latch
    if (c->recycled_incomplete > 0) return report_incomplete(c->recycled_incomplete, 0ull);
    return ERPL_OK;
  }
  HIP_TRY(hipEventSynchronize(c->ring_done[ri]));   // this batch alone: later batches keep running
  if (c->ring_counters[4 * ri + 3] != 0ull) return report_incomplete(ticket, c->ring_counters[4 * ri + 3]);
  return ERPL_OK;
}

int erpl_mc_synchronize(erpl_ctx* c) {
  if (!c) return fail(ERPL_ERR_INVALID, "NULL ctx");
  HIP_TRY(hipSetDevice(c->device));
  return check_all(c);
}

int erpl_mc_set_profiling(erpl_ctx* c, int enable) {
  if (!c) return fail(ERPL_ERR_INVALID, "NULL ctx");
  c->profiling = enable != 0;
  c->profiled_runs = 0;
  return ERPL_OK;
}

int erpl_mc_kernel_ms_history(erpl_ctx* c, int max, float* rail_ms, float* flight_ms, int* n_out) {
  if (!c || !n_out || max < 0) return fail(ERPL_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  long long avail = c->profiled_runs < ERPL_PROFILE_RING ? c->profiled_runs : ERPL_PROFILE_RING;
  long long m = avail < max ? avail : max;
  for (long long k = 0; k < m; ++k) {
    const long long run = c->profiled_runs - m + k;
    hipEvent_t* e = &c->ev[3 * (run % ERPL_PROFILE_RING)];
    HIP_TRY(hipEventSynchronize(e[2]));
    float a = 0.f, b = 0.f;
    HIP_TRY(hipEventElapsedTime(&a, e[0], e[1]));
    HIP_TRY(hipEventElapsedTime(&b, e[1], e[2]));
    if (rail_ms) rail_ms[k] = a;
    if (flight_ms) flight_ms[k] = b;
  }
  *n_out = (int)m;
  return ERPL_OK;
}

int erpl_mc_last_kernel_ms(erpl_ctx* c, float* rail_ms, float* flight_ms) {
  if (!c) return fail(ERPL_ERR_INVALID, "NULL ctx");
  if (c->profiled_runs <= 0) return fail(ERPL_ERR_INVALID, "no profiled run_batch on this context");
  int n = 0;
  return erpl_mc_kernel_ms_history(c, 1, rail_ms, flight_ms, &n);
}

// ------------------------------------------------------------------ legacy RandomState streams
// MT19937 seeded like numpy.random.RandomState(int) (init_genrand), 53-bit doubles and the polar
// gaussian with its one-value cache - the published algorithms of the generator the reference draws
// from (monte_carlo.py:157 `np.random.RandomState(i)`).  Host code; built without FMA contraction so
// that x1*x1 + x2*x2 rounds like the baseline x86-64 build of NumPy.
namespace {
struct LegacyRS {
  uint32_t key[624];
  int pos;
  bool has_gauss;
  double gauss;
  void seed(uint32_t s) {
    for (int i = 0; i < 624; ++i) { key[i] = s; s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u; }
    pos = 624; has_gauss = false; gauss = 0.0;
  }
  void refill() {
    const uint32_t A = 0x9908b0dfu, UP = 0x80000000u, LO = 0x7fffffffu;
    int k = 0;
    for (; k < 624 - 397; ++k) { uint32_t y = (key[k] & UP) | (key[k + 1] & LO); key[k] = key[k + 397] ^ (y >> 1) ^ ((y & 1u) ? A : 0u); }
    for (; k < 623; ++k) { uint32_t y = (key[k] & UP) | (key[k + 1] & LO); key[k] = key[k + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? A : 0u); }
    uint32_t y = (key[623] & UP) | (key[0] & LO);
    key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
    pos = 0;
  }
  uint32_t next32() {
    if (pos == 624) refill();
    uint32_t y = key[pos++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
  }
  double next_double() {
    const uint32_t a = next32() >> 5, b = next32() >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
  }
  double next_gauss() {
    if (has_gauss) { const double g = gauss; has_gauss = false; gauss = 0.0; return g; }
    double x1, x2, r2;
    do {
      x1 = 2.0 * next_double() - 1.0;
      x2 = 2.0 * next_double() - 1.0;
      r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    const double f = sqrt(-2.0 * log(r2) / r2);
    gauss = f * x1; has_gauss = true;
    return f * x2;
  }
};
}  // namespace

int erpl_mc_legacy_random_streams(const uint32_t* seeds, int64_t n, const uint8_t* ops, int32_t m,
                                  double* out, int32_t by_output, int32_t threads) {
  if (n < 0 || m < 0) return fail(ERPL_ERR_INVALID, "negative size");
  if (n == 0 || m == 0) return ERPL_OK;
  if (!seeds || !ops || !out) return fail(ERPL_ERR_INVALID, "NULL buffer");
  for (int32_t j = 0; j < m; ++j)
    if (ops[j] != ERPL_RS_GAUSS && ops[j] != ERPL_RS_DOUBLE) return fail(ERPL_ERR_INVALID, "unknown stream op %d", (int)ops[j]);
  int nthr = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
  if (nthr < 1) nthr = 1;
  if (threads <= 0 && nthr > 32) nthr = 32;   // containers often expose more cores than their quota
  if ((int64_t)nthr * 64 > n) nthr = (int)((n + 63) / 64);   // at least 64 streams per thread
  if (nthr < 1) nthr = 1;
  auto work = [&](int w) {
    LegacyRS rs;
    const int64_t lo = n * w / nthr, hi = n * (w + 1) / nthr;
    for (int64_t i = lo; i < hi; ++i) {
      rs.seed(seeds[i]);
      double* o = by_output ? out + i : out + i * (int64_t)m;
      const int64_t stride = by_output ? n : 1;
      for (int32_t j = 0; j < m; ++j) o[j * stride] = (ops[j] == ERPL_RS_GAUSS) ? rs.next_gauss() : rs.next_double();
    }
  };
  if (nthr == 1) { work(0); return ERPL_OK; }
  std::vector<std::thread> pool;
  for (int w = 0; w < nthr; ++w) pool.emplace_back(work, w);
  for (auto& t : pool) t.join();
  return ERPL_OK;
}

namespace {
int host_threads(int32_t requested, int64_t n) {
  int nthr = requested > 0 ? requested : (int)std::thread::hardware_concurrency();
  if (nthr < 1) nthr = 1;
  if (requested <= 0 && nthr > 32) nthr = 32;   // containers often expose more cores than their quota
  if ((int64_t)nthr * 64 > n) nthr = (int)((n + 63) / 64);   // at least 64 samples per thread
  return nthr < 1 ? 1 : nthr;
}
void run_threads(int nthr, const std::function<void(int)>& work) {
  if (nthr == 1) { work(0); return; }
  std::vector<std::thread> pool;
  for (int w = 0; w < nthr; ++w) pool.emplace_back(work, w);
  for (auto& t : pool) t.join();
}
}  // namespace

int erpl_mc_legacy_wind_profiles(const uint32_t* seeds, int64_t n, int32_t k, const double* sigma,
                                 const double* rho, const double* innov, const double* base,
                                 const double* mean_scale, const double* speed, const double* cdir,
                                 const double* sdir, double* wind, int32_t threads) {
  if (n < 0 || k < 0) return fail(ERPL_ERR_INVALID, "negative size");
  if (n == 0 || k == 0) return ERPL_OK;
  if (!seeds || !sigma || !rho || !innov || !wind) return fail(ERPL_ERR_INVALID, "NULL buffer");
  if (!base && (!mean_scale || !speed || !cdir || !sdir)) return fail(ERPL_ERR_INVALID, "NULL mean-wind inputs");
  const int nthr = host_threads(threads, n);
  run_threads(nthr, [&](int w) {
    LegacyRS rs;
    const int64_t lo = n * w / nthr, hi = n * (w + 1) / nthr;
    for (int64_t s = lo; s < hi; ++s) {
      rs.seed(seeds[s]);
      double* o = wind + s;   // element (i, c) at o[(i * 3 + c) * n]
      double pu, pv, pw;      // previous knot's values
      if (base) {             // environment.py:218-265
        pu = base[0] + (0.0 + sigma[0] * rs.next_gauss());
        pv = base[1] + (0.0 + sigma[0] * rs.next_gauss());
        pw = base[2] + (0.0 + (sigma[0] * 0.3) * rs.next_gauss());
        o[0] = pu; o[n] = pv; o[2 * n] = pw;
        for (int32_t i = 1; i < k; ++i) {
          const double* b0 = base + 3 * (i - 1);
          const double* b1 = base + 3 * i;
          const double tu = rho[i] * (pu - b0[0]) + (0.0 + innov[i] * rs.next_gauss());
          const double tv = rho[i] * (pv - b0[1]) + (0.0 + innov[i] * rs.next_gauss());
          const double tw = rho[i] * (pw - b0[2]) + (0.0 + (innov[i] * 0.3) * rs.next_gauss());
          pu = b1[0] + tu; pv = b1[1] + tv; pw = b1[2] + tw;
          o[(int64_t)(3 * i) * n] = pu; o[(int64_t)(3 * i + 1) * n] = pv; o[(int64_t)(3 * i + 2) * n] = pw;
        }
      } else {                // environment.py:125-200
        const double cd = cdir[s], sd = sdir[s], sp = speed[s];
        double m = sp * mean_scale[0];
        pu = m * cd + (0.0 + sigma[0] * rs.next_gauss());
        pv = m * sd + (0.0 + sigma[0] * rs.next_gauss());
        pw = 0.0 + (sigma[0] * 0.3) * rs.next_gauss();
        o[0] = pu; o[n] = pv; o[2 * n] = pw;
        for (int32_t i = 1; i < k; ++i) {
          const double m1 = sp * mean_scale[i];
          const double tu = rho[i] * (pu - m * cd) + (0.0 + innov[i] * rs.next_gauss());
          const double tv = rho[i] * (pv - m * sd) + (0.0 + innov[i] * rs.next_gauss());
          const double tw = rho[i] * pw + (0.0 + (innov[i] * 0.3) * rs.next_gauss());
          pu = m1 * cd + tu; pv = m1 * sd + tv; pw = tw;
          m = m1;
          o[(int64_t)(3 * i) * n] = pu; o[(int64_t)(3 * i + 1) * n] = pv; o[(int64_t)(3 * i + 2) * n] = pw;
        }
      }
    }
  });
  return ERPL_OK;
}

int erpl_mc_extract_histories(erpl_ctx* c, const erpl_batch* b, int64_t sample, const double* traj, int64_t m,
                              double time_offset, double* out, void* stream) {
  if (!c || !b || !traj || !out) return fail(ERPL_ERR_INVALID, "NULL argument");
  if (!c->has_cfg) return fail(ERPL_ERR_CONFIG, "erpl_mc_set_config has not been called");
  if (b->precision != ERPL_PREC_F64 && b->precision != ERPL_PREC_F64_FAST)
    return fail(ERPL_ERR_INVALID, "history extraction needs a batch with fp64 wind tables");
  if (sample < 0 || sample >= b->n || m < 0) return fail(ERPL_ERR_INVALID, "sample/m out of range");
  if (b->k_wind < 0 || b->k_wind > ERPL_MAX_WIND_KNOTS || (b->k_wind > 0 && (!b->alt_grid || !b->wind)))
    return fail(ERPL_ERR_INVALID, "bad wind arguments");
  if (m == 0) return ERPL_OK;
  HIP_TRY(hipSetDevice(c->device));
  const ErplTables& T = c->h_tables;
  ErplKArgs a;
  fill_common_args(c, b, a);
  a.summary = out; a.traj = const_cast<double*>(traj); a.traj_cap = m; a.n_traj = sample;
  int rc = erpl_launch_extract_f64(a, &T.s64, time_offset, stream);
  if (rc != 0) return fail(ERPL_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
  return ERPL_OK;
}

}  // extern "C"

namespace {
// AR(1) turbulence over the altitude knots + mean wind for n samples at once (environment.py:161-198 /
// :242-263 with caller-supplied standard normals): one thread per (component, sample), sequential over
// the k knots, every access coalesced along the sample index.  fp64 recursion whatever the output type.
template <typename OUT>
__global__ __launch_bounds__(256) void erpl_wind_ar1(const int64_t n, const int k, const double* __restrict__ g,
                                                     const double* __restrict__ sigma, const double* __restrict__ rho,
                                                     const double* __restrict__ innov, const double* __restrict__ base,
                                                     const double* __restrict__ scale, const double* __restrict__ mean_u,
                                                     const double* __restrict__ mean_v, OUT* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 3 * n) return;
  const int c = (int)(idx / n);
  const int64_t s = idx - (int64_t)c * n;
  const double comp = (c == 2) ? 0.3 : 1.0;   // vertical component: 30 % of the horizontal turbulence
  const double m = (c == 0) ? mean_u[s] : ((c == 1) ? mean_v[s] : 0.0);
  double t = 0.0;
  for (int i = 0; i < k; ++i) {
    const double z = g[(int64_t)(i * 3 + c) * n + s];
    t = (i == 0) ? (sigma[0] * comp) * z : rho[i] * t + (innov[i] * comp) * z;
    const double b = base ? base[i * 3 + c] : 0.0;
    out[(int64_t)(i * 3 + c) * n + s] = (OUT)((b + scale[i] * m) + t);
  }
}
}  // namespace

extern "C" {

int erpl_mc_synth_wind(erpl_ctx* c, int64_t n, int32_t k, const double* normals, const double* sigma, const double* rho,
                       const double* innov, const double* base, const double* scale, const double* mean_u,
                       const double* mean_v, void* wind, int32_t precision, void* stream) {
  if (!c) return fail(ERPL_ERR_INVALID, "NULL ctx");
  if (n < 0 || k < 0 || k > ERPL_MAX_WIND_KNOTS) return fail(ERPL_ERR_INVALID, "bad size (n=%lld k=%d)", (long long)n, k);
  if (n == 0 || k == 0) return ERPL_OK;
  if (!normals || !sigma || !rho || !innov || !scale || !mean_u || !mean_v || !wind) return fail(ERPL_ERR_INVALID, "NULL buffer");
  if (precision != ERPL_PREC_F64 && precision != ERPL_PREC_F32 && precision != ERPL_PREC_F64_FAST)
    return fail(ERPL_ERR_INVALID, "unknown precision %d", precision);
  HIP_TRY(hipSetDevice(c->device));
  const int block = 256;
  const int64_t grid = (3 * n + block - 1) / block;
  if (precision == ERPL_PREC_F32)
    hipLaunchKernelGGL(erpl_wind_ar1<float>, dim3((unsigned)grid), dim3(block), 0, (hipStream_t)stream, n, (int)k, normals, sigma,
                       rho, innov, base, scale, mean_u, mean_v, (float*)wind);
  else
    hipLaunchKernelGGL(erpl_wind_ar1<double>, dim3((unsigned)grid), dim3(block), 0, (hipStream_t)stream, n, (int)k, normals, sigma,
                       rho, innov, base, scale, mean_u, mean_v, (double*)wind);
  HIP_TRY(hipGetLastError());
  return ERPL_OK;
}

int erpl_mc_debug_eval(erpl_ctx* c, const erpl_batch* b, int what, int64_t m, const double* in, double* out, void* stream) {
  if (!c || !b || !in || !out) return fail(ERPL_ERR_INVALID, "NULL argument");
  if (!c->has_cfg) return fail(ERPL_ERR_CONFIG, "erpl_mc_set_config has not been called");
  if (what != ERPL_DBG_ATMOSPHERE && what != ERPL_DBG_AERO && what != ERPL_DBG_RHS) return fail(ERPL_ERR_INVALID, "unknown function %d", what);
  if (b->n < 1 || m < 0 || !b->rocket || !b->motor) return fail(ERPL_ERR_INVALID, "need at least one sample with parameters");
  if (b->k_wind < 0 || b->k_wind > ERPL_MAX_WIND_KNOTS || (b->k_wind > 0 && (!b->alt_grid || !b->wind)))
    return fail(ERPL_ERR_INVALID, "bad wind arguments");
  if (m == 0) return ERPL_OK;
  HIP_TRY(hipSetDevice(c->device));
  const ErplTables& T = c->h_tables;
  ErplKArgs a;
  fill_common_args(c, b, a);
  int rc;
  if (b->precision == ERPL_PREC_F64) rc = erpl_launch_debug_f64(a, &T.s64, what, m, in, out, stream);
  else if (b->precision == ERPL_PREC_F64_FAST) rc = erpl_launch_debug_f64f(a, &T.s64, what, m, in, out, stream);
  else if (b->precision == ERPL_PREC_F32) rc = erpl_launch_debug_f32(a, &T.s32, what, m, in, out, stream);
  else return fail(ERPL_ERR_INVALID, "unknown precision %d", b->precision);
  if (rc != 0) return fail(ERPL_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
  return ERPL_OK;
}

int erpl_mc_debug_counters(erpl_ctx* c, double* out16) {
  if (!c || !out16) return fail(ERPL_ERR_INVALID, "NULL argument");
  unsigned long long h[16];
  HIP_TRY(hipSetDevice(c->device));
  const ErplSlot& ls = c->slot[c->last_slot];
  if (ls.used) HIP_TRY(hipEventSynchronize(ls.done));
  HIP_TRY(hipMemcpy(h, ls.d_counters, sizeof(h), hipMemcpyDeviceToHost));
  for (int i = 0; i < 16; ++i) out16[i] = (double)h[i];
  return ERPL_OK;
}

int erpl_mc_last_stats(erpl_ctx* c, double* total_steps, double* wave_iterations) {
  if (!c) return fail(ERPL_ERR_INVALID, "NULL ctx");
  unsigned long long h[4] = {0, 0, 0, 0};
  HIP_TRY(hipSetDevice(c->device));
  const ErplSlot& ls = c->slot[c->last_slot];
  if (ls.used) HIP_TRY(hipEventSynchronize(ls.done));
  HIP_TRY(hipMemcpy(h, ls.d_counters, sizeof(h), hipMemcpyDeviceToHost));
  if (total_steps) *total_steps = (double)h[1];
  if (wave_iterations) *wave_iterations = (double)h[2];
  if (h[3] != 0ull) return fail(ERPL_ERR_HIP, "lane hand-over timed out: the results of the last batch are incomplete");
  return ERPL_OK;
}

/* Device counters of ONE submitted batch (its record in the ticket ring): waits for that batch alone. */
int erpl_mc_ticket_stats(erpl_ctx* c, int64_t ticket, double* total_steps, double* wave_iterations) {
  if (!c) return fail(ERPL_ERR_INVALID, "NULL ctx");
  HIP_TRY(hipSetDevice(c->device));
  const int ri = ring_index(c, ticket);
  if (ri < 0) return fail(ERPL_ERR_INVALID, "ticket %lld is not (or no longer) among the last %d submitted batches",
                          (long long)ticket, (int)ERPL_TICKET_RING);
  HIP_TRY(hipEventSynchronize(c->ring_done[ri]));
  if (total_steps) *total_steps = (double)c->ring_counters[4 * ri + 1];
  if (wave_iterations) *wave_iterations = (double)c->ring_counters[4 * ri + 2];
  if (c->ring_counters[4 * ri + 3] != 0ull) return report_incomplete(ticket, c->ring_counters[4 * ri + 3]);
  return ERPL_OK;
}

}  // extern "C"
