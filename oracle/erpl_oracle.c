/*
 * erpl_oracle.c — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (erpl_monte_carlo_sim_amd/) never does and has no CPU fallback.
 *
 * Plain C99, IEEE fp64, compiled with -ffp-contract=off, one scalar operation per reference
 * operation in the reference's evaluation order.  Every function cites the reference lines it
 * restates (paths relative to /root/reference/rocket_simulation/).  `x**2` on floats in the
 * reference is libm pow(x, 2.0) (probe: numpy/python scalar power is NOT x*x in 0.09 % of cases),
 * so pow() is used wherever the reference writes `**`.
 *
 * Parity pinning: checked in tests/test_oracle_golden.py against the tests/golden fixtures, which
 * were produced by importing the reference in the build container (oracle/gen_golden.py).
 * NumPy's own exp/arctan2/dot kernels differ from glibc's in the last ulp (probe: 5 % / 3 % /
 * 10 % of arguments), so the pinning tolerances are a few ulp on function KATs, 1e-9 relative on
 * healthy-flight scalars.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/erpl_mc.h"

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------- small helpers */

/* Python's builtin max(a, b): returns a unless b > a.  min(a, b): returns a unless b < a. */
static double py_max(double a, double b) { return (b > a) ? b : a; }
static double py_min(double a, double b) { return (b < a) ? b : a; }

static double sq(double x) { return pow(x, 2.0); }

/* np.interp for a scalar x (numpy/_core/src/multiarray/compiled_base.c arr_interp, numpy 2.2):
 * NaN in -> NaN out; x > xp[n-1] -> fp[n-1]; x < xp[0] -> fp[0]; exact knot -> fp[j];
 * otherwise slope*(x - xp[j]) + fp[j] with slope = (fp[j+1]-fp[j])/(xp[j+1]-xp[j]), retried from
 * the right knot if that is NaN.  utils.py:147-149. */
double erpl_oracle_interp(double x, const double* xp, const double* fp, int n) {
  if (isnan(x)) return x;
  if (n <= 0) return NAN;
  if (x > xp[n - 1]) return fp[n - 1];
  if (x < xp[0]) return fp[0];
  int lo = 0, hi = n; /* find j: xp[j] <= x < xp[j+1] */
  while (hi - lo > 1) {
    int mid = (lo + hi) / 2;
    if (x >= xp[mid]) lo = mid; else hi = mid;
  }
  int j = lo;
  if (j == n - 1) return fp[j];
  if (xp[j] == x) return fp[j];
  double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
  double r = slope * (x - xp[j]) + fp[j];
  if (isnan(r)) {
    r = slope * (x - xp[j + 1]) + fp[j + 1];
    if (isnan(r) && fp[j] == fp[j + 1]) r = fp[j];
  }
  return r;
}

/* np.linalg.norm of a short vector = sqrt(dot(x, x)), accumulated left to right. */
static double norm3(const double* v) { return sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }
static double norm4(const double* v) {
  return sqrt(((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]) + v[3] * v[3]);
}
static double dot3(const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

/* utils.py:76-82 */
static void normalize_quaternion(const double* q, double* out) {
  double n = norm4(q);
  if (n > 1e-12) {
    for (int i = 0; i < 4; ++i) out[i] = q[i] / n;
  } else {
    out[0] = 1.0; out[1] = 0.0; out[2] = 0.0; out[3] = 0.0;
  }
}

/* utils.py:100-111 (normalises again) ; R is row-major 3x3, body -> inertial */
static void quaternion_to_rotation_matrix(const double* q_in, double R[9]) {
  double q[4];
  normalize_quaternion(q_in, q);
  double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (sq(y) + sq(z)); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (sq(x) + sq(z)); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (sq(x) + sq(y));
}

/* ---------------------------------------------------------------- L1 physical models */

/* environment.py:26-103.  out = {temperature, pressure, density, speed_of_sound} */
void erpl_oracle_atmosphere(const erpl_config* c, double altitude, double out[4]) {
  double temperature, pressure;
  const double g = c->gravity, Rg = c->gas_constant;
  if (altitude <= c->troposphere_height) {
    temperature = c->sea_level_temperature - c->temperature_lapse_rate * altitude;
    pressure = c->sea_level_pressure *
               pow(temperature / c->sea_level_temperature, g / (Rg * c->temperature_lapse_rate));
  } else if (altitude <= c->stratosphere_height) {
    temperature = c->stratosphere_temp;
    double pressure_11km = c->sea_level_pressure *
        pow(c->stratosphere_temp / c->sea_level_temperature, g / (Rg * c->temperature_lapse_rate));
    pressure = pressure_11km * exp(-g * (altitude - c->troposphere_height) / (Rg * temperature));
  } else {
    if (altitude <= 32000.0) {
      temperature = c->stratosphere_temp + 0.001 * (altitude - c->stratosphere_height);
      temperature = py_min(temperature, 228.65);
      double pressure_20km = c->sea_level_pressure *
          pow(c->stratosphere_temp / c->sea_level_temperature, g / (Rg * c->temperature_lapse_rate));
      pressure_20km *= exp(-g * (c->stratosphere_height - c->troposphere_height) /
                           (Rg * c->stratosphere_temp));
      if (altitude <= 25000.0) {
        pressure = pressure_20km *
                   exp(-g * (altitude - c->stratosphere_height) / (Rg * c->stratosphere_temp));
      } else {
        double pressure_25km = pressure_20km * exp(-g * 5000.0 / (Rg * c->stratosphere_temp));
        double temp_gradient = 0.0028;
        double temp_25km = c->stratosphere_temp;
        pressure = pressure_25km * pow(temperature / temp_25km, g / (Rg * temp_gradient));
      }
    } else {
      temperature = 228.65 - 0.0028 * (altitude - 32000.0);
      temperature = py_max(temperature, 180.0);
      double scale_height = Rg * temperature / g;
      double pressure_32km = 868.02;
      pressure = pressure_32km * exp(-(altitude - 32000.0) / scale_height);
    }
  }
  out[0] = temperature;
  out[1] = pressure;
  out[2] = pressure / (Rg * temperature);
  out[3] = sqrt(1.4 * Rg * temperature); /* self.gamma = 1.4 (environment.py:19, :96) */
}

/* environment.py:105-108 */
double erpl_oracle_gravity(const erpl_config* c, double altitude) {
  double earth_radius = 6.371e6;
  return c->gravity * sq(earth_radius / (earth_radius + altitude));
}

/* rocket.py:110-136.  out = {mass, center_of_mass, Ixx, Iyy, Izz} */
void erpl_oracle_mass_props(const erpl_config* c, double dry_mass, double propellant_mass,
                            double pf, double out[5]) {
  double current_propellant = propellant_mass * pf;
  double total_mass = dry_mass + current_propellant;
  double propellant_cg = c->center_of_mass_dry - 0.5;
  double current_cg = (dry_mass * c->center_of_mass_dry + current_propellant * propellant_cg) / total_mass;
  double propellant_Ixx = current_propellant * sq(c->diameter / 4);
  double propellant_Iyy = current_propellant * (4.0 / 12 + sq(propellant_cg - current_cg));
  out[0] = total_mass;
  out[1] = current_cg;
  out[2] = c->Ixx_dry + propellant_Ixx;
  out[3] = c->Iyy_dry + propellant_Iyy;
  out[4] = out[3];
}

/* rocket.py:138-218 (+ get_dynamic_cp :105-108).
 * out = {cd, cl, cy, cpitch(cm), cyaw, cp_current, cn} ; croll = 0 */
void erpl_oracle_aero(const erpl_config* c, double mach, double alpha, double beta, double cg,
                      int power_on, double out[7]) {
  double cd0 = erpl_oracle_interp(mach, c->cd_mach, c->cd0, c->n_cd);
  double cda = erpl_oracle_interp(mach, c->cd_mach, c->cda, c->n_cd);
  double cd = cd0 + cda * sq(alpha);
  if (!power_on) cd *= c->power_off_drag_factor;
  double stall_angle = 15.0 * (M_PI / 180.0); /* np.radians */
  double max_angle = 45.0 * (M_PI / 180.0);
  double abs_alpha = fabs(alpha);
  double cr = c->fin_root_chord, ct = c->fin_tip_chord, s = c->fin_span;
  double fin_area = 0.5 * (cr + ct) * s;
  double AR = (fin_area > 0) ? 2 * sq(s) / fin_area : 0.0;
  double beta_m = (mach < 1) ? sqrt(fabs(1.0 - sq(mach))) : sqrt(fabs(sq(mach) - 1));
  double cosl = cos(c->fin_sweep_angle);
  double denom = 2 + sqrt(4 + sq(AR * beta_m / py_max(cosl, 1e-6)));
  double cl_alpha = (2 * M_PI * AR / denom) * cosl;
  double cl = cl_alpha * alpha;
  double stall_factor = 1.0;
  double sgn = (alpha > 0) ? 1.0 : ((alpha < 0) ? -1.0 : alpha); /* np.sign; NaN -> NaN, 0 -> 0 */
  if (abs_alpha > stall_angle) {
    stall_factor = py_max(0.0, 1.0 - (abs_alpha - stall_angle) / (max_angle - stall_angle));
    cl = cl_alpha * stall_angle * stall_factor * sgn;
    cd *= 1.0 + 0.5 * (abs_alpha - stall_angle) / (max_angle - stall_angle);
  }
  double cp_current = c->cp_location + erpl_oracle_interp(mach, c->cp_mach, c->cp_shift, c->n_cp);
  double static_margin = cp_current - cg;
  double cm_alpha = -cl_alpha * static_margin;
  double cm = cm_alpha * alpha;
  double cy = cl_alpha * beta;
  double cn = cl_alpha * alpha;
  if (abs_alpha > stall_angle) {
    cy *= stall_factor;
    cn = cl_alpha * stall_angle * stall_factor * sgn;
  }
  double cyaw = -cl_alpha * static_margin * beta;
  out[0] = cd; out[1] = cl; out[2] = cy; out[3] = cm; out[4] = cyaw; out[5] = cp_current; out[6] = cn;
}

/* one sample's view of the batch */
typedef struct sample {
  const erpl_config* c;
  double dry_mass, propellant_mass;
  double thrust, nozzle_exit_area, mass_flow_rate, burn_time;
  int k_wind;
  const double* alt_grid;
  const double* wind;   /* element (k, comp) at wind[(k*3+comp)*wstride] */
  int64_t wstride;
  double curve_thrust[ERPL_MAX_CURVE_KNOTS]; /* per-sample scaled curve (motor.py:105) */
} sample;

/* environment.py:267-276 ; zero when no profile (simulator.py:333-338) */
static void wind_at(const sample* s, double altitude, double w[3]) {
  if (s->k_wind <= 0) { w[0] = w[1] = w[2] = 0.0; return; }
  double col[ERPL_MAX_WIND_KNOTS];
  for (int comp = 0; comp < 3; ++comp) {
    for (int k = 0; k < s->k_wind; ++k) col[k] = s->wind[(int64_t)(k * 3 + comp) * s->wstride];
    w[comp] = erpl_oracle_interp(altitude, s->alt_grid, col, s->k_wind);
  }
}

/* motor.py:152-156 (liquid) / :54-76 (solid) */
static double motor_thrust(const sample* s, double time, double ambient_pressure) {
  if (time < 0 || time > s->burn_time) return 0.0;
  if (s->c->motor_kind == ERPL_MOTOR_SOLID) {
    double thrust_sl = erpl_oracle_interp(time, s->c->curve_time, s->curve_thrust, s->c->n_curve);
    double pressure_correction = s->nozzle_exit_area * (101325.0 - ambient_pressure);
    return thrust_sl + pressure_correction;
  }
  return s->thrust - s->nozzle_exit_area * ambient_pressure;
}

/* motor.py:78-84 / :158-161 */
static double motor_mass_flow(const sample* s, double time) {
  if (time < 0 || time > s->burn_time) return 0.0;
  return s->mass_flow_rate;
}

/* motor.py:86-93 / :163-169 */
static double motor_propellant_remaining(const sample* s, double time) {
  if (time <= 0) return 1.0;
  if (time >= s->burn_time) return 0.0;
  return py_max(0.0, 1.0 - time / s->burn_time);
}

/* utils.py:152-157 */
static double mach_number(const double* v, double temperature) {
  double speed_of_sound = sqrt(1.4 * 287.053 * temperature);
  return norm3(v) / speed_of_sound;
}
/* utils.py:160-164 */
static double angle_of_attack(const double* vb) {
  if (fabs(vb[0]) < 1e-6 && fabs(vb[2]) < 1e-6) return 0.0;
  return atan2(vb[2], vb[0]);
}
/* utils.py:167-172 */
static double sideslip_angle(const double* vb) {
  double V_xz = sqrt(sq(vb[0]) + sq(vb[2]));
  if (V_xz < 1e-6) return 0.0;
  return atan2(vb[1], V_xz);
}

/* simulator.py:295-460.  `chute` is FlightSimulator.parachute_deployed (latched in here). */
static void rocket_dynamics(const sample* s, double t, const double* state, int* chute, double* sd) {
  const erpl_config* c = s->c;
  const double* position = state;
  const double* velocity = state + 3;
  const double* angular_velocity = state + 10;
  double pf = py_max(0.0, state[13]);
  double q[4];
  normalize_quaternion(state + 6, q);
  double mp[5];
  erpl_oracle_mass_props(c, s->dry_mass, s->propellant_mass, pf, mp);
  double mass = mp[0];
  if (mass < s->dry_mass) {
    mass = s->dry_mass;
    erpl_oracle_mass_props(c, s->dry_mass, s->propellant_mass, 0.0, mp);
  }
  double Ixx = mp[2], Iyy = mp[3], Izz = mp[4];
  double R[9];
  quaternion_to_rotation_matrix(q, R);
  double altitude = position[2];
  double atm[4];
  erpl_oracle_atmosphere(c, altitude, atm);
  double density = atm[2], temperature = atm[0];
  double wind[3];
  wind_at(s, altitude, wind);
  double vrel[3] = {velocity[0] - wind[0], velocity[1] - wind[1], velocity[2] - wind[2]};
  double vb[3];
  for (int i = 0; i < 3; ++i) vb[i] = (R[i] * vrel[0] + R[3 + i] * vrel[1]) + R[6 + i] * vrel[2];
  double mach = mach_number(vrel, temperature);
  double alpha = angle_of_attack(vb);
  double beta = sideslip_angle(vb);
  double q_dynamic = 0.5 * density * sq(norm3(vrel));
  double fb[3] = {0, 0, 0}, mb[3] = {0, 0, 0};
  double thrust = 0.0;
  if (pf > 0 && t <= s->burn_time) thrust = motor_thrust(s, t, atm[1]);
  fb[0] += thrust;
  if (!*chute && altitude <= c->parachute_deployment_altitude && velocity[2] < 0) *chute = 1;
  if (*chute) {
    double rel_speed = norm3(vb);
    if (rel_speed > 0) {
      double drag = 0.5 * density * sq(rel_speed) * c->parachute_cd;
      drag *= c->parachute_area;
      for (int i = 0; i < 3; ++i) fb[i] += -drag * vb[i] / rel_speed;
    }
  } else if (q_dynamic > 0) {
    double co[7];
    erpl_oracle_aero(c, mach, alpha, beta, mp[1], pf > 0, co);
    double drag = q_dynamic * co[0] * c->reference_area;
    double lift = q_dynamic * co[1] * c->reference_area;
    double side = q_dynamic * co[2] * c->reference_area;
    double ca = cos(alpha), sa = sin(alpha), cb = cos(beta), sb = sin(beta);
    double Rw[9] = {ca * cb, -sb, sa * cb, ca * sb, cb, sa * sb, -sa, 0.0, ca};
    double fw[3] = {-drag, -side, -lift};
    for (int i = 0; i < 3; ++i) fb[i] += (Rw[3 * i] * fw[0] + Rw[3 * i + 1] * fw[1]) + Rw[3 * i + 2] * fw[2];
    mb[0] += q_dynamic * 0.0 * c->reference_area * c->reference_diameter;
    mb[1] += q_dynamic * co[3] * c->reference_area * c->reference_diameter;
    mb[2] += q_dynamic * co[4] * c->reference_area * c->reference_diameter;
  }
  mb[1] += -c->pitch_damping * angular_velocity[1];
  mb[2] += -c->yaw_damping * angular_velocity[2];
  double fi[3];
  for (int i = 0; i < 3; ++i) fi[i] = (R[3 * i] * fb[0] + R[3 * i + 1] * fb[1]) + R[3 * i + 2] * fb[2];
  double gravity = erpl_oracle_gravity(c, altitude);
  fi[2] -= mass * gravity;
  double aa[3] = {0, 0, 0};
  if (Ixx > 0) aa[0] = (mb[0] - (Izz - Iyy) * angular_velocity[1] * angular_velocity[2]) / Ixx;
  if (Iyy > 0) aa[1] = (mb[1] - (Ixx - Izz) * angular_velocity[2] * angular_velocity[0]) / Iyy;
  if (Izz > 0) aa[2] = (mb[2] - (Iyy - Ixx) * angular_velocity[0] * angular_velocity[1]) / Izz;
  /* utils.py:114-121 with q = normalised quaternion, omega_q = (0, wx, wy, wz) */
  double w1 = q[0], x1 = q[1], y1 = q[2], z1 = q[3];
  double w2 = 0.0, x2 = angular_velocity[0], y2 = angular_velocity[1], z2 = angular_velocity[2];
  double qm[4] = {w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                  w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2};
  double norm_error = (((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]) - 1.0;
  double pf_rate = 0.0;
  if (pf > 0 && t <= s->burn_time) {
    double mass_flow = motor_mass_flow(s, t);
    pf_rate = -mass_flow / s->propellant_mass;
    double remaining_time = (pf_rate != 0) ? pf / fabs(pf_rate) : INFINITY;
    if (remaining_time < 0.01) pf_rate = -pf / 0.01;
  }
  for (int i = 0; i < 3; ++i) sd[i] = velocity[i];
  for (int i = 0; i < 3; ++i) sd[3 + i] = fi[i] / mass;
  for (int i = 0; i < 4; ++i) sd[6 + i] = 0.5 * qm[i] - 0.5 * norm_error * q[i];
  for (int i = 0; i < 3; ++i) sd[10 + i] = aa[i];
  sd[13] = pf_rate;
}

/* simulator.py:42-125.  state[14] updated in place; returns rail-exit time; info = {speed, aoa, sideslip} */
static double simulate_launch_rail(const sample* s, double* state, double info[3]) {
  const erpl_config* c = s->c;
  double position[3] = {state[0], state[1], state[2]};
  double velocity[3] = {state[3], state[4], state[5]};
  const double* quaternion = state + 6;
  double prop_frac = state[13];
  double R[9];
  quaternion_to_rotation_matrix(quaternion, R);
  double direction[3] = {R[0], R[3], R[6]};
  double distance = 0.0, t = 0.0, dt = c->dt_initial;
  while (distance < c->rail_length && t < s->burn_time) {
    double mp[5];
    erpl_oracle_mass_props(c, s->dry_mass, s->propellant_mass, prop_frac, mp);
    double mass = mp[0];
    double atm[4];
    erpl_oracle_atmosphere(c, position[2], atm);
    double wind[3];
    wind_at(s, position[2], wind);
    double speed = dot3(velocity, direction);
    double rel_vel[3];
    for (int i = 0; i < 3; ++i) rel_vel[i] = direction[i] * speed - wind[i];
    double rel_speed = dot3(rel_vel, direction);
    double mach = mach_number(rel_vel, atm[0]);
    double co[7];
    erpl_oracle_aero(c, mach, 0.0, 0.0, mp[1], 1, co);
    double drag = 0.5 * atm[2] * sq(rel_speed) * co[0] * c->reference_area;
    double thrust = motor_thrust(s, t, atm[1]);
    double gravity = erpl_oracle_gravity(c, position[2]);
    double accel = (thrust - mass * gravity - drag) / mass;
    speed += accel * dt;
    for (int i = 0; i < 3; ++i) position[i] += direction[i] * speed * dt;
    distance += speed * dt;
    for (int i = 0; i < 3; ++i) velocity[i] = direction[i] * speed;
    t += dt;
    prop_frac = motor_propellant_remaining(s, t);
  }
  for (int i = 0; i < 3; ++i) { state[i] = position[i]; state[3 + i] = velocity[i]; }
  state[13] = prop_frac;
  double wind[3];
  wind_at(s, position[2], wind);
  double vrel[3] = {velocity[0] - wind[0], velocity[1] - wind[1], velocity[2] - wind[2]};
  double vb[3];
  for (int i = 0; i < 3; ++i) vb[i] = (R[i] * vrel[0] + R[3 + i] * vrel[1]) + R[6 + i] * vrel[2];
  info[0] = norm3(velocity);
  info[1] = angle_of_attack(vb);
  info[2] = sideslip_angle(vb);
  return t;
}

typedef struct recorder {
  double* buf; int64_t cap, stride, len;
} recorder;

static void record(recorder* r, int64_t step, double t, const double* state, int final) {
  if (!r || !r->buf || r->cap <= 0) return;
  if (!(final || (r->stride > 0 && step % r->stride == 0))) return;
  int64_t slot = r->len;
  if (slot >= r->cap) { if (!final) return; slot = r->cap - 1; } else r->len++;
  double* p = r->buf + slot * ERPL_TRAJ_DIM;
  p[0] = t;
  memcpy(p + 1, state, sizeof(double) * ERPL_STATE_DIM);
}

/* simulator.py:127-293 reduced to the scalar results of :488-494 and :579-582 */
static void simulate_flight(const sample* s, const double* ic, int flags, double* summary,
                            int64_t n, int64_t i, int32_t* status, recorder* rec) {
  const erpl_config* c = s->c;
  double state[ERPL_STATE_DIM];
  memcpy(state, ic, sizeof(double) * ERPL_IC_DIM);
  state[13] = 1.0;
  double info[3];
  double rail_time = simulate_launch_rail(s, state, info);
  int chute = 0;
  double dt = py_min(c->dt_initial, 0.005);
  double t = rail_time;
  /* running argmax over altitudes[0..]; np.argmax returns the first NaN if any */
  double apogee = state[2], apogee_t = t;
  int nan_seen = isnan(apogee);
  double first_apogee = apogee, first_apogee_t = t;
  double max_speed = norm3(state + 3);
  if (isnan(max_speed)) max_speed = 0.0;
  int apogee_detected = 0;
  double apogee_time = 0.0, max_coast_time = 0.0;
  int end = ERPL_END_MAX_TIME;
  int64_t steps = 0;
  record(rec, 0, t, state, 0);
  while (t < c->max_time) {
    double k1[14], k2[14], k3[14], k4[14], y[14];
    rocket_dynamics(s, t, state, &chute, k1);
    for (int j = 0; j < 14; ++j) y[j] = state[j] + 0.5 * dt * k1[j];
    rocket_dynamics(s, t + 0.5 * dt, y, &chute, k2);
    for (int j = 0; j < 14; ++j) y[j] = state[j] + 0.5 * dt * k2[j];
    rocket_dynamics(s, t + 0.5 * dt, y, &chute, k3);
    for (int j = 0; j < 14; ++j) y[j] = state[j] + dt * k3[j];
    rocket_dynamics(s, t + dt, y, &chute, k4);
    for (int j = 0; j < 14; ++j) state[j] += (dt / 6.0) * (((k1[j] + 2 * k2[j]) + 2 * k3[j]) + k4[j]);
    double qn[4];
    normalize_quaternion(state + 6, qn);
    memcpy(state + 6, qn, sizeof(qn));
    t += dt;
    steps++;
    double altitude = state[2], vertical_velocity = state[5];
    if (!nan_seen) {
      if (isnan(altitude)) { nan_seen = 1; apogee = altitude; apogee_t = t; }
      else if (altitude > apogee) { apogee = altitude; apogee_t = t; }
    }
    double spd = norm3(state + 3);
    if (spd > max_speed) max_speed = spd;
    int stop = 0;
    if (altitude <= 0.5 && vertical_velocity <= 0) { end = ERPL_END_GROUND; stop = 1; }
    else if (altitude > 100000.0) { end = ERPL_END_ALTITUDE; stop = 1; }
    else {
      if (altitude > 1000.0 && vertical_velocity < 0 && !apogee_detected) {
        apogee_detected = 1;
        apogee_time = t;
        first_apogee = apogee; first_apogee_t = apogee_t;
        if (altitude > 50000.0) max_coast_time = 60.0;
        else if (altitude > 25000.0) max_coast_time = 120.0;
        else max_coast_time = 300.0;
        if (flags & ERPL_FLAG_STOP_AT_APOGEE) { end = ERPL_END_APOGEE; stop = 1; }
      }
      if (!stop && apogee_detected && altitude > 25000.0) {
        double coast_time = t - apogee_time;
        if (coast_time > max_coast_time) { end = ERPL_END_COAST; stop = 1; }
      }
    }
    record(rec, steps, t, state, stop || !(t < c->max_time));
    if (stop) break;
  }
  if (!apogee_detected) { first_apogee = apogee; first_apogee_t = apogee_t; }
  summary[ERPL_SUM_APOGEE_ALT * n + i] = apogee;
  summary[ERPL_SUM_APOGEE_TIME * n + i] = apogee_t - rail_time;
  summary[ERPL_SUM_FIRST_APOGEE_ALT * n + i] = first_apogee;
  summary[ERPL_SUM_FIRST_APOGEE_TIME * n + i] = first_apogee_t - rail_time;
  summary[ERPL_SUM_RANGE * n + i] = sqrt(sq(state[0]) + sq(state[1]));
  summary[ERPL_SUM_FLIGHT_TIME * n + i] = t - rail_time;
  summary[ERPL_SUM_RAIL_EXIT_TIME * n + i] = rail_time;
  summary[ERPL_SUM_RAIL_EXIT_SPEED * n + i] = info[0];
  summary[ERPL_SUM_IMPACT_X * n + i] = state[0];
  summary[ERPL_SUM_IMPACT_Y * n + i] = state[1];
  summary[ERPL_SUM_IMPACT_Z * n + i] = state[2];
  summary[ERPL_SUM_STEPS * n + i] = (double)steps;
  summary[ERPL_SUM_RAIL_EXIT_AOA * n + i] = info[1];
  summary[ERPL_SUM_RAIL_EXIT_SIDESLIP * n + i] = info[2];
  summary[ERPL_SUM_FINAL_VZ * n + i] = state[5];
  summary[ERPL_SUM_MAX_SPEED * n + i] = max_speed;
  status[i] = end | (apogee_detected ? ERPL_ST_APOGEE_LATCHED : 0) | (chute ? ERPL_ST_CHUTE : 0) |
              (nan_seen ? ERPL_ST_NAN : 0);
}

static void load_sample(const erpl_config* c, const erpl_batch* b, int64_t i, sample* s, double* ic) {
  int64_t n = b->n;
  s->c = c;
  for (int k = 0; k < ERPL_IC_DIM; ++k) ic[k] = b->ic[k * n + i];
  s->dry_mass = b->rocket[0 * n + i];
  s->propellant_mass = b->rocket[1 * n + i];
  s->thrust = b->motor[0 * n + i];
  s->nozzle_exit_area = b->motor[1 * n + i];
  s->mass_flow_rate = b->motor[2 * n + i];
  s->burn_time = b->motor[3 * n + i];
  s->k_wind = b->k_wind;
  s->alt_grid = b->alt_grid;
  s->wind = (b->k_wind > 0) ? ((const double*)b->wind) + i : NULL;
  s->wstride = n;
  for (int k = 0; k < c->n_curve; ++k) s->curve_thrust[k] = c->curve_thrust[k] * s->thrust;
}

/* ---------------------------------------------------------------- exported entry points */

/* HOST pointers in batch/out; wind is always double here.  n_threads <= 0 -> all cores. */
int erpl_oracle_run_batch(const erpl_config* c, const erpl_batch* b, const erpl_out* o, int n_threads) {
  if (!c || !b || !o || !o->summary || !o->status) return ERPL_ERR_INVALID;
  if (b->k_wind > ERPL_MAX_WIND_KNOTS) return ERPL_ERR_INVALID;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
  int64_t n = b->n;
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t i = 0; i < n; ++i) {
    sample s;
    double ic[ERPL_IC_DIM];
    load_sample(c, b, i, &s, ic);
    recorder rec = {0, 0, 0, 0};
    int64_t slot = -1;
    for (int64_t m = 0; m < o->n_traj; ++m) if (o->traj_ids[m] == i) slot = m;
    if (slot >= 0 && o->traj) {
      rec.buf = o->traj + slot * o->traj_cap * ERPL_TRAJ_DIM;
      rec.cap = o->traj_cap; rec.stride = o->traj_stride;
    }
    simulate_flight(&s, ic, b->flags, o->summary, n, i, o->status, slot >= 0 ? &rec : NULL);
    if (slot >= 0 && o->traj_len) o->traj_len[slot] = rec.len;
  }
  return ERPL_OK;
}

int erpl_oracle_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* single RHS evaluation for the function-level KATs: sample 0 of the batch supplies parameters */
int erpl_oracle_rhs(const erpl_config* c, const erpl_batch* b, double t, const double* state,
                    int* chute, double* deriv) {
  sample s;
  double ic[ERPL_IC_DIM];
  load_sample(c, b, 0, &s, ic);
  rocket_dynamics(&s, t, state, chute, deriv);
  return ERPL_OK;
}

int erpl_oracle_wind(const erpl_batch* b, double altitude, double w[3]) {
  sample s;
  memset(&s, 0, sizeof(s));
  s.k_wind = b->k_wind; s.alt_grid = b->alt_grid; s.wind = (const double*)b->wind; s.wstride = b->n;
  wind_at(&s, altitude, w);
  return ERPL_OK;
}

/* Per-step diagnostic histories of FlightSimulator._extract_results (simulator.py:496-552) for the
 * m stored records traj[m][15] = {absolute time, 14 state} of sample 0 of the batch.
 * out[m][17] = euler(3), center_of_mass, mass, Ixx, Iyy, Izz, thrust, drag, cd, cl, cm,
 *              cp_location_dynamic, stability_margin, angle_of_attack, sideslip_angle.
 * Quirk kept: the thrust history is evaluated at the rail-shifted time (simulator.py:543). */
int erpl_oracle_extract(const erpl_config* c, const erpl_batch* b, int64_t m, const double* traj,
                        double time_offset, double* out) {
  sample s;
  double ic[ERPL_IC_DIM];
  load_sample(c, b, 0, &s, ic);
  for (int64_t r = 0; r < m; ++r) {
    const double* rec = traj + r * ERPL_TRAJ_DIM;
    const double time_shifted = rec[0] - time_offset;
    const double* st = rec + 1;
    double* o = out + r * 17;
    /* quaternion_to_euler on the stored quaternion (utils.py:139-144 via :46-70) */
    double w = st[6], x = st[7], y = st[8], z = st[9];
    double sinr_cosp = 2 * (w * x + y * z), cosr_cosp = 1 - 2 * (x * x + y * y);
    o[0] = atan2(sinr_cosp, cosr_cosp);
    double sinp = 2 * (w * y - z * x);
    o[1] = (fabs(sinp) >= 1) ? copysign(M_PI / 2, sinp) : asin(sinp);
    double siny_cosp = 2 * (w * z + x * y), cosy_cosp = 1 - 2 * (y * y + z * z);
    o[2] = atan2(siny_cosp, cosy_cosp);
    double mp[5];
    erpl_oracle_mass_props(c, s.dry_mass, s.propellant_mass, st[13], mp);
    o[3] = mp[1]; o[4] = mp[0]; o[5] = mp[2]; o[6] = mp[3]; o[7] = mp[4];
    double atm[4];
    erpl_oracle_atmosphere(c, st[2], atm);
    double wind[3];
    wind_at(&s, st[2], wind);
    double vrel[3] = {st[3] - wind[0], st[4] - wind[1], st[5] - wind[2]};
    double R[9];
    quaternion_to_rotation_matrix(st + 6, R);
    double vb[3];
    for (int i = 0; i < 3; ++i) vb[i] = (R[i] * vrel[0] + R[3 + i] * vrel[1]) + R[6 + i] * vrel[2];
    double mach = mach_number(vrel, atm[0]);
    double aoa = angle_of_attack(vb), beta = sideslip_angle(vb);
    double cp_val = c->cp_location + erpl_oracle_interp(mach, c->cp_mach, c->cp_shift, c->n_cp);
    double co[7];
    erpl_oracle_aero(c, mach, aoa, beta, mp[1], st[13] > 0, co);
    double q_dyn = 0.5 * atm[2] * sq(norm3(vrel));
    o[8] = motor_thrust(&s, time_shifted, atm[1]);
    o[9] = q_dyn * co[0] * c->reference_area;
    o[10] = co[0]; o[11] = co[1]; o[12] = co[3];
    o[13] = cp_val;
    o[14] = (cp_val - mp[1]) / c->reference_diameter;
    o[15] = aoa; o[16] = beta;
  }
  return ERPL_OK;
}

/* motor KATs: out = {thrust, mass_flow, propellant_remaining} for sample 0 */
int erpl_oracle_motor(const erpl_config* c, const erpl_batch* b, double t, double pressure, double out[3]) {
  sample s;
  double ic[ERPL_IC_DIM];
  load_sample(c, b, 0, &s, ic);
  out[0] = motor_thrust(&s, t, pressure);
  out[1] = motor_mass_flow(&s, t);
  out[2] = motor_propellant_remaining(&s, t);
  return ERPL_OK;
}
