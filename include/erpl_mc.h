/*
 * erpl_mc.h — C ABI of the MI355X-native Monte Carlo 6-DOF trajectory engine.
 *
 * The reference (smcconoughey/erpl_monte_carlo_sim) has NO native/FFI layer: its boundary for this
 * path is two Python methods.  Each entry point below names the reference interface it replaces
 * (file:line relative to the reference repo) and is what a ctypes/cffi stub added to the
 * reference would bind.
 *
 *   erpl_mc_run_batch   <-  N x MonteCarloAnalyzer._run_single_simulation -> FlightSimulator.
 *                           simulate_flight   (monte_carlo.py:225-306, simulator.py:127-293):
 *                           launch rail (simulator.py:42-125) + RK4 loop (simulator.py:208-264)
 *                           over _rocket_dynamics (simulator.py:295-460) + scalar results
 *                           (simulator.py:488-494, :579-582)
 *   erpl_config         <-  the attributes of Rocket (rocket.py:14-66), Solid/LiquidMotor
 *                           (motor.py:15-52, :131-150), StandardAtmosphere (environment.py:13-24)
 *                           and FlightSimulator (simulator.py:19-40) that the path reads
 *   erpl_batch          <-  the per-sample objects _run_single_simulation builds
 *                           (monte_carlo.py:228-288): perturbed IC, rocket masses, motor, wind
 *
 * Conventions: plain C, no C++ types, no exceptions across the boundary.  Every function returns
 * 0 on success or a negative erpl_status; erpl_mc_last_error() gives a thread-local message.
 * All per-sample arrays are structure-of-arrays: element (c, i) of a [C][n] array is at c*n + i.
 * Device buffers are CALLER-OWNED (e.g. torch tensors); the library never frees or keeps them
 * past completion of the work it enqueued on the given stream.
 */
#ifndef ERPL_MC_H
#define ERPL_MC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ERPL_MC_ABI_VERSION 3

#define ERPL_STATE_DIM 14   /* x y z vx vy vz q0(w) q1 q2 q3 wx wy wz propellant_fraction (simulator.py:130) */
#define ERPL_IC_DIM 13      /* the same without propellant_fraction (always 1.0 at ignition, simulator.py:161) */
#define ERPL_ROCKET_DIM 2   /* dry_mass, propellant_mass (monte_carlo.py:315-316) */
#define ERPL_MOTOR_DIM 4    /* thrust, nozzle_exit_area, mass_flow_rate, burn_time */
#define ERPL_SUMMARY_DIM 16
#define ERPL_TRAJ_DIM 15    /* absolute time + 14 state */
#define ERPL_MAX_MACH_KNOTS 16
#define ERPL_MAX_CURVE_KNOTS 32
#define ERPL_MAX_WIND_KNOTS 1024

typedef enum erpl_status {
  ERPL_OK = 0,
  ERPL_ERR_INVALID = -1,   /* bad argument / shape / non-finite table */
  ERPL_ERR_HIP = -2,       /* a HIP runtime call failed */
  ERPL_ERR_NO_DEVICE = -3, /* no usable GPU: the product path never falls back to a CPU */
  ERPL_ERR_CONFIG = -4,    /* run_batch before set_config */
  ERPL_ERR_INCOMPLETE = -5 /* a lane hand-over of a finished batch timed out: the samples whose status word still
                              carries ERPL_ST_INCOMPLETE were not integrated (erpl_mc_set_adopt) */
} erpl_status;

enum { ERPL_MOTOR_LIQUID = 0, ERPL_MOTOR_SOLID = 1 };
/* erpl_batch.precision: which build of the kernels integrates the batch.
 *   ERPL_PREC_F64       fp64 in the reference's operation order (IEEE division, libm-grade pow/exp/atan2/sin/cos,
 *                       no FMA contraction): the correctness gate, tracks the CPU reference to ~1e-13 per call
 *   ERPL_PREC_F64_FAST  fp64 arithmetic on the short formulation of the fp32 kernel (one reciprocal per
 *                       denominator, interval-record atmosphere, FMA) for as long as a sample is of physical size;
 *                       a sample whose speed passes 1e6 m/s (the reference's model blows up on every sample with
 *                       sideslip) is finished by the ERPL_PREC_F64 kernel in a sweep launch behind the
 *                       batch, because WHICH intermediate of a blow-up's last steps turns inf and which NaN decides
 *                       how the reference's flight ends (simulator.py:216, :238-242) and only the reference's own
 *                       operation order reproduces it.  The reference's outcome (apogee within 0.1 %, end reason,
 *                       step count) on every sample of the parity sets, at four times the speed of the gate
 *   ERPL_PREC_F32       fp32 state and RHS (time stays fp64): highest throughput; only the first-descent
 *                       apogee of a diverging sample is within 0.1 % of the reference */
enum { ERPL_PREC_F64 = 0, ERPL_PREC_F32 = 1, ERPL_PREC_F64_FAST = 2 };

/* erpl_batch.flags */
enum {
  ERPL_FLAG_STOP_AT_APOGEE = 1, /* extension (BASELINE config 2 "to apogee"): end a trajectory at the
                                  first post-step state with z>1000 and vz<0 (simulator.py:247) */
  ERPL_FLAG_CAPTURE_POSITION_ONLY = 2 /* trajectory capture (erpl_out.traj*) whose records are read for time and position only
                                  (the 'trajectory' of MonteCarloAnalyzer's results: monte_carlo.py:298-302): a sample whose
                                  position has turned non-finite for good - nothing the reference's loop can observe changes
                                  any more, SURVEY fact 9 - is fast-forwarded to max_time like in a batch without capture;
                                  its remaining records carry the exact time stamps and the state as it was then (the
                                  quaternion / angular-velocity / propellant columns are not advanced).  Summaries unchanged. */
};

/* rows of the per-sample summary, double [ERPL_SUMMARY_DIM][n] */
enum {
  ERPL_SUM_APOGEE_ALT = 0,     /* altitudes[argmax(altitudes)]  (simulator.py:488-490) incl. NaN-first rule */
  ERPL_SUM_APOGEE_TIME = 1,    /* time[argmax] - rail_time */
  ERPL_SUM_FIRST_APOGEE_ALT = 2,  /* max altitude up to the first-descent latch (simulator.py:247) */
  ERPL_SUM_FIRST_APOGEE_TIME = 3,
  ERPL_SUM_RANGE = 4,          /* sqrt(x_end^2 + y_end^2)  (simulator.py:493-494) */
  ERPL_SUM_FLIGHT_TIME = 5,    /* t_end - rail_time        (simulator.py:582) */
  ERPL_SUM_RAIL_EXIT_TIME = 6, /* simulator.py:104 */
  ERPL_SUM_RAIL_EXIT_SPEED = 7,/* simulator.py:107 */
  ERPL_SUM_IMPACT_X = 8,       /* final position */
  ERPL_SUM_IMPACT_Y = 9,
  ERPL_SUM_IMPACT_Z = 10,
  ERPL_SUM_STEPS = 11,         /* RK4 steps taken */
  ERPL_SUM_RAIL_EXIT_AOA = 12, /* simulator.py:121 */
  ERPL_SUM_RAIL_EXIT_SIDESLIP = 13, /* simulator.py:122 */
  ERPL_SUM_FINAL_VZ = 14,
  ERPL_SUM_MAX_SPEED = 15      /* max |v| over rail-exit state + all steps (NaN ignored) */
};

/* per-sample int32 status word: low byte = why the RK4 loop ended, bits above = latches */
enum {
  ERPL_END_MAX_TIME = 0,  /* while t < max_time ran out       (simulator.py:216) */
  ERPL_END_GROUND = 1,    /* z<=0.5 and vz<=0                 (simulator.py:238) */
  ERPL_END_ALTITUDE = 2,  /* z>100 km                         (simulator.py:242) */
  ERPL_END_COAST = 3,     /* coast time-out above 25 km       (simulator.py:260-264) */
  ERPL_END_APOGEE = 4,    /* ERPL_FLAG_STOP_AT_APOGEE */
  ERPL_ST_APOGEE_LATCHED = 1 << 8,
  ERPL_ST_CHUTE = 1 << 9, /* parachute latch set (simulator.py:366-369) */
  ERPL_ST_NAN = 1 << 10,  /* altitude became NaN at some step */
  ERPL_ST_INCOMPLETE = 1 << 11 /* no result: every status word is set to this value when its batch starts and is
                              replaced when the trajectory ends; it survives only if a lane hand-over timed out
                              (a logic error or a wedged wave - never observed), and the summary rows of such
                              a sample hold only its rail-exit values */
};

/* Everything shared by all samples of a batch.  Plain attribute values of the reference objects;
 * derived constants are computed inside the library exactly as the reference computes them. */
typedef struct erpl_config {
  /* Rocket (rocket.py:14-66) */
  double diameter;                 /* rocket.py:16  (used for propellant Ixx, rocket.py:122) */
  double center_of_mass_dry;       /* rocket.py:31 */
  double Ixx_dry, Iyy_dry;         /* rocket.py:34-35  (Izz := Iyy, rocket.py:128) */
  double reference_area;           /* rocket.py:39 */
  double reference_diameter;       /* rocket.py:40 */
  double cp_location;              /* rocket.py:56 (Barrowman, host-computed) */
  double fin_root_chord, fin_tip_chord, fin_span, fin_sweep_angle; /* rocket.py:18-22 */
  double parachute_area, parachute_cd, parachute_deployment_altitude; /* rocket.py:59-61 */
  double power_off_drag_factor;    /* rocket.py:66 */
  int32_t n_cd;                    /* Cd_data knots (rocket.py:43-47) */
  int32_t n_cp;                    /* CP_shift_data knots (rocket.py:50-53) */
  double cd_mach[ERPL_MAX_MACH_KNOTS], cd0[ERPL_MAX_MACH_KNOTS], cda[ERPL_MAX_MACH_KNOTS];
  double cp_mach[ERPL_MAX_MACH_KNOTS], cp_shift[ERPL_MAX_MACH_KNOTS];
  /* Motor, shared part (motor.py) */
  int32_t motor_kind;              /* ERPL_MOTOR_LIQUID | ERPL_MOTOR_SOLID */
  int32_t n_curve;                 /* solid thrust-curve knots (motor.py:31-41); 0 for liquid */
  double curve_time[ERPL_MAX_CURVE_KNOTS];
  double curve_thrust[ERPL_MAX_CURVE_KNOTS]; /* UNSCALED; per-sample thrust multiplies it (motor.py:105) */
  /* StandardAtmosphere (environment.py:13-24) */
  double sea_level_pressure, sea_level_temperature, temperature_lapse_rate;
  double gas_constant, gravity;
  double troposphere_height, stratosphere_height, stratosphere_temp;
  /* FlightSimulator (simulator.py:19-40, :42, :209) */
  double dt_initial;               /* rail step; flight step = min(dt_initial, 0.005) */
  double max_time;
  double rail_length;              /* 18.288 default argument (simulator.py:42) */
  double pitch_damping, yaw_damping;
} erpl_config;

/* One batch of independent samples (device pointers for erpl_mc_run_batch). */
typedef struct erpl_batch {
  int64_t n;             /* samples */
  int32_t precision;     /* ERPL_PREC_F64 (gate) | ERPL_PREC_F64_FAST | ERPL_PREC_F32 */
  int32_t k_wind;        /* wind knots; 0 = no profile -> zero wind (simulator.py:333-338) */
  int32_t flags;
  int32_t reserved;
  const double* ic;      /* [13][n]  position, velocity, quaternion (w,x,y,z), angular velocity */
  const double* rocket;  /* [2][n]   dry_mass, propellant_mass */
  const double* motor;   /* [4][n]   thrust (liquid: thrust_vacuum [N]; solid: curve multiplier),
                                     nozzle_exit_area, mass_flow_rate, burn_time */
  const double* alt_grid;/* [k_wind] shared altitude knots, strictly increasing */
  const void* wind;      /* [k_wind][3][n] per-sample u,v,w; float if F32, double otherwise */
} erpl_batch;

/* Outputs (device pointers).  traj_* are optional (NULL / 0 to disable): state history of a
 * caller-chosen subset, replacing the per-step lists of simulator.py:212-231. */
typedef struct erpl_out {
  double* summary;        /* [ERPL_SUMMARY_DIM][n] */
  int32_t* status;        /* [n] */
  int64_t n_traj;         /* samples to record */
  const int64_t* traj_ids;/* [n_traj] sample indices (device) */
  int64_t traj_stride;    /* record every traj_stride-th step (step 0 = rail-exit state, last step always) */
  int64_t traj_cap;       /* records per sample */
  double* traj;           /* [n_traj][traj_cap][ERPL_TRAJ_DIM] */
  int64_t* traj_len;      /* [n_traj] records written */
} erpl_out;

typedef struct erpl_ctx erpl_ctx;

int erpl_mc_abi_version(void);
const char* erpl_mc_last_error(void);

/* One context per GPU (one host thread / torch.distributed rank each).  HOST calls on one context are
 * serialised by the caller; device work of successive batches is ordered by the library (below). */
int erpl_mc_create(int device, erpl_ctx** out);
int erpl_mc_destroy(erpl_ctx* ctx);

/* Copies and validates cfg, derives the interval tables and uploads them. */
int erpl_mc_set_config(erpl_ctx* ctx, const erpl_config* cfg);

/* Grows the context workspace for batches of up to n samples: both workspaces of every lane within the current
 * overlap depth (hipMalloc happens here, never in run_batch / submit_batch once the workspace is large enough, so
 * run_batch is graph-capturable and a run of submits does not allocate half way). */
int erpl_mc_reserve(erpl_ctx* ctx, int64_t n);

/* Enqueues rail phase + flight integration + summaries for the batch on `hip_stream`
 * (a hipStream_t passed as void*; NULL = the default stream).  Asynchronous; stream-ordered: work
 * enqueued on `hip_stream` afterwards sees the results.  A batch uses one of the context's
 * workspaces; if that workspace is still in use by an earlier batch (on any stream), the new batch
 * waits for it on the device, so two streams can never corrupt each other's queues. */
int erpl_mc_run_batch(erpl_ctx* ctx, const erpl_batch* batch, const erpl_out* out, void* hip_stream);

/* More than one batch in flight.  A single pass over a batch that just fills the GPU (BASELINE config 3:
 * ~100 k samples) lasts as long as its longest trajectory while most lanes have already finished; the
 * next batch can use those lanes.  erpl_mc_submit_batch enqueues the batch on one of `depth` internal
 * streams, each with its own workspace (round-robin), AFTER everything enqueued on `hip_stream` so far
 * (inputs written there are visible), and returns a ticket; it does NOT make `hip_stream` wait.
 * erpl_mc_wait_batch makes `hip_stream` wait on the device for that batch (ticket < 0: for every batch
 * submitted so far); the host never blocks.  Output buffers belong to the batch until then.  Results
 * are bitwise those of erpl_mc_run_batch.  depth 1..8; changing it waits for work in flight.
 *
 * Each internal stream needs a hardware queue of its own to overlap with the others, and every lane has two
 * (main launches, sweep launches - erpl_mc_set_adopt).  The HIP runtime gives a process GPU_MAX_HW_QUEUES of
 * them (environment variable, read when the runtime initialises, default 4) and puts further streams on the
 * same queues, where their kernels run one after the other: with the default, four or five batches in flight
 * are slower than three.  The default depth is therefore 3, or (GPU_MAX_HW_QUEUES - 2) / 2 up to 8 if
 * GPU_MAX_HW_QUEUES >= 12 is in the environment at erpl_mc_create (the host sets it before its first HIP call;
 * the Python package sets 24 on import).  erpl_mc_get_overlap returns the depth in use. */
#define ERPL_MAX_OVERLAP 8
int erpl_mc_set_overlap(erpl_ctx* ctx, int depth);
/* How many batches of SHORT flights are in flight (scheduling only; results do not depend on it, bitwise).  With a hardware
 * queue per stream up to `depth` batches run beside each other, which pays for long flights (15 k-step flights: 8 in
 * flight 3 % faster than 4, fp32 7 %) and costs 2-3 % for short ones (2.5 k steps, the dispersed sets whose samples blow
 * up: 3 to 5 in flight 22.6 ms per 131 072-sample pass, 8 in flight 23.2 - fewer busy streams, not a later start: making
 * the eighth batch wait for the fourth without taking its stream away changes nothing).  Once the batches this context has
 * FINISHED averaged fewer than 8192 RK4 steps per trajectory, erpl_mc_submit_batch goes round the first `depth` lanes only.
 * Default 4; 0 = always all lanes of erpl_mc_set_overlap. */
int erpl_mc_set_short_flight_overlap(erpl_ctx* ctx, int depth);
int erpl_mc_get_overlap(erpl_ctx* ctx);
int erpl_mc_submit_batch(erpl_ctx* ctx, const erpl_batch* batch, const erpl_out* out, void* hip_stream, int64_t* ticket);
int erpl_mc_wait_batch(erpl_ctx* ctx, int64_t ticket, void* hip_stream);
/* Host-blocking wait for everything this context has enqueued, then ERPL_OK, or ERPL_ERR_INCOMPLETE naming the first
 * batch a lane hand-over of which timed out.  Reported ONCE: the batches up to the last ticket are acknowledged by
 * this call (and by erpl_mc_check_batch(ctx, -1)), later calls answer for later batches. */
int erpl_mc_synchronize(erpl_ctx* ctx);
/* Where results are consumed: host-blocking wait for batch `ticket` ALONE - its own completion event, batches
 * submitted after it keep running - then ERPL_OK, or ERPL_ERR_INCOMPLETE if one of ITS lane hand-overs timed out
 * (its unfinished samples carry ERPL_ST_INCOMPLETE).  Every ticket has its own record (event + pinned copy of
 * the batch's counters), so the answer does not depend on how often the batch's workspace has been reused since;
 * the records of the last 256 tickets are kept, an unreported failure that leaves the ring is latched and comes
 * back from the next call that checks an older ticket or all of them.  ticket < 0: every batch submitted so far
 * (= erpl_mc_synchronize).  erpl_mc_wait_batch itself never blocks the host and therefore cannot know; it does
 * report ERPL_ERR_INCOMPLETE for unacknowledged batches that had already finished that way when it is called. */
int erpl_mc_check_batch(erpl_ctx* ctx, int64_t ticket);

/* Launch geometry knobs (tuning / tests): threads per workgroup (default 64), max workgroups of the
 * persistent flight kernel (0 = library default), lane-refill threshold. */
int erpl_mc_set_launch(erpl_ctx* ctx, int block_threads, int max_blocks, int refill_threshold);

/* Which build of the fp32 flight kernel run_batch launches: 2 = all 256 VGPRs, two resident waves per
 * SIMD; 3 = capped at 168 VGPRs (spills), three resident waves - faster once the batch is large enough
 * to keep three waves per SIMD busy; 0 (default) = by batch size (3 from 2 304 samples per CU up).
 * Results do not depend on the value (bitwise). */
int erpl_mc_set_waves_per_simd(erpl_ctx* ctx, int waves);

/* Per-GPU compaction (BASELINE config 5): integrate in launches of `chunk_steps` RK4 steps; lanes
 * that are still flying at the end of a chunk park their state densely in a resume queue and the
 * next launch continues them with fully populated waves.  0 = one launch, no compaction; < 0 (the default) =
 * chosen per batch: 2048 for batches submitted with erpl_mc_submit_batch once the batches this context has
 * finished averaged >= 8192 RK4 steps per trajectory (there another batch fills every chunk barrier: -17...-20 %
 * time on 15 k - 42 k-step flights), otherwise one launch.  Results do not depend on the value (bitwise). */
int erpl_mc_set_chunk(erpl_ctx* ctx, int chunk_steps);

/* Lane adoption (no reference counterpart; scheduling only).  Once the sample queue is empty, a wave of the
 * flight kernel left with at most `lanes` flying trajectories writes them to the resume queue and leaves;
 * waves of the same launch that still fly more adopt them into their idle lanes, and two short sweep
 * launches fly out what nobody adopted.  For batches handed over with erpl_mc_submit_batch the sweeps run on
 * a second internal stream of the lane and the lane's next batch (on a second workspace) follows the main
 * launch at once, so the few long trajectories of a batch finish beside the next one.  0 = off; < 0 (the
 * default) = 24 (fp32) / 40 (fp64 builds) for every batch handed over with erpl_mc_submit_batch when the lanes have hardware queues of
 * their own (GPU_MAX_HW_QUEUES >= 2 x depth + 2, see erpl_mc_set_overlap), off without them, for
 * erpl_mc_run_batch (on the caller's one stream a batch is bound by its longest trajectory, which the
 * hand-overs lengthen) and whenever step chunks are in use.  Results do not depend on the value (bitwise). */
int erpl_mc_set_adopt(erpl_ctx* ctx, int lanes);
/* Test knob: how often an adopting lane polls for the ready word of the record it has claimed before it gives up
 * (default 4 194 304 polls of ~0.5 us).  < 0: it gives up at once, published or not - the injected failure of
 * tests/test_gpu_overlap.py: the record is left alone, its sample keeps ERPL_ST_INCOMPLETE and the batch fails
 * with ERPL_ERR_INCOMPLETE where it is checked. */
int erpl_mc_set_adopt_spin(erpl_ctx* ctx, int polls);

/* Diagnostics of the last run_batch on this ctx (after the stream has been synchronised):
 * total RK4 steps integrated over all samples and total wave-iterations executed. */
int erpl_mc_last_stats(erpl_ctx* ctx, double* total_steps, double* wave_iterations);
/* The same two counters of ONE batch handed over with erpl_mc_submit_batch, by its ticket (host-blocking for that batch
 * alone; ERPL_ERR_INVALID once the ticket has left the ring of the last 256 submitted batches, ERPL_ERR_INCOMPLETE as
 * erpl_mc_check_batch). */
int erpl_mc_ticket_stats(erpl_ctx* ctx, int64_t ticket, double* total_steps, double* wave_iterations);

/* Per-step diagnostic histories of FlightSimulator._extract_results (simulator.py:496-552) for m
 * stored records traj[m][ERPL_TRAJ_DIM] (as written by erpl_out.traj) of sample `sample` of an
 * ERPL_PREC_F64 batch: out[m][ERPL_DIAG_DIM] = euler(3), center_of_mass, mass, Ixx, Iyy, Izz, thrust,
 * drag, cd, cl, cm, cp_location_dynamic, stability_margin, angle_of_attack, sideslip_angle.
 * time_offset is the rail-exit time (the reference evaluates the thrust history at the shifted
 * time, simulator.py:543).  Device pointers, asynchronous on the stream. */
#define ERPL_DIAG_DIM 17
int erpl_mc_extract_histories(erpl_ctx* ctx, const erpl_batch* batch, int64_t sample, const double* traj,
                              int64_t m, double time_offset, double* out, void* hip_stream);

/* Host-side input preparation (no device work): the first `m` outputs of NumPy's legacy
 * `np.random.RandomState(seed)` for each of `n` integer seeds, bit for bit - the reference draws
 * every sample's dispersions, motor perturbation and wind turbulence from such per-sample streams
 * (monte_carlo.py:157, :253, :263; motor.py:95-125/:171-186; environment.py:161-198/:242-263).
 * ops[j] selects output j of every stream: ERPL_RS_GAUSS = one standard normal of `normal()` /
 * `randn()` (polar method, second value cached as the generator does), ERPL_RS_DOUBLE = one
 * `random_sample()` double in [0,1) as used by `uniform()`.  out (host memory) is [n][m] row-major,
 * or [m][n] when `by_output` is non-zero (all samples' j-th output contiguous).
 * Samples are independent, so the work is spread over `threads` host threads (<= 0: all cores). */
enum { ERPL_RS_GAUSS = 0, ERPL_RS_DOUBLE = 1 };
int erpl_mc_legacy_random_streams(const uint32_t* seeds, int64_t n, const uint8_t* ops, int32_t m,
                                  double* out, int32_t by_output, int32_t threads);

/* Host-side input preparation (no device work): the wind tables of all samples, each from a fresh
 * RandomState(seeds[s]) with 3 normals per knot in the order u, v, w - AR(1) turbulence over the k
 * altitude knots exactly as environment.py:218-265 (`base` != NULL: WindModel.perturb_wind_profile
 * around the [k][3] baseline profile) or environment.py:125-200 (`base` == NULL:
 * generate_stochastic_profile with the power-law mean wind (speed[s] * mean_scale[i]) * (cdir[s],
 * sdir[s]), mean_scale[i] = (alt_i / 10)^exponent).  sigma / rho / innov are the per-knot turbulence
 * sigma, AR(1) correlation and innovation sigma (rho[0], innov[0] unused).  Same operations in the
 * same order as the reference, fp64, no contraction: bit-identical tables.  wind is [k][3][n] (the
 * erpl_batch layout, host memory). */
int erpl_mc_legacy_wind_profiles(const uint32_t* seeds, int64_t n, int32_t k, const double* sigma,
                                 const double* rho, const double* innov, const double* base,
                                 const double* mean_scale, const double* speed, const double* cdir,
                                 const double* sdir, double* wind, int32_t threads);

/* On-device wind-table synthesis for the 100 k - 10 M throughput sets: the AR(1) turbulence
 * recursion over the k altitude knots of environment.py:161-198 / :242-263 plus the mean wind, for n
 * samples at once, from caller-supplied standard normals (counter-based device RNG; NOT the MT19937
 * streams of the parity sets).  wind[i][c][s] = base[i][c] + scale[i] * mean_c[s] + t_i, with
 * t_0 = sigma[0] f_c z, t_i = rho[i] t_{i-1} + innov[i] f_c z, f = (1, 1, 0.3), z = normals[i][c][s],
 * mean_2 = 0.  CSV mode (monte_carlo.py:268-280): base = the baseline profile, scale = 1, mean = the
 * uniform (speed, direction) offset; synthetic mode (:282-288): base = NULL, scale[i] = (alt_i/10)^p,
 * mean = speed * (cos, sin)(direction).  All pointers are device memory; normals / wind are [k][3][n];
 * wind is float for ERPL_PREC_F32, double otherwise (the recursion itself always runs in fp64). */
int erpl_mc_synth_wind(erpl_ctx* ctx, int64_t n, int32_t k, const double* normals, const double* sigma,
                       const double* rho, const double* innov, const double* base, const double* scale,
                       const double* mean_u, const double* mean_v, void* wind, int32_t precision, void* hip_stream);

/* Known-answer evaluation ON THE DEVICE (tests): one function of the hot path per lane, through the
 * device functions the flight kernel of `batch->precision` inlines.  Case j (0 <= j < m) uses the
 * per-sample parameters and wind table of sample j % batch->n; in / out are device double arrays
 * [rows][m].   what = ERPL_DBG_ATMOSPHERE: in altitude -> out T, P, rho, g (environment.py:26-108);
 * ERPL_DBG_AERO: in mach, alpha, beta, propellant_fraction, power_on -> out cd, cl, cy, cm, cyaw
 * (rocket.py:138-218; the fast builds take power_on = propellant_fraction > 0 as the RHS does);
 * ERPL_DBG_RHS: in t, y[14], parachute latch -> out dy[14], latch (simulator.py:295-460). */
enum { ERPL_DBG_ATMOSPHERE = 0, ERPL_DBG_AERO = 1, ERPL_DBG_RHS = 2 };
int erpl_mc_debug_eval(erpl_ctx* ctx, const erpl_batch* batch, int what, int64_t m, const double* in, double* out,
                       void* hip_stream);

/* Raw device counters of the last run_batch (16 doubles): [0] queue head, [1] RK4 steps, [2] wave
 * iterations, [8..15] per-segment s_memtime sums of a -DERPL_STAMPS=1 diagnostic build (0 otherwise). */
int erpl_mc_debug_counters(erpl_ctx* ctx, double* out16);

/* Kernel timing with HIP events recorded on the SAME stream as the kernels (enable before
 * run_batch; read after the stream has been synchronised).  rail_ms / flight_ms are the device
 * durations of the two kernels of the last run_batch. */
int erpl_mc_set_profiling(erpl_ctx* ctx, int enable);
int erpl_mc_last_kernel_ms(erpl_ctx* ctx, float* rail_ms, float* flight_ms);
/* The events live in a ring of ERPL_PROFILE_RING launches, so a timed loop needs no host sync
 * inside it: afterwards this returns the durations of the last min(max, ring, profiled runs)
 * launches, oldest first; *n_out = how many were written. */
#define ERPL_PROFILE_RING 256
int erpl_mc_kernel_ms_history(erpl_ctx* ctx, int max, float* rail_ms, float* flight_ms, int* n_out);

#ifdef __cplusplus
}
#endif
#endif /* ERPL_MC_H */
