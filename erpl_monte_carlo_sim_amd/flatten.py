"""Flattening of reference-shaped objects into the POD inputs of the C ABI.

`config_from_objects`  : Rocket + motor + StandardAtmosphere + FlightSimulator attributes ->
                         `erpl_config` (shared by all samples of a batch).
`HostBatch`            : per-sample SoA rows (IC 13, rocket 2, motor 4, wind K x 3) in NumPy,
                         the layout `erpl_batch` points at once moved to the GPU.
`dispersed_batch`      : the per-sample construction of MonteCarloAnalyzer._run_single_simulation
                         (monte_carlo.py:225-288) for many samples at once, bit-exact with the
                         reference's legacy-RandomState streams.
"""
import numpy as np

from . import _abi
from .models import LiquidMotor, SolidMotor, knot_constants


class UnsupportedModel(TypeError):
    """Raised when a duck-typed object cannot be flattened (the kernels hard-code the
    reference's model equations; there is no CPU fallback)."""


# Methods of the reference's model classes whose equations the kernels (or the vectorised host
# preparation) hard-code.  A user subclass - or an unrelated duck-typed class - that defines one of them
# would be silently ignored by the GPU path, so it is refused (there is no CPU fallback).
_KNOWN_MODELS = ("Rocket", "SolidMotor", "LiquidMotor", "StandardAtmosphere", "WindModel")
_HARD_CODED = {
    "rocket": ("get_mass_properties", "get_aerodynamic_coefficients", "get_dynamic_cp"),            # rocket.py:105-218
    "motor": ("get_thrust", "get_mass_flow_rate", "get_propellant_remaining", "perturb_for_monte_carlo"),  # motor.py:54-186
    "atmosphere": ("get_properties", "get_gravity"),                                                 # environment.py:26-108
    "wind_model": ("get_wind_at_altitude", "generate_stochastic_profile", "perturb_wind_profile",
                   "power_law_profile"),                                                            # environment.py:118-276
}


def reject_overrides(obj, role):
    """Raise UnsupportedModel if `obj` (playing `role`) redefines a method the GPU path hard-codes: in a
    subclass of one of the known model classes (the reference's or this package's), on the instance, or
    in a class unrelated to them."""
    methods = _HARD_CODED[role]
    patched = [m for m in methods if m in getattr(obj, "__dict__", {})]
    if patched:
        raise UnsupportedModel(f"{role}: instance attribute(s) {patched} replace methods the HIP kernels hard-code")
    for cls in type(obj).__mro__:
        if cls is object:
            break
        if cls.__name__ in _KNOWN_MODELS:
            return            # everything from here up is the model the kernels implement
        over = [m for m in methods if m in cls.__dict__]
        if over:
            raise UnsupportedModel(
                f"{role}: class {cls.__name__} overrides {over}; the HIP kernels hard-code the reference's "
                f"equations for these and there is no CPU fallback")


def euler_to_quaternion(roll, pitch, yaw):
    """'xyz' Euler angles -> scalar-first quaternion (w, x, y, z); utils.py:129-136 via :13-35."""
    cr, sr = np.cos(roll / 2), np.sin(roll / 2)
    cp, sp = np.cos(pitch / 2), np.sin(pitch / 2)
    cy, sy = np.cos(yaw / 2), np.sin(yaw / 2)
    return np.array([cr * cp * cy + sr * sp * sy,
                     sr * cp * cy - cr * sp * sy,
                     cr * sp * cy + sr * cp * sy,
                     cr * cp * sy - sr * sp * cy])


def motor_kind(motor):
    if isinstance(motor, SolidMotor) or hasattr(motor, "thrust_curve_thrust"):
        return _abi.MOTOR_SOLID
    if isinstance(motor, LiquidMotor) or hasattr(motor, "thrust_vacuum"):
        return _abi.MOTOR_LIQUID
    raise UnsupportedModel(f"cannot flatten motor of type {type(motor).__name__}")


def _fill(arr, values, cap, what):
    vals = [float(v) for v in values]
    if len(vals) > cap:
        raise UnsupportedModel(f"{what}: {len(vals)} knots exceed the ABI limit {cap}")
    if len(vals) > 1 and not all(b > a for a, b in zip(vals, vals[1:])) and what.endswith("mach"):
        raise UnsupportedModel(f"{what} must be strictly increasing")
    for i, v in enumerate(vals):
        arr[i] = v
    return len(vals)


def config_from_objects(rocket, motor, atmosphere, dt_initial=0.01, max_time=300.0,
                        rail_length=18.288, pitch_damping=20.0, yaw_damping=20.0):
    """Build the shared `erpl_config`.  Field-by-field sources are listed in include/erpl_mc.h."""
    reject_overrides(rocket, "rocket")
    reject_overrides(motor, "motor")
    reject_overrides(atmosphere, "atmosphere")
    c = _abi.ErplConfig()
    for name in ("diameter", "center_of_mass_dry", "Ixx_dry", "Iyy_dry", "reference_area",
                 "reference_diameter", "cp_location", "fin_root_chord", "fin_tip_chord",
                 "fin_span", "fin_sweep_angle", "parachute_area", "parachute_cd",
                 "parachute_deployment_altitude", "power_off_drag_factor"):
        setattr(c, name, float(getattr(rocket, name)))
    n1 = _fill(c.cd_mach, rocket.Cd_data["mach"], _abi.MAX_MACH_KNOTS, "Cd_data mach")
    n2 = _fill(c.cd0, rocket.Cd_data["cd0"], _abi.MAX_MACH_KNOTS, "Cd_data cd0")
    n3 = _fill(c.cda, rocket.Cd_data["cda"], _abi.MAX_MACH_KNOTS, "Cd_data cda")
    if not (n1 == n2 == n3) or n1 < 1:
        raise UnsupportedModel("Cd_data columns must have equal, non-zero length")
    c.n_cd = n1
    m1 = _fill(c.cp_mach, rocket.CP_shift_data["mach"], _abi.MAX_MACH_KNOTS, "CP_shift_data mach")
    m2 = _fill(c.cp_shift, rocket.CP_shift_data["cp_shift"], _abi.MAX_MACH_KNOTS, "CP_shift_data cp_shift")
    if m1 != m2 or m1 < 1:
        raise UnsupportedModel("CP_shift_data columns must have equal, non-zero length")
    c.n_cp = m1
    c.motor_kind = motor_kind(motor)
    if c.motor_kind == _abi.MOTOR_SOLID:
        # the curve is stored UNSCALED; a per-sample multiplier (motor row 0) scales it in-kernel
        # exactly like motor.py:105.  A perturbed SolidMotor passed directly carries its own
        # already-scaled curve and gets multiplier 1.0 (motor_row).
        base = np.asarray(motor.thrust_curve_thrust, dtype=np.float64)
        k1 = _fill(c.curve_time, motor.thrust_curve_time, _abi.MAX_CURVE_KNOTS, "thrust_curve_time")
        k2 = _fill(c.curve_thrust, base, _abi.MAX_CURVE_KNOTS, "thrust_curve_thrust")
        if k1 != k2 or k1 < 1:
            raise UnsupportedModel("thrust curve columns must have equal, non-zero length")
        c.n_curve = k1
    else:
        c.n_curve = 0
    for name in ("sea_level_pressure", "sea_level_temperature", "temperature_lapse_rate",
                 "gas_constant", "gravity", "troposphere_height", "stratosphere_height",
                 "stratosphere_temp"):
        setattr(c, name, float(getattr(atmosphere, name)))
    c.dt_initial = float(dt_initial)
    c.max_time = float(max_time)
    c.rail_length = float(rail_length)
    c.pitch_damping = float(pitch_damping)
    c.yaw_damping = float(yaw_damping)
    return c


def motor_row(motor):
    """[thrust, nozzle_exit_area, mass_flow_rate, burn_time] of one motor object."""
    if motor_kind(motor) == _abi.MOTOR_SOLID:
        thrust = 1.0
    else:
        thrust = float(motor.thrust_vacuum)
    return [thrust, float(motor.nozzle_exit_area), float(motor.mass_flow_rate), float(motor.burn_time)]


class HostBatch:
    """SoA rows of n samples in host memory (float64)."""

    def __init__(self, n, k_wind=0):
        self.n = int(n)
        self.k_wind = int(k_wind)
        self.ic = np.zeros((_abi.IC_DIM, n))
        self.rocket = np.zeros((_abi.ROCKET_DIM, n))
        self.motor = np.zeros((_abi.MOTOR_DIM, n))
        self.alt_grid = np.zeros(k_wind)
        self.wind = np.zeros((k_wind, 3, n))

    def set_ic(self, i, position, velocity, attitude, angular_velocity):
        self.ic[0:3, i] = position
        self.ic[3:6, i] = velocity
        self.ic[6:10, i] = euler_to_quaternion(attitude[0], attitude[1], attitude[2])
        self.ic[10:13, i] = angular_velocity

    def take(self, idx):
        idx = np.asarray(idx)
        out = HostBatch(len(idx), self.k_wind)
        out.ic = np.ascontiguousarray(self.ic[:, idx])
        out.rocket = np.ascontiguousarray(self.rocket[:, idx])
        out.motor = np.ascontiguousarray(self.motor[:, idx])
        out.alt_grid = self.alt_grid.copy()
        out.wind = np.ascontiguousarray(self.wind[:, :, idx])
        return out


def single_flight_batch(rocket, motor, initial_conditions, wind_profile, altitude_profile):
    """Inputs of one FlightSimulator.simulate_flight call (simulator.py:131-166)."""
    has_wind = wind_profile is not None and altitude_profile is not None and len(wind_profile) > 0
    k = len(altitude_profile) if has_wind else 0
    b = HostBatch(1, k)
    ic = initial_conditions
    b.set_ic(0, ic.get("position", [0.0, 0.0, 0.0]), ic.get("velocity", [0.0, 0.0, 0.0]),
             ic.get("attitude", [0.0, 0.0, 0.0]), ic.get("angular_velocity", [0.0, 0.0, 0.0]))
    b.rocket[:, 0] = [float(rocket.dry_mass), float(rocket.propellant_mass)]
    b.motor[:, 0] = motor_row(motor)
    if has_wind:
        b.alt_grid[:] = np.asarray(altitude_profile, dtype=np.float64)
        b.wind[:, :, 0] = np.asarray(wind_profile, dtype=np.float64)
    return b


def generate_parameter_samples_loop(uncertainty, n_samples, stream="seed_i"):
    """Per-sample Python restatement (kept as the cross-check of the vectorised path): the dispersion
    draws of monte_carlo.py:156-179 (`stream='seed_i'`: RandomState(i) per
    sample) or :181-201 (`stream='seed_42'`: one RandomState(42) stream).  Same calls in the same
    order on the legacy generator, so the values are bit-identical to the reference's."""
    u = uncertainty
    out = []
    rs = np.random.RandomState(42) if stream == "seed_42" else None
    for i in range(n_samples):
        if stream != "seed_42":
            rs = np.random.RandomState(i)
        out.append({
            "initial_position_offset": rs.normal(0, u["initial_position"]),
            "initial_velocity_offset": rs.normal(0, u["initial_velocity"]),
            "initial_attitude_offset": rs.normal(0, u["initial_attitude"]),
            "initial_angular_velocity_offset": rs.normal(0, u["initial_angular_velocity"]),
            "mass_multiplier": rs.normal(1.0, u["mass_uncertainty"]),
            "thrust_multiplier": rs.normal(1.0, u["thrust_uncertainty"]),  # dead
            "wind_speed": rs.uniform(*u["wind_speed_range"]),
            "wind_direction": rs.uniform(*u["wind_direction_range"]),
            "density_multiplier": rs.normal(1.0, u["atmospheric_density_uncertainty"]),  # dead
            "random_seed": i,
        })
    return out


def dispersed_batch_loop(rocket, motor, wind_model, base_initial_conditions, params_list,
                         base_altitude_profile=None, base_wind_profile=None, planar=False):
    """Per-sample Python restatement (kept as the cross-check of the vectorised path).  Per-sample inputs exactly as MonteCarloAnalyzer._run_single_simulation builds them
    (monte_carlo.py:228-288): IC + offsets, masses x mass_multiplier, motor perturbed from a
    fresh RandomState(seed) with propellant mass / burn time re-synchronised (:258-260), wind
    from another fresh RandomState(seed) (CSV baseline + AR(1) + uniform offset, or the 100-knot
    synthetic profile).  `planar=True` zeroes every out-of-plane input (Set P)."""
    n = len(params_list)
    use_base = base_wind_profile is not None and base_altitude_profile is not None
    alt = (np.asarray(base_altitude_profile, dtype=np.float64) if use_base
           else np.linspace(0, 25000, 100))
    b = HostBatch(n, len(alt))
    b.alt_grid[:] = alt
    ic0 = base_initial_conditions
    solid = motor_kind(motor) == _abi.MOTOR_SOLID
    for i, p in enumerate(params_list):
        def off(key, pkey):
            if key in ic0:
                return np.array(ic0[key]) + p[pkey]
            return p[pkey]
        pos = off("position", "initial_position_offset")
        vel = off("velocity", "initial_velocity_offset")
        att = off("attitude", "initial_attitude_offset")
        omg = off("angular_velocity", "initial_angular_velocity_offset")
        if planar:
            base_v = np.array(ic0.get("velocity", [0.0, 0.0, 0.0]))
            base_a = np.array(ic0.get("attitude", [0.0, 0.0, 0.0]))
            vel = base_v + p["initial_velocity_offset"] * [1, 0, 1]
            att = base_a + p["initial_attitude_offset"] * [0, 1, 0]
            omg = p["initial_angular_velocity_offset"] * [0, 1, 0]
        b.set_ic(i, pos, vel, att, omg)
        dry = rocket.dry_mass * p["mass_multiplier"]
        prop = rocket.propellant_mass * p["mass_multiplier"]
        b.rocket[:, i] = [dry, prop]
        pm = motor.perturb_for_monte_carlo(np.random.RandomState(p["random_seed"]))
        pm.propellant_mass = prop
        if hasattr(pm, "mass_flow_rate") and pm.mass_flow_rate > 0:
            pm.burn_time = pm.propellant_mass / pm.mass_flow_rate
        thrust = pm._thrust_multiplier if solid else pm.thrust_vacuum
        b.motor[:, i] = [thrust, pm.nozzle_exit_area, pm.mass_flow_rate, pm.burn_time]
        rs = np.random.RandomState(p["random_seed"])
        if use_base:
            w = wind_model.perturb_wind_profile(alt, base_wind_profile, random_state=rs)
            w[:, 0] += p["wind_speed"] * np.cos(p["wind_direction"])
            if planar:
                w[:, 1] = 0.0
            else:
                w[:, 1] += p["wind_speed"] * np.sin(p["wind_direction"])
        else:
            w = wind_model.generate_stochastic_profile(alt, p["wind_speed"], p["wind_direction"],
                                                       random_state=rs)
            if planar:
                w[:, 1] = 0.0
        b.wind[:, :, i] = w
    return b


# ---------------------------------------------------------------------------------- vectorised
# The same constructions for all samples at once.  The per-sample legacy RandomState(seed) streams
# come from the C ABI (erpl_mc_legacy_random_streams, bit-identical to NumPy's generator); the
# arithmetic on them is the reference's, element for element, as NumPy array expressions - IEEE
# +,-,*,/ and sqrt do not depend on the array shape, and np.cos / np.sin / np.exp give the same bits
# for an array element as for a scalar call (checked by tests/test_host.py against the per-sample
# loops above and against the reference's own captured inputs); `**` does not, so powers are taken
# on scalars exactly where the reference takes them on scalars.
_PARAM_OPS = "g" * 14 + "uu" + "g"   # pos3 vel3 att3 omega3 mass thrust | wind speed, direction | density


def legacy_streams(seeds, ops, threads=0, by_output=False):
    """First len(ops) outputs ('g' normal / 'u' uniform double) of np.random.RandomState(seed) for
    every seed: float64 [n, len(ops)], or [len(ops), n] with by_output=True."""
    import ctypes as C
    seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
    threads = threads if threads > 0 else host_cores()
    code = np.frombuffer(ops.encode(), dtype=np.uint8)
    code = np.ascontiguousarray(np.where(code == ord("g"), _abi.RS_GAUSS, _abi.RS_DOUBLE).astype(np.uint8))
    out = np.empty((code.size, seeds.size) if by_output else (seeds.size, code.size), dtype=np.float64)
    lib = _abi.load_library()
    _abi.check(lib, lib.erpl_mc_legacy_random_streams(
        seeds.ctypes.data_as(C.c_void_p), C.c_int64(seeds.size), code.ctypes.data_as(C.c_void_p),
        C.c_int32(code.size), out.ctypes.data_as(C.c_void_p), C.c_int32(1 if by_output else 0), C.c_int32(threads)),
        "erpl_mc_legacy_random_streams")
    return out


def generate_parameter_arrays(uncertainty, n_samples, stream="seed_i"):
    """monte_carlo.py:156-179 for samples 0..n-1 (`stream='seed_i'`: RandomState(i) each) or :181-201
    (`stream='seed_42'`: ONE RandomState(42) stream drawn sample after sample) as arrays: dict of [n, 3] /
    [n] float64 plus "random_seed"."""
    u = uncertainty
    if stream == "seed_42":   # the same 17 draws per sample, continued on one generator (the polar method's cached
        # second normal carries over from sample to sample exactly as in the reference's loop)
        r = legacy_streams(np.array([42], dtype=np.uint32), _PARAM_OPS * n_samples).reshape(n_samples, len(_PARAM_OPS)) \
            if n_samples else np.empty((0, len(_PARAM_OPS)))
    elif n_samples >= 4 * _PARAM_CHUNK:
        # the streams are per sample: blocks of samples are drawn AND scaled side by side (the C generator and NumPy's
        # element-wise loops both release the interpreter lock; the single-threaded scaling below is otherwise half
        # of the time at 10^6 samples), then concatenated - same values
        from concurrent.futures import ThreadPoolExecutor
        starts = list(range(0, n_samples, _PARAM_CHUNK))
        workers = max(1, min(len(starts), host_workers()))

        def block(a):
            b = min(n_samples, a + _PARAM_CHUNK)
            d = _scale_parameter_draws(u, legacy_streams(np.arange(a, b, dtype=np.uint32), _PARAM_OPS,
                                                         threads=max(1, host_cores() // workers)))   # (host_cores: what the cgroup grants)
            d["random_seed"] = np.arange(a, b, dtype=np.int64)
            return d
        with ThreadPoolExecutor(workers) as ex:
            parts = list(ex.map(block, starts))
        return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    else:
        r = legacy_streams(np.arange(n_samples, dtype=np.uint32), _PARAM_OPS)
    out = _scale_parameter_draws(u, r)
    out["random_seed"] = np.arange(n_samples, dtype=np.int64)
    return out


_PARAM_CHUNK = 65536


def host_cores():
    """CPU threads this process may actually use: min(affinity mask, cgroup cpu quota) - a container often shows the
    host's 256 cores and grants 16."""
    import os
    try:
        n = max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        n = max(1, os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = max(1, min(n, int(q / per + 0.5)))
        except Exception:
            pass
    return n


def host_workers():
    """Python-level workers for host-side sample construction: each one drives the C generator's own threads, so a few
    are enough to keep the NumPy parts off the critical path."""
    return max(1, min(4, host_cores() // 4))


def _scale_parameter_draws(u, r):
    """The arithmetic of monte_carlo.py:160-177 on the raw draws r [n, 17] (element for element the reference's)."""
    sc = lambda key: np.asarray(u[key], dtype=np.float64)[None, :]
    lo_s, hi_s = u["wind_speed_range"]
    lo_d, hi_d = u["wind_direction_range"]
    return {
        "initial_position_offset": 0.0 + sc("initial_position") * r[:, 0:3],
        "initial_velocity_offset": 0.0 + sc("initial_velocity") * r[:, 3:6],
        "initial_attitude_offset": 0.0 + sc("initial_attitude") * r[:, 6:9],
        "initial_angular_velocity_offset": 0.0 + sc("initial_angular_velocity") * r[:, 9:12],
        "mass_multiplier": 1.0 + u["mass_uncertainty"] * r[:, 12],
        "thrust_multiplier": 1.0 + u["thrust_uncertainty"] * r[:, 13],
        "wind_speed": lo_s + (hi_s - lo_s) * r[:, 14],
        "wind_direction": lo_d + (hi_d - lo_d) * r[:, 15],
        "density_multiplier": 1.0 + u["atmospheric_density_uncertainty"] * r[:, 16],
    }


_VEC_KEYS = ("initial_position_offset", "initial_velocity_offset", "initial_attitude_offset",
             "initial_angular_velocity_offset")
_SCALAR_KEYS = ("mass_multiplier", "thrust_multiplier", "wind_speed", "wind_direction", "density_multiplier")


def generate_parameter_samples(uncertainty, n_samples, stream="seed_i"):
    """The dispersion draws of monte_carlo.py:156-179 (`stream='seed_i'`: RandomState(i) per
    sample) or :181-201 (`stream='seed_42'`: one RandomState(42) stream) as the reference's list of
    per-sample dicts, bit-identical to the reference's values."""
    return _arrays_to_params(generate_parameter_arrays(uncertainty, n_samples, stream))


def params_to_arrays(params_list):
    """List of per-sample dicts -> the array form of generate_parameter_arrays."""
    a = {k: np.array([p[k] for p in params_list], dtype=np.float64).reshape(len(params_list), 3) for k in _VEC_KEYS}
    a.update({k: np.array([p[k] for p in params_list], dtype=np.float64) for k in _SCALAR_KEYS if k in params_list[0]})
    a["random_seed"] = np.array([p["random_seed"] for p in params_list], dtype=np.int64)
    return a


def legacy_wind_profiles(wind_model, alt, seeds, base=None, speed=None, cdir=None, sdir=None, threads=0):
    """Wind tables [K, 3, n] of all samples, each from a fresh RandomState(seed): perturb_wind_profile
    around `base` [K, 3] (environment.py:218-265), or generate_stochastic_profile with the power-law
    mean wind of (speed, direction) per sample (environment.py:125-200); erpl_mc_legacy_wind_profiles."""
    import ctypes as C
    seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
    threads = threads if threads > 0 else host_cores()
    n, K = seeds.size, len(alt)
    sigma, rho, innov = knot_constants(wind_model, alt)
    f = lambda v: np.ascontiguousarray(v, dtype=np.float64)
    sigma, rho, innov = f(sigma), f([0.0] + list(rho[1:])), f([0.0] + list(innov[1:]))
    ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    scale = None
    if base is None:   # power law on scalars, as environment.py:118-123 evaluates it (`**` differs between scalar and array code)
        scale = f([(np.float64(a) / 10.0) ** wind_model.power_law_exponent for a in alt])
        speed, cdir, sdir = f(speed), f(cdir), f(sdir)
    else:
        base = f(base)
    out = np.empty((K, 3, n), dtype=np.float64)
    lib = _abi.load_library()
    _abi.check(lib, lib.erpl_mc_legacy_wind_profiles(
        ptr(seeds), C.c_int64(n), C.c_int32(K), ptr(sigma), ptr(rho), ptr(innov), ptr(base), ptr(scale),
        ptr(speed), ptr(cdir), ptr(sdir), ptr(out), C.c_int32(threads)), "erpl_mc_legacy_wind_profiles")
    return out


def _ar1_profiles(wind_model, alt, g, mean_u=None, mean_v=None, base=None):
    """AR(1) turbulence of environment.py:161-198 / :242-263 for all samples: g [3K, n] holds each
    sample's normals in draw order (u, v, w per knot); returns [K, 3, n]."""
    sigma, rho, innov = knot_constants(wind_model, alt)
    K, n = len(sigma), g.shape[1]
    out = np.empty((K, 3, n))
    zero = np.zeros(n)
    mu = (lambda k: mean_u[k]) if mean_u is not None else (lambda k: base[k, 0])
    mv = (lambda k: mean_v[k]) if mean_v is not None else (lambda k: base[k, 1])
    mw = (lambda k: zero) if base is None else (lambda k: base[k, 2])
    out[0, 0] = mu(0) + (0.0 + sigma[0] * g[0])
    out[0, 1] = mv(0) + (0.0 + sigma[0] * g[1])
    w0 = 0.0 + (sigma[0] * 0.3) * g[2]
    out[0, 2] = w0 if base is None else mw(0) + w0
    for i in range(1, K):
        pu = out[i - 1, 0] - mu(i - 1)
        pv = out[i - 1, 1] - mv(i - 1)
        pw = out[i - 1, 2] if base is None else out[i - 1, 2] - mw(i - 1)
        tu = rho[i] * pu + (0.0 + innov[i] * g[3 * i])
        tv = rho[i] * pv + (0.0 + innov[i] * g[3 * i + 1])
        tw = rho[i] * pw + (0.0 + (innov[i] * 0.3) * g[3 * i + 2])
        out[i, 0] = mu(i) + tu
        out[i, 1] = mv(i) + tv
        out[i, 2] = tw if base is None else mw(i) + tw
    return out


def dispersed_batch(rocket, motor, wind_model, base_initial_conditions, params_list,
                    base_altitude_profile=None, base_wind_profile=None, planar=False, threads=0):
    """Per-sample inputs exactly as MonteCarloAnalyzer._run_single_simulation builds them
    (monte_carlo.py:228-288), for all samples at once: IC + offsets, masses x mass_multiplier, motor
    perturbed from a fresh RandomState(seed) with propellant mass / burn time re-synchronised
    (:258-260), wind from another fresh RandomState(seed) (CSV baseline + AR(1) + uniform offset, or
    the 100-knot synthetic profile).  `planar=True` zeroes every out-of-plane input (Set P, SURVEY
    §8d).  `params_list` is the reference's list of dicts or the dict of arrays of
    generate_parameter_arrays.  `threads`: host threads of the C generators (0 = all the process may use; callers that
    prepare several batches side by side share them out)."""
    reject_overrides(wind_model, "wind_model")
    use_base = base_wind_profile is not None and base_altitude_profile is not None
    alt = (np.asarray(base_altitude_profile, dtype=np.float64) if use_base else np.linspace(0, 25000, 100))
    if not isinstance(params_list, dict) and len(params_list) == 0:
        b = HostBatch(0, len(alt))
        b.alt_grid[:] = alt
        return b
    P = params_list if isinstance(params_list, dict) else params_to_arrays(params_list)
    seeds = np.asarray(P["random_seed"])
    n = seeds.size
    if seeds.min() < 0 or seeds.max() > 0xFFFFFFFF:
        raise UnsupportedModel("random_seed must fit an unsigned 32-bit integer (np.random.RandomState(int))")
    b = HostBatch(n, len(alt))
    b.alt_grid[:] = alt
    ic0 = base_initial_conditions

    def off(key, pkey):   # np.array(ic0[key]) + offset, or the offset alone (monte_carlo.py:228-249)
        return (np.array(ic0[key], dtype=np.float64)[None, :] + P[pkey]) if key in ic0 else P[pkey]
    pos = off("position", "initial_position_offset")
    vel = off("velocity", "initial_velocity_offset")
    att = off("attitude", "initial_attitude_offset")
    omg = off("angular_velocity", "initial_angular_velocity_offset")
    if planar:
        base_v = np.array(ic0.get("velocity", [0.0, 0.0, 0.0]), dtype=np.float64)[None, :]
        base_a = np.array(ic0.get("attitude", [0.0, 0.0, 0.0]), dtype=np.float64)[None, :]
        vel = base_v + P["initial_velocity_offset"] * np.array([1, 0, 1])[None, :]
        att = base_a + P["initial_attitude_offset"] * np.array([0, 1, 0])[None, :]
        omg = P["initial_angular_velocity_offset"] * np.array([0, 1, 0])[None, :]
    b.ic[0:3] = pos.T
    b.ic[3:6] = vel.T
    b.ic[6:10] = euler_to_quaternion(att[:, 0], att[:, 1], att[:, 2])
    b.ic[10:13] = omg.T
    dry = rocket.dry_mass * P["mass_multiplier"]
    prop = rocket.propellant_mass * P["mass_multiplier"]
    b.rocket[0], b.rocket[1] = dry, prop

    # motor perturbation from a fresh RandomState(seed) (motor.py:95-125 / :171-186)
    if motor_kind(motor) == _abi.MOTOR_SOLID:
        g = legacy_streams(seeds, "ggg", threads=threads)   # thrust, burn time, impulse (the last two: drawn, then overwritten/unused)
        k = 1.0 + motor.thrust_uncertainty * g[:, 0]
        mdot = 4.26 * k
        b.motor[0] = k
        b.motor[1] = motor.nozzle_exit_area * k
    else:
        g = legacy_streams(seeds, "gg", threads=threads)    # thrust, mass flow
        k = 1.0 + motor.thrust_uncertainty * g[:, 0]
        kf = 1.0 + motor.mass_flow_uncertainty * g[:, 1]
        tv = motor.thrust_vacuum * k
        mdot = motor.mass_flow_rate * kf
        b.motor[0] = tv
        b.motor[1] = (tv - motor.thrust_sea_level * k) / 101325.0
    if not np.all(mdot > 0):   # the re-synchronisation of :258-260 is conditional on a positive mass flow
        return dispersed_batch_loop(rocket, motor, wind_model, base_initial_conditions,
                                    _arrays_to_params(P), base_altitude_profile, base_wind_profile, planar)
    b.motor[2] = mdot
    b.motor[3] = prop / mdot

    # wind from another fresh RandomState(seed): AR(1) turbulence in the C ABI (same operations, same
    # order, fp64), the per-knot constants and the sample-wise offsets as NumPy expressions
    speed, direction = P["wind_speed"], P["wind_direction"]
    cd, sd = np.cos(direction), np.sin(direction)
    if use_base:
        w = legacy_wind_profiles(wind_model, alt, seeds, base=np.asarray(base_wind_profile, dtype=np.float64), threads=threads)
        w[:, 0, :] += speed * cd
        if planar:
            w[:, 1, :] = 0.0
        else:
            w[:, 1, :] += speed * sd
    else:
        w = legacy_wind_profiles(wind_model, alt, seeds, speed=speed, cdir=cd, sdir=sd, threads=threads)
        if planar:
            w[:, 1, :] = 0.0
    b.wind = w
    return b


def _arrays_to_params(P):
    """Dict of arrays -> the reference's list of per-sample dicts (3-vectors as float64 arrays, scalars as
    Python floats).  Built from column lists: the rows of each 3-vector are views into ONE private copy of its
    [n, 3] array (no per-sample allocation - at 1e5 samples the per-sample `.copy()` + `float()` form spent
    140 us per sample, most of it in the cyclic garbage collector walking the growing list of dicts)."""
    import gc
    n = len(P["random_seed"])
    keys = list(_VEC_KEYS) + [k for k in _SCALAR_KEYS if k in P] + ["random_seed"]
    cols = [list(np.array(P[k], dtype=np.float64)) for k in _VEC_KEYS]
    cols += [np.asarray(P[k], dtype=np.float64).tolist() for k in _SCALAR_KEYS if k in P]
    cols.append(np.asarray(P["random_seed"]).astype(np.int64).tolist())
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        out = [dict(zip(keys, row)) for row in zip(*cols)]
    finally:
        if was_enabled:
            gc.enable()
    assert len(out) == n
    return out
