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
from .models import LiquidMotor, SolidMotor


class UnsupportedModel(TypeError):
    """Raised when a duck-typed object cannot be flattened (the kernels hard-code the
    reference's model equations; there is no CPU fallback)."""


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


def generate_parameter_samples(uncertainty, n_samples, stream="seed_i"):
    """The dispersion draws of monte_carlo.py:156-179 (`stream='seed_i'`: RandomState(i) per
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


def dispersed_batch(rocket, motor, wind_model, base_initial_conditions, params_list,
                    base_altitude_profile=None, base_wind_profile=None, planar=False):
    """Per-sample inputs exactly as MonteCarloAnalyzer._run_single_simulation builds them
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
