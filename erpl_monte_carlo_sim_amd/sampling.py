"""Vectorised dispersion + wind-profile synthesis in torch.

The reference draws every sample from a legacy MT19937 `RandomState(i)` in Python loops
(monte_carlo.py:156-179, environment.py:125-265), ~ms per sample.  `flatten.dispersed_batch`
restates that bit for bit and is what the parity sets use.  For the 100 k - 10 M throughput
configurations this module draws the SAME distributions with torch's counter-based generator,
all samples at once, directly into the SoA tensors the kernels read (no host round trip), with the AR(1)
wind recursion in one HIP kernel of the C ABI (erpl_mc_synth_wind).  It is not bit-identical to MT19937
and is therefore never used for parity against the reference; its distribution is tested against the
bit-exact host generator (tests/test_gpu_sampling.py).
"""
import math

import torch

from . import _abi
from .engine import DeviceBatch
from .flatten import motor_kind, reject_overrides
from .models import knot_constants

DEFAULT_UNCERTAINTY = {
    "initial_position": [0.0, 0.0, 0.0],
    "initial_velocity": [0.1, 0.1, 0.1],
    "initial_attitude": [0.005, 0.005, 0.005],
    "initial_angular_velocity": [0.005, 0.005, 0.005],
    "mass_uncertainty": 0.02,
    "thrust_uncertainty": 0.03,
    "wind_speed_range": [0.0, 5.0],
    "wind_direction_range": [0.0, 2 * math.pi],
    "atmospheric_density_uncertainty": 0.05,
}


def _euler_to_quaternion(roll, pitch, yaw):
    cr, sr = torch.cos(roll / 2), torch.sin(roll / 2)
    cp, sp = torch.cos(pitch / 2), torch.sin(pitch / 2)
    cy, sy = torch.cos(yaw / 2), torch.sin(yaw / 2)
    return torch.stack([cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy,
                        cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy])


def _check_ranges(ic, rk, mt, mean_u, mean_v):
    """The input contract `DeviceBatch.from_host` enforces, for batches generated on the device: the
    kernels assume finite inputs, positive masses and mass flow, a finite non-negative burn time.  (The
    wind table is a finite linear map of the finite normals, per-knot constants and mean-wind rows.)"""
    ok = (torch.isfinite(ic).all() & torch.isfinite(rk).all() & torch.isfinite(mt).all()
          & torch.isfinite(mean_u).all() & torch.isfinite(mean_v).all()
          & (rk > 0).all() & (mt[2] > 0).all() & (mt[3] >= 0).all())
    if not bool(ok):
        raise _abi.ErplError("generated dispersions violate the kernel input contract (finite values, positive "
                             "masses and mass flow): check the uncertainty table")


def synthetic_dispersions(n, rocket, motor, wind_model, base_initial_conditions, device,
                          precision=_abi.PREC_F32, seed=1234, uncertainty=None,
                          base_altitude_profile=None, base_wind_profile=None, planar=False,
                          n_wind_knots=100, engine=None):
    """n dispersed samples with the distributions of monte_carlo.py:156-179 / :225-288, drawn with
    torch's counter-based device generator straight into the SoA tensors the kernels read.

    The reference draws a sample's dispersions from `np.random.seed(i)`, its motor perturbation from a
    fresh `RandomState(i)` (monte_carlo.py:253, motor.py:95-125 / :171-186) and its wind turbulence
    from ANOTHER fresh `RandomState(i)` (monte_carlo.py:263, environment.py:161-198): three streams
    with the same seed, i.e. the same leading normals.  That joint law is kept here: one normal
    vector z per sample feeds all three consumers by position - z[0:14] the 14 dispersion normals
    (position, velocity, attitude, angular velocity, mass, thrust[dead]), z[0:2] (liquid) or z[0:3]
    (solid) the motor multipliers, z[3k + c] the turbulence innovation of knot k, component c.  The two
    uniforms (wind speed, direction) and the dead density draw are independent of it.

    Wind: with a base profile (CSV) -> baseline + AR(1) turbulence + uniform (speed, direction)
    offset (monte_carlo.py:268-280); otherwise the synthetic power-law profile on
    linspace(0, 25000, n_wind_knots) (monte_carlo.py:282-288).  The AR(1) recursion runs in fp64 in one
    HIP kernel (erpl_mc_synth_wind); the table is stored in the working precision.  `planar=True`
    zeroes every out-of-plane input (Set P).  Returns an engine.DeviceBatch resident on `device`."""
    import ctypes as C

    import numpy as np
    reject_overrides(wind_model, "wind_model")
    u = dict(DEFAULT_UNCERTAINTY)
    if uncertainty:
        u.update(uncertainty)
    device = torch.device(device)
    f64 = torch.float64
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    if engine is None:
        from .simulator import shared_engine
        engine = shared_engine(device)

    def col(v):
        return torch.tensor(v, dtype=f64, device=device).view(3, 1)

    use_base = base_wind_profile is not None and base_altitude_profile is not None
    alt_np = (np.asarray(base_altitude_profile, dtype=np.float64) if use_base
              else np.linspace(0, 25000, n_wind_knots))
    K = len(alt_np)
    if K > _abi.MAX_WIND_KNOTS:
        raise _abi.ErplError(f"{K} wind knots exceed the ABI limit {_abi.MAX_WIND_KNOTS}")
    rows = max(3 * K, 14)
    z = torch.randn((rows, n), generator=gen, dtype=f64, device=device)   # the shared normal stream
    uni = torch.rand((2, n), generator=gen, dtype=f64, device=device)

    ic0 = base_initial_conditions
    pos = col(ic0.get("position", [0.0] * 3)) + col(u["initial_position"]) * z[0:3]
    vel = col(ic0.get("velocity", [0.0] * 3)) + col(u["initial_velocity"]) * z[3:6]
    att = col(ic0.get("attitude", [0.0] * 3)) + col(u["initial_attitude"]) * z[6:9]
    omg = col(ic0.get("angular_velocity", [0.0] * 3)) + col(u["initial_angular_velocity"]) * z[9:12]
    mass_mult = 1.0 + u["mass_uncertainty"] * z[12]
    lo, hi = u["wind_speed_range"]
    wind_speed = lo + (hi - lo) * uni[0]
    lo, hi = u["wind_direction_range"]
    wind_dir = lo + (hi - lo) * uni[1]
    if planar:
        keep_v = col([1.0, 0.0, 1.0])
        keep_a = col([0.0, 1.0, 0.0])
        vel = col(ic0.get("velocity", [0.0] * 3)) * (1 - keep_v) + vel * keep_v
        att = col(ic0.get("attitude", [0.0] * 3)) * (1 - keep_a) + att * keep_a
        omg = omg * keep_a
    ic = torch.cat([pos, vel, _euler_to_quaternion(att[0], att[1], att[2]), omg]).contiguous()

    dry = rocket.dry_mass * mass_mult
    prop = rocket.propellant_mass * mass_mult
    rk = torch.stack([dry, prop]).contiguous()

    # motor perturbation (motor.py:95-125 / :171-186) + re-sync of burn time (monte_carlo.py:258-260)
    k_thrust = 1.0 + motor.thrust_uncertainty * z[0]
    if motor_kind(motor) == _abi.MOTOR_SOLID:   # z[1], z[2]: burn-time and impulse multipliers, drawn then overwritten / unused
        mdot = 4.26 * k_thrust
        row0 = k_thrust
        ae = motor.nozzle_exit_area * k_thrust
    else:
        mdot = motor.mass_flow_rate * (1.0 + motor.mass_flow_uncertainty * z[1])
        row0 = motor.thrust_vacuum * k_thrust
        ae = (motor.thrust_vacuum * k_thrust - motor.thrust_sea_level * k_thrust) / 101325.0
    burn = prop / mdot
    mt = torch.stack([row0, ae, mdot, burn]).contiguous()

    # wind: per-knot constants on the host exactly as the bit-exact generator computes them, AR(1) on the device
    sigma, rho, innov = knot_constants(wind_model, alt_np)
    dev64 = lambda v: torch.tensor(np.asarray(v, dtype=np.float64), dtype=f64, device=device)
    sigma_d, rho_d, innov_d = dev64(sigma), dev64([0.0] + list(rho[1:])), dev64([0.0] + list(innov[1:]))
    if use_base:
        base_np = np.array(base_wind_profile, dtype=np.float64)
        if planar:
            base_np[:, 1] = 0.0
        base_d = dev64(base_np).contiguous()
        scale_d = torch.ones((K,), dtype=f64, device=device)
    else:
        base_d = None
        scale_d = dev64([(np.float64(a) / 10.0) ** wind_model.power_law_exponent for a in alt_np])  # environment.py:118-123
    mean_u = (wind_speed * torch.cos(wind_dir)).contiguous()
    mean_v = (torch.zeros_like(wind_speed) if planar else wind_speed * torch.sin(wind_dir)).contiguous()
    g = z[: 3 * K].contiguous()
    if planar:
        g = g.clone()
        g.view(K, 3, n)[:, 1, :] = 0.0      # no cross-wind turbulence either
    wdt = torch.float32 if precision == _abi.PREC_F32 else f64
    wind = torch.empty((K, 3, n), dtype=wdt, device=device)
    st = torch.cuda.current_stream(device)
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    rc = engine.lib.erpl_mc_synth_wind(engine._ctx, n, K, ptr(g), ptr(sigma_d), ptr(rho_d), ptr(innov_d), ptr(base_d),
                                       ptr(scale_d), ptr(mean_u), ptr(mean_v), ptr(wind), int(precision),
                                       C.c_void_p(st.cuda_stream))
    _abi.check(engine.lib, rc, "erpl_mc_synth_wind")
    _check_ranges(ic, rk, mt, mean_u, mean_v)
    return DeviceBatch(ic, rk, mt, dev64(alt_np).contiguous(), wind, precision)
