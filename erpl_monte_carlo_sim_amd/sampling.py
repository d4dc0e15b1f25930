"""Vectorised dispersion + wind-profile synthesis in torch.

The reference draws every sample from a legacy MT19937 `RandomState(i)` in Python loops
(monte_carlo.py:156-179, environment.py:125-265), ~ms per sample.  `flatten.dispersed_batch`
restates that bit for bit and is what the parity sets use.  For the 100 k - 10 M throughput
configurations this module draws the SAME distributions with torch's counter-based generator,
all samples at once, directly into the SoA tensors the kernels read (no host round trip).  It is
not bit-identical to MT19937 and is therefore never used for parity against the reference.
"""
import math

import torch

from . import _abi
from .engine import DeviceBatch
from .flatten import motor_kind

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


def _ar1_turbulence(alt, wind_model, n, gen, device, dtype):
    """AR(1) turbulence over the knots (environment.py:161-198 / :242-263) for n samples:
    returns [K, 3, n]."""
    K = alt.numel()
    sigma = wind_model.turbulence_intensity * torch.exp(-alt / 2000.0)
    dz = torch.clamp(alt[1:] - alt[:-1], min=1e-6)
    rho = torch.clamp(torch.exp(-dz / wind_model.correlation_length), 0.1, 0.95)
    innov = sigma[1:] * torch.sqrt(torch.clamp(1 - rho ** 2, min=0.01))
    comp = torch.tensor([1.0, 1.0, 0.3], dtype=dtype, device=device).view(3, 1)
    turb = torch.empty((K, 3, n), dtype=dtype, device=device)
    turb[0] = sigma[0] * comp * torch.randn((3, n), generator=gen, dtype=dtype, device=device)
    for k in range(1, K):
        g = torch.randn((3, n), generator=gen, dtype=dtype, device=device)
        turb[k] = rho[k - 1] * turb[k - 1] + innov[k - 1] * comp * g
    return turb


def synthetic_dispersions(n, rocket, motor, wind_model, base_initial_conditions, device,
                          precision=_abi.PREC_F32, seed=1234, uncertainty=None,
                          base_altitude_profile=None, base_wind_profile=None, planar=False,
                          n_wind_knots=100):
    """n dispersed samples with the distributions of monte_carlo.py:156-179 / :225-288.

    Wind: with a base profile (CSV) -> baseline + AR(1) turbulence + uniform (speed, direction)
    offset (monte_carlo.py:268-280); otherwise the synthetic power-law profile on
    linspace(0, 25000, n_wind_knots) (monte_carlo.py:282-288).  `planar=True` zeroes every
    out-of-plane input (Set P).  Returns an engine.DeviceBatch resident on `device`."""
    u = dict(DEFAULT_UNCERTAINTY)
    if uncertainty:
        u.update(uncertainty)
    device = torch.device(device)
    f64 = torch.float64
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))

    def normal(shape):
        return torch.randn(shape, generator=gen, dtype=f64, device=device)

    def uniform(lo, hi):
        return lo + (hi - lo) * torch.rand((n,), generator=gen, dtype=f64, device=device)

    def col(v):
        return torch.tensor(v, dtype=f64, device=device).view(3, 1)

    ic0 = base_initial_conditions
    pos = col(ic0.get("position", [0.0] * 3)) + col(u["initial_position"]) * normal((3, n))
    vel = col(ic0.get("velocity", [0.0] * 3)) + col(u["initial_velocity"]) * normal((3, n))
    att = col(ic0.get("attitude", [0.0] * 3)) + col(u["initial_attitude"]) * normal((3, n))
    omg = col(ic0.get("angular_velocity", [0.0] * 3)) + col(u["initial_angular_velocity"]) * normal((3, n))
    mass_mult = 1.0 + u["mass_uncertainty"] * normal((n,))
    wind_speed = uniform(*u["wind_speed_range"])
    wind_dir = uniform(*u["wind_direction_range"])
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
    k_thrust = 1.0 + motor.thrust_uncertainty * normal((n,))
    if motor_kind(motor) == _abi.MOTOR_SOLID:
        normal((n,)); normal((n,))  # burn-time and impulse multipliers: drawn, overwritten/unused
        mdot = 4.26 * k_thrust
        row0 = k_thrust
        ae = motor.nozzle_exit_area * k_thrust
    else:
        mdot = motor.mass_flow_rate * (1.0 + motor.mass_flow_uncertainty * normal((n,)))
        row0 = motor.thrust_vacuum * k_thrust
        ae = (motor.thrust_vacuum * k_thrust - motor.thrust_sea_level * k_thrust) / 101325.0
    burn = prop / mdot
    mt = torch.stack([row0, ae, mdot, burn]).contiguous()

    wdt = torch.float32 if precision == _abi.PREC_F32 else f64
    use_base = base_wind_profile is not None and base_altitude_profile is not None
    if use_base:
        alt = torch.as_tensor(base_altitude_profile, dtype=f64, device=device)
        base = torch.as_tensor(base_wind_profile, dtype=f64, device=device)  # [K, 3]
    else:
        alt = torch.linspace(0, 25000, n_wind_knots, dtype=f64, device=device)
    K = alt.numel()
    turb = _ar1_turbulence(alt.to(wdt), wind_model, n, gen, device, wdt)
    cd, sd = torch.cos(wind_dir).to(wdt), torch.sin(wind_dir).to(wdt)
    ws = wind_speed.to(wdt)
    wind = turb
    if use_base:
        wind += base.to(wdt).view(K, 3, 1)
        wind[:, 0, :] += (ws * cd).view(1, n)
        wind[:, 1, :] += (ws * sd).view(1, n)
    else:
        mean = (alt / 10.0) ** wind_model.power_law_exponent  # environment.py:118-123
        wind[:, 0, :] += mean.to(wdt).view(K, 1) * (ws * cd).view(1, n)
        wind[:, 1, :] += mean.to(wdt).view(K, 1) * (ws * sd).view(1, n)
    if planar:
        wind[:, 1, :] = 0
    return DeviceBatch(ic, rk, mt, alt.contiguous(), wind.contiguous(), precision)
