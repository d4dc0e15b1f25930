"""FlightSimulator — drop-in for simulator.py:9-293 of the reference, backed by the HIP kernels.

Same constructor, same mutable attributes, same `simulate_flight(initial_conditions,
wind_profile=None, altitude_profile=None) -> dict` with the reference's result keys for the state
histories, scalar results and rail-exit diagnostics.  One flight = a batch of one sample through
`erpl_mc_run_batch` in fp64 with full-resolution trajectory capture, followed by
`erpl_mc_extract_histories` for the per-step derived diagnostics of `_extract_results`
(mass/thrust/drag/coefficient/stability histories, simulator.py:496-552); nothing is integrated or
evaluated on the CPU.
"""
import math

import numpy as np
import torch

from . import _abi, flatten
from .engine import DeviceBatch, TrajectoryEngine

_ENGINES = {}


def shared_engine(device=None):
    """One TrajectoryEngine (erpl_ctx) per GPU per process."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = (dev.type, dev.index)
    if key not in _ENGINES:
        _ENGINES[key] = TrajectoryEngine(dev)
    return _ENGINES[key]


def _to_jsonable(obj):
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, (np.floating, np.integer)):
        return obj.item()
    if isinstance(obj, dict):
        return {k: _to_jsonable(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_to_jsonable(v) for v in obj]
    return obj


def _quat_to_euler(q):
    """'xyz' Euler angles of a (w, x, y, z) quaternion (utils.py:139-144 via :46-70)."""
    w, x, y, z = q
    roll = np.arctan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y))
    sinp = 2 * (w * y - z * x)
    pitch = np.copysign(np.pi / 2, sinp) if np.abs(sinp) >= 1 else np.arcsin(sinp)
    yaw = np.arctan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))
    return np.array([roll, pitch, yaw])


class FlightSimulator:
    """6-DOF flight simulator; the integration runs on the GPU (HIP), see csrc/erpl_kernels.inc."""

    def __init__(self, rocket, motor, atmosphere, wind_model, device=None):
        self.rocket, self.motor, self.atmosphere, self.wind_model = rocket, motor, atmosphere, wind_model
        self.max_time = 300.0      # simulator.py:19-22
        self.dt_initial = 0.01
        self.rtol = 1e-4           # unused by the reference's RK4 as well
        self.atol = 1e-7
        self.ground_altitude = 0.0
        self.apogee_detected = False
        self.wind_profile = None
        self.altitude_profile = None
        self.pitch_damping = 20.0  # simulator.py:36-37
        self.yaw_damping = 20.0
        self.parachute_deployed = False
        self.device = device
        self.precision = _abi.PREC_F64

    def _config(self):
        return flatten.config_from_objects(self.rocket, self.motor, self.atmosphere, dt_initial=self.dt_initial,
                                           max_time=self.max_time, pitch_damping=self.pitch_damping,
                                           yaw_damping=self.yaw_damping)

    def simulate_flight(self, initial_conditions, wind_profile=None, altitude_profile=None):
        """Simulate one flight (simulator.py:127-293)."""
        flatten.reject_overrides(self.wind_model, "wind_model")   # the kernels interpolate the table themselves
        eng = shared_engine(self.device)
        eng.set_config(self._config())
        hb = flatten.single_flight_batch(self.rocket, self.motor, initial_conditions, wind_profile, altitude_profile)
        db = DeviceBatch.from_host(hb, eng.device, self.precision)
        dt = min(self.dt_initial, 0.005)
        cap = int(math.ceil(max(self.max_time, 0.0) / dt)) + 8
        summ, status, traj, tlen = eng.run(db, traj_ids=[0], traj_stride=1, traj_cap=cap)
        torch.cuda.synchronize(eng.device)
        s = summ[:, 0].cpu().numpy()
        n = int(tlen[0].item())
        # per-step diagnostic histories (_extract_results loop, simulator.py:511-552), on the GPU
        diag = eng.extract_histories(db, 0, traj[0, :n], float(s[_abi.SUM_RAIL_EXIT_TIME])).cpu().numpy().T
        tr = traj[0, :n].cpu().numpy()            # [n, 15]: absolute time + 14 state
        st = int(status[0].item())
        self.wind_profile, self.altitude_profile = wind_profile, altitude_profile
        self.parachute_deployed = bool(st & _abi.ST_CHUTE)
        rail_time = float(s[_abi.SUM_RAIL_EXIT_TIME])
        states = tr[:, 1:].T                       # (14, n)
        positions, velocities = states[0:3], states[3:6]
        results = {
            "time": tr[:, 0] - rail_time,          # shifted to start at rail exit (:464)
            "position": positions, "velocity": velocities, "quaternion": states[6:10],
            "angular_velocity": states[10:13], "propellant_fraction": states[13],
            "altitude": positions[2], "speed": np.linalg.norm(velocities, axis=0),
            "euler_angles": diag[0:3], "center_of_mass": diag[3], "mass": diag[4],
            "moments_of_inertia": diag[5:8], "thrust": diag[8], "drag": diag[9],
            "cd": diag[10], "cl": diag[11], "cm": diag[12], "cp_location_dynamic": diag[13],
            "stability_margin": diag[14], "angle_of_attack": diag[15], "sideslip_angle": diag[16],
            "cp_location": self.rocket.cp_location,
            "thrust_curve_time": getattr(self.motor, "thrust_curve_time", None),
            "thrust_curve_thrust": getattr(self.motor, "thrust_curve_thrust", None),
            "apogee_time": float(s[_abi.SUM_APOGEE_TIME]),
            "apogee_altitude": float(s[_abi.SUM_APOGEE_ALT]),
            "range": float(s[_abi.SUM_RANGE]),
            "flight_time": float(s[_abi.SUM_FLIGHT_TIME]),
            # extensions
            "first_apogee_altitude": float(s[_abi.SUM_FIRST_APOGEE_ALT]),
            "first_apogee_time": float(s[_abi.SUM_FIRST_APOGEE_TIME]),
            "termination": ("max_time", "ground_impact", "excessive_altitude", "coast_timeout", "apogee")[st & 0xFF],
            "parachute_deployed": self.parachute_deployed,
        }
        # rail-exit diagnostics (:103-123)
        quat0 = states[6:10, 0]
        if hb.k_wind:
            wind_at_exit = np.array([np.interp(positions[2, 0], hb.alt_grid, hb.wind[:, c, 0]) for c in range(3)])
        else:
            wind_at_exit = np.array([0.0, 0.0, 0.0])
        results.update({
            "rail_exit_time": rail_time,
            "rail_exit_position": positions[:, 0].copy(),
            "rail_exit_velocity": velocities[:, 0].copy(),
            "rail_exit_speed": float(s[_abi.SUM_RAIL_EXIT_SPEED]),
            "rail_exit_euler": _quat_to_euler(quat0),
            "rail_exit_angle_of_attack": float(s[_abi.SUM_RAIL_EXIT_AOA]),
            "rail_exit_sideslip": float(s[_abi.SUM_RAIL_EXIT_SIDESLIP]),
            "wind_at_exit": wind_at_exit,
        })
        ic = initial_conditions
        results["initial_conditions"] = {
            "position": [float(v) for v in hb.ic[0:3, 0]],
            "velocity": [float(v) for v in ic.get("velocity", [0.0, 0.0, 0.0])],
            "attitude": ic.get("attitude", [0.0, 0.0, 0.0]),
            "angular_velocity": [float(v) for v in hb.ic[10:13, 0]],
        }
        results["rocket_parameters"] = {k: _to_jsonable(v) for k, v in self.rocket.__dict__.items()}
        results["motor_parameters"] = {k: _to_jsonable(v) for k, v in self.motor.__dict__.items()}
        results["simulation_assumptions"] = {"max_time": self.max_time, "dt_initial": self.dt_initial,
                                             "rtol": self.rtol, "atol": self.atol, "rail_length": 18.288}
        if wind_profile is not None and altitude_profile is not None:
            results["wind_profile"] = wind_profile
            results["altitude_profile"] = altitude_profile
        return results
