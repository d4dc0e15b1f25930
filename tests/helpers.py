"""Shared helpers: golden-fixture loading and batch construction (no reference access)."""
import json
import os

import numpy as np

from erpl_monte_carlo_sim_amd import _abi, flatten, models

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

EXAMPLE_IC = {
    "position": [0.0, 0.0, 10.0],
    "velocity": [0, 0, 0.0],
    "attitude": [0.0, -np.pi / 2 + 0.02, 0.0],
    "angular_velocity": [0.0, 0.0, 0.0],
}

CSV_ALT = np.array([0.0, 5000.0, 10000.0, 15000.0, 20000.0, 25000.0])
CSV_WIND = np.array([[2.0, 0, 0], [5, 1, 0], [8, 2, 0], [10, 2, 0], [12, 3, 0], [15, 3, 0]])

UNCERTAINTY = {
    "initial_position": [0.0, 0.0, 0.0], "initial_velocity": [0.1, 0.1, 0.1],
    "initial_attitude": [0.005, 0.005, 0.005], "initial_angular_velocity": [0.005, 0.005, 0.005],
    "mass_uncertainty": 0.02, "thrust_uncertainty": 0.03, "wind_speed_range": [0.0, 5.0],
    "wind_direction_range": [0.0, 2 * np.pi], "atmospheric_density_uncertainty": 0.05,
}


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def load_flights(name):
    idx = load_json(name + ".json")
    arr = np.load(os.path.join(GOLDEN, name + ".npz"))
    return idx, arr


def make_motor(kind):
    return models.SolidMotor() if kind in ("solid", 1) else models.LiquidMotor()


def make_config(kind, **kw):
    return flatten.config_from_objects(models.Rocket(), make_motor(kind), models.StandardAtmosphere(), **kw)


def batch_from_golden(entries, arr):
    """HostBatch from golden flight entries that share motor kind and wind grid."""
    n = len(entries)
    e0 = entries[0]
    k = arr[e0["tag"] + "_altitude_profile"].shape[0] if (e0["tag"] + "_altitude_profile") in arr else 0
    b = flatten.HostBatch(n, k)
    if k:
        b.alt_grid[:] = arr[e0["tag"] + "_altitude_profile"]
    for i, e in enumerate(entries):
        inp = e["inputs"]
        b.ic[0:3, i] = inp["position"]
        b.ic[3:6, i] = inp["velocity"]
        b.ic[6:10, i] = inp["quaternion"]
        b.ic[10:13, i] = inp["angular_velocity"]
        b.rocket[:, i] = [inp["dry_mass"], inp["propellant_mass"]]
        if inp["motor_kind"] == 1:
            base = models.SolidMotor().thrust_curve_thrust
            cur = arr[e["tag"] + "_thrust_curve_thrust"]
            j = int(np.argmax(base))
            mult = cur[j] / base[j]
            # the multiplier must reproduce the golden's scaled curve bit for bit
            if not np.array_equal(base * mult, cur):
                cands = [cur[m] / base[m] for m in range(len(base)) if base[m] != 0]
                mult = next(c for c in cands if np.array_equal(base * c, cur))
            thrust = mult
        else:
            thrust = inp["thrust_vacuum"]
        b.motor[:, i] = [thrust, inp["nozzle_exit_area"], inp["mass_flow_rate"], inp["burn_time"]]
        if k:
            b.wind[:, :, i] = arr[e["tag"] + "_wind_profile"]
    return b


def group_flights(idx):
    """Group golden entries by (motor_kind, wind-key) so each group forms one batch."""
    groups = {}
    for e in idx:
        key = e["key"]
        kind = e["inputs"]["motor_kind"]
        if isinstance(key, list) and len(key) == 4:
            g = (kind, key[1], key[2])
        elif isinstance(key, list):
            g = (kind, "planar")
        else:
            g = (kind, key)
        groups.setdefault(g, []).append(e)
    return groups


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
