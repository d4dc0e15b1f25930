#!/usr/bin/env python3
"""Golden-vector generator (TEST INFRASTRUCTURE, runs only in the build container).

Imports the Python reference from /root/reference/rocket_simulation (read-only, never copied)
and records inputs/outputs of the hot path as small data fixtures under tests/golden/.
The fixtures are DATA (flattened inputs + expected outputs); no reference source text is stored.

Usage:  PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python3 oracle/gen_golden.py [--jobs 8]

What is captured (G4, the sensitivity record, is oracle/gen_sensitivity.py -> sensitivity.json):
  kat.json            function-level known answers (atmosphere, gravity, mass props, aero
                      coefficients, np.interp edge cases, wind lookup, RHS `_rocket_dynamics`)
  params.json         dispersion stream `_generate_parameter_samples` (seed=i) and the seed-42
                      stream, perturbed motors, perturbed CSV / synthetic wind profiles
  flights_*.npz       flattened per-sample inputs (as seen by FlightSimulator.simulate_flight,
                      captured from inside the reference call) + scalar summaries + decimated
                      state histories
  stats.json          `_analyze_results` on synthetic summaries (outlier filter + statistics)

The flattened input layout is the one the C ABI (include/erpl_mc.h) consumes.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time
import warnings

import numpy as np

REF = "/root/reference/rocket_simulation"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")

import environment as ref_env  # noqa: E402
import monte_carlo as ref_mc  # noqa: E402
import motor as ref_motor  # noqa: E402
import rocket as ref_rocket  # noqa: E402
import simulator as ref_sim  # noqa: E402
import utils as ref_utils  # noqa: E402


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def f(x):
    """float -> python float (json keeps full repr precision)."""
    return float(x)


def fl(a):
    return [float(v) for v in np.asarray(a, dtype=np.float64).ravel()]


# ----------------------------------------------------------------------------------------
# flattening of the reference objects, taken from INSIDE the reference's simulate_flight call
# ----------------------------------------------------------------------------------------
def flatten_call(sim, ic, wind_profile, altitude_profile):
    e = ic.get("attitude", [0.0, 0.0, 0.0])
    quat = ref_utils.euler_to_quaternion(e[0], e[1], e[2])
    m = sim.motor
    solid = isinstance(m, ref_motor.SolidMotor)
    d = {
        "motor_kind": 1 if solid else 0,
        "position": fl(ic.get("position", [0, 0, 0])),
        "velocity": fl(ic.get("velocity", [0, 0, 0])),
        "attitude": fl(e),
        "quaternion": fl(quat),
        "angular_velocity": fl(ic.get("angular_velocity", [0, 0, 0])),
        "dry_mass": f(sim.rocket.dry_mass),
        "propellant_mass": f(sim.rocket.propellant_mass),
        "nozzle_exit_area": f(m.nozzle_exit_area),
        "mass_flow_rate": f(m.mass_flow_rate),
        "burn_time": f(m.burn_time),
    }
    if solid:
        d["thrust_curve_time"] = fl(m.thrust_curve_time)
        d["thrust_curve_thrust"] = fl(m.thrust_curve_thrust)
    else:
        d["thrust_vacuum"] = f(m.thrust_vacuum)
    if wind_profile is not None and altitude_profile is not None:
        d["altitude_profile"] = fl(altitude_profile)
        d["wind_profile"] = np.asarray(wind_profile, dtype=np.float64).tolist()
    return d


_orig_simulate = ref_sim.FlightSimulator.simulate_flight


def _capturing_simulate(self, initial_conditions, wind_profile=None, altitude_profile=None):
    cap = flatten_call(self, initial_conditions, wind_profile, altitude_profile)
    res = _orig_simulate(self, initial_conditions, wind_profile, altitude_profile)
    res["_captured"] = cap
    return res


ref_sim.FlightSimulator.simulate_flight = _capturing_simulate


def summarize(res, decim, diag=False):
    """Scalar summary + decimated history of one reference result dict."""
    t = np.asarray(res["time"])  # shifted by rail time
    pos = np.asarray(res["position"])
    vel = np.asarray(res["velocity"])
    alt = pos[2]
    n = len(t)
    # first-descent apogee: first post-step index k>=1 with z>1000 and vz<0 (simulator.py:247)
    k_first = -1
    for k in range(1, n):
        if alt[k] > 1000.0 and vel[2, k] < 0:
            k_first = k
            break
    if k_first >= 0:
        seg = alt[: k_first + 1]
        j = int(np.argmax(seg))
        first_apogee, first_apogee_time = float(seg[j]), float(t[j])
    else:
        j = int(np.argmax(alt))
        first_apogee, first_apogee_time = float(alt[j]), float(t[j])
    states = np.vstack([
        pos, vel, np.asarray(res["quaternion"]), np.asarray(res["angular_velocity"]),
        np.asarray(res["propellant_fraction"])[None, :],
    ])  # (14, n)
    idx = np.arange(0, n, decim)
    if idx[-1] != n - 1:
        idx = np.append(idx, n - 1)
    out = {
        "n_steps": n - 1,
        "apogee_altitude": float(res["apogee_altitude"]),
        "apogee_time": float(res["apogee_time"]),
        "range": float(res["range"]),
        "flight_time": float(res["flight_time"]),
        "first_apogee": first_apogee,
        "first_apogee_time": first_apogee_time,
        "first_descent_step": k_first,
        "rail_exit_time": float(res["rail_exit_time"]),
        "rail_exit_speed": float(res["rail_exit_speed"]),
        "rail_exit_position": fl(res["rail_exit_position"]),
        "rail_exit_velocity": fl(res["rail_exit_velocity"]),
        "rail_exit_euler": fl(res["rail_exit_euler"]),
        "rail_exit_angle_of_attack": float(res["rail_exit_angle_of_attack"]),
        "rail_exit_sideslip": float(res["rail_exit_sideslip"]),
        "wind_at_exit": fl(res["wind_at_exit"]),
        "final_state": fl(states[:, -1]),
        "hist_index": idx.astype(np.int64),
        "hist_time_abs": (t[idx] + float(res["rail_exit_time"])),
        "hist_state": states[:, idx].copy(),
    }
    if diag:
        # per-step diagnostic histories of _extract_results (simulator.py:496-552), same decimation
        rows = [res["euler_angles"][0], res["euler_angles"][1], res["euler_angles"][2], res["center_of_mass"],
                res["mass"], res["moments_of_inertia"][0], res["moments_of_inertia"][1],
                res["moments_of_inertia"][2], res["thrust"], res["drag"], res["cd"], res["cl"], res["cm"],
                res["cp_location_dynamic"], res["stability_margin"], res["angle_of_attack"],
                res["sideslip_angle"]]
        out["hist_diag"] = np.vstack([np.asarray(r, dtype=np.float64)[idx] for r in rows])
    return out


# ----------------------------------------------------------------------------------------
# flight jobs (run in worker processes)
# ----------------------------------------------------------------------------------------
EXAMPLE_IC = {
    "position": [0.0, 0.0, 10.0],
    "velocity": [0, 0, 0.0],
    "attitude": [0.0, -np.pi / 2 + 0.02, 0.0],
    "angular_velocity": [0.0, 0.0, 0.0],
}


def csv_profile():
    return ref_env.WindModel().load_wind_profile_from_csv(os.path.join(REF, "sample_wind.csv"))


def make_motor(kind):
    return ref_motor.SolidMotor() if kind == "solid" else ref_motor.LiquidMotor()


def job_named(spec):
    name, kind, wind = spec
    alt, w = csv_profile()
    if wind == "none":
        alt, w = None, None
    elif wind == "planar":
        w = w.copy()
        w[:, 1] = 0.0
    sim = ref_sim.FlightSimulator(ref_rocket.Rocket(), make_motor(kind),
                                  ref_env.StandardAtmosphere(), ref_env.WindModel())
    with quiet():
        res = sim.simulate_flight(dict(EXAMPLE_IC), w, alt)
    return name, res["_captured"], summarize(res, 100, diag=True)


def make_analyzer(kind, base):
    with quiet():
        an = ref_mc.MonteCarloAnalyzer(ref_rocket.Rocket(), make_motor(kind),
                                       ref_env.StandardAtmosphere(), ref_env.WindModel())
    if base == "csv":
        an.base_altitude_profile, an.base_wind_profile = csv_profile()
    return an


def job_mc(spec):
    """One `_run_single_simulation` of the reference (inputs captured inside it)."""
    kind, base, stream, i = spec
    an = make_analyzer(kind, base)
    with quiet():
        if stream == "seed_i":
            params = an._generate_parameter_samples(i + 1)[i]
        else:
            params = an._generate_parameter_samples_vectorized(i + 1)[i]
        res = an._run_single_simulation(dict(EXAMPLE_IC), params, i)
    return (kind, base, stream, i), res["_captured"], summarize(res, 100)


def job_planar(spec):
    """Set P: reference-drawn dispersion with every out-of-plane input zeroed, so
    sideslip stays identically 0 and the flight is healthy.  The reference's own objects are
    perturbed by its own methods; only the IC/wind handed to simulate_flight are planarised."""
    kind, i = spec
    an = make_analyzer(kind, "csv")
    with quiet():
        params = an._generate_parameter_samples(i + 1)[i]
        rocket = an._perturb_rocket(params)
        motor = an._perturb_motor(params)
        motor.propellant_mass = rocket.propellant_mass
        motor.burn_time = motor.propellant_mass / motor.mass_flow_rate
        wind = an.wind_model.perturb_wind_profile(
            an.base_altitude_profile, an.base_wind_profile,
            random_state=np.random.RandomState(params["random_seed"]))
        wind[:, 0] += params["wind_speed"] * np.cos(params["wind_direction"])
        wind[:, 1] = 0.0
        ic = {
            "position": np.array(EXAMPLE_IC["position"]) + params["initial_position_offset"],
            "velocity": np.array(EXAMPLE_IC["velocity"]) + params["initial_velocity_offset"] * [1, 0, 1],
            "attitude": np.array(EXAMPLE_IC["attitude"]) + params["initial_attitude_offset"] * [0, 1, 0],
            "angular_velocity": params["initial_angular_velocity_offset"] * [0, 1, 0],
        }
        sim = ref_sim.FlightSimulator(rocket, motor, an.atmosphere, an.wind_model)
        res = sim.simulate_flight(ic, wind, an.base_altitude_profile)
    return (kind, i), res["_captured"], summarize(res, 200)


def pack_flights(items, path):
    """items: list of (key, captured, summary) -> one compressed npz (+ json index)."""
    arrays, index = {}, []
    for n, (key, cap, summ) in enumerate(items):
        tag = f"f{n:03d}"
        entry = {"key": key if isinstance(key, str) else list(key), "tag": tag, "inputs": {}, "summary": {}}
        for k, v in cap.items():
            if k in ("wind_profile", "altitude_profile", "thrust_curve_time", "thrust_curve_thrust"):
                arrays[f"{tag}_{k}"] = np.asarray(v, dtype=np.float64)
            else:
                entry["inputs"][k] = v
        for k, v in summ.items():
            if k.startswith("hist_"):
                arrays[f"{tag}_{k}"] = np.asarray(v)
            else:
                entry["summary"][k] = v
        index.append(entry)
    np.savez_compressed(path + ".npz", **arrays)
    with open(path + ".json", "w") as fh:
        json.dump(index, fh, indent=1)


# ----------------------------------------------------------------------------------------
# function-level KATs
# ----------------------------------------------------------------------------------------
def gen_kat():
    rng = np.random.RandomState(20251004)
    kat = {}
    rk = ref_rocket.Rocket()
    atm = ref_env.StandardAtmosphere()
    wm = ref_env.WindModel()
    kat["constants"] = {
        "cp_location": f(rk.cp_location), "reference_area": f(rk.reference_area),
        "liquid": {k: f(getattr(ref_motor.LiquidMotor(), k)) for k in
                   ("thrust_vacuum", "thrust_sea_level", "mass_flow_rate", "propellant_mass",
                    "nozzle_exit_area", "burn_time", "total_impulse")},
        "solid": {**{k: f(getattr(ref_motor.SolidMotor(), k)) for k in
                     ("thrust_vacuum", "thrust_sea_level", "mass_flow_rate", "propellant_mass",
                      "nozzle_exit_area", "burn_time", "total_impulse", "average_thrust")},
                  "thrust_curve_time": fl(ref_motor.SolidMotor().thrust_curve_time),
                  "thrust_curve_thrust": fl(ref_motor.SolidMotor().thrust_curve_thrust)},
    }
    # atmosphere + gravity
    alts = [-100.0, 0.0, 5000.0, 11000.0, 11000.000001, 15000.0, 20000.0, 20000.000001, 22000.0,
            25000.0, 25000.000001, 28000.0, 30000.0, 32000.0, 32000.000001, 40000.0, 49357.0,
            49358.0, 50000.0, 99999.0, 150000.0, -5000.0]
    alts += list(rng.uniform(-200.0, 60000.0, 40))
    rows = []
    for h in alts:
        p = atm.get_properties(h)
        rows.append([f(h), f(p["temperature"]), f(p["pressure"]), f(p["density"]),
                     f(p["speed_of_sound"]), f(atm.get_gravity(h))])
    kat["atmosphere"] = rows
    # mass properties (default + perturbed masses)
    rows = []
    for pf in [1.0, 0.37, 0.0, 0.9416346456692913, 2.5e-4] + list(rng.uniform(0, 1, 6)):
        for mult in (1.0, 1.01522075450294):
            r2 = ref_rocket.Rocket()
            r2.dry_mass *= mult
            r2.propellant_mass *= mult
            mp = r2.get_mass_properties(pf)
            rows.append([f(pf), f(r2.dry_mass), f(r2.propellant_mass), f(mp["mass"]),
                         f(mp["center_of_mass"]), f(mp["Ixx"]), f(mp["Iyy"]), f(mp["Izz"])])
    kat["mass_properties"] = rows
    # aero coefficients
    cases = [(0.3, 0.02, -0.01, 0.9, True), (0.95, 0.1, 0.05, 0.5, True),
             (1.7, 0.4, -0.3, 0.0, False), (3.5, -0.9, 0.2, 0.2, True),
             (0.0, 0.0, 0.0, 1.0, True), (1.0, 0.26179938779914946, 0.0, 0.5, True),
             (0.5, -0.2617993877991495, 0.1, 0.3, False), (0.8, 3.0, -1.2, 0.1, True),
             (1.2, -2.5, 1.5, 0.0, False), (3.0, 0.05, 0.0, 0.0, True), (2.0, 1e-9, -1e-9, 0.7, True)]
    for _ in range(48):
        cases.append((float(rng.uniform(0, 4)), float(rng.normal(0, 0.4)), float(rng.normal(0, 0.3)),
                      float(rng.uniform(0, 1)), bool(rng.randint(2))))
    rows = []
    for (M, a, b, pf, pw) in cases:
        mp = rk.get_mass_properties(pf)
        c = rk.get_aerodynamic_coefficients(M, a, b, mp, power_on=pw)
        rows.append([M, a, b, pf, 1.0 if pw else 0.0, f(c["cd"]), f(c["cl"]), f(c["cy"]),
                     f(c["cpitch"]), f(c["cyaw"]), f(c["cp"]), f(c["cn"])])
    kat["aero"] = rows
    # np.interp edge cases on the reference tables + wind lookup
    alt, w = csv_profile()
    rows = []
    for h in [3456.0, -5.0, 30000.0, 0.0, 5000.0, 25000.0, 24999.999, float("inf"), 12500.0, 1e-300]:
        v = wm.get_wind_at_altitude(h, w, alt)
        rows.append([f(h)] + fl(v))
    kat["wind_csv"] = {"altitude": fl(alt), "wind": w.tolist(), "lookups": rows}
    rows = []
    for M in [0.0, 0.5, 0.8, 1.0, 1.2, 1.5, 2.0, 3.0, 3.0000001, 10.0, 0.25, 0.99, 1.01, 2.5]:
        rows.append([M, f(ref_utils.interpolate_1d(M, rk.Cd_data["mach"], rk.Cd_data["cd0"])),
                     f(ref_utils.interpolate_1d(M, rk.Cd_data["mach"], rk.Cd_data["cda"])),
                     f(rk.get_dynamic_cp(M))])
    kat["mach_tables"] = rows
    sm = ref_motor.SolidMotor()
    lm = ref_motor.LiquidMotor()
    rows = []
    for t in [-0.1, 0.0, 0.1, 0.2, 0.87, 1.0, 4.9, 12.0, 14.5, 14.9, 15.0, 15.0001, 20.0]:
        for P in (101325.0, 54019.9, 0.0):
            rows.append([t, P, f(sm.get_thrust(t, P)), f(lm.get_thrust(t, P)),
                         f(sm.get_mass_flow_rate(t)), f(lm.get_mass_flow_rate(t)),
                         f(sm.get_propellant_remaining(t)), f(lm.get_propellant_remaining(t))])
    kat["motors"] = rows
    # RHS KATs
    q_k = ref_utils.euler_to_quaternion(0.01, -np.pi / 2 + 0.02, 0.03)
    kat["euler_to_quaternion"] = [[0.01, -np.pi / 2 + 0.02, 0.03] + fl(q_k)]
    for e in rng.normal(0, 1.0, (6, 3)):
        kat["euler_to_quaternion"].append(fl(e) + fl(ref_utils.euler_to_quaternion(*e)))
    rhs = []

    def rhs_case(kind, t, state, wind_on, chute_before):
        sim = ref_sim.FlightSimulator(ref_rocket.Rocket(), make_motor(kind), atm, wm)
        if wind_on:
            sim.wind_profile, sim.altitude_profile = w, alt
        sim.parachute_deployed = bool(chute_before)
        d = sim._rocket_dynamics(t, np.array(state, dtype=np.float64))
        rhs.append({"motor": kind, "t": f(t), "state": fl(state), "wind": int(wind_on),
                    "chute_before": int(chute_before), "chute_after": int(sim.parachute_deployed),
                    "deriv": fl(d)})

    def st(z, v, pf, om=(0.01, -0.02, 0.03), q=q_k, xy=(12.0, -7.0)):
        return [xy[0], xy[1], z, v[0], v[1], v[2], q[0], q[1], q[2], q[3], om[0], om[1], om[2], pf]

    for kind in ("liquid", "solid"):
        rhs_case(kind, 6.0, st(3456.0, (15, -4, 310), 0.6), 1, 0)
        rhs_case(kind, 80.0, st(26000.0, (40, 5, -20), 3e-4), 1, 0)
        rhs_case(kind, 150.0, st(450.0, (3, 1, -60), 3e-4), 1, 0)
        rhs_case(kind, 150.0, st(450.0, (3, 1, -60), 3e-4), 1, 1)
        rhs_case(kind, 0.5, st(30.0, (0.5, 0.0, 20.0), 0.97), 0, 0)
        rhs_case(kind, 14.9, st(9000.0, (30, 2, 600), 1e-4), 1, 0)   # burn-out clamp branch
        rhs_case(kind, 3.0, st(800.0, (0, 0, 0), 0.8), 0, 0)          # q_dynamic == 0 branch
        rhs_case(kind, 200.0, st(100.0, (0, 0, 0), 0.0), 0, 1)        # chute, rel_speed == 0
    for _ in range(40):
        kind = "solid" if rng.randint(2) else "liquid"
        e = rng.normal(0, 1.0, 3) * [0.3, 0.5, 0.3] + [0, -np.pi / 2, 0]
        q = ref_utils.euler_to_quaternion(*e) * rng.uniform(0.97, 1.03)  # un-normalised on purpose
        z = float(rng.choice([rng.uniform(0, 11000), rng.uniform(11000, 20000),
                              rng.uniform(20000, 25000), rng.uniform(25000, 32000),
                              rng.uniform(32000, 60000)]))
        v = rng.normal(0, 1, 3) * [80, 80, 400]
        om = rng.normal(0, 0.2, 3)
        pf = float(rng.choice([rng.uniform(0, 1), 0.0, 1e-4]))
        t = float(rng.uniform(0, 30))
        rhs_case(kind, t, st(z, v, pf, om, q), int(rng.randint(2)), 0)
    kat["rhs"] = rhs
    return kat


def gen_params():
    out = {}
    an = make_analyzer("liquid", "csv")
    with quiet():
        samples = an._generate_parameter_samples(64)
        samples42 = an._generate_parameter_samples_vectorized(64)

    def ser(s):
        return {k: (fl(v) if isinstance(v, np.ndarray) else (int(v) if k == "random_seed" else f(v)))
                for k, v in s.items()}

    out["seed_i"] = [ser(s) for s in samples]
    out["seed_42"] = [ser(s) for s in samples42]
    liq, sol = [], []
    for i in range(16):
        m = ref_motor.LiquidMotor().perturb_for_monte_carlo(np.random.RandomState(i))
        liq.append({k: f(getattr(m, k)) for k in
                    ("thrust_vacuum", "thrust_sea_level", "mass_flow_rate", "nozzle_exit_area",
                     "burn_time", "propellant_mass")})
        m = ref_motor.SolidMotor().perturb_for_monte_carlo(np.random.RandomState(i))
        d = {k: f(getattr(m, k)) for k in ("mass_flow_rate", "nozzle_exit_area", "burn_time",
                                           "average_thrust", "thrust_vacuum", "total_impulse")}
        d["thrust_curve_thrust"] = fl(m.thrust_curve_thrust)
        sol.append(d)
    out["liquid_perturbed"] = liq
    out["solid_perturbed"] = sol
    alt, w = csv_profile()
    wm = ref_env.WindModel()
    out["csv_perturbed"] = [wm.perturb_wind_profile(alt, w, np.random.RandomState(i)).tolist()
                            for i in range(4)]
    grid = np.linspace(0, 25000, 100)
    syn = []
    for i in range(4):
        s = samples[i]
        syn.append(wm.generate_stochastic_profile(grid, s["wind_speed"], s["wind_direction"],
                                                  random_state=np.random.RandomState(i)).tolist())
    out["synthetic_profiles"] = syn
    return out


def gen_stats():
    """`_analyze_results` (outlier filter + statistics) on synthetic per-sample summaries."""
    rng = np.random.RandomState(7)
    an = make_analyzer("liquid", "none")
    n = 200
    apo = rng.normal(25000, 1500, n)
    rngs = np.abs(rng.normal(3000, 2000, n))
    ft = rng.normal(205, 6, n)
    apo[3], apo[17], apo[50] = 9.0e4, 50.0, np.nan
    rngs[5], rngs[60] = 3.0e5, np.inf
    ft[9], ft[11] = 700.0, np.nan
    apo[70] = 88100.0  # between 80 km bound and 1.2x energy bound
    with quiet():
        params = an._generate_parameter_samples(n)
        results = [{"apogee_altitude": apo[i], "range": rngs[i], "flight_time": ft[i],
                    "simulation_id": i, "parameters": params[i]} for i in range(n)]
        results[20] = None
        analysis = an._analyze_results(results)
    out = {"inputs": {"apogee_altitude": fl(apo), "range": fl(rngs), "flight_time": fl(ft),
                      "none_index": 20},
           "n_samples": analysis["n_samples"], "n_failed": analysis["n_failed"],
           "n_outliers": analysis["n_outliers"],
           "apogee_altitude": analysis["apogee_altitude"], "range": analysis["range"],
           "flight_time": analysis["flight_time"],
           "valid_ids": [r["simulation_id"] for r in analysis["results"]],
           "outlier_ids": [r["simulation_id"] for r in analysis["outliers"]],
           "outlier_reasons": [r["outlier_reasons"] for r in analysis["outliers"]],
           "parameter_ranges_observed": analysis["parameter_ranges_observed"]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=8)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    t0 = time.time()
    if not args.only or "kat" in args.only:
        with open(os.path.join(OUT, "kat.json"), "w") as fh:
            json.dump(gen_kat(), fh, indent=1)
        with open(os.path.join(OUT, "params.json"), "w") as fh:
            json.dump(gen_params(), fh, indent=1)
        with open(os.path.join(OUT, "stats.json"), "w") as fh:
            json.dump(gen_stats(), fh, indent=1)
        print("kat/params/stats done", time.time() - t0, flush=True)
    if args.only and "flights" not in args.only and "named" not in args.only:
        return
    from multiprocessing import Pool
    named = [("liquid_nowind", "liquid", "none"), ("liquid_planar_csv", "liquid", "planar"),
             ("solid_nowind", "solid", "none"), ("liquid_csv_nominal", "liquid", "csv"),
             ("solid_csv_nominal", "solid", "csv")]
    mc = ([("liquid", "csv", "seed_i", i) for i in range(32)]
          + [("liquid", "csv", "seed_42", i) for i in range(32)]
          + [("liquid", "none", "seed_i", i) for i in range(16)]
          + [("solid", "csv", "seed_i", i) for i in range(8)]
          + [("solid", "none", "seed_i", i) for i in range(8)])
    planar = [("liquid", i) for i in range(8)] + [("solid", i) for i in range(4)]
    if args.only == "named":
        with Pool(args.jobs) as pool:
            pack_flights(pool.map(job_named, named, chunksize=1), os.path.join(OUT, "flights_named"))
        print("named done", time.time() - t0, flush=True)
        return
    with Pool(args.jobs) as pool:
        r_pl = pool.map_async(job_planar, planar, chunksize=1)
        r_nm = pool.map_async(job_named, named, chunksize=1)
        r_mc = pool.map_async(job_mc, mc, chunksize=1)
        pack_flights(r_nm.get(), os.path.join(OUT, "flights_named"))
        print("named done", time.time() - t0, flush=True)
        pack_flights(r_mc.get(), os.path.join(OUT, "flights_mc"))
        print("mc done", time.time() - t0, flush=True)
        pack_flights(r_pl.get(), os.path.join(OUT, "flights_planar"))
        print("planar done", time.time() - t0, flush=True)


if __name__ == "__main__":
    main()
