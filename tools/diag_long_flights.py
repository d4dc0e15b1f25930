#!/usr/bin/env python3
"""Which dispersed samples fly long?  (start a batch's long trajectories first.)  Set R through the CPU
oracle: physics step count of every sample against what is known about it before the flight kernel starts."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import helpers as H
import oracle as orc
from erpl_monte_carlo_sim_amd import _abi, flatten, models

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
P = flatten.generate_parameter_arrays(H.UNCERTAINTY, n)
hb = flatten.dispersed_batch(models.Rocket(), models.LiquidMotor(), models.WindModel(), H.EXAMPLE_IC, P)
cfg = H.make_config("liquid")
out = orc.run_batch(cfg, hb)
summ, status = out[0], out[1]
steps = summ[_abi.SUM_STEPS] - 0
phys = steps.copy()
print("n", n, "steps: mean %.0f median %.0f p99 %.0f max %.0f" % (phys.mean(), np.median(phys), np.percentile(phys, 99), phys.max()))
end = status & 0xFF
for e in range(5):
    m = end == e
    if m.any():
        print(" end", e, "count", int(m.sum()), "steps median %.0f max %.0f" % (np.median(phys[m]), phys[m].max()))
feat = {
    "wind_speed": P["wind_speed"], "mass_mult": P["mass_multiplier"], "thrust_mult": P["thrust_multiplier"], "density_mult": P["density_multiplier"],
    "|att_off|": np.linalg.norm(P["initial_attitude_offset"], axis=1), "|omega_off|": np.linalg.norm(P["initial_angular_velocity_offset"], axis=1),
    "|vel_off|": np.linalg.norm(P["initial_velocity_offset"], axis=1),
    "att_pitch": P["initial_attitude_offset"][:, 1], "att_yaw": P["initial_attitude_offset"][:, 2], "att_roll": P["initial_attitude_offset"][:, 0],
    "om_x": P["initial_angular_velocity_offset"][:, 0], "om_y": P["initial_angular_velocity_offset"][:, 1], "om_z": P["initial_angular_velocity_offset"][:, 2],
    "rail_exit_speed": summ[_abi.SUM_RAIL_EXIT_SPEED], "rail_exit_aoa": summ[_abi.SUM_RAIL_EXIT_AOA], "rail_exit_sideslip": summ[_abi.SUM_RAIL_EXIT_SIDESLIP],
    "wind0": np.hypot(hb.wind[0, 0], hb.wind[0, 1]) if hb.wind.shape[0] == 100 else np.hypot(hb.wind[0, 0], hb.wind[1, 0]),
}
print("wind array shape", hb.wind.shape)
# non-finite samples are dragged to max_time by table: physics steps are what the kernel integrates
long_ = phys > 3 * np.median(phys)
print("long (> 3 x median):", int(long_.sum()), "of", n)
for k, v in feat.items():
    v = np.asarray(v, dtype=np.float64)
    ok = np.isfinite(v)
    c = np.corrcoef(v[ok], np.log(phys[ok]))[0, 1]
    print("%-20s corr(log steps) %+.3f   long: median %.4g (all: %.4g)   long range [%.4g, %.4g]  all range [%.4g, %.4g]" % (
        k, c, np.median(v[long_ & ok]) if (long_ & ok).any() else float("nan"), np.median(v[ok]),
        v[long_ & ok].min() if (long_ & ok).any() else float("nan"), v[long_ & ok].max() if (long_ & ok).any() else float("nan"), v[ok].min(), v[ok].max()))
np.savez("/tmp/long_flights.npz", steps=phys, end=end, **{k.replace("|", "_"): np.asarray(v) for k, v in feat.items()})
