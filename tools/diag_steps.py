#!/usr/bin/env python3
"""Diagnostic: step-length distribution and per-wave imbalance of a synthetic workload (GPU)."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from erpl_monte_carlo_sim_amd import _abi, flatten, models, sampling
from erpl_monte_carlo_sim_amd.engine import TrajectoryEngine
import bench as B

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=131072)
ap.add_argument("--precision", default="f32")
ap.add_argument("--planar", action="store_true")
ap.add_argument("--apogee", action="store_true")
ap.add_argument("--wind", default="syn", choices=["syn", "csv", "none"])
ap.add_argument("--chunk", type=int, default=0)
ap.add_argument("--waves", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda", 0)
rocket, motor, atm, wm = models.Rocket(), models.LiquidMotor(), models.StandardAtmosphere(), models.WindModel()
eng = TrajectoryEngine(dev, lib_path=os.environ.get("ERPL_LIB")); eng.set_config(flatten.config_from_objects(rocket, motor, atm))
prec = _abi.PRECISIONS[a.precision]
kw = dict(base_altitude_profile=B.CSV_ALT, base_wind_profile=B.CSV_WIND) if a.wind == "csv" else {}
db = sampling.synthetic_dispersions(a.n, rocket, motor, wm, B.EXAMPLE_IC, dev, precision=prec, seed=1234, planar=a.planar, **kw)
if a.wind == "none":
    db.wind = None; db.alt_grid = None; db.k_wind = 0
eng.set_profiling(True)
eng.set_chunk(a.chunk)
eng.set_waves_per_simd(a.waves)
flags = _abi.FLAG_STOP_AT_APOGEE if a.apogee else 0
for _ in range(2):
    s, st = eng.run(db, flags=flags)
torch.cuda.synchronize()
rail, fl = eng.last_kernel_ms()
phys, wi = eng.last_stats()
s = s.cpu().numpy(); st = st.cpu().numpy()
steps = s[_abi.SUM_STEPS]
ff = ((st & 0xFF) == 0) & ((st & _abi.ST_NAN) != 0)
ps = np.where(ff, np.nan, steps)
print(f"chunk={a.chunk} n={a.n} prec={a.precision} flight_ms={fl:.2f} rail_ms={rail:.3f} physics_steps={phys:.0f} wave_iters={wi:.0f} util={phys/(64*wi):.3f}")
print(f"  per wave-iteration (2048-wave-equivalent): {fl*1e3/(wi/ (a.n/64)):.3f} us per step per wave if all waves concurrent")
print("  fast-forwarded (NaN) fraction:", ff.mean())
q = np.nanpercentile(ps, [1, 10, 50, 90, 95, 98, 99, 99.9, 100])
print("  physics-lane step percentiles 1/10/50/90/95/98/99/99.9/100:", q)
for thr in (3000, 5000, 10000, 20000, 40000):
    print(f"  fraction of non-FF lanes with steps > {thr}: {np.nanmean(ps > thr):.4f}")
w = np.nan_to_num(ps, nan=0).reshape(-1, 64)
print("  per-wave max steps: mean", w.max(1).mean(), "median", np.median(w.max(1)), " per-wave mean steps", w.mean(1).mean())
print("  end reasons:", {k: int(np.sum((st & 0xFF) == v)) for k, v in (("max_time", 0), ("ground", 1), ("alt", 2), ("coast", 3), ("apogee", 4))})

dc = eng.debug_counters()
if sum(dc[8:16]) > 0:
    names = ["refill/ballots", "rhs: wind lookup", "rhs: atm+attitude+mass+thrust", "rhs: chute+aero", "rhs: forces+derivs",
             "stage combine", "final combine+normalise", "events+finish"]
    tot = sum(dc[8:16])
    print("  stamp shares (cycles per wave-step):")
    for nme, v in zip(names, dc[8:16]):
        print(f"    {nme:34s} {v / wi:9.1f}  {100 * v / tot:5.1f} %")
    print(f"    {'total':34s} {tot / wi:9.1f}")
