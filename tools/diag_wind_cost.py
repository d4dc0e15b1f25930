#!/usr/bin/env python3
"""What the per-lane wind table costs: the same dispersed samples with synthetic wind profiles of K knots
(monte_carlo.py:282-288 uses 100), trajectory steps per second of the overlapped fp32 / f64_fast kernels.

    python tools/diag_wind_cost.py [--n 131072] [--knots 0,6,25,100,400] [--planar]
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from erpl_monte_carlo_sim_amd import _abi, flatten, models, sampling  # noqa: E402
from erpl_monte_carlo_sim_amd.engine import DeviceBatch, TrajectoryEngine  # noqa: E402
from tools.diag_precision import IC  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=131072)
    ap.add_argument("--knots", default="0,6,25,100,400")
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--planar", action="store_true")
    ap.add_argument("--reps", type=int, default=16)
    ap.add_argument("--overlap", type=int, default=6)
    ap.add_argument("--lib", default=None, help="experiment build of the library to load instead of the product")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    rocket, motor, atm, wm = models.Rocket(), models.LiquidMotor(), models.StandardAtmosphere(), models.WindModel()
    eng = TrajectoryEngine(dev, lib_path=a.lib)
    eng.set_config(flatten.config_from_objects(rocket, motor, atm))
    eng.set_overlap(a.overlap)
    prec = _abi.PRECISIONS[a.precision]
    flags = _abi.FLAG_STOP_AT_APOGEE if a.planar else 0
    for k in [int(x) for x in a.knots.split(",")]:
        db = sampling.synthetic_dispersions(a.n, rocket, motor, wm, IC, dev, precision=prec, seed=1234, planar=a.planar,
                                            n_wind_knots=max(k, 2), engine=eng)
        if k == 0:   # no wind table at all: the kernel build without the lookup
            db = DeviceBatch(db.ic, db.rocket, db.motor, None, None, prec)
        outs = [eng.alloc_outputs(db.n) for _ in range(a.overlap)]
        for o in outs:
            eng.submit(db, flags=flags, summary=o[0], status=o[1])
        eng.wait()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.reps):
            o = outs[i % len(outs)]
            eng.submit(db, flags=flags, summary=o[0], status=o[1])
        eng.wait()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.reps * 1e3
        steps, wi = eng.last_stats()
        print(json.dumps({"knots": k, "ms_per_pass": round(ms, 3), "steps_per_traj": round(steps / db.n, 1),
                          "G_traj_steps_per_s": round(steps / ms / 1e6, 2), "G_wave_lane_steps_per_s": round(64 * wi / ms / 1e6, 2),
                          "util": round(steps / 64 / wi, 3)}), flush=True)


if __name__ == "__main__":
    main()
