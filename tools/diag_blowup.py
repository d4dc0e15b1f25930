#!/usr/bin/env python3
"""How long does a trajectory spend at unphysical speeds before it blows up?  (Sizing of the f64_fast -> gate
hand-over: DESIGN.md section 5.)  Flies the first `--n` samples of the bench shard with the fp64 gate kernel and a full
per-step capture, and reports, for speed thresholds 1e4 .. 1e20 m/s: the fraction of samples that ever cross it, the
RK4 steps they make from the first crossing to the end of their physics (non-finite steps excluded), and the share of
all physics steps that is.

    python tools/diag_blowup.py [--n 4096] [--out gpurun_out/blowup.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from erpl_monte_carlo_sim_amd import _abi, flatten, models, sampling  # noqa: E402
from erpl_monte_carlo_sim_amd.engine import DeviceBatch, TrajectoryEngine  # noqa: E402

IC = {"position": [0.0, 0.0, 10.0], "velocity": [0, 0, 0.0],
      "attitude": [0.0, -np.pi / 2 + 0.02, 0.0], "angular_velocity": [0.0, 0.0, 0.0]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--cap", type=int, default=12000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "blowup.json"))
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    rocket, motor, atm, wm = models.Rocket(), models.LiquidMotor(), models.StandardAtmosphere(), models.WindModel()
    eng = TrajectoryEngine(dev)
    eng.set_config(flatten.config_from_objects(rocket, motor, atm))
    db = sampling.synthetic_dispersions(131072, rocket, motor, wm, IC, dev, precision=_abi.PREC_F64, seed=1234, engine=eng)
    thresholds = [1e4, 1e5, 1e6, 1e7, 1e8, 1e10, 1e12, 1e15, 1e20]
    cross = {t: 0 for t in thresholds}
    steps_after = {t: [] for t in thresholds}
    zmin_at = {t: [] for t in thresholds}
    total_phys = 0
    block = 512
    for b0 in range(0, args.n, block):
        ix = torch.arange(b0, min(args.n, b0 + block), device=dev)
        sb = DeviceBatch(db.ic.index_select(1, ix).contiguous(), db.rocket.index_select(1, ix).contiguous(),
                         db.motor.index_select(1, ix).contiguous(), db.alt_grid, db.wind.index_select(2, ix).contiguous(), _abi.PREC_F64)
        s, t, traj, tlen = eng.run(sb, traj_ids=list(range(sb.n)), traj_stride=1, traj_cap=args.cap)
        torch.cuda.synchronize()
        traj, tlen = traj.cpu().numpy(), tlen.cpu().numpy()
        for j in range(sb.n):
            m = int(tlen[j])
            v = traj[j, :m, 4:7]
            z = traj[j, :m, 3]
            with np.errstate(over="ignore", invalid="ignore"):
                sp = np.sqrt(np.sum(v * v, axis=1))
            fin = np.isfinite(traj[j, :m, 1:]).all(axis=1) & np.isfinite(sp)
            n_phys = int(np.argmin(fin)) if not fin.all() else m     # records up to the first non-finite state
            total_phys += n_phys
            for thr in thresholds:
                w = np.nonzero(sp[:n_phys] > thr)[0]
                if len(w):
                    cross[thr] += 1
                    steps_after[thr].append(n_phys - int(w[0]))
                    zmin_at[thr].append(float(z[int(w[0])]))
    out = {"samples": args.n, "physics_steps_total": total_phys, "thresholds": {}}
    for thr in thresholds:
        a = np.array(steps_after[thr]) if steps_after[thr] else np.zeros(1)
        zz = np.array(zmin_at[thr]) if zmin_at[thr] else np.zeros(1)
        out["thresholds"][f"{thr:g}"] = {
            "fraction_of_samples_crossing": cross[thr] / args.n,
            "steps_after_crossing": {"mean": float(a.mean()), "median": float(np.median(a)), "p90": float(np.percentile(a, 90)),
                                     "p99": float(np.percentile(a, 99)), "max": int(a.max())},
            "share_of_all_physics_steps": float(a.sum() / max(total_phys, 1)),
            "altitude_at_crossing": {"min": float(zz.min()), "median": float(np.median(zz)), "max": float(zz.max())}}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
