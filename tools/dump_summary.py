#!/usr/bin/env python3
"""Dump the [16, n] summary + status + a few inputs of the bench shard (f64_fast) to gpurun_out/summary_set_s.npz."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from erpl_monte_carlo_sim_amd import _abi, flatten, models, sampling
from erpl_monte_carlo_sim_amd.engine import DeviceBatch, TrajectoryEngine
IC = {"position": [0.0, 0.0, 10.0], "velocity": [0, 0, 0.0], "attitude": [0.0, -np.pi / 2 + 0.02, 0.0], "angular_velocity": [0.0, 0.0, 0.0]}
dev = torch.device("cuda", 0)
rocket, motor, atm, wm = models.Rocket(), models.LiquidMotor(), models.StandardAtmosphere(), models.WindModel()
eng = TrajectoryEngine(dev); eng.set_config(flatten.config_from_objects(rocket, motor, atm))
db = sampling.synthetic_dispersions(131072, rocket, motor, wm, IC, dev, precision=_abi.PREC_F64_FAST, seed=1234, engine=eng)
s, t = eng.run(db)
torch.cuda.synchronize()
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "summary_set_s.npz"), summary=s.cpu().numpy(), status=t.cpu().numpy(),
                    ic=db.ic.cpu().numpy(), wind0=db.wind[:3].cpu().numpy())
print("ok")
