#!/usr/bin/env python3
"""Where in the bench shard are the samples with the longest PHYSICS integration (the drain of a finite run)?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
from erpl_monte_carlo_sim_amd import _abi, flatten, models, sampling
from erpl_monte_carlo_sim_amd.engine import TrajectoryEngine
dev = torch.device("cuda", 0)
r, m, a, w = models.Rocket(), models.LiquidMotor(), models.StandardAtmosphere(), models.WindModel()
eng = TrajectoryEngine(dev); eng.set_config(flatten.config_from_objects(r, m, a))
for seed in (1234, 1235, 1236, 1237, 4321):
    db = sampling.synthetic_dispersions(131072, r, m, w, B.EXAMPLE_IC, dev, precision=_abi.PREC_F64, seed=seed, engine=eng)
    s, t = eng.run(db)
    s, t = s.cpu().numpy(), t.cpu().numpy()
    steps = s[_abi.SUM_STEPS]; end = t & 0xFF
    phys = np.where(end == 0, 0, steps)        # max_time enders are dragged by table once non-finite (their physics part is short)
    order = np.argsort(-phys)[:6]
    print("seed", seed, "longest physics integrations (sample index / 131072, steps, end):",
          [(round(int(i) / 131072, 3), int(phys[i]), int(end[i])) for i in order], flush=True)
eng.close()
