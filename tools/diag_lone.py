#!/usr/bin/env python3
"""Diagnostic: Set P to-apogee with a configurable number of resident blocks (lone-wave experiments)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from erpl_monte_carlo_sim_amd import _abi, flatten, models, sampling
from erpl_monte_carlo_sim_amd.engine import TrajectoryEngine
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda", 0)
r, m, a, w = models.Rocket(), models.LiquidMotor(), models.StandardAtmosphere(), models.WindModel()
eng = TrajectoryEngine(dev); eng.set_config(flatten.config_from_objects(r, m, a)); eng.set_profiling(True)
db = sampling.synthetic_dispersions(n, r, m, w, B.EXAMPLE_IC, dev, precision=_abi.PREC_F32, seed=1234, planar=True)
eng.set_launch(256, mb, 8)
for _ in range(2):
    eng.run(db, flags=_abi.FLAG_STOP_AT_APOGEE)
torch.cuda.synchronize()
ph, wi = eng.last_stats()
print(f"max_blocks={mb} n={n} flight_ms={eng.last_kernel_ms()[1]:.2f} wave_iters={wi:.0f} steps={ph:.0f}")
