#!/usr/bin/env python3
"""The samples of bench shards 4-7 on which the fp64 throughput build misses the gate kernel's apogee_altitude at the 0.1 % bar."""
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
names = {v: k for k, v in vars(_abi).items() if k.startswith("SUM_")}
for j in range(4, 8):
    db = sampling.synthetic_dispersions(131072, r, m, w, B.EXAMPLE_IC, dev, precision=_abi.PREC_F64, seed=1234 + 7919 * j, engine=eng)
    gs, gt = (x.cpu().numpy() for x in eng.run(db))
    fs, ft = (x.cpu().numpy() for x in eng.run(B.as_precision(db, _abi.PREC_F64_FAST)))
    e = B.relerr(fs[_abi.SUM_APOGEE_ALT], gs[_abi.SUM_APOGEE_ALT])
    bad = np.nonzero(~(e <= 1e-3))[0]
    print("shard", j, "misses", bad.tolist(), flush=True)
    for i in bad:
        for row in range(gs.shape[0]):
            print("   %-28s gate %-24.16g fast %-24.16g" % (names.get(row, row), gs[row, i], fs[row, i]))
        print("   status gate %x fast %x" % (gt[i], ft[i]))
eng.close()
