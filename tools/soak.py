#!/usr/bin/env python3
"""Soak: random batch sizes / workloads / precisions / launch geometries for a fixed wall time; every
(size, workload, precision) must give bitwise identical summaries under every geometry and compaction
setting, run after run.  python tools/soak.py [seconds]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
from erpl_monte_carlo_sim_amd import _abi, flatten, models, sampling
from erpl_monte_carlo_sim_amd.engine import TrajectoryEngine

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
dev = torch.device("cuda", 0)
rs = np.random.RandomState(2024)
rocket, atm, wm = models.Rocket(), models.StandardAtmosphere(), models.WindModel()
eng = TrajectoryEngine(dev)
base = {}
t0 = time.time(); runs = 0
while time.time() - t0 < budget:
    n = int(rs.choice([1, 2, 63, 64, 65, 255, 1000, 4097, 65537, 131072, 300000]))
    kind = str(rs.choice(["liquid", "solid"]))
    wl = str(rs.choice(["set_s", "set_p_apogee", "csv_chute"]))
    prec = _abi.PREC_F32 if (rs.rand() < 0.8 or n > 5000) else _abi.PREC_F64
    motor = models.SolidMotor() if kind == "solid" else models.LiquidMotor()
    eng.set_config(flatten.config_from_objects(rocket, motor, atm))
    csv = wl == "csv_chute"
    db = sampling.synthetic_dispersions(n, rocket, motor, wm, B.EXAMPLE_IC, dev, precision=prec, seed=7,
                                        planar=(wl != "set_s"), base_altitude_profile=B.CSV_ALT if csv else None,
                                        base_wind_profile=B.CSV_WIND if csv else None)
    flags = _abi.FLAG_STOP_AT_APOGEE if wl == "set_p_apogee" else 0
    block = int(rs.choice([64, 128, 256])); mb = int(rs.choice([0, 0, 1, 7, 300])); refill = int(rs.choice([1, 8, 40]))
    chunk = int(rs.choice([0, 0, 97, 1000, 5000]))
    eng.set_launch(block, mb, refill); eng.set_chunk(chunk); eng.set_waves_per_simd(int(rs.choice([0, 2, 3])))
    s, t = eng.run(db, flags=flags)
    torch.cuda.synchronize()
    key = (n, kind, wl, prec)
    if key not in base:
        base[key] = (s.clone(), t.clone())
    else:
        bs, bt = base[key]
        assert torch.equal(t, bt), (key, block, mb, refill, chunk)
        assert bool(((s == bs) | (s.isnan() & bs.isnan())).all()), (key, block, mb, refill, chunk)
    runs += 1
    if runs % 20 == 0:
        print(f"{runs} runs, {len(base)} distinct cases, {time.time() - t0:.0f} s", flush=True)
print(f"soak ok: {runs} runs over {len(base)} cases in {time.time() - t0:.0f} s")
