#!/usr/bin/env python3
"""Cycles per RK4 step of ONE wave of the fp64 throughput kernel against how many other waves share its CU: 256 / 512 / 1024 /
2048 workgroups of one wave on 256 CUs (Set P planar flights to apogee, all lanes busy).  Is the distance between a
lone wave in the drain of a run (4.9 us per step) and one wave per SIMD everywhere (6.5 us) the clock, or the CU?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
from erpl_monte_carlo_sim_amd import _abi, flatten, models, sampling
from erpl_monte_carlo_sim_amd.engine import TrajectoryEngine
dev = torch.device("cuda", 0)
r, m, a, w = models.Rocket(), models.LiquidMotor(), models.StandardAtmosphere(), models.WindModel()
eng = TrajectoryEngine(dev)
eng.set_config(flatten.config_from_objects(r, m, a)); eng.set_profiling(True); eng.set_adopt(0); eng.set_chunk(0)
for prec_name in ("f64_fast", "f32"):
    prec = _abi.PRECISIONS[prec_name]
    for nblk in (1, 64, 256, 512, 1024, 2048):
        db = sampling.synthetic_dispersions(64 * nblk, r, m, w, B.EXAMPLE_IC, dev, precision=prec, seed=1234, planar=True, engine=eng)
        eng.set_launch(64, nblk, 1)
        if prec_name == "f32":
            eng.set_waves_per_simd(2)
        for _ in range(2):
            eng.run(db, flags=_abi.FLAG_STOP_AT_APOGEE)
        torch.cuda.synchronize()
        ph, wi = eng.last_stats()
        ms = eng.last_kernel_ms()[1]
        it = wi / nblk
        print(f"{prec_name} {nblk:5d} waves on the chip: {ms:8.2f} ms, {it:7.0f} iterations/wave, {ms * 1e3 / it:6.3f} us per wave-step "
              f"({ms * 1e-3 * 2.4e9 / it:7.0f} nominal cycles), lane utilisation {ph / (64 * wi):.3f}", flush=True)
eng.close()
