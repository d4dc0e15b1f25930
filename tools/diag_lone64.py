#!/usr/bin/env python3
"""Cycles per RK4 step of the fp64 throughput kernel with ONE and with TWO resident waves per SIMD (Set P planar flights
to apogee: ~15 k steps each, all lanes busy): is a wave bound by its own dependent-issue latency or by the port?
    python tools/diag_lone64.py [lib.so ...]      (ERPL variants built with tools/ab/build_variant.sh; "" = shipped)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
from erpl_monte_carlo_sim_amd import _abi, flatten, models, sampling
from erpl_monte_carlo_sim_amd.engine import TrajectoryEngine, DeviceBatch
dev = torch.device("cuda", 0)
r, m, a, w = models.Rocket(), models.LiquidMotor(), models.StandardAtmosphere(), models.WindModel()
libs = sys.argv[1:] or [""]
CLK = 2.4e9
for lib in libs:
    eng = TrajectoryEngine(dev, lib_path=(os.path.join(ROOT, "tools", "ab", f"liberpl_mc_{lib}.so") if lib else None))
    eng.set_config(flatten.config_from_objects(r, m, a)); eng.set_profiling(True); eng.set_adopt(0); eng.set_chunk(0)
    for prec_name in ("f64_fast",):
        prec = _abi.PRECISIONS[prec_name]
        for waves_per_simd in (1, 2):
            nblk = 1024 * waves_per_simd
            db = sampling.synthetic_dispersions(64 * nblk, r, m, w, B.EXAMPLE_IC, dev, precision=prec, seed=1234, planar=True, engine=eng)
            eng.set_launch(64, nblk, 1)
            for _ in range(2):
                eng.run(db, flags=_abi.FLAG_STOP_AT_APOGEE)
            torch.cuda.synchronize()
            ph, wi = eng.last_stats()
            ms = eng.last_kernel_ms()[1]
            iters_per_wave = wi / nblk
            cyc = ms * 1e-3 * CLK / iters_per_wave          # wall cycles per loop iteration of a wave
            print(f"{lib or 'shipped':14s} {prec_name} {waves_per_simd} wave(s)/SIMD: {ms:8.2f} ms, {iters_per_wave:7.0f} iterations/wave, "
                  f"{cyc:7.0f} cycles per wave-step ({cyc / waves_per_simd:6.0f} per step and SIMD), lane utilisation {ph / (64 * wi):.3f}")
    eng.close()
