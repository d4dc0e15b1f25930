#!/usr/bin/env python3
"""Per-chunk cost of host-side sample construction on this box (single Python thread): parameter draws, dispersed_batch
(MT19937 motor / wind streams in C threads + NumPy), validation + upload."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import erpl_monte_carlo_sim_amd as E
from erpl_monte_carlo_sim_amd import flatten, sampling, _abi
from erpl_monte_carlo_sim_amd.engine import DeviceBatch
import helpers as H
n = 131072
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
r, m, w = E.Rocket(), E.LiquidMotor(), E.WindModel()
for rep in range(2):
    t = time.time(); p = flatten.generate_parameter_arrays(sampling.DEFAULT_UNCERTAINTY, n); t1 = time.time() - t
    t = time.time(); hb = flatten.dispersed_batch(r, m, w, H.EXAMPLE_IC, p, None, None); t2 = time.time() - t
    t = time.time(); db = DeviceBatch.from_host(hb, dev, _abi.PREC_F64_FAST); torch.cuda.synchronize(); t3 = time.time() - t
    print(f"n = {n}: parameter draws {t1*1e3:.0f} ms, dispersed_batch {t2*1e3:.0f} ms, validate + upload {t3*1e3:.0f} ms  (cores {flatten.host_cores()})")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); hb = flatten.dispersed_batch(r, m, w, H.EXAMPLE_IC, p, None, None); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(8)
