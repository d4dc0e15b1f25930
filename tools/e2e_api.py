#!/usr/bin/env python3
"""End-to-end rate of the named API at BASELINE sizes: MonteCarloAnalyzer.run_monte_carlo (host
MT19937 streams, chunked pipeline, lazy results) and run_monte_carlo_device.  usage: e2e_api.py [n] [precision]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import erpl_monte_carlo_sim_amd as E
import torch
import helpers as H
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
precision = sys.argv[2] if len(sys.argv) > 2 else "f64_fast"
mc = E.MonteCarloAnalyzer(E.Rocket(), E.LiquidMotor(), E.StandardAtmosphere(), E.WindModel(), verbose=False)
mc.precision = precision
mc.run_monte_carlo(dict(H.EXAMPLE_IC), n_samples=2000)     # warm-up: library load, first launches
out = {}
for label, kw in (("run_monte_carlo", {}), ("run_monte_carlo_optimized", {"optimized": True})):
    t = time.time(); r = mc.run_monte_carlo(dict(H.EXAMPLE_IC), n_samples=n, **kw); el = time.time() - t
    out[label] = {"n": n, "seconds": el, "samples_per_s": n / el, "n_valid": r["n_samples"], "n_outliers": r["n_outliers"]}
    t = time.time(); first = r["results"][0]; el2 = time.time() - t
    out[label]["first_record_ms"] = el2 * 1e3
mc.run_monte_carlo_device(dict(H.EXAMPLE_IC), n, precision=precision)
r = mc.run_monte_carlo_device(dict(H.EXAMPLE_IC), n, precision=precision)
out["run_monte_carlo_device"] = {k: r["performance"][k] for k in r["performance"]}
print(json.dumps(out, indent=1))
