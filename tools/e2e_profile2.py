#!/usr/bin/env python3
"""Where the time of run_monte_carlo(n = 10^6) goes on the host (main thread; cProfile)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import erpl_monte_carlo_sim_amd as E
import helpers as H
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
mc = E.MonteCarloAnalyzer(E.Rocket(), E.LiquidMotor(), E.StandardAtmosphere(), E.WindModel(), verbose=False)
mc.run_monte_carlo(dict(H.EXAMPLE_IC), n_samples=2000)
pr = cProfile.Profile(); t = time.time(); pr.enable()
r = mc.run_monte_carlo(dict(H.EXAMPLE_IC), n_samples=n)
pr.disable(); print("total", time.time() - t)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
