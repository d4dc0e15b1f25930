"""End-to-end profile of the drop-in MonteCarloAnalyzer.run_monte_carlo (host preparation + HIP pass + analysis)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
from erpl_monte_carlo_sim_amd import Rocket, LiquidMotor, StandardAtmosphere, WindModel, MonteCarloAnalyzer
mc = MonteCarloAnalyzer(Rocket(), LiquidMotor(), StandardAtmosphere(), WindModel(), verbose=False)
mc.n_trajectories = 0
mc.run_monte_carlo(H.EXAMPLE_IC, n_samples=64)
for n in (1000, 20000):
    t = time.time(); out = mc.run_monte_carlo(H.EXAMPLE_IC, n_samples=n); el = time.time() - t
    print(f"run_monte_carlo n={n}: {el:.2f} s ({el/n*1e6:.0f} us/sample)")
pr = cProfile.Profile(); pr.enable(); mc.run_monte_carlo(H.EXAMPLE_IC, n_samples=20000); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
