"""MI355X-native Monte Carlo 6-DOF trajectory engine (drop-in for the hot path of
smcconoughey/erpl_monte_carlo_sim: MonteCarloAnalyzer.run_monte_carlo / FlightSimulator.simulate_flight).
"""
import os as _os

# Eight batches in flight need a hardware queue each (include/erpl_mc.h, erpl_mc_set_overlap); the HIP
# runtime reads the limit once, at its first call - which a bare `import torch` has not made yet.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

from .models import Rocket, SolidMotor, LiquidMotor, StandardAtmosphere, WindModel  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):
    # torch-dependent modules are imported lazily so that `import erpl_monte_carlo_sim_amd`
    # stays cheap for host-only users (flattening, sampling).
    if name == "FlightSimulator":
        from .simulator import FlightSimulator
        return FlightSimulator
    if name == "MonteCarloAnalyzer":
        from .monte_carlo import MonteCarloAnalyzer
        return MonteCarloAnalyzer
    if name == "TrajectoryEngine":
        from .engine import TrajectoryEngine
        return TrajectoryEngine
    raise AttributeError(name)


__all__ = ["Rocket", "SolidMotor", "LiquidMotor", "StandardAtmosphere", "WindModel",
           "FlightSimulator", "MonteCarloAnalyzer", "TrajectoryEngine"]
