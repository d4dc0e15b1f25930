"""MI355X-native Monte Carlo 6-DOF trajectory engine (drop-in for the hot path of
smcconoughey/erpl_monte_carlo_sim: MonteCarloAnalyzer.run_monte_carlo / FlightSimulator.simulate_flight).
"""
import os as _os
import sys as _sys


def _hip_already_started():
    """True when the HIP runtime may already have read GPU_MAX_HW_QUEUES: torch.cuda is initialised, or a tool that
    starts HIP before the program does is attached (rocprofv3 and friends preload their library, which initialises
    the runtime - with --pmc certainly - before the first line of Python runs)."""
    env = _os.environ
    if any(k in env for k in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCPROF_OUTPUT_PATH")) or \
            any(w in env.get("LD_PRELOAD", "") for w in ("rocprof", "roctracer", "rocprofiler")):
        return True
    t = _sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return False


# Eight batches in flight need a hardware queue each (include/erpl_mc.h, erpl_mc_set_overlap); the HIP
# runtime reads the limit once, at its first call - which a bare `import torch` has not made yet.  If the
# runtime is already up (the caller used torch.cuda before importing this package, or a profiler preloaded it)
# the variable is left alone: the library sizes its overlap depth from it and must not believe in queues the
# process does not have (measured: eight lanes on four queues are 2x slower than three).
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    if _hip_already_started():
        import warnings as _warnings
        _warnings.warn("HIP was (or may have been: a profiler is attached) initialised before erpl_monte_carlo_sim_amd was "
                       "imported: GPU_MAX_HW_QUEUES stays at the runtime's default (4 hardware queues), so submitted batches "
                       "overlap three deep instead of eight.  Export GPU_MAX_HW_QUEUES=24 in the environment of the "
                       "process (or import this package before the first torch.cuda call).")
    else:
        _os.environ["GPU_MAX_HW_QUEUES"] = "24"

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
