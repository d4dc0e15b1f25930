"""cProfile of sampling.synthetic_dispersions (host-side cost per sub-batch of run_monte_carlo_device)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import erpl_monte_carlo_sim_amd as E
from erpl_monte_carlo_sim_amd import _abi, sampling
from erpl_monte_carlo_sim_amd.simulator import shared_engine
import helpers as H
dev = torch.device("cuda", 0)
eng = shared_engine(dev)
args = (131072, E.Rocket(), E.LiquidMotor(), E.WindModel(), dict(H.EXAMPLE_IC), dev)
for _ in range(3):
    sampling.synthetic_dispersions(*args, precision=_abi.PREC_F64_FAST, engine=eng)
torch.cuda.synchronize()
t = time.time()
for j in range(8):
    sampling.synthetic_dispersions(*args, precision=_abi.PREC_F64_FAST, seed=j, engine=eng)
print("host time per call, no sync: %.2f ms" % ((time.time() - t) / 8 * 1e3))
torch.cuda.synchronize()
print("with final sync: %.2f ms" % ((time.time() - t) / 8 * 1e3))
pr = cProfile.Profile(); pr.enable()
for j in range(8):
    sampling.synthetic_dispersions(*args, precision=_abi.PREC_F64_FAST, seed=j, engine=eng)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
