#!/usr/bin/env python3
"""Soak of the lane hand-over protocol: many full-size batches in flight with lane adoption on, EVERY batch's
summaries and statuses compared bit for bit with a plain erpl_mc_run_batch of the same inputs.

    python tools/soak_adopt.py [--n 131072] [--rounds 40] [--depth 8] [--precision f32]
"""
import argparse
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from erpl_monte_carlo_sim_amd import _abi, flatten, models, sampling  # noqa: E402
from erpl_monte_carlo_sim_amd.engine import TrajectoryEngine  # noqa: E402
from tools.diag_precision import IC  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=131072)
    ap.add_argument("--rounds", type=int, default=40)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--adopt", type=int, default=-1)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    rocket, motor, atm, wm = models.Rocket(), models.LiquidMotor(), models.StandardAtmosphere(), models.WindModel()
    eng = TrajectoryEngine(dev)
    eng.set_config(flatten.config_from_objects(rocket, motor, atm))
    prec = _abi.PRECISIONS[a.precision]
    dbs = [sampling.synthetic_dispersions(a.n - 4099 * i, rocket, motor, wm, IC, dev, precision=prec, seed=1234 + i, engine=eng)
           for i in range(3)]
    eng.set_adopt(0)
    refs = [tuple(x.clone() for x in eng.run(db)) for db in dbs]
    torch.cuda.synchronize()
    eng.set_adopt(a.adopt)
    eng.set_overlap(a.depth)
    outs = [[eng.alloc_outputs(db.n) for _ in range(a.depth)] for db in dbs]
    bad = total = 0
    for r in range(a.rounds):
        k = r % 3
        for j in range(a.depth):
            s, t = outs[k][j]
            s.fill_(-1.0); t.fill_(-1)
            eng.submit(dbs[k], summary=s, status=t)
        eng.wait()
        eng.synchronize()     # raises on a hand-over time-out
        for j in range(a.depth):
            s, t = outs[k][j]
            ok = torch.equal(t, refs[k][1]) and bool(((s == refs[k][0]) | (s.isnan() & refs[k][0].isnan())).all())
            bad += 0 if ok else 1
            total += 1
    steps, wi = eng.last_stats()
    print(f"soak {a.precision}: {total} batches of ~{a.n} samples, depth {a.depth}, lane utilisation {steps / 64 / wi:.3f}: "
          f"{bad} differ from erpl_mc_run_batch")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
