#!/usr/bin/env python3
"""Launch-shape sweep for overlapped submission (erpl_mc_submit_batch): block size x step-chunk x
overlap depth x kernel build, on the bench shard.  Prints one JSON line per point (ms per pass,
trajectories/s) and writes them all to --out."""
import argparse
import itertools
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from erpl_monte_carlo_sim_amd import _abi, flatten, models, sampling  # noqa: E402
from erpl_monte_carlo_sim_amd.engine import DeviceBatch, TrajectoryEngine  # noqa: E402

IC = {"position": [0.0, 0.0, 10.0], "velocity": [0, 0, 0.0],
      "attitude": [0.0, -np.pi / 2 + 0.02, 0.0], "angular_velocity": [0.0, 0.0, 0.0]}


def timed(eng, db, flags, reps, overlap):
    outs = [eng.alloc_outputs(db.n) for _ in range(max(overlap, 1))]
    if overlap:
        eng.set_overlap(overlap)
    go = eng.submit if overlap else eng.run
    for k in range(len(outs)):
        go(db, flags=flags, summary=outs[k][0], status=outs[k][1])
    if overlap:
        eng.wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        k = i % len(outs)
        go(db, flags=flags, summary=outs[k][0], status=outs[k][1])
    if overlap:
        eng.wait()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=131072)
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--blocks", default="256,64")
    ap.add_argument("--chunks", default="0,512,1024,2048")
    ap.add_argument("--overlaps", default="0,2,3,4")
    ap.add_argument("--waves", default="2,3")
    ap.add_argument("--max-blocks", default="0", help="workgroup caps to sweep (0 = library default)")
    ap.add_argument("--adopts", default="0", help="lane-adoption limits to sweep (0 = off)")
    ap.add_argument("--check", action="store_true", help="compare every point's results with the first point's, bit for bit")
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--planar", action="store_true")
    ap.add_argument("--lib", default=None, help="experiment build of the library to load instead of the product")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep_overlap.json"))
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    rocket, motor, atm, wm = models.Rocket(), models.LiquidMotor(), models.StandardAtmosphere(), models.WindModel()
    eng = TrajectoryEngine(dev, lib_path=a.lib)
    eng.set_config(flatten.config_from_objects(rocket, motor, atm))
    prec = _abi.PRECISIONS[a.precision]
    db = sampling.synthetic_dispersions(a.n, rocket, motor, wm, IC, dev, precision=prec, seed=1234, planar=a.planar)
    flags = _abi.FLAG_STOP_AT_APOGEE if a.planar else 0
    rows = []
    ints = lambda s: [int(x) for x in s.split(",")]
    ref = None
    for block, chunk, waves, ov, mb, adopt in itertools.product(ints(a.blocks), ints(a.chunks), ints(a.waves), ints(a.overlaps),
                                                              ints(a.max_blocks), ints(a.adopts)):
        eng.set_launch(block, mb, 1)
        eng.set_adopt(adopt)
        eng.set_chunk(chunk)
        eng.set_waves_per_simd(waves)
        ms = timed(eng, db, flags, a.reps, ov)
        same = None
        if a.check:
            s_, t_ = eng.run(db, flags=flags)
            torch.cuda.synchronize()
            if ref is None:
                ref = (s_.clone(), t_.clone())
            same = bool(torch.equal(t_, ref[1]) and ((s_ == ref[0]) | (s_.isnan() & ref[0].isnan())).all())
        steps, wi = eng.last_stats()
        row = {"publish_timeouts": eng.debug_counters()[3], "bitwise_equal_to_first": same, "util": round(steps / 64 / wi, 3) if wi else None,"lib": os.path.basename(a.lib) if a.lib else "product", "precision": a.precision, "n": a.n, "block": block, "chunk": chunk, "waves": waves, "overlap": ov, "max_blocks": mb, "adopt": adopt,
               "ms_per_pass": round(ms, 3), "traj_per_s": round(a.n / ms * 1e3)}
        rows.append(row)
        print(json.dumps(row), flush=True)
        json.dump(rows, open(a.out, "w"), indent=0)


if __name__ == "__main__":
    main()
