#!/usr/bin/env python3
"""In-kernel workgroup compaction (erpl_mc_set_block_compaction): bitwise check against the plain launch and timing.
    python tools/try_block_compaction.py --n 4096 --precision f32 --blocks 256 --syncs 64 [--overlap 3]"""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from erpl_monte_carlo_sim_amd import _abi, flatten, models, sampling
from erpl_monte_carlo_sim_amd.engine import TrajectoryEngine
import bench as B

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--precision", default="f32")
ap.add_argument("--blocks", default="256")
ap.add_argument("--syncs", default="0,64")
ap.add_argument("--overlap", type=int, default=0)
ap.add_argument("--waves", type=int, default=0)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--workload", default="set_s")
a = ap.parse_args()
dev = torch.device("cuda", 0)
rocket, motor, atm, wm = models.Rocket(), models.LiquidMotor(), models.StandardAtmosphere(), models.WindModel()
eng = TrajectoryEngine(dev); eng.set_config(flatten.config_from_objects(rocket, motor, atm))
prec = _abi.PRECISIONS[a.precision]
csv = a.workload == "csv_chute"
db = sampling.synthetic_dispersions(a.n, rocket, motor, wm, B.EXAMPLE_IC, dev, precision=prec, seed=1234, planar=csv or a.workload.startswith("set_p"),
                                    base_altitude_profile=B.CSV_ALT if csv else None, base_wind_profile=B.CSV_WIND if csv else None, engine=eng)
flags = _abi.FLAG_STOP_AT_APOGEE if a.workload == "set_p_apogee" else 0
eng.set_waves_per_simd(a.waves)
eng.set_launch(64, 0, 1); eng.set_block_compaction(0)
ref_s, ref_t = (x.clone() for x in eng.run(db, flags=flags)); torch.cuda.synchronize()
print("reference done", flush=True)
ints = lambda s: [int(x) for x in s.split(",")]
for block in ints(a.blocks):
    for sync in ints(a.syncs):
        eng.set_launch(block, 0, 1); eng.set_block_compaction(sync)
        s, t = eng.run(db, flags=flags); torch.cuda.synchronize()
        same = bool(torch.equal(t, ref_t) and ((s == ref_s) | (s.isnan() & ref_s.isnan())).all())
        steps, wi = eng.last_stats()
        outs = [eng.alloc_outputs(a.n) for _ in range(max(a.overlap, 1))]
        if a.overlap: eng.set_overlap(a.overlap)
        go = eng.submit if a.overlap else eng.run
        for k in range(len(outs)): go(db, flags=flags, summary=outs[k][0], status=outs[k][1])
        if a.overlap: eng.wait()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(a.reps): go(db, flags=flags, summary=outs[i % len(outs)][0], status=outs[i % len(outs)][1])
        if a.overlap: eng.wait()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / a.reps * 1e3
        print(json.dumps({"precision": a.precision, "workload": a.workload, "n": a.n, "block": block, "sync": sync, "overlap": a.overlap,
                          "bitwise_equal": same, "ms": round(ms, 3), "traj_per_s": round(a.n / ms * 1e3), "util": round(steps / 64 / wi, 3)}), flush=True)
        if not same:
            sys.exit("MISMATCH")
