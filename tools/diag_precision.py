#!/usr/bin/env python3
"""Precision table: the reference's apogee_altitude (global argmax,
simulator.py:488-490), first-descent apogee and end reason of every kernel build against the fp64
reference-order gate kernel (which tracks the CPU oracle / the Python reference on 100 % of the
cfg-2 set), with the throughput of each build on the same batch.

    python tools/diag_precision.py [--n 131072] [--mixed] [--out gpurun_out/precision.json]

Builds compared: f64 (gate), f64_fast, f32 and - with --mixed - the experiment library
liberpl_mc_mixed.so (fp32 RHS, fp64 state and RK4 combination; `make -C .../csrc mixed`).
"""
import argparse
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
CSV_ALT = np.array([0.0, 5000.0, 10000.0, 15000.0, 20000.0, 25000.0])
CSV_WIND = np.array([[2.0, 0, 0], [5, 1, 0], [8, 2, 0], [10, 2, 0], [12, 3, 0], [15, 3, 0]])


def relerr(a, b):
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        e = np.abs(a - b) / np.abs(b)
    return np.where(same, 0.0, np.where(np.isnan(e), np.inf, e))


def compare(ref, got):
    """Match rates of `got` against `ref` = (summary, status), overall and per class of the reference."""
    rs, rt = ref
    gs, gt = got
    e_ap = relerr(gs[_abi.SUM_APOGEE_ALT], rs[_abi.SUM_APOGEE_ALT])
    e_fa = relerr(gs[_abi.SUM_FIRST_APOGEE_ALT], rs[_abi.SUM_FIRST_APOGEE_ALT])
    same_end = (gt & 0xFF) == (rt & 0xFF)
    out = {"n": int(rs.shape[1]),
           "apogee_argmax_match_0p1pct": float(np.mean(e_ap <= 1e-3)),
           "first_apogee_match_0p1pct": float(np.mean(e_fa <= 1e-3)),
           "same_end_reason": float(np.mean(same_end)),
           "same_step_count": float(np.mean(gs[_abi.SUM_STEPS] == rs[_abi.SUM_STEPS])),
           "median_apogee_err": float(np.median(e_ap)), "median_first_apogee_err": float(np.median(e_fa))}
    # classes of the REFERENCE outcome: how the flight ended, whether the altitude ever went NaN, and
    # whether the reference apogee IS the first-descent apogee (i.e. the argmax was reached before the
    # tumble / blow-up) - the only class an fp32 integration can be expected to reproduce
    reason = rt & 0xFF
    nan = (rt & _abi.ST_NAN) != 0
    calm = (~nan) & (rs[_abi.SUM_APOGEE_ALT] == rs[_abi.SUM_FIRST_APOGEE_ALT])
    cls = {"apogee_is_first_descent_apogee": calm, "apogee_set_after_first_descent": (~nan) & ~calm, "altitude_went_nan": nan}
    for k, name in ((0, "end_max_time"), (1, "end_ground"), (2, "end_altitude_100km"), (3, "end_coast")):
        cls[name] = reason == k
    out["by_reference_class"] = {}
    for name, m in cls.items():
        if m.sum() == 0:
            continue
        out["by_reference_class"][name] = {
            "fraction": float(np.mean(m)),
            "apogee_argmax_match": float(np.mean(e_ap[m] <= 1e-3)),
            "first_apogee_match": float(np.mean(e_fa[m] <= 1e-3)),
            "same_end_reason": float(np.mean(same_end[m]))}
    return out


def timed(eng, db, flags, reps, overlap=0):
    """ms per pass of `reps` back-to-back passes (overlap = depth of submit(); 0 = run())."""
    outs = [eng.alloc_outputs(db.n) for _ in range(max(overlap, 1))]
    if overlap:
        eng.set_overlap(overlap)
    for k in range(len(outs)):   # warm-up (also sizes every workspace)
        (eng.submit if overlap else eng.run)(db, flags=flags, summary=outs[k][0], status=outs[k][1])
    if overlap:
        eng.wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        k = i % len(outs)
        (eng.submit if overlap else eng.run)(db, flags=flags, summary=outs[k][0], status=outs[k][1])
    if overlap:
        eng.wait()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=131072)
    ap.add_argument("--mixed", action="store_true")
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "precision.json"))
    ap.add_argument("--skip-set-p", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    rocket, motor, atm, wm = models.Rocket(), models.LiquidMotor(), models.StandardAtmosphere(), models.WindModel()
    cfg = flatten.config_from_objects(rocket, motor, atm)
    engines = {"product": TrajectoryEngine(dev)}
    if args.mixed:
        engines["mixed"] = TrajectoryEngine(dev, lib_path=os.path.join(ROOT, "erpl_monte_carlo_sim_amd", "csrc", "liberpl_mc_mixed.so"))
    for e in engines.values():
        e.set_config(cfg)
    builds = [("f64", "product", _abi.PREC_F64), ("f64_fast", "product", _abi.PREC_F64_FAST), ("f32", "product", _abi.PREC_F32)]
    if args.mixed:
        builds.append(("mixed_f32rhs_f64state", "mixed", _abi.PREC_F32))
    report = {}

    def run_all(tag, make_db, flags=0, time_reps=args.reps, overlaps=(0, 2, 3, 4)):
        res, rows = {}, {}
        for name, ek, prec in builds:
            eng = engines[ek]
            db = make_db(prec)
            s, t = eng.run(db, flags=flags)
            torch.cuda.synchronize()
            res[name] = (s.cpu().numpy(), t.cpu().numpy())
            steps, _ = eng.last_stats()
            row = {"physics_steps": steps}
            for ov in overlaps:
                if name == "f64" and ov not in (0, 2):
                    continue
                ms = timed(eng, db, flags, time_reps if name != "f64" else max(2, time_reps // 2), ov)
                row["ms_per_pass" + (f"_overlap{ov}" if ov else "")] = ms
                row["traj_per_s" + (f"_overlap{ov}" if ov else "")] = db.n / ms * 1e3
            rows[name] = row
            print(tag, name, json.dumps(row), flush=True)
        for name, _, _ in builds[1:]:
            rows[name]["vs_f64_gate"] = compare(res["f64"], res[name])
            print(tag, name, json.dumps(rows[name]["vs_f64_gate"]), flush=True)
        report[tag] = rows
        json.dump(report, open(args.out, "w"), indent=1)

    # ---- Set R: 1000 reference-faithful samples (seed = i streams, CSV wind) = BASELINE configs[1]
    pl = flatten.generate_parameter_samples(sampling.DEFAULT_UNCERTAINTY, 1000)
    hbr = flatten.dispersed_batch(rocket, motor, wm, IC, pl, CSV_ALT, CSV_WIND)
    run_all("set_r_1k", lambda prec: DeviceBatch.from_host(hbr, dev, prec), overlaps=(0,))

    # ---- Set S: the bench shard (synthetic dispersions, K = 100 wind)
    def as_precision(db64, prec):   # the SAME samples for every build: fp64 draws, wind rounded for fp32
        if prec != _abi.PREC_F32:
            return DeviceBatch(db64.ic, db64.rocket, db64.motor, db64.alt_grid, db64.wind, prec)
        return DeviceBatch(db64.ic, db64.rocket, db64.motor, db64.alt_grid, db64.wind.float().contiguous(), prec)

    db_s = sampling.synthetic_dispersions(args.n, rocket, motor, wm, IC, dev, precision=_abi.PREC_F64, seed=1234)
    run_all(f"set_s_{args.n}", lambda prec: as_precision(db_s, prec))
    del db_s

    # ---- Set P to apogee (healthy planar flights, ~15 k steps each)
    if not args.skip_set_p:
        db_p = sampling.synthetic_dispersions(args.n, rocket, motor, wm, IC, dev, precision=_abi.PREC_F64, seed=1234, planar=True)
        run_all(f"set_p_apogee_{args.n}", lambda prec: as_precision(db_p, prec), flags=_abi.FLAG_STOP_AT_APOGEE,
                time_reps=2, overlaps=(0, 2))
    print("written", args.out)


if __name__ == "__main__":
    main()
