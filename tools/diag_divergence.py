#!/usr/bin/env python3
"""Where does the fp64 throughput build (f64_fast) leave the fp64 reference-order gate kernel?

For the samples of a batch whose reference `apogee_altitude` (or end reason) the two builds disagree on, both
builds re-fly the sample with a full per-step capture, and the first step at which
  (a) any state component differs in CLASS (finite / +inf / -inf / NaN),
  (b) the relative difference of the state exceeds 1e-9 / 1e-6 / 1e-3
is reported with the magnitudes of the state there.  Then the RK4 stages of the step before (a) are replayed on
the host through erpl_mc_debug_eval of BOTH builds from the gate kernel's state, to name the stage and the
derivative component in which the class first differs.

    python tools/diag_divergence.py [--n 131072] [--max-samples 48] [--out gpurun_out/diverge]
"""
import argparse
import json
import os
import sys

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
NAMES = ["x", "y", "z", "vx", "vy", "vz", "q0", "q1", "q2", "q3", "wx", "wy", "wz", "pf"]


def relerr(a, b):
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        e = np.abs(a - b) / np.abs(b)
    return np.where(same, 0.0, np.where(np.isnan(e), np.inf, e))


def klass(a):
    """0 finite, 1 +inf, 2 -inf, 3 NaN"""
    return np.where(np.isnan(a), 3, np.where(np.isposinf(a), 1, np.where(np.isneginf(a), 2, 0)))


def sub_batch(db, idx, prec):
    ix = torch.as_tensor(np.asarray(idx, dtype=np.int64), device=db.ic.device)
    wind = db.wind.index_select(2, ix).contiguous() if db.wind is not None else None
    return DeviceBatch(db.ic.index_select(1, ix).contiguous(), db.rocket.index_select(1, ix).contiguous(),
                       db.motor.index_select(1, ix).contiguous(), db.alt_grid, wind, prec)


def fly(eng, db, prec, cap):
    d = DeviceBatch(db.ic, db.rocket, db.motor, db.alt_grid, db.wind, prec)
    s, t, traj, tlen = eng.run(d, traj_ids=list(range(db.n)), traj_stride=1, traj_cap=cap)
    torch.cuda.synchronize()
    return s.cpu().numpy(), t.cpu().numpy(), traj.cpu().numpy(), tlen.cpu().numpy()


def rhs(eng, db, prec, t, y, chute):
    d = DeviceBatch(db.ic, db.rocket, db.motor, db.alt_grid, db.wind, prec)
    inp = np.concatenate([t[None, :], y, chute[None, :].astype(np.float64)], axis=0)
    out = eng.debug_eval(d, _abi.DBG_RHS, inp)
    return out[:14], out[14] > 0


def fmt(v):
    return "[" + ", ".join(f"{x:.6g}" for x in v) + "]"


def analyse(eng, db, tag, lines, max_samples, cap, dt):
    gs, gt = eng.run(DeviceBatch(db.ic, db.rocket, db.motor, db.alt_grid, db.wind, _abi.PREC_F64))
    fs, ft = eng.run(DeviceBatch(db.ic, db.rocket, db.motor, db.alt_grid, db.wind, _abi.PREC_F64_FAST))
    torch.cuda.synchronize()
    gs, gt, fs, ft = gs.cpu().numpy(), gt.cpu().numpy(), fs.cpu().numpy(), ft.cpu().numpy()
    e_ap = relerr(fs[_abi.SUM_APOGEE_ALT], gs[_abi.SUM_APOGEE_ALT])
    same_end = (gt & 0xFF) == (ft & 0xFF)
    bad = np.nonzero((e_ap > 1e-3) | ~same_end)[0]
    nan = (gt & _abi.ST_NAN) != 0
    calm = (~nan) & (gs[_abi.SUM_APOGEE_ALT] == gs[_abi.SUM_FIRST_APOGEE_ALT])
    lines.append(f"== {tag}: n = {db.n}, apogee match {np.mean(e_ap <= 1e-3):.5f}, same end {np.mean(same_end):.5f}, "
                 f"{len(bad)} disagree (ref class: nan {int(nan[bad].sum())}, after-first-descent {int(((~nan) & ~calm)[bad].sum())}, "
                 f"before {int(calm[bad].sum())})")
    # nan class first (the bulk), then the rest
    order = np.concatenate([bad[nan[bad]], bad[~nan[bad]]])
    half = max_samples // 2
    pick = np.concatenate([bad[nan[bad]][:half], bad[~nan[bad]][:max_samples - half]])
    if len(pick) == 0:
        return {"n": int(db.n), "disagree": 0}
    sb = sub_batch(db, pick, _abi.PREC_F64)
    g_s, g_t, g_tr, g_len = fly(eng, sb, _abi.PREC_F64, cap)
    f_s, f_t, f_tr, f_len = fly(eng, sb, _abi.PREC_F64_FAST, cap)
    records = []
    for j, i in enumerate(pick):
        m = int(min(g_len[j], f_len[j]))
        G, F = g_tr[j, :m, 1:], f_tr[j, :m, 1:]
        kc = klass(G) != klass(F)
        step_class = int(np.argmax(kc.any(axis=1))) if kc.any() else -1
        with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
            scale = np.maximum(np.abs(G), 1e-300)
            rel = np.where((G == F) | (np.isnan(G) & np.isnan(F)), 0.0, np.abs(G - F) / scale)
            rel = np.where(np.isnan(rel), np.inf, rel)
        # quaternion / rates are O(1) quantities: measure them absolutely
        worst = rel.max(axis=1)
        first = {}
        for thr in (1e-9, 1e-6, 1e-3):
            w = np.nonzero(worst > thr)[0]
            first[thr] = int(w[0]) if len(w) else -1
        gn = np.nonzero(np.isnan(G).any(axis=1))[0]
        fn = np.nonzero(np.isnan(F).any(axis=1))[0]
        gi = np.nonzero(np.isinf(G).any(axis=1))[0]
        fi = np.nonzero(np.isinf(F).any(axis=1))[0]
        rec = {"sample": int(i), "ref_class": "nan" if nan[i] else ("before" if calm[i] else "after"),
               "gate": {"apogee": float(gs[0, i]), "end": int(gt[i] & 0xFF), "steps": int(gs[_abi.SUM_STEPS, i]), "status": int(gt[i])},
               "fast": {"apogee": float(fs[0, i]), "end": int(ft[i] & 0xFF), "steps": int(fs[_abi.SUM_STEPS, i]), "status": int(ft[i])},
               "captured_steps": m, "first_class_difference_step": step_class,
               "first_rel_1e-9": first[1e-9], "first_rel_1e-6": first[1e-6], "first_rel_1e-3": first[1e-3],
               "gate_first_nan": int(gn[0]) if len(gn) else -1, "fast_first_nan": int(fn[0]) if len(fn) else -1,
               "gate_first_inf": int(gi[0]) if len(gi) else -1, "fast_first_inf": int(fi[0]) if len(fi) else -1}
        for key, st in (("at_rel_1e-6", first[1e-6]), ("at_class", step_class)):
            if st >= 0:
                rec[key] = {"max_abs_gate": float(np.nanmax(np.abs(np.where(np.isinf(G[st]), np.nan, G[st])), initial=0.0))}
        records.append(rec)
        lines.append(f"-- sample {i} ({rec['ref_class']}): gate apogee {gs[0, i]:.9g} end {gt[i] & 0xFF} steps {int(gs[_abi.SUM_STEPS, i])} | "
                     f"fast apogee {fs[0, i]:.9g} end {ft[i] & 0xFF} steps {int(fs[_abi.SUM_STEPS, i])}")
        lines.append(f"   first rel>1e-9 @ {first[1e-9]}, >1e-6 @ {first[1e-6]}, >1e-3 @ {first[1e-3]}, class diff @ {step_class}; "
                     f"first nan gate/fast {rec['gate_first_nan']}/{rec['fast_first_nan']}, first inf {rec['gate_first_inf']}/{rec['fast_first_inf']}")
        for st in sorted({s for s in (first[1e-6], step_class - 1, step_class) if s is not None and s >= 0}):
            lines.append(f"   step {st}: gate {fmt(G[st])}")
            lines.append(f"   step {st}: fast {fmt(F[st])}")
        rec["_j"] = j
        rec["_G"], rec["_F"] = G, F
        rec["_t"] = g_tr[j, :m, 0]
    # ---- replay the RK4 stages of the step before the first class difference, both builds, from the GATE state ----
    sel = [r for r in records if r["first_class_difference_step"] > 0]
    if sel:
        J = np.array([r["_j"] for r in sel])
        sb2 = sub_batch(sb, J, _abi.PREC_F64)
        y0 = np.stack([r["_G"][r["first_class_difference_step"] - 1] for r in sel], axis=1)
        t0 = np.array([r["_t"][r["first_class_difference_step"] - 1] for r in sel])
        # parachute latch state is not in the capture: below 500 m descending -> assume latched if gate status says so
        ch = np.array([(r["gate"]["status"] & _abi.ST_CHUTE) != 0 and False for r in sel])
        ys = {p: y0.copy() for p in (_abi.PREC_F64, _abi.PREC_F64_FAST)}
        chs = {p: ch.copy() for p in ys}
        acc = {p: np.zeros_like(y0) for p in ys}
        found = [None] * len(sel)
        for stage in range(4):
            ts = t0 + (0.0 if stage == 0 else (dt if stage == 3 else 0.5 * dt))
            ks = {}
            for p in ys:
                # both builds evaluate the SAME stage state (the gate's) so that the difference is the RHS's alone
                ks[p], chs[p] = rhs(eng, sb2, p, ts, ys[_abi.PREC_F64], chs[p])
            kG, kF = ks[_abi.PREC_F64], ks[_abi.PREC_F64_FAST]
            dc = klass(kG) != klass(kF)
            for c in range(len(sel)):
                if found[c] is None and dc[:, c].any():
                    comp = [NAMES[k] for k in np.nonzero(dc[:, c])[0]]
                    found[c] = {"stage": stage + 1, "components": comp, "stage_state": ys[_abi.PREC_F64][:, c].tolist(),
                                "dy_gate": kG[:, c].tolist(), "dy_fast": kF[:, c].tolist()}
            wgt = 1.0 if stage in (0, 3) else 2.0
            adv = dt if stage == 2 else 0.5 * dt
            with np.errstate(all="ignore"):
                acc[_abi.PREC_F64] = acc[_abi.PREC_F64] + wgt * kG
                if stage < 3:
                    ys[_abi.PREC_F64] = y0 + adv * kG
        for r, f in zip(sel, found):
            r["rhs_replay"] = f
            lines.append(f"-- replay sample {r['sample']} step {r['first_class_difference_step'] - 1}: " +
                         ("no class difference of the RHS on the gate's stage states" if f is None else
                          f"stage {f['stage']} components {f['components']}"))
            if f is not None:
                lines.append(f"   stage state {fmt(f['stage_state'])}")
                lines.append(f"   dy gate     {fmt(f['dy_gate'])}")
                lines.append(f"   dy fast     {fmt(f['dy_fast'])}")
    for r in records:
        for k in ("_j", "_G", "_F", "_t"):
            r.pop(k, None)
    return {"n": int(db.n), "apogee_match": float(np.mean(e_ap <= 1e-3)), "same_end": float(np.mean(same_end)),
            "disagree": int(len(bad)), "disagree_nan_class": int(nan[bad].sum()), "records": records}


def host_of(db, idx):
    """Samples `idx` of a DeviceBatch as a flatten.HostBatch for the CPU oracle."""
    ix = torch.as_tensor(np.asarray(idx, dtype=np.int64), device=db.ic.device)
    hb = flatten.HostBatch(len(idx), db.k_wind)
    hb.ic = np.ascontiguousarray(db.ic.index_select(1, ix).cpu().numpy())
    hb.rocket = np.ascontiguousarray(db.rocket.index_select(1, ix).cpu().numpy())
    hb.motor = np.ascontiguousarray(db.motor.index_select(1, ix).cpu().numpy())
    if db.k_wind:
        hb.alt_grid = db.alt_grid.cpu().numpy().astype(np.float64)
        hb.wind = np.ascontiguousarray(db.wind.index_select(2, ix).double().cpu().numpy())
    return hb


def analyse_vs_oracle(eng, cfg, db, m, lines, max_samples, cap, dt, threads):
    """The fp64 gate kernel against the CPU oracle on the first m samples: where do two implementations of the SAME
    operation order part?  (libm differences: device pow / exp / atan2 / sincos against glibc's.)"""
    from oracle import oracle as orc
    hb = host_of(db, np.arange(m))
    osum, ostat = orc.run_batch(cfg, hb, threads=threads)
    gs, gt = eng.run(sub_batch(db, np.arange(m), _abi.PREC_F64))
    torch.cuda.synchronize()
    gs, gt = gs.cpu().numpy(), gt.cpu().numpy()
    e_ap = relerr(gs[_abi.SUM_APOGEE_ALT], osum[_abi.SUM_APOGEE_ALT])
    same_end = (gt & 0xFF) == (ostat & 0xFF)
    bad = np.nonzero((e_ap > 1e-3) | ~same_end)[0]
    lines.append(f"== gate kernel vs CPU oracle: n = {m}, apogee match {np.mean(e_ap <= 1e-3):.5f}, same end {np.mean(same_end):.5f}, {len(bad)} disagree")
    pick = bad[:max_samples]
    if len(pick) == 0:
        return {"n": m, "disagree": 0}
    hs = host_of(db, pick)
    o_s, o_t, o_tr, o_len = orc.run_batch(cfg, hs, threads=threads, traj_ids=list(range(len(pick))), traj_stride=1, traj_cap=cap)
    sb = sub_batch(db, pick, _abi.PREC_F64)
    g_s, g_t, g_tr, g_len = fly(eng, sb, _abi.PREC_F64, cap)
    recs = []
    for j, i in enumerate(pick):
        mm = int(min(g_len[j], o_len[j]))
        G, O = g_tr[j, :mm, 1:], o_tr[j, :mm, 1:]
        kc = klass(G) != klass(O)
        step_class = int(np.argmax(kc.any(axis=1))) if kc.any() else -1
        with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
            rel = np.where((G == O) | (np.isnan(G) & np.isnan(O)), 0.0, np.abs(G - O) / np.maximum(np.abs(O), 1e-300))
            rel = np.where(np.isnan(rel), np.inf, rel)
        worst = rel.max(axis=1)
        first = {thr: (int(np.nonzero(worst > thr)[0][0]) if (worst > thr).any() else -1) for thr in (1e-12, 1e-9, 1e-6, 1e-3)}
        lines.append(f"-- sample {i}: oracle apogee {osum[0, i]:.9g} end {ostat[i] & 0xFF} steps {int(osum[_abi.SUM_STEPS, i])} | "
                     f"gate apogee {gs[0, i]:.9g} end {gt[i] & 0xFF} steps {int(gs[_abi.SUM_STEPS, i])}")
        lines.append(f"   first rel>1e-12 @ {first[1e-12]}, >1e-9 @ {first[1e-9]}, >1e-6 @ {first[1e-6]}, >1e-3 @ {first[1e-3]}, class diff @ {step_class}")
        for st in sorted({s for s in (step_class - 2, step_class - 1, step_class) if s >= 0}):
            lines.append(f"   step {st}: oracle {fmt(O[st])}")
            lines.append(f"   step {st}: gate   {fmt(G[st])}")
        rec = {"sample": int(i), "first_class_difference_step": step_class, "first": {str(k): v for k, v in first.items()}}
        # replay the RK4 stages of the step before the class difference from the ORACLE's state with both RHS
        if step_class > 0:
            one = host_of(db, [i])
            y0, t0 = O[step_class - 1].copy(), float(o_tr[j, step_class - 1, 0])
            sb1 = sub_batch(db, [i], _abi.PREC_F64)
            ys, ch_o, ch_g = y0.copy(), 0, np.array([False])
            for stage in range(4):
                ts = t0 + (0.0 if stage == 0 else (dt if stage == 3 else 0.5 * dt))
                k_o, ch_o = orc.rhs(cfg, one, ts, ys, ch_o)
                k_g, ch_g = rhs(eng, sb1, _abi.PREC_F64, np.array([ts]), ys[:, None], ch_g)
                k_g = k_g[:, 0]
                with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
                    r = np.where((k_o == k_g) | (np.isnan(k_o) & np.isnan(k_g)), 0.0, np.abs(k_o - k_g) / np.maximum(np.abs(k_o), 1e-300))
                dc = klass(k_o) != klass(k_g)
                lines.append(f"   replay stage {stage + 1}: max rel diff of dy {np.nanmax(np.where(np.isinf(r), np.nan, r)):.3g}" +
                             (f", CLASS differs in {[NAMES[k] for k in np.nonzero(dc)[0]]}" if dc.any() else ""))
                if dc.any() or stage == 3:
                    lines.append(f"     stage state {fmt(ys)}")
                    lines.append(f"     dy oracle   {fmt(k_o)}")
                    lines.append(f"     dy gate     {fmt(k_g)}")
                if dc.any():
                    rec["rhs_replay"] = {"stage": stage + 1, "components": [NAMES[k] for k in np.nonzero(dc)[0]],
                                         "stage_state": ys.tolist(), "dy_oracle": k_o.tolist(), "dy_gate": k_g.tolist()}
                    break
                adv = dt if stage == 2 else 0.5 * dt
                with np.errstate(all="ignore"):
                    ys = y0 + adv * k_o
            # the same step from the GATE's own state with the ORACLE's RHS and a NumPy RK4 combination (the reference's
            # expression, simulator.py:217-224): does the difference sit in the state (sensitivity) or in the kernel?
            def full_step(y_start, use_gate):
                ys, ch_o, ch_g = y_start.copy(), 0, np.array([False])
                ks = []
                for stage in range(4):
                    ts = t0 + (0.0 if stage == 0 else (dt if stage == 3 else 0.5 * dt))
                    if use_gate:
                        k, ch_g = rhs(eng, sb1, _abi.PREC_F64, np.array([ts]), ys[:, None], ch_g)
                        k = k[:, 0]
                    else:
                        k, ch_o = orc.rhs(cfg, one, ts, ys, ch_o)
                    ks.append(k)
                    with np.errstate(all="ignore"):
                        ys = y_start + (dt if stage == 2 else 0.5 * dt) * k
                with np.errstate(all="ignore"):
                    yn = y_start + (dt / 6.0) * (((ks[0] + 2 * ks[1]) + 2 * ks[2]) + ks[3])
                return yn
            for label, y_start in (("oracle state", O[step_class - 1]), ("gate state  ", G[step_class - 1])):
                for use_gate in (False, True):
                    yn = full_step(y_start, use_gate)
                    lines.append(f"   host RK4 from {label} with {'gate RHS  ' if use_gate else 'oracle RHS'}: {fmt(yn[:6])}")
            # three steps with the parachute latch carried along (it is part of the state, simulator.py:366-369)
            for use_gate in (False, True):
                k0 = max(step_class - 3, 0)
                y, chs = (G if use_gate else O)[k0].copy(), (np.array([False]) if use_gate else 0)
                for kk in range(k0, step_class):
                    tk = float((g_tr if use_gate else o_tr)[j, kk, 0])
                    ys, ksl = y.copy(), []
                    for stage in range(4):
                        ts = tk + (0.0 if stage == 0 else (dt if stage == 3 else 0.5 * dt))
                        if use_gate:
                            k, chs = rhs(eng, sb1, _abi.PREC_F64, np.array([ts]), ys[:, None], chs)
                            k = k[:, 0]
                        else:
                            k, chs = orc.rhs(cfg, one, ts, ys, chs)
                        ksl.append(k)
                        with np.errstate(all="ignore"):
                            ys = y + (dt if stage == 2 else 0.5 * dt) * k
                    with np.errstate(all="ignore"):
                        y = y + (dt / 6.0) * (((ksl[0] + 2 * ksl[1]) + 2 * ksl[2]) + ksl[3])
                        nrm = np.sqrt(((y[6] * y[6] + y[7] * y[7]) + y[8] * y[8]) + y[9] * y[9])
                        if nrm > 1e-12:
                            y[6:10] = y[6:10] / nrm
                        else:
                            y[6:10] = [1.0, 0.0, 0.0, 0.0]
                    lines.append(f"   host steps with {'gate RHS  ' if use_gate else 'oracle RHS'} -> step {kk + 1}: {fmt(y[:6])} latch {int(np.asarray(chs).ravel()[0])}")
            lines.append(f"   recorded next state: oracle {fmt(O[step_class][:6])}")
            lines.append(f"   recorded next state: gate   {fmt(G[step_class][:6])}")
            d01 = G[step_class - 1] - O[step_class - 1]
            with np.errstate(all="ignore"):
                lines.append(f"   rel diff of the two states before the step: {fmt(np.abs(d01) / np.maximum(np.abs(O[step_class - 1]), 1e-300))}")
        recs.append(rec)
    return {"n": m, "apogee_match": float(np.mean(e_ap <= 1e-3)), "disagree": int(len(bad)), "records": recs}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=131072)
    ap.add_argument("--max-samples", type=int, default=48)
    ap.add_argument("--cap", type=int, default=12000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "diverge"))
    ap.add_argument("--skip-set-r", action="store_true")
    ap.add_argument("--vs-oracle", type=int, default=0, help="gate kernel vs CPU oracle on the first N samples of the shard instead")
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--motor", default="liquid", choices=["liquid", "solid"])
    ap.add_argument("--wind", default="syn", choices=["syn", "csv", "none"], help="Set S wind: synthetic K=100, CSV base K=6, or none")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    dev = torch.device("cuda", 0)
    rocket, atm, wm = models.Rocket(), models.StandardAtmosphere(), models.WindModel()
    motor = models.SolidMotor() if args.motor == "solid" else models.LiquidMotor()
    cfg = flatten.config_from_objects(rocket, motor, atm)
    eng = TrajectoryEngine(dev, lib_path=os.environ.get("ERPL_LIB"))
    eng.set_config(cfg)
    dt = min(float(cfg.dt_initial), 0.005)
    lines, report = [], {}
    kw = dict(base_altitude_profile=CSV_ALT, base_wind_profile=CSV_WIND) if args.wind == "csv" else {}
    db = sampling.synthetic_dispersions(args.n, rocket, motor, wm, IC, dev, precision=_abi.PREC_F64, seed=1234, engine=eng, **kw)
    if args.wind == "none":
        db = DeviceBatch(db.ic, db.rocket, db.motor, None, None, _abi.PREC_F64)
    if args.vs_oracle > 0:
        report["gate_vs_oracle"] = analyse_vs_oracle(eng, cfg, db, args.vs_oracle, lines, args.max_samples, args.cap, dt, args.threads)
    else:
        report["set_s"] = analyse(eng, db, f"Set S ({args.motor} motor, wind {args.wind})", lines, args.max_samples, args.cap, dt)
    if not args.skip_set_r and args.vs_oracle == 0:
        pl = flatten.generate_parameter_samples(sampling.DEFAULT_UNCERTAINTY, 4000)
        hbr = flatten.dispersed_batch(rocket, motor, wm, IC, pl, *((CSV_ALT, CSV_WIND) if args.wind != "none" else (None, None)))
        dbr = DeviceBatch.from_host(hbr, dev, _abi.PREC_F64)
        report["set_r"] = analyse(eng, dbr, "Set R (cfg 2 recipe, 4000 samples)", lines, args.max_samples, args.cap, dt)
    with open(os.path.join(args.out, "report.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(args.out, "report.json"), "w") as f:
        json.dump(report, f, indent=1)
    print("\n".join(lines[:400]))


if __name__ == "__main__":
    main()
