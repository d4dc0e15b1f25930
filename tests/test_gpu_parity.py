"""GPU parity tests (run on a real MI355X: `pytest -m gpu`).  Every test drives the HIP kernels
through the C ABI (erpl_mc_run_batch) and compares with
  (1) the golden vectors captured from the imported Python reference, and
  (2) the CPU oracle on the same seeded inputs.

Tolerances (stated per BASELINE north_star: per-sample apogee within 0.1 % of the CPU reference):
  fp64 kernel, healthy flights : 1e-9 relative on apogee / first-descent apogee / range,
                                 identical step counts and termination reasons
  fp64 kernel, diverging flights (error amplification 1e3..1e4): 1e-6 on
                                 first-descent apogee, match-RATE >= 99 % at 1e-3 on the reference's
                                 global-argmax apogee
  fp32 kernel                  : 1e-3 (the north-star 0.1 %) on first-descent apogee of healthy
                                 flights, match-rate reported for diverging ones
"""
import numpy as np
import pytest
import torch

from erpl_monte_carlo_sim_amd import _abi, flatten, models

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from erpl_monte_carlo_sim_amd.engine import TrajectoryEngine
    eng = TrajectoryEngine(torch.device("cuda", 0))
    yield eng
    eng.close()


def run_gpu(engine, cfg, hb, prec=_abi.PREC_F64, flags=0, **kw):
    from erpl_monte_carlo_sim_amd.engine import DeviceBatch
    engine.set_config(cfg)
    db = DeviceBatch.from_host(hb, engine.device, prec)
    out = engine.run(db, flags=flags, **kw)
    torch.cuda.synchronize()
    return tuple(o.cpu().numpy() for o in out)


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    with np.errstate(invalid="ignore", divide="ignore"):
        e = np.abs(a - b) / np.abs(b)
    return np.where(same, 0.0, e)


def mc_batch(kind, n, base="csv", stream="seed_i", planar=False, start=0):
    pl = flatten.generate_parameter_samples(H.UNCERTAINTY, start + n, stream=stream)[start:]
    kw = dict(base_altitude_profile=H.CSV_ALT, base_wind_profile=H.CSV_WIND) if base == "csv" else {}
    return flatten.dispersed_batch(models.Rocket(), H.make_motor(kind), models.WindModel(), H.EXAMPLE_IC, pl,
                                   planar=planar, **kw)


# ------------------------------------------------------------------ vs golden (the reference itself)
@pytest.mark.parametrize("name", ["flights_named", "flights_planar", "flights_mc"])
def test_fp64_vs_reference_golden(engine, oracle, name):
    idx, arr = H.load_flights(name)
    for g, entries in H.group_flights(idx).items():
        cfg = H.make_config(g[0])
        hb = H.batch_from_golden(entries, arr)
        summ, status = run_gpu(engine, cfg, hb)
        osum, ostat = oracle.run_batch(cfg, hb)
        for i, e in enumerate(entries):
            s = e["summary"]
            assert summ[_abi.SUM_RAIL_EXIT_TIME, i] == s["rail_exit_time"]
            assert relerr(summ[_abi.SUM_RAIL_EXIT_SPEED, i], s["rail_exit_speed"]) < 1e-12
            healthy = np.isfinite(s["range"]) and s["range"] < 1e5 and s["n_steps"] > 20000
            tol = 1e-9 if healthy else 1e-6
            assert relerr(summ[_abi.SUM_FIRST_APOGEE_ALT, i], s["first_apogee"]) < tol, (e["key"],)
            if healthy:
                assert int(summ[_abi.SUM_STEPS, i]) == s["n_steps"]
                assert relerr(summ[_abi.SUM_APOGEE_ALT, i], s["apogee_altitude"]) < 1e-9
                assert relerr(summ[_abi.SUM_RANGE, i], s["range"]) < 1e-7
                assert relerr(summ[_abi.SUM_FLIGHT_TIME, i], s["flight_time"]) < 1e-14
                assert (status[i] & 0xFF) == _abi.END_GROUND and (status[i] & _abi.ST_CHUTE)
            elif np.isfinite(s["apogee_altitude"]):
                assert relerr(summ[_abi.SUM_APOGEE_ALT, i], s["apogee_altitude"]) < 1e-5, (e["key"],)
            else:
                assert status[i] & _abi.ST_NAN
        # and the oracle agrees on how every trajectory ended
        assert np.array_equal(status & 0xFF, ostat & 0xFF)


# ------------------------------------------------------------------ vs oracle, BASELINE config 2
def test_cfg2_set_r_1k_fp64_match_rate(engine, oracle):
    """1k reference-faithful dispersed samples (seed=i stream, example config, CSV wind), fp64,
    full reference termination logic.  Gate: apogee match-rate at 0.1 %."""
    hb = mc_batch("liquid", 1000)
    cfg = H.make_config("liquid")
    summ, status = run_gpu(engine, cfg, hb)
    osum, ostat = oracle.run_batch(cfg, hb)
    e_ap = relerr(summ[_abi.SUM_APOGEE_ALT], osum[_abi.SUM_APOGEE_ALT])
    e_fa = relerr(summ[_abi.SUM_FIRST_APOGEE_ALT], osum[_abi.SUM_FIRST_APOGEE_ALT])
    rate_ap = np.mean(e_ap <= 1e-3)
    rate_fa = np.mean(e_fa <= 1e-3)
    print(f"cfg2 Set R fp64: apogee match {rate_ap:.4f}, first-apogee match {rate_fa:.4f}, "
          f"median err {np.median(e_ap):.2e}/{np.median(e_fa):.2e}, "
          f"same end reason {np.mean((status & 0xFF) == (ostat & 0xFF)):.4f}, "
          f"same step count {np.mean(summ[_abi.SUM_STEPS] == osum[_abi.SUM_STEPS]):.4f}")
    # the correctness gate: every sample (tests/golden/sensitivity.json - the same source with another rounding pattern
    # keeps 4000 / 4000 outcomes - is what makes "all of them" the right bar)
    assert rate_ap == 1.0 and rate_fa == 1.0
    assert np.array_equal(status & 0xFF, ostat & 0xFF)
    assert np.array_equal(summ[_abi.SUM_STEPS], osum[_abi.SUM_STEPS])
    # rail phase is exact
    assert np.array_equal(summ[_abi.SUM_RAIL_EXIT_TIME], osum[_abi.SUM_RAIL_EXIT_TIME])
    assert np.max(relerr(summ[_abi.SUM_RAIL_EXIT_SPEED], osum[_abi.SUM_RAIL_EXIT_SPEED])) < 1e-12


def test_cfg2_set_p_to_apogee_fp64(engine, oracle):
    """Planar healthy dispersions integrated to apogee (BASELINE cfg 2 'to apogee'): 1e-9."""
    hb = mc_batch("liquid", 256, planar=True)
    cfg = H.make_config("liquid")
    summ, status = run_gpu(engine, cfg, hb, flags=_abi.FLAG_STOP_AT_APOGEE)
    osum, ostat = oracle.run_batch(cfg, hb, flags=_abi.FLAG_STOP_AT_APOGEE)
    assert np.array_equal(status, ostat)
    assert np.array_equal(summ[_abi.SUM_STEPS], osum[_abi.SUM_STEPS])
    for row in (_abi.SUM_FIRST_APOGEE_ALT, _abi.SUM_APOGEE_ALT, _abi.SUM_RANGE, _abi.SUM_MAX_SPEED):
        assert np.max(relerr(summ[row], osum[row])) < 1e-9, row
    assert np.array_equal(summ[_abi.SUM_FLIGHT_TIME], osum[_abi.SUM_FLIGHT_TIME])
    assert np.array_equal(summ[_abi.SUM_FIRST_APOGEE_TIME], osum[_abi.SUM_FIRST_APOGEE_TIME])


@pytest.mark.parametrize("kind,base", [("solid", "csv"), ("liquid", "none"), ("solid", "none")])
def test_motor_and_wind_variants_fp64(engine, oracle, kind, base):
    """Solid thrust curve, 100-knot synthetic wind."""
    hb = mc_batch(kind, 128, base=base, planar=True)
    cfg = H.make_config(kind)
    summ, status = run_gpu(engine, cfg, hb, flags=_abi.FLAG_STOP_AT_APOGEE)
    osum, ostat = oracle.run_batch(cfg, hb, flags=_abi.FLAG_STOP_AT_APOGEE)
    assert np.array_equal(status, ostat)
    assert np.max(relerr(summ[_abi.SUM_FIRST_APOGEE_ALT], osum[_abi.SUM_FIRST_APOGEE_ALT])) < 1e-9
    assert np.max(relerr(summ[_abi.SUM_RANGE], osum[_abi.SUM_RANGE])) < 1e-8


def test_no_wind_profile_fp64(engine, oracle):
    hb = mc_batch("liquid", 64, planar=True)
    hb0 = flatten.HostBatch(hb.n, 0)
    hb0.ic, hb0.rocket, hb0.motor = hb.ic, hb.rocket, hb.motor
    cfg = H.make_config("liquid")
    summ, status = run_gpu(engine, cfg, hb0)
    osum, ostat = oracle.run_batch(cfg, hb0)
    assert np.array_equal(status & 0xFF, ostat & 0xFF)
    assert np.max(relerr(summ[_abi.SUM_APOGEE_ALT], osum[_abi.SUM_APOGEE_ALT])) < 1e-9
    ok = osum[_abi.SUM_RANGE] < 1e5
    assert np.max(relerr(summ[_abi.SUM_RANGE], osum[_abi.SUM_RANGE])[ok]) < 1e-7


def test_full_flight_with_parachute_fp64(engine, oracle):
    """cfg 5 ingredient: CSV wind + parachute-deploy event, flights to touchdown."""
    hb = mc_batch("liquid", 64, planar=True)
    cfg = H.make_config("liquid")
    summ, status = run_gpu(engine, cfg, hb)
    osum, ostat = oracle.run_batch(cfg, hb)
    assert np.array_equal(status & 0xFF, ostat & 0xFF)
    landed = ((ostat & 0xFF) == _abi.END_GROUND) & (osum[_abi.SUM_RANGE] < 1e5)
    assert landed.sum() > 40
    assert np.all((status[landed] & _abi.ST_CHUTE) != 0)
    assert np.array_equal(summ[_abi.SUM_STEPS][landed], osum[_abi.SUM_STEPS][landed])
    assert np.max(relerr(summ[_abi.SUM_RANGE], osum[_abi.SUM_RANGE])[landed]) < 1e-7
    assert np.max(relerr(summ[_abi.SUM_FINAL_VZ], osum[_abi.SUM_FINAL_VZ])[landed]) < 1e-7


def test_nan_trajectories_fast_forward_is_exact(engine, oracle):
    """Non-finite trajectories run to max_time in the reference.  The kernel
    fast-forwards them through the host-tabulated time accumulation: steps and flight_time
    must equal the oracle's brute-force loop exactly."""
    hb = mc_batch("liquid", 64, stream="seed_42")
    cfg = H.make_config("liquid")
    summ, status = run_gpu(engine, cfg, hb)
    osum, ostat = oracle.run_batch(cfg, hb)
    nanrun = ((ostat & 0xFF) == _abi.END_MAX_TIME)
    assert nanrun.sum() >= 3
    assert np.array_equal((status & 0xFF)[nanrun], (ostat & 0xFF)[nanrun])
    assert np.array_equal(summ[_abi.SUM_STEPS][nanrun], osum[_abi.SUM_STEPS][nanrun])
    assert np.array_equal(summ[_abi.SUM_FLIGHT_TIME][nanrun], osum[_abi.SUM_FLIGHT_TIME][nanrun])
    assert np.all(np.isnan(summ[_abi.SUM_APOGEE_ALT][nanrun]))
    assert np.all((status[nanrun] & _abi.ST_NAN) != 0)
    # apogee_time = time of the FIRST NaN altitude (np.argmax semantics)
    same = np.abs(summ[_abi.SUM_APOGEE_TIME][nanrun] - osum[_abi.SUM_APOGEE_TIME][nanrun]) <= 0.0051
    assert np.all(same)


# ------------------------------------------------------------------ fp32 kernel
def test_fp32_first_apogee_within_0p1_percent(engine, oracle):
    hb = mc_batch("liquid", 512, planar=True)
    cfg = H.make_config("liquid")
    summ, status = run_gpu(engine, cfg, hb, prec=_abi.PREC_F32, flags=_abi.FLAG_STOP_AT_APOGEE)
    osum, ostat = oracle.run_batch(cfg, hb, flags=_abi.FLAG_STOP_AT_APOGEE)
    e = relerr(summ[_abi.SUM_FIRST_APOGEE_ALT], osum[_abi.SUM_FIRST_APOGEE_ALT])
    print(f"fp32 Set P: first-apogee max rel err {e.max():.2e}, median {np.median(e):.2e}")
    assert e.max() < 1e-3
    assert np.max(np.abs(summ[_abi.SUM_STEPS] - osum[_abi.SUM_STEPS])) <= 25  # apogee time within 0.125 s
    assert np.array_equal(summ[_abi.SUM_RAIL_EXIT_TIME], osum[_abi.SUM_RAIL_EXIT_TIME])
    assert np.max(relerr(summ[_abi.SUM_RAIL_EXIT_SPEED], osum[_abi.SUM_RAIL_EXIT_SPEED])) < 2e-5


@pytest.mark.parametrize("kind,base", [("solid", "csv"), ("liquid", "none"), ("solid", "none"), ("liquid", "nowind")])
def test_fp32_kernel_specialisations(engine, oracle, kind, base):
    """The fp32 flight kernel is compiled in four specialisations (wind table present or not x
    liquid / solid motor, chosen by the launcher): each one against the oracle, same 0.1 % bar."""
    hb = mc_batch(kind, 192, base="none" if base == "nowind" else base, planar=True)
    if base == "nowind":  # k_wind = 0: the RHS sees still air (simulate_flight without a profile)
        hb0 = flatten.HostBatch(hb.n, 0)
        hb0.ic, hb0.rocket, hb0.motor = hb.ic, hb.rocket, hb.motor
        hb = hb0
    cfg = H.make_config(kind)
    summ, status = run_gpu(engine, cfg, hb, prec=_abi.PREC_F32, flags=_abi.FLAG_STOP_AT_APOGEE)
    osum, ostat = oracle.run_batch(cfg, hb, flags=_abi.FLAG_STOP_AT_APOGEE)
    assert np.array_equal(status & 0xFF, ostat & 0xFF)
    e = relerr(summ[_abi.SUM_FIRST_APOGEE_ALT], osum[_abi.SUM_FIRST_APOGEE_ALT])
    print(f"fp32 {kind}/{base}: first-apogee max rel err {e.max():.2e}")
    assert e.max() < 1e-3
    assert np.max(relerr(summ[_abi.SUM_MAX_SPEED], osum[_abi.SUM_MAX_SPEED])) < 1e-3
    assert np.max(np.abs(summ[_abi.SUM_STEPS] - osum[_abi.SUM_STEPS])) <= 25


def test_fp32_set_r_match_rate(engine, oracle):
    """Diverging samples: fp32 parity is only meaningful on the first-descent apogee (fact 6)."""
    hb = mc_batch("liquid", 1000)
    cfg = H.make_config("liquid")
    summ, status = run_gpu(engine, cfg, hb, prec=_abi.PREC_F32)
    osum, ostat = oracle.run_batch(cfg, hb)
    e_fa = relerr(summ[_abi.SUM_FIRST_APOGEE_ALT], osum[_abi.SUM_FIRST_APOGEE_ALT])
    e_ap = relerr(summ[_abi.SUM_APOGEE_ALT], osum[_abi.SUM_APOGEE_ALT])
    print(f"fp32 Set R: first-apogee match-rate@1e-3 {np.mean(e_fa <= 1e-3):.4f}, "
          f"apogee(argmax) match-rate@1e-3 {np.mean(e_ap <= 1e-3):.4f}")
    assert np.mean(e_fa <= 1e-3) >= 0.90


def test_fp32_full_flight_landing(engine, oracle):
    hb = mc_batch("liquid", 128, planar=True)
    cfg = H.make_config("liquid")
    summ, status = run_gpu(engine, cfg, hb, prec=_abi.PREC_F32)
    osum, ostat = oracle.run_batch(cfg, hb)
    landed = ((ostat & 0xFF) == _abi.END_GROUND) & (osum[_abi.SUM_RANGE] < 1e5)
    agree = (status & 0xFF)[landed] == _abi.END_GROUND
    # the post-apogee descent tumbles (stall model + destabilising yaw term) and is chaotic: a
    # few percent of fp32 trajectories leave the fp64 solution there; the rate is
    # reported, apogee and flight time are asserted on the ones that stay together
    print(f"fp32 full flight: {agree.mean():.3f} of fp64-landed samples also land in fp32")
    assert agree.mean() > 0.80
    sel = np.where(landed)[0][agree]
    assert np.max(relerr(summ[_abi.SUM_APOGEE_ALT][sel], osum[_abi.SUM_APOGEE_ALT][sel])) < 1e-3
    # touchdown time after a tumbling descent + parachute phase: a fraction of a second in ~220 s
    assert np.median(relerr(summ[_abi.SUM_FLIGHT_TIME][sel], osum[_abi.SUM_FLIGHT_TIME][sel])) < 1e-2


# ------------------------------------------------------------------ structure of the launch
def test_launch_geometry_does_not_change_results(engine):
    """Sharding equivalence (SURVEY §4 iv): block size, resident-block cap and refill threshold
    only change which lane integrates which sample -> bitwise identical summaries."""
    hb = mc_batch("liquid", 1500)
    cfg = H.make_config("liquid")
    base = None
    try:
        for block, max_blocks, refill, chunk in ((256, 0, 8, 0), (64, 0, 1, 0), (256, 2, 1, 0), (128, 3, 64, 0),
                                                 (64, 5, 17, 0), (256, 0, 8, 256), (256, 0, 8, 100), (64, 3, 5, 977),
                                                 (128, 0, 64, 31)):
            engine.set_launch(block, max_blocks, refill)
            engine.set_chunk(chunk)
            for prec in (_abi.PREC_F64, _abi.PREC_F32):
                summ, status = run_gpu(engine, cfg, hb, prec=prec)
                if base is None:
                    base = {}
                if prec not in base:
                    base[prec] = (summ, status)
                else:
                    assert np.array_equal(status, base[prec][1]), (block, max_blocks, refill, chunk, prec)
                    assert np.array_equal(summ, base[prec][0], equal_nan=True), (block, max_blocks, refill, chunk, prec)
    finally:
        engine.set_launch(256, 0, 1)
        engine.set_chunk(0)


@pytest.mark.parametrize("n", [1, 63, 65, 257])
def test_ragged_batch_sizes(engine, oracle, n):
    hb = mc_batch("liquid", n, planar=True)
    cfg = H.make_config("liquid")
    summ, status = run_gpu(engine, cfg, hb, flags=_abi.FLAG_STOP_AT_APOGEE)
    osum, ostat = oracle.run_batch(cfg, hb, flags=_abi.FLAG_STOP_AT_APOGEE)
    assert np.array_equal(status, ostat)
    assert np.max(relerr(summ[_abi.SUM_FIRST_APOGEE_ALT], osum[_abi.SUM_FIRST_APOGEE_ALT])) < 1e-9


def test_trajectory_capture(engine, oracle):
    hb = mc_batch("liquid", 96, planar=True)
    cfg = H.make_config("liquid")
    ids = [0, 17, 95]
    summ, status, traj, tlen = run_gpu(engine, cfg, hb, traj_ids=ids, traj_stride=50, traj_cap=1200)
    osum, ostat, otraj, otlen = oracle.run_batch(cfg, hb, traj_ids=ids, traj_stride=50, traj_cap=1200)
    assert np.array_equal(tlen, otlen)
    for m in range(len(ids)):
        k = int(tlen[m])
        healthy = osum[_abi.SUM_RANGE, ids[m]] < 1e5
        assert np.array_equal(traj[m, :k, 0], otraj[m, :k, 0])  # time stamps are exact
        if healthy:
            scale = np.maximum(np.abs(otraj[m, :k, 1:]), 1e-6)
            assert np.max(np.abs(traj[m, :k, 1:] - otraj[m, :k, 1:]) / scale) < 1e-6
    # capturing must not change the summaries
    s2, st2 = run_gpu(engine, cfg, hb)
    assert np.array_equal(summ, s2, equal_nan=True) and np.array_equal(status, st2)


def test_error_paths(engine):
    import ctypes as C
    lib = engine.lib
    ctx = C.c_void_p()
    assert lib.erpl_mc_create(0, C.byref(ctx)) == 0
    b, o = _abi.ErplBatch(), _abi.ErplOut()
    b.n = 4
    assert lib.erpl_mc_run_batch(ctx, C.byref(b), C.byref(o), None) == -4  # no config yet
    cfg = H.make_config("liquid")
    cfg.n_cd = 0
    assert lib.erpl_mc_set_config(ctx, C.byref(cfg)) == -1
    cfg = H.make_config("liquid")
    cfg.cd_mach[1] = float("nan")
    assert lib.erpl_mc_set_config(ctx, C.byref(cfg)) == -1
    cfg = H.make_config("liquid")
    cfg.dt_initial = 1e-12   # 3e14 steps: int32 step counter / `t += dt` cannot advance -> refused, not hung
    assert lib.erpl_mc_set_config(ctx, C.byref(cfg)) == -1
    cfg = H.make_config("liquid")
    assert lib.erpl_mc_set_config(ctx, C.byref(cfg)) == 0
    assert lib.erpl_mc_run_batch(ctx, C.byref(b), C.byref(o), None) == -1  # NULL buffers
    assert b"NULL" in lib.erpl_mc_last_error()
    b.n = 0
    assert lib.erpl_mc_run_batch(ctx, C.byref(b), C.byref(o), None) == 0   # empty batch is a no-op
    assert lib.erpl_mc_set_launch(ctx, 100, 0, 8) == -1
    assert lib.erpl_mc_create(99, C.byref(C.c_void_p())) == -1
    assert lib.erpl_mc_destroy(ctx) == 0


def test_extract_histories_vs_oracle(engine, oracle):
    """erpl_mc_extract_histories on GPU-integrated records == oracle evaluation of the same records."""
    from erpl_monte_carlo_sim_amd.engine import DeviceBatch
    for kind in ("liquid", "solid"):
        hb = mc_batch(kind, 8, planar=(kind == "liquid"))
        cfg = H.make_config(kind)
        engine.set_config(cfg)
        db = DeviceBatch.from_host(hb, engine.device, _abi.PREC_F64)
        summ, status, traj, tlen = engine.run(db, traj_ids=[3], traj_stride=37, traj_cap=2000)
        torch.cuda.synchronize()
        k = int(tlen[0].item())
        t_rail = float(summ[_abi.SUM_RAIL_EXIT_TIME, 3].item())
        got = engine.extract_histories(db, 3, traj[0, :k], t_rail).cpu().numpy()
        exp = oracle.extract(cfg, hb.take([3]), traj[0, :k].cpu().numpy(), t_rail)
        assert np.array_equal(np.isfinite(got), np.isfinite(exp))
        fin = np.isfinite(exp)
        scale = np.maximum(np.abs(exp), 1e-9 * np.nanmax(np.abs(np.where(fin, exp, 0)), axis=0, keepdims=True) + 1e-300)
        assert np.max((np.abs(got - exp) / scale)[fin]) < 1e-9, kind


def test_chunked_compaction_with_trajectory_capture(engine, oracle):
    """Step-chunked launches + resume queue must also carry the trajectory-capture state."""
    hb = mc_batch("liquid", 40, planar=True)
    cfg = H.make_config("liquid")
    ids = [1, 39]
    try:
        engine.set_chunk(0)
        ref = run_gpu(engine, cfg, hb, traj_ids=ids, traj_stride=50, traj_cap=1200)
        engine.set_chunk(300)
        got = run_gpu(engine, cfg, hb, traj_ids=ids, traj_stride=50, traj_cap=1200)
    finally:
        engine.set_chunk(0)
    for a, b in zip(ref, got):
        assert np.array_equal(a, b, equal_nan=True)


# ------------------------------------------------------------------ full size (BASELINE cfg 3 / 4 share)
@pytest.mark.parametrize("precision", ["f32", "f64_fast"])
def test_full_size_batch_properties(engine, oracle, precision):
    """131 072 dispersed samples (one GPU's share of BASELINE config 3, the bench shard's shape: K = 100 wind knots),
    reference dispersion model drawn on the device, in both throughput builds.  The oracle cannot integrate that
    many in a test, so the full-size run is checked through properties that do not depend on the size:
      * launch geometry / compaction independence: bitwise identical summaries,
      * sample independence: any sub-batch integrated alone reproduces its rows bit for bit,
      * bookkeeping: every sample ends exactly once, with a reason, flags consistent with the values,
      * and a random 256-sample subset against the oracle at the 0.1 % bar: the first-descent apogee for fp32, the
        reference's apogee_altitude, end reason and step count for the fp64 throughput build."""
    from erpl_monte_carlo_sim_amd import sampling
    from erpl_monte_carlo_sim_amd.engine import DeviceBatch
    n = 131072
    prec = _abi.PRECISIONS[precision]
    rocket, motor, wm = models.Rocket(), models.LiquidMotor(), models.WindModel()
    cfg = H.make_config("liquid")
    engine.set_config(cfg)
    db = sampling.synthetic_dispersions(n, rocket, motor, wm, H.EXAMPLE_IC, engine.device, precision=prec, seed=77)
    try:
        s0, t0 = (x.clone() for x in engine.run(db))
        engine.set_launch(128, 300, 16)
        engine.set_chunk(700)
        s1, t1 = (x.clone() for x in engine.run(db))
    finally:
        engine.set_launch(256, 0, 1)
        engine.set_chunk(0)
    torch.cuda.synchronize()
    assert torch.equal(t0, t1)
    assert bool(((s0 == s1) | (s0.isnan() & s1.isnan())).all())
    # a sub-batch alone (strided pick, so lanes / waves / queue order all differ)
    idx = torch.arange(5, n, 257, device=engine.device)
    sub = DeviceBatch(db.ic[:, idx].contiguous(), db.rocket[:, idx].contiguous(), db.motor[:, idx].contiguous(),
                      db.alt_grid, db.wind[:, :, idx].contiguous(), prec)
    s2, t2 = engine.run(sub)
    torch.cuda.synchronize()
    assert torch.equal(t2, t0[idx])
    assert bool(((s2 == s0[:, idx]) | (s2.isnan() & s0[:, idx].isnan())).all())
    summ, status = s0.cpu().numpy(), t0.cpu().numpy()
    reason = status & 0xFF
    assert np.all(reason <= _abi.END_APOGEE) and np.sum(np.bincount(reason, minlength=5)) == n
    assert np.all(summ[_abi.SUM_STEPS] >= 1) and np.all(summ[_abi.SUM_FLIGHT_TIME] > 0)
    assert np.all(summ[_abi.SUM_FLIGHT_TIME] <= cfg.max_time + 2 * cfg.dt_initial)
    nan_flag = (status & _abi.ST_NAN) != 0
    assert np.array_equal(nan_flag, np.isnan(summ[_abi.SUM_APOGEE_ALT]))          # argmax hit a NaN altitude
    assert np.all(reason[np.isnan(summ[_abi.SUM_IMPACT_Z])] == _abi.END_MAX_TIME)   # NaN states only end by time
    grounded = reason == _abi.END_GROUND
    assert np.all(summ[_abi.SUM_IMPACT_Z][grounded] <= 0.5) and np.all(summ[_abi.SUM_FINAL_VZ][grounded] <= 0)
    assert np.all(summ[_abi.SUM_IMPACT_Z][reason == _abi.END_ALTITUDE] > 100000.0)
    fin = np.isfinite(summ[_abi.SUM_APOGEE_ALT]) & np.isfinite(summ[_abi.SUM_FIRST_APOGEE_ALT])
    assert np.all(summ[_abi.SUM_APOGEE_ALT][fin] >= summ[_abi.SUM_FIRST_APOGEE_ALT][fin])  # global max >= latched max
    # random subset vs the oracle
    pick = np.sort(np.random.RandomState(3).choice(n, 256, replace=False))
    hb = flatten.HostBatch(len(pick), db.k_wind)
    tp = torch.as_tensor(pick, device=engine.device)
    hb.ic = db.ic[:, tp].cpu().numpy(); hb.rocket = db.rocket[:, tp].cpu().numpy(); hb.motor = db.motor[:, tp].cpu().numpy()
    hb.alt_grid = db.alt_grid.cpu().numpy(); hb.wind = db.wind[:, :, tp].double().cpu().numpy()
    osum, ostat = oracle.run_batch(cfg, hb)
    e = relerr(summ[_abi.SUM_FIRST_APOGEE_ALT][pick], osum[_abi.SUM_FIRST_APOGEE_ALT])
    print(f"full-size {precision} subset: first-apogee match-rate@1e-3 {np.mean(e <= 1e-3):.3f}")
    assert np.mean(e <= 1e-3) >= (0.95 if precision == "f32" else 1.0)
    assert np.array_equal(summ[_abi.SUM_RAIL_EXIT_TIME][pick], osum[_abi.SUM_RAIL_EXIT_TIME])
    if precision == "f64_fast":   # the build that has to carry north_star's apogee bar on the diverging samples too
        ea = relerr(summ[_abi.SUM_APOGEE_ALT][pick], osum[_abi.SUM_APOGEE_ALT])
        same_end = (status[pick] & 0xFF) == (ostat & 0xFF)
        print(f"full-size f64_fast subset: apogee match-rate@1e-3 {np.mean(ea <= 1e-3):.4f}, same end reason {np.mean(same_end):.4f}")
        assert np.mean(ea <= 1e-3) == 1.0 and np.mean(same_end) == 1.0       # (blow-ups finish in the reference-order kernel)
        assert np.array_equal(summ[_abi.SUM_STEPS][pick], osum[_abi.SUM_STEPS])


def test_two_and_three_wave_builds_agree_bitwise(engine):
    """The fp32 flight kernel exists in a 256-VGPR build (two resident waves per SIMD) and a 168-VGPR
    build (three, with spills), picked by batch size: same arithmetic, so identical bits - a sample's
    result must not depend on the size of the batch it travels in."""
    from erpl_monte_carlo_sim_amd import sampling
    rocket, wm = models.Rocket(), models.WindModel()
    try:
        for kind, csv in (("liquid", False), ("solid", True)):
            motor = H.make_motor(kind)
            engine.set_config(H.make_config(kind))
            db = sampling.synthetic_dispersions(20000, rocket, motor, wm, H.EXAMPLE_IC, engine.device,
                                                precision=_abi.PREC_F32, seed=5, planar=csv,
                                                base_altitude_profile=H.CSV_ALT if csv else None,
                                                base_wind_profile=H.CSV_WIND if csv else None)
            out = {}
            for w in (2, 3):
                engine.set_waves_per_simd(w)
                s, t = engine.run(db)
                out[w] = (s.clone(), t.clone())
            torch.cuda.synchronize()
            assert torch.equal(out[2][1], out[3][1]), kind
            assert bool(((out[2][0] == out[3][0]) | (out[2][0].isnan() & out[3][0].isnan())).all()), kind
    finally:
        engine.set_waves_per_simd(0)


# ------------------------------------------------------------------ configuration space
@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_randomised_configurations(engine, oracle, seed):
    """Rocket / motor / atmosphere / simulator attributes away from the reference's defaults: table
    sizes up to the ABI limits, other time steps, rail lengths, parachute altitudes, damping, wind grids
    with 2..1024 knots.  fp64 kernel vs oracle at the healthy-flight bar, the fp64 throughput build (whose Mach and
    atmosphere records are read by index from the workgroup's tables and whose altitude grid sits in dynamic LDS)
    at 1e-8 on the same healthy flights, fp32 at 0.1 %."""
    rs = np.random.RandomState(seed)
    kind = "solid" if seed % 2 else "liquid"
    rocket, motor = models.Rocket(), H.make_motor(kind)
    rocket.diameter *= rs.uniform(0.8, 1.3)
    rocket.reference_area = np.pi * (rocket.diameter / 2) ** 2
    rocket.reference_diameter = rocket.diameter
    rocket.center_of_mass_dry = rs.uniform(4.6, 5.2)    # ahead of the CP: statically stable flights
    rocket.fin_span *= rs.uniform(0.8, 1.4)
    rocket.fin_sweep_angle = rs.uniform(0.0, 0.5)
    rocket.parachute_deployment_altitude = rs.uniform(200.0, 3000.0)
    rocket.parachute_cd = rs.uniform(0.8, 2.5)
    rocket.power_off_drag_factor = rs.uniform(1.0, 1.5)
    nm = int(rs.choice([2, 5, 11, _abi.MAX_MACH_KNOTS]))
    mach = np.sort(np.concatenate([[0.0], rs.uniform(0.05, 6.0, nm - 1)]))
    rocket.Cd_data = {"mach": list(mach), "cd0": list(rs.uniform(0.3, 0.7, nm)), "cda": list(rs.uniform(1.0, 1.5, nm))}
    nc = int(rs.choice([2, 7, _abi.MAX_MACH_KNOTS]))
    rocket.CP_shift_data = {"mach": list(np.sort(rs.uniform(0.0, 5.0, nc))), "cp_shift": list(rs.uniform(-0.1, 0.05, nc))}
    rocket.cp_location = rocket._calculate_center_of_pressure()
    if kind == "solid":
        nt = int(rs.choice([3, 10, _abi.MAX_CURVE_KNOTS]))
        motor.thrust_curve_time = np.concatenate([[0.0], np.sort(rs.uniform(0.1, 14.0, nt - 2)), [15.0]])
        motor.thrust_curve_thrust = np.concatenate([[0.0], rs.uniform(4000.0, 16000.0, nt - 2), [0.0]])
    atm = models.StandardAtmosphere()
    atm.sea_level_temperature = rs.uniform(270.0, 305.0)
    atm.sea_level_pressure = rs.uniform(95000.0, 104000.0)
    cfg = flatten.config_from_objects(rocket, motor, atm, dt_initial=float(rs.choice([0.002, 0.01, 0.02])),
                                      max_time=float(rs.choice([40.0, 120.0, 300.0])),
                                      pitch_damping=rs.uniform(200.0, 2000.0), yaw_damping=rs.uniform(200.0, 2000.0))
    cfg.rail_length = float(rs.choice([0.0, 3.0, 18.0]))
    n = 96
    pl = flatten.generate_parameter_samples(H.UNCERTAINTY, n)
    k = int(rs.choice([2, 37, _abi.MAX_WIND_KNOTS]))
    alt = np.sort(np.concatenate([[0.0], rs.uniform(10.0, 40000.0, k - 1)]))
    basew = np.stack([rs.uniform(-8, 8, k), np.zeros(k), rs.uniform(-0.5, 0.5, k)], axis=1)
    hb = flatten.dispersed_batch(rocket, motor, models.WindModel(), H.EXAMPLE_IC, pl, alt, basew, planar=True)
    flags = _abi.FLAG_STOP_AT_APOGEE if seed % 3 else 0
    osum, ostat = oracle.run_batch(cfg, hb, flags=flags)
    summ, status = run_gpu(engine, cfg, hb, flags=flags)
    assert np.array_equal(status & 0xFF, ostat & 0xFF)
    same = (status == ostat) & np.isfinite(osum[_abi.SUM_RANGE]) & (osum[_abi.SUM_RANGE] < 1e5)
    assert same.sum() >= 0.7 * n
    assert np.array_equal(summ[_abi.SUM_RAIL_EXIT_TIME], osum[_abi.SUM_RAIL_EXIT_TIME])
    for row, tol in ((_abi.SUM_FIRST_APOGEE_ALT, 1e-8), (_abi.SUM_APOGEE_ALT, 1e-8), (_abi.SUM_MAX_SPEED, 1e-8)):
        assert np.max(relerr(summ[row][same], osum[row][same])) < tol, (row, seed)
    assert np.array_equal(summ[_abi.SUM_STEPS][same], osum[_abi.SUM_STEPS][same])
    sf, tf = run_gpu(engine, cfg, hb, prec=_abi.PREC_F64_FAST, flags=flags)
    same_f = same & (tf == ostat)
    assert np.array_equal(tf & 0xFF, ostat & 0xFF) and same_f.sum() == same.sum(), (seed, same_f.sum(), same.sum())
    assert np.array_equal(sf[_abi.SUM_RAIL_EXIT_TIME], osum[_abi.SUM_RAIL_EXIT_TIME])
    for row in (_abi.SUM_FIRST_APOGEE_ALT, _abi.SUM_APOGEE_ALT, _abi.SUM_MAX_SPEED):
        assert np.max(relerr(sf[row][same_f], osum[row][same_f])) < 1e-8, (row, seed)
    assert np.array_equal(sf[_abi.SUM_STEPS][same_f], osum[_abi.SUM_STEPS][same_f])
    s32, t32 = run_gpu(engine, cfg, hb, prec=_abi.PREC_F32, flags=flags)
    e = relerr(s32[_abi.SUM_FIRST_APOGEE_ALT][same], osum[_abi.SUM_FIRST_APOGEE_ALT][same])
    print(f"config {seed} ({kind}, K={k}, mach knots {nm}/{nc}): healthy {same.sum()}/{n}, fp32 first-apogee max err {e.max():.2e}")
    assert np.mean(e <= 1e-3) >= 0.97
