#!/usr/bin/env python3
"""Headline benchmark: Monte Carlo trajectories/sec + apogee-match rate (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (rail kernel + RK4 flight kernel [+ RCCL all-gather of the
per-sample summaries when N > 1]) over one batch of synthetic dispersions ALREADY RESIDENT in HBM.
Every pass integrates a DIFFERENT shard (all of them generated on the device before the warm-up; `--shards 1`
replays one shard, as rounds 1-3 did: cache-warm inputs, identical end times, and shard 0's one 16 029-step
trajectory as the tail of every pass - DESIGN.md section 3.1).  The first timed pass runs shard 0: parity, end
reasons and the CPU baseline refer to it.
Workload at N = 1: 131 072 dispersed samples per GPU (BASELINE configs[2]; x 8 GPUs = the 1 048 576
samples of configs[3]), LiquidMotor, reference dispersion model (monte_carlo.py:156-179), synthetic
100-knot wind profile, full reference termination logic.  Weak scaling (default): every rank integrates its
own 131 072-sample shard; `--total-samples N` splits a fixed N over the ranks instead (strong scaling:
configs[3]'s 1 048 576 at 1 / 2 / 4 / 8 GPUs).

The headline leg is the fp64 throughput build (ERPL_PREC_F64_FAST): it is the build that meets north_star's
"per-sample apogee within 0.1 %" (apogee_match_rate 1.0 since round 4: samples that blow up - speed above 1e6 m/s -
finish in the reference-order kernel; the fp32 build, which configs[2] names, reproduces the reference's
apogee_altitude on 17 % of these samples and is reported as the secondary `f32` leg).

Passes are handed to the library with erpl_mc_submit_batch: up to `--overlap` of them are in flight
on the library's internal streams (a pass over a batch that just fills the GPU lasts as long as its
longest trajectory; the next pass uses the lanes that have already finished).  All K passes complete
inside the timed region.

Prints ONE JSON line (rank 0).  `value`, `dtype`, `apogee_match_rate`, `roofline` all describe the SAME
kernel build (--precision, default f64_fast); the match rates are those of shard 0:
  apogee_match_rate  fraction of the shard's samples whose `apogee_altitude` (the reference's global
                argmax, simulator.py:488-490) is within 0.1 % of the fp64 reference-order gate kernel
                run on the same inputs (that kernel tracks the CPU oracle on 100 % of the cfg-2 set:
                `parity.cfg2_set_r_1k`); first-descent apogee, end reasons and a breakdown by class
                of the reference outcome are under `parity`.
  roofline      dominant kernel = erpl_flight_<dtype>; the path is vector-ALU bound:
                `achieved` = RK4 steps integrated per launch x 1570 algorithmic flops / the GPU time per
                launch (HIP events on the stream around the timed region / K); the per-dispatch
                duration the library's own events (and rocprofv3) see is reported beside it - with
                `overlap` launches in flight a dispatch lasts about `overlap` times the time per launch.
  cpu_baseline  the CPU oracle (C fp64 restatement, OpenMP over samples, all host cores) on a
                bounded sample of the same shard; a reported baseline, not the target.
  f32           the same K passes with the ERPL_PREC_F32 build: its own value / apogee_match_rate / roofline.
  cfg5_share    BASELINE configs[4]'s per-GPU share (1.25 M samples, CSV wind + parachute, full flights) in the
                headline build, with its 256-sample oracle subset (outside the headline timed region).
  api_end_to_end  MonteCarloAnalyzer.run_monte_carlo / run_monte_carlo_device at 10^6 samples.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the package asks for one hardware queue per batch in flight (GPU_MAX_HW_QUEUES=24, include/erpl_mc.h) BEFORE HIP starts -
# unless something has started HIP already (a profiler's preloaded library): then it leaves the default and warns,
# and the library runs three deep (tools/*.sh export the variable for profiled runs)
import erpl_monte_carlo_sim_amd  # noqa: E402,F401

import numpy as np  # noqa: E402
import torch  # noqa: E402

from erpl_monte_carlo_sim_amd import _abi, flatten, models, sampling  # noqa: E402
from erpl_monte_carlo_sim_amd.engine import DeviceBatch, TrajectoryEngine  # noqa: E402

FLOPS_PER_STEP = 1570.0          # SURVEY.md 8d algorithmic count (4 RHS x 343 + ~200)
PEAK_TFLOPS = {"f32": 157.3, "f64": 78.65, "f64_fast": 78.65}   # MI355X_MICROARCH.md vector peaks
PEAK_HBM_GBPS = 8000.0

EXAMPLE_IC = {"position": [0.0, 0.0, 10.0], "velocity": [0, 0, 0.0],
              "attitude": [0.0, -np.pi / 2 + 0.02, 0.0], "angular_velocity": [0.0, 0.0, 0.0]}
CSV_ALT = np.array([0.0, 5000.0, 10000.0, 15000.0, 20000.0, 25000.0])
CSV_WIND = np.array([[2.0, 0, 0], [5, 1, 0], [8, 2, 0], [10, 2, 0], [12, 3, 0], [15, 3, 0]])


def host_cores():
    """CPU threads this process may actually use: min(affinity mask, cgroup cpu quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = max(1, min(n, int(q / per + 0.5)))
        except Exception:
            pass
    return n


def relerr(a, b):
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        e = np.abs(a - b) / np.abs(b)
    return np.where(same, 0.0, np.where(np.isnan(e), np.inf, e))


def match_report(ref_s, ref_t, got_s, got_t):
    """Match rates of (got) against (ref) at the north-star 0.1 % bar, overall and by class of the
    REFERENCE outcome: apogee reached before the first descent (the flight never topped it afterwards),
    apogee set after the first descent (tumble / blow-up), altitude turned NaN."""
    e_ap = relerr(got_s[_abi.SUM_APOGEE_ALT], ref_s[_abi.SUM_APOGEE_ALT])
    e_fa = relerr(got_s[_abi.SUM_FIRST_APOGEE_ALT], ref_s[_abi.SUM_FIRST_APOGEE_ALT])
    same_end = (got_t & 0xFF) == (ref_t & 0xFF)
    fin = np.isfinite(e_ap)
    out = {"n": int(ref_s.shape[1]),
           "apogee_match_rate_0p1pct": float(np.mean(e_ap <= 1e-3)),
           "first_apogee_match_rate_0p1pct": float(np.mean(e_fa <= 1e-3)),
           "same_end_reason": float(np.mean(same_end)),
           "same_step_count": float(np.mean(got_s[_abi.SUM_STEPS] == ref_s[_abi.SUM_STEPS])),
           # how far from the 0.1 % bar the build is (a regression shows here long before it costs a match)
           "apogee_err_median": float(np.median(e_ap[fin])) if fin.any() else None,
           "apogee_err_p99": float(np.percentile(e_ap[fin], 99)) if fin.any() else None,
           "apogee_err_max_finite": float(e_ap[fin].max()) if fin.any() else None,
           "by_reference_class": {}}
    nan = (ref_t & _abi.ST_NAN) != 0
    calm = (~nan) & (ref_s[_abi.SUM_APOGEE_ALT] == ref_s[_abi.SUM_FIRST_APOGEE_ALT])
    for name, m in (("apogee_before_first_descent", calm), ("apogee_after_first_descent", (~nan) & ~calm),
                    ("altitude_turned_nan", nan)):
        if m.any():
            out["by_reference_class"][name] = {"fraction": float(np.mean(m)),
                                               "apogee_match_rate": float(np.mean(e_ap[m] <= 1e-3)),
                                               "first_apogee_match_rate": float(np.mean(e_fa[m] <= 1e-3)),
                                               "same_end_reason": float(np.mean(same_end[m]))}
    return out


def as_precision(db, prec):
    """The SAME samples for another kernel build (fp64 draws; the wind table rounded for fp32)."""
    wind = db.wind
    if prec == _abi.PREC_F32 and wind.dtype != torch.float32:
        wind = wind.float().contiguous()
    elif prec != _abi.PREC_F32 and wind.dtype != torch.float64:
        raise SystemExit("cannot widen an fp32 wind table")
    return DeviceBatch(db.ic, db.rocket, db.motor, db.alt_grid, wind, prec)


def host_slice(db, m):
    """First m samples of a DeviceBatch as a flatten.HostBatch (fp64) for the oracle."""
    hb = flatten.HostBatch(m, db.k_wind)
    hb.ic = np.ascontiguousarray(db.ic[:, :m].cpu().numpy())
    hb.rocket = np.ascontiguousarray(db.rocket[:, :m].cpu().numpy())
    hb.motor = np.ascontiguousarray(db.motor[:, :m].cpu().numpy())
    hb.alt_grid = db.alt_grid.cpu().numpy().astype(np.float64)
    hb.wind = np.ascontiguousarray(db.wind[:, :, :m].double().cpu().numpy())
    return hb


def cfg5_share(eng, device, rocket, atm, wm, args, n=1250000, passes=2, subset=256):
    """BASELINE configs[4] ("10M samples, CSV wind profile + parachute-deploy event detection, 8xMI355X with per-GPU
    compaction"): one GPU's share - 1.25 M samples, CSV base wind, full flights to the ground under the parachute -
    in the headline (fp64 throughput) build with the library's automatic step chunks, outside the headline timed
    region.  The first `subset` samples are also run through the CPU oracle: landing and parachute latch must agree."""
    from oracle import oracle as orc
    motor = models.LiquidMotor()
    cfg5 = flatten.config_from_objects(rocket, motor, atm)
    eng.set_config(cfg5)
    db = sampling.synthetic_dispersions(n, rocket, motor, wm, EXAMPLE_IC, device, precision=_abi.PREC_F64_FAST, seed=4321,
                                        planar=True, base_altitude_profile=CSV_ALT, base_wind_profile=CSV_WIND, engine=eng)
    depth = eng.get_overlap()
    eng.set_overlap(depth)
    outs = [eng.alloc_outputs(n) for _ in range(2)]
    for i in range(2):          # the library learns the trajectory length from finished batches (automatic step chunks)
        eng.submit(db, summary=outs[i % 2][0], status=outs[i % 2][1])
        eng.wait()
        torch.cuda.synchronize()
    # `passes` batches in flight together (another batch fills every chunk barrier), as a 10 M-sample run would submit them
    outs = [eng.alloc_outputs(n) for _ in range(passes)]
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record()
    for i in range(passes):
        eng.submit(db, summary=outs[i][0], status=outs[i][1])
    eng.wait()
    ev1.record()
    torch.cuda.synchronize()
    eng.synchronize()
    ms = ev0.elapsed_time(ev1) / passes
    steps, wi = eng.last_stats()
    summ, status = outs[-1]
    st = status.cpu().numpy()
    tf = steps * FLOPS_PER_STEP / (ms * 1e-3) / 1e12
    hb = host_slice(db, subset)
    osum, ostat = orc.run_batch(cfg5, hb, threads=host_cores())
    gs, gt = summ[:, :subset].cpu().numpy(), st[:subset]
    rep = match_report(osum, ostat, gs, gt)
    rep.pop("by_reference_class")
    landed_ref = (ostat & 0xFF) == 1
    rep["oracle_landed_under_parachute"] = int(np.sum(landed_ref & ((ostat & _abi.ST_CHUTE) != 0)))
    rep["same_landing_and_parachute_latch"] = float(np.mean(((gt & 0xFF) == (ostat & 0xFF)) &
                                                            ((gt & _abi.ST_CHUTE) == (ostat & _abi.ST_CHUTE))))
    return {
        "workload": f"{n} samples (BASELINE configs[4] / 8 GPUs), planar dispersions, CSV base wind K=6 + AR(1) turbulence, "
                    f"full flights with parachute latch, f64_fast, automatic step chunks, {passes} batches in flight",
        "value": n / ms * 1e3, "unit": "trajectories/s", "ms_per_pass": ms, "dtype": "f64_fast",
        "steps_per_trajectory_mean": steps / n, "lane_utilisation": steps / (64.0 * wi) if wi else None,
        "roofline": {"bound": "valu", "achieved": tf, "peak": PEAK_TFLOPS["f64_fast"], "unit": "TFLOP/s",
                     "frac": tf / PEAK_TFLOPS["f64_fast"], "kernel": "erpl_flight_f64f", "rk4_steps_per_launch": steps},
        "end_reasons": {k: int(np.sum((st & 0xFF) == v)) for k, v in
                        (("max_time", 0), ("ground", 1), ("altitude_100km", 2), ("coast", 3), ("apogee", 4))},
        "parachute_deployed": int(np.sum((st & _abi.ST_CHUTE) != 0)),
        "landed_under_parachute_fraction": float(np.mean(((st & 0xFF) == 1) & ((st & _abi.ST_CHUTE) != 0))),
        "oracle_subset": {"n": subset, **rep},
    }


def api_end_to_end(device, rocket, motor, atm, wm, n=1000000):
    """The named API end to end at BASELINE size (outside the timed region): run_monte_carlo_device (dispersions drawn
    on the device, sub-batches overlapped, statistics on the device) and run_monte_carlo (the reference-exact drop-in:
    per-sample MT19937 streams on the host, chunked pipeline, lazy result records; monte_carlo.py:52-90)."""
    import erpl_monte_carlo_sim_amd as E
    mc = E.MonteCarloAnalyzer(rocket, motor, atm, wm, device=device, verbose=False)
    out = {"n_samples": n}
    for precision in ("f64_fast", "f32"):
        mc.run_monte_carlo_device(dict(EXAMPLE_IC), n, precision=precision)      # warm-up (allocations, first launches)
        r = mc.run_monte_carlo_device(dict(EXAMPLE_IC), n, precision=precision)
        out["run_monte_carlo_device_" + precision] = {**r["performance"], "n_valid": r["n_samples"], "n_outliers": r["n_outliers"]}
    for precision, m in (("f64_fast", n), ("f64", 262144)):   # (the gate kernel: two chunks, so that the pipeline shows)
        mc.precision = precision
        t1 = time.perf_counter()
        try:
            res = mc.run_monte_carlo(dict(EXAMPLE_IC), n_samples=m)
            shape = {"n_valid": res["n_samples"], "n_outliers": res["n_outliers"]}
            t2 = time.perf_counter()
            first = res["results"][0]
            shape["first_record_ms"] = (time.perf_counter() - t2) * 1e3
            shape["record_keys"] = len(first)
        except ValueError as e:      # the reference's own behaviour when no sample survives the outlier rules
            shape = {"raised": str(e)[:60]}
        el = (t2 if "n_valid" in shape else time.perf_counter()) - t1
        out["run_monte_carlo_" + precision] = {"n_samples": m, "total_time": el, "simulations_per_second": m / el, **shape}
    out["note"] = ("run_monte_carlo defaults to precision 'f64_fast' (the fp64 throughput build; its blow-ups finish in the "
                   "reference-order kernel: the same outcomes on every sample of the parity sets, see `parity`); 'f64' runs every "
                   "sample in the reference-order gate kernel")
    return out


def self_launch(n_ranks):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as FRESH child processes (RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, the same command line) BEFORE anything in this process
    touches the GPU, pass their output through (rank 0 prints the ONE JSON line), and return non-zero if any rank
    failed.  No re-exec: this process stays the parent and never initialises HIP (device_count() does not, on this image).
    Replaces the pool start-up of monte_carlo.py:67-83."""
    import socket
    import subprocess
    backend = os.environ.get("ERPL_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < n_ranks:
        print(f"bench.py: {n_ranks} GPUs requested, {ndev} visible (RCCL needs one GPU per rank)", file=sys.stderr, flush=True)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK=str(r),
                   WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, kill_at = 0, None
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code if 0 < code < 256 else 1
                for q in alive:      # a rank that died leaves the others waiting in a collective: end exactly those
                    q.terminate()
                kill_at = time.monotonic() + 15.0
        if kill_at is not None and time.monotonic() > kill_at:
            for q in alive:
                q.kill()
            kill_at = None
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--samples-per-gpu", type=int, default=131072)
    ap.add_argument("--workload", default="set_s", choices=["set_s", "set_p_apogee", "set_p_full", "csv_chute"])
    ap.add_argument("--motor", default="liquid", choices=["liquid", "solid"])
    ap.add_argument("--precision", default="f64_fast", choices=["f32", "f64", "f64_fast"])
    ap.add_argument("--total-samples", type=int, default=0,
                    help="strong scaling: split this many samples over the ranks (0 = weak scaling, --samples-per-gpu each)")
    ap.add_argument("--no-cfg5", action="store_true", help="skip the cfg5_share block")
    ap.add_argument("--no-api", action="store_true", help="skip the api_end_to_end block")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration (0 = skip)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-second-leg", action="store_true", help="skip the secondary leg (f32 beside f64_fast and vice versa)")
    ap.add_argument("--block", type=int, default=0, help="threads per workgroup (0 = library default)")
    ap.add_argument("--max-blocks", type=int, default=0)
    ap.add_argument("--refill", type=int, default=1)
    ap.add_argument("--adopt", type=int, default=-1, help="lane adoption limit (erpl_mc_set_adopt); 0 = off; -1 = library default")
    ap.add_argument("--overlap", type=int, default=-1, help="passes in flight (erpl_mc_set_overlap); 0 = erpl_mc_run_batch on the stream; -1 = library default")
    ap.add_argument("--shards", type=int, default=0, help="distinct input shards the passes cycle through: 0 = one per timed pass (capped at "
                    "48 GB of inputs), 1 = every pass replays one resident shard (what rounds 1-3 timed)")
    ap.add_argument("--short-overlap", type=int, default=-1, help="erpl_mc_set_short_flight_overlap (-1 = library default: 4; 0 = off)")
    ap.add_argument("--waves", type=int, default=0, help="fp32 kernel build: 2 or 3 waves per SIMD (0 = library default)")
    ap.add_argument("--chunk", type=int, default=-1, help="steps per launch between compactions (-1 = library default)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` typed as is: this process becomes the launcher (it never touches the GPU) and the
        # N ranks are fresh children, exactly what torch.distributed.run would start
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU "
                         f"(python bench.py --gpus N starts them itself)")
    ndev = torch.cuda.device_count()
    if world > 1 and os.environ.get("ERPL_BENCH_BACKEND", "nccl") == "nccl" and ndev < world:
        raise SystemExit(f"{world} GPUs requested, {ndev} visible: RCCL needs one GPU per rank")
    dev_index = local_rank % max(ndev, 1)   # one rank per GPU on the driver's 8-GPU node
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    gloo_rehearsal = False
    if world > 1:
        import torch.distributed as dist
        # "nccl" is RCCL on ROCm.  ERPL_BENCH_BACKEND=gloo exists only to rehearse the N > 1 code path on
        # a box with a single GPU (RCCL refuses two ranks on one device); it is never used for numbers.
        backend = os.environ.get("ERPL_BENCH_BACKEND", "nccl")
        gloo_rehearsal = backend != "nccl"
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    rocket, atm, wm = models.Rocket(), models.StandardAtmosphere(), models.WindModel()
    motor = models.SolidMotor() if args.motor == "solid" else models.LiquidMotor()
    cfg = flatten.config_from_objects(rocket, motor, atm)
    eng = TrajectoryEngine(device, lib_path=os.environ.get("ERPL_LIB"))   # ERPL_LIB: an experiment build of the library (A/B runs)
    eng.set_config(cfg)
    if args.block > 0 or args.max_blocks > 0 or args.refill != 1:
        eng.set_launch(args.block if args.block > 0 else 64, args.max_blocks, args.refill)
    if args.chunk >= 0:
        eng.set_chunk(args.chunk)
    eng.set_waves_per_simd(args.waves)
    if args.adopt >= 0:
        eng.set_adopt(args.adopt)
    if args.short_overlap >= 0:
        eng.set_short_flight_overlap(args.short_overlap)
    lib_depth = eng.get_overlap()     # 8 when the process has a hardware queue per batch in flight, else 3

    def leg_depth(precision):
        """Passes in flight for a leg: the library's own default (8 with a hardware queue per stream, else 3) unless
        --overlap says otherwise.  All K passes finish inside the timed region, so the pipeline's fill and drain count."""
        return args.overlap if args.overlap >= 0 else lib_depth

    depth = leg_depth(args.precision)
    n = -(-args.total_samples // world) if args.total_samples > 0 else args.samples_per_gpu
    scaling = "strong" if args.total_samples > 0 else "weak"
    planar = args.workload.startswith("set_p")
    flags = _abi.FLAG_STOP_AT_APOGEE if args.workload == "set_p_apogee" else 0
    csv = args.workload == "csv_chute"
    # one set of samples for every kernel build: fp64 draws, the wind table rounded once for fp32
    shard_cache = {}

    def shard64(j):
        """Shard j of this rank (fp64 draws, resident in HBM): j = 0 is the shard every parity figure refers to."""
        if j not in shard_cache:
            shard_cache[j] = sampling.synthetic_dispersions(
                n, rocket, motor, wm, EXAMPLE_IC, device, precision=_abi.PREC_F64, seed=1234 + rank + 7919 * j, planar=planar or csv,
                base_altitude_profile=CSV_ALT if csv else None, base_wind_profile=CSV_WIND if csv else None, engine=eng)
        return shard_cache[j]
    db64 = shard64(0)

    def n_shards(steps, depth):
        """Distinct shards the passes of a leg cycle through (a replayed shard is cache-warm and its lanes
        end at identical times in every pass): one per timed pass unless --shards says otherwise; fp64 + fp32 copies of
        the inputs stay under 48 GB; erpl_mc_run_batch legs (--overlap 0) replay one shard (no per-ticket counters)."""
        if depth <= 0:
            return 1
        want = args.shards if args.shards > 0 else steps
        return max(1, min(want, steps, int(48e9 // (1.5 * db64.input_bytes()))))
    eng.reserve(n)
    side = torch.cuda.Stream(device) if world > 1 else None

    def timed_leg(precision, steps=None, warmup=None):
        """W warm-up + K timed passes of one kernel build; returns the measurements of this rank."""
        steps = args.steps if steps is None else steps
        warmup = args.warmup if warmup is None else warmup
        prec = _abi.PRECISIONS[precision]
        depth = leg_depth(precision)
        S = n_shards(steps, depth)
        dbs = [as_precision(shard64(j), prec) for j in range(S)]
        db = dbs[0]
        if depth > 0:
            eng.set_overlap(depth)
        # every lane of the library owns two workspaces (a lane's next batch starts while the sweeps of its previous one
        # still write results): pass i + depth may run beside pass i, so the output buffers rotate over 2 x depth sets
        # (+ 1 for the gather that still reads one) - a buffer is reused only behind the batch that last wrote it
        nbuf = 2 * max(depth, 1) + (1 if world > 1 else 0)
        # (set `nbuf` belongs to the FIRST timed pass alone - shard 0, the one the parity figures are about - and is never reused)
        outs = [eng.alloc_outputs(n) for _ in range(nbuf + 1)]
        last_ticket_of = [0] * (nbuf + 1)
        gath = []
        for _ in range(nbuf + 1 if world > 1 else 0):
            gdev = "cpu" if gloo_rehearsal else device
            gath.append((torch.empty((world * _abi.SUMMARY_DIM, n), dtype=torch.float64, device=gdev),
                         torch.empty((world * n,), dtype=torch.int32, device=gdev)))
        pending = [None] * (nbuf + 1)
        tickets = []
        last_writer = {}     # output set -> (timed pass, shard) that wrote it last

        def wait_gather(k):
            if pending[k] is not None:
                for w in pending[k]:
                    w.wait()
                pending[k] = None

        def step(i, timed=False):
            k = nbuf if (timed and i == 0) else i % nbuf
            db_i = dbs[i % S]
            wait_gather(k)  # the gather that last read these output buffers
            s_k, t_k = outs[k]
            if depth > 0:
                if last_ticket_of[k]:
                    eng.wait(last_ticket_of[k])   # (device-side; a no-op in practice: that batch is 2 x depth passes back)
                eng.submit(db_i, flags=flags, summary=s_k, status=t_k)
                last_ticket_of[k] = eng.last_ticket
                if timed:
                    tickets.append(eng.last_ticket)
                    last_writer[k] = i % S
            else:
                eng.run(db_i, flags=flags, summary=s_k, status=t_k)
            if world > 1:
                # all-gather of the per-sample summaries (monte_carlo.py:76-83) on a side stream that waits
                # for THIS pass only: RCCL over xGMI overlaps the kernels of the following passes
                g_s, g_t = gath[k]
                with torch.cuda.stream(side):
                    if depth > 0:
                        eng.wait(eng.last_ticket, side)
                    else:
                        side.wait_stream(torch.cuda.current_stream(device))
                    if gloo_rehearsal:
                        side.synchronize()
                        s_k, t_k = s_k.cpu(), t_k.cpu()
                    pending[k] = [dist.all_gather_into_tensor(g_s, s_k, async_op=True),
                                  dist.all_gather_into_tensor(g_t, t_k, async_op=True)]

        def drain():
            if depth > 0:
                eng.wait()          # the current stream waits for every pass in flight
            for k in range(nbuf + 1):
                wait_gather(k)

        eng.set_profiling(True)
        for i in range(warmup):
            step(i)
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for i in range(steps):
            step(i, timed=True)
        drain()
        ev1.record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        gpu_ms = ev0.elapsed_time(ev1)
        eng.synchronize()     # raises if a lane hand-over of any pass timed out (its samples would carry ST_INCOMPLETE)
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if gloo_rehearsal else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        summary, status = outs[nbuf]     # the first timed pass: shard 0
        if world > 1:  # every rank must hold every rank's summaries: rank-major rows, own block == own results
            g_s, g_t = gath[nbuf]
            own = g_s[rank * _abi.SUMMARY_DIM:(rank + 1) * _abi.SUMMARY_DIM].to(summary.device)
            if not bool(((own == summary) | (own.isnan() & summary.isnan())).all()) or \
                    not torch.equal(g_t[rank * n:(rank + 1) * n].to(status.device), status):
                raise SystemExit("all-gather result does not contain this rank's summaries")
        rail_ms, flight_ms = eng.kernel_ms_history(steps)
        if tickets:      # device counters of every timed pass (the shards differ): mean per launch
            per_pass = [eng.ticket_stats(t) for t in tickets[-250:]]   # (the library keeps the records of its last 256 tickets)
            phys_steps = float(np.mean([x[0] for x in per_pass]))
            wave_iters = float(np.mean([x[1] for x in per_pass]))
        else:
            phys_steps, wave_iters = eng.last_stats()
        phys_total = phys_steps
        if world > 1:
            tot = torch.tensor([phys_steps], dtype=torch.float64, device="cpu" if gloo_rehearsal else device)
            dist.all_reduce(tot)
            phys_total = float(tot.item())
        # results of the timed passes that are still in their output sets (a set is reused nbuf passes later): shard -> (summary, status)
        kept = {j: outs[k] for k, j in last_writer.items()}
        return {"steps": steps, "precision": precision, "prec": prec, "db": db, "elapsed": elapsed, "gpu_ms": gpu_ms, "depth": depth, "shards": S,
                "kept": kept,
                "summary": summary, "status": status, "rail_ms": rail_ms, "flight_ms": flight_ms,
                "phys_steps": phys_steps, "wave_iters": wave_iters, "phys_total": phys_total}

    def leg_json(L):
        """The value / roofline part of the JSON line for one leg (rank 0)."""
        precision = L["precision"]
        total_traj = n * world
        value = total_traj * L["steps"] / L["elapsed"]
        fl, rl = float(np.mean(L["flight_ms"])), float(np.mean(L["rail_ms"]))
        per_launch_ms = L["gpu_ms"] / L["steps"]     # GPU time per launch over the timed region (HIP events)
        peak = PEAK_TFLOPS[precision]
        achieved_tf = L["phys_steps"] * FLOPS_PER_STEP / (per_launch_ms * 1e-3) / 1e12
        dispatch_tf = L["phys_steps"] * FLOPS_PER_STEP / (fl * 1e-3) / 1e12
        es = 4 if L["prec"] == _abi.PREC_F32 else 8
        algo_bytes = (L["db"].input_bytes() + 2 * n * (14 * es + 8 + 4) + n * (_abi.SUMMARY_DIM * 8 + 4))
        if precision == "f64_fast":      # + the hand-over record of (nearly) every diverging sample, written once and read once
            algo_bytes += 2 * n * (20 * 8 + 5 * 8 + 6 * 4)
        hbm_gbps = algo_bytes / (per_launch_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
        # (FETCH_SIZE / WRITE_SIZE need the profiler, they cannot be read from inside this process)
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "r4_hbm_traffic.json")
        if os.path.exists(tpath) and args.workload == "set_s" and n == 131072:
            tj = json.load(open(tpath)).get(precision)
            if tj:
                traffic, traffic_src = tj.get("traffic_bytes_per_launch"), "profiles/r4_hbm_traffic.json (rocprofv3 --pmc passes of this command)"
        st = L["status"].cpu().numpy()
        steps_col = L["summary"][_abi.SUM_STEPS].cpu().numpy()
        return {
            "value": value, "unit": "trajectories/s", "ms_per_step": L["elapsed"] / L["steps"] * 1e3, "dtype": precision,
            "trajectory_steps_per_s": L["phys_total"] * L["steps"] / L["elapsed"],
            "steps_per_trajectory": {"mean": float(steps_col.mean()), "max": float(steps_col.max()),
                                     "physics_mean": L["phys_steps"] / n},
            "lane_utilisation": L["phys_steps"] / (64.0 * L["wave_iters"]) if L["wave_iters"] else None,
            "end_reasons": {k: int(np.sum((st & 0xFF) == v)) for k, v in
                            (("max_time", 0), ("ground", 1), ("altitude_100km", 2), ("coast", 3), ("apogee", 4))},
            "nan_fraction": float(np.mean((st & _abi.ST_NAN) != 0)),
            "kernel_ms": {"gpu_ms_per_launch": per_launch_ms, "erpl_flight_dispatch_mean": fl, "erpl_rail_dispatch_mean": rl,
                          "launches_in_flight": max(L["depth"], 1),
                          "per_dispatch_flight_ms": [round(x, 3) for x in L["flight_ms"]]},
            "roofline": {
                "bound": "valu", "achieved": achieved_tf, "peak": peak, "unit": "TFLOP/s", "frac": achieved_tf / peak,
                "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "erpl_flight_" + {"f32": "f32", "f64": "f64", "f64_fast": "f64f"}[precision],
                "algorithmic_flops_per_step": FLOPS_PER_STEP, "rk4_steps_per_launch": L["phys_steps"],
                "launch_duration_ms": per_launch_ms,
                "timeline": "profiles/r4_bench_timeline.json (rocprofv3 --kernel-trace of this command: union of the erpl_flight dispatch intervals / passes)",
                "per_dispatch": {"duration_ms": fl, "achieved": dispatch_tf, "frac": dispatch_tf / peak,
                                 "note": "what rocprofv3 --kernel-trace reports per dispatch; dispatches overlap"},
                "note": "non-MFMA vector-ALU bound (SURVEY 8d); peak = vector rate of the dtype at the 2.4 GHz peak engine clock",
                "sustained_peak_measured": ({"value": 67.9, "unit": "TFLOP/s", "clock_ghz": 2.19,
                                             "source": "profiles/r4_ubench_clock.txt: a pure v_fma_f64 stream, 4 waves per SIMD; the chip clocks "
                                                       "at 2.19-2.27 GHz under fp64 load", "frac_of_it": achieved_tf / 67.9}
                                            if precision != "f32" else None),
                "hbm": {"achieved": hbm_gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": hbm_gbps / PEAK_HBM_GBPS, "algorithmic_bytes_per_launch": algo_bytes},
            },
        }

    main_leg = timed_leg(args.precision)
    second = None
    second_name = {"f64_fast": "f32", "f32": "f64_fast", "f64": "f64_fast"}[args.precision]
    if not args.no_second_leg:
        second = timed_leg(second_name)

    out = None
    if rank == 0:
        total_traj = n * world
        out = {"metric": "Monte Carlo trajectories/sec (whole node) + apogee-match rate"}
        body = leg_json(main_leg)
        out.update({"value": body.pop("value"), "unit": body.pop("unit"), "n_gpus": world, "steps": args.steps,
                    "warmup": args.warmup, "ms_per_step": body.pop("ms_per_step"), "higher_is_better": True,
                    "scaling": scaling, "vs_baseline": None, "dtype": body.pop("dtype"), "data": "synthetic",
                    "config": {
                        "workload": f"{args.workload}: {n} dispersed 6-DOF samples/GPU ({total_traj} total), "
                                    f"{args.motor} motor, reference dispersion model, "
                                    f"{'CSV base wind K=6' if csv else 'synthetic wind K=100'}, "
                                    f"rail dt=0.01 + RK4 dt=0.005, "
                                    f"{'to first-descent apogee' if flags else 'full reference termination logic'}",
                        "samples_per_gpu": n, "precision": args.precision, "passes_in_flight": max(depth, 1),
                        "distinct_shards": main_leg["shards"],
                        "shards_note": "every timed pass integrates a DIFFERENT resident shard (seed 1234 + rank + 7919 x pass; --shards 1 replays "
                                       "one); the parity figures and end_reasons are those of the first timed pass (shard 0); counters are means over the passes", "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                        "lane_adoption": "library default" if args.adopt < 0 else args.adopt,
                        "parallelism": f"sample-shard x{world}" + (" + RCCL all-gather of [16,n] summaries overlapped with the following passes" if world > 1 else ""),
                    }})
        out.update(body)
        if second is not None:
            out[second_name] = leg_json(second)
            if second_name == "f32":
                out["f32"]["note"] = ("BASELINE configs[2] names fp32; it is the secondary leg because it does not reproduce the "
                                      "reference's apogee_altitude on diverging samples (see its apogee_match_rate; DESIGN.md section 5)")

    # ---------------- parity + CPU baseline (N = 1 only, rank 0, outside the timed region) ----------------
    if rank == 0 and world > 1:
        out["cpu_baseline"] = None   # measured by the N = 1 run only (the host cores are shared by the ranks)
    if rank == 0 and world == 1 and not args.no_parity:
        # the fp64 reference-order gate kernel on the SAME shard: the per-sample reference for every timed build, and a
        # timed leg of its own on the same footing as the others (erpl_mc_submit_batch at the library depth; fewer
        # passes: one lasts ten times longer)
        gate_leg = main_leg if args.precision == "f64" else timed_leg("f64", steps=max(2, min(args.steps, 8)), warmup=min(args.warmup, 2))
        gj = leg_json(gate_leg)
        out["f64_gate"] = {"note": "the fp64 reference-order kernel (ERPL_PREC_F64; MonteCarloAnalyzer.precision = 'f64'): "
                                   f"{gate_leg['steps']} passes over the same shard through erpl_mc_submit_batch, like the other legs",
                           "value": gj["value"], "unit": gj["unit"], "ms_per_step": gj["ms_per_step"], "steps": gate_leg["steps"],
                           "lane_utilisation": gj["lane_utilisation"], "kernel_ms": gj["kernel_ms"],
                           "roofline": {k: gj["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel",
                                                                       "rk4_steps_per_launch", "launch_duration_ms")}}
        gs, gt = gate_leg["summary"], gate_leg["status"]
        gs, gt = gs.cpu().numpy(), gt.cpu().numpy()
        rep = match_report(gs, gt, main_leg["summary"].cpu().numpy(), main_leg["status"].cpu().numpy())
        out["apogee_match_rate"] = rep["apogee_match_rate_0p1pct"]
        out["parity"] = {"timed_shard_vs_fp64_gate_kernel": rep}
        # the other shards both legs still hold the results of (the gate leg runs fewer passes): the same comparison, pooled
        both = sorted(j for j in set(main_leg["kept"]) & set(gate_leg["kept"]) if j != 0) if gate_leg is not main_leg else []
        if both:
            cat = lambda leg, c: np.concatenate([leg["kept"][j][c].cpu().numpy() for j in both], axis=-1)
            more = match_report(cat(gate_leg, 0), cat(gate_leg, 1), cat(main_leg, 0), cat(main_leg, 1))
            more["shards"] = both
            out["parity"]["other_timed_shards_vs_fp64_gate_kernel"] = more
        if second is not None:
            rep2 = match_report(gs, gt, second["summary"].cpu().numpy(), second["status"].cpu().numpy())
            out[second_name]["apogee_match_rate"] = rep2["apogee_match_rate_0p1pct"]
            out[second_name]["parity"] = {"timed_shard_vs_fp64_gate_kernel": rep2}
    if rank == 0 and world == 1 and (args.cpu_seconds > 0 or not args.no_parity):
        from oracle import oracle as orc
        cores = host_cores()
        if args.cpu_seconds > 0:
            m = min(n, 256 * cores)
            cpu_t, osum, ostat = 0.0, None, None
            for _attempt in range(3):  # grow the sample until it is a 10-30 s measurement
                hb = host_slice(db64, m)
                t1 = time.perf_counter()
                osum, ostat = orc.run_batch(cfg, hb, flags=flags, threads=cores)
                cpu_t = time.perf_counter() - t1
                if cpu_t >= 0.6 * args.cpu_seconds or m >= n:
                    break
                m = int(min(n, max(m + 1, m * args.cpu_seconds / max(cpu_t, 1e-3))))
            # RK4 steps with physics in them (the GPU's unit): the oracle brute-forces the ~57 k no-op steps
            # of every non-finite trajectory, the kernels fast-forward them; count the same m samples' physics
            # steps with the gate kernel (identical trajectories)
            sub = DeviceBatch.from_host(hb, device, _abi.PREC_F64)
            eng.run(sub, flags=flags)
            phys_m, _ = eng.last_stats()
            out["cpu_baseline"] = {
                "value": m / cpu_t, "unit": "trajectories/s", "cores": cores, "kind": "port",
                "sample": f"first {m} samples of rank 0's shard, CPU oracle (C fp64, OpenMP, {cores} threads), {cpu_t:.1f} s",
                "all_loop_steps_per_s": float(osum[_abi.SUM_STEPS].sum() / cpu_t),
                "physics_steps_per_s": float(phys_m / cpu_t),
                "note": "physics_steps_per_s is the unit of the GPU's trajectory_steps_per_s; all_loop_steps_per_s also counts the no-op steps of non-finite trajectories the oracle runs to max_time",
            }
            if not args.no_parity:
                for L, dst in ((main_leg, out), (second, out.get(second_name))):
                    if L is None:
                        continue
                    got_s, got_t = L["summary"][:, :m].cpu().numpy(), L["status"][:m].cpu().numpy()
                    dst["parity"]["timed_shard_sample_vs_cpu_oracle"] = match_report(osum, ostat, got_s, got_t)
                # the reference of the whole-shard rates is itself measured against the oracle at this scale
                out["parity"]["fp64_gate_kernel_sample_vs_cpu_oracle"] = match_report(osum, ostat, gs[:, :m], gt[:m])
        if not args.no_parity:
            # BASELINE configs[1]: 1 k reference-faithful samples (seed=i stream, CSV wind) against the CPU oracle
            pl = flatten.generate_parameter_samples(sampling.DEFAULT_UNCERTAINTY, 1000)
            hbr = flatten.dispersed_batch(rocket, models.LiquidMotor(), wm, EXAMPLE_IC, pl, CSV_ALT, CSV_WIND)
            cfg_l = flatten.config_from_objects(rocket, models.LiquidMotor(), atm)
            osum, ostat = orc.run_batch(cfg_l, hbr, threads=cores)
            eng.set_config(cfg_l)
            res = {}
            for name, p in _abi.PRECISIONS.items():
                dbr = DeviceBatch.from_host(hbr, device, p)
                s2, st2 = eng.run(dbr)
                torch.cuda.synchronize()
                r = match_report(osum, ostat, s2.cpu().numpy(), st2.cpu().numpy())
                r.pop("by_reference_class")
                res[name] = r
            out["parity"]["cfg2_set_r_1k_vs_cpu_oracle"] = res
            eng.set_config(cfg)
    if rank == 0 and world == 1 and not args.no_cfg5 and args.workload == "set_s" and args.samples_per_gpu == 131072:
        out["cfg5_share"] = cfg5_share(eng, device, rocket, atm, wm, args)
        eng.set_config(cfg)
    if rank == 0 and world == 1 and not args.no_api and args.workload == "set_s":
        out["api_end_to_end"] = api_end_to_end(device, rocket, motor, atm, wm)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
