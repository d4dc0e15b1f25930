#!/usr/bin/env python3
"""Headline benchmark: Monte Carlo trajectories/sec + apogee-match rate (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (rail kernel + RK4 flight kernel [+ RCCL all-gather of the
per-sample summaries when N > 1]) over one batch of synthetic dispersions ALREADY RESIDENT in HBM.
Workload at N = 1: the fp32 throughput configuration (BASELINE configs[2]/[3]): 131 072 dispersed
samples per GPU (x 8 GPUs = the 1 048 576 samples of configs[3]), LiquidMotor, reference
dispersion model (monte_carlo.py:156-179), synthetic 100-knot wind profile, full reference
termination logic.  Weak scaling: every rank integrates its own 131 072-sample shard.

Prints ONE JSON line (rank 0) with `roofline` and `cpu_baseline` objects:
  roofline      dominant kernel = erpl_flight_f32; the path is vector-ALU bound, so
                `achieved` = RK4 steps integrated per launch x 1570 algorithmic flops / the kernel's
                HIP-event duration, against the 157.3 TFLOP/s fp32 vector peak; the HBM view that
                north_star asks for is reported alongside (`hbm`).
  cpu_baseline  the CPU oracle (C fp64 restatement, OpenMP over samples, all host cores) on a
                bounded sample of the same shard; it is a reported baseline, not the target.
  parity        apogee-match rates of the GPU results against that oracle (fp32 shard sample)
                and of the fp64 gate on 1 k reference-faithful samples (BASELINE configs[1]).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from erpl_monte_carlo_sim_amd import _abi, flatten, models, sampling  # noqa: E402
from erpl_monte_carlo_sim_amd.engine import DeviceBatch, TrajectoryEngine  # noqa: E402

FLOPS_PER_STEP = 1570.0        # SURVEY.md §8d algorithmic count (4 RHS x 343 + ~200)
PEAK_FP32_VECTOR_TFLOPS = 157.3  # MI355X_MICROARCH.md chip table
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
    with np.errstate(invalid="ignore", divide="ignore"):
        e = np.abs(a - b) / np.abs(b)
    return np.where(same, 0.0, e)


def host_slice(db, m):
    """First m samples of a DeviceBatch as a flatten.HostBatch (fp64) for the oracle."""
    hb = flatten.HostBatch(m, db.k_wind)
    hb.ic = np.ascontiguousarray(db.ic[:, :m].cpu().numpy())
    hb.rocket = np.ascontiguousarray(db.rocket[:, :m].cpu().numpy())
    hb.motor = np.ascontiguousarray(db.motor[:, :m].cpu().numpy())
    hb.alt_grid = db.alt_grid.cpu().numpy().astype(np.float64)
    hb.wind = np.ascontiguousarray(db.wind[:, :, :m].double().cpu().numpy())
    return hb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--samples-per-gpu", type=int, default=131072)
    ap.add_argument("--workload", default="set_s", choices=["set_s", "set_p_apogee", "set_p_full", "csv_chute"])
    ap.add_argument("--motor", default="liquid", choices=["liquid", "solid"])
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration (0 = skip)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--block", type=int, default=256)
    ap.add_argument("--max-blocks", type=int, default=0)
    ap.add_argument("--refill", type=int, default=1)
    ap.add_argument("--pipeline", type=int, default=2, help="depth of the extra pipelined measurement (1 = skip)")
    ap.add_argument("--waves", type=int, default=0, help="fp32 kernel build: 2 or 3 waves per SIMD (0 = library default, by batch size)")
    ap.add_argument("--chunk", type=int, default=-1, help="steps per launch between compactions (-1 = library default)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)   # one rank per GPU on the driver's 8-GPU node
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        # "nccl" is RCCL on ROCm.  ERPL_BENCH_BACKEND=gloo exists only to rehearse the N > 1 code path on
        # a box with a single GPU (RCCL refuses two ranks on one device); it is never used for numbers.
        backend = os.environ.get("ERPL_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    rocket, atm, wm = models.Rocket(), models.StandardAtmosphere(), models.WindModel()
    motor = models.SolidMotor() if args.motor == "solid" else models.LiquidMotor()
    cfg = flatten.config_from_objects(rocket, motor, atm)
    prec = _abi.PREC_F32 if args.precision == "f32" else _abi.PREC_F64
    eng = TrajectoryEngine(device)
    eng.set_config(cfg)
    eng.set_launch(args.block, args.max_blocks, args.refill)
    if args.chunk >= 0:
        eng.set_chunk(args.chunk)
    eng.set_waves_per_simd(args.waves)
    n = args.samples_per_gpu
    planar = args.workload.startswith("set_p")
    flags = _abi.FLAG_STOP_AT_APOGEE if args.workload == "set_p_apogee" else 0
    csv = args.workload == "csv_chute"
    db = sampling.synthetic_dispersions(
        n, rocket, motor, wm, EXAMPLE_IC, device, precision=prec, seed=1234 + rank, planar=planar or csv,
        base_altitude_profile=CSV_ALT if csv else None, base_wind_profile=CSV_WIND if csv else None)
    eng.reserve(n)
    gloo_rehearsal = world > 1 and os.environ.get("ERPL_BENCH_BACKEND", "nccl") != "nccl"
    # N > 1: the all-gather of pass i (RCCL over xGMI, on RCCL's own stream) overlaps the kernels of pass
    # i+1, so outputs and gather buffers are double-buffered; every gather issued inside the timed region
    # is waited for before the closing barrier.
    nbuf = 2 if world > 1 else 1
    outs = [eng.alloc_outputs(n) for _ in range(nbuf)]
    gath = []
    for _ in range(nbuf if world > 1 else 0):
        gdev = "cpu" if gloo_rehearsal else device
        gath.append((torch.empty((world * _abi.SUMMARY_DIM, n), dtype=torch.float64, device=gdev),
                     torch.empty((world * n,), dtype=torch.int32, device=gdev)))
    pending = [None] * nbuf

    def wait_gather(k):
        if pending[k] is not None:
            for w in pending[k]:
                w.wait()
            pending[k] = None

    def step(i):
        k = i % nbuf
        wait_gather(k)  # the gather that last read these output buffers
        s_k, t_k = outs[k]
        eng.run(db, flags=flags, summary=s_k, status=t_k)
        if world > 1:  # all-gather of the per-sample summaries (monte_carlo.py:76-83)
            g_s, g_t = gath[k]
            if gloo_rehearsal:
                torch.cuda.synchronize()
                s_k, t_k = s_k.cpu(), t_k.cpu()
            pending[k] = [dist.all_gather_into_tensor(g_s, s_k, async_op=True),
                          dist.all_gather_into_tensor(g_t, t_k, async_op=True)]

    def drain():
        for k in range(nbuf):
            wait_gather(k)

    eng.set_profiling(True)
    for i in range(args.warmup):
        step(i)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if gloo_rehearsal else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    summary, status = outs[(args.steps - 1) % nbuf]
    if world > 1:  # every rank must hold every rank's summaries: rank-major rows, own block == own results
        g_s, g_t = gath[(args.steps - 1) % nbuf]
        own = g_s[rank * _abi.SUMMARY_DIM:(rank + 1) * _abi.SUMMARY_DIM].to(summary.device)
        if not bool(((own == summary) | (own.isnan() & summary.isnan())).all()) or \
                not torch.equal(g_t[rank * n:(rank + 1) * n].to(status.device), status):
            raise SystemExit("all-gather result does not contain this rank's summaries")

    rail_ms, flight_ms = eng.kernel_ms_history(args.steps)
    phys_steps, wave_iters = eng.last_stats()
    st = status.cpu().numpy()
    sm = summary.cpu().numpy()
    steps_col = sm[_abi.SUM_STEPS]
    # ---- extra (N = 1 only, never `value`): the same K passes software-pipelined two deep on two
    # streams / two contexts / two output buffers.  A single pass is bound by the sequential latency of
    # its longest trajectory; independent passes overlap each other's sparse tails. ----
    pipelined = None
    if world == 1 and args.pipeline > 1:
        engs = [eng] + [TrajectoryEngine(device) for _ in range(args.pipeline - 1)]
        pouts = [(summary, status)] + [eng.alloc_outputs(n) for _ in range(args.pipeline - 1)]
        streams = [torch.cuda.Stream(device) for _ in range(args.pipeline)]
        for e in engs[1:]:
            e.set_config(cfg); e.set_launch(args.block, args.max_blocks, args.refill); e.reserve(n)
            if args.chunk >= 0:
                e.set_chunk(args.chunk)
        for k in range(args.pipeline):       # warm-up of the extra contexts
            with torch.cuda.stream(streams[k]):
                engs[k].run(db, flags=flags, summary=pouts[k][0], status=pouts[k][1], stream=streams[k])
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for i in range(args.steps):
            k = i % args.pipeline
            engs[k].run(db, flags=flags, summary=pouts[k][0], status=pouts[k][1], stream=streams[k])
        torch.cuda.synchronize()
        el_p = time.perf_counter() - tp
        pipelined = {"depth": args.pipeline, "value": n * args.steps / el_p, "unit": "trajectories/s",
                     "ms_per_step": el_p / args.steps * 1e3,
                     "note": "independent passes overlapped on separate streams/contexts; not the headline"}
        for e in engs[1:]:
            e.close()

    if world > 1:
        tot = torch.tensor([phys_steps], dtype=torch.float64, device="cpu" if gloo_rehearsal else device)
        dist.all_reduce(tot)
        phys_total = float(tot.item())
    else:
        phys_total = phys_steps

    out = None
    if rank == 0:
        total_traj = n * world
        value = total_traj * args.steps / elapsed
        fl = float(np.mean(flight_ms))
        rl = float(np.mean(rail_ms))
        achieved_tf = phys_steps * FLOPS_PER_STEP / (fl * 1e-3) / 1e12
        es = 4 if prec == _abi.PREC_F32 else 8
        algo_bytes = (db.input_bytes() + 2 * n * (14 * es + 8 + 4) + n * (_abi.SUMMARY_DIM * 8 + 4))
        hbm_gbps = algo_bytes / ((fl + rl) * 1e-3) / 1e9
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
        # (FETCH_SIZE / WRITE_SIZE need the profiler, they cannot be read from inside this process)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r1_hbm_traffic.json")
        if os.path.exists(tpath) and args.workload == "set_s" and n == 131072 and args.precision == "f32":
            traffic = json.load(open(tpath)).get("traffic_bytes_per_launch")
        out = {
            "metric": "Monte Carlo trajectories/sec (whole node) + apogee-match rate",
            "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {n} dispersed 6-DOF samples/GPU ({total_traj} total), "
                            f"{args.motor} motor, reference dispersion model, "
                            f"{'CSV base wind K=6' if csv else 'synthetic wind K=100'}, "
                            f"rail dt=0.01 + RK4 dt=0.005, "
                            f"{'to first-descent apogee' if flags else 'full reference termination logic'}",
                "samples_per_gpu": n, "precision": args.precision,
                "parallelism": f"sample-shard x{world}" + (" + RCCL all-gather of [16,n] summaries overlapped with the next pass" if world > 1 else ""),
            },
            "trajectory_steps_per_s": phys_total * args.steps / elapsed,
            "steps_per_trajectory": {"mean": float(steps_col.mean()), "max": float(steps_col.max()),
                                     "physics_mean": phys_steps / n},
            "lane_utilisation": phys_steps / (64.0 * wave_iters) if wave_iters else None,
            "pipelined": pipelined,
            "end_reasons": {k: int(np.sum((st & 0xFF) == v)) for k, v in
                            (("max_time", 0), ("ground", 1), ("altitude_100km", 2), ("coast", 3), ("apogee", 4))},
            "nan_fraction": float(np.mean((st & _abi.ST_NAN) != 0)),
            "kernel_ms": {"erpl_flight": fl, "erpl_rail": rl, "per_launch_flight_ms": [round(x, 3) for x in flight_ms]},
            "roofline": {
                "bound": "valu", "achieved": achieved_tf, "peak": PEAK_FP32_VECTOR_TFLOPS if prec == _abi.PREC_F32 else PEAK_FP32_VECTOR_TFLOPS / 2,
                "unit": "TFLOP/s",
                "frac": achieved_tf / (PEAK_FP32_VECTOR_TFLOPS if prec == _abi.PREC_F32 else PEAK_FP32_VECTOR_TFLOPS / 2),
                "traffic": traffic,
                "kernel": "erpl_flight_" + args.precision,
                "algorithmic_flops_per_step": FLOPS_PER_STEP, "rk4_steps_per_launch": phys_steps,
                "note": "non-MFMA vector-ALU bound (SURVEY 8d); peak = fp32 vector (= f32 MFMA) rate",
                "hbm": {"achieved": hbm_gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": hbm_gbps / PEAK_HBM_GBPS, "algorithmic_bytes_per_launch": algo_bytes},
            },
        }

    # ---------------- CPU baseline + parity (N = 1 only, rank 0, outside the timed region) ----------------
    if rank == 0 and world > 1:
        out["cpu_baseline"] = None   # measured by the N = 1 run only (the host cores are shared by the ranks)
    if rank == 0 and world == 1 and (args.cpu_seconds > 0 or not args.no_parity):
        from oracle import oracle as orc
        cores = host_cores()
        if args.cpu_seconds > 0:
            m = min(n, 256 * cores)
            cpu_t, osum, ostat = 0.0, None, None
            for _attempt in range(3):  # grow the sample until it is a 10-30 s measurement
                hb = host_slice(db, m)
                t1 = time.perf_counter()
                osum, ostat = orc.run_batch(cfg, hb, flags=flags, threads=cores)
                cpu_t = time.perf_counter() - t1
                if cpu_t >= 0.6 * args.cpu_seconds or m >= n:
                    break
                m = int(min(n, max(m + 1, m * args.cpu_seconds / max(cpu_t, 1e-3))))
            out["cpu_baseline"] = {
                "value": m / cpu_t, "unit": "trajectories/s", "cores": cores, "kind": "port",
                "sample": f"first {m} samples of rank 0's shard, CPU oracle (C fp64, OpenMP, {cores} threads), {cpu_t:.1f} s",
                "trajectory_steps_per_s": float(osum[_abi.SUM_STEPS].sum() / cpu_t),
            }
            got = sm[:, :m]
            e_fa = relerr(got[_abi.SUM_FIRST_APOGEE_ALT], osum[_abi.SUM_FIRST_APOGEE_ALT])
            e_ap = relerr(got[_abi.SUM_APOGEE_ALT], osum[_abi.SUM_APOGEE_ALT])
            out["parity"] = {
                "shard_sample": {"n": m, "precision": args.precision,
                                 "first_apogee_match_rate_0p1pct": float(np.mean(e_fa <= 1e-3)),
                                 "apogee_argmax_match_rate_0p1pct": float(np.mean(e_ap <= 1e-3)),
                                 "same_end_reason": float(np.mean((st[:m] & 0xFF) == (ostat & 0xFF)))}}
        if not args.no_parity:
            # BASELINE configs[1]: 1 k reference-faithful samples (seed=i stream, CSV wind), fp64 gate
            pl = flatten.generate_parameter_samples(sampling.DEFAULT_UNCERTAINTY, 1000)
            hbr = flatten.dispersed_batch(rocket, models.LiquidMotor(), wm, EXAMPLE_IC, pl, CSV_ALT, CSV_WIND)
            cfg_l = flatten.config_from_objects(rocket, models.LiquidMotor(), atm)
            osum, ostat = orc.run_batch(cfg_l, hbr, threads=cores)
            eng.set_config(cfg_l)
            res = {}
            for name, p in (("f64", _abi.PREC_F64), ("f32", _abi.PREC_F32)):
                dbr = DeviceBatch.from_host(hbr, device, p)
                s2, st2 = eng.run(dbr)
                torch.cuda.synchronize()
                s2 = s2.cpu().numpy()
                res[name] = {
                    "apogee_match_rate_0p1pct": float(np.mean(relerr(s2[_abi.SUM_APOGEE_ALT], osum[_abi.SUM_APOGEE_ALT]) <= 1e-3)),
                    "first_apogee_match_rate_0p1pct": float(np.mean(relerr(s2[_abi.SUM_FIRST_APOGEE_ALT], osum[_abi.SUM_FIRST_APOGEE_ALT]) <= 1e-3)),
                }
            out.setdefault("parity", {})["cfg2_set_r_1k"] = res
            out["apogee_match_rate"] = res["f64"]["apogee_match_rate_0p1pct"]
            eng.set_config(cfg)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
