"""MonteCarloAnalyzer — drop-in for monte_carlo.py:17-473 of the reference.

`run_monte_carlo(initial_conditions, n_samples, n_processes=None, optimized=False)` keeps its
signature and the structure of the returned analysis dict.  The ProcessPoolExecutor fan-out of the
reference (one future per sample, results pickled back through pipes) becomes: flatten all samples
into SoA tensors -> one `erpl_mc_run_batch` per GPU on this rank's shard -> one all-gather of the
[16, n] summaries (RCCL when torch.distributed is initialised with "nccl").  Dispersion draws,
motor perturbation and wind synthesis stay on the host and are bit-identical to the reference
(flatten.py); the integration itself runs only on the GPU.
"""
import os
import time

import numpy as np
import torch

from . import _abi, analysis, dist, flatten, reports
from . import results as results_mod
from .engine import DeviceBatch, TrajectoryEngine
from .sampling import DEFAULT_UNCERTAINTY
from .simulator import shared_engine

_END = results_mod.END_NAMES


class _Shard:
    """Stand-in for a HostBatch in dist.run_local_shard: only its size is looked at (the shard is built chunk by
    chunk inside the runner)."""

    def __init__(self, n):
        self.n = n


class MonteCarloAnalyzer:
    def __init__(self, rocket, motor, atmosphere, wind_model, device=None, verbose=True):
        self.rocket, self.motor, self.atmosphere, self.wind_model = rocket, motor, atmosphere, wind_model
        self.n_cores = os.cpu_count()
        self.base_altitude_profile = None    # monte_carlo.py:27-32
        self.base_wind_profile = None
        self.uncertainty_params = {k: (list(v) if isinstance(v, list) else v) for k, v in DEFAULT_UNCERTAINTY.items()}
        self.device = device
        self.verbose = verbose
        # Kernel build: "f64_fast" (default since round 4) = the fp64 throughput build - the reference's outcome on every
        # sample of the parity sets (apogee, end reason, step count: 131 072 / 131 072 against the reference-order
        # kernel, 60 000 / 60 000 of those against the CPU oracle), healthy flights to 1e-9, 10 x the rate;
        # "f64" = the reference-order gate kernel (the reference's arithmetic op for op); "f32" = first-descent
        # apogee / healthy flights only
        self.precision = "f64_fast"
        self.n_trajectories = 50             # samples that carry a 'trajectory' (plots use the first 50)
        self.trajectory_stride = 20
        # simulator attributes a user could have changed on FlightSimulator
        self.max_time, self.dt_initial, self.pitch_damping, self.yaw_damping = 300.0, 0.01, 20.0, 20.0
        if verbose:
            print(f"Initialized Monte Carlo analyzer with {self.n_cores} cores")

    # -- parameter streams (bit-identical to the reference) -------------------------------------
    def _generate_parameter_samples(self, n_samples):
        return flatten.generate_parameter_samples(self.uncertainty_params, n_samples, stream="seed_i")

    def _generate_parameter_samples_vectorized(self, n_samples):
        return flatten.generate_parameter_samples(self.uncertainty_params, n_samples, stream="seed_42")

    # -- the hot path ------------------------------------------------------------------------------
    def _config(self):
        return flatten.config_from_objects(self.rocket, self.motor, self.atmosphere, dt_initial=self.dt_initial,
                                           max_time=self.max_time, pitch_damping=self.pitch_damping,
                                           yaw_damping=self.yaw_damping)

    CHUNK = 131072   # run_monte_carlo: samples per submitted batch - one batch fills an MI355X (256 CUs x 4 SIMDs x
    #                  2 waves x 64 lanes) and the host prepares the next one meanwhile
    DEVICE_CHUNK = 1 << 20   # run_monte_carlo_device: samples per sub-batch.  Nothing is prepared on the host there, and a
    #                  larger batch keeps the lanes busy from its own queue (5.8 M trajectories/s in one 1 M-sample launch
    #                  against 4.9 M for eight 131 072-sample batches, whose tails all fall at the end of the run)

    def _integrate_shard(self, initial_conditions, params, lo, hi, n_traj_global):
        """Integrate samples [lo, hi) of `params` (dict of arrays or list of dicts) on this rank's GPU; returns
        (summary [16, hi-lo] device tensor, status [hi-lo] device tensor, captured trajectories or None).

        The samples that carry a trajectory (global index < n_traj_global: the first ones of the shard) go through
        the trajectory-capture build in a small batch of their own; everything else runs in chunks of CHUNK samples
        handed to erpl_mc_submit_batch, so that the host builds chunk i+1 (MT19937 streams and AR(1) wind tables in
        C threads, the rest as NumPy expressions) while the GPU integrates chunk i.  Samples are independent: the
        split does not change a single bit of the summaries (tested)."""
        eng = shared_engine(self.device)
        eng.set_config(self._config())
        prec = _abi.PRECISIONS[self.precision]
        take = (lambda a, b: {k: v[a:b] for k, v in params.items()}) if isinstance(params, dict) else (lambda a, b: params[a:b])

        def host_batch(a, b, threads=0):
            return flatten.dispersed_batch(self.rocket, self.motor, self.wind_model, initial_conditions, take(a, b),
                                           self.base_altitude_profile, self.base_wind_profile, threads=threads)
        parts, traj, capture_inputs = [], None, None
        m = max(0, min(hi, n_traj_global) - lo)   # local samples 0..m-1 are captured
        if m > 0:
            db = DeviceBatch.from_host(host_batch(lo, lo + m), eng.device, prec)
            dt = min(self.dt_initial, 0.005)
            cap = int(np.ceil(self.max_time / dt / self.trajectory_stride)) + 4
            ids = list(range(m))
            # on a stream of its own: the capture build follows a non-finite sample step by step to max_time (it has
            # records to write), 0.6 s for one wave - and erpl_mc_submit_batch starts a batch behind everything that is
            # on the caller's stream, so on the current stream it held up every chunk below (round 4: 1.5 -> 0.9 s at 10^6)
            cur = torch.cuda.current_stream(eng.device)
            capture_stream = torch.cuda.Stream(eng.device)
            capture_stream.wait_stream(cur)           # the upload above
            with torch.cuda.stream(capture_stream):   # (its output buffers are allocated and pre-filled on that stream too)
                summ, status, tr, tlen = eng.run(db, traj_ids=ids, traj_stride=self.trajectory_stride, traj_cap=cap,
                                                 flags=_abi.FLAG_CAPTURE_POSITION_ONLY)   # 'trajectory' = time, altitude, position
            for t_ in (summ, status, tr, tlen):
                t_.record_stream(cur)                 # consumed on the current stream, behind the wait_stream below
            # the inputs were allocated on the current stream and are read on the other one until the capture ends: they must
            # not go back to the allocator before (the chunk loop below would otherwise hand their memory to the next upload)
            capture_inputs = db
            for t_ in (db.ic, db.rocket, db.motor, db.alt_grid, db.wind):
                if t_ is not None:
                    t_.record_stream(capture_stream)
            traj = (ids, tr, tlen)
            parts.append((summ, status))
        inflight = []    # (ticket, inputs): the inputs must outlive their batch (it runs on the library's own streams)
        # look-ahead: `depth` batches run, one more is queued behind them while the host builds the next chunk.
        # erpl_mc_check_batch(T) waits for T's own event (batches behind it keep running) and answers for T itself.
        keep = eng.get_overlap() + 1
        incomplete = None

        def retire(ticket):
            nonlocal incomplete
            try:
                eng.check(ticket)   # host-blocking for that batch alone; raises if one of its lane hand-overs timed out
            except _abi.IncompleteBatch as e:
                # not raised here: in a multi-rank run the other ranks are on their way to the all-gather, and the
                # status words of the lost samples carry ERPL_ST_INCOMPLETE through it - run_batch_arrays raises on
                # EVERY rank from the gathered status
                incomplete = incomplete or e
        # Host preparation of the chunks runs in a few Python workers side by side (round 4): the C generators
        # (MT19937 streams, AR(1) wind tables) have their own threads, but the NumPy expressions between them are
        # single-threaded and were half of a chunk's preparation; NumPy and ctypes release the interpreter lock.
        # Chunks are consumed in order, so nothing about the results depends on it.  (131 072 samples with a
        # 100-knot wind table are 335 MB per prepared chunk, on the host and then on the device: at most workers + 1
        # wait to be submitted at a time.)
        import collections
        from concurrent.futures import ThreadPoolExecutor
        chunks = [(a, min(hi, a + self.CHUNK)) for a in range(lo + m, hi, self.CHUNK)]
        workers = max(1, min(len(chunks), flatten.host_workers()))
        def device_batch(a, b):
            # validation and the upload (335 MB of pageable memory per chunk: 70 ms at the 4.6 GB/s such a copy gets) in
            # the worker too: on the default stream of the engine's device, before the submission that follows it there
            with torch.cuda.device(eng.device):
                return DeviceBatch.from_host(host_batch(a, b, threads=max(1, flatten.host_cores() // workers)), eng.device, prec)
        with ThreadPoolExecutor(workers) as pool:
            todo = iter(chunks)
            ready = collections.deque()
            for _ in range(workers + 1):
                c = next(todo, None)
                if c is not None:
                    ready.append(pool.submit(device_batch, *c))
            while ready:
                db = ready.popleft().result()
                c = next(todo, None)
                if c is not None:
                    ready.append(pool.submit(device_batch, *c))
                parts.append(eng.submit(db))
                inflight.append((eng.last_ticket, db))
                while len(inflight) > keep:
                    retire(inflight.pop(0)[0])
        if inflight:
            eng.wait()
            retire(-1)
        if m > 0:
            torch.cuda.current_stream(eng.device).wait_stream(capture_stream)
            del capture_inputs
        if len(parts) == 1:
            return parts[0][0], parts[0][1], traj
        return torch.cat([p[0] for p in parts], dim=1), torch.cat([p[1] for p in parts]), traj

    def run_batch_arrays(self, initial_conditions, parameter_samples):
        """Integrate the given dispersed samples; returns (summary [16, n], status [n]) NumPy arrays
        (identical on every rank) and this rank's captured trajectories.  Each rank builds only its own
        shard [lo, hi) of the samples on the host (the per-sample seeds make them independent,
        monte_carlo.py:156-179)."""
        n = len(parameter_samples["random_seed"]) if isinstance(parameter_samples, dict) else len(parameter_samples)
        rank, ws = dist.world()
        lo, hi, _ = dist.shard_bounds(n, rank, ws)
        box = {}

        def runner(_unused):
            summ, status, box["traj"] = self._integrate_shard(initial_conditions, parameter_samples, lo, hi,
                                                              min(self.n_trajectories, n))
            return summ, status
        local = _Shard(hi - lo) if hi > lo else None
        summ, status = dist.run_local_shard(n, local, runner)
        TrajectoryEngine.raise_if_incomplete(status)
        return summ, status, box.get("traj"), lo

    def _sample_table(self, summ, status, parameter_samples, traj, lo):
        """Columns of the run as a results.SampleTable (per-sample dicts are built on access)."""
        trajectories = {}
        if traj is not None:
            ids, tr, tlen = traj
            tr, tlen = tr.cpu().numpy(), tlen.cpu().numpy()
            for m, i in enumerate(ids):
                k = int(tlen[m])
                rec = tr[m, :k]
                trajectories[lo + i] = {           # monte_carlo.py:298-302
                    "time": rec[:, 0] - summ[_abi.SUM_RAIL_EXIT_TIME, lo + i],
                    "altitude": rec[:, 3].copy(),
                    "position": rec[:, 1:4].copy(),
                }
        return results_mod.SampleTable(summ, status, parameter_samples, trajectories)

    def _result_dicts(self, summ, status, parameter_samples, traj, lo):
        """The eager list of per-sample dicts (what round 2 returned; kept as the cross-check of LazyResults)."""
        table = self._sample_table(summ, status, parameter_samples, traj, lo)
        return [table.record(i) for i in range(table.n)]

    def run_monte_carlo(self, initial_conditions, n_samples=1000, n_processes=None, optimized=False):
        """monte_carlo.py:52-90 (and :92-154 when optimized=True: seed-42 stream + 'performance').

        Same inputs (bit for bit), same analysis dict.  `analysis['results']` / `['outliers']` are lazy sequences
        (results.LazyResults): `len`, indexing, slicing, iteration and `+` behave like the reference's lists and
        yield the same per-sample dicts, built on access; the outlier filter and the statistics run on the summary
        columns.  `self.precision` picks the kernel build: "f64_fast" (default) is the fp64 throughput build - the
        reference's outcome on every sample of the parity sets at 4-5 M trajectories/s (its blow-ups finish in the
        reference-order kernel); "f64" runs everything in the reference-order gate kernel
        (1.2 M trajectories/s)."""
        if self.verbose:
            print(f"Running Monte Carlo analysis with {n_samples} samples...")
        t0 = time.time()
        if optimized:   # one sequential RandomState(42) stream (monte_carlo.py:181-201)
            params = flatten.generate_parameter_arrays(self.uncertainty_params, n_samples, stream="seed_42")
        else:           # per-sample RandomState(i) streams
            params = flatten.generate_parameter_arrays(self.uncertainty_params, n_samples)
        summ, status, traj, lo = self.run_batch_arrays(initial_conditions, params)
        table = self._sample_table(summ, status, params, traj, lo)
        if self.verbose:
            print(f"Completed {table.n} out of {n_samples} simulations")
        out = results_mod.analyze_table(table, verbose=self.verbose)
        if optimized:
            el = time.time() - t0
            _, ws = dist.world()
            out["performance"] = {"total_time": el, "simulations_per_second": table.n / el,
                                  "cores_used": self.n_cores, "gpus_used": ws}
        return out

    def run_monte_carlo_device(self, initial_conditions, n_samples, seed=1234, precision="f64_fast", planar=False):
        """Throughput form for 100 k - 10 M samples (BASELINE configs 3-5): dispersions are drawn on the
        device (`sampling.synthetic_dispersions`, same distributions, torch generator), each rank
        integrates `n_samples / world` of them, summaries are all-gathered and the outlier filter +
        statistics run on the device (`analysis.device_statistics`).  Returns the statistics part of
        the analysis dict plus the gathered tensors; no per-sample dicts.

        precision: "f64_fast" (default) gives the reference's `apogee_altitude` - the global argmax over all
        steps, simulator.py:488-490 - end reason and step count on every sample of the parity sets (its blow-ups
        finish in the reference-order kernel), and with them the reference's outlier filter and statistics;
        "f32" is ~2.7x faster but on diverging samples only its `first_apogee_altitude` is within 0.1 % (its
        `apogee_altitude` on ~17 %, its end reason on ~52 %: DESIGN.md section 5), so n_outliers and the
        statistics differ from the reference's; "f64" runs everything in the reference-order gate kernel."""
        from . import sampling
        eng = shared_engine(self.device)
        eng.set_config(self._config())
        rank, ws = dist.world()
        lo, hi, _ = dist.shard_bounds(n_samples, rank, ws)
        prec = _abi.PRECISIONS[precision]
        torch.cuda.synchronize(eng.device)
        t0 = time.time()
        # This rank's samples in sub-batches of DEVICE_CHUNK (the draws of sub-batch j come from seed + rank and j alone, so a
        # run is reproducible for a given n_samples, world size and seed), every one handed to erpl_mc_submit_batch
        # as soon as it is drawn: generation, up to `depth` integrations and their tails overlap on the GPU.
        m = max(hi - lo, 1)
        gen_s = 0.0
        parts, inflight = [], []
        starts = list(range(0, m, self.DEVICE_CHUNK))
        # Sub-batches drawn before any of them is submitted: a sample budget (4 M samples, ~10 GB of wind tables at
        # K = 100), not a count.  A generation kernel enqueued behind a running flight launch waits for its waves to
        # drain, so a group is drawn on an idle stream first (1.5 ms per sub-batch) and then handed over in one go;
        # the next group is drawn while the tails of this one fly.  A sub-batch's inputs are released as soon as ITS
        # ticket has been checked (erpl_mc_check_batch waits for that batch alone).
        GROUP = max(1, (4 << 20) // self.DEVICE_CHUNK)
        keep = 2 * eng.get_overlap() + GROUP
        err = None
        try:
            for g0 in range(0, len(starts), GROUP):
                tg = time.time()
                group = []
                for j in range(g0, min(g0 + GROUP, len(starts))):
                    a = starts[j]
                    group.append(sampling.synthetic_dispersions(
                        min(self.DEVICE_CHUNK, m - a), self.rocket, self.motor, self.wind_model, initial_conditions, eng.device,
                        precision=prec, seed=seed + rank + 1000003 * j, uncertainty=self.uncertainty_params,
                        base_altitude_profile=self.base_altitude_profile, base_wind_profile=self.base_wind_profile,
                        planar=planar, engine=eng))
                gen_s += time.time() - tg
                for db in group:
                    parts.append(eng.submit(db))
                    inflight.append((eng.last_ticket, db))    # inputs outlive their batch
                while len(inflight) > keep:
                    eng.check(inflight.pop(0)[0])
            eng.wait()
            eng.check()                # host-blocking; raises if a lane hand-over timed out
            inflight.clear()
            summ = parts[0][0] if len(parts) == 1 else torch.cat([p[0] for p in parts], dim=1)
            status = parts[0][1] if len(parts) == 1 else torch.cat([p[1] for p in parts])
            summ, status = summ[:, : hi - lo], status[: hi - lo]
        except Exception as e:   # noqa: BLE001 - re-raised below
            if ws == 1:
                raise
            # the other ranks are waiting in the all-gather: take part with a shard marked incomplete, then raise
            err = e
            torch.cuda.synchronize(eng.device)
            summ, status = dist.failed_shard(hi - lo)
            if summ.device != eng.device and torch.distributed.get_backend() == "nccl":
                summ, status = summ.to(eng.device), status.to(eng.device)
        summ, status = dist.all_gather_summaries(summ, status, n_samples)
        if err is not None:
            raise err
        if ws > 1:
            TrajectoryEngine.raise_if_incomplete(status)   # another rank's shard failed: every rank refuses the result
        torch.cuda.synchronize(eng.device)
        t2 = time.time()
        out = analysis.device_statistics(summ, status)
        torch.cuda.synchronize(eng.device)
        t3 = time.time()
        out["summary"], out["status"] = summ, status
        out["performance"] = {"total_time": t3 - t0, "simulations_per_second": n_samples / (t3 - t0), "gpus_used": ws,
                              "precision": precision, "generate_s": gen_s, "integrate_and_gather_s": t2 - t0 - gen_s,
                              "statistics_s": t3 - t2, "sub_batches": len(parts), "in_flight": eng.get_overlap()}
        return out

    def run_optimized_monte_carlo(self, initial_conditions, n_samples=1000, chunk_size=None):
        return self.run_monte_carlo(initial_conditions, n_samples, optimized=True)

    def _create_output_directory(self):
        """monte_carlo.py:475-480."""
        return reports.create_output_directory()

    def _save_report(self, analysis, output_dir):
        """monte_carlo.py:482-560 (same files, keys and text format)."""
        return reports.save_report(self, analysis, output_dir)

    def _filter_physics_outliers(self, results):
        """monte_carlo.py:337-398."""
        valid, outliers = [], []
        for r in results:
            reasons = analysis.outlier_reasons(r.get("apogee_altitude", 0), r.get("range", 0), r.get("flight_time", 0))
            if reasons:
                r["outlier_reasons"] = reasons
                outliers.append(r)
            else:
                valid.append(r)
        return valid, outliers

    def _analyze_results(self, results):
        """monte_carlo.py:400-473."""
        return analysis.analyze(results, verbose=self.verbose)
