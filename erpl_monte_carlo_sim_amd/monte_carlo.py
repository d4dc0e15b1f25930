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
from .engine import DeviceBatch
from .sampling import DEFAULT_UNCERTAINTY
from .simulator import shared_engine

_END = ("max_time", "ground_impact", "excessive_altitude", "coast_timeout", "apogee")


class MonteCarloAnalyzer:
    def __init__(self, rocket, motor, atmosphere, wind_model, device=None, verbose=True):
        self.rocket, self.motor, self.atmosphere, self.wind_model = rocket, motor, atmosphere, wind_model
        self.n_cores = os.cpu_count()
        self.base_altitude_profile = None    # monte_carlo.py:27-32
        self.base_wind_profile = None
        self.uncertainty_params = {k: (list(v) if isinstance(v, list) else v) for k, v in DEFAULT_UNCERTAINTY.items()}
        self.device = device
        self.verbose = verbose
        self.precision = "f64"               # "f64": reference-faithful gate; "f32": throughput kernel
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

    def _gpu_runner(self, n_traj_global, lo):
        """runner(HostBatch of this rank's shard) for dist.run_local_shard.  The samples that carry a
        trajectory (global index < n_traj_global: the first ones of the shard) go through the
        trajectory-capture build in a small batch of their own; everything else runs through the
        specialised non-capturing build (which also fast-forwards non-finite trajectories).  Samples are
        independent, so the split does not change a single bit of the summaries (tested)."""
        eng = shared_engine(self.device)
        eng.set_config(self._config())
        prec = _abi.PRECISIONS[self.precision]
        box = {}

        def runner(hb):
            m = max(0, min(hb.n, n_traj_global - lo))   # local samples 0..m-1 are captured
            parts = []
            if m > 0:
                head = hb.take(np.arange(m)) if m < hb.n else hb
                db = DeviceBatch.from_host(head, eng.device, prec)
                dt = min(self.dt_initial, 0.005)
                cap = int(np.ceil(self.max_time / dt / self.trajectory_stride)) + 4
                ids = list(range(m))
                summ, status, traj, tlen = eng.run(db, traj_ids=ids, traj_stride=self.trajectory_stride, traj_cap=cap)
                box["traj"] = (ids, traj, tlen)
                parts.append((summ, status))
            if m < hb.n:
                tail = hb.take(np.arange(m, hb.n)) if m > 0 else hb
                parts.append(eng.run(DeviceBatch.from_host(tail, eng.device, prec)))
            if len(parts) == 1:
                return parts[0]
            return torch.cat([p[0] for p in parts], dim=1), torch.cat([p[1] for p in parts])
        return runner, box

    def run_batch_arrays(self, initial_conditions, parameter_samples):
        """Integrate the given dispersed samples; returns (summary [16, n], status [n]) NumPy arrays
        (identical on every rank) and this rank's captured trajectories.  Each rank builds only its own
        shard [lo, hi) of the samples on the host (the per-sample seeds make them independent,
        monte_carlo.py:156-179)."""
        n = len(parameter_samples["random_seed"]) if isinstance(parameter_samples, dict) else len(parameter_samples)
        rank, ws = dist.world()
        lo, hi, _ = dist.shard_bounds(n, rank, ws)
        if isinstance(parameter_samples, dict):
            mine = {k: v[lo:hi] for k, v in parameter_samples.items()}
        else:
            mine = parameter_samples[lo:hi]
        hb = flatten.dispersed_batch(self.rocket, self.motor, self.wind_model, initial_conditions,
                                     mine, self.base_altitude_profile, self.base_wind_profile) if hi > lo else None
        runner, box = self._gpu_runner(min(self.n_trajectories, n), lo)
        summ, status = dist.run_local_shard(n, hb, runner)
        return summ, status, box.get("traj"), lo

    def _result_dicts(self, summ, status, parameter_samples, traj, lo):
        cols = {name: summ[row].tolist() for name, row in (
            ("apogee_altitude", _abi.SUM_APOGEE_ALT), ("apogee_time", _abi.SUM_APOGEE_TIME), ("range", _abi.SUM_RANGE),
            ("flight_time", _abi.SUM_FLIGHT_TIME), ("rail_exit_time", _abi.SUM_RAIL_EXIT_TIME),
            ("rail_exit_speed", _abi.SUM_RAIL_EXIT_SPEED), ("rail_exit_angle_of_attack", _abi.SUM_RAIL_EXIT_AOA),
            ("rail_exit_sideslip", _abi.SUM_RAIL_EXIT_SIDESLIP), ("first_apogee_altitude", _abi.SUM_FIRST_APOGEE_ALT))}
        impact = summ[[_abi.SUM_IMPACT_X, _abi.SUM_IMPACT_Y, _abi.SUM_IMPACT_Z]].T.tolist()
        steps = summ[_abi.SUM_STEPS].astype(np.int64).tolist()
        reason = [_END[k] for k in (status & 0xFF).tolist()]
        chute = ((status & _abi.ST_CHUTE) != 0).tolist()
        keys = list(cols)
        results = []
        for i, (params, *vals) in enumerate(zip(parameter_samples, *[cols[k] for k in keys])):
            r = dict(zip(keys, vals))
            r["impact_position"] = impact[i]
            r["n_steps"] = steps[i]
            r["termination"] = reason[i]
            r["parachute_deployed"] = chute[i]
            r["simulation_id"] = i
            r["parameters"] = params
            results.append(r)
        if traj is not None:
            ids, tr, tlen = traj
            tr, tlen = tr.cpu().numpy(), tlen.cpu().numpy()
            for m, i in enumerate(ids):
                k = int(tlen[m])
                rec = tr[m, :k]
                results[lo + i]["trajectory"] = {           # monte_carlo.py:298-302
                    "time": rec[:, 0] - summ[_abi.SUM_RAIL_EXIT_TIME, lo + i],
                    "altitude": rec[:, 3].copy(),
                    "position": rec[:, 1:4].copy(),
                }
        return results

    def run_monte_carlo(self, initial_conditions, n_samples=1000, n_processes=None, optimized=False):
        """monte_carlo.py:52-90 (and :92-154 when optimized=True: seed-42 stream + 'performance')."""
        if self.verbose:
            print(f"Running Monte Carlo analysis with {n_samples} samples...")
        t0 = time.time()
        if optimized:   # one sequential RandomState(42) stream (monte_carlo.py:181-201): a list by construction
            params = batch_params = self._generate_parameter_samples_vectorized(n_samples)
        else:           # per-sample RandomState(i) streams: keep the array form for the batch construction
            batch_params = flatten.generate_parameter_arrays(self.uncertainty_params, n_samples)
            params = flatten._arrays_to_params(batch_params)
        summ, status, traj, lo = self.run_batch_arrays(initial_conditions, batch_params)
        results = self._result_dicts(summ, status, params, traj, lo)
        if self.verbose:
            print(f"Completed {len(results)} out of {n_samples} simulations")
        out = self._analyze_results(results)
        if optimized:
            el = time.time() - t0
            _, ws = dist.world()
            out["performance"] = {"total_time": el, "simulations_per_second": len(results) / el,
                                  "cores_used": self.n_cores, "gpus_used": ws}
        return out

    def run_monte_carlo_device(self, initial_conditions, n_samples, seed=1234, precision="f64_fast", planar=False):
        """Throughput form for 100 k - 10 M samples (BASELINE configs 3-5): dispersions are drawn on the
        device (`sampling.synthetic_dispersions`, same distributions, torch generator), each rank
        integrates `n_samples / world` of them, summaries are all-gathered and the outlier filter +
        statistics run on the device (`analysis.device_statistics`).  Returns the statistics part of
        the analysis dict plus the gathered tensors; no per-sample dicts.

        precision: "f64_fast" (default) keeps the reference's `apogee_altitude` - the global argmax over all
        steps, simulator.py:488-490 - within 0.1 % on 99.9 % of reference-faithful samples, and with it the
        outlier filter and the statistics; "f32" is ~4x faster but on diverging samples only its
        `first_apogee_altitude` is within 0.1 % (its `apogee_altitude` on ~17 %, its end reason on ~52 %:
        DESIGN.md section 5), so n_outliers and the statistics differ from the reference's; "f64" is the
        reference-order gate kernel."""
        from . import sampling
        eng = shared_engine(self.device)
        eng.set_config(self._config())
        rank, ws = dist.world()
        lo, hi, _ = dist.shard_bounds(n_samples, rank, ws)
        prec = _abi.PRECISIONS[precision]
        torch.cuda.synchronize(eng.device)
        t0 = time.time()
        db = sampling.synthetic_dispersions(max(hi - lo, 1), self.rocket, self.motor, self.wind_model,
                                            initial_conditions, eng.device, precision=prec, seed=seed + rank,
                                            uncertainty=self.uncertainty_params,
                                            base_altitude_profile=self.base_altitude_profile,
                                            base_wind_profile=self.base_wind_profile, planar=planar, engine=eng)
        torch.cuda.synchronize(eng.device)
        t1 = time.time()
        summ, status = eng.run(db)
        summ, status = summ[:, : hi - lo], status[: hi - lo]
        summ, status = dist.all_gather_summaries(summ, status, n_samples)
        torch.cuda.synchronize(eng.device)
        t2 = time.time()
        out = analysis.device_statistics(summ, status)
        torch.cuda.synchronize(eng.device)
        t3 = time.time()
        out["summary"], out["status"] = summ, status
        out["performance"] = {"total_time": t3 - t0, "simulations_per_second": n_samples / (t3 - t0), "gpus_used": ws,
                              "precision": precision, "generate_s": t1 - t0, "integrate_and_gather_s": t2 - t1,
                              "statistics_s": t3 - t2}
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
