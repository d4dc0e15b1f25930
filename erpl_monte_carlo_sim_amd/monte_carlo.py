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
        eng = shared_engine(self.device)
        eng.set_config(self._config())
        prec = _abi.PRECISIONS[self.precision]
        box = {}

        def runner(hb):
            db = DeviceBatch.from_host(hb, eng.device, prec)
            ids = [i for i in range(hb.n) if lo + i < n_traj_global]
            if ids:
                dt = min(self.dt_initial, 0.005)
                cap = int(np.ceil(self.max_time / dt / self.trajectory_stride)) + 4
                summ, status, traj, tlen = eng.run(db, traj_ids=ids, traj_stride=self.trajectory_stride, traj_cap=cap)
                box["traj"] = (ids, traj, tlen)
            else:
                summ, status = eng.run(db)
            return summ, status
        return runner, box

    def run_batch_arrays(self, initial_conditions, parameter_samples):
        """Integrate the given dispersed samples; returns (summary [16, n], status [n]) NumPy arrays
        (identical on every rank) and this rank's captured trajectories."""
        hb = flatten.dispersed_batch(self.rocket, self.motor, self.wind_model, initial_conditions,
                                     parameter_samples, self.base_altitude_profile, self.base_wind_profile)
        rank, ws = dist.world()
        lo, _, _ = dist.shard_bounds(hb.n, rank, ws)
        runner, box = self._gpu_runner(min(self.n_trajectories, hb.n), lo)
        summ, status = dist.run_sharded(hb, runner)
        return summ, status, box.get("traj"), lo

    def _result_dicts(self, summ, status, parameter_samples, traj, lo):
        results = []
        for i, params in enumerate(parameter_samples):
            st = int(status[i])
            results.append({
                "apogee_altitude": float(summ[_abi.SUM_APOGEE_ALT, i]),
                "apogee_time": float(summ[_abi.SUM_APOGEE_TIME, i]),
                "range": float(summ[_abi.SUM_RANGE, i]),
                "flight_time": float(summ[_abi.SUM_FLIGHT_TIME, i]),
                "rail_exit_time": float(summ[_abi.SUM_RAIL_EXIT_TIME, i]),
                "rail_exit_speed": float(summ[_abi.SUM_RAIL_EXIT_SPEED, i]),
                "rail_exit_angle_of_attack": float(summ[_abi.SUM_RAIL_EXIT_AOA, i]),
                "rail_exit_sideslip": float(summ[_abi.SUM_RAIL_EXIT_SIDESLIP, i]),
                "impact_position": [float(summ[r, i]) for r in (_abi.SUM_IMPACT_X, _abi.SUM_IMPACT_Y, _abi.SUM_IMPACT_Z)],
                "first_apogee_altitude": float(summ[_abi.SUM_FIRST_APOGEE_ALT, i]),
                "n_steps": int(summ[_abi.SUM_STEPS, i]),
                "termination": _END[st & 0xFF],
                "parachute_deployed": bool(st & _abi.ST_CHUTE),
                "simulation_id": i,
                "parameters": params,
            })
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
        params = (self._generate_parameter_samples_vectorized(n_samples) if optimized
                  else self._generate_parameter_samples(n_samples))
        summ, status, traj, lo = self.run_batch_arrays(initial_conditions, params)
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

    def run_monte_carlo_device(self, initial_conditions, n_samples, seed=1234, precision="f32", planar=False):
        """Throughput form for 100 k - 10 M samples (BASELINE configs 3-5): dispersions are drawn on the
        device (`sampling.synthetic_dispersions`, same distributions, torch generator), each rank
        integrates `n_samples / world` of them, summaries are all-gathered and the outlier filter +
        statistics run on the device (`analysis.device_statistics`).  Returns the statistics part of
        the analysis dict plus the gathered tensors; no per-sample dicts."""
        from . import sampling
        eng = shared_engine(self.device)
        eng.set_config(self._config())
        rank, ws = dist.world()
        lo, hi, _ = dist.shard_bounds(n_samples, rank, ws)
        prec = _abi.PRECISIONS[precision]
        t0 = time.time()
        db = sampling.synthetic_dispersions(max(hi - lo, 1), self.rocket, self.motor, self.wind_model,
                                            initial_conditions, eng.device, precision=prec, seed=seed + rank,
                                            uncertainty=self.uncertainty_params,
                                            base_altitude_profile=self.base_altitude_profile,
                                            base_wind_profile=self.base_wind_profile, planar=planar)
        summ, status = eng.run(db)
        summ, status = summ[:, : hi - lo], status[: hi - lo]
        summ, status = dist.all_gather_summaries(summ, status, n_samples)
        out = analysis.device_statistics(summ, status)
        torch.cuda.synchronize(eng.device)
        el = time.time() - t0
        out["summary"], out["status"] = summ, status
        out["performance"] = {"total_time": el, "simulations_per_second": n_samples / el, "gpus_used": ws}
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
