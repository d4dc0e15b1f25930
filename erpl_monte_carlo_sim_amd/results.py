"""Per-sample result records of MonteCarloAnalyzer.run_monte_carlo as a lazy sequence.

The reference returns `analysis['results']` / `analysis['outliers']` as Python lists of dicts, one per
sample (monte_carlo.py:296-302, :337-398).  At 10^5 - 10^7 samples building those dicts costs more than
integrating the trajectories (6 us per sample against 0.2 us on the GPU), and nothing in the analysis needs
them: the outlier filter and the statistics work on the summary columns.  `LazyResults` keeps the columns
and materialises the dict of a sample - same keys in the same order, same values and value types as the
eager list the round-2 build produced - when it is indexed, sliced or iterated.
"""
from collections.abc import Sequence

import numpy as np

from . import _abi, analysis

END_NAMES = ("max_time", "ground_impact", "excessive_altitude", "coast_timeout", "apogee")

# result keys in the order of the eager construction (erpl_monte_carlo_sim_amd/monte_carlo.py, round 2)
SCALAR_COLUMNS = (("apogee_altitude", _abi.SUM_APOGEE_ALT), ("apogee_time", _abi.SUM_APOGEE_TIME),
                  ("range", _abi.SUM_RANGE), ("flight_time", _abi.SUM_FLIGHT_TIME),
                  ("rail_exit_time", _abi.SUM_RAIL_EXIT_TIME), ("rail_exit_speed", _abi.SUM_RAIL_EXIT_SPEED),
                  ("rail_exit_angle_of_attack", _abi.SUM_RAIL_EXIT_AOA),
                  ("rail_exit_sideslip", _abi.SUM_RAIL_EXIT_SIDESLIP),
                  ("first_apogee_altitude", _abi.SUM_FIRST_APOGEE_ALT))

VEC_PARAMS = ("initial_position_offset", "initial_velocity_offset", "initial_attitude_offset",
              "initial_angular_velocity_offset")
SCALAR_PARAMS = ("mass_multiplier", "thrust_multiplier", "wind_speed", "wind_direction", "density_multiplier")


class SampleTable:
    """Columns of all samples of one run: the [16, n] summary, the [n] status words, the parameter arrays
    (dict of [n, 3] / [n] arrays, flatten.generate_parameter_arrays) or the reference's list of parameter
    dicts, and the captured trajectories {sample id: dict}."""

    def __init__(self, summary, status, params, trajectories=None):
        self.summary = np.asarray(summary, dtype=np.float64)
        self.status = np.asarray(status)
        self.params = params
        self.trajectories = trajectories or {}
        self.n = self.summary.shape[1]
        if isinstance(params, dict):
            self._vec = {k: np.array(params[k], dtype=np.float64) for k in VEC_PARAMS}   # private copies: rows are handed out as views
            self._sca = {k: np.asarray(params[k], dtype=np.float64) for k in SCALAR_PARAMS if k in params}
            self._seed = np.asarray(params["random_seed"]).astype(np.int64)

    def parameters(self, i):
        if not isinstance(self.params, dict):
            return self.params[i]
        p = {k: self._vec[k][i] for k in VEC_PARAMS}
        for k, v in self._sca.items():
            p[k] = float(v[i])
        p["random_seed"] = int(self._seed[i])
        return p

    def record(self, i, with_reasons=False):
        s = self.summary
        r = {name: float(s[row, i]) for name, row in SCALAR_COLUMNS}
        r["impact_position"] = [float(s[_abi.SUM_IMPACT_X, i]), float(s[_abi.SUM_IMPACT_Y, i]), float(s[_abi.SUM_IMPACT_Z, i])]
        r["n_steps"] = int(s[_abi.SUM_STEPS, i])
        st = int(self.status[i])
        r["termination"] = END_NAMES[st & 0xFF]
        r["parachute_deployed"] = bool(st & _abi.ST_CHUTE)
        r["simulation_id"] = int(i)
        r["parameters"] = self.parameters(i)
        if i in self.trajectories:
            r["trajectory"] = self.trajectories[i]
        if with_reasons:
            r["outlier_reasons"] = analysis.outlier_reasons(r["apogee_altitude"], r["range"], r["flight_time"])
        return r


class LazyResults(Sequence):
    """`ids` (ascending sample ids) of a SampleTable as a sequence of result dicts built on access.

    A record that has been handed out is kept: indexing the same sample again returns the SAME dict, so that
    reference-style consumers which annotate records in place (`r['x'] = ...` while iterating) find their writes
    again.  The sequence itself is fixed (no append / sort / del); `tolist()` gives the plain mutable
    list the reference returns (monte_carlo.py:296-302) - the deviation is listed in INTEGRATION.md."""

    def __init__(self, table, ids, with_reasons=False):
        self.table, self.ids, self.with_reasons = table, np.asarray(ids, dtype=np.int64), with_reasons
        self._made = {}

    def __len__(self):
        return int(self.ids.size)

    def _record(self, i):
        r = self._made.get(i)
        if r is None:
            r = self._made[i] = self.table.record(i, self.with_reasons)
        return r

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self._record(int(i)) for i in self.ids[k]]
        return self._record(int(self.ids[k]))

    def __iter__(self):
        for i in self.ids:
            yield self._record(int(i))

    def tolist(self):
        """The eager list of dicts (what the reference returns): every record is materialised (and kept)."""
        return [self._record(int(i)) for i in self.ids]

    def __add__(self, other):        # `results + outliers` as the reference's lists allow
        return list(self) + list(other)

    def __radd__(self, other):
        return list(other) + list(self)

    def __eq__(self, other):
        return len(self) == len(other) and all(_same_record(a, b) for a, b in zip(self, other))

    def __repr__(self):
        return f"<LazyResults of {len(self)} samples>"

    def column(self, name):
        """One summary column of these samples as an array (no dicts built)."""
        row = dict(SCALAR_COLUMNS)[name]
        return self.table.summary[row, self.ids]


def _same_value(a, b):
    if isinstance(a, dict) and isinstance(b, dict):
        return a.keys() == b.keys() and all(_same_value(a[k], b[k]) for k in a)
    if isinstance(a, (np.ndarray, list, tuple)) or isinstance(b, (np.ndarray, list, tuple)):
        a, b = np.asarray(a), np.asarray(b)
        return a.shape == b.shape and bool(np.all((a == b) | ((a != a) & (b != b))))
    if isinstance(a, float) and isinstance(b, float):
        return a == b or (a != a and b != b)
    return a == b


def _same_record(a, b):
    return list(a.keys()) == list(b.keys()) and all(_same_value(a[k], b[k]) for k in a)


def parameter_ranges_of(table, ids):
    """analysis.parameter_ranges over the samples `ids`, from the parameter arrays (monte_carlo.py:425-441)."""
    if not isinstance(table.params, dict):
        return analysis.parameter_ranges(table.params[int(i)] for i in ids)
    out = {}
    if len(ids) == 0:
        return out
    for k in VEC_PARAMS:
        a = table._vec[k][ids]
        out[k] = {"min": a.min(axis=0).tolist(), "max": a.max(axis=0).tolist()}
    for k, v in table._sca.items():
        a = v[ids]
        out[k] = {"min": float(a.min()), "max": float(a.max())}
    seeds = table._seed[ids].astype(np.float64)
    out["random_seed"] = {"min": float(seeds.min()), "max": float(seeds.max())}
    return out


def analyze_table(table, verbose=False):
    """`MonteCarloAnalyzer._analyze_results` (monte_carlo.py:400-473) on the columns of a SampleTable: the same
    numbers as analysis.analyze on the eager list of dicts, with 'results' / 'outliers' as LazyResults."""
    s = table.summary
    if table.n == 0:
        raise ValueError("No valid simulation results")
    apo, rng, ft = s[_abi.SUM_APOGEE_ALT], s[_abi.SUM_RANGE], s[_abi.SUM_FLIGHT_TIME]
    bad = analysis.outlier_mask(apo, rng, ft)
    valid_ids, outlier_ids = np.flatnonzero(~bad), np.flatnonzero(bad)
    if verbose:
        for i in outlier_ids:
            print(f"Filtered outlier simulation {i}: {', '.join(analysis.outlier_reasons(apo[i], rng[i], ft[i]))}")
        print(f"Physics-based filtering: {valid_ids.size} valid, {outlier_ids.size} outliers")
    if valid_ids.size == 0:
        raise ValueError("No physically reasonable simulation results after outlier filtering")
    ok = ~bad
    return {
        "n_samples": int(valid_ids.size),
        "n_failed": 0,
        "n_outliers": int(outlier_ids.size),
        "apogee_altitude": analysis.calc_stats(apo[ok]),
        "range": analysis.calc_stats(rng[ok]),
        "flight_time": analysis.calc_stats(ft[ok]),
        "results": LazyResults(table, valid_ids),
        "outliers": LazyResults(table, outlier_ids, with_reasons=True),
        "parameter_ranges_observed": parameter_ranges_of(table, valid_ids),
    }
