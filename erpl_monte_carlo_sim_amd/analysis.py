"""Outlier filter + statistics of the Monte Carlo driver (thin host glue).

Same bounds, keys and reason strings as monte_carlo.py:337-398 / :400-473 of the reference, but
evaluated on arrays (one pass over the gathered per-sample summaries) instead of a Python loop
over result dicts.  Checked against tests/golden/stats.json.
"""
import numpy as np

MAX_REASONABLE_APOGEE = 80000.0
MAX_REASONABLE_RANGE = 200000.0
MAX_REASONABLE_FLIGHT_TIME = 600.0
MIN_REASONABLE_APOGEE = 100.0
_THEORETICAL_MAX_ALTITUDE = 1200.0 ** 2 / (2 * 9.81)


def outlier_reasons(apogee, range_val, flight_time):
    """Reason strings of one sample, in the reference's order (monte_carlo.py:356-388)."""
    reasons = []
    if not np.isfinite(apogee) or not np.isfinite(range_val) or not np.isfinite(flight_time):
        reasons.append("non-finite values")
    if apogee > MAX_REASONABLE_APOGEE:
        reasons.append(f"apogee {apogee/1000:.1f} km > {MAX_REASONABLE_APOGEE/1000:.1f} km")
    elif apogee < MIN_REASONABLE_APOGEE:
        reasons.append(f"apogee {apogee:.1f} m < {MIN_REASONABLE_APOGEE:.1f} m")
    if range_val > MAX_REASONABLE_RANGE:
        reasons.append(f"range {range_val/1000:.1f} km > {MAX_REASONABLE_RANGE/1000:.1f} km")
    if flight_time > MAX_REASONABLE_FLIGHT_TIME:
        reasons.append(f"flight time {flight_time:.1f} s > {MAX_REASONABLE_FLIGHT_TIME:.1f} s")
    if apogee > _THEORETICAL_MAX_ALTITUDE * 1.2:
        reasons.append("apogee exceeds theoretical energy limit")
    return reasons


def outlier_mask(apogee, range_val, flight_time):
    """Vectorised form of the same tests: True where the sample is an outlier."""
    apogee, range_val, flight_time = (np.asarray(a, dtype=np.float64) for a in (apogee, range_val, flight_time))
    with np.errstate(invalid="ignore"):
        bad = ~np.isfinite(apogee) | ~np.isfinite(range_val) | ~np.isfinite(flight_time)
        bad |= (apogee > MAX_REASONABLE_APOGEE) | (apogee < MIN_REASONABLE_APOGEE)
        bad |= range_val > MAX_REASONABLE_RANGE
        bad |= flight_time > MAX_REASONABLE_FLIGHT_TIME
        bad |= apogee > _THEORETICAL_MAX_ALTITUDE * 1.2
    return bad


def calc_stats(values):
    """monte_carlo.py:444-459."""
    values = np.asarray(values, dtype=np.float64)
    values = values[np.isfinite(values)]
    if len(values) == 0:
        nan = float("nan")
        return {"mean": nan, "std": nan, "min": nan, "max": nan, "percentiles": [nan] * 5}
    return {"mean": float(np.mean(values)), "std": float(np.std(values)), "min": float(np.min(values)),
            "max": float(np.max(values)), "percentiles": np.percentile(values, [5, 25, 50, 75, 95]).tolist()}


def parameter_ranges(param_dicts):
    """Observed min/max of every sampled parameter over the valid samples (monte_carlo.py:425-441)."""
    out = {}
    for params in param_dicts:
        for key, val in params.items():
            arr = np.array(val)
            if key not in out:
                out[key] = {"min": arr.astype(float), "max": arr.astype(float)}
            else:
                out[key]["min"] = np.minimum(out[key]["min"], arr)
                out[key]["max"] = np.maximum(out[key]["max"], arr)
    for key in out:
        out[key]["min"] = out[key]["min"].tolist()
        out[key]["max"] = out[key]["max"].tolist()
    return out


def analyze(results, verbose=False):
    """`MonteCarloAnalyzer._analyze_results` on a list of per-sample dicts (None = failed sample).
    Raises the reference's ValueErrors when nothing (reasonable) is left (monte_carlo.py:405-412)."""
    initial = [r for r in results if r is not None]
    if len(initial) == 0:
        raise ValueError("No valid simulation results")
    apo = np.array([r.get("apogee_altitude", 0) for r in initial], dtype=np.float64)
    rng = np.array([r.get("range", 0) for r in initial], dtype=np.float64)
    ft = np.array([r.get("flight_time", 0) for r in initial], dtype=np.float64)
    bad = outlier_mask(apo, rng, ft)
    valid, outliers = [], []
    for r, b, a_, r_, f_ in zip(initial, bad, apo, rng, ft):
        if b:
            r["outlier_reasons"] = outlier_reasons(a_, r_, f_)
            outliers.append(r)
            if verbose:
                print(f"Filtered outlier simulation {r.get('simulation_id', '?')}: {', '.join(r['outlier_reasons'])}")
        else:
            valid.append(r)
    if verbose:
        print(f"Physics-based filtering: {len(valid)} valid, {len(outliers)} outliers")
    if len(valid) == 0:
        raise ValueError("No physically reasonable simulation results after outlier filtering")
    ok = ~bad
    return {
        "n_samples": len(valid),
        "n_failed": len(results) - len(initial),
        "n_outliers": len(outliers),
        "apogee_altitude": calc_stats(apo[ok]),
        "range": calc_stats(rng[ok]),
        "flight_time": calc_stats(ft[ok]),
        "results": valid,
        "outliers": outliers,
        "parameter_ranges_observed": parameter_ranges(r.get("parameters", {}) for r in valid),
    }


def device_statistics(summary, status=None):
    """Outlier filter + statistics on the gathered [16, n] summary tensor WITHOUT leaving the device
    (SURVEY.md §8f-3): the same bounds as `outlier_mask` and the same statistics as `calc_stats`
    (NumPy-style linear-interpolated percentiles, population std) with torch ops, for the 100 k - 10 M
    sample runs where a Python list of result dicts is not an option.  Works on CPU tensors too."""
    import torch
    from . import _abi
    s = summary.to(torch.float64)
    apo, rng, ft = s[_abi.SUM_APOGEE_ALT], s[_abi.SUM_RANGE], s[_abi.SUM_FLIGHT_TIME]
    bad = ~torch.isfinite(apo) | ~torch.isfinite(rng) | ~torch.isfinite(ft)
    bad |= (apo > MAX_REASONABLE_APOGEE) | (apo < MIN_REASONABLE_APOGEE)
    bad |= rng > MAX_REASONABLE_RANGE
    bad |= ft > MAX_REASONABLE_FLIGHT_TIME
    bad |= apo > _THEORETICAL_MAX_ALTITUDE * 1.2
    ok = ~bad
    n_valid = int(ok.sum().item())
    if n_valid == 0:
        raise ValueError("No physically reasonable simulation results after outlier filtering")
    q = torch.tensor([0.05, 0.25, 0.5, 0.75, 0.95], dtype=torch.float64, device=s.device)

    def stats(v):
        v = v[ok]
        # torch.quantile is limited to 16 M elements; sort-based linear interpolation has no limit
        vs, _ = torch.sort(v)
        pos = q * (vs.numel() - 1)
        lo = pos.floor().long()
        hi = torch.clamp(lo + 1, max=vs.numel() - 1)
        frac = pos - lo.to(torch.float64)
        pct = vs[lo] + (vs[hi] - vs[lo]) * frac
        return {"mean": float(v.mean().item()), "std": float(v.std(unbiased=False).item()),
                "min": float(vs[0].item()), "max": float(vs[-1].item()), "percentiles": pct.tolist()}

    out = {"n_samples": n_valid, "n_failed": 0, "n_outliers": int(bad.sum().item()),
           "apogee_altitude": stats(apo), "range": stats(rng), "flight_time": stats(ft),
           "valid_mask": ok}
    if status is not None:
        st = status.to(torch.int64) & 0xFF
        out["termination_counts"] = {name: int((st == code).sum().item()) for code, name in
                                     enumerate(("max_time", "ground_impact", "excessive_altitude",
                                                "coast_timeout", "apogee"))}
        out["n_non_finite"] = int(((status.to(torch.int64) & _abi.ST_NAN) != 0).sum().item())
        n_inc = int(((status.to(torch.int64) & _abi.ST_INCOMPLETE) != 0).sum().item())
        if n_inc:   # a lane hand-over timed out: these samples were never integrated (include/erpl_mc.h)
            raise _abi.IncompleteBatch(f"{n_inc} sample(s) carry ERPL_ST_INCOMPLETE: refusing to compute statistics on them")
    return out
