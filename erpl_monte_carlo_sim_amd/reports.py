"""Report writers of the Monte Carlo driver: the on-disk contract of
monte_carlo.py:475-560 — `monte_carlo_report.json` (same keys), `simulation_results/sim_<id>.json`
per valid sample, and the human-readable `monte_carlo_report.txt` (same lines and number formats).
Per-sample dumps are capped: at 100 k+ samples nobody wants 100 k files."""
import json
import os
from datetime import datetime

import numpy as np


def to_serializable(obj):
    """NumPy / torch containers -> plain Python for json (utils.py:208-223)."""
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, (np.floating, np.integer, np.bool_)):
        return obj.item()
    if isinstance(obj, dict):
        return {k: to_serializable(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [to_serializable(v) for v in obj]
    if hasattr(obj, "detach") and hasattr(obj, "cpu"):  # torch tensor
        return obj.detach().cpu().tolist()
    return obj


def object_parameters(obj):
    return {k: to_serializable(v) for k, v in obj.__dict__.items() if not k.startswith("_")}


def create_output_directory(root="outputs"):
    """monte_carlo.py:475-480."""
    out = os.path.join(root, "monte_carlo_" + datetime.now().strftime("%Y%m%d_%H%M%S"))
    os.makedirs(out, exist_ok=True)
    return out


_SECTIONS = (("Apogee Altitude Statistics:", "apogee_altitude_stats", "m"),
             ("Range Statistics:", "range_stats", "m"),
             ("Flight Time Statistics:", "flight_time_stats", "s"))


def report_text(report):
    """Lines of monte_carlo_report.txt (monte_carlo.py:520-560), without trailing newline handling."""
    s = report["simulation_summary"]
    lines = ["Monte Carlo Analysis Report", "=" * 50, "", f"Generated: {report['timestamp']}", "",
             "Simulation Summary:",
             f"  Valid simulations: {s['total_simulations']}",
             f"  Failed simulations: {s['failed_simulations']}",
             f"  Outlier simulations: {s['outlier_simulations']}",
             f"  Success rate: {s['success_rate']:.1f}%", ""]
    for title, key, unit in _SECTIONS:
        st = report[key]
        lines += [title,
                  f"  Mean: {st['mean']:.1f} {unit}",
                  f"  Standard Deviation: {st['std']:.1f} {unit}",
                  f"  Min: {st['min']:.1f} {unit}",
                  f"  Max: {st['max']:.1f} {unit}",
                  f"  95% Confidence Interval: [{st['percentiles'][0]:.1f}, {st['percentiles'][4]:.1f}] {unit}", ""]
    if "performance" in report:
        p = report["performance"]
        lines += ["Performance Statistics:", f"  Total time: {p['total_time']:.2f} s",
                  f"  Simulations per second: {p['simulations_per_second']:.1f}",
                  f"  Cores used: {p.get('cores_used', 0)}"]
        if "gpus_used" in p:
            lines.append(f"  GPUs used: {p['gpus_used']}")
        lines.append("")
    return lines


def build_report(analyzer, analysis):
    """The dict written to monte_carlo_report.json (monte_carlo.py:484-504)."""
    n_ok, n_fail, n_out = analysis["n_samples"], analysis["n_failed"], analysis["n_outliers"]
    report = {
        "timestamp": datetime.now().isoformat(),
        "simulation_summary": {"total_simulations": n_ok, "failed_simulations": n_fail,
                               "outlier_simulations": n_out,
                               "success_rate": n_ok / (n_ok + n_fail + n_out) * 100},
        "apogee_altitude_stats": analysis["apogee_altitude"],
        "range_stats": analysis["range"],
        "flight_time_stats": analysis["flight_time"],
        "uncertainty_parameters": to_serializable(analyzer.uncertainty_params),
        "parameter_ranges_observed": analysis.get("parameter_ranges_observed"),
        "rocket_parameters": object_parameters(analyzer.rocket),
        "motor_parameters": object_parameters(analyzer.motor),
        "atmosphere_parameters": object_parameters(analyzer.atmosphere),
        "wind_model_parameters": object_parameters(analyzer.wind_model),
    }
    if "performance" in analysis:
        report["performance"] = to_serializable(analysis["performance"])
    return report


def save_report(analyzer, analysis, output_dir, max_sample_dumps=1000):
    """monte_carlo.py:482-560.  Returns the report dict."""
    os.makedirs(output_dir, exist_ok=True)
    report = build_report(analyzer, analysis)
    with open(os.path.join(output_dir, "monte_carlo_report.json"), "w") as fh:
        json.dump(to_serializable(report), fh, indent=2)
    sims_dir = os.path.join(output_dir, "simulation_results")
    os.makedirs(sims_dir, exist_ok=True)
    for k, result in enumerate(analysis.get("results", [])):
        if k >= max_sample_dumps:
            break
        sim_id = result.get("simulation_id", k)
        with open(os.path.join(sims_dir, f"sim_{sim_id}.json"), "w") as fh:
            json.dump(to_serializable(result), fh)
    with open(os.path.join(output_dir, "monte_carlo_report.txt"), "w") as fh:
        fh.write("\n".join(report_text(report)) + "\n")
    return report
