#!/usr/bin/env python3
"""Distil gpurun_out/<tag>_* (written by tools/refresh_profiles.sh on the GPU box) into profiles/:
   tools/collect_profiles.py <tag> <round-prefix, e.g. r1>"""
import collections
import csv
import glob
import json
import os
import sys

tag, pre = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go = os.path.join(root, "gpurun_out")
prof = os.path.join(root, "profiles")


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0][:60]


def pmc(sub):
    """Counter totals per PASS over a batch.  A pass is one erpl_rail_* dispatch followed by one erpl_flight_*
    dispatch per phase (with lane adoption: the main launch and two sweep launches), so the flight kernel's
    counters are summed over its dispatches and divided by the number of rail dispatches of the same build."""
    tot, cnt = collections.Counter(), collections.Counter()
    for f in glob.glob(os.path.join(go, f"{tag}_{sub}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.startswith("erpl_"):
                k = k.split("<")[0] + "." + r["Counter_Name"]
                tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    passes = {}
    for k in tot:
        kern, counter = k.split(".")
        rail = kern.replace("erpl_flight_", "erpl_rail_") + "." + counter
        if rail not in cnt and kern == "erpl_flight_f64":      # the hand-over sweep of the f64_fast passes (no gate leg in this run)
            rail = "erpl_rail_f64f." + counter
        passes[k] = cnt.get(rail, cnt[k])
    return {k: tot[k] / passes[k] for k in sorted(tot)}, (max(passes.values()) if passes else 0)


# bench lines
line = json.loads(open(os.path.join(go, f"{tag}_bench.json")).read().strip().splitlines()[-1])
json.dump(line, open(os.path.join(prof, f"{pre}_bench_line_final.json"), "w"), indent=1)
extra = {}
keys = ("value", "ms_per_step", "dtype", "trajectory_steps_per_s", "lane_utilisation", "roofline")
for w in ("set_p_apogee", "set_p_apogee_gate", "set_p_full", "csv_chute", "set_s_1m", "cfg5_share", "cfg5_share_compaction"):
    p = os.path.join(go, f"{tag}_bench_{w}.json")
    if os.path.exists(p) and os.path.getsize(p):
        d = json.loads(open(p).read().strip().splitlines()[-1])
        extra[w] = {k: d[k] for k in keys if k in d}
        extra[w]["workload"] = d["config"]["workload"]
        for leg in ("f64_fast", "f32"):      # the secondary leg of the run, whichever it is
            if leg in d:
                extra[w][leg] = {k: d[leg][k] for k in keys if k in d[leg]}
json.dump(extra, open(os.path.join(prof, f"{pre}_bench_other_workloads.json"), "w"), indent=1)

# kernel stats (names truncated: torch's are hundreds of characters long)
for sub, out in (("stats", "bench_kernel_stats"), ("stats_gate", "bench_f64_gate_kernel_stats")):
    rows = []
    for f in glob.glob(os.path.join(go, f"{tag}_{sub}", "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
    if not rows:
        continue
    with open(os.path.join(prof, f"{pre}_{out}.csv"), "w") as o:
        o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
        for r in rows[:12]:
            o.write(",".join(['"%s"' % short(r["Name"])] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")]) + "\n")

# HBM traffic per launch, per kernel build (the bench command runs the f32 leg, then the f64_fast leg)
fe, n = pmc("pmc_fetch")
wr, _ = pmc("pmc_write")
traffic = {"command": "rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python3 bench.py --steps 6 --warmup 3 --cpu-seconds 0 --no-parity --no-cfg5 --no-api",
           "workload": line["config"]["workload"], "passes_averaged": n, "gfx950_fetch_correction": 2.0,
           "note": "FETCH_SIZE on gfx950 reads 1/2 of a wide coalesced stream (MI355X_MICROARCH.md HBM section), so the read side is doubled; our loads are 4- and 8-byte-per-lane, for which the guide calls the counter uncalibrated: the corrected figure is an upper bound, the raw one a lower bound. The written bytes are 12 scattered 8-byte summary rows per sample, each costing a 32/64-byte write transaction, plus the rail kernel's resume records."}
for name, suf in (("f32", "f32"), ("f64_fast", "f64f")):
    fk = fe.get(f"erpl_flight_{suf}.FETCH_SIZE", 0.0); wk = wr.get(f"erpl_flight_{suf}.WRITE_SIZE", 0.0)
    if name == "f64_fast":      # + the sweep of its hand-over queue by the reference-order kernel (part of every pass)
        sf, sw = fe.get("erpl_flight_f64.FETCH_SIZE", 0.0), wr.get("erpl_flight_f64.WRITE_SIZE", 0.0)
        traffic.setdefault("f64_fast_handoff_sweep", {"kernel": "erpl_flight_f64<..,2>", "fetch_size_kb_raw": sf, "write_size_kb": sw})
        fk, wk = fk + sf, wk + sw
    rf = fe.get(f"erpl_rail_{suf}.FETCH_SIZE", 0.0); rw = wr.get(f"erpl_rail_{suf}.WRITE_SIZE", 0.0)
    traffic[name] = {"kernel": f"erpl_flight_{suf}", "fetch_size_kb_raw": fk, "write_size_kb": wk, "rail_fetch_size_kb_raw": rf,
                     "rail_write_size_kb": rw, "traffic_bytes_per_launch": (2.0 * fk + wk) * 1024.0,
                     "launch": "one pass over the batch = all erpl_flight dispatches behind one erpl_rail dispatch"}
json.dump(traffic, open(os.path.join(prof, f"{pre}_hbm_traffic.json"), "w"), indent=1)

out = {}
for sub, key in (("pmc_sq", "bench_command_f64_fast_then_f32"), ("pmc_sq_gate", "bench_command_f64_gate")):
    sq, n = pmc(sub)
    if sq:
        out[key] = {"passes_averaged": n, "counters_per_pass": sq}
json.dump(out, open(os.path.join(prof, f"{pre}_pmc_sq_counters.json"), "w"), indent=1)
print(json.dumps(line)[:600])
print(json.dumps(extra)[:3000])
