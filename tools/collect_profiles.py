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
    tot, cnt = collections.Counter(), collections.Counter()
    for f in glob.glob(os.path.join(go, f"{tag}_{sub}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.startswith("erpl_"):
                k = k.split("<")[0] + "." + r["Counter_Name"]
                tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in sorted(tot)}, (max(cnt.values()) if cnt else 0)


# bench lines
line = json.loads(open(os.path.join(go, f"{tag}_bench.json")).read().strip().splitlines()[-1])
json.dump(line, open(os.path.join(prof, f"{pre}_bench_line_final.json"), "w"), indent=1)
extra = {}
for w in ("set_p_apogee", "set_p_full", "csv_chute"):
    p = os.path.join(go, f"{tag}_bench_{w}.json")
    if os.path.exists(p) and os.path.getsize(p):
        d = json.loads(open(p).read().strip().splitlines()[-1])
        extra[w] = {k: d[k] for k in ("value", "ms_per_step", "trajectory_steps_per_s", "lane_utilisation", "roofline") if k in d}
json.dump(extra, open(os.path.join(prof, f"{pre}_bench_other_workloads.json"), "w"), indent=1)

# kernel stats (names truncated: torch's are hundreds of characters long)
rows = []
for f in glob.glob(os.path.join(go, f"{tag}_stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
with open(os.path.join(prof, f"{pre}_bench_kernel_stats.csv"), "w") as o:
    o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
    for r in rows[:12]:
        o.write(",".join(['"%s"' % short(r["Name"])] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")]) + "\n")

# HBM traffic
fe, n = pmc("pmc_fetch")
wr, _ = pmc("pmc_write")
fk = fe.get("erpl_flight_f32.FETCH_SIZE", 0.0); wk = wr.get("erpl_flight_f32.WRITE_SIZE", 0.0)
rf = fe.get("erpl_rail_f32.FETCH_SIZE", 0.0); rw = wr.get("erpl_rail_f32.WRITE_SIZE", 0.0)
json.dump({
    "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-parity --pipeline 1",
    "kernel": "erpl_flight_f32", "workload": line["config"]["workload"], "launches_averaged": n,
    "fetch_size_kb_raw": fk, "write_size_kb": wk, "rail_fetch_size_kb_raw": rf, "rail_write_size_kb": rw,
    "gfx950_fetch_correction": 2.0,
    "traffic_bytes_per_launch": (2.0 * fk + wk) * 1024.0,
    "note": "FETCH_SIZE on gfx950 reads 1/2 of a wide coalesced stream (MI355X_MICROARCH.md HBM section), so the read side is doubled; our loads are 4-byte-per-lane dwords, for which the guide calls the counter uncalibrated: the corrected figure is an upper bound, the raw one a lower bound. The written bytes are 12 scattered 8-byte summary rows per sample, each costing a 32/64-byte write transaction, plus the rail kernel's resume records.",
}, open(os.path.join(prof, f"{pre}_hbm_traffic.json"), "w"), indent=1)

sq, n = pmc("pmc_sq")
p = os.path.join(prof, f"{pre}_pmc_sq_counters.json")
old = json.load(open(p)) if os.path.exists(p) else {}
old["sq_bench_final_kernel"] = sq
old["sq_bench_final_kernel_launches_averaged"] = n
json.dump(old, open(p, "w"), indent=1)
print(json.dumps(line)[:600])
print(extra)
