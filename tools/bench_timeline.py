#!/usr/bin/env python3
"""Reduce a rocprofv3 --kernel-trace of the driver's bench command to the timeline figures the roofline cites:
per timed leg, the union of the erpl_flight_* dispatch intervals / passes (what a pass costs on the GPU when 8
passes overlap) and the mean number of concurrent dispatches.

    tools/bench_timeline.py <dir with *kernel_trace.csv> <bench json line of the same run> <out.json> [steps] [warmup]

A leg = W + K passes of one kernel build; pass i starts with dispatch i of erpl_rail_<build>.  The timed window of
a leg runs from the start of its (W+1)-th rail dispatch to the end of the last flight dispatch of that build
that started before the next leg's first rail dispatch."""
import csv
import glob
import json
import os
import sys


def load(dirname):
    rows = []
    for f in glob.glob(os.path.join(dirname, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         int(r.get("VGPR_Count") or r.get("Arch_VGPR_Count") or 0), int(r.get("Accum_VGPR_Count") or 0),
                         int(r.get("LDS_Block_Size") or 0), int(r.get("Scratch_Size") or 0)))
    rows.sort()
    return rows


def union_ns(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def main():
    d, line_path, out_path = sys.argv[1:4]
    K = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    W = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    rows = load(d)
    line = json.loads(open(line_path).read().strip().splitlines()[-1])
    legs = [(line["dtype"], line["ms_per_step"])]
    for k in ("f32", "f64_fast"):
        if k in line and k != line["dtype"]:
            legs.append((k, line[k]["ms_per_step"]))
    suffix = {"f32": "f32", "f64_fast": "f64f", "f64": "f64"}
    out = {"command": "GPU_MAX_HW_QUEUES=24 rocprofv3 --kernel-trace -- python3 bench.py --gpus 1 --steps %d --warmup %d" % (K, W),
           "note": "union = total time during which at least one erpl_flight dispatch of the leg's timed passes is running; "
                   "bench_ms_per_step is what the same (profiled) run printed; registers = (VGPR_Count, Accum_VGPR_Count) as "
                   "rocprofv3 reports them for a wave64 kernel: half of the per-lane counts of profiles/r4_kernel_resource_usage.txt "
                   "(128 = the 256-register two-wave build, 84 = the 168-register three-wave build)", "legs": {}}
    first_rail_after = {}
    def is_kernel(name, kind, build):
        """kind = "rail" | "flight"; the name may be mangled (erpl_rail_f64fE..., erpl_flight_f64fILb0...) or demangled."""
        key = "erpl_%s_%s" % (kind, build)
        i = name.find(key)
        if i < 0:
            return False
        nxt = name[i + len(key): i + len(key) + 1]
        return not (nxt.isalnum() and nxt not in "EI") or nxt in "EI<("     # "f64" must not match "f64f"

    rails_by = {p: [r for r in rows if is_kernel(r[2], "rail", suffix[p])] for p, _ in legs}
    for idx, (p, bench_ms) in enumerate(legs):
        rails = rails_by[p]
        if len(rails) < W + K:
            out["legs"][p] = {"error": "only %d rail dispatches found" % len(rails)}
            continue
        rails = rails[: W + K]                      # the leg's own passes (later blocks of the bench reuse the builds)
        t_begin = rails[W][0]
        t_next = None
        if idx + 1 < len(legs) and rails_by[legs[idx + 1][0]]:
            t_next = rails_by[legs[idx + 1][0]][0][0]
        after = [r for r in rows if r[0] > rails[-1][1] and "erpl_rail_" in r[2]]
        t_limit = t_next if t_next is not None else (after[0][0] if after else rows[-1][1] + 1)
        fl = [r for r in rows if is_kernel(r[2], "flight", suffix[p]) and t_begin <= r[0] < t_limit]
        # round 4: the fp64 throughput build hands its blow-ups to the reference-order kernel - one more flight dispatch
        # (erpl_flight_f64, the gate's own one-wave-per-SIMD instantiation) behind the launches of every pass; part of the pass
        sweep = [r for r in rows if p == "f64_fast" and is_kernel(r[2], "flight", "f64") and t_begin <= r[0] < t_limit]
        fl = sorted(fl + sweep)
        iv = [(s, e) for s, e, *_ in fl]
        u = union_ns(iv)
        tot = sum(e - s for s, e in iv)
        window = (max(e for _, e in iv) - t_begin) if iv else 0
        out["legs"][p] = {"passes": K, "flight_dispatches": len(fl), "window_ms": window / 1e6, "union_ms": u / 1e6,
                          "union_ms_per_pass": u / 1e6 / K, "window_ms_per_pass": window / 1e6 / K,
                          "sum_of_dispatch_ms": tot / 1e6, "mean_concurrent_dispatches": tot / u if u else None,
                          "mean_dispatch_ms": tot / 1e6 / len(fl) if fl else None, "bench_ms_per_step": bench_ms,
                          "union_over_bench": (u / 1e6 / K) / bench_ms if bench_ms else None,
                          "registers": sorted({(r[3], r[4]) for r in fl}), "lds_bytes": sorted({r[5] for r in fl}),
                          "scratch_bytes": sorted({r[6] for r in fl})}
        if sweep:
            out["legs"][p]["handoff_sweep"] = {"kernel": "erpl_flight_f64 (reference-order kernel; its waves take a whole SIMD and wait for one: the dispatch lasts, the work is two steps per record)",
                                               "dispatches": len(sweep), "sum_of_dispatch_ms": sum(e - s for s, e, *_ in sweep) / 1e6,
                                               "mean_dispatch_ms": sum(e - s for s, e, *_ in sweep) / 1e6 / len(sweep),
                                               "share_of_flight_dispatch_time": sum(e - s for s, e, *_ in sweep) / tot if tot else None}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
