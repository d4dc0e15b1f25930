#!/usr/bin/env python3
"""Distil tools/pmc_port.sh: per flight dispatch, vector-ALU port occupancy = SQ_ACTIVE_INST_VALU (quad-cycles,
summed over the SIMDs) * 4 / (1024 SIMDs * wall cycles), wall cycles = GRBM_GUI_ACTIVE / 8 (sum over the XCDs)."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for prec in ("f32", "f64_fast"):
    rows = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(root, "gpurun_out", f"{tag}_port_{prec}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "erpl_flight" in r["Kernel_Name"]:
                rows[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    best = max(rows.values(), key=lambda d: d.get("SQ_INSTS_VALU", 0)) if rows else None
    if not best:
        continue
    wall = best["GRBM_GUI_ACTIVE"] / 8.0
    out[prec] = {"counters": best, "wall_cycles": wall,
                 "valu_port_busy": best["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * wall),
                 "cycles_per_valu_instruction": best["SQ_ACTIVE_INST_VALU"] * 4.0 / best["SQ_INSTS_VALU"],
                 "resident_waves_per_simd": best["SQ_WAVE_CYCLES"] * 4.0 / (1024.0 * wall),
                 "note": "largest erpl_flight dispatch of: bench.py --samples-per-gpu 1048576 --overlap 0 (one dense launch)"}
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(root, "profiles", sys.argv[2] if len(sys.argv) > 2 else tag + "_pmc_valu_port.json"), "w"), indent=1)
