#!/usr/bin/env python3
"""Durations of the hand-over sweep dispatches (erpl_flight_f64<..,2>) in a rocprofv3 --kernel-trace directory."""
import csv, glob, sys
import numpy as np
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
sw = [(e - s) / 1e6 for s, e, n in rows if "erpl_flight_f64<" in n or "erpl_flight_f64I" in n]
fl = [(e - s) / 1e6 for s, e, n in rows if "erpl_flight_f64f" in n and (e - s) > 1e6]
print(f"{d}: sweep dispatches {len(sw)}, ms mean {np.mean(sw):.2f} median {np.median(sw):.2f} min {np.min(sw):.2f} max {np.max(sw):.2f}; "
      f"f64f dispatches > 1 ms: {len(fl)}, mean {np.mean(fl):.1f}")
