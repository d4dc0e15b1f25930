#!/bin/bash
# usage: ov_sweep.sh <label> <bench args...>
L=$1; shift
for o in 8 4 3 5 8 4 3 5; do
  timeout -k 10 200 python bench.py --cpu-seconds 0 --no-parity --no-cfg5 --no-api --overlap $o "$@" > gpurun_out/ovs.json 2>gpurun_out/ovs.err || { echo FAILED; tail -n 3 gpurun_out/ovs.err; exit 1; }
  python - "$L" $o <<'PY'
import json,sys
d=json.load(open("gpurun_out/ovs.json"))
s="%s overlap %s: %s %.3f M traj/s %.2f ms frac %.4f" % (sys.argv[1], sys.argv[2], d["dtype"], d["value"]/1e6, d["ms_per_step"], d["roofline"]["frac"])
for k in ("f32","f64_fast"):
    if k in d and isinstance(d[k],dict) and "value" in d[k]: s+=" | %s %.3f M %.2f ms" % (k, d[k]["value"]/1e6, d[k]["ms_per_step"])
print(s, flush=True)
PY
done
