#!/bin/bash
# usage (on the GPU box): tools/ab/run_ab.sh <out prefix> <bench args...> -- <variant tags...>   ("default" = the shipped library)
P=$1; shift
ARGS=(); while [ "$1" != "--" ]; do ARGS+=("$1"); shift; done; shift
for v in "$@"; do
  if [ "$v" = default ]; then L=""; else L=tools/ab/liberpl_mc_$v.so; fi
  ERPL_LIB=$L timeout -k 10 300 python bench.py --precision ${PREC:-f64_fast} --no-second-leg --cpu-seconds 0 --no-parity "${ARGS[@]}" > gpurun_out/${P}_$v.json 2>gpurun_out/${P}_$v.err || { echo "$v FAILED"; tail -3 gpurun_out/${P}_$v.err; exit 1; }
  python - "$v" gpurun_out/${P}_$v.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2]))
print("%-22s %8.3f M traj/s  %7.2f ms/pass  frac %.4f  lane_util %.3f" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], d["roofline"]["frac"], d["lane_utilisation"]))
PY
done
