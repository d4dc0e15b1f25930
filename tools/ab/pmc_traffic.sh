#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per pass of the fp64 throughput leg for a library variant: tools/ab/pmc_traffic.sh <tag> [variant]
tag=$1; v=$2; cd "$(dirname "$0")/../.." && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=24
[ -n "$v" ] && export ERPL_LIB=tools/ab/liberpl_mc_$v.so
B="python3 bench.py --steps 6 --warmup 3 --cpu-seconds 0 --no-parity --no-cfg5 --no-api --no-second-leg"
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c -d gpurun_out/${tag}_$c -o run --output-format csv -- $B > gpurun_out/${tag}_$c.log 2>&1 || exit 1; done
python3 - "$tag" <<'PY'
import csv,glob,sys,collections
tag=sys.argv[1]
for c in ("FETCH_SIZE","WRITE_SIZE"):
    tot=collections.Counter(); n=collections.Counter()
    for f in glob.glob(f"gpurun_out/{tag}_{c}/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            nm=r["Kernel_Name"]
            k="rail" if "erpl_rail" in nm else ("sweep_f64" if "erpl_flight_f64<" in nm or "erpl_flight_f64I" in nm else ("flight" if "erpl_flight" in nm else None))
            if k: tot[k]+=float(r["Counter_Value"]); n[k]+=1
    print(c, "KB per pass: flight_f64f %.0f, hand-over sweep (erpl_flight_f64) %.0f, rail %.0f"%(tot["flight"]/max(n["rail"],1), tot["sweep_f64"]/max(n["rail"],1), tot["rail"]/max(n["rail"],1)), "(dispatches", n["flight"], n["sweep_f64"], "passes", n["rail"],")")
PY
