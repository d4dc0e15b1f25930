#!/bin/bash
# What the waves of the fp64 throughput kernel wait for, on ONE dense dispatch (1 M Set S samples, no overlap):
#   tools/pmc_wait.sh <tag>   ->  gpurun_out/<tag>_wait_{a,b}/ ; prints per-wave-cycle shares
tag="$1"; cd "$(dirname "$0")/.." && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=24
B="python3 bench.py --samples-per-gpu 1048576 --overlap 0 --steps 2 --warmup 1 --cpu-seconds 0 --no-parity --no-cfg5 --no-api --no-second-leg --precision f64_fast"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d gpurun_out/${tag}_wait_a -o run --output-format csv -- $B > gpurun_out/${tag}_wait_a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INST_CYCLES_SALU -d gpurun_out/${tag}_wait_b -o run --output-format csv -- $B > gpurun_out/${tag}_wait_b.log 2>&1 || exit 1
python3 - "$tag" <<'PY'
import csv, collections, glob, sys
tag = sys.argv[1]
rows = collections.defaultdict(dict)
for f in glob.glob(f'gpurun_out/{tag}_wait_?/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'erpl_flight' in r['Kernel_Name']:
            rows[(f.split('_wait_')[1][0], r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
best = {}
for (p, d), c in rows.items():
    k = 'SQ_WAVE_CYCLES' if p == 'a' else 'SQ_INSTS_VALU'
    if p not in best or c.get(k, 0) > best[p].get(k, 0): best[p] = c
c = {**best.get('a', {}), **best.get('b', {})}
wc = c['SQ_WAVE_CYCLES']
for k in sorted(c): print(f'{k:26s} {c[k]:16.0f}  {c[k]/wc:8.4f} of wave-cycles   {c[k]/c["SQ_INSTS_VALU"]:8.4f} per VALU instruction')
PY
