#!/bin/bash
# Dynamic instruction mix of an fp64 flight kernel (two PMC passes): tools/pmc_mix64.sh <tag> [diag_steps args]
# Run on the GPU box from the repo root; summaries land in gpurun_out/pmc_mix_<tag>_{a,b}.
tag="$1"; shift
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 \
  -d gpurun_out/pmc_mix_${tag}_a -o run --output-format csv -- python3 tools/diag_steps.py "$@" > gpurun_out/pmc_mix_${tag}_a.log 2>&1 &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_INT64 \
  -d gpurun_out/pmc_mix_${tag}_b -o run --output-format csv -- python3 tools/diag_steps.py "$@" > gpurun_out/pmc_mix_${tag}_b.log 2>&1
python3 - "$tag" <<'PY'
import csv, collections, glob, sys, re
tag = sys.argv[1]
c = collections.Counter(); n = collections.Counter()
for f in glob.glob(f'gpurun_out/pmc_mix_{tag}_?/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'erpl_flight' in r['Kernel_Name']:   # (all flight dispatches of a run: main launch + the hand-over sweep of the f64_fast build)
            c[r['Counter_Name']] += float(r['Counter_Value'])
        if 'erpl_rail' in r['Kernel_Name']:     # one per run
            n[r['Counter_Name']] += 1
it = None
for l in open(f'gpurun_out/pmc_mix_{tag}_a.log'):
    m = re.search(r'wave_iters=(\d+)', l)
    if m: it = int(m.group(1))
print('wave_iters', it, 'runs', dict(n))
for k in sorted(c): print(f'{k:28s} {c[k]/n[k]/it:10.1f} per wave-step')
PY
