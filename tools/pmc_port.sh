#!/bin/bash
# Vector-ALU port occupancy of the fp32 flight kernel on ONE dense dispatch (1 M Set S samples, no overlap:
# rocprofv3 --pmc serialises dispatches, which does not matter for a single large one).
#   tools/pmc_port.sh <tag>   ->  gpurun_out/<tag>_port/  (distil with tools/pmc_port.py <tag>)
tag="$1"; cd "$(dirname "$0")/.." && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=24
for prec in ${PRECS:-f32 f64_fast}; do
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/${tag}_port_$prec -o run --output-format csv -- \
  python3 bench.py --samples-per-gpu 1048576 --overlap 0 --steps 2 --warmup 1 --cpu-seconds 0 --no-parity --no-cfg5 --no-api --no-second-leg --precision $prec > gpurun_out/${tag}_port_$prec.log 2>&1 || exit 1
done
