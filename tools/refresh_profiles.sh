#!/bin/bash
# Re-measure the judged artefacts on the GPU box (run from the repo root through gpurun):
#   tools/refresh_profiles.sh <tag>
# writes gpurun_out/<tag>_*; tools/collect_profiles.py <tag> then distils them into profiles/.
tag="$1"
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
# rocprofv3 starts the HIP runtime before python does: the hardware-queue limit the package sets on import must already be there
export GPU_MAX_HW_QUEUES=24
B="python3 bench.py --steps 6 --warmup 3 --cpu-seconds 0 --no-parity"
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err &&
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_stats -o run --output-format csv -- $B > gpurun_out/${tag}_stats.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${tag}_pmc_fetch -o run --output-format csv -- $B > gpurun_out/${tag}_pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${tag}_pmc_write -o run --output-format csv -- $B > gpurun_out/${tag}_pmc_write.log 2>&1 &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM \
  -d gpurun_out/${tag}_pmc_sq -o run --output-format csv -- $B > gpurun_out/${tag}_pmc_sq.log 2>&1 &&
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_stats_gate -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-parity --no-second-leg --precision f64 > gpurun_out/${tag}_stats_gate.log 2>&1 &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM \
  -d gpurun_out/${tag}_pmc_sq_gate -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-parity --no-second-leg --precision f64 > gpurun_out/${tag}_pmc_sq_gate.log 2>&1 &&
python3 bench.py --workload set_p_apogee --steps 6 --warmup 3 --cpu-seconds 0 --no-parity > gpurun_out/${tag}_bench_set_p_apogee.json 2>/dev/null &&
python3 bench.py --workload set_p_apogee --steps 2 --warmup 1 --cpu-seconds 0 --no-parity --no-second-leg --precision f64 > gpurun_out/${tag}_bench_set_p_apogee_gate.json 2>/dev/null &&
python3 bench.py --workload set_p_full --steps 6 --warmup 3 --cpu-seconds 0 --no-parity > gpurun_out/${tag}_bench_set_p_full.json 2>/dev/null &&
python3 bench.py --workload csv_chute --steps 6 --warmup 3 --cpu-seconds 0 --no-parity > gpurun_out/${tag}_bench_csv_chute.json 2>/dev/null &&
python3 bench.py --samples-per-gpu 1048576 --steps 6 --warmup 3 --cpu-seconds 0 --no-parity > gpurun_out/${tag}_bench_set_s_1m.json 2>/dev/null &&
python3 bench.py --workload csv_chute --samples-per-gpu 1250000 --steps 4 --warmup 2 --cpu-seconds 0 --no-parity > gpurun_out/${tag}_bench_cfg5_share.json 2>/dev/null &&
python3 bench.py --workload csv_chute --samples-per-gpu 1250000 --chunk 2048 --overlap 2 --steps 2 --warmup 1 --cpu-seconds 0 --no-parity > gpurun_out/${tag}_bench_cfg5_share_compaction.json 2>/dev/null
