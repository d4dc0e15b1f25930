#!/bin/bash
# Re-measure the judged artefacts on the GPU box (run from the repo root through gpurun, in two or three calls -
# every step is skipped when its output exists):   tools/refresh_profiles.sh <tag> [part: a|b|c]
# writes gpurun_out/<tag>_*; tools/collect_profiles.py <tag> <prefix> and tools/bench_timeline.py distil them into profiles/.
tag="$1"; part="${2:-abc}"
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
# rocprofv3 starts the HIP runtime before python does: the hardware-queue limit the package sets on import must already be there
export GPU_MAX_HW_QUEUES=24
S="--cpu-seconds 0 --no-parity --no-cfg5 --no-api"
B="python3 bench.py --steps 6 --warmup 3 $S"
run() { out="$1"; shift; [ -s "$out" ] || "$@" > "$out" 2> "${out%.json}.err" || { echo "FAILED: $*"; exit 1; }; }
if [[ $part == *a* ]]; then
run gpurun_out/${tag}_bench.json python3 bench.py --gpus 1 --steps 20 --warmup 5
# the driver's command under the kernel trace: timeline of the two timed legs (tools/bench_timeline.py) + per-kernel stats
[ -d gpurun_out/${tag}_stats ] || rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_stats -o run --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_stats.json 2> gpurun_out/${tag}_stats.log || exit 1
python3 tools/bench_timeline.py gpurun_out/${tag}_stats gpurun_out/${tag}_stats.json gpurun_out/${tag}_timeline.json 20 5 > /dev/null || exit 1
[ -d gpurun_out/${tag}_pmc_fetch ] || rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${tag}_pmc_fetch -o run --output-format csv -- $B > gpurun_out/${tag}_pmc_fetch.log 2>&1 || exit 1
[ -d gpurun_out/${tag}_pmc_write ] || rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${tag}_pmc_write -o run --output-format csv -- $B > gpurun_out/${tag}_pmc_write.log 2>&1 || exit 1
[ -d gpurun_out/${tag}_pmc_sq ] || rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM \
  -d gpurun_out/${tag}_pmc_sq -o run --output-format csv -- $B > gpurun_out/${tag}_pmc_sq.log 2>&1 || exit 1
fi
if [[ $part == *b* ]]; then
[ -d gpurun_out/${tag}_stats_gate ] || rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_stats_gate -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 $S --no-second-leg --precision f64 > gpurun_out/${tag}_stats_gate.log 2>&1 || exit 1
[ -d gpurun_out/${tag}_pmc_sq_gate ] || rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM \
  -d gpurun_out/${tag}_pmc_sq_gate -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 $S --no-second-leg --precision f64 > gpurun_out/${tag}_pmc_sq_gate.log 2>&1 || exit 1
tools/pmc_mix64.sh ${tag}_f64f --precision f64_fast --planar --apogee --n 32768 > gpurun_out/${tag}_pmc_mix_f64f.txt 2>&1 || exit 1
tools/pmc_port.sh ${tag} || exit 1
python3 tools/pmc_port.py ${tag} ${tag}_pmc_valu_port.json > /dev/null || exit 1
tools/pmc_wait.sh ${tag} > gpurun_out/${tag}_pmc_wait.txt 2>&1 || exit 1
fi
if [[ $part == *c* ]]; then
run gpurun_out/${tag}_bench_set_p_apogee.json python3 bench.py --workload set_p_apogee --steps 6 --warmup 3 $S
run gpurun_out/${tag}_bench_set_p_apogee_gate.json python3 bench.py --workload set_p_apogee --steps 2 --warmup 1 $S --no-second-leg --precision f64
run gpurun_out/${tag}_bench_set_p_full.json python3 bench.py --workload set_p_full --steps 6 --warmup 3 $S
run gpurun_out/${tag}_bench_csv_chute.json python3 bench.py --workload csv_chute --steps 6 --warmup 3 $S
run gpurun_out/${tag}_bench_set_s_1m.json python3 bench.py --samples-per-gpu 1048576 --steps 6 --warmup 3 $S
run gpurun_out/${tag}_bench_cfg5_share.json python3 bench.py --workload csv_chute --samples-per-gpu 1250000 --steps 4 --warmup 2 $S
fi
echo refreshed
