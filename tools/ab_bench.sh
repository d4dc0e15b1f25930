#!/bin/bash
# A/B builds of liberpl_mc.so with bench.py: tools/ab_bench.sh "<bench args>" lib1.so lib2.so ...
args="$1"; shift
for lib in "$@"; do
  cp "$lib" erpl_monte_carlo_sim_amd/csrc/liberpl_mc.so
  timeout -k 10 300 python bench.py $args --cpu-seconds 0 --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', 'traj/s', round(d['value']), 'steps/s %.3g'%d['trajectory_steps_per_s'], 'util', round(d['lane_utilisation'],3), 'TF', round(d['roofline']['achieved'],2), 'ms', round(d['ms_per_step'],1))"
done
