#!/bin/bash
# A/B different builds of liberpl_mc.so on the same box: tools/ab_libs.sh "<diag args>" lib1.so lib2.so ...
args="$1"; shift
for lib in "$@"; do
  cp "$lib" erpl_monte_carlo_sim_amd/csrc/liberpl_mc.so
  echo "== $lib $args"
  timeout -k 10 200 python tools/diag_steps.py $args 2>&1 | grep -E "^chunk="
done
