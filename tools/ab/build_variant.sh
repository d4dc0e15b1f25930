#!/bin/bash
# usage: build_variant.sh <tag> [extra flags for the fp64 throughput unit]  ->  tools/ab/liberpl_mc_<tag>.so
# (A/B experiment builds of the library; loaded through ERPL_LIB, never shipped)
set -e
cd "$(dirname "$0")/../../erpl_monte_carlo_sim_amd/csrc"
T=$1; shift
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../../include -ffp-contract=fast -fno-slp-vectorize"
SCHED="-mllvm -amdgpu-sched-strategy=iterative-maxocc"
for a in "$@"; do [ "$a" = "-nosched" ] && SCHED=""; done
ARGS=(); for a in "$@"; do [ "$a" != "-nosched" ] && ARGS+=("$a"); done
/opt/rocm/bin/hipcc $FL $SCHED "${ARGS[@]}" -c erpl_k64f.hip -o /tmp/k64f_$T.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/liberpl_mc_$T.so erpl_k64.o erpl_k32.o /tmp/k64f_$T.o erpl_api.o -lpthread
echo built tools/ab/liberpl_mc_$T.so
