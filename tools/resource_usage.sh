#!/bin/bash
# Register / scratch / LDS table of every kernel of the three builds, as the compiler reports it
# (-Rpass-analysis=kernel-resource-usage, same flags as csrc/Makefile).  No GPU needed.
#   tools/resource_usage.sh > profiles/r3_kernel_resource_usage.txt
cd "$(dirname "$0")/../erpl_monte_carlo_sim_amd/csrc"
C="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../../include -S --cuda-device-only -Rpass-analysis=kernel-resource-usage -o /dev/null"
echo "# hipcc -Rpass-analysis=kernel-resource-usage, gfx950, flags of csrc/Makefile ($(/opt/rocm/bin/hipcc --version | grep -m1 -o 'HIP version.*'))"
echo "# template arguments of erpl_flight_*: <trajectory capture, specialisation (bit 0 wind table, bit 1 solid motor; n1 = run time), min waves per SIMD>"
printf "%-46s %5s %5s %5s %8s %6s %10s %10s %8s\n" kernel SGPR VGPR AGPR scratchB waves sgpr_spill vgpr_spill LDS_B
for u in "k64f -ffp-contract=fast -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-maxocc" "k32 -ffp-contract=fast -fno-slp-vectorize" "k64 -ffp-contract=off -mllvm -amdgpu-sched-strategy=max-ilp"; do
  set -- $u; unit=$1; shift
  /opt/rocm/bin/hipcc $C "$@" erpl_$unit.hip 2>&1 | python3 -c '
import re,sys
t=sys.stdin.read()
for blk in t.split("Function Name: ")[1:]:
    name=blk.split()[0]
    g=lambda k: re.search(k+r": (\d+)",blk).group(1)
    short=re.sub(r"_ZN12_GLOBAL__N_1\d+","",name)
    short=re.sub(r"E?v?9ErplKArgs.*$","",short)
    short=re.sub(r"ILb(\d)ELi(n?\d)ELi(\d)E+$",lambda m:"<%s,%s,%s>"%(m.group(1),m.group(2),m.group(3)),short)
    print("%-46s %5s %5s %5s %8s %6s %10s %10s %8s"%(short,g("TotalSGPRs"),g("VGPRs"),g("AGPRs"),g(r"ScratchSize \[bytes/lane\]"),g(r"Occupancy \[waves/SIMD\]"),g("SGPRs Spill"),g("VGPRs Spill"),g(r"LDS Size \[bytes/block\]")))
'
done
