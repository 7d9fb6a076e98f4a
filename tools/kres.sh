#!/bin/bash
# Compile one translation unit of the library and print the register / scratch figures of its kernels (no GPU needed).
# Usage: tools/kres.sh [rg_kernels.hip] [-O3] [kernel-name-pattern] [extra flags]
cd "$(dirname "$0")/../rogue-gym_amd/csrc"
src=${1:-rg_kernels.hip}; opt=${2:--O3}; pat=${3:-k_step_w32}; shift 3
hipcc --offload-arch=gfx950 -std=c++17 -fPIC -Wall -Wno-unused-function -DRG_BUILD_ID='"x"' $opt "$@" -c $src -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name:/ {if (line) print line; line=$NF} /(VGPRs|ScratchSize|Occupancy|VGPRs Spill|SGPRs Spill).*:/ {sub(/.*remark: [^ ]* +/, ""); sub(/ \[-Rpass.*/, ""); line=line " | " $0} END {print line}' | grep -E "$pat"
