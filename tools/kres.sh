#!/bin/bash
# Compile one translation unit of the library and print the register / scratch figures of its kernels (no GPU needed).
# Usage: tools/kres.sh [rg_kernels.hip] [-O3] [mangled-name-pattern, e.g. k_step_w32]
cd "$(dirname "$0")/../rogue-gym_amd/csrc"
src=${1:-rg_kernels.hip}; opt=${2:--O3}; pat=${3:-k_step_w32}
hipcc --offload-arch=gfx950 -std=c++17 -fPIC -Wall -Wno-unused-function -DRG_BUILD_ID='"x"' $opt -c $src -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage > /tmp/kres.txt 2>&1
grep -E "error|warning" /tmp/kres.txt | head
grep -A10 "Function Name: .*$pat" /tmp/kres.txt | grep -E "Function Name|VGPRs:|ScratchSize|VGPRs Spill|Occupancy" | sed 's/.*remark: *//; s/ \[-Rpass.*//'
