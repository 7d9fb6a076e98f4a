#!/bin/bash
# A/B builds: tools/build_variant.sh NAME "-DFLAG ..."  ->  rogue-gym_amd/variants/librogue_NAME.so (select with ROGUE_GYM_HIP_LIB=...).
# The variant .so files are git-ignored and travel to the GPU box with the snapshot, so one gpurun call can compare several builds on one box.
set -e
name=$1; shift
cd "$(dirname "$0")/../rogue-gym_amd/csrc"
B=../build_$name; mkdir -p $B ../variants
F="--offload-arch=gfx950 -std=c++17 -fPIC -Wall -Wno-unused-function -DRG_BUILD_ID=\"variant-$name\" $*"
hipcc $F -O3 -c rg_kernels.hip -o $B/rg_kernels.o &
hipcc $F -Os -c rg_obs.hip -o $B/rg_obs.o &
hipcc $F -O3 -c rg_regen_lanes.hip -o $B/rg_regen_lanes.o &
hipcc $F -O2 -c rg_api.cpp -o $B/rg_api.o &
hipcc $F -O2 -c rg_config.cpp -o $B/rg_config.o &
hipcc $F -O2 -c rg_items.cpp -o $B/rg_items.o &
wait
hipcc --offload-arch=gfx950 -shared $B/rg_kernels.o $B/rg_regen_lanes.o $B/rg_obs.o $B/rg_api.o $B/rg_config.o $B/rg_items.o -o ../variants/librogue_$name.so
echo "built variants/librogue_$name.so"
