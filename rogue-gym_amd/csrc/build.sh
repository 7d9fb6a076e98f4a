#!/bin/bash
# Builds librogue_gym_hip.so for gfx950 in-tree (cross-compiles without a GPU).
# The step kernels want -O3; the bandwidth-bound render/observation kernels are faster with -Os (less unrolling).
set -e
cd "$(dirname "$0")"
OUT=../librogue_gym_hip.so
ID=$(python3 -c "import sys; sys.path.insert(0, '../..'); import __graft_entry__ as g; print(g.source_id())")   # compiled in as rg_build_id()
F="--offload-arch=gfx950 -std=c++17 -fPIC -Wall -Wno-unused-function -DRG_BUILD_ID=\"$ID\""
mkdir -p ../build
hipcc $F -O3 -c rg_kernels.hip -o ../build/rg_kernels.o
hipcc $F -Os -c rg_obs.hip -o ../build/rg_obs.o
hipcc $F -O3 -c rg_regen_lanes.hip -o ../build/rg_regen_lanes.o
hipcc $F -O2 -c rg_api.cpp -o ../build/rg_api.o
hipcc $F -O2 -c rg_config.cpp -o ../build/rg_config.o
hipcc $F -O2 -c rg_items.cpp -o ../build/rg_items.o
hipcc --offload-arch=gfx950 -shared ../build/rg_kernels.o ../build/rg_regen_lanes.o ../build/rg_obs.o ../build/rg_api.o ../build/rg_config.o ../build/rg_items.o -o "$OUT"
echo "built $OUT"
