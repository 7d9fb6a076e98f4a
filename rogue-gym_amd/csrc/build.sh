#!/bin/bash
# Builds librogue_gym_hip.so for gfx950 in-tree (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../librogue_gym_hip.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
    rg_kernels.hip rg_api.cpp rg_config.cpp -o "$OUT"
echo "built $OUT"
