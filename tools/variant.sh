#!/bin/bash
# Developer tool: builds the CURRENT tree with extra flags into build/variants/lib_<name>.so (A/B on one GPU box with tools/ab.sh; name it
# build/lib_<name>.so in the tools/gpu.sh command: only the variants a command names travel to the box).
# usage: tools/variant.sh <name> [-DRR_SHADOW_WAVES=3 ...]
set -e
name=$1; shift
cd "$(dirname "$0")/../rustray_amd/csrc"
mkdir -p ../../build/variants
hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -shared -o ../../build/variants/lib_$name.so rr_api.hip rr_bvh.cpp
echo "built build/variants/lib_$name.so ($*)"
