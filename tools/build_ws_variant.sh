#!/bin/bash
# build tools/ubench/libearl_ws_<tag>.so = the shipped library with tabletop.hip recompiled under extra -D flags (tuning experiments of the
# fused tabletop rollout; see the macro list at the head of csrc/tabletop_rollout_ws.h).  usage: build_ws_variant.sh <tag> [-DFLAG ...]
set -e
cd "$(dirname "$0")/../earl_benchmark_amd/csrc"
tag=$1; shift
FLAGS="-DEARL_WS_EXPERIMENTS --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fPIC"
mkdir -p ../../tools/ubench
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o ../../tools/ubench/tabletop_$tag.o tabletop.hip
/opt/rocm/bin/hipcc $FLAGS -shared -o ../../tools/ubench/libearl_ws_$tag.so ../../tools/ubench/tabletop_$tag.o glue.o physics.o physics_w8.o physics_mt.o physics_l64.o physics_kitchen.o
rm -f ../../tools/ubench/tabletop_$tag.o
echo built libearl_ws_$tag.so
