#!/bin/bash
# build tools/ubench/libearl_phys_<tag>.so = the shipped library with the five articulated-body units (physics*.hip) recompiled under extra -D flags
# (e.g. -DEARL_PHYS_NO_CONTRACT: no floating-point contraction, the measurement DESIGN.md "contraction" quotes; -DEARL_MT_BLOCKS=2).  usage: build_phys_variant.sh <tag> [-DFLAG ...]
set -e
cd "$(dirname "$0")/../earl_benchmark_amd/csrc"
tag=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fPIC"
U=../../tools/ubench
mkdir -p $U
for u in physics physics_w8 physics_mt physics_l64 physics_kitchen; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c -o $U/${u}_$tag.o $u.hip &
done
wait
/opt/rocm/bin/hipcc $FLAGS -shared -o $U/libearl_phys_$tag.so tabletop.o glue.o $U/physics_$tag.o $U/physics_w8_$tag.o $U/physics_mt_$tag.o $U/physics_l64_$tag.o $U/physics_kitchen_$tag.o
rm -f $U/physics*_$tag.o
echo built libearl_phys_$tag.so
