#!/bin/bash
# A variant of the library for same-box A/B runs and debug counters: bash tools/build_variant.sh <name> <extra hipcc flags...>
# -> _ab/lib<name>.so (git-ignored; select it with PT_HIP_LIBRARY=$PWD/_ab/lib<name>.so)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../pathtracer_amd/csrc"
mkdir -p ../../_ab /tmp/ab
make -s pt_plugin.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -fno-slp-vectorize "$@" -c pt_kernels.hip -o /tmp/ab/k_$NAME.o
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,nodelete -o ../../_ab/lib$NAME.so /tmp/ab/k_$NAME.o pt_plugin.o -ldl
echo built _ab/lib$NAME.so
