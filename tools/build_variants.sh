#!/bin/bash
# Build kernel variants for A/B runs: tools/build_variants.sh NAME "-DRT_X=1 -DRT_Y=2" [NAME FLAGS ...]
# (SLP="" in the environment builds with the SLP vectoriser, which the Makefile turns off for kernels.hip)
# -> raytracer_amd/lib/variants/librtamd_NAME.so (run with RTAMD_LIB=..., see tools/sweep_variants.sh)
set -e
cd "$(dirname "$0")/../raytracer_amd/csrc"
mkdir -p ../lib/variants build/var
COMMON="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math ${SLP--fno-slp-vectorize}"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc $COMMON --offload-arch=gfx950 $flags -c kernels.hip -o build/var/kernels_$name.o \
      -Rpass-analysis=kernel-resource-usage 2> build/var/res_$name.txt;
    g++ -shared -fPIC -pthread -o ../lib/variants/librtamd_$name.so build/var/kernels_$name.o build/lbvh.o build/grid_build.o build/api.o build/bvh.o build/shadow_grid.o build/flyscene_host.o ) &
done
wait
ls ../lib/variants
