#!/bin/bash
# Registers and occupancy of the main kernels for a set of compiler flags: tools/kres.sh "-DRT_X=1 ..." [name filter]
# (writes the assembly to /tmp/kres.s)
cd "$(dirname "$0")/../raytracer_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize --offload-arch=gfx950 $1 --cuda-device-only -S kernels.hip -o /tmp/kres.s -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name:/{name=$5} / VGPRs:/{v=$4} /ScratchSize/{sc=$5} /Occupancy/{print name, "VGPRs", v, "scratch", sc, "waves", $5}' |
  grep -E "${2:-shadow_grid_kernelILb0|shadow_walk_kernelILb0|walk_steal_kernelILi1ELb0|face_raster_kernelILb0|compact_kernelILb1ELi2|shade_kernelILb1}" | sed 's/_ZN5rtamd12_GLOBAL__N_1[0-9]*//; s/EEvNS_[A-Za-z_0-9]*//'
