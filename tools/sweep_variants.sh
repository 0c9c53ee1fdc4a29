#!/bin/bash
# A/B of kernel build variants raytracer_amd/lib/variants/librtamd_<name>.so (see tools/build_variants.sh)
for wl in ${WORKLOADS:-bunny_1000 field_4k}; do
  for lib in raytracer_amd/lib/variants/*.so; do
      out=$(RTAMD_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --steps 60 --warmup 10 --cpu-seconds 0 --in-flight ${IN_FLIGHT:-2} --workload $wl 2>/dev/null | tail -1)
      echo "$wl $(basename $lib) $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels"]; print("Mrays/s %.0f  ms %.3f  trace %.3f raster %.3f shadow %.3f compact %.3f shade %.3f" % (d["value"], d["ms_per_step"], k["trace_ms_per_frame"], k["raster_ms_per_frame"], k["shadow_ms_per_frame"], k["compact_ms_per_frame"], k["shade_ms_per_frame"]))')"
  done
done
