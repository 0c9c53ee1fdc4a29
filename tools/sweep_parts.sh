#!/bin/bash
# One rank's share of a frame (tools/inflight.py) for every library under raytracer_amd/lib/variants
export GPU_MAX_HW_QUEUES=16
for lib in raytracer_amd/lib/variants/*.so; do
  echo "== $(basename $lib)"
  RTAMD_LIB=$PWD/$lib timeout -k 10 120 python3 tools/inflight.py ${1:-field_4k} ${2:-8} ${3:-1,8} 2>/dev/null || exit 1
done
