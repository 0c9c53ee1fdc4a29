#!/bin/bash
# Frames in flight against hardware queues: RTAMD_LIB must allow that many frame slots (-DRT_MAX_FRAME_SLOTS=8).
for wl in ${WORKLOADS:-bunny_1000 bunny_1080p field_4k}; do
  for q in ${QUEUES:-4 8}; do
    for f in ${IN_FLIGHT:-4 6 8}; do
      out=$(GPU_MAX_HW_QUEUES=$q RTAMD_BENCH_MAX_IN_FLIGHT=16 timeout -k 10 120 python3 bench.py --steps 96 --warmup 16 --cpu-seconds 0 --no-kernel-events --in-flight $f --workload $wl 2>/dev/null | tail -1) || exit 1
      echo "$wl queues $q in-flight $f $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("Mrays/s %.0f  ms %.4f" % (d["value"], d["ms_per_step"]))')"
    done
  done
done
