#!/bin/bash
# Occupancy sensitivity: extra LDS per 1-wave workgroup (8 KiB stack + pad) -> resident waves per CU = 160 KiB / (8 KiB + pad)
for wl in ${WORKLOADS:-bunny_1000}; do
  for pad in 0 8192 24576 57344; do
      out=$(RTAMD_WALK_LDS_PAD=$pad timeout -k 10 120 python3 bench.py --steps 30 --warmup 5 --cpu-seconds 0 --workload $wl 2>/dev/null | tail -1)
      echo "$wl pad=$pad waves/CU=$((163840/(8192+pad))) $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels"]; print("Mrays/s %.0f  ms %.3f  trace %.3f shadow %.3f" % (d["value"], d["ms_per_step"], k["trace_ms_per_frame"], k["shadow_ms_per_frame"]))')"
  done
done
