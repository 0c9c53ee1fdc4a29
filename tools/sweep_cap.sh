#!/bin/bash
# Grid-size cap of the walk kernels (experiment hook RTAMD_WALK_MAX_BLOCKS): frame time per workload.
out=gpurun_out/sweep_cap.txt
: > $out
for wl in bunny_1080p field_4k; do
  for cap in 8192 32768 65536 131072 262144 1048576; do
    echo "== $wl cap=$cap" >> $out
    RTAMD_WALK_MAX_BLOCKS=$cap timeout -k 10 120 python -u bench.py --workload $wl --steps 50 --warmup 5 --cpu-seconds 0 2>/dev/null \
      | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['kernels'])" >> $out || exit 1
  done
done
