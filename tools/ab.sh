#!/bin/bash
# A/B of library builds or environment switches on the GPU box: prints ms_per_step and the per-kernel times of bench.py.
# usage: tools/ab.sh OUT "ENV1" "ENV2" ... [-- bench args]   e.g. tools/ab.sh gpurun_out/ab "RTAMD_SHADOW_POOL=0" "RTAMD_SHADOW_POOL=1"
OUT=$1; shift
mkdir -p "$OUT"
ARGS="--steps 100 --warmup 20 --repeats 7 --cpu-seconds 0"
ENVS=()
while [ $# -gt 0 ]; do
  if [ "$1" = "--" ]; then shift; ARGS="$ARGS $*"; break; fi
  ENVS+=("$1"); shift
done
i=0
for e in "${ENVS[@]}"; do
  i=$((i+1))
  env $e timeout -k 10 200 python3 bench.py $ARGS > "$OUT/run$i.json" 2> "$OUT/run$i.err" || { echo "run $i ($e) failed"; tail -5 "$OUT/run$i.err"; exit 1; }
  python3 - "$OUT/run$i.json" "$e" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("kernels", {})
print("%-40s ms/step %.4f (min %.4f) single %s | trace %.4f shadow %.4f raster %.4f compact %.4f shade %.4f frame_ev %.4f" % (
    sys.argv[2] or "(default)", d["ms_per_step"], d["ms_per_step_min"], d.get("frame_ms_single"), k.get("trace_ms_per_frame", 0), k.get("shadow_ms_per_frame", 0),
    k.get("raster_ms_per_frame", 0), k.get("compact_ms_per_frame", 0), k.get("shade_ms_per_frame", 0), k.get("frame_ms_events", 0)))
PY
done
