#!/bin/bash
# Like tools/ab.sh but varies bench.py arguments instead of the environment.
# usage: tools/ab_args.sh OUT "ARGS1" "ARGS2" ...
OUT=$1; shift
mkdir -p "$OUT"
i=0
for a in "$@"; do
  i=$((i+1))
  timeout -k 10 200 python3 bench.py --steps 100 --warmup 20 --repeats 7 --cpu-seconds 0 $a > "$OUT/run$i.json" 2> "$OUT/run$i.err" || { echo "run $i ($a) failed"; tail -5 "$OUT/run$i.err"; exit 1; }
  python3 - "$OUT/run$i.json" "$a" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("kernels", {})
r = d["config"]["rays_per_frame"]
print("%-44s ms/step %.4f (min %.4f) single %s | trace %.4f shadow %.4f raster %.4f compact %.4f shade %.4f frame_ev %.4f | rays %d/%d/%d" % (
    sys.argv[2] or "(default)", d["ms_per_step"], d["ms_per_step_min"], d.get("frame_ms_single"), k.get("trace_ms_per_frame", 0), k.get("shadow_ms_per_frame", 0),
    k.get("raster_ms_per_frame", 0), k.get("compact_ms_per_frame", 0), k.get("shade_ms_per_frame", 0), k.get("frame_ms_events", 0), r["primary"], r["shadow"], r["bounce"]))
PY
done
