#!/bin/bash
# Evidence for profiles/ (bench lines, kernel traces, counters): run on the GPU box from the repository root.
#   tools/collect_profiles.sh TAG counters WORKLOAD   -> counters and the kernels-alone trace of one workload (copied to profiles/ on the box)
#   tools/collect_profiles.sh TAG lines               -> the default-command trace, the driver's command and every workload's bench line
# Order: counters first (for every workload), then the lines, so that the bench lines quote the counters of the very build they
# measure (bench.py reads them from profiles/).  Results: gpurun_out/profiles_TAG/; tools/adopt_profiles.py TAG copies them to profiles/.
set -u
TAG=${1:-r3}
WHAT=${2:-lines}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
if [ "$WHAT" = counters ]; then
  WL=${3:-bunny_1080p}
  # 1. hardware counters, one pass per rocprofv3 run
  bash tools/pmc_passes.sh gpurun_out/profiles_$TAG/pmc_$WL $WL || exit 1
  python3 tools/pmc_summary.py gpurun_out/profiles_$TAG/pmc_$WL/* > "$OUT/pmc_$WL.txt"
  cp "$OUT/pmc_$WL.txt" "$ROOT/profiles/${TAG}_pmc_$WL.txt"
  rm -rf "$OUT/pmc_$WL" # (the raw counter tables are large; what comes back from the box is bounded)
  cd /tmp && export TMPDIR=/tmp
  export GPU_MAX_HW_QUEUES=16
  # 2. kernel trace with every kernel alone on the GPU and one kernel per launch: these averages are the ones the roofline's
  # avg_launch_ms must match (RTAMD_FULL_WAVES=1: bounce launches in full waves as the default command runs them; a scene
  # rendered on one stream only would spread its bounce rays over narrower waves.  RTAMD_FUSE=0: the default command pairs the
  # shadow rays of a level with the closest hits of the next in one launch, which the event pass of bench.py takes apart)
  STEPS=$([ $WL = field_mirrors_8k ] && echo "--steps 10 --warmup 3 --repeats 3" || echo "")
  RTAMD_FULL_WAVES=1 RTAMD_FUSE=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_serial_$WL" -- \
    python3 "$ROOT/bench.py" --cpu-seconds 0 --in-flight 1 --workload $WL $STEPS > "$OUT/bench_under_rocprof_serial_$WL.json" 2> "$OUT/trace_serial_$WL.log" || exit 1
  cp "$(find "$OUT/trace_serial_$WL" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_${WL}_serial.csv"
  cp "$OUT/kernel_stats_${WL}_serial.csv" "$ROOT/profiles/${TAG}_kernel_stats_${WL}_serial.csv"
  rm -rf "$OUT/trace_serial_$WL"
  exit 0
fi
cd /tmp && export TMPDIR=/tmp
# (the profiler starts the HIP runtime before bench.py can ask for its hardware queues, so the shell does)
export GPU_MAX_HW_QUEUES=16
# 3. kernel trace of the default command (four frames in flight: kernels of different frames overlap)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_default" -- \
  python3 "$ROOT/bench.py" --cpu-seconds 0 > "$OUT/bench_under_rocprof_default_bunny_1080p.json" 2> "$OUT/trace_default.log" || exit 1
cp "$(find "$OUT/trace_default" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_bunny_1080p_default.csv"
rm -rf "$OUT/trace_default"
# 4. the driver's command, the default bench line, and the other single-GPU workloads
cd "$ROOT"
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > "$OUT/bench_bunny_1080p_steps20.json" || exit 1
for wl in bunny_1080p bunny_1000 bunny_4k bunny_sub2_4k field_4k field_mirrors_8k; do
  timeout -k 10 400 python3 bench.py --workload $wl $([ $wl = bunny_1080p ] || echo --cpu-seconds 0) 2>/dev/null | tail -1 > "$OUT/bench_$wl.json" || exit 1
done
ls "$OUT"
