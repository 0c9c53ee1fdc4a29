#!/bin/bash
# Evidence for profiles/ (bench lines, kernel traces, counters): run on the GPU box from the repository root.
#   tools/collect_profiles.sh TAG      -> gpurun_out/profiles_TAG/
# Order: counters and the kernels-alone trace first, copied to profiles/ on the box, so that the bench lines taken
# afterwards quote the counters of the very build they measure (bench.py reads them from profiles/).
set -u
TAG=${1:-r2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
# 1. hardware counters, one pass per rocprofv3 run
bash tools/pmc_passes.sh gpurun_out/profiles_$TAG/pmc bunny_1080p || exit 1
python3 tools/pmc_summary.py gpurun_out/profiles_$TAG/pmc/* > "$OUT/pmc_bunny_1080p.txt"
cp "$OUT/pmc_bunny_1080p.txt" "$ROOT/profiles/${TAG}_pmc_bunny_1080p.txt"
cd /tmp && export TMPDIR=/tmp
# (the profiler starts the HIP runtime before bench.py can ask for its hardware queues, so the shell does)
export GPU_MAX_HW_QUEUES=16
# 2. kernel trace with every kernel alone on the GPU and one kernel per launch: these averages are the ones the roofline's
# avg_launch_ms must match (RTAMD_FULL_WAVES=1: bounce launches in full waves as the default command runs them; a scene
# rendered on one stream only would spread its bounce rays over narrower waves.  RTAMD_FUSE=0: the default command pairs the
# shadow rays of a level with the closest hits of the next in one launch, which the event pass of bench.py takes apart)
RTAMD_FULL_WAVES=1 RTAMD_FUSE=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_serial" -- \
  python3 "$ROOT/bench.py" --cpu-seconds 0 --in-flight 1 > "$OUT/bench_under_rocprof_serial.json" 2> "$OUT/trace_serial.log" || exit 1
cp "$(find "$OUT/trace_serial" -name '*kernel_stats.csv' | head -1)" "$ROOT/profiles/${TAG}_kernel_stats_bunny_1080p_serial.csv"
# 3. kernel trace of the default command (four frames in flight: kernels of different frames overlap)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_default" -- \
  python3 "$ROOT/bench.py" --cpu-seconds 0 > "$OUT/bench_under_rocprof_default.json" 2> "$OUT/trace_default.log" || exit 1
# 4. the driver's command, the default bench line, and the other single-GPU workloads
cd "$ROOT"
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > "$OUT/bench_bunny_1080p_steps20.json" || exit 1
for wl in bunny_1080p bunny_1000 bunny_4k bunny_sub2_4k field_4k field_mirrors_8k; do
  timeout -k 10 300 python3 bench.py --workload $wl $([ $wl = bunny_1080p ] || echo --cpu-seconds 0) 2>/dev/null | tail -1 > "$OUT/bench_$wl.json" || exit 1
done
find "$OUT" -name "*kernel_stats.csv"
