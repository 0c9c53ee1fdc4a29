#!/usr/bin/env python3
"""Benchmark of the render pass: Mrays/s (primary + shadow + reflection) and frame time.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

One step = one frame: every launch of Flyscene::raytraceScene's replacement (trace + shade per
recursion level) over the whole image, scene and output buffers resident in HBM.  With N > 1 (one
process per GPU, started by torch.distributed.run) the frame is cut into row tiles dealt round-robin to
the ranks, each rank renders its tiles from its own copy of the scene, and the step ends with the one
gather of the tiles on rank 0 (RCCL).  The total work per step is fixed, so scaling is "strong".

Rank 0 prints one JSON line.  `roofline` describes the kernel that takes most of the frame; its
duration comes from HIP events recorded on the render stream around every launch of a pass of K frames
that follows the timed region (one frame at a time, one kernel per launch: in the timed region several
frames overlap and independent kernels share launches), its algorithmic work from an instrumented
(untimed) frame that counts slab and triangle tests.
`cpu_baseline` times the CPU oracle (the restatement of the reference algorithm, compiled with the
reference's Release flags) on a bounded sample of the same frame's rows on this machine's host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _hardware_queues():
    """Frames in flight overlap only while each of their streams has a hardware queue of its own (and streams that
    wait for each other, as the frames and the collective do, must not outnumber the queues at all); the HIP runtime
    reads GPU_MAX_HW_QUEUES (four by default) once, at its first call: set it before torch loads the runtime."""
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--hw-queues", type=int, default=16)
    queues = pre.parse_known_args()[0].hw_queues
    if queues > 0:
        os.environ["GPU_MAX_HW_QUEUES"] = str(min(queues, 16))


_hardware_queues()

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import raytracer_amd as R  # noqa: E402
from raytracer_amd import _lib as L  # noqa: E402
from raytracer_amd import tiles, workloads  # noqa: E402

# Peaks from the MI355X microarchitecture guide: FP32 vector 157.3 TFLOP/s = 256 CU x 4 SIMD x 32 lanes x 2 flops
# (one FMA) x 2.4 GHz; HBM3E 8 TB/s.  A wave64 VALU instruction takes a SIMD-32 two cycles, so the chip issues at most
# 256 x 4 x 2.4e9 / 2 = 1228.8 G wave-level VALU instructions per second.  What the path's small scene is really served
# from: the XCDs' L2 (34.5 TB/s aggregate) and, for records gathered at random, the Infinity Cache (8.6 TB/s measured
# for random rows of a 38 MB table).
PEAK_FP32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
PEAK_VALU_WAVE_INSTS_PER_S = 256 * 4 * 2.4e9 / 2
PEAK_L2_GBS = 34500.0
PEAK_INFINITY_CACHE_GATHER_GBS = 8600.0

# Algorithmic work per test: the reference's operation counts (one ray/AABB slab test,
# src/boundingBox.h:257-286: 6 sub, 6 div, 3 swap-compares, 10 min/max, 2 compares = 27; plane distance,
# src/flyscene.cpp:758-760: 14; hit point + three edge functions, :771-780: 69; Phong per light ~150)
# and bytes (one AABB 32 B, plane n,D 16 B, vertex ids + vertices 48 B).
FLOPS_BOX, FLOPS_PLANE, FLOPS_EDGE, FLOPS_SHADE = 27, 14, 69, 150
BYTES_BOX, BYTES_PLANE, BYTES_EDGE = 32, 16, 48


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=25,
                    help="the timed region of K steps is run this many times, each bracketed by barrier + synchronize; the line "
                         "reports the median repeat as ms_per_step / value and the extremes beside it")
    ap.add_argument("--workload", default=None, help="default: bunny_1080p (1 GPU) / field_4k (several GPUs)")
    ap.add_argument("--size", default=None, help="WxH: render the workload at another image size (experiments; the line names it)")
    ap.add_argument("--soft-points", type=int, default=None, help="experiments: soft-shadow samples per light instead of the workload's (the line names it)")
    ap.add_argument("--max-reflections", type=int, default=None, help="experiments: bounce depth instead of the workload's (the line names it)")
    ap.add_argument("--row-tile", type=int, default=0,
                    help="rows per tile of the partition over GPUs; 0 = about six tiles per GPU (height / (6 N), at least 16): coarse enough "
                         "that a rank drops most faces of other ranks' rows unseen, fine enough to balance the ranks (1.1 M-face field at "
                         "4K, slowest of eight ranks: 0.052 ms per frame with tiles of 45 rows, 0.052 with 16, 0.060 with 135; four ranks: "
                         "0.090 with 90 rows, 0.096 with 16)")
    ap.add_argument("--max-leaf", type=int, default=0, help="BVH: faces per leaf (0 = library default)")
    ap.add_argument("--sah-bins", type=int, default=0, help="BVH: SAH bins (0 = library default)")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="frames enqueued at a time, each on a stream of its own (1..32; default: 4 on one GPU -- the GPU runs four "
                         "kernels at a time, one per hardware pipe, and eight or sixteen frames in flight are no faster in the steady "
                         "state and slower over a short batch, on one GPU and for one rank's share of a frame alike; fewer when the "
                         "frames' workspaces would take more than 64 GB)")
    ap.add_argument("--workspace-limit-mb", type=int, default=-1,
                    help="experiments: bound on one frame's work buffers, beyond which the library renders the frame in bands of rows "
                         "(rtamd_scene_set_workspace_limit; 0 = no bound, default: the library's 2048)")
    ap.add_argument("--pipeline-depth", type=int, default=8,
                    help="several GPUs: frame buffers of the gather pipeline (at least the frames in flight); the gather of a frame may "
                         "lag that many frames behind its tracing")
    ap.add_argument("--hw-queues", type=int, default=16,
                    help="hardware queues asked of the HIP runtime (GPU_MAX_HW_QUEUES; 0 = leave the environment alone, and then "
                         "at most 4 frames in flight on one GPU, 2 with several)")
    ap.add_argument("--single-stream", action="store_true",
                    help="accepted for older command lines; frames always run entirely on their caller's stream. With --in-flight 1 (and "
                         "RTAMD_FUSE=0 for one kernel per launch) every kernel runs alone, the mode the per-kernel profile under profiles/ is taken in")
    ap.add_argument("--force-pipeline", action="store_true",
                    help="run the several-GPU code path (gather pipeline, both payloads, single-GPU comparison) on one GPU: a rehearsal "
                         "of everything but the collective itself")
    ap.add_argument("--gather", choices=("u8", "f32"), default="f32",
                    help="what the ranks send to rank 0: the float framebuffer (pixel_data, what the 1e-3 parity bar is stated on), or "
                         "the 8-bit pixels the reference writes to result.ppm (a quarter of the bytes; timed as other_gather)")
    ap.add_argument("--exchange", choices=("all_to_all", "gather", "rotate"), default="all_to_all",
                    help="several GPUs: how a frame's row tiles come together.  all_to_all: frames in batches of N, frame k whole on rank "
                         "k mod N, ONE all-to-all per batch (every rank's shares leave over all its xGMI links at once); gather: one gather "
                         "per frame to rank 0 (every share crosses one link, one frame after the other); rotate: one gather per frame to rank k mod N")
    ap.add_argument("--builder", type=int, default=0, help="0 = host SAH builder, 1 = on-GPU Morton-order builder")
    ap.add_argument("--shadow-rays", type=int, default=0, help="0 = candidate grids when possible, 1 = shadow rays walk the BVH")
    ap.add_argument("--primary-rays", type=int, default=0, help="0 = screen binning, 1 = primary rays walk the BVH")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample; 0 = skip")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket launches with HIP events in the timed region")
    return ap.parse_args()


def cpu_baseline(w, cam, budget_s):
    """Time the CPU oracle (reference algorithm, reference Release flags) on every k-th row."""
    from oracle.oracle_py import Camera, OracleScene  # test infrastructure; CPU leg only

    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    arrays = workloads.mesh_arrays(w)
    lights = [tuple(l) + (1.0, 1.0, 1.0) for l in w["lights"]]
    orc = OracleScene(arrays=arrays, lights=lights, hard_shadow=True, soft_points=w["soft_points"], fast=True)
    oc = Camera()
    oc.origin[:] = list(cam.origin)
    oc.inv_view[:] = list(cam.inv_view)
    oc.viewport[:] = list(cam.viewport)
    oc.aspect_ratio, oc.fovy = cam.aspect_ratio, cam.fovy
    W, H = w["width"], w["height"]
    kw = dict(max_reflections=w["max_reflections"], num_threads=threads)
    # calibrate on every 64th row, then pick the stride that fills the budget
    probe = orc.render(W, H, camera=oc, planes=False, row_offset=H // 128, row_step=64, **kw)["counters"]
    rate = probe.rays / max(probe.seconds, 1e-9)
    full_rays_est = probe.rays * 64.0
    step = max(1, int(np.ceil(full_rays_est / max(rate * budget_s, 1.0))))
    # a frame shorter than the budget is repeated so that the sample still holds several seconds of CPU work
    # (until the budget is used: the probe's few rows spread badly over the threads and underestimate the rate)
    rays = seconds = box = tri = 0.0
    reps = 0
    while reps == 0 or (step == 1 and seconds < 0.9 * budget_s and reps < 2000):
        res = orc.render(W, H, camera=oc, planes=False, row_offset=0, row_step=step, **kw)["counters"]
        rays, seconds, box, tri = rays + res.rays, seconds + res.seconds, box + res.box_tests, tri + res.tri_plane
        reps += 1
    orc.close()

    class res:  # totals over the repetitions
        pass
    res.rays, res.seconds, res.box_tests, res.tri_plane = int(rays), seconds, box, tri
    return {
        "value": round(res.rays / res.seconds / 1e6, 4),
        "unit": "Mrays/s",
        "cores": threads,
        "host_cores": os.cpu_count(),
        "kind": "port",
        "sample": "every %d-th image row of the same frame, %d time(s) (%d rays, %.1f s), oracle built -O3 -ffast-math -march=x86-64-v3, "
                  "rows interleaved over %d threads as the reference does" % (step, reps, res.rays, res.seconds, threads),
        "box_tests_per_ray": round(res.box_tests / max(res.rays, 1), 1),
        "tri_tests_per_ray": round(res.tri_plane / max(res.rays, 1), 1),
    }


def pmc_traffic(workload, kernel_key):
    """HBM-side bytes per launch of one kernel from the committed counter summary of this workload
    (profiles/*_pmc_<workload>.txt, written by tools/pmc_passes.sh + tools/pmc_summary.py): FETCH_SIZE and
    WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests of 16-byte-per-lane loads at
    64 bytes, hence the factor 2 on the read side."""
    import glob
    import re

    best = None
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_%s.txt" % workload))):
        best = path
    if best is None:
        return None, None
    fetch = write = None
    current = None
    for line in open(best):
        if not line.startswith(" "):
            current = line.strip()
            continue
        if current != kernel_key:
            continue
        m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+mean\s+([0-9.]+)", line)
        if m:
            if m.group(1) == "FETCH_SIZE":
                fetch = float(m.group(2))
            else:
                write = float(m.group(2))
    if fetch is None or write is None:
        return None, None
    return (2.0 * fetch + write) * 1024.0, os.path.basename(best)


def profile_table(workload):
    """Per-kernel figures from the committed profiles of this workload, so that the line shows every kernel and not
    only the one `roofline` describes: wave-level VALU instructions and active lanes per VALU instruction
    (SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU), bytes at the L2's memory side per launch
    ((2 * FETCH_SIZE + WRITE_SIZE) KiB) from profiles/*_pmc_<workload>.txt, and with the average launch duration of
    profiles/*_kernel_stats_<workload>_serial.csv (every kernel alone on the GPU) the fabric-side GB/s."""
    import csv
    import glob
    import re

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    pmc = sorted(glob.glob(os.path.join(here, "*_pmc_%s.txt" % workload)))
    if not pmc:
        return None
    counters, current = {}, None
    for line in open(pmc[-1]):
        if not line.startswith(" "):
            current = line.strip()
            counters[current] = {}
            continue
        m = re.match(r"\s+(\S+)\s+mean\s+([0-9.]+)", line)
        if m and current:
            counters[current][m.group(1)] = float(m.group(2))
    avg_us = {}
    stats = sorted(glob.glob(os.path.join(here, "*_kernel_stats_%s_serial.csv" % workload)))
    if stats:
        for row in csv.DictReader(open(stats[-1])):
            m = re.search(r"(\w+_kernel(?:<[^>]*>)?)\(", row["Name"])
            if m:
                avg_us[m.group(1)] = float(row["AverageNs"]) / 1e3
    table = {}
    for kern, c in sorted(counters.items()):
        if kern.replace(" ", "").endswith("true>") and not kern.startswith(("compact", "shade")):
            continue  # the instrumented (work-counting) variants are not part of a timed frame
        if "SQ_INSTS_VALU" not in c:
            continue
        row = {"valu_wave_insts": round(c["SQ_INSTS_VALU"]),
               "lanes_per_valu_inst": round(c.get("SQ_THREAD_CYCLES_VALU", 0.0) / max(c.get("SQ_ACTIVE_INST_VALU", 0.0), 1.0), 1)}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            row["fabric_bytes"] = round((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0)
            if kern in avg_us:
                row["avg_us_alone"] = round(avg_us[kern], 1)
                row["fabric_GBs"] = round(row["fabric_bytes"] / avg_us[kern] / 1e3, 1)
        table[kern] = row
    return {"per_launch": table, "source": [os.path.basename(pmc[-1])] + ([os.path.basename(stats[-1])] if stats else [])}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be started with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
        a.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    # --force-pipeline under torch.distributed.run with ONE process: the several-GPU code path with the real collective
    # (a gather among one), RCCL's stream and the barriers included
    use_dist = world > 1 or (a.force_pipeline and "RANK" in os.environ)
    dev = torch.device("cuda", local_rank)
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)

    name = a.workload or ("bunny_1080p" if world == 1 else "field_4k")
    w = workloads.get(name)
    if a.size:
        w = dict(w)
        w["width"], w["height"] = (int(x) for x in a.size.lower().split("x"))
        name = "%s@%s" % (name, a.size.lower())
    if a.soft_points is not None or a.max_reflections is not None:
        w = dict(w)
        if a.soft_points is not None:
            w["soft_points"] = a.soft_points
            name += "@soft%d" % a.soft_points
        if a.max_reflections is not None:
            w["max_reflections"] = a.max_reflections
            name += "@depth%d" % a.max_reflections
    t0 = time.time()
    fly, params, cam = workloads.make_flyscene(w, max_leaf_faces=a.max_leaf, sah_bins=a.sah_bins, shadow_rays=a.shadow_rays, primary_rays=a.primary_rays, builder=a.builder)
    info = fly.info()
    if a.workspace_limit_mb >= 0:
        fly.set_workspace_limit(a.workspace_limit_mb << 20)
    init_s = time.time() - t0
    W, H = w["width"], w["height"]

    if a.row_tile <= 0:
        a.row_tile = 16 if world == 1 else max(16, H // (6 * world))
    base = fly.params_copy(row_tile=a.row_tile)
    mine = tiles.part_params(base, world, rank)
    rows = fly.local_rows(mine)
    # Frames in flight: frame k is enqueued on stream k % F into buffer k % F (the scene keeps one workspace per
    # stream), so the small, latency-bound launches of one frame overlap the other frames' work.
    # With the gather, the frames' streams and the collective's stream wait for each other: together they must stay
    # below the number of hardware queues, or the waits stall for good.
    queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    piped = world > 1 or a.force_pipeline
    # a frame's workspace per ray slot (rtamd_scene_workspace_bytes): 48 bytes + one per shadow ray without reflections, 130 with
    # one bounce, 210 with more
    ns = len(w["lights"]) * (1 + w["soft_points"])
    per_slot = 48 + ns if w["max_reflections"] == 0 else (112 if w["max_reflections"] == 1 else 192) + 2 * ns
    slot_bytes = (W * ((H + world - 1) // world)) * per_slot
    in_flight = max(1, min(a.in_flight or 4, 32, max(1, queues - 2) if piped else (32 if queues >= 16 else max(4, queues)),
                           max(2, int(64e9 // max(slot_bytes, 1)))))
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(in_flight - 1)]
    turn = [0]
    if not piped:
        rgbs = [torch.zeros((max(rows, 1), W, 3), dtype=torch.float32, device=dev) for _ in range(in_flight)]

        def frame(p):
            k = turn[0] % in_flight
            turn[0] += 1
            fly.render_device(rgbs[k], params=p, camera=cam, stream=streams[k])

        def flush():
            pass
    else:
        # A buffer per frame in flight: the gather of frame k (RCCL, on its own stream) overlaps the tracing of the next frames.
        # Every frame is still gathered and put in row order on rank 0 inside the timed region.
        # What is exchanged is what the reference keeps of a frame: the 8-bit pixels of result.ppm, quantised on
        # every rank with the file writer's expression (a quarter of the float framebuffer's bytes); --gather f32
        # sends the floats instead.
        def make_pipeline(as_bytes):
            dt = torch.uint8 if as_bytes else torch.float32
            if a.exchange == "all_to_all":  # `depth` counts batches of `world` frames there
                pipe = tiles.AllToAllPipeline(mine, W, 3, dev, dtype=dt, depth=max(2, -(-max(in_flight, a.pipeline_depth) // max(world, 1))),
                                              rehearse=not use_dist, collective_alone=use_dist)
            else:
                pipe = tiles.GatherPipeline(mine, W, 3, dev, dtype=dt, depth=max(in_flight, a.pipeline_depth),
                                            rehearse=not use_dist, collective_alone=use_dist, rotate=a.exchange == "rotate")
            bufs = [torch.zeros((max(rows, 1), W, 3), dtype=torch.float32, device=dev) for _ in range(in_flight)] if as_bytes else None

            def frame(p):
                k = turn[0] % in_flight
                turn[0] += 1
                buf = pipe.acquire(streams[k])
                if as_bytes:
                    fly.render_device(bufs[k], params=p, camera=cam, stream=streams[k])
                    tiles.quantize_ppm(bufs[k], buf, rows * W * 3, stream=streams[k])
                else:
                    fly.render_device(buf, params=p, camera=cam, stream=streams[k])
                pipe.submit(streams[k])

            return frame, pipe.drain

        frame, flush = make_pipeline(a.gather == "u8")

    def sync(flush_now=None):
        (flush_now or flush)()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- instrumented frame (untimed): algorithmic work of each kernel
    counted = tiles.part_params(base, world, rank)
    counted.flags = L.FLAG_COUNT_WORK | (L.FLAG_SINGLE_STREAM if a.single_stream else 0)
    frame(counted)
    flush()
    torch.cuda.synchronize(dev)
    work = fly.stats()

    timed = tiles.part_params(base, world, rank)
    timed.flags = L.FLAG_SINGLE_STREAM if a.single_stream else 0
    # set-up, not warm-up: the scene allocates a stream's workspace at the first frame it sees on that stream
    # (hipMalloc, a device-wide wait), so every stream renders once before the warm-up steps begin
    for _ in range(in_flight):
        frame(timed)
    sync()
    for _ in range(a.warmup):
        frame(timed)
    sync()
    # The timed region: exactly K steps between barrier + synchronize on both sides, repeated; a K-step region of a
    # sub-millisecond frame is a few milliseconds long, so one sample of it is not a measurement.
    repeat_s = []
    for _ in range(max(1, a.repeats)):
        t0 = time.perf_counter()
        for _ in range(a.steps):
            frame(timed)
        sync()
        repeat_s.append(time.perf_counter() - t0)
    st = fly.stats()
    # ---- the same frame again, one at a time on one stream, with HIP events around every launch: per-kernel
    # durations for the roofline.  Outside the timed region: the events cost about a tenth of a frame, and with
    # several frames in flight an event pair would also span the other frames' kernels.
    if not a.no_kernel_events:
        evented = tiles.part_params(base, world, rank)
        evented.flags = L.FLAG_TIME_KERNELS | L.FLAG_SINGLE_STREAM  # every kernel alone on the GPU
        scratch = torch.zeros((max(rows, 1), W, 3), dtype=torch.float32, device=dev)
        for _ in range(a.steps):
            fly.render_device(scratch, params=evented, camera=cam, stream=streams[0])
        torch.cuda.synchronize(dev)
        st = fly.stats()
        del scratch

    # ---- one frame alone, as a caller with a single view gets it: a second scene object, rendered on ONE stream (the library
    # then gives the small launches of the deeper stages narrower waves, the shortest chain on an idle chip), each frame enqueued
    # on an idle GPU and waited for; wall clock from the call to the end of the frame, and the frame's own HIP events
    fly1 = None
    single = None
    if world == 1 and not a.force_pipeline:
        fly1, _, cam1 = workloads.make_flyscene(w, max_leaf_faces=a.max_leaf, sah_bins=a.sah_bins, shadow_rays=a.shadow_rays,
                                                primary_rays=a.primary_rays, builder=a.builder)
        lone = tiles.part_params(base, 1, 0)
        buf1 = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        for _ in range(5):
            fly1.render_device(buf1, params=lone, camera=cam, stream=streams[0])
            torch.cuda.synchronize(dev)
        wall = []
        for _ in range(max(25, a.repeats)):
            t0 = time.perf_counter()
            fly1.render_device(buf1, params=lone, camera=cam, stream=streams[0])
            torch.cuda.synchronize(dev)
            wall.append((time.perf_counter() - t0) * 1e3)
        wall.sort()
        # GPU time of the same frames, first to last launch: a pair of torch events on the frame's stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gpu = []
        for _ in range(max(25, a.repeats)):
            e0.record(streams[0])
            fly1.render_device(buf1, params=lone, camera=cam, stream=streams[0])
            e1.record(streams[0])
            torch.cuda.synchronize(dev)
            gpu.append(e0.elapsed_time(e1))
        gpu.sort()
        single = {"wall_ms": round(wall[len(wall) // 2], 4), "wall_ms_min": round(wall[0], 4), "wall_ms_max": round(wall[-1], 4),
                  "events_ms": round(gpu[len(gpu) // 2], 4), "frames": len(wall)}
        del buf1

    # ---- the same frames with the other payload of the gather (floats if bytes were timed above, or the reverse)
    other_elapsed = None
    if piped:
        turn[0] = 0
        other_frame, other_flush = make_pipeline(a.gather != "u8")
        for _ in range(a.warmup):
            other_frame(timed)
        sync(other_flush)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            other_frame(timed)
        sync(other_flush)
        other_elapsed = time.perf_counter() - t0

    # ---- reduce over ranks: slowest rank's time of every repeat, all ranks' rays
    t_el = torch.tensor(repeat_s + [other_elapsed or 0.0], dtype=torch.float64, device=dev)
    rays = torch.tensor([st.rays_primary, st.rays_shadow, st.rays_bounce], dtype=torch.float64, device=dev)
    kern = torch.tensor([st.ms_trace, st.ms_shadow, st.launches_trace, st.launches_shadow, st.ms_compact, st.ms_shade, st.ms_frames,
                         st.ms_raster, st.launches_raster], dtype=torch.float64, device=dev)
    wk = torch.tensor([work.box_tests[0], work.tri_plane[0], work.tri_edge[0], work.box_tests[1], work.tri_plane[1], work.tri_edge[1],
                       work.box_tests[2], work.tri_plane[2], work.tri_edge[2]], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(wk, op=dist.ReduceOp.SUM)
        kmax = kern.clone()
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
    else:
        kmax = kern
    repeat_ms = sorted(float(x) / a.steps * 1e3 for x in t_el[:-1].tolist())
    other_elapsed = float(t_el[-1].item())
    rays_frame = float(rays.sum().item())
    ms_per_step = repeat_ms[len(repeat_ms) // 2] if len(repeat_ms) % 2 else 0.5 * (repeat_ms[len(repeat_ms) // 2 - 1] + repeat_ms[len(repeat_ms) // 2])
    value = rays_frame / ms_per_step / 1e3

    if rank == 0:
        out = {
            "metric": "Mrays/s (primary+shadow+reflection)",
            "value": round(value, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "repeats": len(repeat_ms),
            "ms_per_step_min": round(repeat_ms[0], 4),
            "ms_per_step_max": round(repeat_ms[-1], 4),
            "frame_ms_single": None if single is None else single["wall_ms"],
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": ("synthetic" if w["scene"] != "bunny" else
                     "reference scene (bunny.ply converted to OBJ), default camera" +
                     (", material modified: Ks raised from 0.1 to %g so that reflection rays occur" % w["ks"] if w.get("ks") is not None else "")),
            "config": {
                "workload": name, "material_override": ({"ks": w["ks"]} if w.get("ks") is not None else None), "width": W, "height": H, "triangles": int(info.num_faces), "lights": len(w["lights"]),
                "samples_per_light": 1 + w["soft_points"], "max_reflections": w["max_reflections"],
                "rays_per_frame": {"primary": int(rays[0].item()), "shadow": int(rays[1].item()), "bounce": int(rays[2].item())},
                "parallelism": ("one GPU, no exchange" if world == 1 else
                                "row tiles of %d rows round-robin over %d GPUs, %s of %s pixels, overlapped with the next frames" %
                                (a.row_tile, world, {"all_to_all": "one all-to-all per %d frames (frame k whole on rank k mod %d)" % (world, world),
                                                     "gather": "one gather per frame to rank 0", "rotate": "one gather per frame to rank k mod %d" % world}[a.exchange],
                                 "8-bit (PPM)" if a.gather == "u8" else "float32")),
                "frames_in_flight": in_flight,
                "shadow_rays": "per-sample candidate grids (%d face references)" % info.shadow_grid_entries if info.shadow_grid_entries else "BVH walk",
                "bvh": {"nodes": int(info.num_nodes), "leaves": int(info.num_leaves), "depth": int(info.depth),
                        "build_ms": round(float(info.build_ms), 1),
                        "builder": ("host SAH", "device LBVH")[int(info.builder)], "hierarchy_ms": round(float(info.hierarchy_ms), 1), "device_MB": round(info.device_bytes / 1e6, 2)},
                "workspace_MB": round(fly.workspace_bytes() / 1e6, 1),
                "scene_init_s": round(init_s, 2),
            },
        }
        # ---- roofline of the dominant kernel (slowest rank's event times; work summed over ranks / ranks).
        # Three groups are timed: the BVH walk for closest hits, the shadow-ray kernel, and the screen-space pass of the
        # primary rays (face_raster_kernel; with forced bins the three binning kernels + raster_kernel, timed as one).
        if not a.no_kernel_events and kmax[3].item() > 0:
            group_ms = [kmax[0].item(), kmax[1].item(), kmax[7].item()]
            group_launches = [kmax[2].item(), kmax[3].item(), kmax[8].item()]
            wl = wk.cpu().numpy() / world  # per-GPU share of one frame
            flops = [wl[3 * g] * FLOPS_BOX + wl[3 * g + 1] * FLOPS_PLANE + wl[3 * g + 2] * FLOPS_EDGE for g in range(3)]
            byts = [wl[3 * g] * BYTES_BOX + wl[3 * g + 1] * BYTES_PLANE + wl[3 * g + 2] * BYTES_EDGE for g in range(3)]
            k = max(range(3), key=lambda g: group_ms[g])
            bounce = w["max_reflections"] > 0 and rays[2].item() > 0
            closest_key = "walk_steal_kernel<1, false>" if bounce else "walk_steal_kernel<0, false>"
            shadow_key = "shadow_grid_kernel<false>" if info.shadow_grid_entries else "walk_kernel<2, false>"
            raster_key = "raster_kernel<false>" if a.primary_rays == 2 else "face_raster_kernel<false>"
            kname = ("walk_steal_kernel (closest hits of the reflection rays)" if bounce else "walk_steal_kernel (closest hits, primary rays)",
                     "shadow_grid_kernel" if info.shadow_grid_entries else "walk_kernel<shadow>",
                     "raster_kernel (+ bin_count, bin_offsets, bin_fill)" if a.primary_rays == 2 else "face_raster_kernel")[k]
            launches_per_frame = group_launches[k] / a.steps
            per_launch_ms = group_ms[k] / group_launches[k]
            flops_launch = flops[k] / launches_per_frame
            bytes_launch = byts[k] / launches_per_frame
            # Two rooflines of that kernel, both from its ALGORITHMIC work per launch (counted tests x bytes or flops per
            # test) over its average launch duration.  `roofline` is the memory one the bench contract names; the path
            # is bound by neither (dependent fetches of small records, divergent FP32 VALU work), which is why both
            # fractions are far from 1 and why `traffic`, the bytes that really reached HBM, is below the algorithmic
            # bytes: the caches serve the re-reads.
            ach_b = bytes_launch / (per_launch_ms * 1e-3) / 1e9
            traffic, traffic_src = pmc_traffic(name, (closest_key, shadow_key, raster_key)[k])
            # What binds it is neither roofline: a closest-hit walk is a chain of dependent fetches per ray (latency), the list
            # kernels issue VALU work at a third of the lanes; the fractions against both peaks are reported all the same.
            out["roofline"] = {"kernel": kname, "bound": "latency" if k == 0 else "valu_issue", "roofline": "hbm",
                               "achieved": round(ach_b, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": round(ach_b / PEAK_HBM_GBS, 4), "traffic": None if traffic is None else round(traffic),
                               "algorithmic_bytes_per_launch": round(bytes_launch),
                               "avg_launch_ms": round(per_launch_ms, 4), "launches_per_frame": launches_per_frame,
                               "share_of_frame": round(group_ms[k] / max(kmax[6].item(), 1e-9), 3),
                               "note": "frac = algorithmic bytes / time against HBM's 8 TB/s as the bench contract defines it; those bytes are re-reads "
                                       "served by L2 / Infinity Cache (the scene is %.1f MB), so frac is no bound and can pass 1 on large frames: "
                                       "hbm_side_frac (counter bytes) and cache_level are the rates against what really serves them; avg_launch_ms: HIP events, "
                                       "one frame at a time on one stream after the timed region, one kernel per launch and each alone on the GPU "
                                       "(in the timed region %d frames overlap, and the shadow rays of a level share a launch with the closest hits "
                                       "of the next: shadow_walk_kernel)" % (info.device_bytes / 1e6, in_flight)}
            if traffic is not None:
                out["roofline"]["traffic_unit"] = "bytes/launch at the L2's memory side, (2*FETCH_SIZE + WRITE_SIZE) KiB, from profiles/" + traffic_src
                out["roofline"]["hbm_side_GBs"] = round(traffic / (per_launch_ms * 1e-3) / 1e9, 1)
                out["roofline"]["hbm_side_frac"] = round(traffic / (per_launch_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
            # The algorithmic bytes are re-reads of a scene of a few MB to a few hundred MB: they come out of L2 and the Infinity
            # Cache, so `frac` (against HBM, as the bench contract asks) bounds nothing and may exceed 1 on a large frame; the
            # same rate against the levels that serve it, and the instruction issue rate the kernel really runs against:
            out["roofline"]["cache_level"] = {"achieved_GBs": round(ach_b, 1), "l2_peak_GBs": PEAK_L2_GBS, "frac_of_l2": round(ach_b / PEAK_L2_GBS, 4),
                                              "infinity_cache_gather_GBs": PEAK_INFINITY_CACHE_GATHER_GBS,
                                              "frac_of_infinity_cache_gather": round(ach_b / PEAK_INFINITY_CACHE_GATHER_GBS, 4)}
            prof = profile_table(name)
            insts = None if prof is None else prof["per_launch"].get((closest_key, shadow_key, raster_key)[k], {})
            if insts and insts.get("valu_wave_insts"):
                rate = insts["valu_wave_insts"] / (per_launch_ms * 1e-3)
                out["roofline"]["issue"] = {"valu_wave_insts_per_launch": insts["valu_wave_insts"], "lanes_per_valu_inst": insts.get("lanes_per_valu_inst"),
                                            "issue_frac": round(rate / PEAK_VALU_WAVE_INSTS_PER_S, 4),
                                            "active_lane_frac": round(rate / PEAK_VALU_WAVE_INSTS_PER_S * insts.get("lanes_per_valu_inst", 64.0) / 64.0, 4),
                                            "peak_wave_insts_per_s": PEAK_VALU_WAVE_INSTS_PER_S,
                                            "note": "wave-level VALU instructions of one launch (profiles/, PMC) over avg_launch_ms against 256 CU x 4 SIMD x "
                                                    "2.4 GHz / 2 cycles; active_lane_frac = the same weighted by the lanes that were switched on"}
            # the same figures for all three groups, so that the runner-up shows as well (on the 1080p bunny the closest-hit walk and
            # the two shadow launches are within a few per cent of each other, and which one leads differs from box to box)
            names_all = ("closest_hits", "shadow_rays", "primary_rays_screen_space")
            keys_all = (closest_key, shadow_key, raster_key)
            out["roofline_all"] = {}
            for g in range(3):
                if group_launches[g] <= 0 or group_ms[g] <= 0:
                    continue
                lpf = group_launches[g] / a.steps
                ms_l = group_ms[g] / group_launches[g]
                tr, _ = pmc_traffic(name, keys_all[g])
                gb = byts[g] / lpf / (ms_l * 1e-3) / 1e9
                tf = flops[g] / lpf / (ms_l * 1e-3) / 1e12
                row = {"kernel": keys_all[g], "avg_launch_ms": round(ms_l, 4), "launches_per_frame": lpf,
                       "algorithmic_GBs": round(gb, 1), "hbm_frac": round(gb / PEAK_HBM_GBS, 4), "l2_frac": round(gb / PEAK_L2_GBS, 4),
                       "valu_TFLOPs": round(tf, 3), "valu_frac": round(tf / PEAK_FP32_TFLOPS, 4),
                       "traffic": None if tr is None else round(tr)}
                if tr is not None:
                    row["hbm_side_frac"] = round(tr / (ms_l * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
                pi = None if prof is None else prof["per_launch"].get(keys_all[g])
                if pi and pi.get("valu_wave_insts"):
                    row["issue_frac"] = round(pi["valu_wave_insts"] / (ms_l * 1e-3) / PEAK_VALU_WAVE_INSTS_PER_S, 4)
                out["roofline_all"][names_all[g]] = row
            ach = flops_launch / (per_launch_ms * 1e-3) / 1e12
            out["roofline_valu"] = {"kernel": kname, "bound": "valu", "achieved": round(ach, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                                    "frac": round(ach / PEAK_FP32_TFLOPS, 4), "algorithmic_flops_per_launch": round(flops_launch)}
            out["kernels"] = {"trace_ms_per_frame": round(group_ms[0] / a.steps, 4), "shadow_ms_per_frame": round(group_ms[1] / a.steps, 4),
                              "raster_ms_per_frame": round(group_ms[2] / a.steps, 4),
                              "compact_ms_per_frame": round(kmax[4].item() / a.steps, 4), "shade_ms_per_frame": round(kmax[5].item() / a.steps, 4),
                              "frame_ms_events": round(kmax[6].item() / a.steps, 4),
                              "tests_per_ray": {"box": round(float(wk[0::3].sum().item()) / rays_frame, 2),
                                                "tri_plane": round(float(wk[1::3].sum().item()) / rays_frame, 2),
                                                "tri_edge": round(float(wk[2::3].sum().item()) / rays_frame, 2)}}
        table = profile_table(name)
        if table is not None:
            out["kernels_profiled"] = table
        if single is not None:
            single["note"] = ("one frame at a time on one stream of an otherwise idle GPU: host call to end of frame (wall_ms) and first "
                              "to last launch of the frame by HIP events (events_ms); ms_per_step is the throughput figure, with "
                              "%d frames in flight" % in_flight)
            out["single_frame"] = single
        if world == 1:
            # the same frame through the blocking host-buffer entry point (upload of nothing, download of
            # rgb + face ids over PCIe): reported beside `value`, never as `value`
            plain = tiles.part_params(base, 1, 0)
            h_st = L.Stats()
            lib = L.lib()

            def host_ms(rgb_ptr, fid_ptr):
                def host_frame():
                    L.check(lib.rtamd_render((fly1 or fly).scene, C.byref(cam), C.byref(plain), C.cast(rgb_ptr, C.POINTER(C.c_float)),
                                             C.cast(fid_ptr, C.POINTER(C.c_int32)), C.byref(h_st)), "rtamd_render")

                host_frame()
                host_frame()
                ts = []
                for _ in range(15):
                    t0 = time.perf_counter()
                    host_frame()
                    ts.append((time.perf_counter() - t0) * 1e3)
                return sorted(ts)[len(ts) // 2]

            h_rgb = np.zeros((H, W, 3), np.float32)
            h_fid = np.zeros((H, W), np.int32)
            pageable_ms = host_ms(h_rgb.ctypes.data, h_fid.ctypes.data)
            p_rgb, p_fid = C.c_void_p(), C.c_void_p()
            L.check(lib.rtamd_host_alloc(C.byref(p_rgb), H * W * 12), "rtamd_host_alloc")
            L.check(lib.rtamd_host_alloc(C.byref(p_fid), H * W * 4), "rtamd_host_alloc")
            locked_ms = host_ms(p_rgb.value, p_fid.value)
            lib.rtamd_host_free(p_rgb)
            lib.rtamd_host_free(p_fid)
            # ... and what Flyscene::raytraceScene itself needs of a frame: the 8-bit pixels of result.ppm (rtamd_render_ppm,
            # quantised on the device; a quarter of the float framebuffer's bytes cross the link, no face ids)
            p_pix = C.c_void_p()
            L.check(lib.rtamd_host_alloc(C.byref(p_pix), H * W * 3), "rtamd_host_alloc")
            exact = C.c_int32(0)

            def ppm_frame():
                L.check(lib.rtamd_render_ppm((fly1 or fly).scene, C.byref(cam), C.byref(plain), p_pix, C.byref(exact), None), "rtamd_render_ppm")

            ppm_frame()
            ppm_frame()
            ts = []
            for _ in range(25):
                t0 = time.perf_counter()
                ppm_frame()
                ts.append((time.perf_counter() - t0) * 1e3)
            ppm_ms = sorted(ts)[len(ts) // 2]
            lib.rtamd_host_free(p_pix)
            out["host_boundary"] = {"ms_per_frame": round(ppm_ms, 3), "Mrays_per_s": round(rays_frame / ppm_ms / 1e3, 1),
                                    "ms_per_frame_float_planes": round(locked_ms, 3), "ms_per_frame_float_planes_pageable": round(pageable_ms, 3),
                                    "note": "blocking entry points, one frame at a time, PCIe download included.  ms_per_frame: rtamd_render_ppm, the "
                                            "path of Flyscene::raytraceScene (the 8-bit pixels of result.ppm, %.1f MB, page-locked buffer; bytes exact: %d); "
                                            "ms_per_frame_float_planes: rtamd_render with rgb + face ids as floats / ints (%.1f MB) into page-locked "
                                            "buffers, ..._pageable: into ordinary malloc'ed buffers" % (H * W * 3 / 1e6, exact.value, H * W * 16 / 1e6)}
        if piped:
            out["other_gather"] = {"payload": "float32" if a.gather == "u8" else "8-bit (PPM pixels)",
                                   "ms_per_step": round(other_elapsed / a.steps * 1e3, 4),
                                   "value": round(rays_frame * a.steps / other_elapsed / 1e6, 3)}
            # the whole frame of the SAME workload on this one GPU, no exchange: the single-GPU figure that the
            # N-GPU value above is to be compared with (the N = 1 default of this script is another workload)
            whole = tiles.part_params(base, 1, 0)
            full = [torch.zeros((H, W, 3), dtype=torch.float32, device=dev) for _ in range(in_flight)]

            def whole_frames(n):
                for i in range(n):
                    fly.render_device(full[i % in_flight], params=whole, camera=cam, stream=streams[i % in_flight])
                torch.cuda.synchronize(dev)

            whole_frames(max(a.warmup, in_flight) + 1)  # every stream's workspace grows to the whole frame first
            t0 = time.perf_counter()
            whole_frames(a.steps)
            one_ms = (time.perf_counter() - t0) / a.steps * 1e3
            out["single_gpu_same_workload"] = {"ms_per_step": round(one_ms, 4), "value": round(rays_frame / one_ms / 1e3, 3),
                                               "speedup": round(one_ms / ms_per_step, 3)}
        if world == 1 and a.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(w, cam, a.cpu_seconds)
            out["gpu_over_cpu"] = round(value / max(out["cpu_baseline"]["value"], 1e-9), 1)
            out["gpu_over_cpu_note"] = ("value / cpu_baseline.value: one GPU against %d host threads (the GPU box's CPU share; the machine "
                                        "has %d)" % (out["cpu_baseline"]["cores"], os.cpu_count() or 0))
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
