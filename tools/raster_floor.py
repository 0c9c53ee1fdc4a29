#!/usr/bin/env python3
"""How much of the screen-bin pass is floor (launch, idle waves, binning of faces) and how much is the lists:
the default view of a workload against the same view turned away from the mesh, kernels timed alone by HIP events.
usage: tools/raster_floor.py [workload]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import _lib as L, workloads

name = sys.argv[1] if len(sys.argv) > 1 else "bunny_1080p"
w = workloads.get(name)
fly, params, cam = workloads.make_flyscene(w)
W, H = w["width"], w["height"]
dev = torch.device("cuda", 0)
rgb = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
for label, turn in (("default view", 0.0), ("turned away", 3.14159)):
    if turn:
        fly.camera_rotate(0.0, turn)
        cam = fly.getCamera()
    p = fly.params_copy(flags=L.FLAG_TIME_KERNELS | L.FLAG_SINGLE_STREAM)
    for _ in range(3):
        fly.render_device(rgb, params=p, camera=cam)
    torch.cuda.synchronize()
    fly.stats()
    n = 20
    for _ in range(n):
        fly.render_device(rgb, params=p, camera=cam)
    torch.cuda.synchronize()
    st = fly.stats()
    print("%-14s raster pass %.4f ms  shadow %.4f  compact %.4f  shade %.4f  trace %.4f frame %.4f | hits %d bin entries %d nonempty bins %d path %d" % (
        label, st.ms_raster / n, st.ms_shadow / n, st.ms_compact / n, st.ms_shade / n, st.ms_trace / n, st.ms_frames / n, st.hits, st.bin_entries, st.bins_nonempty, st.primary_path))
