"""How the per-kernel times of the bunny bounce frame move with the image size (one frame at a time, HIP events)."""
import sys

import torch

sys.path.insert(0, ".")
from raytracer_amd import _lib as L  # noqa: E402
from raytracer_amd import workloads  # noqa: E402

for scale in (8, 4, 2, 1, 0.5):
    w = dict(workloads.get("bunny_1080p"))
    w["width"], w["height"] = int(1920 / scale), int(1080 / scale)
    fly, params, cam = workloads.make_flyscene(w)
    p = fly.params_copy()
    p.flags = L.FLAG_TIME_KERNELS | L.FLAG_SINGLE_STREAM
    rgb = torch.zeros((w["height"], w["width"], 3), dtype=torch.float32, device="cuda")
    for _ in range(3):
        fly.render_device(rgb, params=p, camera=cam)
    torch.cuda.synchronize()
    fly.stats()
    n = 20
    for _ in range(n):
        fly.render_device(rgb, params=p, camera=cam)
    torch.cuda.synchronize()
    st = fly.stats()
    print("%dx%d: bounce rays %d hits %d | walk %.3f ms, shadow %.3f, raster pass %.3f, frame %.3f" %
          (w["width"], w["height"], st.rays_bounce, st.hits, st.ms_trace / n, st.ms_shadow / n, st.ms_raster / n, st.ms_frames / n), flush=True)
    fly.close()
