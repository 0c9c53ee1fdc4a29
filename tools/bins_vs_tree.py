"""Primary rays through the screen bins and through the tree, over image sizes: time and work per pixel."""
import sys

import torch

sys.path.insert(0, ".")
from raytracer_amd import _lib as L  # noqa: E402
from raytracer_amd import workloads  # noqa: E402

name = sys.argv[1]
base = workloads.get(name)
for scale in (8, 4, 2, 1):
    row = []
    for mode in (L.PRIMARY_RAYS_AUTO, L.PRIMARY_RAYS_TREE):
        w = dict(base)
        w["width"], w["height"] = int(base["width"] / scale), int(base["height"] / scale)
        w["max_reflections"] = 0
        fly, params, cam = workloads.make_flyscene(w, primary_rays=mode)
        rgb = torch.zeros((w["height"], w["width"], 3), dtype=torch.float32, device="cuda")
        p = fly.params_copy()
        p.flags = L.FLAG_COUNT_WORK | L.FLAG_SINGLE_STREAM
        fly.render_device(rgb, params=p, camera=cam)
        torch.cuda.synchronize()
        c = fly.stats()
        p.flags = L.FLAG_TIME_KERNELS | L.FLAG_SINGLE_STREAM
        for _ in range(3):
            fly.render_device(rgb, params=p, camera=cam)
        torch.cuda.synchronize()
        fly.stats()
        n = 20
        for _ in range(n):
            fly.render_device(rgb, params=p, camera=cam)
        torch.cuda.synchronize()
        st = fly.stats()
        px = w["width"] * w["height"]
        if st.launches_raster:
            row.append("bins %.3f ms + walk %.3f (%.1f candidates/px, list %.1f, taken %d)" %
                       (st.ms_raster / n, st.ms_trace / n, c.tri_plane[2] / px, st.bin_entries / max(st.bins_nonempty, 1), st.primary_path))
        else:
            row.append("tree %.3f ms (%.1f boxes/px, %.1f faces/px)" % (st.ms_trace / n, c.box_tests[0] / px, c.tri_plane[0] / px))
        fly.close()
    print("%s %dx%d (%.2f faces/px): %s" % (name, w["width"], w["height"], fly.info().num_faces / px if False else 0, " | ".join(row)), flush=True)
