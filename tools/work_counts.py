"""Algorithmic work per ray class of one frame: python tools/work_counts.py WORKLOAD"""
import sys

import torch

sys.path.insert(0, ".")
from raytracer_amd import _lib as L  # noqa: E402
from raytracer_amd import workloads  # noqa: E402

w = workloads.get(sys.argv[1])
fly, params, cam = workloads.make_flyscene(w)
p = fly.params_copy()
p.flags = L.FLAG_COUNT_WORK
rgb = torch.zeros((w["height"], w["width"], 3), dtype=torch.float32, device="cuda")
fly.render_device(rgb, params=p, camera=cam)
torch.cuda.synchronize()
st = fly.stats()
print("rays: primary %d shadow %d bounce %d, hits %d" % (st.rays_primary, st.rays_shadow, st.rays_bounce, st.hits))
for k, (name, rays) in enumerate((("BVH walk (closest)", st.rays_bounce), ("shadow", st.rays_shadow), ("screen bins", st.rays_primary))):
    r = max(rays, 1)
    print("%-20s per ray: box %.2f plane %.2f edge %.2f" % (name, st.box_tests[k] / r, st.tri_plane[k] / r, st.tri_edge[k] / r))
