#!/usr/bin/env python3
"""Work per ray of one workload under each of its lights separately (device counters)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import workloads, _lib as L

name = sys.argv[1] if len(sys.argv) > 1 else "field2_mirrors_small"
w = workloads.get(name)
lights = list(w["lights"])
for sel in [lights] + [[l] for l in lights]:
    ww = dict(w, lights=sel)
    fly, params, cam = workloads.make_flyscene(ww)
    p = fly.params_copy()
    p.flags = L.FLAG_COUNT_WORK
    import torch
    rgb = torch.zeros((w["height"], w["width"], 3), dtype=torch.float32, device="cuda")
    fly.render_device(rgb, params=p, camera=cam)
    torch.cuda.synchronize()
    st = fly.stats()
    rs = st.rays_shadow or 1
    rc = st.rays_primary + st.rays_bounce
    print([tuple(round(float(x), 2) for x in l) for l in sel] if len(sel) == 1 else "all",
          "rays p/s/b", st.rays_primary, st.rays_shadow, st.rays_bounce,
          "closest box/ray %.1f plane %.1f" % (st.box_tests[0] / rc, st.tri_plane[0] / rc),
          "shadow box/ray %.1f plane %.1f edge %.1f" % (st.box_tests[1] / rs, st.tri_plane[1] / rs, st.tri_edge[1] / rs), flush=True)
