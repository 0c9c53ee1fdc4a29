"""The benchmark workloads at full size: the default path (primary rays face by face, shadow grids), the screen bins and
the device builders against the tree walk for every ray, bit for bit:  python tools/soak_workloads.py [WORKLOAD ...]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from raytracer_amd import _lib as L  # noqa: E402
from raytracer_amd import workloads  # noqa: E402

bad = 0
for name in sys.argv[1:] or ["bunny_1080p", "bunny_4k", "bunny_sub2_4k", "field_4k"]:
    w = workloads.get(name)
    outs = []
    for build in (dict(primary_rays=L.PRIMARY_RAYS_TREE, shadow_rays=L.SHADOW_RAYS_TREE), dict(), dict(builder=L.BUILDER_DEVICE_LBVH),
                  dict(primary_rays=L.PRIMARY_RAYS_BINS), dict(primary_rays=L.PRIMARY_RAYS_BINS, builder=L.BUILDER_DEVICE_LBVH)):
        fly, params, cam = workloads.make_flyscene(w, **build)
        rgb = torch.zeros((w["height"], w["width"], 3), dtype=torch.float32, device="cuda")
        fid = torch.zeros((w["height"], w["width"]), dtype=torch.int32, device="cuda")
        fly.render_device(rgb, fid, camera=cam)
        torch.cuda.synchronize()
        st = fly.stats()
        outs.append((rgb.cpu().numpy().view(np.uint32), fid.cpu().numpy(), (st.rays_primary, st.rays_shadow, st.rays_bounce, st.hits), st.primary_path))
        del fly, rgb, fid
    for k in (1, 2, 3, 4):
        same = np.array_equal(outs[k][0], outs[0][0]) and np.array_equal(outs[k][1], outs[0][1]) and outs[k][2] == outs[0][2]
        print("%s: %s vs tree walk: %s (primary path %d, rays %s)" % (name, ("faces + grids", "faces + grids, device builders", "bins + grids", "bins + grids, device builders")[k - 1],
                                                                     "identical" if same else "DIFFERENT", outs[k][3], outs[k][2]), flush=True)
        bad += 0 if same else 1
sys.exit(1 if bad else 0)
