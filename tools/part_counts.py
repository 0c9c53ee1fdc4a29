"""Algorithmic work of one rank's share of a frame (device counters): python tools/part_counts.py WORKLOAD WORLD [ROW_TILE]
Prints, for the screen-space pass of the primary rays: faces projected, (face, pixel) pairs tested, edge tests."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from raytracer_amd import _lib as L  # noqa: E402
from raytracer_amd import tiles, workloads  # noqa: E402

name, world = sys.argv[1], int(sys.argv[2])
row_tile = int(sys.argv[3]) if len(sys.argv) > 3 else 16
w = workloads.get(name)
fly, params, cam = workloads.make_flyscene(w)
for part in sorted({0, world // 2, world - 1}):
    p = tiles.part_params(fly.params_copy(row_tile=row_tile), world, part)
    p.flags = L.FLAG_COUNT_WORK
    rgb = torch.zeros((max(fly.local_rows(p), 1), w["width"], 3), dtype=torch.float32, device="cuda")
    fly.render_device(rgb, params=p, camera=cam)
    torch.cuda.synchronize()
    st = fly.stats()
    print("%s part %d of %d (row tiles of %d): %d rows, primary pass: %d faces projected, %d pairs, %d edge tests; hits %d, shadow rays %d"
          % (name, part, world, row_tile, fly.local_rows(p), st.box_tests[2], st.tri_plane[2], st.tri_edge[2], st.hits, st.rays_shadow), flush=True)
