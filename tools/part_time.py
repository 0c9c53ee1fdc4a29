"""Time what ONE rank of an N-rank run traces (no exchange): python tools/part_time.py WORKLOAD [row_tile]."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from raytracer_amd import tiles, workloads  # noqa: E402

name = sys.argv[1]
row_tile = int(sys.argv[2]) if len(sys.argv) > 2 else 16
worlds = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8]
w = workloads.get(name)
fly, params, cam = workloads.make_flyscene(w)
base = fly.params_copy(row_tile=row_tile)
for world in worlds:
    for rank in (range(world) if len(sys.argv) > 4 else sorted({0, world - 1})):
        p = tiles.part_params(base, world, rank)
        rows = fly.local_rows(p)
        rgb = torch.zeros((rows, w["width"], 3), dtype=torch.float32, device="cuda")
        for _ in range(5):
            fly.render_device(rgb, params=p, camera=cam)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(30):
            fly.render_device(rgb, params=p, camera=cam)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 30 * 1e3
        print("%s world %d rank %d: %d rows, %.3f ms/frame" % (name, world, rank, rows, ms), flush=True)
