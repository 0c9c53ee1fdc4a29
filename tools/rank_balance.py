"""What every rank of WORLD traces, frames in flight, for a row-tile height: the slowest rank sets the frame rate.
    GPU_MAX_HW_QUEUES=8 python tools/rank_balance.py WORKLOAD WORLD ROW_TILE[,ROW_TILE...] [IN_FLIGHT]"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from raytracer_amd import tiles, workloads  # noqa: E402

name, world = sys.argv[1], int(sys.argv[2])
row_tiles = [int(x) for x in sys.argv[3].split(",")]
inflight = int(sys.argv[4]) if len(sys.argv) > 4 else 4
w = workloads.get(name)
fly, _, cam = workloads.make_flyscene(w)
streams = [torch.cuda.Stream() for _ in range(inflight)]
for row_tile in row_tiles:
    ms = []
    for rank in range(world):
        p = tiles.part_params(fly.params_copy(row_tile=row_tile), world, rank)
        bufs = [torch.zeros((max(fly.local_rows(p), 1), w["width"], 3), dtype=torch.float32, device="cuda") for _ in range(inflight)]

        def run(n):
            for i in range(n):
                fly.render_device(bufs[i % inflight], params=p, camera=cam, stream=streams[i % inflight])

        run(16)
        torch.cuda.synchronize()
        t = time.perf_counter()
        run(96)
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t) / 96 * 1e3)
    print("%s, %d ranks, row tiles of %d, %d in flight: slowest %.4f mean %.4f ms/frame  (%s)" % (
        name, world, row_tile, inflight, max(ms), sum(ms) / len(ms), " ".join("%.3f" % x for x in ms)), flush=True)
