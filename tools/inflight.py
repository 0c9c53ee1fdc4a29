"""Throughput of one rank's share with several frames in flight (one scene, a stream per frame):
    [GPU_MAX_HW_QUEUES=8] python tools/inflight.py WORKLOAD [WORLD [F,F,... [ROW_TILE]]]"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from raytracer_amd import tiles, workloads  # noqa: E402

name = sys.argv[1]
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
counts = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 6, 8]
w = workloads.get(name)
fly, _, cam = workloads.make_flyscene(w)
row_tile = int(sys.argv[4]) if len(sys.argv) > 4 else 16
p = tiles.part_params(fly.params_copy(row_tile=row_tile), world, 0)
for inflight in counts:
    streams = [torch.cuda.Stream() for _ in range(inflight)]
    bufs = [torch.zeros((fly.local_rows(p), w["width"], 3), dtype=torch.float32, device="cuda") for _ in range(inflight)]

    def run(n):
        for i in range(n):
            fly.render_device(bufs[i % inflight], params=p, camera=cam, stream=streams[i % inflight])

    run(16)
    torch.cuda.synchronize()
    t = time.perf_counter()
    run(96)
    torch.cuda.synchronize()
    print("%s, rank 0 of %d (row tiles of %d): %d in flight: %.3f ms/frame" % (name, world, row_tile, inflight, (time.perf_counter() - t) / 96 * 1e3), flush=True)
