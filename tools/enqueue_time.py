"""Host time to enqueue a frame vs GPU time per frame: python tools/enqueue_time.py WORKLOAD [IN_FLIGHT [WORLD [ROW_TILE]]]
(WORLD > 1: rank 0's share of a frame partitioned over WORLD ranks)"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from raytracer_amd import tiles, workloads  # noqa: E402

name = sys.argv[1]
F = int(sys.argv[2]) if len(sys.argv) > 2 else 2
world = int(sys.argv[3]) if len(sys.argv) > 3 else 1
w = workloads.get(name)
fly, params, cam = workloads.make_flyscene(w)
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(F - 1)]
row_tile = int(sys.argv[4]) if len(sys.argv) > 4 else 16
p = tiles.part_params(fly.params_copy(row_tile=row_tile), world, 0) if world > 1 else fly.params_copy()
rows = fly.local_rows(p) if world > 1 else w["height"]
bufs = [torch.zeros((rows, w["width"], 3), dtype=torch.float32, device="cuda") for _ in range(F)]
for n in (20, 200):
    for i in range(8):
        fly.render_device(bufs[i % F], params=p, camera=cam, stream=streams[i % F])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fly.render_device(bufs[i % F], params=p, camera=cam, stream=streams[i % F])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s (rank 0 of %d) F=%d n=%d: host enqueue %.3f ms/frame, total %.3f ms/frame" % (name, world, F, n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
