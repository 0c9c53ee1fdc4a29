"""Throughput with several frames in flight (one scene handle + stream each): python tools/inflight.py WORKLOAD WORLD"""
import sys
import time

import torch

sys.path.insert(0, ".")
from raytracer_amd import _lib as L  # noqa: E402
from raytracer_amd import tiles, workloads  # noqa: E402

name = sys.argv[1]
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w = workloads.get(name)
for inflight in (1, 2, 3, 4):
    flies = [workloads.make_flyscene(w) for _ in range(inflight)]
    streams = [torch.cuda.Stream() for _ in range(inflight)]
    cam = flies[0][2]
    for single in ((0, L.FLAG_SINGLE_STREAM) if inflight == 1 else (L.FLAG_SINGLE_STREAM,)):
        ps, bufs = [], []
        for fly, _, _ in flies:
            p = tiles.part_params(fly.params_copy(row_tile=16), world, 0)
            p.flags = single
            ps.append(p)
            bufs.append(torch.zeros((fly.local_rows(p), w["width"], 3), dtype=torch.float32, device="cuda"))

        def run(n):
            for i in range(n):
                k = i % inflight
                flies[k][0].render_device(bufs[k], params=ps[k], camera=cam, stream=streams[k])

        run(8)
        torch.cuda.synchronize()
        t = time.perf_counter()
        run(60)
        torch.cuda.synchronize()
        print("%s world %d: %d in flight, flags %d: %.3f ms/frame" % (name, world, inflight, single, (time.perf_counter() - t) / 60 * 1e3), flush=True)
    del flies
