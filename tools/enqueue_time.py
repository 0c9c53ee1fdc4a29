"""Host time to enqueue a frame vs GPU time per frame: python tools/enqueue_time.py WORKLOAD [IN_FLIGHT]"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from raytracer_amd import workloads  # noqa: E402

name = sys.argv[1]
F = int(sys.argv[2]) if len(sys.argv) > 2 else 2
w = workloads.get(name)
fly, params, cam = workloads.make_flyscene(w)
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(F - 1)]
bufs = [torch.zeros((w["height"], w["width"], 3), dtype=torch.float32, device="cuda") for _ in range(F)]
p = fly.params_copy()
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
    print("%s F=%d n=%d: host enqueue %.3f ms/frame, total %.3f ms/frame" % (name, F, n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
