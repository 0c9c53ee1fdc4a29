"""One rank's share of a frame, one frame at a time on one stream, for a kernel trace:
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/part_kernels.py WORKLOAD WORLD [ROW_TILE]"""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from raytracer_amd import _lib as L  # noqa: E402
from raytracer_amd import tiles, workloads  # noqa: E402

name, world = sys.argv[1], int(sys.argv[2])
w = workloads.get(name)
fly, params, cam = workloads.make_flyscene(w)
row_tile = int(sys.argv[3]) if len(sys.argv) > 3 else 16
p = tiles.part_params(fly.params_copy(row_tile=row_tile), world, 0)
p.flags = L.FLAG_SINGLE_STREAM
rgb = torch.zeros((fly.local_rows(p), w["width"], 3), dtype=torch.float32, device="cuda")
for _ in range(50):
    fly.render_device(rgb, params=p, camera=cam)
torch.cuda.synchronize()
