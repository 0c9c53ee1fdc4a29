"""What the binning pass of a workload's view finds: python tools/bin_stats.py WORKLOAD"""
import sys

import torch

sys.path.insert(0, ".")
from raytracer_amd import workloads  # noqa: E402

w = workloads.get(sys.argv[1])
fly, params, cam = workloads.make_flyscene(w)
rgb = torch.zeros((w["height"], w["width"], 3), dtype=torch.float32, device="cuda")
for k in range(4):
    fly.render_device(rgb, camera=cam)
    torch.cuda.synchronize()
    st = fly.stats()
    print("frame %d: bin entries %d, non-empty bins %d, list %.1f, primary path %d, overflow %d" %
          (k, st.bin_entries, st.bins_nonempty, st.bin_entries / max(st.bins_nonempty, 1), st.primary_path, st.bin_overflow), flush=True)
