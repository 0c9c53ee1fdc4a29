"""sha256 of a workload's frame (float image bits + face ids), to compare builds and environment settings:
    python tools/frame_digest.py WORKLOAD"""
import hashlib
import sys

import torch

sys.path.insert(0, ".")
from raytracer_amd import workloads  # noqa: E402

w = workloads.get(sys.argv[1])
fly, params, cam = workloads.make_flyscene(w)
rgb = torch.zeros((w["height"], w["width"], 3), dtype=torch.float32, device="cuda")
fid = torch.zeros((w["height"], w["width"]), dtype=torch.int32, device="cuda")
fly.render_device(rgb, fid, camera=cam)
torch.cuda.synchronize()
h = hashlib.sha256(rgb.cpu().numpy().tobytes() + fid.cpu().numpy().tobytes()).hexdigest()
st = fly.stats()
print(sys.argv[1], h[:16], (st.rays_primary, st.rays_shadow, st.rays_bounce, st.hits), flush=True)
