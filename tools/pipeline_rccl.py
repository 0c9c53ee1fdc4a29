"""Rehearsal of the gather pipeline with the nccl backend (RCCL), rank r on GPU r modulo the GPUs present:
    python -m torch.distributed.run --nproc-per-node 2 tools/pipeline_rccl.py
With fewer GPUs than ranks RCCL refuses two ranks on one GPU (seen on the one-GPU box); the script then says so
and exits 0."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, ".")
import raytracer_amd as R  # noqa: E402
from raytracer_amd import tiles  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
index = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
torch.cuda.set_device(index)
dev = torch.device("cuda", index)
try:
    dist.init_process_group("nccl", device_id=dev)
    x = torch.ones(4, device=dev)
    dist.all_reduce(x)
    torch.cuda.synchronize()
except Exception as e:  # noqa: BLE001
    print("rank %d: RCCL did not start (two ranks on one GPU?): %s" % (rank, str(e).splitlines()[0][:200]), flush=True)
    sys.exit(0)
W, H = 64, 48
base = R.default_params(W, H, row_tile=8)
mine = tiles.part_params(base, world, rank)
rows = R._lib.lib().rtamd_local_rows(__import__("ctypes").byref(mine))
streams = [torch.cuda.current_stream(dev), torch.cuda.Stream(dev)]
for dtype in (torch.uint8, torch.float32):
    pipe = tiles.GatherPipeline(mine, W, 3, dev, dtype=dtype)
    got = []
    for k in range(6):
        s = streams[k % 2]
        buf = pipe.acquire(s)
        with torch.cuda.stream(s):
            buf[:rows] = k * 10 + rank
        img = pipe.submit(s)
        if img is not None:
            got.append(img)
    got += pipe.drain()
    torch.cuda.synchronize()
    if rank == 0:
        assert len(got) == 6, len(got)
        for k, img in enumerate(got):
            vals = sorted(set(img.reshape(-1).tolist()))
            assert vals == [k * 10, k * 10 + 1], (k, vals)
        print("pipeline ok with", dtype, flush=True)
dist.barrier()
dist.destroy_process_group()
