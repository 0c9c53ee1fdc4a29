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
F = int(os.environ.get("FRAMES_IN_FLIGHT", "2"))  # with more than 2, export GPU_MAX_HW_QUEUES=16 first
streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(F - 1)]
FRAMES = 3 * F
for dtype in (torch.uint8, torch.float32):
    pipe = tiles.GatherPipeline(mine, W, 3, dev, dtype=dtype, depth=F, collective_alone=True)  # one rank: RCCL all the same
    got = []
    for k in range(FRAMES):
        s = streams[k % F]
        buf = pipe.acquire(s)
        with torch.cuda.stream(s):
            buf[:rows] = k * 10 + rank
        img = pipe.submit(s)
        if img is not None:
            got.append(img)
    got += pipe.drain()
    torch.cuda.synchronize()
    if rank == 0:
        assert len(got) == FRAMES, len(got)
        for k, img in enumerate(got):
            vals = sorted(set(img.reshape(-1).tolist()))
            want = sorted({(k * 10 + r) % 256 if dtype == torch.uint8 else k * 10 + r for r in range(world)})
            assert vals == want, (k, vals, want)
        print("pipeline ok with", dtype, "-", world, "rank(s),", F, "frames in flight", flush=True)
dist.barrier()
dist.destroy_process_group()
