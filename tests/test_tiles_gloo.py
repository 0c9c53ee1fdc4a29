"""Row-tile partition + gather across ranks (the N > 1 path of bench.py) with world_size 2 and 3 on the
gloo backend: every rank fills its packed rows with a function of the IMAGE row, rank 0 must assemble
the full image.  No GPU: only the partition arithmetic of the C ABI and the collective are exercised."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import raytracer_amd as R
from raytracer_amd import _lib as L
from raytracer_amd import tiles


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, width, height, tile, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        base = R.default_params(width, height, row_tile=tile)
        mine = tiles.part_params(base, world, rank)
        lib = L.lib()
        rows = lib.rtamd_local_rows(C.byref(mine))
        local = torch.zeros((rows, width, 3), dtype=torch.float32)
        for lr in range(rows):
            y = lib.rtamd_global_row(C.byref(mine), lr)
            local[lr, :, 0] = y
            local[lr, :, 1] = torch.arange(width, dtype=torch.float32)
            local[lr, :, 2] = rank
        img = tiles.gather_rows(local, mine, dst=0)
        if rank == 0:
            q.put(img.numpy())
        else:
            assert img is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,width,height,tile", [(2, 16, 40, 8), (2, 8, 37, 16), (3, 8, 50, 4)])
def test_gather_assembles_the_image(world, width, height, tile):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, width, height, tile, q)) for r in range(world)]
    for p in procs:
        p.start()
    img = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert img.shape == (height, width, 3)
    ys = np.arange(height, dtype=np.float32)
    assert np.array_equal(img[:, 0, 0], ys)
    assert np.array_equal(img[:, :, 1], np.tile(np.arange(width, dtype=np.float32), (height, 1)))
    assert np.array_equal(img[:, 0, 2], ((ys // tile) % world).astype(np.float32))


def test_single_process_path_is_identity():
    p = R.default_params(4, 6)
    local = torch.arange(6 * 4 * 3, dtype=torch.float32).reshape(6, 4, 3)
    assert torch.equal(tiles.gather_rows(local, p), local)
    idx = tiles.row_permutation(R.default_params(4, 10, row_tile=2), 3)
    assert sorted(idx.tolist()) == list(range(10))


def _pipeline_worker(rank, world, port, width, height, tile, frames, q, dtype=torch.float32, depth=2):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        base = R.default_params(width, height, row_tile=tile)
        mine = tiles.part_params(base, world, rank)
        lib = L.lib()
        rows = lib.rtamd_local_rows(C.byref(mine))
        pipe = tiles.GatherPipeline(mine, width, 3, "cpu", dtype=dtype, depth=depth)
        got = []
        for k in range(frames):
            buf = pipe.acquire()
            for lr in range(rows):
                buf[lr, :, 0] = lib.rtamd_global_row(C.byref(mine), lr)
                buf[lr, :, 1] = k
                buf[lr, :, 2] = rank
            img = pipe.submit()
            if img is not None:
                got.append(img.numpy().copy())
        got += [x.numpy().copy() for x in pipe.drain()]
        if rank == 0:
            q.put(got)
        else:
            assert got == []
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,height,tile,frames,dtype,depth", [(2, 37, 8, 5, torch.float32, 2), (3, 50, 4, 1, torch.float32, 2),
                                                                  (2, 16, 16, 2, torch.float32, 2), (2, 37, 8, 4, torch.uint8, 2),
                                                                  (2, 37, 8, 11, torch.uint8, 4), (2, 21, 4, 3, torch.float32, 8)])
def test_double_buffered_gather_returns_every_frame_in_order(world, height, tile, frames, dtype, depth):
    """Also with 8-bit pixels, which is what the ranks exchange by default (the bytes of result.ppm), and with more
    than two frames in flight (more buffers than frames included)."""
    width = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, width, height, tile, frames, q, dtype, depth)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(got) == frames
    ys = np.arange(height, dtype=np.float32)
    for k, img in enumerate(got):
        assert img.shape == (height, width, 3)
        assert np.array_equal(img[:, 0, 0], ys)
        assert (img[:, :, 1] == k).all()
        assert np.array_equal(img[:, 0, 2], ((ys // tile) % world).astype(np.float32))


@pytest.mark.parametrize("depth,frames", [(2, 3), (3, 7), (8, 5)])
def test_pipeline_single_process(depth, frames):
    p = R.default_params(4, 6)
    pipe = tiles.GatherPipeline(p, 4, 3, "cpu", depth=depth)
    out = []
    for k in range(frames):
        buf = pipe.acquire()
        buf.fill_(k)
        img = pipe.submit()
        if img is not None:
            out.append(img)
    out += pipe.drain()
    assert [float(x[0, 0, 0]) for x in out] == [float(k) for k in range(frames)]


def _frames_worker(rank, world, port, width, height, tile, frames, q, kind, dtype, depth):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        base = R.default_params(width, height, row_tile=tile)
        mine = tiles.part_params(base, world, rank)
        lib = L.lib()
        rows = lib.rtamd_local_rows(C.byref(mine))
        if kind == "rotate":
            pipe = tiles.GatherPipeline(mine, width, 3, "cpu", dtype=dtype, depth=depth, rotate=True)
        else:
            pipe = tiles.AllToAllPipeline(mine, width, 3, "cpu", dtype=dtype, depth=depth)
        got = []
        for k in range(frames):
            buf = pipe.acquire()
            for lr in range(rows):
                buf[lr, :, 0] = lib.rtamd_global_row(C.byref(mine), lr)
                buf[lr, :, 1] = k
                buf[lr, :, 2] = rank
            img = pipe.submit()
            if img is not None:
                got.append(img.numpy().copy())
        got += [x.numpy().copy() for x in pipe.drain()]
        q.put((rank, got))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,world,height,tile,frames,dtype,depth", [
    ("rotate", 2, 37, 8, 5, torch.float32, 2), ("rotate", 3, 50, 4, 7, torch.float32, 4), ("rotate", 2, 21, 4, 3, torch.uint8, 8),
    ("all_to_all", 2, 37, 8, 6, torch.float32, 2), ("all_to_all", 3, 50, 4, 7, torch.float32, 2), ("all_to_all", 3, 16, 16, 1, torch.float32, 2),
    ("all_to_all", 2, 37, 8, 11, torch.uint8, 3), ("all_to_all", 3, 21, 4, 9, torch.float32, 8)])
def test_frames_delivered_on_rotating_roots(kind, world, height, tile, frames, dtype, depth):
    """Frame k whole on rank k % world, every rank's frames in order: by one gather per frame with a rotating root, or by
    one all-to-all per batch of `world` frames (a last batch may be partial)."""
    width = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_frames_worker, args=(r, world, port, width, height, tile, frames, q, kind, dtype, depth)) for r in range(world)]
    for p in procs:
        p.start()
    per_rank = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ys = np.arange(height, dtype=np.float32)
    for r in range(world):
        want = list(range(r, frames, world))
        assert [int(img[0, 0, 1]) for img in per_rank[r]] == want, (r, want)
        for img in per_rank[r]:
            assert img.shape == (height, width, 3)
            assert np.array_equal(img[:, 0, 0], ys)
            assert (img[:, :, 1] == img[0, 0, 1]).all()
            assert np.array_equal(img[:, 0, 2], ((ys // tile) % world).astype(np.float32))


@pytest.mark.parametrize("depth,frames", [(2, 3), (3, 7)])
def test_all_to_all_pipeline_single_process(depth, frames):
    p = R.default_params(4, 6)
    pipe = tiles.AllToAllPipeline(p, 4, 3, "cpu", depth=depth)
    out = []
    for k in range(frames):
        buf = pipe.acquire()
        buf.fill_(k)
        img = pipe.submit()
        if img is not None:
            out.append(img.clone())
    out += pipe.drain()
    assert [float(x[0, 0, 0]) for x in out] == [float(k) for k in range(frames)]
