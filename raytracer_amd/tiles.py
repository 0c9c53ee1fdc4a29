"""Row-tile partition of a frame across the GPUs of one node and the single gather that follows.

The reference splits rows over CPU threads (`y = tid; y += N`, src/flyscene.cpp:809).  Here the image
is cut into tiles of `row_tile` rows, tile k goes to rank k % world; every rank holds the whole
read-only scene and renders its tiles into a packed buffer.  The only exchange is one gather of the
packed buffers to rank 0 (torch.distributed.gather: on the nccl backend, i.e. RCCL, that is one
grouped send/recv per peer, so the root's xGMI links are used concurrently).
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib as L


def part_params(params, world, rank, row_tile=None):
    """Copy of `params` restricted to the rows of `rank`."""
    p = L.RenderParams()
    C.memmove(C.byref(p), C.byref(params), C.sizeof(p))
    if row_tile is not None:
        p.row_tile = int(row_tile)
    p.num_parts, p.part = int(world), int(rank)
    return p


def local_rows(params, world, rank):
    return L.lib().rtamd_local_rows(C.byref(part_params(params, world, rank)))


def row_permutation(params, world, device="cpu"):
    """index[y] = row of the part-major concatenation (rank 0's rows, then rank 1's, ...) that holds
    image row y."""
    lib = L.lib()
    height = params.height
    index = torch.empty(height, dtype=torch.long)
    base = 0
    for r in range(world):
        p = part_params(params, world, r)
        n = lib.rtamd_local_rows(C.byref(p))
        for lr in range(n):
            index[lib.rtamd_global_row(C.byref(p), lr)] = base + lr
        base += n
    assert base == height, (base, height)
    return index.to(device)


def gather_rows(local, params, group=None, dst=0):
    """Gather every rank's packed rows (`local`: [rows, width, ch]) on `dst` and return the assembled
    [height, width, ch] image there (None elsewhere).  One collective, no other traffic."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        return local[: params.height]
    rows_max = max(local_rows(params, world, r) for r in range(world))
    shape = (rows_max,) + tuple(local.shape[1:])
    send = local
    if local.shape[0] != rows_max:  # pad so that every rank contributes the same size
        send = local.new_zeros(shape)
        send[: local.shape[0]] = local
    send = send.contiguous()
    if rank == dst:
        recv = [torch.empty(shape, dtype=local.dtype, device=local.device) for _ in range(world)]
        dist.gather(send, recv, dst=dst, group=group)
        parts = [recv[r][: local_rows(params, world, r)] for r in range(world)]
        packed = torch.cat(parts, dim=0)
        return packed.index_select(0, row_permutation(params, world, packed.device))
    dist.gather(send, None, dst=dst, group=group)
    return None
