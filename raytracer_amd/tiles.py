"""Row-tile partition of a frame across the GPUs of one node and the one exchange that follows.

The reference splits rows over CPU threads (`y = tid; y += N`, src/flyscene.cpp:809).  Here the image
is cut into tiles of `row_tile` rows, tile k goes to rank k % world; every rank holds the whole
read-only scene and renders its tiles into a packed buffer.  The only exchange brings the packed
buffers of a frame together on one rank, in one of two forms:

* GatherPipeline: one gather per frame (torch.distributed.gather: on the nccl backend, i.e. RCCL, one
  grouped send/recv per peer) to a fixed rank, or to rank k % world for frame k.  Every frame costs
  every sender its share over ONE of its links (xGMI is point to point), whichever rank is the root.
* AllToAllPipeline: frames in batches of `world`; frame b * world + k is delivered whole on rank k by
  ONE all-to-all per batch, in which every rank sends its shares of the batch's frames over all its
  links at once.  The same bytes per frame, a `world`-th of the time on the wire.
"""
import contextlib
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


def quantize_ppm(rgb, out, count=None, stream=None):
    """out[i] = min(255, int(255 * rgb[i])) on the device: the bytes the reference's writePPMImage stores
    (utils/ppmIO.hpp:203).  rgb: float32 CUDA tensor, out: uint8 CUDA tensor, both contiguous."""
    n = rgb.numel() if count is None else int(count)
    if rgb.dtype != torch.float32 or out.dtype != torch.uint8 or not rgb.is_cuda or not out.is_cuda:
        raise ValueError("quantize_ppm wants a float32 and a uint8 CUDA tensor")
    if not rgb.is_contiguous() or not out.is_contiguous() or rgb.numel() < n or out.numel() < n:
        raise ValueError("quantize_ppm wants contiguous tensors of at least %d elements" % n)
    s = stream if stream is not None else torch.cuda.current_stream(rgb.device)
    L.check(L.lib().rtamd_quantize_ppm(C.c_void_p(rgb.data_ptr()), C.c_void_p(out.data_ptr()), C.c_uint64(n), C.c_void_p(s.cuda_stream)),
            "rtamd_quantize_ppm")
    return out


_layout_cache = {}


def layout(params, world):
    """(rows of every rank, largest of them, index) for the partition `params` describes over `world` ranks;
    index[y] = row of the part-major concatenation of PADDED blocks (rank r's rows start at r * rows_max)
    that holds image row y.  Cached: a frame loop must not pay a Python loop over the image rows per frame."""
    key = (params.height, params.row_tile, world)
    hit = _layout_cache.get(key)
    if hit is not None:
        return hit
    lib = L.lib()
    rows = []
    index_padded = torch.empty(params.height, dtype=torch.long)
    index_packed = torch.empty(params.height, dtype=torch.long)
    for r in range(world):
        p = part_params(params, world, r)
        rows.append(lib.rtamd_local_rows(C.byref(p)))
    rows_max = max(rows) if rows else 0
    base = 0
    for r in range(world):
        p = part_params(params, world, r)
        for lr in range(rows[r]):
            y = lib.rtamd_global_row(C.byref(p), lr)
            index_padded[y] = r * rows_max + lr
            index_packed[y] = base + lr
        base += rows[r]
    assert base == params.height, (base, params.height)
    out = (rows, rows_max, index_padded, index_packed, {})
    _layout_cache[key] = out
    return out


def row_permutation(params, world, device="cpu"):
    """index[y] = row of the part-major concatenation (rank 0's rows, then rank 1's, ...) that holds
    image row y."""
    return layout(params, world)[3].to(device)


def gather_rows(local, params, group=None, dst=0):
    """Gather every rank's packed rows (`local`: [rows, width, ch], or already padded to the largest row count
    of any rank) on `dst` and return the assembled [height, width, ch] image there (None elsewhere).  One
    collective, no other traffic; the root restores row order in one pass over the gathered buffer."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        return local[: params.height]
    rows, rows_max, index_padded, _, on_device = layout(params, world)
    shape = (rows_max,) + tuple(local.shape[1:])
    send = local
    if local.shape[0] != rows_max:  # pad so that every rank contributes the same size
        send = local.new_zeros(shape)
        send[: local.shape[0]] = local
    send = send.contiguous()
    if rank == dst:
        key = (str(local.device), local.dtype, shape)
        cached = on_device.get(key)
        if cached is None:
            cached = (torch.empty((world,) + shape, dtype=local.dtype, device=local.device), index_padded.to(local.device))
            on_device[key] = cached
        recv_all, index = cached
        dist.gather(send, list(recv_all.unbind(0)), dst=dst, group=group)
        return recv_all.view((world * rows_max,) + tuple(local.shape[1:])).index_select(0, index)
    dist.gather(send, None, dst=dst, group=group)
    return None


class GatherPipeline:
    """Frames of a render loop in `depth` buffers: frame k's gather runs (asynchronously, on the collective's own
    stream) while the next frames are rendered into the other buffers.  Usage, every rank alike:

        pipe = GatherPipeline(params, width, channels, device, depth=D)
        for k in range(frames):
            buf = pipe.acquire()          # waits (on the stream, not the host) until the gather D frames back is done
            render into buf[:local_rows]
            image = pipe.submit()         # starts this frame's gather; returns the frame finished by acquire(), if any
        last = pipe.drain()               # list of the frames still in flight, oldest first

    Images are returned on rank `dst` only (None elsewhere) and lag the submissions by up to `depth` frames.
    acquire() and submit() take the stream the frame is rendered on (default: the current stream), so that several
    frames may also be TRACED at the same time, each on a stream of its own: buffer k % D goes with stream k % D.
    `rehearse` (one process only) puts a device copy on a side stream where the collective would be, with the same
    waits between the streams: the stream topology of a several-GPU run on one GPU."""

    def __init__(self, params, width, channels, device, dtype=torch.float32, group=None, dst=0, depth=2, rehearse=False, collective_alone=False,
                 rotate=False):
        """rotate: frame k is gathered on rank (dst + k) % world instead of always on `dst` (every rank then holds
        receive buffers and returns the frames it was the root of, in order)."""
        self.params, self.group, self.dst = params, group, dst
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rotate = bool(rotate) and self.world > 1
        self.frames = 0
        # collective_alone: a group of ONE process still goes through the collective (a test of the plumbing around it)
        self.single = self.world == 1 and not (collective_alone and dist.is_initialized())
        self.depth = max(2, int(depth))
        rows, rows_max, index_padded, _, _ = layout(params, self.world)
        self.rows_max = max(rows_max, 1)
        shape = (self.rows_max, width, channels)
        self.send = [torch.zeros(shape, dtype=dtype, device=device) for _ in range(self.depth)]
        self.recv = None
        if not self.single and (self.rank == dst or self.rotate):
            self.recv = [torch.empty((self.world,) + shape, dtype=dtype, device=device) for _ in range(self.depth)]
            self.index = index_padded.to(device)
        self.root = [dst] * self.depth  # root of the frame that buffer b holds
        self.side = None
        if self.single and rehearse:
            self.side = torch.cuda.Stream(device)
            self.recv = [torch.empty(shape, dtype=dtype, device=device) for _ in range(self.depth)]
        self.work = [None] * self.depth
        self.turn = 0
        self.done = None

    def _finish(self, b):
        """Frame of buffer b: wait for its gather and put the rows in image order."""
        if self.work[b] is None:
            return None
        if self.work[b] is not True:
            self.work[b].wait()  # a collective's handle or the rehearsal's event: the current stream waits, not the host
        self.work[b] = None
        if self.single:
            return (self.recv if self.side is not None else self.send)[b][: self.params.height].clone()
        if self.rank != self.root[b]:
            return None
        flat = self.recv[b].view((self.world * self.rows_max,) + tuple(self.send[b].shape[1:]))
        return flat.index_select(0, self.index)

    @staticmethod
    def _on(stream):
        return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()

    def acquire(self, stream=None):
        with self._on(stream):  # the wait for the buffer's previous gather, and its assembly, go to the frame's stream
            self.done = self._finish(self.turn)
        return self.send[self.turn]

    def submit(self, stream=None):
        b = self.turn
        with self._on(stream):  # the collective starts behind what this stream has enqueued
            if self.single:
                self.work[b] = True
                if self.side is not None:
                    self.side.wait_stream(torch.cuda.current_stream(self.send[b].device))
                    with torch.cuda.stream(self.side):
                        self.recv[b].copy_(self.send[b])
                        self.work[b] = torch.cuda.Event()
                        self.work[b].record(self.side)
            else:
                root = (self.dst + self.frames) % self.world if self.rotate else self.dst
                self.root[b] = root
                if self.rank == root:
                    self.work[b] = dist.gather(self.send[b], list(self.recv[b].unbind(0)), dst=root, group=self.group, async_op=True)
                else:
                    self.work[b] = dist.gather(self.send[b], None, dst=root, group=self.group, async_op=True)
        self.frames += 1
        self.turn = (self.turn + 1) % self.depth
        out, self.done = self.done, None
        return out

    def drain(self):
        out = [self._finish((self.turn + i) % self.depth) for i in range(self.depth)]
        return [x for x in out if x is not None]


class AllToAllPipeline:
    """Frames of a render loop in batches of `world`: every rank renders its rows of each frame of a batch into one
    chunk of a send buffer, and ONE all-to-all per batch delivers frame b * world + k whole on rank k.  Same usage as
    GatherPipeline, every rank alike:

        pipe = AllToAllPipeline(params, width, channels, device, depth=D)
        for k in range(frames):
            buf = pipe.acquire(stream)     # first frame of a batch: waits (on the stream) for the exchange D batches back
            render into buf[:local_rows]
            image = pipe.submit(stream)    # last frame of a batch: starts the batch's exchange
        rest = pipe.drain()                # exchanges a last, partial batch and returns the frames still under way

    Every rank is the root of one frame per batch and gets its frames back in order (rank r: frames r, r + world, ...).
    Why batches: xGMI links are point to point.  Gathering a 4K frame of float pixels, every sender moves its eighth
    (12.4 MB) over the one link to the root while its six other links idle, and a rank's collectives run one after the
    other; in the all-to-all the eight shares a rank holds for the batch's eight roots leave over seven links at once.
    `depth` = batches in flight (send and receive buffers of `world` chunks each)."""

    def __init__(self, params, width, channels, device, dtype=torch.float32, group=None, depth=2, rehearse=False, collective_alone=False):
        self.params, self.group = params, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.single = self.world == 1 and not (collective_alone and dist.is_initialized())
        self.depth = max(2, int(depth))
        self.cuda = torch.device(device).type == "cuda"
        rows, rows_max, index_padded, _, _ = layout(params, self.world)
        self.rows_max = max(rows_max, 1)
        shape = (self.world, self.rows_max, width, channels)
        self.send = [torch.zeros(shape, dtype=dtype, device=device) for _ in range(self.depth)]
        self.side = torch.cuda.Stream(device) if (self.single and rehearse and self.cuda) else None
        need_recv = not self.single or self.side is not None
        self.recv = [torch.empty(shape, dtype=dtype, device=device) for _ in range(self.depth)] if need_recv else None
        self.index = index_padded.to(device)
        self.work = [None] * self.depth     # the exchange of the batch in slot b: a collective's handle, an event, or True
        self.filled = [0] * self.depth      # frames that batch held
        self.free = [None] * self.depth     # event: slot b's previous exchange is over and its frame assembled
        self.rendered = [[] for _ in range(self.depth)]  # events behind the frames of the batch being rendered into slot b
        self.slot, self.k = 0, 0
        self.done = None

    _on = staticmethod(GatherPipeline._on)

    def _finish(self, b):
        if self.work[b] is None:
            return None
        if self.work[b] is not True:
            self.work[b].wait()  # the current stream waits, not the host
        self.work[b] = None
        if self.rank >= self.filled[b]:
            return None  # a last, partial batch held no frame for this rank
        src = self.recv[b] if self.recv is not None else self.send[b]
        flat = src.view((self.world * self.rows_max,) + tuple(src.shape[2:]))
        return flat.index_select(0, self.index)

    def acquire(self, stream=None):
        b = self.slot
        with self._on(stream):
            if self.k == 0:
                self.done = self._finish(b)
                if self.cuda:
                    self.free[b] = torch.cuda.Event()
                    self.free[b].record(torch.cuda.current_stream(self.send[b].device))
            elif self.cuda and self.free[b] is not None:  # the other frames of the batch write other chunks of the same buffer
                torch.cuda.current_stream(self.send[b].device).wait_event(self.free[b])
        return self.send[b][self.k]

    def _exchange(self, b):
        """Starts the exchange of the batch in slot b behind every frame rendered into it (called on the stream of the last)."""
        if self.cuda:
            cur = torch.cuda.current_stream(self.send[b].device)
            for ev in self.rendered[b]:
                cur.wait_event(ev)
        self.rendered[b] = []
        self.filled[b] = self.k
        if self.single:
            self.work[b] = True
            if self.side is not None:
                self.side.wait_stream(torch.cuda.current_stream(self.send[b].device))
                with torch.cuda.stream(self.side):
                    self.recv[b].copy_(self.send[b])
                    self.work[b] = torch.cuda.Event()
                    self.work[b].record(self.side)
        else:
            self.work[b] = dist.all_to_all_single(self.recv[b].view(-1), self.send[b].view(-1), group=self.group, async_op=True)
        self.slot, self.k = (self.slot + 1) % self.depth, 0

    def submit(self, stream=None):
        b = self.slot
        with self._on(stream):
            if self.cuda:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.send[b].device))
                self.rendered[b].append(ev)
            self.k += 1
            if self.k == self.world:
                self._exchange(b)
        out, self.done = self.done, None
        return out

    def drain(self, stream=None):
        with self._on(stream):
            if self.k > 0:
                self._exchange(self.slot)
            out = [self._finish((self.slot + i) % self.depth) for i in range(self.depth)]
        first, self.done = self.done, None
        return [x for x in [first] + out if x is not None]
