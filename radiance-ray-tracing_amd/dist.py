"""dist.py -- image-tile sharding of a frame across the GPUs of one node and the framebuffer gather.

New functionality with no reference counterpart (the reference is single-device).
One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  The path shards
naturally -- every pixel is independent and its RNG streams depend only on the GLOBAL pixel index,
the frame id and the bounce depth (samples/shader.cl:205,523) -- so there is no data-path collective
while rendering.  The single exchange happens at frame end: every rank's owned tiles travel to rank 0
as one contiguous buffer (direct per-link transfers into the root; see DESIGN.md "Multi-GPU").

Tile ownership: tiles of tile_w x tile_h pixels are numbered row-major; tile t belongs to rank
t % world (interleaved, so cheap sky tiles and expensive geometry tiles spread evenly).  The packed
buffer of a rank lists its tiles in ascending id, each tile row-major and always tile_w*tile_h
entries long (border tiles are padded), which makes every rank's message the same shape up to one
tile -- `gather` pads to the maximum.
"""
import numpy as np


def tile_counts(width, height, tile_w, tile_h):
    return (width + tile_w - 1) // tile_w, (height + tile_h - 1) // tile_h


def owned_tile_ids(rank, world, width, height, tile_w, tile_h):
    tx, ty = tile_counts(width, height, tile_w, tile_h)
    return np.arange(rank, tx * ty, world, dtype=np.int64)


def max_owned_tiles(world, width, height, tile_w, tile_h):
    tx, ty = tile_counts(width, height, tile_w, tile_h)
    return (tx * ty + world - 1) // world


def owned_pixels(rank, world, width, height, tile_w, tile_h):
    """global pixel indices rendered by `rank`, in the order the device runtime enumerates them"""
    tx, _ = tile_counts(width, height, tile_w, tile_h)
    out = []
    for t in owned_tile_ids(rank, world, width, height, tile_w, tile_h):
        x0, y0 = (t % tx) * tile_w, (t // tx) * tile_h
        ys = np.arange(y0, min(height, y0 + tile_h))
        xs = np.arange(x0, min(width, x0 + tile_w))
        out.append((ys[:, None] * width + xs[None, :]).reshape(-1))
    return np.concatenate(out).astype(np.uint32) if out else np.zeros(0, np.uint32)


def _tile_index_map(rank, world, width, height, tile_w, tile_h):
    """(packed slot, image pixel) pairs of the in-image entries of rank's packed buffer"""
    tx, _ = tile_counts(width, height, tile_w, tile_h)
    slots, pixels = [], []
    for k, t in enumerate(owned_tile_ids(rank, world, width, height, tile_w, tile_h)):
        x0, y0 = (t % tx) * tile_w, (t // tx) * tile_h
        yy, xx = np.meshgrid(np.arange(tile_h), np.arange(tile_w), indexing="ij")
        ok = (y0 + yy < height) & (x0 + xx < width)
        slots.append((k * tile_w * tile_h + yy * tile_w + xx)[ok])
        pixels.append(((y0 + yy) * width + (x0 + xx))[ok])
    if not slots:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    return np.concatenate(slots), np.concatenate(pixels)


def pack_tiles_np(image, rank, world, width, height, tile_w, tile_h):
    """host equivalent of rdx_pack_tiles: image (H*W, C) -> packed (max-free, owned_tiles*tw*th, C)"""
    image = image.reshape(width * height, -1)
    n = len(owned_tile_ids(rank, world, width, height, tile_w, tile_h)) * tile_w * tile_h
    packed = np.zeros((n, image.shape[1]), image.dtype)
    s, p = _tile_index_map(rank, world, width, height, tile_w, tile_h)
    packed[s] = image[p]
    return packed


def unpack_tiles_np(packed, image, rank, world, width, height, tile_w, tile_h):
    image = image.reshape(width * height, -1)
    s, p = _tile_index_map(rank, world, width, height, tile_w, tile_h)
    image[p] = packed.reshape(-1, image.shape[1])[s]
    return image


def gather_to_root(packed, world, dst=0, recv=None):
    """One exchange per frame: every rank sends its packed tile buffer (a torch tensor, all ranks the
    same padded length) to `dst`.  Returns the list of per-rank tensors on dst (`recv` if given, so
    the landing buffers can be allocated once), None elsewhere.  With the nccl backend this is
    RCCL's gather: direct sends into the root over each peer's own xGMI link rather than a ring."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return [packed]
    rank = dist.get_rank()
    bufs = None
    if rank == dst:
        bufs = recv if recv is not None else [torch.empty_like(packed) for _ in range(world)]
    if packed.is_cuda and dist.get_backend() == "gloo":
        # rehearsal path (several ranks sharing one GPU, where RCCL refuses duplicate devices): stage
        # through host memory; the production backend is nccl = RCCL
        host = [torch.empty(packed.shape, dtype=packed.dtype) for _ in range(world)] if rank == dst else None
        dist.gather(packed.cpu(), host, dst=dst)
        if rank == dst:
            for b, h in zip(bufs, host):
                b.copy_(h)
        return bufs
    dist.gather(packed, bufs, dst=dst)
    return bufs


class FrameSharder:
    """Per-rank helper used by bench.py: owns the packed staging tensors (torch device memory wrapped
    once as RD buffers), packs after TraceRays, gathers, and unpacks on rank 0."""

    def __init__(self, rd, plt, width, height, rank, world, tile_w=64, tile_h=64, device=None):
        import torch
        self.rd, self.plt = rd, plt
        self.width, self.height, self.rank, self.world = width, height, rank, world
        self.tile_w, self.tile_h = tile_w, tile_h
        self.max_tiles = max_owned_tiles(world, width, height, tile_w, tile_h)
        n = self.max_tiles * tile_w * tile_h * 4
        # Two staging buffers, used alternately, each with an event recorded behind its gather: with the nccl backend
        # dist.gather only enqueues the send (torch's current stream waits for RCCL's stream, the host does not), while
        # rdx_pack_tiles runs on the library's own stream -- so before a buffer is packed again the host waits for the
        # event of the frame that last sent it (two frames ago: by then it has long completed), and a pending send can
        # never read the next frame's tiles.
        self.packed = [torch.zeros(n, dtype=torch.uint8, device=device) for _ in range(2)]
        self.packed_buf = [rd.WrapDeviceMemory(plt, t.data_ptr(), n, t) for t in self.packed]
        self.sent = [None, None]
        self.frame = 0
        self.recv, self.recv_bufs = None, None
        if rank == 0 and world > 1:
            self.recv = [torch.zeros(n, dtype=torch.uint8, device=device) for _ in range(world)]
            self.recv_bufs = [rd.WrapDeviceMemory(plt, t.data_ptr(), n, t) for t in self.recv]
        rd.SetShard(rank, world, tile_w, tile_h)

    def gather_image(self, image_buffer):
        """pack this rank's RGBA8 tiles, gather to rank 0, unpack there into `image_buffer`"""
        import torch
        from . import _lib
        if self.world == 1:
            return
        L, rd = _lib.lib(), self.rd
        b = self.frame & 1
        self.frame += 1
        if self.sent[b] is not None:
            self.sent[b].synchronize()          # the send that last read this buffer has completed
        if L.rdx_pack_tiles(image_buffer.handle, self.packed_buf[b].handle, self.width, self.height, 4, self.rank, self.world):
            raise rd.RadianceError(_lib.last_error())
        gather_to_root(self.packed[b], self.world, 0, self.recv)
        if self.packed[b].is_cuda:
            ev = torch.cuda.Event()
            ev.record()                         # on torch's current stream, which dist.gather made wait for the collective
            self.sent[b] = ev
        if self.rank == 0:
            torch.cuda.synchronize()
            import ctypes as C
            arr = (C.c_void_p * (self.world - 1))(*[self.recv_bufs[r].handle for r in range(1, self.world)])
            if L.rdx_unpack_tiles_multi(arr, 1, self.world - 1, image_buffer.handle, self.width, self.height, 4, self.world):
                raise rd.RadianceError(_lib.last_error())
