"""Query-sharded data parallelism: reads are independent, the index is
replicated.  One process per GPU; ONE broadcast of the index image at start-up
(torch.distributed: backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests),
then no communication per batch.
"""
import torch
import torch.distributed as dist

from . import _native as N
from .index import GenieIndex


def shard_bounds(n_reads: int, rank: int, world: int):
    """Contiguous block partition: rank r owns reads [lo, hi)."""
    per = (n_reads + world - 1) // world
    lo = min(n_reads, rank * per)
    return lo, min(n_reads, lo + per)


def broadcast_image(image, src: int = 0, group=None, device=None):
    """Broadcast the serialized index image from `src`; returns this rank's copy.
    `image` is the uint8 tensor on `src` (ignored elsewhere); `device` is where copies live."""
    rank = dist.get_rank(group)
    device = torch.device(device) if device is not None else (image.device if image is not None else None)
    nbytes = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == src:
        nbytes[0] = image.numel()
    dist.broadcast(nbytes, src=src, group=group)
    if rank == src:
        buf = image.to(device)
    else:
        buf = torch.empty(int(nbytes.item()), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src=src, group=group)
    return buf


def broadcast_index(index, src: int = 0, group=None, device=None):
    """Rank `src` passes its GenieIndex; every rank gets a GenieIndex bound to its own device."""
    rank = dist.get_rank(group)
    device = torch.device(device)
    if rank == src:
        if index.blob is None or index.device != device:
            index.to(device)
        broadcast_image(index.blob, src, group, device)
        return index
    buf = broadcast_image(None, src, group, device)
    return GenieIndex.from_image(buf)


def header_of(image):
    """The fixed-size header bytes of an image (host tensor)."""
    return image[:N.HEADER_BYTES].cpu()
