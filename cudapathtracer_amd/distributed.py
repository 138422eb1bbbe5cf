"""Screen-tile sharding across the GPUs of one node.

The reference is single-GPU. Every pixel's estimate depends only on (scene, camera, global pixel
index, seed), so the framebuffer shards with no data-path collective: rank r renders the
8x8-pixel tiles {t : t mod world == r} (interleaved for load balance) of a replicated scene into
a compact tile-major buffer, and ONE gather (RCCL over xGMI on GPUs, gloo in the CPU tests)
brings the buffers to rank 0, which de-interleaves them into scan-line order. Because the RNG
streams are keyed by the global index y*w+x the result is bit-identical for any world size.
"""
from __future__ import annotations

import numpy as np

from . import api


def tiles_of_rank(w, h, rank, world):
    """Global tile ids owned by `rank` (row-major grid of ceil(w/8) x ceil(h/8) tiles)."""
    return np.arange(rank, api.n_tiles(w, h), world, dtype=np.int64)


def padded_tile_count(w, h, world):
    """Equal per-rank count used for the gather (ranks with one tile fewer pad with zeros)."""
    return (api.n_tiles(w, h) + world - 1) // world


def untile_host(w, h, tile_buf, tile_ids, out=None):
    """NumPy de-interleave: tile_buf [n,64,4] for global tiles `tile_ids` -> colors [h,w,4]."""
    out = np.zeros((h, w, 4), np.float32) if out is None else out
    tiles_x = (w + 7) // 8
    tile_buf = np.asarray(tile_buf).reshape(-1, 8, 8, 4)
    for k, t in enumerate(tile_ids):
        x0, y0 = (int(t) % tiles_x) * 8, (int(t) // tiles_x) * 8
        xs, ys = min(8, w - x0), min(8, h - y0)
        out[y0:y0 + ys, x0:x0 + xs] = tile_buf[k, :ys, :xs]
    return out


def gather_tiles(local_tiles, w, h, rank, world, group=None):
    """The path's single collective: gather every rank's padded tile buffer on rank 0.

    local_tiles: torch tensor [padded_tile_count, 64, 4] float32 on this rank's device (cuda with
    the nccl/RCCL backend, cpu with gloo). Returns the list of per-rank tensors on rank 0, else None.
    """
    import torch
    import torch.distributed as dist

    if world == 1:
        return [local_tiles]
    if local_tiles.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal on a shared GPU / CPU tests: gloo gathers host tensors
        host = local_tiles.cpu()
        bufs = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
        dist.gather(host, gather_list=bufs, dst=0, group=group)
        return [b.to(local_tiles.device) for b in bufs] if rank == 0 else None
    bufs = [torch.empty_like(local_tiles) for _ in range(world)] if rank == 0 else None
    dist.gather(local_tiles, gather_list=bufs, dst=0, group=group)
    return bufs


def assemble_host(gathered, w, h, world):
    """Rank 0, host path: per-rank tile tensors -> scan-line [h,w,4] (used by the gloo tests)."""
    out = np.zeros((h, w, 4), np.float32)
    for r in range(world):
        ids = tiles_of_rank(w, h, r, world)
        untile_host(w, h, gathered[r].cpu().numpy()[: len(ids)], ids, out)
    return out


def assemble_device(gathered, w, h, world, colors, stream=0):
    """Rank 0, device path: pt_untile_device per rank into the scan-line `colors` tensor [h,w,4]."""
    for r in range(world):
        tr = api.rank_tiles(w, h, r, world)
        api.untile_device(w, h, gathered[r].data_ptr(), colors.data_ptr(), tr, stream)
    return colors
