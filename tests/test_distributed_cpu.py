"""The N>1 path on CPU: world_size-2 gloo processes shard the tile grid, fill their tile buffers
with a position-dependent pattern, run the path's single gather, and rank 0 de-interleaves."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _pattern(w, h):
    y, x = np.mgrid[0:h, 0:w]
    return np.stack([x, y, x * 1000 + y, np.ones_like(x)], -1).astype(np.float32)


def _worker(rank, world, port, w, h, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cudapathtracer_amd import distributed as D
    ids = D.tiles_of_rank(w, h, rank, world)
    tiles_x = (w + 7) // 8
    pat = _pattern(w, h)
    buf = np.zeros((D.padded_tile_count(w, h, world), 64, 4), np.float32)
    for k, t in enumerate(ids):                       # what the megakernel would write: [tile][ly*8+lx]
        x0, y0 = (t % tiles_x) * 8, (t // tiles_x) * 8
        blk = np.zeros((8, 8, 4), np.float32)
        ys, xs = min(8, h - y0), min(8, w - x0)
        blk[:ys, :xs] = pat[y0:y0 + ys, x0:x0 + xs]
        buf[k] = blk.reshape(64, 4)
    gathered = D.gather_tiles(torch.from_numpy(buf), w, h, rank, world)
    if rank == 0:
        np.save(out_path, D.assemble_host(gathered, w, h, world))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("w,h", [(64, 32), (37, 21)])
def test_gloo_world2_gather_reassembles_the_frame(tmp_path, w, h):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), w, h, out), nprocs=2, join=True)
    assert np.array_equal(np.load(out), _pattern(w, h))


def test_untile_host_matches_layout():
    from cudapathtracer_amd import distributed as D
    w, h = 19, 10
    ids = D.tiles_of_rank(w, h, 0, 1)
    buf = np.zeros((len(ids), 64, 4), np.float32)
    buf[:, :, 0] = np.arange(len(ids))[:, None]
    buf[:, :, 1] = np.arange(64)[None, :]
    img = D.untile_host(w, h, buf, ids)
    assert img[9, 18, 0] == 1 * 3 + 2 and img[9, 18, 1] == 1 * 8 + 2      # tile (2,1), lane ly=1,lx=2
