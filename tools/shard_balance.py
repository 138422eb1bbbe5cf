#!/usr/bin/env python3
"""How well does 1/N of the frame (the tiles one rank of N renders) fill ONE GPU? Kernel time of the shard
vs. kernel time of the full frame / N — the single-GPU part of strong-scaling efficiency (no gather)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cudapathtracer_amd import api, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
wl = sys.argv[2] if len(sys.argv) > 2 else "cornell"
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 8
opts = api.parse_options(sys.argv[4:])                     # e.g. waves_hbm=2 slice_iters=256
w, h = 1920, 1080
s = getattr(scenes, wl)(tempfile.mkdtemp(), width=w, height=h, spp=spp, max_depth=depth)
hs = api.HostScene(s["config"]); sc = api.Scene(hs, options=opts)
print("# %s %d spp depth %d options %s" % (wl, spp, depth, opts))
cam = hs.camera()
buf = torch.zeros(api.n_tiles(w, h), 64, 4, device="cuda")
def run(world):
    tr = api.rank_tiles(w, h, 0, world)
    best = 1e9
    for _ in range(3):
        buf.zero_()
        sc.render_tiles_device(cam, w, h, spp, hs.info["max_depth"], buf.data_ptr(), tiles=tr)
        torch.cuda.synchronize()
        best = min(best, sc.last_kernel_ms())
    return best, sc.flags()
full, fl = run(1)
print("full frame %.1f ms" % full)
for n in (2, 4, 8):
    t, fl = run(n)
    print("hbm_kernel %s simple %s | " % (fl["hbm_kernel"], fl["simple"]), end="")
    print("1/%d of the tiles: %.1f ms = %.2fx of full/%d -> per-GPU efficiency %.1f %%, speedup bound %.2fx" % (n, t, t / (full / n), n, 100 * full / n / t, full / t))
