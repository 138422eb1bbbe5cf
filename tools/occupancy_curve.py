#!/usr/bin/env python3
"""Single-wave speed vs. full occupancy: kernel time of 1/N of the tiles (N = 32: about one wave per SIMD)
against full/N. One tile per wave (option persistent=0), so that the waves spread over the CUs as workgroups are dispatched."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cudapathtracer_amd import api, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
w, h = 1920, 1080
s = scenes.cornell(tempfile.mkdtemp(), width=w, height=h, spp=spp, max_depth=8)
hs = api.HostScene(s["config"]); sc = api.Scene(hs, options={"persistent": 0})
cam = hs.camera()
buf = torch.zeros(api.n_tiles(w, h), 64, 4, device="cuda")
def run(world):
    tr = api.rank_tiles(w, h, 0, world)
    best = 1e9
    for _ in range(3):
        buf.zero_()
        sc.render_tiles_device(cam, w, h, spp, 8, buf.data_ptr(), tiles=tr)
        torch.cuda.synchronize()
        best = min(best, sc.last_kernel_ms())
    return best
full = run(1)
print("full %.1f ms" % full)
for n in (8, 16, 32, 64):
    t = run(n)
    print("1/%d: %.1f ms; waves/SIMD ~%.2f; throughput vs full %.2f" % (n, t, 32400 / n / 1024, (full / n) / t))
