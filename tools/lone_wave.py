#!/usr/bin/env python3
"""One launch with about one wave per SIMD (1/32 of the 1080p tiles, one tile per wave: option persistent=0), for PMC passes."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cudapathtracer_amd import api, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
world = int(sys.argv[2]) if len(sys.argv) > 2 else 32
w, h = 1920, 1080
s = scenes.cornell(tempfile.mkdtemp(), width=w, height=h, spp=spp, max_depth=8)
hs = api.HostScene(s["config"]); sc = api.Scene(hs, options={"persistent": 0})
buf = torch.zeros(api.n_tiles(w, h), 64, 4, device="cuda")
sc.render_tiles_device(hs.camera(), w, h, spp, 8, buf.data_ptr(), tiles=api.rank_tiles(w, h, 0, world))
torch.cuda.synchronize()
print("kernel ms", sc.last_kernel_ms())
