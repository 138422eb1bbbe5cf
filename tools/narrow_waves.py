#!/usr/bin/env python3
"""Lone-wave iteration time vs. active lanes: the top 8, 4, 2, 1 pixel rows of
the Cornell view (240 tiles, far fewer than wave slots, so every wave runs alone on its SIMD). Same columns,
so the time ratio is the per-iteration cost of a wave with 64 / 32 / 16 / 8 live lanes."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cudapathtracer_amd import api, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 512
s = scenes.cornell(tempfile.mkdtemp(), width=1920, height=1080, spp=spp, max_depth=8)
hs = api.HostScene(s["config"]); sc = api.Scene(hs)
for rows in (8, 4, 2, 1):
    # same vertical field of view per pixel: keep fovScale, shrink the image to `rows` rows around the centre
    cam = hs.camera()                 # the camera keeps the 1920x1080 frame: rows 0..rows-1 of the real image
    w, h = 1920, rows
    buf = torch.zeros(api.n_tiles(w, h), 64, 4, device="cuda")
    best = 1e9
    for _ in range(3):
        buf.zero_()
        sc.render_tiles_device(cam, w, h, spp, 8, buf.data_ptr())
        torch.cuda.synchronize()
        best = min(best, sc.last_kernel_ms())
    print("rows %d (%2d lanes per wave): %.2f ms" % (rows, rows * 8, best))
