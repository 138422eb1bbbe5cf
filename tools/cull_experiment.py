#!/usr/bin/env python3
"""How often does opt-in box culling (pt_set_culling) change a frame on the SAME arithmetic? Renders 1920x1080 frames
exactly and culled, counts differing pixels and the work saved. usage: cull_experiment.py scene spp [depth]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cudapathtracer_amd import api, scenes
wl, spp = sys.argv[1], int(sys.argv[2])
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 8
s = getattr(scenes, wl)(tempfile.mkdtemp(), width=1920, height=1080, spp=spp, max_depth=depth)
hs = api.HostScene(s["config"]); sc = api.Scene(hs)
def frame(culled):
    sc.set_culling(culled)
    col, cnt = sc.render(hs.camera(), 1920, 1080, spp, depth, counters=True)
    sc.render(hs.camera(), 1920, 1080, spp, depth)
    return col, cnt, sc.last_kernel_ms()
a, ca, ta = frame(False)
b, cb, tb = frame(True)
diff = (a.view(np.uint32) != b.view(np.uint32)).any(axis=-1)
rays = int(ca[..., 0].sum() + ca[..., 1].sum())
print("%s %d spp depth %d: %.3g rays, %.3g triangle tests; culling: box tests %.3g -> %.3g, triangle tests -> %.3g, kernel %.1f -> %.1f ms; pixels differing: %d of %d" %
      (wl, spp, depth, rays, int(ca[..., 4].sum()), int(ca[..., 3].sum()), int(cb[..., 3].sum()), int(cb[..., 4].sum()), ta, tb, int(diff.sum()), diff.size))
