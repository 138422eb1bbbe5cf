#!/usr/bin/env python3
"""Lanes carried per trip through the traversal loops, from a -DPT_UTIL diagnostic build (PT_LIB_PATH).

usage: PT_LIB_PATH=.../lib_util.so python tools/lane_util.py [spp] [cornell|blob_in_box|atrium] [depth]
A wave walks its node loop as long as ANY lane is descending and its triangle loop as long as any lane is in
a leaf; the counting kernels of a PT_UTIL build count both the trips of the wave and the lanes that were active
in each. lanes_per_trip / 64 is the fraction of the vector unit's work in that loop that is useful.
"""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cudapathtracer_amd import api, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wl = sys.argv[2] if len(sys.argv) > 2 else "atrium"
depth = int(sys.argv[3]) if len(sys.argv) > 3 else (16 if wl == "atrium" else 8)
tmp = tempfile.mkdtemp()
s = getattr(scenes, wl)(tmp, width=1920, height=1080, spp=spp, max_depth=depth)
hs = api.HostScene(s["config"]); sc = api.Scene(hs)
tiles = torch.zeros(api.n_tiles(1920, 1080), 64, 4, device="cuda")
sc.reset_counters()
sc.render_tiles_device(hs.camera(), 1920, 1080, spp, depth, tiles.data_ptr(), count_work=True)
torch.cuda.synchronize()
out = {"scene": wl, "spp": spp, "depth": depth, "kernel_flags": sc.flags(), "lane_util": sc.debug_lane_util(), "counters": sc.counters()}
print(json.dumps(out))
