#!/usr/bin/env python3
"""A/B of the pair form of FLAT on a Cornell box with a mirror and a glass box (generic bounce): python tools/pair_generic_ab.py [spp]"""
import json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cudapathtracer_amd import api, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s = scenes.cornell(tempfile.mkdtemp(), 1920, 1080, spp, 8, tall_material=19, short_material=5, name="mix")
hs = api.HostScene(s["config"])
row = {"scene": "cornell, mirror + glass boxes", "spp": spp}
for flat2 in (1, 0, 1, 0):
    sc = api.Scene(hs, options={"flat2": flat2})
    tiles = torch.zeros(api.n_tiles(1920, 1080), 64, 4, device="cuda")
    sc.render_tiles_device(hs.camera(), 1920, 1080, spp, 8, tiles.data_ptr())
    ms = sc.last_kernel_ms()
    key = "pair" if sc.flags()["flat_pair"] else "single"
    row[key] = min(row.get(key, 1e9), round(ms, 2))
    sc.close()
print(json.dumps(row))
