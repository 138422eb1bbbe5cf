#!/usr/bin/env python3
"""Where does the FLAT closest-hit traversal (lockstep walk over ALL internal nodes, 64-bit masks) stop paying against the
per-lane stack walk? Cornell with 0..N extra diffuse boxes (36 + 12 k triangles) at 1080p, option A/B on one scene object.
usage: python tools/flat_crossover.py [spp]"""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cudapathtracer_amd import api, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for extra in (0, 1, 2, 3, 4, 6):
    tmp = tempfile.mkdtemp()
    s = scenes.cornell(tmp, 1920, 1080, spp, 8, extra_boxes=extra, name="x%d" % extra)
    hs = api.HostScene(s["config"])
    row = {"extra_boxes": extra, "triangles": hs.info["n_tris"], "bvh_nodes": hs.info["n_nodes"]}
    for flat in (2, 0, 2, 0):
        sc = api.Scene(hs, options={"flat": flat})
        tiles = torch.zeros(api.n_tiles(1920, 1080), 64, 4, device="cuda")
        sc.render_tiles_device(hs.camera(), 1920, 1080, spp, 8, tiles.data_ptr())
        ms = sc.last_kernel_ms()
        f = sc.flags()
        key = "flat" if f["flat"] else ("stack_onchip" if f["onchip"] else "stack_general")
        row[key] = min(row.get(key, 1e9), round(ms, 2))
        row["simple"] = f["simple"]
        sc.close()
    print(json.dumps(row), flush=True)
