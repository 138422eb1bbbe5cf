#!/usr/bin/env python3
"""Where scene set-up time goes: loader (parse + host BVH), device BVH build, pt_scene_create (host re-pack + upload)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cudapathtracer_amd import api, scenes
torch.cuda.init()
for name, gen in (("blob82k", scenes.blob_in_box), ("atrium262k", scenes.atrium)):
    s = gen(tempfile.mkdtemp())
    t = time.perf_counter(); hs = api.HostScene(s["config"]); t_load = time.perf_counter() - t
    t = time.perf_counter(); hd = api.HostScene(s["config"], bvh_builder="device"); t_load_dev = time.perf_counter() - t
    t = time.perf_counter(); sc = api.Scene(hs); torch.cuda.synchronize(); t_create = time.perf_counter() - t
    t = time.perf_counter(); sc2 = api.Scene(hs); torch.cuda.synchronize(); t_create2 = time.perf_counter() - t
    t = time.perf_counter(); sm = api.Scene.from_mesh(hs); torch.cuda.synchronize(); t_mesh = time.perf_counter() - t
    t = time.perf_counter(); sm2 = api.Scene.from_mesh(hs); torch.cuda.synchronize(); t_mesh2 = time.perf_counter() - t
    _, _, st = api.build_bvh(hs.array("points"), hs.array("mesh"), hs.info["leaf_size"], where="host")
    print("   pt_scene_create_from_mesh (device build + device re-layout): %.1f ms (again %.1f ms; kernels %.2f ms) vs host builder %.0f ms + pt_scene_create %.0f ms" %
          (t_mesh * 1e3, t_mesh2 * 1e3, sm2.build_stats["device_ms"], st["total_ms"], t_create2 * 1e3))
    print("%s: %d tris; loader %.0f ms (host builder inside: %.0f ms), loader with device builder %.0f ms; pt_scene_create %.0f ms (again: %.0f ms)" %
          (name, hs.info["n_tris"], t_load * 1e3, st["total_ms"], t_load_dev * 1e3, t_create * 1e3, t_create2 * 1e3))
