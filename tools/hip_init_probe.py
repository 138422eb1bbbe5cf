"""Diagnostic: which HIP runtimes live in the process and does the first call into libptamd matter."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1] if len(sys.argv) > 1 else "api_first"
import numpy as np
if order == "api_first":
    from cudapathtracer_amd import api
    api.lib()
import torch
print("torch avail", torch.cuda.is_available()); torch.cuda.set_device(0)
from cudapathtracer_amd import api
L = api.lib()
maps = sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l or "hsa-runtime" in l})
print("\n".join(maps))
first = sys.argv[2] if len(sys.argv) > 2 else "build"
from tests.bvh_cases import cases
pts, mesh = cases()["tiny3"]
if first == "count":
    print("device_count", L.pt_device_count())
try:
    print("build", api.build_bvh(pts, mesh, 4)[2])
except Exception as e:
    print("build failed:", e)
print("device_count", L.pt_device_count())
try:
    print("build again", api.build_bvh(pts, mesh, 4)[2])
except Exception as e:
    print("build again failed:", e)
