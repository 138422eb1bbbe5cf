#!/usr/bin/env python3
"""BVH build: the kept host builder (the reference's buildBVH on one core) next to pt_bvh_build_device
 on the same arrays; checks the two outputs are identical and prints one JSON line per scene."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cudapathtracer_amd import api, scenes  # noqa: E402
from tests.bvh_cases import arrays_of  # noqa: E402


def soup(n, seed=3):
    rng = np.random.default_rng(seed)
    c = (rng.random((n, 1, 3), np.float32) - 0.5) * 20
    return arrays_of(c + (rng.random((n, 3, 3), np.float32) - 0.5) * 0.05)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default="cornell,blob82k,atrium262k,soup1m,soup4m")
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--leaf", type=int, default=4)
    a = ap.parse_args()
    tmp = tempfile.mkdtemp()
    for name in a.scenes.split(","):
        if name.startswith("soup"):
            n = int(name[4:-1]) * (10**6 if name.endswith("m") else 10**3)
            pts, mesh = soup(n)
            leaf = a.leaf
        else:
            cfg = {"cornell": scenes.cornell, "blob82k": scenes.blob_in_box, "atrium262k": scenes.atrium}[name](tmp)["config"]
            hs = api.HostScene(cfg)
            pts, mesh, leaf = hs.array("points"), hs.array("mesh"), hs.info["leaf_size"]
            hs.close()
        t0 = time.perf_counter()
        hn, hi, hst = api.build_bvh(pts, mesh, leaf, where="host")
        host_ms = (time.perf_counter() - t0) * 1e3
        best = None
        for _ in range(a.repeat):
            dn, di, dst = api.build_bvh(pts, mesh, leaf, where="device")
            if best is None or dst["device_ms"] < best["device_ms"]:
                best = dst
        same = bool(np.array_equal(dn, hn) and np.array_equal(di, hi))
        n_tris = np.asarray(mesh).view(np.uint8).size // 80
        print(json.dumps({"scene": name, "triangles": n_tris, "nodes": hst["n_nodes"], "levels": best["levels"],
                          "host_ms": round(host_ms, 2), "host_builder_ms": round(hst["total_ms"], 2),
                          "device_ms": round(best["device_ms"], 3), "device_total_ms": round(best["total_ms"], 2),
                          "speedup_kernels": round(hst["total_ms"] / best["device_ms"], 1),
                          "Mtris_per_s_device": round(n_tris / best["device_ms"] / 1e3, 2),
                          "identical": same, "backups": best["backups"], "sort_fallbacks": best["sort_fallbacks"]}), flush=True)
        if not same:
            sys.exit(1)


if __name__ == "__main__":
    main()
