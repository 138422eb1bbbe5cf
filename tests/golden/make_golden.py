#!/usr/bin/env python3
"""Regenerates tests/golden/*: the small scene files and the oracle's outputs on them.

    python tests/golden/make_golden.py

These are REGRESSION pins of this repo's CPU oracle (oracle/), not reference pins: the reference
ships no golden vectors and cannot be built here (oracle/README.md, "PARITY UNPINNED"). Each .npz
holds inputs (w, h, spp, max_depth, integrator, seed) and expected outputs: the per-pixel radiance
SUM (float32 [h,w,4], bit patterns) and the per-pixel work counters (uint32 [h,w,8]).
xorwow_kat.json holds the first 8 u32 outputs / uniforms of the streams SURVEY.md §8c lists.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from cudapathtracer_amd import scenes  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

CASES = [
    # name, scene kwargs, render kwargs
    ("cornell32_mis", dict(width=32, height=32, spp=8, max_depth=4, name="cornell32"), dict(integrator=0)),
    ("cornell32_naive", dict(width=32, height=32, spp=8, max_depth=4, name="cornell32"), dict(integrator=2)),
    ("cornell64_mis", dict(width=64, height=64, spp=4, max_depth=4, name="cornell64"), dict(integrator=0)),
    ("mixed32_mis", dict(width=32, height=32, spp=8, max_depth=6, tall_material=19, short_material=5, nested=True, name="mixed32"), dict(integrator=0)),
    ("mixed32_naive", dict(width=32, height=32, spp=8, max_depth=6, tall_material=19, short_material=5, nested=True, name="mixed32"), dict(integrator=2)),
    ("metal32_mis", dict(width=40, height=24, spp=8, max_depth=5, tall_material=4, short_material=7, name="metal32"), dict(integrator=0)),
    # textured Lambert (11, 12) + leaf materials (13, 16) with procedural BMP textures; own directory
    ("textured32_mis", dict(gen="textured", sub="scenes_tex", width=32, height=24, spp=8, max_depth=5, name="textured32"), dict(integrator=0)),
    ("textured32_naive", dict(gen="textured", sub="scenes_tex", width=32, height=24, spp=8, max_depth=5, name="textured32"), dict(integrator=2)),
]


def main():
    for name, skw, rkw in CASES:
        skw = dict(skw)
        sdir = os.path.join(HERE, skw.pop("sub", "scenes"))
        s = getattr(scenes, skw.pop("gen", "cornell"))(sdir, **skw)
        sc = O.OracleScene(s["config"])
        col, cnt, _ = sc.render(counters=True, threads=4, **rkw)
        i = sc.info
        np.savez_compressed(os.path.join(HERE, name + ".npz"), colors=col, counters=cnt, w=i["width"], h=i["height"], spp=i["spp"],
                            max_depth=i["max_depth"], integrator=rkw["integrator"], seed=103033, scene=skw["name"], scene_dir=os.path.basename(sdir),
                            scene_sha256=s["sha256"])
        print(name, "mean", col[..., :3].mean(), "rays", int(cnt[..., 0].sum() + cnt[..., 1].sum()))
    kat = {}
    for sub in (0, 1, 2, 3, 255, 65535, 2073599):
        st = O.xorwow_init(103033, sub)
        st2 = st.copy()
        kat[str(sub)] = {"state": [int(v) for v in st], "u32": [int(v) for v in O.xorwow_next(st, 8)],
                         "uniform_bits": [int(v) for v in O.xorwow_uniform(st2, 8).view(np.uint32)]}
    json.dump({"seed": 103033, "streams": kat}, open(os.path.join(HERE, "xorwow_kat.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
