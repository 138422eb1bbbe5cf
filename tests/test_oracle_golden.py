"""The oracle against the committed golden fixtures (regression pins; see make_golden.py)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_case_scene, golden_cases, golden_scene, window_cases, window_scene
from util import assert_bits_equal

CASES = golden_cases()


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_golden(oracle, case):
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    sc = oracle.OracleScene(golden_case_scene(g))
    col, cnt, _ = sc.render(spp=int(g["spp"]), max_depth=int(g["max_depth"]), integrator=int(g["integrator"]),
                            seed=int(g["seed"]), counters=True, threads=2)
    assert_bits_equal(col, g["colors"], case)
    assert np.array_equal(cnt, g["counters"])


def test_region_and_thread_invariance(oracle):
    g = np.load(os.path.join(GOLDEN, "cornell32_mis.npz"))
    sc = oracle.OracleScene(golden_scene("cornell32"))
    col, _, _ = sc.render(integrator=0, rect=(8, 4, 24, 20), threads=3)
    assert_bits_equal(col[4:20, 8:24], g["colors"][4:20, 8:24])
    assert not col[:4].any() and not col[:, :8].any()


def test_accumulates_into_existing_colors(oracle):
    sc = oracle.OracleScene(golden_scene("cornell32"))
    a, _, _ = sc.render(spp=2, integrator=0)
    pre = np.full((32, 32, 4), 0.25, np.float32)
    b, _, _ = sc.render(spp=2, integrator=0, colors=pre.copy())
    # (0.25 + L0) + L1 in float32, per pixel, and w untouched
    assert np.array_equal(b[..., 3], pre[..., 3])
    assert np.allclose(b[..., :3], a[..., :3] + 0.25, rtol=1e-6, atol=1e-6)


def test_naive_and_mis_converge(oracle, scene_dir):
    """The reference's README (:70-92) argues both integrators estimate the same integral. With
    the whole ceiling emitting (no reachable emitter back side) they must agree statistically."""
    from cudapathtracer_amd import scenes
    s = scenes.cornell(os.path.join(scene_dir, "cl"), 16, 16, 4, 4, ceiling_light=True, name="cl")
    sc = oracle.OracleScene(s["config"])
    a, _, _ = sc.render(spp=384, max_depth=40, integrator=0, threads=4)
    b, _, _ = sc.render(spp=384, max_depth=40, integrator=2, threads=4)
    ma, mb = a[..., :3].mean(axis=(0, 1)), b[..., :3].mean(axis=(0, 1))
    assert np.all(np.abs(ma - mb) / ma < 0.02), (ma, mb)


def test_finalise(oracle):
    c = np.zeros((2, 2, 4), np.float32)
    c[0, 0] = [4, 8, 12, 0]; c[0, 1] = [np.nan, 1, 1, 0]; c[1, 0] = [np.inf, 1, 1, 0]
    f = oracle.finalise(c, 4).reshape(2, 2, 4)
    assert np.array_equal(f[0, 0, :3], [1, 2, 3]) and np.array_equal(f[0, 1, :3], [1, 0, 1]) and np.array_equal(f[1, 0, :3], [0, 1, 0])


def test_baseline_config0_host_loop(oracle, scene_dir):
    """BASELINE.json configs[0]: Cornell 256x256, 16 spp, 4 bounces through the single-thread host loop.
    Pinned by digest (a regression pin of the restatement, not a reference pin: oracle/README.md) and checked
    against the 8-thread run — the per-pixel streams make the image independent of the thread count."""
    import hashlib
    from cudapathtracer_amd import scenes
    s = scenes.cornell(os.path.join(scene_dir, "c1"), 256, 256, 16, 4, name="c1")
    assert s["sha256"] == "a8a5cd2e4b5eab82d1ef22e22ac7589bf78280791e05652bea0d4195c63bfc63"   # the generated scene itself
    sc = oracle.OracleScene(s["config"])
    a, ca, _ = sc.render(counters=True, threads=1)
    b, cb, _ = sc.render(counters=True, threads=8)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(ca, cb)
    assert int(ca[..., 0].sum()) == 4720774                                   # closest-hit rays of the frame
    assert hashlib.sha256(a.tobytes()).hexdigest() == "615b8397923e2b526aa64713ad524803550a64fb0f741e40b09348a2f8e73927"
    assert hashlib.sha256(ca.tobytes()).hexdigest() == "de43006ee22d1cbab2f3e305d5d450501b05f6458da2a0757b45355193e0e23c"


@pytest.mark.parametrize("case", window_cases())
def test_oracle_reproduces_window_fixtures(oracle, scene_dir, case):
    """64x64 windows of the real BASELINE scenes (82 k triangles at depth 8; 263 k triangles at depth 16) at the 1080p
    camera: the regenerated scene is the one the fixture was made from, and the oracle's `rect` render reproduces it."""
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    s = window_scene(g, os.path.join(scene_dir, case))
    assert s["sha256"] == str(g["scene_sha256"]), "scenes.py no longer generates the scene this fixture pins"
    sc = oracle.OracleScene(s["config"])
    assert sc.info["n_tris"] == int(g["n_tris"]) > 80000
    for k, (x0, y0, x1, y1) in enumerate(g["rects"]):
        col, cnt, _ = sc.render(rect=(int(x0), int(y0), int(x1), int(y1)), counters=True, threads=8, integrator=int(g["integrator"]))
        assert_bits_equal(col[y0:y1, x0:x1], g["colors"][k], "%s window %d" % (case, k))
        assert np.array_equal(cnt[y0:y1, x0:x1], g["counters"][k])
    if "atrium" in case and int(g["integrator"]) == 0:
        assert int(g["max_depth"]) == 16 and int(g["counters"][..., 7].max()) > 64      # paths long past the roulette depth
    if int(g["integrator"]) == 2:                                                       # Li_naive_unidirectional: maxDepth is a hard cap, no shadow rays
        assert int(g["counters"][..., 7].max()) == int(g["spp"]) * int(g["max_depth"]) and int(g["counters"][..., 1].sum()) == 0
