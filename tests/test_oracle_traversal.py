"""BVH traversal of the oracle vs the reference's own brute-force scan
(sceneIntersection, integratorUtilities.cuh:290-335), plus hand-built intersection cases."""
import os

import numpy as np

from conftest import golden_scene
from util import random_rays


def _check_vs_bruteforce(sc, rays):
    oi, of, cnt = sc.trace_closest(rays)
    bi, bf, _ = sc.trace_closest(rays, brute_force=True)
    assert np.array_equal(oi[:, 0], bi[:, 0])
    hit = oi[:, 0] == 1
    # equal-t ties may pick a different triangle (leaf order vs index order); t itself must agree
    assert np.array_equal(of[hit, 0], bf[hit, 0])
    same = oi[hit, 1] == bi[hit, 1]
    assert same.mean() > 0.99
    assert np.array_equal(of[hit][same], bf[hit][same])
    return cnt


def test_cornell_bvh_vs_bruteforce(oracle):
    sc = oracle.OracleScene(golden_scene("cornell32"))
    cnt = _check_vs_bruteforce(sc, random_rays(np.random.default_rng(3), 20000))
    assert cnt["rays_closest"] == 20000 and cnt["box_tests"] > 0


def test_blob_bvh_vs_bruteforce(oracle, scene_dir):
    from cudapathtracer_amd import scenes
    s = scenes.blob_in_box(os.path.join(scene_dir, "blob3"), 64, 36, 1, 4, subdiv=3, name="blob3")
    sc = oracle.OracleScene(s["config"])
    assert sc.info["n_tris"] == 20 * 4 ** 3 + 12
    _check_vs_bruteforce(sc, random_rays(np.random.default_rng(4), 3000))


def test_known_hits(oracle):
    sc = oracle.OracleScene(golden_scene("cornell32"))
    zf = 1.0 - 1.0 / np.tan(np.radians(30.0))
    # straight down the axis: back wall at z = zf - 2.5, facing the ray
    oi, of, _ = sc.trace_closest(np.array([[0, 0, 1, 0, 0, -1]], np.float32))
    assert oi[0, 0] == 1 and oi[0, 2] == 2 and oi[0, 3] == 0
    assert abs(of[0, 0] - (1.0 - (zf - 2.5))) < 1e-5 and np.allclose(of[0, 6:9], [0, 0, 1])
    # pointing out of the open front: miss
    oi, _, _ = sc.trace_closest(np.array([[0, 0, 0, 0, 0, 1]], np.float32))
    assert oi[0, 0] == 0
    # left wall is red (material 6), right wall green (23)
    oi, _, _ = sc.trace_closest(np.array([[0, 0.5, -1.0, -1, 0, 0], [0, 0.5, -1.0, 1, 0, 0]], np.float32))
    assert list(oi[:, 2]) == [6, 23]
    # from outside through a wall: back face, normal flipped to face the ray
    oi, of, _ = sc.trace_closest(np.array([[-5, 0.5, -1.0, 1, 0, 0]], np.float32))
    assert oi[0, 2] == 6 and oi[0, 3] == 1 and np.allclose(of[0, 6:9], [-1, 0, 0])


def test_shadow_rays(oracle):
    sc = oracle.OracleScene(golden_scene("cornell32"))
    rays = np.array([[0, 0, -1.0, 0, 1, 0], [0, 0, -1.0, 0, 1, 0], [0, 0, 0.5, 0, 0, 1]], np.float32)
    thr, _ = sc.trace_shadow(rays, np.array([5.0, 0.5, 5.0], np.float32))
    assert np.array_equal(thr[0], [0, 0, 0])      # ceiling (and light) in the way
    assert np.array_equal(thr[1], [1, 1, 1])      # max_t stops short
    assert np.array_equal(thr[2], [1, 1, 1])      # leaves through the open front
