"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): integer indices / counters bit-exact; radiance within 1e-4
relative. Because both sides implement one fixed arithmetic the radiance is in fact
compared BIT-EXACTLY here; the 1e-4 tolerance is asserted separately as the contractual bound.
"""
import ctypes
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_case_scene, golden_cases, golden_scene, window_cases, window_scene
from util import assert_bits_equal, random_rays

pytestmark = pytest.mark.gpu

COUNTER_KEYS = ("rays_closest", "rays_shadow", "node_pops", "box_tests", "tri_tests", "hits")


def test_rng_streams(api, oracle, gpu_ready):
    subs = np.array([0, 1, 2, 3, 255, 65535, 2073599, 2**22 + 12345, 2**31 + 7, 2**32 - 1] + list(np.random.default_rng(0).integers(0, 2**21, 200)), np.uint32)
    st, u, f = api.probe_rng(subs, 16)
    for k, s in enumerate(subs):
        o = oracle.xorwow_init(103033, int(s))
        assert np.array_equal(st[k], o), s
        o2 = o.copy()
        assert np.array_equal(u[k], oracle.xorwow_next(o, 16))
        assert np.array_equal(f[k].view(np.uint32), oracle.xorwow_uniform(o2, 16).view(np.uint32))


def test_math_contract(api, oracle, gpu_ready):
    x = np.concatenate([np.linspace(0, 6.2832, 300001), np.linspace(-40, 40, 100001), np.random.default_rng(1).random(100000) * 1e-3]).astype(np.float32)
    g = api.probe_math(x)
    s, c = oracle.sincosf(x)
    assert_bits_equal(g["sin"], s, "sin"); assert_bits_equal(g["cos"], c, "cos")
    xe = np.linspace(-100, 95, 400001).astype(np.float32)
    assert_bits_equal(api.probe_math(xe)["exp"], oracle.expf(xe), "exp")
    xp = (np.random.default_rng(2).random(200000).astype(np.float32) * 50 + np.float32(1e-6))
    g = api.probe_math(xp)
    assert_bits_equal(g["rsqrt"], oracle.rsqrtf(xp), "rsqrt"); assert_bits_equal(g["pow5"], oracle.pow5(xp), "pow5")


def test_rcp_exact_equals_the_ieee_quotient_for_every_binary32_input(api, gpu_ready):
    """pt_device.h rcp_exact (v_rcp_f32 + one Newton step for 1e-12 <= |a| <= 1e30, IEEE division elsewhere) stands for
    `f = 1.0 / a` (integratorUtilities.cuh:22), 1 / dir (:50-55) and rsqrtf (util.cuh:129). Its bit-exactness is a property of
    this device's v_rcp_f32 table, so it is re-proven on every box that runs the suite: all 2^32 inputs, no mismatch."""
    r = api.probe_rcp_exhaustive()
    assert r["mismatches"] == 0 and r["first_bad"] == 0xffffffff, r
    lo, hi = np.float32(1e-12).view(np.uint32), np.float32(1.0e30).view(np.uint32)
    assert r["in_fast_range"] == 2 * (int(hi) - int(lo) + 1), r          # both signs of [1e-12, 1e30]: the fast arm covers what the kernels feed it
    assert r["bare_wrong_outside"] > 0, r                                 # ... and the range guard is not decoration


def test_camera_rays(api, oracle, gpu_ready):
    for cam in (api.Camera.Pinhole((0, 0, 1), 40, 24), api.Camera.NotPinhole((0.2, -0.1, 1.5), 40, 24, (5, 20, -3), 50.0, 0.05, 2.5),
                api.Camera.NotPinhole((0, 0, 1), 40, 24, (0, 0, 0), 60.0, 0.0, 1.0)):
        xy = np.array([(x, y) for y in range(24) for x in range(40)], np.int32)
        g = api.probe_camera_rays(cam, xy)
        cb = np.frombuffer(cam.tobytes(), np.uint8)
        o = np.stack([oracle.camera_ray(cb, int(x), int(y)) for x, y in xy])
        assert_bits_equal(g, o, "camera rays")


def _scene_pair(api, oracle, cfg, options=None):
    hs = api.HostScene(cfg)
    return api.Scene(hs, options=options), hs, oracle.OracleScene(cfg)


def _chain_arrays(api, n=50):
    """Hand-built left-deep chain BVH over n triangles stacked along -z (array-level boundary, the
    data model of main.cu:469-557): every internal node has the FARTHEST remaining triangle as its
    leaf child, so a ray from +z descends n-1 internal nodes with one far child pending at each
    level — deeper than the 32-entry LDS stack, which exercises the global spill path."""
    hs = api.HostScene(golden_scene("cornell32"))
    mats = hs.array("materials")
    pts = np.zeros((3 * n + 3, 4), np.float32)
    mesh = np.zeros((n + 1, 20), np.int32)
    meshf = mesh.view(np.float32)
    for k in range(n):
        z = -1.0 - 0.05 * k
        dx, dy = 0.03 * np.sin(k), 0.03 * np.cos(1.7 * k)
        pts[3 * k:3 * k + 3, :3] = [(-1 + dx, -1 + dy, z), (1 + dx, -1 + dy, z), (dx, 1.2 + dy, z)]
    pts[3 * n:3 * n + 3, :3] = [(-0.5, 2.0, -2.0), (0.5, 2.0, -2.0), (0.0, 2.0, -1.0)]        # one light triangle above
    for k in range(n + 1):
        mesh[k, 0:3] = [3 * k, 3 * k + 1, 3 * k + 2]
        mesh[k, 3:6] = 1 if k == n else 0
        mesh[k, 6:9] = 0
        mesh[k, 9] = [2, 6, 23, 17][k % 4] if k < n else 2
        mesh[k, 16], mesh[k, 17] = (0, k) if k == n else (-51, k)
    meshf[n, 12:15] = [5.0, 5.0, 5.0]
    normals = np.array([[0, 0, 1, 0], [0, -1, 0, 0]], np.float32)
    uvs = np.zeros((1, 2), np.float32)
    lo = pts[:, :3].reshape(-1, 3, 3).min(axis=1) - 1e-6
    hi = pts[:, :3].reshape(-1, 3, 3).max(axis=1) + 1e-6
    order = list(range(n + 1))                       # BVHindices: leaf i holds triangle order[i]
    nodes = np.zeros((2 * (n + 1) - 1, 12), np.float32)
    ni = nodes.view(np.int32)
    # node 0 = root over everything; chain: internal node j (j = 0..n-1) has left = leaf of the farthest remaining tri
    remaining = list(range(n + 1))                   # light (index n) is peeled first, then tri n-1, n-2, ...
    idx = 0
    for j in range(n):
        far = remaining.pop()                        # farthest remaining
        leaf_id, next_id = idx + 1, idx + 2
        nodes[idx, 0:3] = np.minimum(lo[remaining + [far]].min(axis=0), lo[far]); nodes[idx, 4:7] = hi[remaining + [far]].max(axis=0)
        ni[idx, 8:12] = [leaf_id, next_id, -1, 0]
        nodes[leaf_id, 0:3] = lo[far]; nodes[leaf_id, 4:7] = hi[far]
        ni[leaf_id, 8:12] = [-1, -1, far, 1]
        idx = next_id
    nodes[idx, 0:3] = lo[0]; nodes[idx, 4:7] = hi[0]
    ni[idx, 8:12] = [-1, -1, 0, 1]
    lights = mesh[n:n + 1].copy()
    return dict(points=pts, normals=normals, uvs=uvs, mesh=mesh, lights=lights, bvh=nodes, indices=np.array(order, np.int32), materials=mats)


@pytest.mark.parametrize("which", ["cornell32", "mixed32", "metal32", "blob3", "deep"])
def test_traversal_probes(api, oracle, gpu_ready, scene_dir, which):
    from cudapathtracer_amd import scenes
    if which == "blob3":
        cfg = scenes.blob_in_box(os.path.join(scene_dir, "blob3g"), 64, 36, 1, 4, subdiv=3, name="blob3")["config"]
    elif which == "deep":
        cfg = None
    else:
        cfg = golden_scene(which)
    if which == "deep":
        arr = _chain_arrays(api)
        gs, osc = api.Scene.from_arrays(arr), oracle.OracleScene(arrays=arr)
    else:
        gs, hs, osc = _scene_pair(api, oracle, cfg)
    rays = random_rays(np.random.default_rng(5), 8192)
    if which == "deep":
        rays[:, :3] = [0.0, 0.0, 1.0] + 0.2 * (np.random.default_rng(9).random((8192, 3)) - 0.5)
        rays[:, 3:] = np.array([0.0, 0.0, -1.0]) + 0.5 * (np.random.default_rng(6).random((8192, 3)) - 0.5)
        rays[::7, 3:] *= -1                          # some rays from behind / missing
    gi, gf, gc = gs.trace_closest(rays)
    oi, of, oc = osc.trace_closest(rays)
    assert np.array_equal(gi, oi)
    assert_bits_equal(gf, of, "closest hit records")
    assert {k: gc[k] for k in COUNTER_KEYS} == {k: oc[k] for k in COUNTER_KEYS}
    assert gi[:, 0].sum() > 100
    max_t = (np.random.default_rng(7).random(8192) * 4).astype(np.float32)
    gt, gc = gs.trace_shadow(rays, max_t)
    ot, oc = osc.trace_shadow(rays, max_t)
    assert_bits_equal(gt, ot, "shadow throughput")
    assert {k: gc[k] for k in COUNTER_KEYS} == {k: oc[k] for k in COUNTER_KEYS}


def test_bsdf_probes(api, oracle, gpu_ready):
    gs, hs, osc = _scene_pair(api, oracle, golden_scene("mixed32"))
    rng = np.random.default_rng(8)
    n = 3000
    mats = rng.choice([0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 13, 14, 16, 17, 18, 19, 20, 23], n).astype(np.int32)
    wi = rng.standard_normal((n, 3)); wi /= np.linalg.norm(wi, axis=1, keepdims=True); wi[:, 2] = -np.abs(wi[:, 2])
    wi = wi.astype(np.float32)
    back = rng.integers(0, 2, n).astype(np.int32)
    sub = rng.integers(0, 2**20, n).astype(np.uint32)
    g = gs.bsdf_sample(mats, wi, back, sub, eta_i=1.2)
    o = np.stack([osc.bsdf_sample(int(mats[k]), wi[k], bool(back[k]), eta_i=1.2, subseq=int(sub[k])) for k in range(n)])
    assert_bits_equal(g, o, "sample_f_eval")
    wo = rng.standard_normal((n, 3)); wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    wo = wo.astype(np.float32)
    g = gs.bsdf_eval(mats, wi, wo, eta_i=1.2)
    o = np.stack([osc.bsdf_eval(int(mats[k]), wi[k], wo[k], eta_i=1.2) for k in range(n)])
    assert_bits_equal(g, o, "f_eval / pdf_eval")


CASES = golden_cases()


@pytest.mark.parametrize("case", CASES)
def test_render_matches_golden(api, gpu_ready, case):
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    hs = api.HostScene(golden_case_scene(g))
    sc = api.Scene(hs)
    w, h = int(g["w"]), int(g["h"])
    col, cnt = sc.render(hs.camera(), w, h, int(g["spp"]), int(g["max_depth"]), integrator=int(g["integrator"]), seed=int(g["seed"]), counters=True)
    assert np.array_equal(cnt, g["counters"]), case                         # integer work counters: bit-exact
    assert_bits_equal(col, g["colors"], case)                               # radiance: bit-exact ...
    ref = g["colors"][..., :3]
    assert np.all(np.abs(col[..., :3] - ref) <= 1e-4 * np.abs(ref) + 1e-12)   # ... hence within the contractual 1e-4 relative
    tot = sc.counters()
    assert tot["rays_closest"] == int(g["counters"][..., 0].sum()) and tot["tri_tests"] == int(g["counters"][..., 4].sum())


@pytest.mark.parametrize("sched", [("0", "31", "1", "1", "1"), ("4", "3", "2", "1", "1"), ("8", "7", "0", "0", "2"), ("512", "31", "2", "0", "2"),
                                   ("4", "3", "2", "0", "0")])
def test_time_sliced_tile_queue(api, gpu_ready, sched):
    """The timed (counters-off) kernels with their scheduling machinery driven hard: tiles are yielded after
    4-8 bounce iterations, queued again and continued by whichever wave is free (production: 512), with and
    without issue-priority steering, through all three instantiations (LDS-resident; onchip=0: the
    6-waves-per-SIMD kernel for scenes in HBM, forced by waves_hbm=2 although these frames have few tiles;
    waves_hbm=0: the general 4-wave kernel). Scheduling and
    register budget must not reach the image: golden colours bit for bit."""
    opts = {k: int(v) for k, v in zip(("slice_iters", "sched_mask", "lpt_prio", "onchip", "waves_hbm"), sched)}
    for case in CASES:
        g = np.load(os.path.join(GOLDEN, case + ".npz"))
        hs = api.HostScene(golden_case_scene(g))
        sc = api.Scene(hs, options=opts)
        w, h = int(g["w"]), int(g["h"])
        for _ in range(2):                                  # second launch: queue re-initialised
            col, _ = sc.render(hs.camera(), w, h, int(g["spp"]), int(g["max_depth"]), integrator=int(g["integrator"]), seed=int(g["seed"]))
            assert_bits_equal(col, g["colors"], case)
        assert sc.last_kernel_ms() > 0.0                    # also reads the queue's error word
        sc.close()


def test_queue_waiters_that_give_up_do_not_fail_a_complete_frame(api, gpu_ready):
    """The tile queue's waits are bounded (a logic error must not hang the GPU): a waiter that sees no progress for
    "queue_timeout_ms" raises the queue's flag and every waiter leaves. Waiters hold no tile, so that is an error only if a tile
    is left unfinished — a stalled device (round 3: four persistent kernels co-resident on one device, a 30 s stall, every tile
    finished) costs the waiters, not the frame. With a timeout of zero the waiters leave at once: the frame is exact and the
    stall is counted; with time slices of four iterations on top, a tile yielded to a queue nobody waits on any more may stay
    unfinished, and THAT is reported (-4, never a silent partial frame)."""
    g = np.load(os.path.join(GOLDEN, "cornell64_mis.npz"))
    hs = api.HostScene(golden_case_scene(g))
    w, h, spp, md = int(g["w"]), int(g["h"]), int(g["spp"]), int(g["max_depth"])
    for opts in ({"queue_timeout_ms": 0}, {"queue_timeout_ms": 0, "onchip": 0, "waves_hbm": 0}, {"queue_timeout_ms": 0, "onchip": 0, "waves_hbm": 2}):
        sc = api.Scene(hs, options=opts)
        for _ in range(2):
            col, _ = sc.render(hs.camera(), w, h, spp, md)
            assert_bits_equal(col, g["colors"], "waiters gave up at once, %s" % opts)
        assert sc.queue_stalls() >= 1 and sc.tile_handovers() == 0, (opts, sc.queue_stalls())
        sc.close()
    outcomes = set()
    for opts in ({"queue_timeout_ms": 0, "slice_iters": 4, "sched_mask": 3}, {"queue_timeout_ms": 0, "slice_iters": 4, "sched_mask": 3, "onchip": 0, "waves_hbm": 2}):
        sc = api.Scene(hs, options=opts)
        try:
            col, _ = sc.render(hs.camera(), w, h, spp, md)
            assert_bits_equal(col, g["colors"], "complete although the waiters had left, %s" % opts)
            outcomes.add("complete")
        except api.PtError as e:
            assert "incomplete" in str(e) and "tiles finished" in str(e), e
            outcomes.add("incomplete")
        sc.close()
    assert outcomes <= {"complete", "incomplete"} and outcomes
    sc = api.Scene(hs)                                     # the default: 30 s, nobody gives up on a healthy device
    col, _ = sc.render(hs.camera(), w, h, spp, md)
    assert_bits_equal(col, g["colors"], "default timeout")
    assert sc.queue_stalls() == 0
    # the queue's header words, each in its own place (round 3: the wait bound once sat on the words the issue-priority steering
    # sums into, and a 1/8 share ran 12 % slower for it): everything claimed and finished, no stall, the steering's sums back at
    # zero, the bound where the waiters read it
    q = sc.queue_header()
    tiles = ((w + 7) // 8) * ((h + 7) // 8)
    assert q[0] >= tiles and q[1] >= tiles and q[2] == tiles and q[3] == 0, q
    assert q[4] == 0 and q[5] == 0, q
    assert (q[8] & 0xffffffff) | (q[9] << 32) == 30000 * 100000, q
    sc.close()
    sc = api.Scene(hs, options={"lpt_prio": 0, "queue_timeout_ms": 7})
    sc.render(hs.camera(), w, h, spp, md)
    q = sc.queue_header()
    assert q[2] == tiles and q[4] == 0 and q[5] == 0 and q[8] == 700000 and q[9] == 0, q
    sc.close()


@pytest.mark.parametrize("knobs", [("0", "4", "0", "0", "2", "0", "512", "0"),      # everything off: a wave stays in each loop until its last lane
                                   ("0", "4", "15", "15", "2", "0", "4", "2"),     # loops left as soon as ONE lane is through, shadow rays traced in the bounce
                                   ("1", "1", "8", "8", "2", "0", "4", "2"),       # production shape, traversal left only for the last sixteenth
                                   ("1", "15", "15", "1", "0", "0", "8", "1"),     # 4-wave kernel; traversal and node loop left at the first finished lane; speculation for closest-hit rays only
                                   ("1", "4", "1", "15", "2", "0", "0", "2"),      # no time slices
                                   ("1", "4", "8", "8", "2", "0", "4", "0"),       # resumable traversal without speculative descent
                                   ("1", "4", "0", "0", "2", "0", "16", "2"),      # speculative descent with the loop exits off
                                   ("2", "4", "8", "8", "1", "1", "4", "2")])      # REFILL also for the LDS-resident instantiation (A/B only)
def test_loop_exits_and_refill(api, gpu_ready, knobs):
    """pt_trace.h LoopExit / trace_resume: when a wave leaves its node loop, its triangle loop or the traversal
    (lanes that are through go on, the others resume later) is scheduling, not arithmetic — golden colours AND the
    per-pixel work counters bit for bit at every threshold, on all kernels (timed and counting instantiations)."""
    opts = {k: int(v) for k, v in zip(("refill", "refill_keep", "node_keep", "tri_keep", "waves_hbm", "onchip", "slice_iters", "spec"), knobs)}
    opts["sched_mask"] = 3
    if not api.has_experimental():                       # default library: no speculative descent, no REFILL for LDS-resident scenes
        del opts["spec"]
        if opts["refill"] == 2:
            pytest.skip("refill = 2 is an EXPERIMENTAL=1 build (DESIGN.md §6)")
    used = []
    for case in CASES:
        g = np.load(os.path.join(GOLDEN, case + ".npz"))
        hs = api.HostScene(golden_case_scene(g))
        sc = api.Scene(hs, options=opts)
        w, h = int(g["w"]), int(g["h"])
        col, _ = sc.render(hs.camera(), w, h, int(g["spp"]), int(g["max_depth"]), integrator=int(g["integrator"]), seed=int(g["seed"]))
        assert_bits_equal(col, g["colors"], case)
        used.append(sc.flags()["refill"])
        col, cnt = sc.render(hs.camera(), w, h, int(g["spp"]), int(g["max_depth"]), integrator=int(g["integrator"]), seed=int(g["seed"]), counters=True)
        assert np.array_equal(cnt, g["counters"]), case
        assert_bits_equal(col, g["colors"], case)
        assert sc.last_kernel_ms() > 0.0
        sc.close()
    assert any(used) == (knobs[0] != "0"), used             # the REFILL instantiation really ran (or really did not)


@pytest.mark.parametrize("integrator", [0, 2])
def test_flat_kernel_and_ties(api, oracle, gpu_ready, scene_dir, integrator):
    """pt_trace.h FLAT (scenes with at most 64 nodes / triangles): lockstep node walk + dealt-out triangle tests must
    give the stack walk's image bit for bit — also where two triangles return the SAME t and the reference keeps the
    one it visits first: the tall box exists twice, once diffuse and once as a mirror / as glass."""
    from cudapathtracer_amd import scenes
    cfgs = [scenes.cornell(os.path.join(scene_dir, "twin19"), 40, 24, 6, 6, doubled=19, name="twin19")["config"],
            scenes.cornell(os.path.join(scene_dir, "twin5"), 33, 17, 5, 8, doubled=5, tall_material=19, ceiling_light=True, name="twin5")["config"]]
    for cfg in cfgs:
        gs, hs, osc = _scene_pair(api, oracle, cfg)
        i = hs.info
        assert i["n_tris"] <= 64
        col, _ = gs.render(hs.camera(), i["width"], i["height"], i["spp"], i["max_depth"], integrator=integrator)
        assert gs.flags()["flat"] and gs.flags()["onchip"], gs.flags()
        ocol, ocnt, _ = osc.render(integrator=integrator, counters=True, threads=8)
        assert_bits_equal(col, ocol, cfg)
        col2, cnt = gs.render(hs.camera(), i["width"], i["height"], i["spp"], i["max_depth"], integrator=integrator, counters=True)   # the stack walk
        assert np.array_equal(cnt, ocnt) and not gs.flags()["flat"]
        assert_bits_equal(col2, ocol, cfg)
        gs0 = api.Scene(hs, options={"flat": 0})
        col3, _ = gs0.render(hs.camera(), i["width"], i["height"], i["spp"], i["max_depth"], integrator=integrator)
        assert not gs0.flags()["flat"]
        assert_bits_equal(col3, ocol, cfg)


def test_flat_kernel_single_leaf_scene(api, oracle, gpu_ready):
    """The whole scene is ONE leaf (root ref < 0, no internal node): the FLAT kernel's degenerate case."""
    arr = _chain_arrays(api, n=3)
    n = 4                                            # 3 stacked triangles + the light
    nodes = np.zeros((1, 12), np.float32)
    pts = arr["points"][:, :3].reshape(-1, 3, 3)
    nodes[0, 0:3] = pts.min(axis=(0, 1)) - 1e-6; nodes[0, 4:7] = pts.max(axis=(0, 1)) + 1e-6
    nodes.view(np.int32)[0, 8:12] = [-1, -1, 0, n]
    arr = dict(arr, bvh=nodes, indices=np.arange(n, dtype=np.int32))
    gs, osc = api.Scene.from_arrays(arr), oracle.OracleScene(arrays=arr)
    cam = api.Camera.Pinhole((0, 0, 1), 24, 16)
    col, _ = gs.render(cam, 24, 16, 4, 5)
    assert gs.flags()["flat"], gs.flags()
    ocol, ocnt, _ = osc.render(camera=np.frombuffer(cam.tobytes(), np.uint8), width=24, height=16, spp=4, max_depth=5, counters=True)
    assert_bits_equal(col, ocol, "single-leaf scene, FLAT kernel")
    col2, cnt = gs.render(cam, 24, 16, 4, 5, counters=True)
    assert np.array_equal(cnt, ocnt)
    assert_bits_equal(col2, ocol, "single-leaf scene, stack walk")
    assert ocnt[..., 5].sum() > 50


@pytest.mark.parametrize("integrator", [0, 2])
def test_render_fresh_scenes_vs_oracle(api, oracle, gpu_ready, scene_dir, integrator):
    from cudapathtracer_amd import scenes
    cfgs = [scenes.blob_in_box(os.path.join(scene_dir, "blob3r"), 45, 27, 3, 6, subdiv=3, name="blob3r")["config"],      # ragged size
            scenes.cornell(os.path.join(scene_dir, "cl2"), 24, 24, 6, 12, ceiling_light=True, tall_material=18, short_material=8, nested=True, name="cl2")["config"]]
    for cfg in cfgs:
        gs, hs, osc = _scene_pair(api, oracle, cfg)
        i = hs.info
        col, cnt = gs.render(hs.camera(), i["width"], i["height"], i["spp"], i["max_depth"], integrator=integrator, counters=True)
        ocol, ocnt, _ = osc.render(integrator=integrator, counters=True, threads=8)
        assert np.array_equal(cnt, ocnt), cfg
        assert_bits_equal(col, ocol, cfg)


@pytest.mark.parametrize("layout", ["queue", "one_tile_per_wave", "xcd_bands"])
@pytest.mark.parametrize("integrator", [0, 2])
def test_render_hand_built_deep_tree(api, oracle, gpu_ready, integrator, layout):
    """Array-level boundary + a tree deeper than the LDS stack, through the full render loop — in the 12-wave
    workgroups of the kernel for scenes in HBM (shared scene cache, per-wave spill areas), fed by the tile queue,
    one tile per wave (persistent=0) or in per-XCD bands."""
    opts = {"waves_hbm": 2}                              # the 6-wave kernel also for this 6-tile frame
    if layout == "one_tile_per_wave": opts["persistent"] = 0
    if layout == "xcd_bands":
        if not api.has_experimental():
            pytest.skip("xcd_bands is an EXPERIMENTAL=1 build (DESIGN.md §6)")
        opts["xcd_bands"] = 1
    arr = _chain_arrays(api)
    gs, osc = api.Scene.from_arrays(arr, options=opts), oracle.OracleScene(arrays=arr)
    cam = api.Camera.Pinhole((0, 0, 1), 24, 16)
    col, cnt = gs.render(cam, 24, 16, 4, 5, integrator=integrator, counters=True)
    ocol, ocnt, _ = osc.render(camera=np.frombuffer(cam.tobytes(), np.uint8), width=24, height=16, spp=4, max_depth=5, integrator=integrator, counters=True)
    assert np.array_equal(cnt, ocnt)
    assert_bits_equal(col, ocol, "deep chain render")
    assert cnt[..., 5].sum() > 100
    # counters off: the timed instantiation for scenes in HBM (8-entry LDS stack, so this tree spills 42 deep)
    col2, _ = gs.render(cam, 24, 16, 4, 5, integrator=integrator)
    assert_bits_equal(col2, ocol, "deep chain render, timed kernel")
    assert gs.flags()["hbm_kernel"] and not gs.flags()["onchip"]


@pytest.mark.parametrize("case", CASES)
def test_wavefront_variant_matches_golden(api, gpu_ready, case):
    """the stream-compacted variant must give the megakernel's image and counters."""
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    hs = api.HostScene(golden_case_scene(g))
    sc = api.Scene(hs).set_variant("wavefront")
    w, h = int(g["w"]), int(g["h"])
    col, cnt = sc.render(hs.camera(), w, h, int(g["spp"]), int(g["max_depth"]), integrator=int(g["integrator"]), seed=int(g["seed"]), counters=True)
    assert np.array_equal(cnt, g["counters"]), case
    assert_bits_equal(col, g["colors"], case)


def test_wavefront_variant_fresh_scenes(api, oracle, gpu_ready, scene_dir):
    from cudapathtracer_amd import scenes
    cfgs = [scenes.blob_in_box(os.path.join(scene_dir, "blob3w"), 45, 27, 3, 6, subdiv=3, name="blob3w")["config"],
            scenes.cornell(os.path.join(scene_dir, "cl3"), 24, 24, 6, 12, ceiling_light=True, tall_material=18, short_material=8, nested=True, name="cl3")["config"]]
    for cfg in cfgs:
        gs, hs, osc = _scene_pair(api, oracle, cfg)
        gs.set_variant("wavefront")
        i = hs.info
        col, cnt = gs.render(hs.camera(), i["width"], i["height"], i["spp"], i["max_depth"], counters=True)
        ocol, ocnt, _ = osc.render(counters=True, threads=8)
        assert np.array_equal(cnt, ocnt), cfg
        assert_bits_equal(col, ocol, cfg)
    arr = _chain_arrays(api)                       # deep tree: global stack spill inside the persistent trace kernel
    gs, osc = api.Scene.from_arrays(arr).set_variant("wavefront"), oracle.OracleScene(arrays=arr)
    cam = api.Camera.Pinhole((0, 0, 1), 24, 16)
    col, cnt = gs.render(cam, 24, 16, 4, 5, counters=True)
    ocol, ocnt, _ = osc.render(camera=np.frombuffer(cam.tobytes(), np.uint8), width=24, height=16, spp=4, max_depth=5, integrator=0, counters=True)
    assert np.array_equal(cnt, ocnt)
    assert_bits_equal(col, ocol, "deep chain, wavefront")
    tiles = np.zeros_like(col)                     # tile sharding works for the variant too
    for r in range(2):
        gs.render(cam, 24, 16, 4, 5, tiles=api.rank_tiles(24, 16, r, 2), out=tiles)
    assert_bits_equal(tiles, col, "wavefront tile sharding")


def test_tile_ranges_and_accumulation(api, gpu_ready):
    hs = api.HostScene(golden_scene("cornell64"))
    sc = api.Scene(hs)
    cam = hs.camera()
    full, _ = sc.render(cam, 64, 64, 4, 4)
    parts = np.zeros_like(full)
    for r in range(3):                                # three "ranks", interleaved tiles, same global streams
        sc.render(cam, 64, 64, 4, 4, tiles=api.rank_tiles(64, 64, r, 3), out=parts)
    assert_bits_equal(parts, full, "tile-sharded render")
    pre = np.full((64, 64, 4), 0.5, np.float32)
    acc, _ = sc.render(cam, 64, 64, 4, 4, out=pre.copy())
    assert np.array_equal(acc[..., 3], pre[..., 3]) and np.allclose(acc[..., :3], full[..., :3] + 0.5, rtol=1e-6, atol=1e-6)
    with pytest.raises(api.PtError):
        sc.render(cam, 64, 64, 1, 4, tiles=api.TileRange(60, 1, 10))
    with pytest.raises(api.PtError):
        sc.render(cam, 64, 64, 1, 4, integrator=3)    # VCM is out of scope


def test_reference_launcher_shape_and_init_render(api, oracle, gpu_ready):
    torch = gpu_ready
    cfg = golden_scene("cornell32")
    hs = api.HostScene(cfg)
    sc = api.Scene(hs)
    g = np.load(os.path.join(GOLDEN, "cornell32_mis.npz"))
    colors = torch.zeros(32, 32, 4, device="cuda")                          # out_colors, main.cu:337-339
    sc.launch_unidirectional(4, hs.camera(), 8, True, 32, 32, colors.data_ptr())
    assert_bits_equal(colors.cpu().numpy(), g["colors"], "pt_launch_unidirectional")
    assert sc.last_kernel_ms() > 0
    g2 = np.load(os.path.join(GOLDEN, "cornell32_naive.npz"))
    colors.zero_()
    sc.launch_naive_unidirectional(4, hs.camera(), 8, True, 32, 32, colors.data_ptr())
    assert_bits_equal(colors.cpu().numpy(), g2["colors"], "pt_launch_naive_unidirectional")
    img = api.init_render(cfg)
    assert_bits_equal(img, oracle.finalise(g["colors"], 8).reshape(32, 32, 4), "novum_init_render")


@pytest.mark.parametrize("variant", ["megakernel", "wavefront"])
def test_progressive_launcher(api, gpu_ready, tmp_path, variant):
    """chunked rendering with the preview hook equals the one-shot launcher bit for bit."""
    torch = gpu_ready
    g = np.load(os.path.join(GOLDEN, "mixed32_mis.npz"))
    hs = api.HostScene(golden_case_scene(g))
    sc = api.Scene(hs).set_variant(variant)
    colors = torch.zeros(32, 32, 4, device="cuda")
    seen = []
    sc.launch_progressive(0, int(g["max_depth"]), hs.camera(), int(g["spp"]), True, 32, 32, colors.data_ptr(), 3, lambda done: seen.append(done))
    assert seen == [3, 6, 8]
    assert_bits_equal(colors.cpu().numpy(), g["colors"], "progressive == one-shot")
    colors.zero_()
    sc.launch_progressive(0, int(g["max_depth"]), hs.camera(), int(g["spp"]), True, 32, 32, colors.data_ptr(), 3, lambda done: done >= 6)
    part = colors.cpu().numpy()
    assert not np.array_equal(part, g["colors"]) and np.isfinite(part[..., :3]).all()      # stopped after 6 of 8 samples


def test_init_render_with_preview_files(api, oracle, gpu_ready, tmp_path):
    g = np.load(os.path.join(GOLDEN, "cornell32_mis.npz"))
    bmp, pbmp, pcsv = str(tmp_path / "final.bmp"), str(tmp_path / "render.bmp"), str(tmp_path / "renderCSV.csv")
    img = api.init_render(golden_case_scene(g), bmp_path=bmp, preview_bmp=pbmp, preview_csv=pcsv, interval_seconds=0.0, chunk_spp=4)
    assert_bits_equal(img, oracle.finalise(g["colors"], 8).reshape(32, 32, 4), "init_render with previews")
    ref_bmp = str(tmp_path / "oracle.bmp")
    oracle.save_bmp(ref_bmp, oracle.finalise(g["colors"], 8).reshape(32, 32, 4), post_process=True)     # config says "Post Process: true"
    for f in (bmp, pbmp):                                                   # the last preview holds all 8 samples
        b = open(f, "rb").read()
        assert b[:2] == b"BM" and len(b) == 54 + 32 * 32 * 3 and int.from_bytes(b[18:22], "little") == 32
        assert b == open(ref_bmp, "rb").read(), "BMP bytes differ from the oracle's saveImageBMP"
    ref_csv = str(tmp_path / "oracle.csv")
    oracle.save_csv_mono(ref_csv, oracle.finalise(g["colors"], 8).reshape(32, 32, 4), 0)
    assert open(pcsv, "rb").read() == open(ref_csv, "rb").read()
    # "Post Process: false" (objects.cuh:844-943 key): the writer skips toneMap / gammaCorrect
    cfg2 = str(tmp_path / "nopost.rendertron")
    src = golden_case_scene(g)
    open(cfg2, "w").write(open(src).read().replace("Post Process: true", "Post Process: false"))
    bmp2 = str(tmp_path / "nopost.bmp")
    img2 = api.init_render(cfg2, base_dir=os.path.dirname(src), bmp_path=bmp2)
    assert_bits_equal(img2, img, "post-processing must not touch the returned radiance")
    oracle.save_bmp(ref_bmp, img2, post_process=False)
    assert open(bmp2, "rb").read() == open(ref_bmp, "rb").read()
    rows = open(pcsv).read().strip().split("\n")
    assert len(rows) == 32 and len(rows[0].split(",")) == 32 and "e" in rows[0].split(",")[0]
    # last preview = all 8 samples: CSV row 0 is image row y = 0 (bottom), channel 0, 3 significant decimals
    assert abs(float(rows[5].split(",")[7]) - img[5, 7, 0]) <= 5e-4 * max(1.0, abs(img[5, 7, 0]))


def test_device_tile_path_and_untile(api, gpu_ready):
    torch = gpu_ready
    hs = api.HostScene(golden_scene("cornell64"))
    sc = api.Scene(hs)
    g = np.load(os.path.join(GOLDEN, "cornell64_mis.npz"))
    frame = torch.zeros(64, 64, 4, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for r in range(2):
        tr = api.rank_tiles(64, 64, r, 2)
        tiles = torch.zeros(tr.count, 64, 4, device="cuda")
        sc.render_tiles_device(hs.camera(), 64, 64, 4, 4, tiles.data_ptr(), tiles=tr, stream=stream)
        api.untile_device(64, 64, tiles.data_ptr(), frame.data_ptr(), tr, stream)
    torch.cuda.synchronize()
    assert_bits_equal(frame.cpu().numpy(), g["colors"], "device tile path")


def test_baseline_config0_cornell256(api, oracle, gpu_ready, scene_dir):
    """BASELINE.json configs[0] — Cornell box 256x256, 16 spp, 4 bounces, the reference's CPU-runnable case — on the
    GPU against the host loop of the oracle: radiance and work counters bit for bit, both integrators' launchers."""
    from cudapathtracer_amd import scenes
    cfg = scenes.cornell(os.path.join(scene_dir, "c1"), 256, 256, 16, 4, name="c1")["config"]
    gs, hs, osc = _scene_pair(api, oracle, cfg)
    col, cnt = gs.render(hs.camera(), 256, 256, 16, 4, counters=True)
    ocol, ocnt, _ = osc.render(counters=True, threads=8)
    assert np.array_equal(cnt, ocnt)
    assert_bits_equal(col, ocol, "C1 Cornell 256x256x16spp")
    timed, _ = gs.render(hs.camera(), 256, 256, 16, 4)                       # the counters-off instantiation
    assert_bits_equal(timed, ocol, "C1, timed kernel")
    gs.set_variant("wavefront")
    wf, _ = gs.render(hs.camera(), 256, 256, 16, 4)
    assert_bits_equal(wf, ocol, "C1, wavefront variant")


@pytest.mark.parametrize("shape", [(1, 1, 3, 4), (7, 9, 2, 3), (33, 9, 1, 1), (16, 8, 5, 0), (8, 8, 0, 4), (130, 3, 2, 2)])
def test_edge_frames(api, oracle, gpu_ready, scene_dir, shape):
    """Frames smaller than a tile, ragged edges, one sample, zero samples, depth 0 and 1 — timed and counted kernels."""
    from cudapathtracer_amd import scenes
    w, h, spp, depth = shape
    cfg = scenes.cornell(os.path.join(scene_dir, "edge"), 16, 16, 2, 4, name="edge")["config"]
    gs, hs, osc = _scene_pair(api, oracle, cfg)
    cam = api.Camera.Pinhole((0, 0, 1), w, h)
    ocol, ocnt, _ = osc.render(camera=np.frombuffer(cam.tobytes(), np.uint8), width=w, height=h, spp=spp, max_depth=depth, counters=True)
    for integ in (0, 2):
        if integ == 2:
            ocol, ocnt, _ = osc.render(camera=np.frombuffer(cam.tobytes(), np.uint8), width=w, height=h, spp=spp, max_depth=depth, integrator=2, counters=True)
        col, cnt = gs.render(cam, w, h, spp, depth, integrator=integ, counters=True)
        assert np.array_equal(cnt, ocnt), (shape, integ)
        assert_bits_equal(col, ocol, "edge frame %s integrator %d" % (shape, integ))
        timed, _ = gs.render(cam, w, h, spp, depth, integrator=integ)
        assert_bits_equal(timed, ocol, "edge frame %s integrator %d, timed kernel" % (shape, integ))


def test_scene_without_lights_and_camera_that_sees_nothing(api, oracle, gpu_ready, scene_dir):
    """nextEventEstimation with lightNum == 0 draws nothing and adds nothing (deviceCode.cu:87-156), and every ray of a camera
    that looks away from the scene misses: black frames, but the RNG streams, the counters and the kernels' bookkeeping
    (pair pass with no shadow ray at all, waves whose paths all end at once) must still be the reference's."""
    from cudapathtracer_amd import scenes
    cfg = scenes.cornell(os.path.join(scene_dir, "dark"), 40, 24, 3, 6, light_mult=0.0, name="dark")["config"]
    gs, hs, osc = _scene_pair(api, oracle, cfg)
    i = hs.info
    assert i["n_lights"] == 0
    away = api.Camera.NotPinhole((0.0, 0.0, 1.0), i["width"], i["height"], (0.0, 180.0, 0.0), 60.0, 0.0, 1.0)
    for cam in (hs.camera(), away):
        for integ in (0, 2):
            ocol, ocnt, _ = osc.render(camera=np.frombuffer(cam.tobytes(), np.uint8), integrator=integ, counters=True)
            col, cnt = gs.render(cam, i["width"], i["height"], i["spp"], i["max_depth"], integrator=integ, counters=True)
            assert np.array_equal(cnt, ocnt), integ
            assert_bits_equal(col, ocol, "dark scene, integrator %d" % integ)
            timed, _ = gs.render(cam, i["width"], i["height"], i["spp"], i["max_depth"], integrator=integ)
            assert_bits_equal(timed, ocol, "dark scene, integrator %d, timed kernel" % integ)
            assert not np.any(ocol[..., :3]) and not np.any(ocnt[..., 1])      # black, and not one shadow ray


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_scenes(api, oracle, gpu_ready, scene_dir, seed):
    """Seeded random scenes (scenes.fuzz): random materials from the whole table on walls and objects, nested and
    interpenetrating dielectrics, leaves in front of lights, several emitters, random leaf size. Both integrators,
    counted and timed kernels, wavefront variant (where every material has a dispatch arm): bit for bit."""
    from cudapathtracer_amd import scenes
    cfg = scenes.fuzz(os.path.join(scene_dir, "fuzz%d" % seed), seed)["config"]
    gs, hs, osc = _scene_pair(api, oracle, cfg)
    i = hs.info
    for integ in (0, 2):
        ocol, ocnt, _ = osc.render(integrator=integ, counters=True, threads=8)
        col, cnt = gs.render(hs.camera(), i["width"], i["height"], i["spp"], i["max_depth"], integrator=integ, counters=True)
        assert np.array_equal(cnt, ocnt), (seed, integ)
        assert_bits_equal(col, ocol, "fuzz %d integrator %d" % (seed, integ))
        timed, _ = gs.render(hs.camera(), i["width"], i["height"], i["spp"], i["max_depth"], integrator=integ)
        assert_bits_equal(timed, ocol, "fuzz %d integrator %d, timed kernel" % (seed, integ))
    try:
        gs.set_variant("wavefront")
        wf, _ = gs.render(hs.camera(), i["width"], i["height"], i["spp"], i["max_depth"], integrator=0)
    except api.PtError as e:
        assert "dispatch arm" in str(e)                     # armless material types: the variant refuses, by design
    else:
        ocol, _, _ = osc.render(integrator=0, threads=8)
        assert_bits_equal(wf, ocol, "fuzz %d, wavefront" % seed)


def test_opt_in_culling(api, gpu_ready, scene_dir):
    """pt_set_culling is NOT the reference's visiting set (pt_api.h) and is off by default: at full scale it changes
    about 3 pixels per 1e9 rays (tools/cull_experiment.py). What this test pins is the plumbing: at the scale of the
    golden scenes, eight random scenes and a small frame of the 263 k-triangle atrium the image is the same bit for
    bit while fewer boxes are tested. Forced onto the kernel for scenes in HBM (the only one with the instantiation)."""
    from cudapathtracer_amd import scenes
    opts = {"onchip": 0, "waves_hbm": 2}
    cfgs = [golden_case_scene(np.load(os.path.join(GOLDEN, c + ".npz"))) for c in CASES]
    cfgs += [scenes.fuzz(os.path.join(scene_dir, "cfuzz%d" % k), k)["config"] for k in range(8)]
    cfgs.append(scenes.atrium(os.path.join(scene_dir, "catrium"), 96, 54, 4, 8, name="catrium")["config"])
    fewer = 0
    for cfg in cfgs:
        hs = api.HostScene(cfg)
        sc = api.Scene(hs, options=opts)
        i = hs.info
        exact, ce = sc.render(hs.camera(), i["width"], i["height"], i["spp"], i["max_depth"], counters=True)
        assert not sc.flags()["culling"]
        sc.set_culling(True)
        culled, cc = sc.render(hs.camera(), i["width"], i["height"], i["spp"], i["max_depth"], counters=True)
        timed, _ = sc.render(hs.camera(), i["width"], i["height"], i["spp"], i["max_depth"])
        assert sc.flags()["culling"]
        assert_bits_equal(culled, exact, "culled vs exact, %s" % cfg)
        assert_bits_equal(timed, exact, "culled timed kernel vs exact, %s" % cfg)
        assert np.array_equal(cc[..., [0, 1, 5, 6, 7]], ce[..., [0, 1, 5, 6, 7]])          # rays, hits, draws, iterations: unchanged
        assert np.all(cc[..., 3] <= ce[..., 3]) and np.all(cc[..., 4] <= ce[..., 4])        # boxes, triangles: never more
        fewer += int(ce[..., 3].sum() - cc[..., 3].sum())
        sc.close()
    assert fewer > 0


def test_thin_lens_camera_render(api, oracle, gpu_ready, scene_dir):
    """Camera::NotPinhole (objects.cuh:237-264): aperture 0.08, focal distance 2.2, rotated — through the full loop."""
    from cudapathtracer_amd import scenes
    cfg = scenes.cornell(os.path.join(scene_dir, "lens"), 48, 32, 6, 5, name="lens")["config"]
    gs, hs, osc = _scene_pair(api, oracle, cfg)
    cam = api.Camera.NotPinhole((0.15, -0.1, 1.2), 48, 32, (3.0, -8.0, 2.0), 55.0, 0.08, 2.2)
    col, cnt = gs.render(cam, 48, 32, 6, 5, counters=True)
    ocol, ocnt, _ = osc.render(camera=np.frombuffer(cam.tobytes(), np.uint8), width=48, height=32, spp=6, max_depth=5, counters=True)
    assert np.array_equal(cnt, ocnt)
    assert_bits_equal(col, ocol, "thin-lens render")


def test_full_frame_scheduling_invariance(api, gpu_ready, scene_dir):
    """1920x1080 Cornell, 48 spp: the frame from one-tile-per-wave workgroups (no queue at all) against the
    persistent kernel forced to yield every 32 iterations — some 300 000 hand-overs of tile state between waves
    on all eight XCDs — and against the production setting. Bit for bit, twice (no run-to-run variation)."""
    from cudapathtracer_amd import scenes
    cfg = scenes.cornell(os.path.join(scene_dir, "c2s"), 1920, 1080, 48, 8, name="c2s")["config"]
    hs = api.HostScene(cfg)
    frames = []
    for env in ({"persistent": 0}, {"slice_iters": 32, "sched_mask": 7}, {}, {"slice_iters": 32, "sched_mask": 7, "onchip": 0, "waves_hbm": 2}):
        sc = api.Scene(hs, options=env)
        a, _ = sc.render(hs.camera(), 1920, 1080, 48, 8)
        b, _ = sc.render(hs.camera(), 1920, 1080, 48, 8)
        assert sc.last_kernel_ms() > 0.0
        assert_bits_equal(a, b, "run-to-run %s" % env)
        frames.append(a)
        sc.close()
    for f in frames[1:]:
        assert_bits_equal(f, frames[0], "scheduling reached the image")


def test_full_frame_kernels_agree_on_a_scene_in_hbm(api, gpu_ready, scene_dir):
    """BASELINE C3's geometry class (82 k triangles, tree in HBM) at 1920x1080, 3 spp, depth 6: the image must not depend
    on which instantiation rendered it — the production kernel (12-wave workgroups, loop exits, resumable traversal),
    the same with plain loops and the shadow ray inside the bounce, the 4-wave kernel, the counting kernel, the
    FLAT switch (a no-op here) and the wavefront variant. Plus the counter identities that need no oracle."""
    from cudapathtracer_amd import scenes
    cfg = scenes.blob_in_box(os.path.join(scene_dir, "b82"), 1920, 1080, 3, 6, name="b82")["config"]
    hs = api.HostScene(cfg)
    assert hs.info["n_tris"] > 80000
    frames = []
    for env in ({}, {"refill": 0, "node_keep": 0, "tri_keep": 0}, {"waves_hbm": 0}, {"refill": 1, "refill_keep": 12, "slice_iters": 64},
                {"persistent": 0}):
        sc = api.Scene(hs, options=env)
        a, _ = sc.render(hs.camera(), 1920, 1080, 3, 6)
        fl = sc.flags()
        assert not fl["onchip"] and not fl["flat"]
        assert fl["refill"] == (env.get("refill", 1) != 0) and fl["hbm_kernel"] == (env.get("waves_hbm", 1) != 0), (env, fl)
        frames.append(a)
        if not env:
            b, cnt = sc.render(hs.camera(), 1920, 1080, 3, 6, counters=True)          # counting kernel
            frames.append(b)
            assert int(cnt[..., 0].sum()) == int(cnt[..., 7].sum())                    # one closest-hit ray per bounce iteration
            assert np.all(cnt[..., 3] % 2 == 0) and np.all(cnt[..., 5] <= cnt[..., 0])  # boxes in pairs; hits <= rays
            w, _ = sc.set_variant("wavefront").render(hs.camera(), 1920, 1080, 3, 6)
            frames.append(w)
        sc.close()
    for f in frames[1:]:
        assert_bits_equal(f, frames[0], "the instantiation reached the image")
    bad = int((~np.isfinite(frames[0][..., :3])).any(axis=-1).sum())
    print("non-finite pixels:", bad)
    assert bad < 2073600 // 10000 and np.nansum(frames[0][..., :3]) > 0, bad      # (the reference's arithmetic has no guards either)


def test_full_size_properties(api, gpu_ready, scene_dir):
    """BASELINE C2 geometry at full 1920x1080 (2 spp): properties that need no oracle run."""
    from cudapathtracer_amd import scenes
    cfg = scenes.cornell(os.path.join(scene_dir, "c2"), 1920, 1080, 2, 8, name="c2")["config"]
    hs = api.HostScene(cfg)
    sc = api.Scene(hs)
    cam = hs.camera()
    a, ca = sc.render(cam, 1920, 1080, 2, 8, counters=True)
    b, _ = sc.render(cam, 1920, 1080, 2, 8)
    assert_bits_equal(a, b, "determinism")
    assert np.isfinite(a[..., :3]).mean() > 0.99999 and np.nanmin(a[..., :3]) >= 0.0
    halves = np.zeros_like(a)
    for r in range(8):
        sc.render(cam, 1920, 1080, 2, 8, tiles=api.rank_tiles(1920, 1080, r, 8), out=halves)
    assert_bits_equal(halves, a, "8-way tile sharding")
    assert np.all(ca[..., 7] == ca[..., 0])                                  # one closest-hit ray per loop iteration
    assert np.all(ca[..., 1] <= ca[..., 0]) and np.all(ca[..., 5] <= ca[..., 0])
    assert np.all(ca[..., 3] % 2 == 0)                                       # boxes are tested in pairs
    assert ca[..., 0].min() >= 2                                             # every pixel traced its 2 camera rays
    m = np.nanmean(a[..., :3]) / 2
    assert 0.05 < m < 5.0


def test_options_api(api, gpu_ready):
    """pt_set_option / pt_get_option: the explicit replacement of round 1's environment switches. Unknown names and
    values out of range are errors; a scene starts from the documented defaults whatever the process environment holds."""
    os.environ["PT_FLAT"] = "0"; os.environ["PT_CULL"] = "1"; os.environ["PT_ONCHIP"] = "0"     # must be ignored
    try:
        hs = api.HostScene(golden_scene("cornell32"))
        sc = api.Scene(hs)
    finally:
        for k in ("PT_FLAT", "PT_CULL", "PT_ONCHIP"):
            del os.environ[k]
    defaults = {"flat": 1, "onchip": 1, "waves_hbm": 1, "refill": 1, "refill_keep": 6, "node_keep": 10, "tri_keep": 8, "defer_shadow": 0,
                "slice_iters": 512, "slice_always": 1, "sched_mask": 31, "lpt_prio": 2, "persistent": 1, "xcd_bands": 0, "culling": 0, "spec": 2, "simple": 1, "flat2": 1, "leaf_boxes": 1, "wide": 0, "compact": 0, "wf_wide_wg": 1, "lean": 1, "queue_timeout_ms": 30000}
    assert {k: sc.get_option(k) for k in defaults} == defaults
    sc.render(hs.camera(), 32, 32, 1, 4)
    assert sc.flags()["flat"] and sc.flags()["onchip"] and not sc.flags()["culling"]
    for name, bad in (("no_such_option", 1), ("sched_mask", 6), ("node_keep", 16), ("waves_hbm", 3), ("flat", -1)):
        with pytest.raises(api.PtError):
            sc.set_option(name, bad)
    sc.set_options({"flat": 0, "sched_mask": 7, "waves_hbm": 2})
    assert (sc.get_option("flat"), sc.get_option("sched_mask"), sc.get_option("waves_hbm")) == (0, 7, 2)
    # variants that lost their A/B are not in the default library: only their "off" value is accepted (rc -3 otherwise)
    for name, on in (("wide", 1), ("compact", 1), ("defer_shadow", 1), ("xcd_bands", 1), ("refill", 2), ("spec", 1)):
        if api.has_experimental():
            sc.set_option(name, on)
        else:
            with pytest.raises(api.PtError, match="experimental"):
                sc.set_option(name, on)
        sc.set_option(name, defaults[name])


def _render_window(api, sc, cam, w, h, spp, md, rect, counters=False, integrator=0):
    """A 64x64 window = 8 runs of 8 consecutive tiles (pt_tile_range is an arithmetic progression of tile ids)."""
    x0, y0, x1, y1 = (int(v) for v in rect)
    tiles_x = (w + 7) // 8
    col = np.zeros((h, w, 4), np.float32)
    cnt = np.zeros((h, w, 8), np.uint32) if counters else None
    for ty in range(y0 // 8, y1 // 8):
        tr = api.TileRange(ty * tiles_x + x0 // 8, 1, (x1 - x0) // 8)
        if counters:
            rc = api.lib().pt_render_counted(sc.h, ctypes.byref(cam), w, h, spp, md, integrator, 1, api.SEED, ctypes.byref(tr), col.ctypes.data_as(ctypes.c_void_p),
                                             cnt.ctypes.data_as(ctypes.c_void_p))
            assert rc == 0, api.lib().pt_last_error()
        else:
            sc.render(cam, w, h, spp, md, tiles=tr, out=col, integrator=integrator)
    return col[y0:y1, x0:x1], (cnt[y0:y1, x0:x1] if counters else None)


@pytest.mark.parametrize("mode", ["production", "generic_bounce", "generic_bounce_all_arms", "plain_loops_4wave", "counted", "wavefront", "compact"])
@pytest.mark.parametrize("case", window_cases())
def test_real_scene_windows_vs_oracle(api, oracle, gpu_ready, scene_dir, case, mode):
    """BASELINE C3 / C4 / C5 geometry and depth at the 1080p camera, against the oracle: 64x64 windows of the 82 k-triangle
    scene (depth 8) and of the 263 k-triangle atrium (depth 16: BVHSceneIntersect stacks 27 deep, Russian roulette after 16
    bounces, paths of up to 100 iterations) — the production kernel for scenes in HBM (12-wave workgroups, loop exits,
    REFILL; forced although a window has few tiles), the 4-wave kernel with plain loops, the counting kernel (all eight
    per-pixel counters) and the wavefront variant (C5). Committed fixture AND a live oracle run of the first window."""
    if mode == "compact" and not api.has_experimental():
        pytest.skip("compact nodes are an EXPERIMENTAL=1 build (DESIGN.md §6)")
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    s = window_scene(g, os.path.join(scene_dir, case))
    assert s["sha256"] == str(g["scene_sha256"])
    hs = api.HostScene(s["config"])
    w, h, spp, md, integ = int(g["w"]), int(g["h"]), int(g["spp"]), int(g["max_depth"]), int(g["integrator"])
    opts = {"production": {"waves_hbm": 2}, "generic_bounce": {"waves_hbm": 2, "simple": 0}, "generic_bounce_all_arms": {"waves_hbm": 2, "simple": 0, "lean": 0}, "plain_loops_4wave": {"waves_hbm": 0, "refill": 0, "node_keep": 0, "tri_keep": 0},
            "counted": {"waves_hbm": 2}, "wavefront": {"wf_wide_wg": 2}, "compact": {"waves_hbm": 2, "compact": 1}}[mode]
    sc = api.Scene(hs, options=opts)
    if mode == "wavefront":
        sc.set_variant("wavefront")
    cam = hs.camera()
    for k, rect in enumerate(g["rects"]):
        col, cnt = _render_window(api, sc, cam, w, h, spp, md, rect, counters=(mode == "counted"), integrator=integ)
        assert_bits_equal(col, g["colors"][k], "%s window %d (%s)" % (case, k, mode))
        if mode == "counted":
            assert np.array_equal(cnt, g["counters"][k]), (case, k)
        if mode in ("production", "generic_bounce", "generic_bounce_all_arms", "compact"):
            fl = sc.flags()
            assert fl["hbm_kernel"] and fl["refill"] and not fl["onchip"] and not fl["culling"], fl
            assert fl["simple"] == (not mode.startswith("generic_bounce")), fl          # both scenes are diffuse-only: the SIMPLE bounce is what production runs
            assert fl["lean"] == (mode == "generic_bounce"), fl                         # ... and without it the generic bounce drops its leaf / texture arms (no such material here)
    x0, y0, x1, y1 = (int(v) for v in g["rects"][1])
    ocol, ocnt, _ = oracle.OracleScene(s["config"]).render(rect=(x0, y0, x1, y1), counters=True, threads=8, integrator=integ)
    assert_bits_equal(ocol[y0:y1, x0:x1], g["colors"][1], "fixture == live oracle")
    assert np.array_equal(ocnt[y0:y1, x0:x1], g["counters"][1])


def _oracle_tile(osc, w, h, spp, md, x0, y0, integrator=0):
    """The oracle on ONE 8x8 tile, its eight rows on eight host threads (a pixel's stream is keyed by its global index, so
    a row is independent of its neighbours; the oracle's own threading is per 8x8 tile and would leave seven cores idle)."""
    from concurrent.futures import ThreadPoolExecutor

    def row(k):
        col = np.zeros((h, w, 4), np.float32)
        osc.render(rect=(x0, y0 + k, x0 + 8, y0 + k + 1), spp=spp, max_depth=md, integrator=integrator, threads=1, colors=col)
        return col[y0 + k, x0:x0 + 8].copy()
    with ThreadPoolExecutor(8) as ex:
        return np.stack(list(ex.map(row, range(8))))


@pytest.mark.parametrize("config", ["C2_cornell_1024spp_depth8", "C3_blob82k_1024spp_depth8", "C4_atrium263k_4096spp_depth16", "C3_blob82k_naive_1024spp_depth8",
                                    "mixed_cornell_1024spp_depth8", "glass_blob82k_256spp_depth8"])
def test_deep_stream_one_tile_at_the_configs_own_sample_count(api, oracle, gpu_ready, scene_dir, config):
    """The reference carries ONE XORWOW stream per pixel through all samples of a render (state reloaded and stored around every
    sample, deviceCode.cu:294, 541, 568-573). The window fixtures stop at 4 spp; here ONE 8x8 tile (pt_tile_range of one tile)
    of each BASELINE config is rendered at the config's OWN sample count and depth through the production kernels and compared
    bit for bit with a live oracle run of the same 64 pixels: 64 streams of 1024 (4096) samples each, where rare events — the
    range guard of rcp_exact, stack spills, equal-t ties, NaN paths, roulette at depth 16 — have 256 (1024) times the windows'
    chances to occur, and where the tile changes hands between waves dozens of times (time slices of 512 iterations)."""
    from cudapathtracer_amd import scenes
    gen, kw, tile_xy, opts, integ = {
        "C2_cornell_1024spp_depth8": ("cornell", dict(spp=1024, max_depth=8), (1112, 624), {}, 0),              # the short box's edge against the floor
        "C3_blob82k_1024spp_depth8": ("blob_in_box", dict(spp=1024, max_depth=8), (1088, 640), {"waves_hbm": 2}, 0),   # the blob's silhouette
        "C4_atrium263k_4096spp_depth16": ("atrium", dict(spp=4096, max_depth=16), (960, 536), {"waves_hbm": 2}, 0),
        "C3_blob82k_naive_1024spp_depth8": ("blob_in_box", dict(spp=1024, max_depth=8), (896, 480), {"waves_hbm": 2}, 2),
        # the general bounce (bench.py's secondary workloads): mirror / glass with nested water / GGX gold on the glass box's edge; the glass blob
        "mixed_cornell_1024spp_depth8": ("cornell", dict(spp=1024, max_depth=8, tall_material=19, short_material=5, nested=True, extra_boxes=1, extra_materials=[4]),
                                         (1112, 624), {}, 0),
        "glass_blob82k_256spp_depth8": ("blob_in_box", dict(spp=256, max_depth=8, material=5), (1088, 640), {"waves_hbm": 2}, 0),
    }[config]
    w, h = 1920, 1080
    s = getattr(scenes, gen)(os.path.join(scene_dir, "deep_" + config), width=w, height=h, name="deep_" + gen, **kw)
    hs = api.HostScene(s["config"])
    sc = api.Scene(hs, options=opts)
    x0, y0 = tile_xy
    tile = (y0 // 8) * ((w + 7) // 8) + x0 // 8
    col = np.zeros((h, w, 4), np.float32)
    sc.render(hs.camera(), w, h, kw["spp"], kw["max_depth"], tiles=api.TileRange(tile, 1, 1), out=col, integrator=integ)
    fl = sc.flags()
    generic = config.startswith(("mixed", "glass"))
    assert fl["lean"] == generic, fl
    assert fl["time_slices"] and (fl["flat_pair"] if gen == "cornell" else (fl["hbm_kernel"] and fl["refill"])) and fl["simple"] == (not generic), fl
    assert sc.tile_handovers() >= 4, sc.tile_handovers()          # the one tile's state really travelled between waves
    want = _oracle_tile(oracle.OracleScene(s["config"]), w, h, kw["spp"], kw["max_depth"], x0, y0, integ)
    got = col[y0:y0 + 8, x0:x0 + 8]
    assert np.isfinite(want[..., :3]).all() and float(want[..., :3].sum()) > 0.0
    assert_bits_equal(got, want, config)
    outside = col.copy(); outside[y0:y0 + 8, x0:x0 + 8] = 0
    assert not outside.any()                                       # nothing outside the tile was touched


@pytest.mark.parametrize("config", ["C2_cornell", "C3_blob82k", "C4_atrium263k", "mixed_cornell", "glass_blob82k", "C2_cornell_naive", "C3_blob82k_naive"])
def test_full_1080p_frame_vs_oracle(api, oracle, gpu_ready, scene_dir, config):
    """Every pixel of the 1920x1080 frame of each BASELINE scene — all 32 400 tiles through the production kernels in ONE launch,
    tile queue, time slices, scheduling and all — against a live oracle render of the whole frame at a low sample count (the
    per-pixel streams at full length are the deep-stream tests' subject): 2 073 600 pixels bit for bit, and the frame's hash is
    what bench.py would print."""
    from cudapathtracer_amd import scenes
    gen, kw = {"C2_cornell": ("cornell", dict(spp=4, max_depth=8)), "C3_blob82k": ("blob_in_box", dict(spp=2, max_depth=8)),
               "C4_atrium263k": ("atrium", dict(spp=1, max_depth=16)),
               # the general bounce (bench.py's secondary workloads) and Li_naive_unidirectional (deviceCode.cu:158-205)
               "mixed_cornell": ("cornell", dict(spp=4, max_depth=8, tall_material=19, short_material=5, nested=True, extra_boxes=1, extra_materials=[4])),
               "glass_blob82k": ("blob_in_box", dict(spp=2, max_depth=8, material=5)),
               "C2_cornell_naive": ("cornell", dict(spp=4, max_depth=8)), "C3_blob82k_naive": ("blob_in_box", dict(spp=2, max_depth=8))}[config]
    integ = 2 if config.endswith("_naive") else 0
    generic = config in ("mixed_cornell", "glass_blob82k")
    w, h = 1920, 1080
    s = getattr(scenes, gen)(os.path.join(scene_dir, "full_" + config), width=w, height=h, name="full_" + gen, **kw)
    hs = api.HostScene(s["config"])
    sc = api.Scene(hs)
    got, _ = sc.render(hs.camera(), w, h, kw["spp"], kw["max_depth"], integrator=integ)
    fl = sc.flags()
    assert fl["simple"] == (not generic) and fl["lean"] == generic and fl["persistent"], fl
    assert (fl["flat"] and fl["flat_pair"] == (integ == 0)) if gen == "cornell" else (fl["hbm_kernel"] and fl["refill"]), fl
    want, _, secs = oracle.OracleScene(s["config"]).render(threads=16, integrator=integ)
    assert want.shape == (h, w, 4) and float(np.nan_to_num(want[..., :3], nan=0.0, posinf=0.0, neginf=0.0).sum()) > 0.0     # (NaN pixels the reference itself produces are part of the frame)
    assert_bits_equal(got, want, "%s, full 1080p frame at %d spp (oracle: %.1f s)" % (config, kw["spp"], secs))


def test_fuzz_scene_depth16(api, oracle, gpu_ready, scene_dir):
    """A random scene over the whole material table at the C4 depth: Russian roulette only after 16 bounces."""
    from cudapathtracer_amd import scenes
    cfg = scenes.fuzz(os.path.join(scene_dir, "fuzzd16"), 101, width=24, height=16, spp=4, max_depth=16, name="fuzzd16")["config"]
    for opts in ({}, {"onchip": 0, "waves_hbm": 2}):
        gs, hs, osc = _scene_pair(api, oracle, cfg, options=opts)
        i = hs.info
        assert i["max_depth"] == 16
        for integ in (0, 2):
            ocol, ocnt, _ = osc.render(integrator=integ, counters=True, threads=8)
            col, cnt = gs.render(hs.camera(), i["width"], i["height"], i["spp"], i["max_depth"], integrator=integ, counters=True)
            assert np.array_equal(cnt, ocnt), integ
            assert_bits_equal(col, ocol, "fuzz depth 16, integrator %d" % integ)
            timed, _ = gs.render(hs.camera(), i["width"], i["height"], i["spp"], i["max_depth"], integrator=integ)
            assert_bits_equal(timed, ocol, "fuzz depth 16, integrator %d, timed kernel" % integ)
        assert int(ocnt[..., 7].max()) > 16


def test_simple_bounce_specialisation(api, oracle, gpu_ready, scene_dir):
    """pt_path.h SIMPLE: scenes whose triangles are all untextured MAT_DIFFUSE (the reference's Cornell configuration, and
    BASELINE's C2-C4 scenes) get a bounce with one arm per dispatcher and no medium stack. Same image as the generic
    bounce and as the oracle; scenes with any other material, a texture, or absorbing 'air' must not qualify."""
    from cudapathtracer_amd import scenes
    for case in ("cornell32_mis", "cornell32_naive", "cornell64_mis"):
        g = np.load(os.path.join(GOLDEN, case + ".npz"))
        hs = api.HostScene(golden_case_scene(g))
        w, h = int(g["w"]), int(g["h"])
        for simple in (1, 0):
            sc = api.Scene(hs, options={"simple": simple})
            col, _ = sc.render(hs.camera(), w, h, int(g["spp"]), int(g["max_depth"]), integrator=int(g["integrator"]))
            assert sc.flags()["flat"] and sc.flags()["simple"] == bool(simple), sc.flags()
            assert_bits_equal(col, g["colors"], "%s simple=%d" % (case, simple))
    for case in ("mixed32_mis", "metal32_mis", "textured32_mis"):                      # mirror / glass / conductors / textures: generic bounce
        g = np.load(os.path.join(GOLDEN, case + ".npz"))
        hs = api.HostScene(golden_case_scene(g))
        sc = api.Scene(hs)
        col, _ = sc.render(hs.camera(), int(g["w"]), int(g["h"]), int(g["spp"]), int(g["max_depth"]), integrator=int(g["integrator"]))
        assert not sc.flags()["simple"], (case, sc.flags())
        assert_bits_equal(col, g["colors"], case)
    # a diffuse-only scene at depth 16 with long paths, both integrators, against a live oracle run
    cfg = scenes.cornell(os.path.join(scene_dir, "simple16"), 40, 24, 6, 16, ceiling_light=True, name="simple16")["config"]
    gs, hs, osc = _scene_pair(api, oracle, cfg)
    for integ in (0, 2):
        ocol, _, _ = osc.render(integrator=integ, threads=8)
        col, _ = gs.render(hs.camera(), 40, 24, 6, 16, integrator=integ)
        assert gs.flags()["simple"]
        assert_bits_equal(col, ocol, "diffuse-only scene, depth 16, integrator %d" % integ)


@pytest.mark.parametrize("integrator", [0, 2])
def test_flat_128_and_larger_workgroups(api, oracle, gpu_ready, scene_dir, integrator):
    """LDS-resident scenes beyond Cornell: 65-128 triangles run the FLAT traversal with 128-bit masks (incl. a doubled box:
    every hit on it is a two-way tie), and scenes whose records need more LDS than a 4-wave workgroup's share run in
    workgroups of 8 or 16 waves that hold one copy of the scene — FLAT up to 128 triangles, the stack walk beyond.
    Timed and counting kernels against the oracle, bit for bit."""
    from cudapathtracer_amd import scenes
    cases = [("x3", dict(extra_boxes=3), True),                                                    # 72 triangles, diffuse only (SIMPLE)
             ("x5mix", dict(extra_boxes=5, tall_material=19, short_material=5, nested=True), True),      # 108: mirror, glass, nested water
             ("x5twin", dict(extra_boxes=5, doubled=19), True),                                    # 108 with a doubled box: ties
             ("x7", dict(extra_boxes=7, ceiling_light=True), True)]                                # 120 triangles
    for name, kw, want_flat in cases:
        cfg = scenes.cornell(os.path.join(scene_dir, "f128_" + name), 40, 24, 5, 7, name=name, **kw)["config"]
        gs, hs, osc = _scene_pair(api, oracle, cfg)
        i = hs.info
        assert 64 < i["n_tris"] <= 128, i["n_tris"]
        ocol, ocnt, _ = osc.render(integrator=integrator, counters=True, threads=8)
        col, _ = gs.render(hs.camera(), i["width"], i["height"], i["spp"], i["max_depth"], integrator=integrator)
        fl = gs.flags()
        assert fl["onchip"] and fl["flat"] == want_flat, (name, fl)
        assert_bits_equal(col, ocol, "%s: FLAT with 128-bit masks" % name)
        col2, cnt = gs.render(hs.camera(), i["width"], i["height"], i["spp"], i["max_depth"], integrator=integrator, counters=True)
        assert np.array_equal(cnt, ocnt), name
        assert_bits_equal(col2, ocol, "%s: counting kernel" % name)
        gs0 = api.Scene(hs, options={"flat": 0})                                                    # the stack walk in the same workgroup shape
        col3, _ = gs0.render(hs.camera(), i["width"], i["height"], i["spp"], i["max_depth"], integrator=integrator)
        assert gs0.flags()["onchip"] and not gs0.flags()["flat"]
        assert_bits_equal(col3, ocol, "%s: stack walk" % name)
    # 356 triangles: LDS-resident only in 16-wave workgroups, stack walk (too many nodes for FLAT)
    cfg = scenes.blob_in_box(os.path.join(scene_dir, "blob2"), 40, 24, 4, 6, subdiv=2, name="blob2")["config"]
    gs, hs, osc = _scene_pair(api, oracle, cfg)
    i = hs.info
    assert 300 < i["n_tris"] < 400
    ocol, ocnt, _ = osc.render(integrator=integrator, counters=True, threads=8)
    col, _ = gs.render(hs.camera(), i["width"], i["height"], i["spp"], i["max_depth"], integrator=integrator)
    assert gs.flags()["onchip"] and not gs.flags()["flat"], gs.flags()
    assert_bits_equal(col, ocol, "356 triangles in 16-wave workgroups")
    col2, cnt = gs.render(hs.camera(), i["width"], i["height"], i["spp"], i["max_depth"], integrator=integrator, counters=True)
    assert np.array_equal(cnt, ocnt)
    assert_bits_equal(col2, ocol, "356 triangles, counting kernel")


def test_flat_pair_kernel(api, oracle, gpu_ready, scene_dir):
    """pt_trace.h trace_pair_flat ("flat2" = 1): SIMPLE scenes of at most 64 nodes / triangles trace the shadow ray of a bounce
    and the next extension ray in ONE FLAT pass (shared lockstep node walk, tests of both rays dealt out together, DEFER
    logic step). Golden Cornell frames, a doubled box (every hit on it a tie), the C1 configuration — bit for bit."""
    from cudapathtracer_amd import scenes
    for case in ("cornell32_mis", "cornell64_mis"):
        g = np.load(os.path.join(GOLDEN, case + ".npz"))
        hs = api.HostScene(golden_case_scene(g))
        sc = api.Scene(hs, options={"flat2": 1, "slice_iters": 8, "sched_mask": 3})
        for _ in range(2):
            col, _ = sc.render(hs.camera(), int(g["w"]), int(g["h"]), int(g["spp"]), int(g["max_depth"]))
            assert sc.flags()["flat"] and sc.flags()["simple"] and sc.flags()["flat_pair"], sc.flags()
            assert_bits_equal(col, g["colors"], case + " flat2")
        naive = np.load(os.path.join(GOLDEN, "cornell32_naive.npz"))                 # the naive integrator has no shadow rays: plain FLAT
    cfgs = [scenes.cornell(os.path.join(scene_dir, "p2twin"), 40, 24, 6, 16, doubled=17, ceiling_light=True, name="p2twin")["config"],
            scenes.cornell(os.path.join(scene_dir, "p2c1"), 256, 256, 16, 4, name="p2c1")["config"],
            scenes.cornell(os.path.join(scene_dir, "p2x2"), 45, 27, 5, 9, extra_boxes=2, name="p2x2")["config"]]
    for cfg in cfgs:
        gs, hs, osc = _scene_pair(api, oracle, cfg, options={"flat2": 1})
        i = hs.info
        ocol, _, _ = osc.render(threads=8)
        col, _ = gs.render(hs.camera(), i["width"], i["height"], i["spp"], i["max_depth"])
        assert gs.flags()["flat"] and gs.flags()["simple"] and gs.flags()["flat_pair"], gs.flags()
        assert_bits_equal(col, ocol, cfg)
        g0 = api.Scene(hs, options={"flat2": 0})                                     # one ray per pass: the round-1 shape of FLAT
        col0, _ = g0.render(hs.camera(), i["width"], i["height"], i["spp"], i["max_depth"])
        assert g0.flags()["flat"] and not g0.flags()["flat_pair"]
        assert_bits_equal(col0, ocol, cfg + " flat2=0")


def test_flat_leaf_boxes_and_degenerate_directions(api, oracle, gpu_ready, scene_dir):
    """The FLAT kernels decide which leaves a ray visits from the leaves' own boxes (nested boxes + a monotone slab test:
    pt_trace.h) — unless a ray of the wave has a zero direction component, where 0 * inf = NaN breaks the argument and the
    wave walks the nodes. Both forms against the oracle, and a camera that looks exactly along -z from the room's centre
    line with a zero-aperture lens... whose primary rays through the image centre column / row have an exact 0 component."""
    from cudapathtracer_amd import scenes
    cfg = scenes.cornell(os.path.join(scene_dir, "lb"), 41, 25, 6, 9, name="lb")["config"]           # odd size: a centre column and row exist
    hs = api.HostScene(cfg)
    osc = oracle.OracleScene(cfg)
    cam = api.Camera.NotPinhole((0.0, 0.0, 1.0), 41, 25, (0.0, 0.0, 0.0), 60.0, 0.0, 1.0)              # aperture 0: two uniforms per camera ray, no lens offset
    cb = np.frombuffer(cam.tobytes(), np.uint8)
    for integ in (0, 2):
        ocol, _, _ = osc.render(camera=cb, width=41, height=25, spp=6, max_depth=9, integrator=integ, threads=8)
        for opts in ({}, {"leaf_boxes": 0}, {"flat2": 0}, {"flat2": 0, "leaf_boxes": 0}, {"simple": 0}):
            sc = api.Scene(hs, options=opts)
            col, _ = sc.render(cam, 41, 25, 6, 9, integrator=integ)
            assert sc.flags()["flat"], sc.flags()
            assert_bits_equal(col, ocol, "leaf boxes %s integrator %d" % (opts, integ))
    # rays with exact zero components through the probes: axis-parallel directions from inside the room
    rays = np.array([[0.0, 0.0, 0.5, 0.0, 0.0, -1.0], [0.1, -0.2, 0.0, 1.0, 0.0, 0.0], [0.1, -0.2, -0.5, 0.0, 1.0, 0.0],
                     [0.0, 0.0, 0.0, 0.0, -1.0, 0.0], [0.3, 0.1, -0.2, -1.0, 0.0, 0.0], [0.2, 0.0, 0.3, 0.70710677, 0.0, -0.70710677]], np.float32)
    gi, gf, _ = api.Scene(hs).trace_closest(rays)
    oi, of, _ = osc.trace_closest(rays)
    assert np.array_equal(gi, oi)
    assert_bits_equal(gf, of, "axis-parallel rays")


def test_caller_tree_with_loose_boxes_takes_the_reference_walk(api, oracle, gpu_ready):
    """pt_scene_create also takes a caller's own BVH arrays (array-level boundary, main.cu:469-557). visited(leaf) ==
    slab(leaf's own box) holds only for finite, nested boxes; a tree whose inner box does NOT contain its children (a refit
    gone wrong, here: an internal node's box shrunk to a quarter) must render as the reference's walk over the boxes as
    given — the FLAT kernels fall back to the lockstep node walk (flag leaf_table off), bit for bit the oracle's frame."""
    hs = api.HostScene(golden_scene("cornell32"))
    names = ("points", "normals", "uvs", "mesh", "lights", "bvh", "indices", "materials")
    arr = {k: hs.array(k) for k in names}
    cam = hs.camera()
    cb = np.frombuffer(cam.tobytes(), np.uint8)
    nested = api.Scene.from_arrays(arr)
    nested.render(cam, 32, 32, 1, 4)
    assert nested.flags()["flat"] and nested.flags()["leaf_table"], nested.flags()
    bvh = arr["bvh"].view(np.float32).reshape(-1, 12).copy()
    bi = bvh.view(np.int32)
    inner = [i for i in range(1, len(bvh)) if bi[i, 11] <= 0]                # internal nodes below the root
    assert len(inner) >= 3
    differs = 0
    for victim, how in ((inner[0], "shrunk"), (inner[len(inner) // 2], "shrunk"), (inner[-1], "infinite")):
        b = bvh.copy()
        if how == "shrunk":
            c, e = 0.5 * (b[victim, 0:3] + b[victim, 4:7]), 0.5 * (b[victim, 4:7] - b[victim, 0:3])
            b[victim, 0:3], b[victim, 4:7] = c - 0.25 * e, c + 0.25 * e
        else:
            b[victim, 4] = np.inf                                             # still contains its children, but is not finite
        loose = dict(arr, bvh=b.view(np.uint8).reshape(-1))
        osc = oracle.OracleScene(arrays=loose)
        for integ in (0, 2):
            ocol, _, _ = osc.render(camera=cb, width=32, height=32, spp=4, max_depth=4, integrator=integ, threads=8)
            for opts in ({}, {"flat2": 0}, {"flat": 0}):
                sc = api.Scene.from_arrays(loose, options=opts)
                col, _ = sc.render(cam, 32, 32, 4, 4, integrator=integ)
                fl = sc.flags()
                assert fl["flat"] == (opts.get("flat", 1) == 1) and not fl["leaf_table"], (victim, how, opts, fl)
                assert_bits_equal(col, ocol, "loose tree: node %d %s, integrator %d, %s" % (victim, how, integ, opts))
            if how == "shrunk":
                good, _ = nested.render(cam, 32, 32, 4, 4, integrator=integ)
                differs += int(not np.array_equal(good.view(np.uint32), ocol.view(np.uint32)))
    assert differs >= 1          # the shrunk boxes really hide geometry: the leaf table would have rendered a different frame


def test_tie_scenes_and_axis_parallel_rays_in_hbm_kernels(api, oracle, gpu_ready, scene_dir):
    """The production kernel for scenes in HBM on the cases built for the opt-in trees (EXPERIMENTAL=1 builds add those:
    pt_trace_experimental.h trace_resume_w4: SIMPLE scenes in HBM traverse the reference tree collapsed to 4-wide nodes, children in no
    particular order; equal-t ties and rays with a zero direction component fall back to the reference traversal. A scene
    whose every second quad is DOUBLED (each hit on them a two-way tie between different leaves), an axis-aligned camera
    (primary rays with exact zero components), and a blob — against the oracle. Opt-in ("wide" = 1): measured slower.
    The same cases for trace_resume_q ("compact" = 1: quantised inner nodes, exact leaf boxes, ties decided by a walk down
    the reference's nodes; rays with a zero direction component read the reference's nodes in the same loop)."""
    from cudapathtracer_amd import scenes
    cfgs = [scenes.blob_in_box(os.path.join(scene_dir, "wblob4"), 64, 40, 3, 8, subdiv=4, name="wblob4")["config"],
            scenes.cornell(os.path.join(scene_dir, "wtwin"), 48, 32, 4, 10, doubled=17, extra_boxes=7, ceiling_light=True, name="wtwin")["config"]]
    for cfg in cfgs:
        hs = api.HostScene(cfg)
        osc = oracle.OracleScene(cfg)
        i = hs.info
        for cam in (hs.camera(), api.Camera.NotPinhole((0.0, 0.0, 1.0), i["width"], i["height"], (0.0, 0.0, 0.0), 60.0, 0.0, 1.0)):
            ocol, _, _ = osc.render(camera=np.frombuffer(cam.tobytes(), np.uint8), threads=8)
            variants = ({"onchip": 0, "waves_hbm": 2}, {"onchip": 0, "waves_hbm": 2, "slice_iters": 8, "sched_mask": 3, "refill_keep": 12})
            if api.has_experimental():
                variants += ({"onchip": 0, "waves_hbm": 2, "wide": 1}, {"onchip": 0, "waves_hbm": 2, "wide": 1, "slice_iters": 8, "sched_mask": 3, "refill_keep": 12},
                             {"onchip": 0, "waves_hbm": 2, "compact": 1}, {"onchip": 0, "waves_hbm": 2, "compact": 1, "slice_iters": 8, "sched_mask": 3, "refill_keep": 12})
            for opts in variants:
                sc = api.Scene(hs, options=opts)
                col, _ = sc.render(cam, i["width"], i["height"], i["spp"], i["max_depth"])
                fl = sc.flags()
                assert fl["hbm_kernel"] and fl["simple"] and not fl["onchip"], fl
                assert_bits_equal(col, ocol, "%s %s" % (os.path.basename(cfg), opts))


def test_flat_pair_kernel_generic_bounce(api, oracle, gpu_ready, scene_dir):
    """The pair form of FLAT needs an order-free shadow ray (no MAT_LEAF triangle in the scene) and an exact DEFER step (every
    material has a dispatch arm), not the SIMPLE bounce: mirror / glass / nested water / conductors take it with the generic
    bounce; a scene with leaf materials must not."""
    from cudapathtracer_amd import scenes
    for case, pair in (("mixed32_mis", True), ("metal32_mis", True), ("textured32_mis", False)):
        g = np.load(os.path.join(GOLDEN, case + ".npz"))
        hs = api.HostScene(golden_case_scene(g))
        for opts in ({}, {"flat2": 0}, {"slice_iters": 8, "sched_mask": 3}, {"lean": 0}):
            sc = api.Scene(hs, options=opts)
            col, _ = sc.render(hs.camera(), int(g["w"]), int(g["h"]), int(g["spp"]), int(g["max_depth"]))
            fl = sc.flags()
            assert fl["flat"] and not fl["simple"] and fl["flat_pair"] == (pair and opts.get("flat2", 1) == 1), (case, opts, fl)
            assert fl["lean"] == (fl["flat_pair"] and opts.get("lean", 1) == 1), (case, opts, fl)        # mirror / glass / metals: no leaf arm, no texture
            assert_bits_equal(col, g["colors"], "%s %s" % (case, opts))
    cfg = scenes.cornell(os.path.join(scene_dir, "pgmix"), 48, 32, 6, 12, tall_material=19, short_material=18, nested=True, doubled=4, name="pgmix")["config"]
    gs, hs, osc = _scene_pair(api, oracle, cfg)
    i = hs.info
    ocol, _, _ = osc.render(threads=8)
    col, _ = gs.render(hs.camera(), i["width"], i["height"], i["spp"], i["max_depth"])
    assert gs.flags()["flat_pair"] and not gs.flags()["simple"], gs.flags()
    assert_bits_equal(col, ocol, "mirror + diamond + nested water + doubled gold box, pair form")
