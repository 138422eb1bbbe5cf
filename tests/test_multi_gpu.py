"""The multi-GPU entry points of the C ABI (include/pt_api.h "multi-GPU"): pt_rank_tiles, pt_multi_*,
pt_render_multi. A one-GPU box can check everything except the xGMI hop itself: the tile math, one host thread per rank,
replicated scenes, the gather into rank slots, the de-interleave, `+=` semantics, error paths — with several ranks on
device 0 (peer-copy transport) and with the RCCL transport on a one-rank communicator (the tile buffer is sent through
ncclSend / ncclRecv to itself). Frames must equal pt_render's bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_case_scene, golden_scene
from util import assert_bits_equal


def test_rank_tiles_matches_the_partition_rule(api):
    for (w, h) in ((1920, 1080), (33, 9), (8, 8), (1, 1), (70, 41), (64, 64)):
        total = ((w + 7) // 8) * ((h + 7) // 8)
        for world in (1, 2, 3, 4, 8, 16):
            seen = []
            for r in range(world):
                tr = api.rank_tiles(w, h, r, world)
                assert (tr.first, tr.stride) == (r, world)
                assert tr.count == len(range(r, total, world))
                seen += [tr.first + k * tr.stride for k in range(tr.count)]
            assert sorted(seen) == list(range(total))


def test_multi_create_fails_cleanly_without_a_device_or_with_bad_arguments(api):
    import torch
    L = api.lib()
    assert L.pt_multi_create(None, 1, None) is None and b"null desc" in L.pt_last_error()
    hs_desc = api.SceneDesc()
    if not torch.cuda.is_available():
        assert L.pt_multi_create(C.byref(hs_desc), 1, None) is None
        assert b"no usable HIP device" in L.pt_last_error() or b"failed" in L.pt_last_error()
    assert L.pt_multi_render(None, None, 8, 8, 1, 1, 0, 1, 0, None, None) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["cornell64_mis", "mixed32_naive"])
def test_multi_render_equals_single_device_render(api, gpu_ready, case):
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    hs = api.HostScene(golden_case_scene(g))
    w, h, spp, md, integ = int(g["w"]), int(g["h"]), int(g["spp"]), int(g["max_depth"]), int(g["integrator"])
    cam = hs.camera()
    for n, opts, transport in ((1, {}, "none"), (1, {"self_gather": 1, "gather": 1}, "rccl"), (1, {"self_gather": 1, "gather": 2}, "peer_copy"),
                               (2, {}, "peer_copy"), (3, {"gather": 2}, "peer_copy"), (5, {"slice_iters": 8, "sched_mask": 3}, "peer_copy")):
        ms = api.MultiScene(hs, n, device_ids=[0] * n, options=opts)
        col = ms.render(cam, w, h, spp, md, integrator=integ)
        assert ms.stats["n_devices"] == n and ms.stats["gather"] == transport, ms.stats
        assert all(t > 0 for t in ms.stats["kernel_ms"]) and ms.stats["total_ms"] >= ms.stats["render_ms"] > 0
        assert_bits_equal(col, g["colors"], "%s on %d ranks %s" % (case, n, opts))
        again = ms.render(cam, w, h, spp, md, integrator=integ)          # buffers and communicators are reused
        assert_bits_equal(again, g["colors"], "second frame")
        ms.close()


@pytest.mark.gpu
def test_multi_render_accumulates_and_handles_ragged_frames(api, oracle, gpu_ready):
    hs = api.HostScene(golden_scene("cornell32"))
    sc = api.Scene(hs)
    cam = api.Camera.Pinhole((0, 0, 1), 33, 9)                           # 5 x 2 tiles: with 3 or 4 ranks the counts differ (padding)
    pre = np.random.default_rng(3).random((9, 33, 4)).astype(np.float32)
    want, _ = sc.render(cam, 33, 9, 3, 4, out=pre.copy())
    for n in (3, 4, 16):
        ms = api.MultiScene(hs, n, device_ids=[0] * n)
        got = ms.render(cam, 33, 9, 3, 4, out=pre.copy())
        assert_bits_equal(got, want, "ragged frame, %d ranks, += semantics" % n)
        ms.close()
    one = api.render_multi(hs, 2, cam, 33, 9, 3, 4, device_ids=[0, 0])   # pt_render_multi, the one-shot form
    zero, _ = sc.render(cam, 33, 9, 3, 4)
    assert_bits_equal(one, zero, "pt_render_multi")
    wf = api.MultiScene(hs, 2, device_ids=[0, 0]).set_variant("wavefront")
    assert_bits_equal(wf.render(cam, 33, 9, 3, 4), zero, "wavefront variant, 2 ranks")


@pytest.mark.gpu
def test_multi_errors(api, gpu_ready):
    hs = api.HostScene(golden_scene("cornell32"))
    with pytest.raises(api.PtError, match="device id"):
        api.MultiScene(hs, 2, device_ids=[0, 97])
    with pytest.raises(api.PtError, match="n_devices"):
        api.MultiScene(hs, 0)
    ms = api.MultiScene(hs, 2, device_ids=[0, 0], options={"gather": 1})
    with pytest.raises(api.PtError, match="two ranks on one device"):     # RCCL proper needs distinct devices
        ms.render(hs.camera(), 32, 32, 1, 4)
    with pytest.raises(api.PtError):
        ms.set_option("gather", 7)
    with pytest.raises(api.PtError, match="out of scope"):
        ms.set_option("gather", 2).render(hs.camera(), 32, 32, 1, 4, integrator=3)


def _frame_82k(api, scene_dir):
    from cudapathtracer_amd import scenes
    cfg = scenes.blob_in_box(os.path.join(scene_dir, "mb82"), 1920, 1080, 2, 6, name="mb82")["config"]
    hs = api.HostScene(cfg)
    full, _ = api.Scene(hs).render(hs.camera(), 1920, 1080, 2, 6)
    return hs, full


def _explain(api, hs, got, full, stats):
    """Say WHICH ranks' tiles differ, and how, before failing."""
    bad = np.argwhere((got.view(np.uint32) != full.view(np.uint32)).any(axis=-1))
    tiles = (bad[:, 0] // 8) * 240 + bad[:, 1] // 8
    again, _ = api.Scene(hs).render(hs.camera(), 1920, 1080, 2, 6)
    g, f = got[bad[:, 0], bad[:, 1], :3], full[bad[:, 0], bad[:, 1], :3]
    return ("differing pixels %d in %d tiles of ranks %s; got == 0 in %d, got == 2 x expected in %d, got NaN in %d; first (y, x) %s got %s want %s; "
            "single-device frame repeatable %s; stats %s" %
            (len(bad), len(set(tiles.tolist())), sorted(set((tiles % 8).tolist())), int((g == 0).all(axis=1).sum()), int((g == 2 * f).all(axis=1).sum()),
             int(np.isnan(g).any(axis=1).sum()), bad[0].tolist(), g[0].tolist(), f[0].tolist(), np.array_equal(again.view(np.uint32), full.view(np.uint32)), stats))


@pytest.mark.gpu
@pytest.mark.parametrize("same_device", [1, 0])
def test_multi_render_full_frame_on_a_scene_in_hbm(api, gpu_ready, scene_dir, same_device):
    """1080p, 82 k triangles, 8 ranks' worth of tile shares rendered through pt_multi (all on this box's one device): the
    frame equals the single-launch frame; each share is the 4050-tile launch an 8-GPU run gives every GPU. Eight host
    threads run the library concurrently in both modes; with same_device = 0 the eight persistent megakernels also
    co-reside on the device (eight streams), with 1 they run in the order they reach the device's one stream."""
    hs, full = _frame_82k(api, scene_dir)
    ms = api.MultiScene(hs, 8, device_ids=[0] * 8, options={"same_device": same_device})
    for frame in range(2):                                                   # second frame: every buffer, stream and event reused
        got = ms.render(hs.camera(), 1920, 1080, 2, 6)
        if not np.array_equal(got.view(np.uint32), full.view(np.uint32)):
            pytest.fail("frame %d, same_device %d: %s" % (frame, same_device, _explain(api, hs, got, full, ms.stats)))
    assert ms.stats["gather"] == "peer_copy" and len(ms.stats["kernel_ms"]) == 8
    ms.close()


@pytest.mark.gpu
@pytest.mark.parametrize("bounce", ["simple", "generic"])
def test_concurrent_megakernels_of_separate_scenes_from_one_host_thread(api, gpu_ready, scene_dir, bounce):
    """Kernel co-residency WITHOUT host concurrency: one host thread launches the eight 4050-tile shares of the frame on eight
    replicas of the scene and eight streams back to back (pt_render_tiles_device is asynchronous), so eight persistent
    megakernels — each with its own tile queue, RNG states, spill area and private segment — are in flight on one device at
    once; then it waits. Every share must equal the single-launch frame's tiles. `generic` forces the bounce with the
    material dispatch (216 bytes of private segment per lane instead of 32): the co-resident kernels then also share the
    device's scratch backing store. No tile changes hands here (2 spp: far below a 512-iteration slice) — asserted."""
    import torch
    hs, full = _frame_82k(api, scene_dir)
    w, h = 1920, 1080
    cam = hs.camera()
    opts = {"simple": 0} if bounce == "generic" else {}
    scenes_ = [api.Scene(hs, options=opts) for _ in range(8)]
    streams = [torch.cuda.Stream() for _ in range(8)]
    bufs = [torch.zeros(4050, 64, 4, device="cuda") for _ in range(8)]
    torch.cuda.synchronize()
    for rnd in range(2):                                                     # second round: queues, RNG and spill buffers reused
        for b in bufs:
            b.zero_()
        torch.cuda.synchronize()
        for r in range(8):
            scenes_[r].render_tiles_device(cam, w, h, 2, 6, bufs[r].data_ptr(), tiles=api.rank_tiles(w, h, r, 8), stream=streams[r].cuda_stream)
        torch.cuda.synchronize()
        frame = torch.zeros(h, w, 4, device="cuda")
        for r in range(8):
            assert scenes_[r].last_kernel_ms() > 0.0
            fl = scenes_[r].flags()
            assert not fl["hbm_kernel"] and fl["refill"] and fl["simple"] == (bounce == "simple"), fl      # the 4-wave kernels of an 8-GPU share
            assert scenes_[r].tile_handovers() == 0
            api.untile_device(w, h, bufs[r].data_ptr(), frame.data_ptr(), api.rank_tiles(w, h, r, 8))
        torch.cuda.synchronize()
        got = frame.cpu().numpy()
        if not np.array_equal(got.view(np.uint32), full.view(np.uint32)):
            pytest.fail("round %d, %s bounce: %s" % (rnd, bounce, _explain(api, hs, got, full, None)))
    for s in scenes_:
        s.close()
