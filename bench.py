#!/usr/bin/env python3
"""bench.py — throughput of the unidirectional path-tracing hot path on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full render of the workload BASELINE.json's metric is quoted on (configs[1]):
Cornell box, 1920x1080, 1024 spp, depth 8, MIS integrator, seed 103033, synthetic scene from
cudapathtracer_amd.scenes. With N > 1 the frame is sharded by interleaved 8x8 tiles (the scene is
replicated, no data-path collective) and one RCCL gather brings the tile buffers to rank 0, which
de-interleaves them; that gather + de-interleave is inside the timed region.

Rank 0 prints ONE JSON line. `value` is whole-job Mray/s (closest-hit + shadow traversals, all
ranks) with the scene resident in HBM when the clock starts. Ray / box / triangle counts come from
a counted, untimed pass of the identical (deterministic) workload, and the timed frame must equal
the counted frame bit for bit (`frame_sha`; the run fails otherwise). `roofline` prices the
megakernel against the bound that binds it (tools/roofline.py): VALU issue slots for scenes that
live in LDS (the headline), the L1's cache-line rate for scenes in HBM — both fractions are <= 1 by
construction; SURVEY.md §8(d)'s algorithmic bytes are kept as the labelled field `algorithmic`.
`secondary` (N = 1; one timed step each, labelled) holds the same measurement for BASELINE C3 at its full
sample count, the C4 scene at 32 of its 4096 spp, the GENERAL bounce — a 1080p Cornell box with mirror,
glass, nested water and a GGX box (reflectors.cuh:588-629 dispatch, medium stack deviceCode.cu:347-432)
and the 82 k-triangle scene with a glass blob — and C5, the wavefront variant on the C4 scene.
`projected_scaling` (N = 1) is kernel time of the full frame / kernel time of rank 0's 1/N share on this
one GPU: what N GPUs could reach before the gather — a projection, not a scaling curve.
Counters from a PMC pass are only used if `code_sha256` of their entry in profiles/roofline_inputs.json
equals the hash of the kernel in the library being timed; otherwise `frac` is null and `stale_profile`
true. `cpu_baseline` is the CPU oracle on a bounded sample of the headline workload (rank 0, N = 1 only)
— a reported reference point, not the target.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(1, os.path.join(ROOT, "tools"))

import hashlib  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import roofline as RF  # noqa: E402  (tools/roofline.py: every roofline number of the line below is computed there)

WORKLOADS = {
    # name: (generator, kwargs)  — C2 is the headline; C3/C4 are selectable for profiling
    "cornell_1920x1080_1024spp_depth8_mis": ("cornell", dict(width=1920, height=1080, spp=1024, max_depth=8)),
    "blob82k_1920x1080_1024spp_depth8_mis": ("blob_in_box", dict(width=1920, height=1080, spp=1024, max_depth=8)),
    "atrium262k_1920x1080_4096spp_depth16_mis": ("atrium", dict(width=1920, height=1080, spp=4096, max_depth=16)),
    # the general bounce: mirror tall box, glass short box with a water box nested inside, a gold (GGX) box
    "cornell_mixed_1920x1080_1024spp_depth8_mis": ("cornell", dict(width=1920, height=1080, spp=1024, max_depth=8, tall_material=19, short_material=5,
                                                                  nested=True, extra_boxes=1, extra_materials=[4], name="cornell_mixed")),
    "blob82k_glass_1920x1080_1024spp_depth8_mis": ("blob_in_box", dict(width=1920, height=1080, spp=1024, max_depth=8, material=5, name="blob_glass")),
}
_SCENES = {}                                   # generated scenes of this process, by workload


def host_threads():
    """Threads the CPU baseline may use: the cgroup CPU quota if one is set (a 1-GPU box is given a
    16-CPU share of a larger host), else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return min(n, int(os.environ.get("PT_BENCH_CPU_THREADS", "16")))


def cpu_baseline(cfg, info, seconds_budget=25.0):
    """The oracle (kind "port": this repo's CPU restatement; the CUDA reference cannot be built here)
    on a bounded sample of the workload: the full frame at a reduced sample count."""
    from oracle import oracle_py as O          # checker / reported baseline only
    cores = host_threads()
    sc = O.OracleScene(cfg)
    w, h = info["width"], info["height"]
    _, _, t1 = sc.render(spp=1, threads=cores)                       # calibrate: whole frame, 1 spp
    spp = int(max(1, min(64, seconds_budget / max(t1, 1e-3))))
    col, cnt, secs = sc.render(spp=spp, counters=True, threads=cores)
    rays = int(cnt[..., 0].sum() + cnt[..., 1].sum())
    return {"value": rays / secs / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
            "msample_per_s": w * h * spp / secs / 1e6,
            "sample": "full %dx%d frame at %d of %d spp (%.1f s), same scene/seed/depth, %d host threads" % (w, h, spp, info["spp"], secs, cores)}


def _tf(b):
    return "true" if b else "false"


def kernel_name(flags, variant):
    if variant != "megakernel":
        return "pt::wf_logic_kernel + pt::wf_trace_kernel (all launches of one frame)"
    lean = _tf(flags.get("lean", False))
    if flags["hbm_kernel"]:
        if flags.get("simple"):
            return "pt::megakernel_hbm_simple<0>"
        return "pt::megakernel_hbm<0, false, %s, %s, false, %s>" % (_tf(flags["culling"]), _tf(flags["refill"]), lean)
    if flags.get("simple") and not flags["onchip"]:
        return "pt::megakernel<0, false, false, false, true, false, true, 1, false>"       # small shares of a diffuse-only scene in HBM
    if flags.get("flat_pair"):
        return "pt::megakernel_flat2<0, %s, %s>" % (_tf(flags.get("simple", False)), lean)
    if flags.get("simple"):
        return "pt::megakernel<0, false, false, true, false, true, true, 1, false>"
    return "pt::megakernel<0, false, false, %s, %s, %s, false, 1, %s>" % (_tf(flags["onchip"]), _tf(flags["refill"]), _tf(flags["flat"]), lean)


def measure(ctx, workload, spp_override, steps, warmup, opts, variant, culling):
    """One workload: scene into HBM, counted pass, `steps` timed passes, frame check. Returns the result dict on rank 0."""
    from cudapathtracer_amd import api, scenes
    from cudapathtracer_amd import distributed as D
    rank, world, red_dev, share = ctx["rank"], ctx["world"], ctx["red_dev"], ctx["share"]
    gen, kw = WORKLOADS[workload]
    if workload not in _SCENES:
        tmp = tempfile.mkdtemp(prefix="ptbench_r%d_" % rank)
        sinfo = getattr(scenes, gen)(tmp, **kw)
        _SCENES[workload] = (sinfo, api.HostScene(sinfo["config"]))
    sinfo, host = _SCENES[workload]
    info = host.info
    w, h, md = info["width"], info["height"], info["max_depth"]
    spp = spp_override or info["spp"]
    cam = host.camera()
    scene = api.Scene(host, options=opts).set_variant(variant)           # scene resident in HBM from here on
    if culling:
        scene.set_culling(True)

    tr = api.rank_tiles(w, h, rank, world)
    pad = D.padded_tile_count(w, h, world)
    tiles = torch.zeros(pad, 64, 4, device="cuda")
    frame = torch.zeros(h, w, 4, device="cuda") if rank == 0 else None
    stream = torch.cuda.current_stream().cuda_stream

    def step(count_work=False):
        tiles.zero_()                                           # out_colors starts at 0 (main.cu:339)
        scene.render_tiles_device(cam, w, h, spp, md, tiles.data_ptr(), tiles=tr, count_work=count_work, stream=stream)
        gathered = D.gather_tiles(tiles, w, h, rank, world)     # the path's single collective (N > 1)
        if rank == 0:
            D.assemble_device(gathered, w, h, world, frame, stream)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def frame_sha():
        if rank != 0:
            return None
        f = frame.cpu().numpy().copy()
        f[np.isnan(f)] = np.float32(np.nan)                      # one NaN bit pattern (a NaN's payload is not part of the image)
        return hashlib.sha256(f.tobytes()).hexdigest()

    # counted pass (untimed): deterministic workload => these counts are exactly the timed steps' counts
    scene.reset_counters()
    step(count_work=True)
    fence()
    scene.last_kernel_ms()                                       # raises if the counted launch did not complete its frame
    sha_counted = frame_sha()
    cnt = scene.counters()
    gnodes = scene.global_node_fetches() if variant == "megakernel" else 0
    cvec = torch.tensor([cnt[k] for k in api.COUNTER_KEYS], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(cvec)
    total = dict(zip(api.COUNTER_KEYS, (int(v) for v in cvec.tolist())))

    for _ in range(max(0, warmup - 1)):                          # the counted pass is the first warm-up
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    # per-launch megakernel time from HIP events on its stream (last launch; all launches are identical);
    # raises if the tile queue reported an incomplete frame
    kernel_ms = scene.last_kernel_ms()
    sha_timed = frame_sha()
    et = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
    elapsed, _ = et.tolist()
    if rank != 0:
        scene.close()
        return None, None, None
    if sha_timed != sha_counted:
        raise SystemExit("bench.py: the timed frame differs from the counted frame (%s vs %s) on %s" % (sha_timed, sha_counted, workload))

    flags = scene.flags()
    rays = total["rays_closest"] + total["rays_shadow"]
    n_px = w * h
    kname = kernel_name(flags, variant)
    # roofline of the dominant kernel (the megakernel) on THIS rank: what it did / its duration
    own_bytes = RF.alg_bytes(cnt, tr.count * 64)
    entry = RF.find_entry(RF.load_inputs(), workload, spp, kname) if (world == 1 and variant == "megakernel") else None
    # counters of a PMC pass describe the code they were measured on: use them only if the library being timed holds that very code
    stale = entry is not None and not RF.profile_is_current(entry, ctx["code_hashes"])
    if stale:
        entry = None
    if variant != "megakernel":
        rf = {"bound": "l1_lines", "achieved": None, "peak": RF.PEAK_L1_LINES / 1e9, "unit": "Gline-access/s", "frac": None,
              "note": "A/B variant: ~200 launches per frame, no single dominant launch; see value / ms_per_step against the megakernel's"}
    elif not flags["onchip"]:
        busy = RF.ta_busy(entry)                                 # from the committed PMC pass of this workload: how busy the unit behind this bound was
        rf = {"ta_busy_frac": busy}                              # leads the block: the independent evidence that the unit behind the bound is saturated
        rf.update(RF.l1_roofline(gnodes, cnt["tri_tests"], kernel_ms))
        rf["peak_source"] = "micro-benchmark tools/ta_rate/quad_fetch.hip (one cache-line access per CU per clock), not a figure of MI355X_MICROARCH.md"
        rf["global_node_fetches_per_launch"] = gnodes
        rf["global_tri_tests_per_launch"] = cnt["tri_tests"]
        if entry is not None:
            rf["profile"] = {"file": "profiles/roofline_inputs.json", "workload": workload, "spp": spp, "source": entry.get("source")}
    else:
        rf = RF.valu_roofline(entry, kernel_ms)
        if rf is None:                                           # no PMC pass of this exact workload / kernel / code is committed
            rf = {"bound": "valu", "achieved": None, "peak": RF.PEAK_VALU / 1e9, "unit": "Gwave-instr/s", "frac": None}
        else:
            rf["profile"] = {"file": "profiles/roofline_inputs.json", "workload": workload, "spp": spp, "source": entry.get("source")}
    if stale:
        rf["stale_profile"] = True                               # a PMC pass exists, but of other code: re-run tools/profile_round.sh
    issue = RF.issue_block(entry, 8 if "hbm_simple" in kname else (6 if "megakernel_hbm" in kname else 4)) if variant == "megakernel" else None
    if issue is not None:
        rf["issue"] = issue                                      # every instruction class against the CU's measured issue ceiling
    rf["code_sha256"] = ctx["code_hashes"].get(kname)
    rf["traffic"] = RF.traffic_bytes(entry)
    rf["kernel"] = kname
    rf["kernel_ms"] = kernel_ms
    rf["algorithmic"] = RF.algorithmic(own_bytes, kernel_ms)
    out = {
        "metric": "Mray/s", "value": rays * steps / elapsed / 1e6, "unit": "Mray/s",
        "msample_per_s": n_px * spp * steps / elapsed / 1e6,
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": (workload if not spp_override else workload + " [spp overridden to %d]" % spp) +
                               (" [opt-in box culling: not the reference's visiting set]" if culling else ""),
                   "resolution": [w, h], "spp": spp, "max_depth": md, "integrator": "UNIDIRECTIONAL (MIS)", "variant": variant, "seed": api.SEED,
                   "triangles": info["n_tris"], "bvh_nodes": info["n_nodes"], "kernel_flags": flags, "options": opts,
                   "sharding": ("interleaved 8x8 tiles, 1 gather" + (" [REHEARSAL: all ranks share cuda:0, gloo]" if share else "")) if world > 1 else "none",
                   "rays_per_step": rays, "box_tests_per_step": total["box_tests"], "tri_tests_per_step": total["tri_tests"]},
        "frame_sha": sha_timed, "frame_equals_counted_frame": True,
        "roofline": rf,
    }
    if ctx.get("project_shares") and world == 1 and variant == "megakernel":
        # what N GPUs could reach on this frame before the gather: kernel time of the full frame / kernel time of rank 0's 1/N share
        proj = {}
        for n in (2, 4, 8):
            trn = api.rank_tiles(w, h, 0, n)
            best = None
            for _ in range(2):                                   # the first launch of a share's kernel also loads its code
                tiles.zero_()
                scene.render_tiles_device(cam, w, h, spp, md, tiles.data_ptr(), tiles=trn, stream=stream)
                torch.cuda.synchronize()
                best = scene.last_kernel_ms() if best is None else min(best, scene.last_kernel_ms())
            proj[str(n)] = kernel_ms / best
        out["projected_scaling"] = dict(proj, note="one-GPU share measurements (kernel only, no gather): a projection, not a scaling curve")
    scene.close()
    return out, sinfo, info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell_1920x1080_1024spp_depth8_mis", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override spp (diagnostics only; the JSON then says so)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C3 / C4 secondary measurements (N = 1)")
    ap.add_argument("--variant", default="megakernel", choices=["megakernel", "wavefront"],
                    help="kernel organisation (SURVEY §8 f-1 A/B); the headline is the megakernel")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="diagnostic: pt_set_option on the scene (kernel selection / scheduling A/B, include/pt_api.h); the JSON lists them")
    ap.add_argument("--culling", action="store_true",
                    help="diagnostic: opt-in box culling (pt_set_culling) - NOT the reference's visiting set, never the headline")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    # PT_BENCH_SHARE_GPU=1 is a REHEARSAL mode for a 1-GPU box: every rank uses cuda:0 and the gather
    # goes through gloo; the JSON says so. The real N > 1 run is one GPU per rank over RCCL.
    share = os.environ.get("PT_BENCH_SHARE_GPU") == "1"
    dev = 0 if share else local_rank
    torch.cuda.set_device(dev)
    red_dev = "cpu" if share else "cuda"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))

    from cudapathtracer_amd import api
    ctx = {"rank": rank, "world": world, "red_dev": red_dev, "share": share, "code_hashes": RF.kernel_code_hashes(api.LIB_PATH)}
    opts = api.parse_options(args.opt)
    ctx["project_shares"] = world == 1 and not args.opt and not args.culling and not args.no_secondary
    out, sinfo, info = measure(ctx, args.workload, args.spp, args.steps, args.warmup, opts, args.variant, args.culling)
    ctx["project_shares"] = False

    plain = world == 1 and not args.spp and not args.opt and not args.culling and args.variant == "megakernel"
    if rank == 0 and plain and not args.no_secondary and args.workload == "cornell_1920x1080_1024spp_depth8_mis":
        # Driver-visible numbers beyond the headline, one counted + one timed step each, labelled: the kernel that serves
        # scenes in HBM (BASELINE C3 at its full 1024 spp; the C4 / C5 scene at 32 of its 4096 spp, depth 16), the GENERAL
        # bounce (material dispatch, medium stack, dielectric / GGX / mirror arms) on an LDS-resident scene at the headline's
        # size and on the 82 k-triangle scene, and C5: the wavefront variant on the C4 scene.
        sec = []
        for wl, spp, variant in (("blob82k_1920x1080_1024spp_depth8_mis", 0, "megakernel"), ("atrium262k_1920x1080_4096spp_depth16_mis", 32, "megakernel"),
                                 ("cornell_mixed_1920x1080_1024spp_depth8_mis", 0, "megakernel"), ("blob82k_glass_1920x1080_1024spp_depth8_mis", 128, "megakernel"),
                                 ("atrium262k_1920x1080_4096spp_depth16_mis", 32, "wavefront")):
            r, _, _ = measure(ctx, wl, spp, 1, 1, {}, variant, False)
            sec.append({k: r[k] for k in ("value", "unit", "msample_per_s", "steps", "ms_per_step", "config", "frame_sha", "frame_equals_counted_frame", "roofline")})
        out["secondary"] = sec
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sinfo["config"], info)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
