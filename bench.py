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
a counted, untimed pass of the identical (deterministic) workload. `roofline` prices the megakernel
against the 8 TB/s HBM peak with SURVEY.md §8(d)'s ALGORITHMIC bytes and the kernel's own duration
from HIP events on its launch stream. `cpu_baseline` is the CPU oracle on a bounded sample of the
same workload (rank 0, N = 1 only) — a reported reference point, not the target.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # name: (generator, kwargs)  — C2 is the headline; C3/C4 are selectable for profiling
    "cornell_1920x1080_1024spp_depth8_mis": ("cornell", dict(width=1920, height=1080, spp=1024, max_depth=8)),
    "blob82k_1920x1080_1024spp_depth8_mis": ("blob_in_box", dict(width=1920, height=1080, spp=1024, max_depth=8)),
    "atrium262k_1920x1080_4096spp_depth16_mis": ("atrium", dict(width=1920, height=1080, spp=4096, max_depth=16)),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s


def alg_bytes(c, n_px):
    """32 B/box test, 16 B/node pop, 52 B/triangle test, 96 B/accepted hit, 16 B/pixel."""
    return 32 * c["box_tests"] + 16 * c["node_pops"] + 52 * c["tri_tests"] + 96 * c["hits"] + 16 * n_px


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


def pmc_traffic(workload, spp):
    """HBM bytes per megakernel launch from the committed rocprofv3 PMC passes (profiles/), if they
    were taken on this workload; None otherwise. bench.py cannot run the profiler on itself."""
    try:
        for t in json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["entries"]:
            if t.get("workload") == workload and t.get("spp") == spp:
                return t["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell_1920x1080_1024spp_depth8_mis", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override spp (diagnostics only; the JSON then says so)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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

    from cudapathtracer_amd import api, scenes
    from cudapathtracer_amd import distributed as D

    gen, kw = WORKLOADS[args.workload]
    tmp = tempfile.mkdtemp(prefix="ptbench_r%d_" % rank)
    sinfo = getattr(scenes, gen)(tmp, **kw)
    host = api.HostScene(sinfo["config"])
    info = host.info
    w, h, md = info["width"], info["height"], info["max_depth"]
    spp = args.spp or info["spp"]
    cam = host.camera()
    opts = api.parse_options(args.opt)
    scene = api.Scene(host, options=opts).set_variant(args.variant)           # scene resident in HBM from here on
    if args.culling:
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

    # counted pass (untimed): deterministic workload => these counts are exactly the timed steps' counts
    scene.reset_counters()
    step(count_work=True)
    fence()
    cnt = scene.counters()
    cvec = torch.tensor([cnt[k] for k in api.COUNTER_KEYS], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(cvec)
    total = dict(zip(api.COUNTER_KEYS, (int(v) for v in cvec.tolist())))

    for _ in range(max(0, args.warmup - 1)):                    # the counted pass is the first warm-up
        step()
    fence()
    kernel_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    # per-launch megakernel time from HIP events on its stream (last launch; all launches are identical)
    kernel_ms = scene.last_kernel_ms()
    et = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
    elapsed, kernel_ms_max = et.tolist()

    if rank == 0:
        rays = total["rays_closest"] + total["rays_shadow"]
        n_px = w * h
        ms_per_step = elapsed / args.steps * 1e3
        # roofline of the dominant kernel (the megakernel) on THIS rank: its algorithmic bytes / its duration
        own_bytes = alg_bytes(cnt, tr.count * 64)
        achieved = own_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        out = {
            "metric": "Mray/s", "value": rays * args.steps / elapsed / 1e6, "unit": "Mray/s",
            "msample_per_s": n_px * spp * args.steps / elapsed / 1e6,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (args.workload if not args.spp else args.workload + " [spp overridden to %d]" % spp) +
                                   (" [opt-in box culling: not the reference's visiting set]" if args.culling else ""),
                       "resolution": [w, h], "spp": spp, "max_depth": md, "integrator": "UNIDIRECTIONAL (MIS)", "variant": args.variant, "seed": api.SEED,
                       "triangles": info["n_tris"], "bvh_nodes": info["n_nodes"], "kernel_flags": scene.flags(), "sharding": ("interleaved 8x8 tiles, 1 gather" + (" [REHEARSAL: all ranks share cuda:0, gloo]" if share else "")) if world > 1 else "none",
                       "rays_per_step": rays, "box_tests_per_step": total["box_tests"], "tri_tests_per_step": total["tri_tests"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(args.workload, spp) if (world == 1 and args.variant == "megakernel") else None,
                         "kernel": (("pt::megakernel_hbm<0, false, %s, %s>" % (_tf(scene.flags()["culling"]), _tf(scene.flags()["refill"]))) if scene.flags()["hbm_kernel"] else "pt::megakernel<0, false, false, %s, %s, %s>" % (_tf(scene.flags()["onchip"]), _tf(scene.flags()["refill"]), _tf(scene.flags()["flat"]))) if args.variant == "megakernel" else "pt::wf_logic_kernel + pt::wf_trace_kernel (all launches of one frame)", "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": own_bytes},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sinfo["config"], info)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
