"""bench.py's N > 1 path as the driver launches it — `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` —
rehearsed on a one-GPU box: PT_BENCH_SHARE_GPU=1 puts both ranks on cuda:0 and the gather on gloo (the JSON says so).

What this can and cannot show: the rank / tile arithmetic, the rendezvous, one process per rank, the gather into rank
slots, the de-interleave and the JSON contract are the real code; the RCCL transport over xGMI between two DEVICES is not
exercised here (this pool hands out one GPU). The frame of the two-rank run must be the one-rank frame bit for bit
(`frame_sha`): a pixel's stream is keyed by its global index (deviceCode.cu:59-60), so sharding cannot reach the image.

The file sorts first on purpose: its child processes are started before this (parent) process has touched the GPU.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _bench(n, extra_env=None):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **(extra_env or {}))
    args = ["bench.py", "--gpus", str(n), "--steps", "1", "--warmup", "1", "--spp", "8", "--no-cpu-baseline", "--no-secondary"]
    if n > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + args
    else:
        cmd = [sys.executable] + args
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, "bench.py --gpus %d failed:\n%s\n%s" % (n, p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                       # rank 0 prints ONE JSON line, the other ranks none
    return json.loads(lines[0])


def test_bench_two_ranks_render_the_one_rank_frame():
    one = _bench(1)
    two = _bench(2, {"PT_BENCH_SHARE_GPU": "1"})
    for b, n in ((one, 1), (two, 2)):
        assert b["metric"] == "Mray/s" and b["unit"] == "Mray/s" and b["n_gpus"] == n and b["steps"] == 1 and b["scaling"] == "strong"
        assert b["value"] > 0 and b["ms_per_step"] > 0 and b["frame_equals_counted_frame"] and b["config"]["spp"] == 8
        assert b["config"]["workload"].startswith("cornell_1920x1080_1024spp_depth8_mis [spp overridden to 8]")
    assert one["config"]["sharding"] == "none"
    assert two["config"]["sharding"].startswith("interleaved 8x8 tiles, 1 gather") and "REHEARSAL" in two["config"]["sharding"]
    assert two["frame_sha"] == one["frame_sha"]                     # sharding cannot reach the image
    assert two["config"]["rays_per_step"] == one["config"]["rays_per_step"]      # ... nor the work: counters summed over the ranks
