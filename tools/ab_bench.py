#!/usr/bin/env python3
"""Interleaved A/B of kernel builds: runs bench.py once per variant per round in subprocesses
(PT_LIB_PATH picks the build) and prints Mray/s + megakernel ms. Usage:
    python tools/ab_bench.py --spp 128 --rounds 2 [--golden] name1 name2 ...
"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=128)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--workload", default="cornell_1920x1080_1024spp_depth8_mis")
ap.add_argument("--golden", action="store_true", help="also run the golden parity tests on each build")
ap.add_argument("names", nargs="+")
a = ap.parse_args()
res = {n: [] for n in a.names}
for r in range(a.rounds):
    for n in a.names:
        base, _, opt = n.partition("+")              # "v3+defer" = lib_v3.so with --opt defer_shadow=1
        lib = os.path.join(ROOT, "cudapathtracer_amd", "csrc", "variants", "lib_%s.so" % base)
        env = dict(os.environ, PT_LIB_PATH=lib)
        extra = []
        if opt == "defer":
            extra = ["--opt", "defer_shadow=1"]
        elif "=" in opt:                             # "base+compact=1": any pt_set_option
            extra = ["--opt", opt]
        if a.golden and r == 0:
            t = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-q", "-x", "-m", "gpu", "-k", "golden or fresh or deep"], cwd=ROOT, env=env, capture_output=True, text=True)
            print(n, "parity:", t.stdout.strip().splitlines()[-1] if t.stdout.strip() else t.stderr[-300:], flush=True)
        p = subprocess.run([sys.executable, "bench.py", "--spp", str(a.spp), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--workload", a.workload] + extra,
                           cwd=ROOT, env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(n, "FAILED", p.stderr[-400:], flush=True); continue
        j = json.loads(line[-1])
        res[n].append((j["value"], j["roofline"]["kernel_ms"]))
        print("round %d %-12s %9.1f Mray/s  kernel %8.2f ms" % (r, n, j["value"], j["roofline"]["kernel_ms"]), flush=True)
print("\nsummary (best kernel ms):")
for n, v in res.items():
    if v: print("  %-12s %9.1f Mray/s  %8.2f ms" % (n, max(x[0] for x in v), min(x[1] for x in v)))
