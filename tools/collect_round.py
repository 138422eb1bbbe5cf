#!/usr/bin/env python3
"""Bring what tools/profile_round.sh TAG wrote under gpurun_out/TAG (scratch) into profiles/ (tracked):

    python tools/collect_round.py TAG

  profiles/TAG_bench.json                 the bench line (and TAG_bench_atrium262k_fullspp.json if that run exists)
  profiles/TAG_kernel_stats.csv           rocprofv3 --kernel-trace --stats of the same command
  profiles/TAG_pmc_<workload>.json        the summed counters of the PMC passes per workload (tools/pmc_sum.py)
  profiles/roofline_inputs.json           the entries of this round appended, their `source` pointing at the files above
"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench.json"))
for f in glob.glob(os.path.join(src, "bench_*_fullspp.json")):
    shutil.copy(f, os.path.join(dst, tag + "_" + os.path.basename(f)))
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
stats = glob.glob(os.path.join(src, "stats_headline", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats_headline.csv"))
inputs = json.load(open(os.path.join(dst, "roofline_inputs.json")))
have = {(e.get("workload"), e.get("spp"), e.get("kernel"), e.get("code_sha256")) for e in inputs["entries"]}
for ef in sorted(glob.glob(os.path.join(src, "entry_*.json"))):
    name = os.path.basename(ef)[len("entry_"):-len(".json")]
    out = os.path.join(dst, "%s_pmc_%s.json" % (tag, name))
    summed = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_sum.py"), os.path.join(src, "pmc_" + name)], capture_output=True, text=True, check=True).stdout
    open(out, "w").write(summed)
    e = json.load(open(ef))
    e["source"] = os.path.relpath(out, ROOT)
    key = (e.get("workload"), e.get("spp"), e.get("kernel"), e.get("code_sha256"))
    if key in have:                                      # the same code measured again: the newer pass replaces the older one
        inputs["entries"] = [x for x in inputs["entries"] if (x.get("workload"), x.get("spp"), x.get("kernel"), x.get("code_sha256")) != key]
    inputs["entries"].append(e)
    have.add(key)
    print("entry", name, e["kernel"], (e.get("code_sha256") or "")[:12], "kernel_ms", e.get("kernel_ms"))
json.dump(inputs, open(os.path.join(dst, "roofline_inputs.json"), "w"), indent=1)
print(len(inputs["entries"]), "entries in profiles/roofline_inputs.json")
