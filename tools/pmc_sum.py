#!/usr/bin/env python3
"""Sum the counters of tools/pmc_passes.sh per kernel: tools/pmc_sum.py outdir [kernel-substring]."""
import csv, glob, json, os, sys
out = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else "megakernel"
tot = {}
for f in sorted(glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if pat not in r["Kernel_Name"]: continue
            k = (r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])
            tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
res = {}
for (kn, c), v in sorted(tot.items()): res.setdefault(kn, {})[c] = v
print(json.dumps(res, indent=1))
