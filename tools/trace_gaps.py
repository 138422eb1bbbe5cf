#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace: busy time vs. gaps between consecutive kernels, per kernel name."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = [r for r in rows if pat in r[2]]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
print(f"kernels {len(rows)} busy {busy/1e3:.1f} us span {span/1e3:.1f} us gaps {(span-busy)/1e3:.1f} us")
by = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    k = n.split("(")[1] if n.startswith("(anon") else n
    k = n.replace("(anonymous namespace)::", "").split("(")[0]
    by[k][0] += 1; by[k][1] += e - s
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:28s} calls {c:5d} total {t/1e3:9.1f} us avg {t/c/1e3:8.1f} us")
