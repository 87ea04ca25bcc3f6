#!/usr/bin/env python3
"""Summarises a rocprofv3 --pmc counter_collection.csv: mean counter value per launch, per kernel.
usage: pmc_summary.py <dir-or-csv> [out.csv]"""
import csv, glob, os, sys
from collections import defaultdict

src = sys.argv[1]
files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: defaultdict(float))
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        acc[(k, row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
lines = ["kernel,counter,launches,mean_per_launch"]
for (k, c), d in sorted(acc.items()):
    name = '"%s"' % k if "," in k else k   # template arguments: lk_fb_kernel<7, 21>
    lines.append("%s,%s,%d,%.1f" % (name, c, len(d), sum(d.values()) / len(d)))
out = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
print(out)
