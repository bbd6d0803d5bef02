#!/usr/bin/env python3
"""Summarise rocprofv3 output directories written by tools/prof.sh: per-kernel time (kernel-trace stats) and
per-kernel PMC sums/averages.  Usage: prof_summary.py <dir>"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    name = name.split("(")[0]
    for pre in ("void ", "mpg::"):
        name = name.replace(pre, "")
    return name[:70]


for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, root))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print("  %-70s calls %6s  total %10.3f ms  avg %10.3f ms  %5s%%" % (
            short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"]))

for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        print("== counters:", os.path.relpath(f, root))
        agg = defaultdict(lambda: defaultdict(float))
        calls = defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add(r["Dispatch_Id"])
        for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[:6]:
            n = len(calls[k])
            print("  %-60s dispatches %d" % (k, n))
            for c, v in sorted(agg[k].items()):
                print("      %-28s per-dispatch %.6g" % (c, v / n))
