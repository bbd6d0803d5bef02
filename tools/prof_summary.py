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


# ---- HBM traffic of the dominant (walk) kernel for bench.py's roofline.traffic -------------------------------------------
# MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a
# wide (16 B/lane) coalesced read stream -> doubled.  Collected in separate --pmc passes (tools/prof.sh).
import json
vals = {}
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        tot = defaultdict(float)
        n = defaultdict(set)
        for r in csv.DictReader(open(f)):
            targs = [a.strip() for a in r["Kernel_Name"].split("<")[-1].split(">")[0].split(",")]
            if r["Counter_Name"] == name and "k_grav_walk" in r["Kernel_Name"] and len(targs) > 1 and targs[1] == "false":
                k = short(r["Kernel_Name"])
                tot[k] += float(r["Counter_Value"])
                n[k].add(r["Dispatch_Id"])
        for k in tot:
            vals.setdefault(k, {})[name] = tot[k] / len(n[k])
for k, v in vals.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        out = {"kernel": k, "fetch_bytes_per_launch": 2 * 1024 * v["FETCH_SIZE"], "write_bytes_per_launch": 1024 * v["WRITE_SIZE"],
               "hbm_bytes_per_launch": 2 * 1024 * v["FETCH_SIZE"] + 1024 * v["WRITE_SIZE"],
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KiB -> bytes; FETCH_SIZE doubled "
                         "(gfx950 correction for 16 B/lane reads, MI355X_MICROARCH.md section HBM)"}
        print("== walk traffic:", json.dumps(out))
        with open(os.path.join(root, "walk_traffic_%s.json" % k.split("<")[0]), "w") as fo:
            json.dump(out, fo, indent=1)
