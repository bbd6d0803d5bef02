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
    name = name.replace("(anonymous namespace)::", "")
    name = name.split("(")[0]
    for pre in ("void ", "mpg::"):
        name = name.replace(pre, "")
    return name[:70]


for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True) + glob.glob(os.path.join(root, "trace_hydro", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, root))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print("  %-70s calls %6s  total %10.3f ms  avg %10.3f ms  %5s%%" % (
            short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"]))

# ---- one walk (bench.py's roofline "launch") from the kernel trace: kernels 6 run as k_walk_lists / k_walk_eval pairs over slices,
# on two streams; a walk's duration is the span from its first list kernel to the end of its last evaluation kernel.  This is the
# figure that must agree with roofline.avg_launch_ms (HIP events inside bench.py); the per-kernel averages above overlap each other.
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    walks, cur = [], None
    for r in rows:
        nm = r["Kernel_Name"]
        w = "k_walk_lists8<" in nm or "k_walk_eval<" in nm
        if w:
            if cur is None:
                cur = dict(t0=int(r["Start_Timestamp"]), t1=0, count=("k_walk_lists8<true" in nm), n=0)
            cur["t1"] = max(cur["t1"], int(r["End_Timestamp"]))
            cur["n"] += 1
        elif cur is not None and "rocclr" not in nm and "k_grav_walk" not in nm:
            walks.append(cur)
            cur = None
    if cur is not None:
        walks.append(cur)
    timed = [w for w in walks if not w["count"]]
    if timed:
        spans = [(w["t1"] - w["t0"]) / 1e6 for w in timed]
        print("== walks (first list kernel start .. last evaluation kernel end):", len(walks), "of which", len(timed), "without counters")
        print("  span ms:", " ".join("%.2f" % x for x in spans), "  kernels per walk:", timed[-1]["n"])
        # the headline steps come first in the command's run (set-up walk, warm-up, timed steps); the legs behind them (sub-steps, host
        # path in slices, other inputs, hydro) are walks of other sizes
        head = spans[1:1 + int(os.environ.get("MPG_HEADLINE_WALKS", "3"))] or spans[-2:]
        print("  headline walks (spans 2 .. %d): %.3f ms per walk" % (1 + len(head), sum(head) / len(head)))

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
WALK_KERNELS = {"1": ("k_grav_walk<",), "4": ("k_grav_walk_coop<",), "6": ("k_walk_lists8<", "k_walk_eval<")}
METHOD = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KiB -> bytes; FETCH_SIZE doubled "
          "(gfx950 correction for 16 B/lane reads, MI355X_MICROARCH.md section HBM); summed over the dispatches of one walk")


def is_count_build(kname):
    # the COUNT template flag (instrumented untimed pass) is the 2nd parameter of k_grav_walk*, the 1st of k_walk_lists
    targs = [a.strip() for a in kname.split("<")[-1].split(">")[0].split(",")]
    if "k_walk_lists8<" in kname:
        return targs[0] == "true"
    if "k_walk_eval<" in kname:
        return False
    return len(targs) > 1 and targs[1] == "true"


variants = {}
for var, pats in WALK_KERNELS.items():
    tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
    walks = {"FETCH_SIZE": 0, "WRITE_SIZE": 0}
    for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
            ndisp = defaultdict(set)
            for r in csv.DictReader(open(f)):
                kn = r["Kernel_Name"]
                if r["Counter_Name"] != name or not any(p_ in kn for p_ in pats) or is_count_build(kn):
                    continue
                if var == "1" and "k_grav_walk_" in kn:
                    continue
                tot[name] += float(r["Counter_Value"])
                ndisp[[p_ for p_ in pats if p_ in kn][0]].add(r["Dispatch_Id"])
            if ndisp:
                # dispatches of the first kernel of the group per walk: 1 for kernels 1 and 4, ceil(N / 2^21) slices for 6
                per_walk = int(os.environ.get("MPG_SLICES_PER_WALK", "1")) if var == "6" else 1   # 256^3 is one slice (list capacity 1024)
                first = len(ndisp[pats[0]])   # (the list kernel opens a walk)
                walks[name] += first / per_walk
    if walks["FETCH_SIZE"] and walks["WRITE_SIZE"]:
        fb = 2 * 1024 * tot["FETCH_SIZE"] / walks["FETCH_SIZE"]
        wb = 1024 * tot["WRITE_SIZE"] / walks["WRITE_SIZE"]
        variants[var] = {"kernel": " + ".join(p_.rstrip("<") for p_ in pats), "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                         "hbm_bytes_per_launch": fb + wb, "walks_profiled": walks["FETCH_SIZE"], "method": METHOD}
if variants:
    # the library these counters belong to and the input set: bench.py reports the traffic only for the same build (mpg_build_stamp)
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stamp, ic = None, None
    try:
        stamp = open(os.path.join(here, "mp-gadget_amd", "libmpgadget_hip.so.stamp")).read().strip()
    except OSError:
        pass
    try:
        line = [x for x in open(os.path.join(root, "bench_trace.json")) if x.startswith("{")][-1]
        jl = json.loads(line)
        wl = jl["config"]["workload"]
        ic = [k for k in ("s_grid", "s_zel", "s_clust") if k in wl][0]
        # list entries of one walk (leaf + node entries): bench.py prices the ranks of a multi-GPU run per entry
        rf = jl.get("roofline", {})
        if "6" in variants and rf.get("leaf_entries_per_launch") and rf.get("node_entries_per_launch"):
            variants["6"]["list_entries_per_launch"] = rf["leaf_entries_per_launch"] + rf["node_entries_per_launch"]
    except (OSError, IndexError, KeyError, ValueError):
        pass
    print("== walk traffic (%s, build %s):" % (ic, stamp), json.dumps(variants))
    with open(os.path.join(root, "walk_traffic.json"), "w") as fo:
        json.dump({"build_stamp": stamp, "by_ic": {ic or "unknown": variants}}, fo, indent=1)


# ---- HBM traffic of the SPH kernels (bench.py --workload hydro) for its roofline.traffic ---------------------------------------------
# The largest dispatch of each kernel is a full launch (all gas targets); the Hsml iteration's later passes are smaller and are left out.
try:
    stamp = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mp-gadget_amd", "libmpgadget_hip.so.stamp")).read().strip()
except OSError:
    stamp = None
sph = {}
for kern in ("k_density", "k_hydro"):
    val = {}
    for name, sub in (("FETCH_SIZE", "pmc_hydro_fetch"), ("WRITE_SIZE", "pmc_hydro_write")):
        per = defaultdict(float)
        for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == name and short(r["Kernel_Name"]) == kern:
                    per[r["Dispatch_Id"]] += float(r["Counter_Value"])
        if per:
            full = sorted(per.values())[-3:]          # the full launches (one per step) are the largest
            val[name] = sum(full) / len(full)
    if len(val) == 2:
        fb, wb = 2 * 1024 * val["FETCH_SIZE"], 1024 * val["WRITE_SIZE"]
        sph[kern] = {"fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb, "method": METHOD.replace("summed over the dispatches of one walk", "mean of the three largest dispatches (full launches)")}
if sph:
    print("== SPH traffic (build %s):" % stamp, json.dumps(sph))
    with open(os.path.join(root, "sph_traffic.json"), "w") as fo:
        json.dump({"build_stamp": stamp, "kernels": sph}, fo, indent=1)
