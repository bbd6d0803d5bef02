#!/usr/bin/env python3
"""bench.py -- particle-updates/s of one TreePM gravity force step on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over the synchronised particle set, all particles active (a PM step of
run.c:522-548): gravpm_force (CIC deposit, 5 FFTs, transfers, readout) + force_tree_full (tree build + moments)
+ grav_short_tree (short-range walk with the relative opening criterion, OldAcc from the previous step).
Inputs are resident in HBM when the timed region starts.

N = 1 : 256^3 dark-matter particles, Nmesh = 512 (BASELINE.json configs[1]), Zel'dovich-displaced synthetic ICs.
N > 1 : weak scaling, ~256^3 particles per GPU (n = 320 / 400 / 512 per dimension for N = 2 / 4 / 8, Nmesh = 2n): particles on the
        owners of their Peano-Hilbert TopLeaves, PM by shipping particles to x-slabs, ghost import, all-reduced top of the tree
        (the library's choreography, csrc/dist.hip); the line carries `parity_check`, the forces of sampled
        particles against the one-GPU path.

One JSON line is printed by rank 0 (contract of the task statement), with `roofline` for the dominant kernel
(the short-range walk; HIP events on the engine stream inside the timed region) and `cpu_baseline` (the oracle
built with the reference's compiler flags, timed on the host cores of this box on a bounded sample).
"""
import argparse
import importlib
import json
import os
import sys
import math
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

G = 43.0071
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec


def emit(out):
    """The ONE JSON line, as the last thing on stdout: RCCL prints a version banner through C stdio, which sits in libc's buffer
    until exit when stdout is a pipe - flush it first."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


FP64_VALU_PEAK_TF = 78.6   # MI355X_MICROARCH.md: fp64 vector peak (256 CUs x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz)
FLOP_PER_INTERACTION = 38  # flop-equivalents of one particle-particle or particle-node evaluation


def walk_roofline(eng, cnt, walk_ms, walk_launches, traffic, traffic_note):
    """`roofline` of the dominant kernel, the short-range walk.  The ceiling that binds it is fp64 vector issue, not HBM
    (pairwise 1/r^2 with a table lookup: the gathers are served by L2/LDS), so `achieved` is the reference-required flops
    38 x (pair interactions + nodes used) per walk over the walk's duration from HIP events on the engine stream; the node
    tests of the traversal, which the reference also performs, are NOT counted as flops.  The HBM side is reported next to it:
    algorithmic bytes (SURVEY 8(d) B_walk), measured HBM traffic (PMC) as a fraction of the 8 TB/s peak, and their ratio."""
    variant, list_cap, list_ovf = eng.walk_choice()
    kernels = {1: "k_grav_walk", 4: "k_grav_walk_coop", 6: "k_walk_lists8 + k_walk_eval",
               7: "k_walk_leaf"}.get(variant, "?")
    t = walk_ms / max(walk_launches, 1) * 1e-3
    flops = (cnt["pp"] + cnt["nodes_used"]) * float(FLOP_PER_INTERACTION)
    b_alg = cnt["targets"] * 64 + cnt["pp"] * 28 + cnt["nodes_visited"] * 72   # B_walk
    ach = flops / t / 1e12
    r = {"bound": "fp64_valu", "kernel": kernels, "achieved": ach, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
         "frac": ach / FP64_VALU_PEAK_TF, "traffic": traffic, "traffic_note": traffic_note,
         "flop_per_launch": flops, "flop_per_interaction": FLOP_PER_INTERACTION,
         "avg_launch_ms": t * 1e3, "launches_timed": walk_launches,
         "pp_interactions_per_launch": cnt["pp"], "nodes_visited_per_launch": cnt["nodes_visited"],
         "nodes_used_per_launch": cnt["nodes_used"], "targets_per_launch": cnt["targets"],
         "algorithmic_bytes_per_launch": b_alg,
         "hbm_measured_frac": (traffic / t / 1e9 / HBM_PEAK_GBS) if traffic else None,
         "reuse": (b_alg / traffic) if traffic else None,
         "walk_variant": variant, "list_capacity": list_cap, "targets_to_fallback_kernel": list_ovf,
         "children_per_node_step": round(cnt["node_lanes"] / max(cnt["node_steps"], 1), 2),
         "node_steps_per_launch": cnt["node_steps"], "node_lanes_per_launch": cnt["node_lanes"],
         "leaf_entries_per_launch": cnt["int_steps"] if variant == 6 else None,
         "node_entries_per_launch": cnt["int_lanes"] if variant == 6 else None,
         # round 6: target passes of the list kernel whose fp32 pre-classification was ambiguous and that ran the fp64 tests instead
         # (counted by the COUNT builds; every other pass took its decisions from fp32 - DESIGN 3.2, round 6)
         "fp32_fallback_passes_per_launch": cnt.get("f32_fallback") if variant == 6 else None,
         "note": "one launch = one short-range walk over all targets; the walk is bound by fp64 VALU issue (pairwise kernel with a "
                 "per-pair window-table lookup; MFMA does not apply), so frac = 38 flop x (N_pp + N_nodes_used) / t / 78.6 TFLOP/s; "
                 "hbm_measured_frac = PMC traffic / t / 8 TB/s; reuse = SURVEY 8(d) B_walk / PMC traffic"}
    return r


def quick_gravity_steps(pkg, torch, eng, ic, n, nmesh, dev, steps=3):
    """ms per force step of another input set of SURVEY 8(d) on the already configured engine (device-resident, as the headline)."""
    pos, mass, box = getattr(pkg.ics, ic)(n)
    N = len(pos)
    d_pos, d_mass = torch.from_numpy(pos).to(dev), torch.from_numpy(mass).to(dev)
    z3 = lambda: torch.zeros(N, 3, dtype=torch.float64, device=dev)
    gravpm, acc, prev, pot = z3(), z3(), z3(), torch.zeros(N, dtype=torch.float64, device=dev)
    eng.gravpm_init_periodic(box, 1.5, nmesh, G)
    eng.set_gravshort_treepar(ErrTolForceAcc=0.002, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=2, Rcut=6.0,
                              FractionalGravitySoftening=1. / 30.)
    eng.gravshort_set_softenings(box / n)
    eng.dev_bind_particles(d_pos, d_mass, box)

    def step():
        nonlocal acc, prev
        eng.dev_gravpm_force(gravpm, pot)
        eng.dev_force_tree_build()
        prev, acc = acc, prev
        eng.dev_grav_short_tree(acc, prev_accel=prev, gravpm=gravpm, potential=pot)
    for _ in range(3):     # Barnes-Hut first walk, list-capacity adaptation, one relative-criterion walk
        step()
    torch.cuda.synchronize()
    eng.walk_events_collect()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    wms, wl = eng.walk_events_collect()
    return {"ms_per_step": round(1e3 * el / steps, 3), "particles_per_s": N * steps / el, "walk_ms": round(wms / max(wl, 1), 3), "steps": steps}


def host_path_steps(pkg, eng, pos, mass, box, steps=3):
    """SURVEY 8(d)'s metric as the reference's callers see it: the drop-in (host pointer) calls on struct particle_data records in
    host memory, results written back into them - PCIe transfers and AoS packing included.  Not `value`.  Measured twice: with the
    overlap the shim switches on (mpg_set_host_overlap: one packing pass per epoch, OldAcc on the device, gravpm_force's results written
    into P[] by a host thread while the tree build and the walk run; everything is in P[] when grav_short_tree returns) - `ms_per_step` -
    and with every call finishing its own transfers before it returns - `synchronous`."""
    eng.set_host_overlap(True)
    out = _host_path_steps(pkg, eng, pos, mass, box, steps)
    # round 6: the same calls with the step's packing pass + uploads started EARLY (mpg_host_prefetch, what shim/timestep-hip.c does at the end of
    # drift_all_particles) and `host_gap_ms` of host time between that point and gravpm_force - run.c:420-522 spends far more there
    # (domain_maintain, the active list; density and hydro_force in a gas run).  ms_per_step counts the three force calls only: what the
    # reference's step waits for when the gap hides the upload.  `ms_per_step` above (no prefetch) stays the drop-in's headline number.
    pre = _host_path_steps(pkg, eng, pos, mass, box, steps, prefetch_gap=0.04)
    out["prefetched"] = {k: pre[k] for k in ("ms_per_step", "particles_per_s", "calls_ms", "prefetch_call_ms", "host_gap_ms")}
    eng.set_host_overlap(False)
    sync = _host_path_steps(pkg, eng, pos, mass, box, steps)
    out["synchronous"] = {k: sync[k] for k in ("ms_per_step", "particles_per_s", "calls_ms")}
    # (the same three columns of P[] by both routes; not bit for bit: the CIC deposit sums with atomics)
    a, b = out.pop("_columns"), sync.pop("_columns")
    out["max_rel_difference_to_synchronous"] = float(max(np.abs(x - y).max() / np.abs(y).max() for x, y in zip(a, b)))
    return out


def _host_path_steps(pkg, eng, pos, mass, box, steps=3, prefetch_gap=None):
    P = pkg.make_particles(pos, mass)
    N = len(pos)
    ts = []
    tpre = []
    for it in range(steps + 2):
        # what shim/gravity-hip.c does at every entry point (mpg_shim_sync): one table epoch per step, so that Pos / Mass / Type go up
        # once per step and not once per call (rounds 1-3 timed three uploads per step here)
        eng.set_particle_epoch(it + 1)
        if prefetch_gap is not None:
            ta = time.perf_counter()
            eng.host_prefetch(P, box)
            tpre.append(time.perf_counter() - ta)
            time.sleep(prefetch_gap)            # (stands for the host's own work between the drift and the first force call)
        t0 = time.perf_counter()
        eng.gravpm_force(P)
        t1 = time.perf_counter()
        eng.force_tree_full(P, box)      # (with shim/forcetree-hip.c this is the whole cost of run.c:546: no host tree is built)
        t2 = time.perf_counter()
        eng.grav_short_tree(P)
        t3 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2))
    eng.set_particle_epoch(0)
    ts = np.array(ts[2:])      # the first two steps allocate the pinned staging and run the Barnes-Hut walk
    tot = ts.sum(1).mean()
    return {"ms_per_step": round(1e3 * tot, 2), "particles_per_s": N / tot,
            "prefetch_call_ms": round(1e3 * float(np.mean(tpre[2:])), 3) if tpre else None,
            "host_gap_ms": None if prefetch_gap is None else 1e3 * prefetch_gap,
            "_columns": (P["GravPM"].copy(), P["FullTreeGravAccel"].copy(), P["Potential"].copy()),
            "calls_ms": {"gravpm_force": round(1e3 * ts[:, 0].mean(), 2), "force_tree_full": round(1e3 * ts[:, 1].mean(), 2),
                         "grav_short_tree": round(1e3 * ts[:, 2].mean(), 2)},
            "note": "mpg_gravpm_force + mpg_force_tree_full + mpg_grav_short_tree on %d 160-byte particle_data records in pageable host "
                    "memory, one table epoch per step as the shim declares it (one packing pass and one H2D of Pos / Mass / Type / Potential / "
                    "FullTreeGravAccel per step, GravPM / FullTreeGravAccel / Potential down, packing on host threads) = the force part of "
                    "run.c:522-548 with shim/ in the link, no host tree anywhere; the device-resident rate is `value`" % N}


def resident_path_steps(pkg, eng, pos, mass, box, steps=3):
    """The drop-in calls in the device-resident mode (mpg_resident_begin): the same three host-pointer calls on the same 160-byte records,
    but the table stays in HBM between them (one upload, timed separately); the host fetches FullTreeGravAccel when a module of its own
    needs it (timed separately)."""
    P = pkg.make_particles(pos, mass)
    N = len(pos)
    t0 = time.perf_counter()
    eng.resident_begin(P, box)
    t_begin = time.perf_counter() - t0
    ts = []
    for it in range(steps + 2):
        t0 = time.perf_counter()
        eng.gravpm_force(P)
        eng.force_tree_full(P, box)
        eng.grav_short_tree(P)
        eng.synchronize()
        ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    eng.resident_fetch(P, eng.FIELD_ACCEL)
    t_fetch = time.perf_counter() - t0
    t0 = time.perf_counter()
    eng.resident_end(P)
    t_end = time.perf_counter() - t0
    tot = float(np.mean(ts[2:]))
    return {"ms_per_step": round(1e3 * tot, 2), "particles_per_s": N / tot, "upload_once_ms": round(1e3 * t_begin, 1),
            "fetch_accel_ms": round(1e3 * t_fetch, 1), "end_fetch_all_ms": round(1e3 * t_end, 1),
            "mean_abs_accel": float(np.abs(P["FullTreeGravAccel"]).mean()),
            "note": "mpg_resident_begin once, then mpg_gravpm_force + mpg_force_tree_full + mpg_grav_short_tree on the same %d records per step "
                    "with the table resident in HBM (drift / kicks through mpg_dev_* on mpg_resident_arrays); fetch_accel_ms = "
                    "mpg_resident_fetch(FullTreeGravAccel) into P[] on demand" % N}


def configure_gravity(eng, args, box, n, nmesh):
    eng.use_torch_stream()
    eng.set_walk_threshold(args.thresh)
    eng.set_walk_variant(args.variant)
    eng.gravshort_fill_ntab(0, 1.5)
    eng.gravpm_init_periodic(box, 1.5, nmesh, G)
    # (timing experiments on ONE kernel with wrong results must not change the lists through the opening input: MPG_BENCH_TREEUSEBH=1
    # walks every step with the geometric criterion, MPG_BENCH_BHANGLE sets its angle; diagnostics only, the line says so in `config`)
    eng.set_gravshort_treepar(ErrTolForceAcc=0.002, BHOpeningAngle=float(os.environ.get("MPG_BENCH_BHANGLE", "0.175")), MaxBHOpeningAngle=0.9,
                              TreeUseBH=int(os.environ.get("MPG_BENCH_TREEUSEBH", "2")), Rcut=6.0, FractionalGravitySoftening=1. / 30.)
    eng.gravshort_set_softenings(box / n)


def walk_traffic(pkg, N, ic, variant):
    """HBM bytes per walk from the committed PMC passes (tools/prof.sh -> profiles/walk_traffic.json), valid only for the library
    they were taken with: the file carries the build stamp (mpg_build_stamp) of that library, and a line produced by another build
    reports no traffic instead of a stale one."""
    tpath = os.path.join(ROOT, "profiles", "walk_traffic.json")
    if not os.path.exists(tpath) or N != 256 ** 3:
        return None, "no PMC summary committed for this configuration"
    tj = json.load(open(tpath))
    stamp = pkg.engine.load_library().mpg_build_stamp().decode()
    if tj.get("build_stamp") != stamp:
        return None, ("profiles/walk_traffic.json was measured with library build %s, this run uses %s: re-run tools/prof.sh"
                      % (str(tj.get("build_stamp"))[:12], stamp[:12]))
    e = tj.get("by_ic", {}).get(ic, {}).get(str(variant))
    if not e:
        return None, "no PMC summary committed for this input set"
    return e["hbm_bytes_per_launch"], "bytes per walk (%s) from %s; library build %s" % (e["kernel"], e["method"], stamp[:12])


def _pmc_child_runs(child_args, kernels):
    """{counter: {kernel key: {dispatch id: counter value}}} of `python bench.py --traffic-child <child_args>` run under rocprofv3 --pmc
    FETCH_SIZE and --pmc WRITE_SIZE (separate passes, no tracing domain: MI355X_MICROARCH.md, HBM / rocprofv3).  kernels: {key: (substring
    of the kernel name, predicate on the full name or None)}.  Raises RuntimeError with the reason when a pass cannot be had."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        raise RuntimeError("rocprofv3 not found")
    tmp = tempfile.mkdtemp(prefix="mpg_traffic_", dir="/tmp")
    res = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [rp, "--output-format", "csv", "--pmc", counter, "-d", out, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--traffic-child"] + list(child_args)
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                raise RuntimeError("rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, (r.stderr or "")[-200:].replace("\n", " ")))
            per = {k: {} for k in kernels}
            for f in files:
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] != counter:
                        continue
                    kn = row["Kernel_Name"]
                    for k, (sub, pred) in kernels.items():
                        if sub in kn and (pred is None or pred(kn)):
                            d = int(row["Dispatch_Id"])
                            per[k][d] = per[k].get(d, 0.0) + float(row["Counter_Value"])
            res[counter] = per
    except (OSError, subprocess.SubprocessError, KeyError, ValueError) as e:
        raise RuntimeError("live PMC passes failed: %r" % (e,))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return res


PMC_METHOD = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE on two child processes of this command (separate passes); KiB -> bytes, FETCH_SIZE "
              "x 2 and WRITE_SIZE x 1 as calibrated on this rocprofv3 with known byte counts in the kernels' own access patterns "
              "(tools/pmc_calib.hip, profiles/r05_calib/calibration.txt)")


def live_walk_traffic(args):
    """HBM bytes per walk measured IN THIS RUN: two child processes of this very command (--traffic-child: the headline workload, set-up +
    1 warm-up + 2 steps, nothing else) under rocprofv3, and the counters of the walk's two kernels averaged over the last two walks.
    FETCH_SIZE x 2, WRITE_SIZE x 1: calibrated (0.500 for 16-B, 4-B, grouped 4-B and 32-B gather reads alike; 1.000 / 1.07 for 4-B streaming
    / run appends).  Returns (bytes, note) or (None, why)."""
    kernels = {"lists": ("k_walk_lists8<", lambda kn: kn.split("k_walk_lists8<")[1].split(",")[0].strip() != "true"),   # (not the counting build)
               "eval": ("k_walk_eval<", None)}
    try:
        res = _pmc_child_runs(["--ic", args.ic, "--n", str(args.n or 256), "--thresh", str(args.thresh), "--variant", str(args.variant)], kernels)
    except RuntimeError as e:
        return None, str(e)
    if any(len(res[c][k]) < 2 for c in res for k in kernels):
        return None, "the profiled child ran fewer than two walks"
    last = lambda m: sum(m[d] for d in sorted(m)[-2:]) / 2.0
    f = {k: 2.0 * 1024.0 * last(res["FETCH_SIZE"][k]) for k in kernels}
    w = {k: 1024.0 * last(res["WRITE_SIZE"][k]) for k in kernels}
    note = ("measured in this run: %s; the last two walks of each pass; per walk: k_walk_lists8 fetched %.2f + wrote %.2f GB, k_walk_eval "
            "fetched %.2f + wrote %.2f GB" % (PMC_METHOD, f["lists"] / 1e9, w["lists"] / 1e9, f["eval"] / 1e9, w["eval"] / 1e9))
    return sum(f.values()) + sum(w.values()), note


def live_sph_traffic(n, PE):
    """HBM bytes per full launch of k_density / k_hydro measured in this run (see live_walk_traffic): the mean of the three largest dispatches
    of each kernel in a child that runs the hydro workload (the Hsml iteration's later passes are smaller launches).
    Returns {kernel: (bytes, note)} or {kernel: (None, why)}."""
    kernels = {"k_density": ("k_density(", None), "k_hydro": ("k_hydro(", None)}
    try:
        res = _pmc_child_runs(["--workload", "hydro", "--n", str(n), "--sph", "pe" if PE else "de"], kernels)
    except RuntimeError as e:
        return {k: (None, str(e)) for k in kernels}
    out = {}
    for k in kernels:
        if not res["FETCH_SIZE"][k] or not res["WRITE_SIZE"][k]:
            out[k] = (None, "kernel not seen by the profiled child")
            continue
        top3 = lambda m: sum(sorted(m.values())[-3:]) / len(sorted(m.values())[-3:])
        f, w = 2.0 * 1024.0 * top3(res["FETCH_SIZE"][k]), 1024.0 * top3(res["WRITE_SIZE"][k])
        out[k] = (f + w, "measured in this run: %s; mean of the three largest dispatches (full launches): fetched %.2f + wrote %.2f GB"
                  % (PMC_METHOD, f / 1e9, w / 1e9))
    return out


def sph_traffic(pkg, n, kernel):
    """HBM bytes per full launch of k_density / k_hydro from the committed PMC passes (tools/prof.sh -> profiles/sph_traffic.json), for
    the library build they were taken with only (see walk_traffic)."""
    tpath = os.path.join(ROOT, "profiles", "sph_traffic.json")
    if not os.path.exists(tpath) or n != 128:
        return None, "no PMC summary committed for this configuration"
    tj = json.load(open(tpath))
    stamp = pkg.engine.load_library().mpg_build_stamp().decode()
    if tj.get("build_stamp") != stamp:
        return None, "profiles/sph_traffic.json was measured with library build %s, this run uses %s: re-run tools/prof.sh" % (
            str(tj.get("build_stamp"))[:12], stamp[:12])
    e = tj.get("kernels", {}).get(kernel)
    if not e:
        return None, "kernel not in the PMC summary"
    return e["hbm_bytes_per_launch"], "bytes per full launch from %s; library build %s" % (e["method"], stamp[:12])


def substep_measure(pkg, torch, eng, N, acc, prev, gravpm, pot, dev, fracs=(1. / 8, 1. / 64, 1. / 512), reps=3):
    """SURVEY 8(d) metric (ii), the short-range-only step, as production runs spend most of their walks (timestep.c:296-599): an
    ActiveParticle list of N f randomly chosen particles (a) on the tree of ALL particles (the sub-steps of run.c:392-470: tree build +
    walk of the active ones) and (b) on a tree of the active particles only (force_tree_active_moments, the hierarchical gravity
    levels).  Times per call in ms (device-resident), active particles per second for (a)."""
    g = torch.Generator(device=dev).manual_seed(99)
    out = {}
    for f in fracs:
        na = max(int(N * f), 1)
        act = torch.randperm(N, device=dev, generator=g)[:na].sort().values.to(torch.int32).contiguous()
        res = {}
        for name, full in (("all_particle_tree", True), ("active_only_tree", False)):
            def one():
                if full:
                    eng.dev_force_tree_build()
                else:
                    eng.dev_force_tree_active_moments(act)
                eng.dev_grav_short_tree(acc, prev_accel=prev, gravpm=gravpm, potential=pot if full else None, active=act)
            one()
            torch.cuda.synchronize()
            eng.walk_events_collect()
            t0 = time.perf_counter()
            for _ in range(reps):
                one()
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / reps
            wms, wl = eng.walk_events_collect()
            res[name] = {"ms_per_substep": round(1e3 * el, 3), "walk_ms": round(wms / max(wl, 1), 3), "walk_kernel": eng.walk_choice()[0],
                         "active_per_s": na / el}
        out["1/%d" % round(1 / f)] = dict(active=na, **res)
    eng.dev_force_tree_build()      # leave the tree of all particles in place
    return out


FLOP_PER_DENSITY_NGB, FLOP_PER_HYDRO_PAIR, FLOP_PER_CANDIDATE = 60, 120, 8   # flop-equivalents of density_ngbiter / hydro_ngbiter / one distance test


def hydro_measure(pkg, torch, args, dev, n=128, steps=3, PE=0):
    """BASELINE configs[2] (2 x n^3 DM + gas, density-entropy SPH): ms per force step (gravity + gas tree + density + hmax + hydro) and
    the two SPH kernels graded against the fp64 vector peak (their neighbour arithmetic is what the reference requires; the gathers are
    served by L2)."""
    nmesh = 2 * n
    pos, mass, typ, box = hydro_ics(pkg, n)
    N = len(pos)
    f8 = torch.float64
    d_pos, d_mass, d_type = torch.from_numpy(pos).to(dev), torch.from_numpy(mass).to(dev), torch.from_numpy(typ).to(dev)
    eng = pkg.Engine(dev.index or 0)
    eng.use_torch_stream()
    eng.set_walk_variant(args.variant)
    eng.gravshort_fill_ntab(0, 1.5)
    eng.gravpm_init_periodic(box, 1.5, nmesh, G)
    eng.set_gravshort_treepar(TreeUseBH=2)
    eng.gravshort_set_softenings(box / n)
    eng.set_densitypar(1.0, 2.0, 2.0, 99999., pkg.engine.DENSITY_KERNEL_QUINTIC_SPLINE, 0.006)
    eng.set_hydropar(PE, 100.0, 0.75)
    eng.dev_bind_particles(d_pos, d_mass, box, type=d_type)
    z1 = lambda: torch.zeros(N, dtype=f8, device=dev)
    z3 = lambda: torch.zeros(N, 3, dtype=f8, device=dev)
    a = dict(hsml=z1(), dthsml=z1(), vel=z3(), entropy=torch.ones(N, dtype=f8, device=dev), density=z1(), egywtdensity=z1(),
             dhsmlegyfac=z1(), divvel=z1(), curlvel=z1(), hydroacc_out=z3(), dtentropy_out=z1(), maxsignalvel=z1())
    gravpm, acc, prev, pot = z3(), z3(), z3(), z1()
    t = pkg.SphTimes()
    t.atime, t.hubble = 0.1, 0.1
    for i in range(47):
        t.dloga_bin[i] = 0.01
    eng.dev_force_tree_rebuild_mask(pkg.engine.GASMASK + pkg.engine.BHMASK, with_moments=True)
    eng.dev_set_init_hsml(a, box / n)
    iters = []

    def step():
        nonlocal acc, prev
        eng.dev_gravpm_force(gravpm, pot)
        eng.dev_force_tree_build()
        prev, acc = acc, prev
        eng.dev_grav_short_tree(acc, prev_accel=prev, gravpm=gravpm, potential=pot)
        eng.dev_force_tree_rebuild_mask(pkg.engine.GASMASK)
        eng.dev_density(a, t, DoEgyDensity=PE)
        iters.append(eng.sph_stats()["iterations"])
        eng.dev_force_tree_calc_hmax()
        eng.dev_hydro_force(a, t)

    for _ in range(max(args.warmup, 2)):      # (the first density loop converges Hsml from set_init_hsml's estimate: several passes)
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # untimed pass with events around the phases
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
    ev[0].record()
    eng.dev_gravpm_force(gravpm, pot)
    eng.dev_force_tree_build()
    eng.dev_grav_short_tree(acc, prev_accel=prev, gravpm=gravpm, potential=pot)
    ev[1].record()
    eng.dev_force_tree_rebuild_mask(pkg.engine.GASMASK)
    ev[2].record()
    eng.dev_density(a, t, DoEgyDensity=PE)
    sd = eng.sph_stats()
    ev[3].record()
    eng.dev_force_tree_calc_hmax()
    ev[4].record()
    eng.dev_hydro_force(a, t)
    sh = eng.sph_stats()
    ev[5].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(5)]
    ngas = n ** 3
    b_dens = sd["targets"] * 128 + sd["candidates"] * 28 + sd["interactions"] * 32
    b_hyd = ngas * 176 + sh["candidates"] * 36 + sh["interactions"] * 100

    live = {} if (args.no_live_traffic or args.traffic_child) else live_sph_traffic(n, PE)

    def roof(kernel, flops, b, t_ms, note):
        ach = flops / (t_ms * 1e-3) / 1e12
        traffic, tnote = sph_traffic(pkg, n, kernel)
        if live.get(kernel, (None,))[0] is not None:
            traffic, tnote = live[kernel]
        elif kernel in live:
            tnote += "; live measurement not available: " + live[kernel][1]
        return {"bound": "fp64_valu", "kernel": kernel, "achieved": ach, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": ach / FP64_VALU_PEAK_TF,
                "traffic": traffic, "traffic_note": tnote, "flop_per_launch": flops, "algorithmic_bytes_per_launch": b,
                "algorithmic_bytes_over_hbm_peak": b / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": t_ms, "note": note}
    out = {"ms_per_step": 1e3 * el / steps, "particles_per_s": N * steps / el, "particles": N, "steps": steps,
           "workload": "2x%d^3 DM+gas TreePM + %s SPH force step, Nmesh=%d, s_zel ICs, quintic kernel" % (n, "pressure-entropy" if PE else "density-entropy", nmesh),
           "density_iterations": iters[-steps:],
           "roofline": roof("k_density", sd["interactions"] * FLOP_PER_DENSITY_NGB + sd["candidates"] * FLOP_PER_CANDIDATE, b_dens, ms[2],
                            "one density() incl. queue set-up, predictions and every Hsml pass; flops = %d x N_ngb + %d x N_cand (%d neighbours, %d "
                            "candidates, %d target visits); algorithmic bytes (SURVEY 8(d) B_dens) are served by L2, not HBM: their ratio to the "
                            "HBM peak is reported, not graded" % (FLOP_PER_DENSITY_NGB, FLOP_PER_CANDIDATE, sd["interactions"], sd["candidates"], sd["targets"])),
           "roofline_hydro": roof("k_hydro", sh["interactions"] * FLOP_PER_HYDRO_PAIR + sh["candidates"] * FLOP_PER_CANDIDATE, b_hyd, ms[4],
                                  "flops = %d x N_pair + %d x N_cand (%d pairs, %d candidates); B_hyd as SURVEY 8(d)"
                                  % (FLOP_PER_HYDRO_PAIR, FLOP_PER_CANDIDATE, sh["interactions"], sh["candidates"])),
           "phases_ms": {"gravity_pm_tree_walk": round(ms[0], 3), "gas_tree": round(ms[1], 3), "density": round(ms[2], 3),
                         "hmax": round(ms[3], 3), "hydro": round(ms[4], 3)}}
    eng.close()
    del d_pos, d_mass, d_type, a, gravpm, acc, prev, pot
    torch.cuda.empty_cache()
    return out


def hydro_bench(pkg, torch, args, dev):
    """--workload hydro: BASELINE.json configs[2] as its own line"""
    n = args.n or 128
    PE = 1 if args.sph == "pe" else 0
    h = hydro_measure(pkg, torch, args, dev, n=n, steps=args.steps, PE=PE)
    out = {"metric": "particle-updates/sec (gravity + SPH force step)", "value": h["particles_per_s"], "unit": "particles/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": h["ms_per_step"], "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": h["workload"], "particles": h["particles"], "density_iterations": h["density_iterations"]},
           "roofline": h["roofline"], "roofline_hydro": h["roofline_hydro"], "phases_ms": h["phases_ms"]}
    emit(out)
    return out


def parity_check_ranks(pkg, torch, dist, args, dev, rank, world, host_pos, host_mass, box, n, nmesh, own_ids, loc, dforce, eng, nsample=2048):
    """The multi-rank force step checks itself (the reference's runs do, through run_gravity_test, runtests.c:89-232): nsample own
    particles, spread over the ranks, are recomputed on ONE GPU (rank 0) from the whole particle set through the single-GPU path the
    parity tests pin to the oracle - PM of all particles, tree of all particles, walk of the sampled targets with the SAME opening
    input (the previous step's acceleration of those particles as the ranks hold it) - and compared with what the ranks computed:
    GravPM and the short-range acceleration of the last step, and the interaction counters of a walk restricted to the sample
    (pair interactions, nodes visited, nodes used: equal when every rank took the reference's decisions on its local tree).
    Untimed; collective."""
    n_own = int(own_ids.shape[0])
    k = max(nsample // world, 1)
    sel = torch.linspace(0, max(n_own - 1, 0), steps=min(k, n_own), device=dev).long().unique()
    ns = int(sel.shape[0])
    # counters of the N-rank walk restricted to the sample (same tree as the last step: build, then walk the sample with counting on)
    eng.set_instrumentation(False, True)
    acc_s = torch.zeros_like(loc["acc"])
    dforce.force_tree_build(loc["pos"], loc["mass"])
    dforce.grav_short_tree(acc_s, prev_accel=loc["prev"], gravpm=loc["gravpm"], active=sel.to(torch.int32).contiguous())
    cnt = eng.walk_counters()
    eng.set_instrumentation(False, False)
    c_n = torch.tensor([cnt["pp"], cnt["nodes_visited"], cnt["nodes_used"]], dtype=torch.int64, device=dev)
    dist.all_reduce(c_n)
    # (the restricted walk repeats the step's values to rounding: the order of a target's list entries depends on its wave-mates)
    same_walk = float((acc_s[sel] - loc["acc"][sel]).abs().max() / loc["acc"][sel].abs().mean().clamp_min(1e-300)) if ns else 0.0
    # the sample's rows to rank 0: id, prev accel (the opening input), GravPM, acceleration
    rows = torch.cat([own_ids[sel].double()[:, None], loc["prev"][sel], loc["gravpm"][sel], loc["acc"][sel]], dim=1)
    pad = torch.zeros(k, 10, dtype=torch.float64, device=dev)
    pad[:ns] = rows
    cntv = torch.tensor([ns], dtype=torch.int64, device=dev)
    allc = [torch.zeros_like(cntv) for _ in range(world)]
    allr = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(allc, cntv)
    dist.all_gather(allr, pad)
    res = None
    if rank == 0:
        got = torch.cat([allr[r][:int(allc[r].item())] for r in range(world)])
        ids = got[:, 0].long()
        N = len(host_pos)
        e1 = pkg.Engine(dev.index or 0)
        e1.use_torch_stream()
        configure_gravity(e1, args, box, n, nmesh)
        e1.set_gravshort_treepar(ErrTolForceAcc=0.002, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=0, Rcut=6.0,
                                 FractionalGravitySoftening=1. / 30.)
        p1, m1 = torch.from_numpy(host_pos).to(dev), torch.from_numpy(host_mass).to(dev)
        e1.dev_bind_particles(p1, m1, box)
        g1 = torch.zeros(N, 3, dtype=torch.float64, device=dev)
        a1 = torch.zeros(N, 3, dtype=torch.float64, device=dev)
        pv = torch.zeros(N, 3, dtype=torch.float64, device=dev)
        pv[ids] = got[:, 1:4]
        e1.dev_gravpm_force(g1, None)
        e1.dev_force_tree_build()
        e1.set_instrumentation(False, True)
        order = torch.argsort(ids)
        tg = ids[order].to(torch.int32).contiguous()
        e1.dev_grav_short_tree(a1, prev_accel=pv, gravpm=g1, active=tg)
        torch.cuda.synchronize()
        c1 = e1.walk_counters()
        d_pm = (g1[ids] - got[:, 4:7]).norm(dim=1) / g1[ids].norm(dim=1).mean()
        an = a1[ids].norm(dim=1)
        d_a = (a1[ids] - got[:, 7:10]).norm(dim=1) / an.clamp_min(1e-300)
        cn = [int(x) for x in c_n.cpu()]
        c1v = [c1["pp"], c1["nodes_visited"], c1["nodes_used"]]
        res = {"n": int(ids.shape[0]), "max_rel": float(d_a.max()), "median_rel": float(d_a.median()),
               "p999_rel": float(torch.quantile(d_a, 0.999)), "gravpm_max_rel_to_mean": float(d_pm.max()),
               "worst_over_mean_accel": float(((a1[ids] - got[:, 7:10]).norm(dim=1)).max() / an.mean()),
               # pair interactions and nodes used unopened must be EQUAL (the same interaction sets); nodes visited can only be fewer
               # on the ranks: a top-level cell none of whose particles a rank needs is absent from its local tree, where the global
               # walk visits and discards it
               "counters_equal": cn[0] == c1v[0] and cn[2] == c1v[2] and cn[1] <= c1v[1], "counters_ranks": cn, "counters_one_gpu": c1v,
               "counters": "[pair interactions, nodes visited, nodes used unopened] of a walk restricted to the sample", "restricted_walk_max_rel_rank0": same_walk,
               "gate": "SURVEY 8(d): median <= 1e-12, 99.9 % <= 1e-9, no particle worse than 2 ErrTolForceAcc <|a|>; GravPM <= 1e-11 <|GravPM|>",
               "reference": "the same particles recomputed on one GPU from all %d particles through the single-GPU path (pinned to the oracle "
                            "by tests/test_gpu_gravity.py); opening input = the ranks' previous acceleration" % N}
        res["ok"] = bool(res["median_rel"] <= 1e-12 and res["p999_rel"] <= 1e-9 and res["worst_over_mean_accel"] <= 2 * 0.002 and
                         res["gravpm_max_rel_to_mean"] <= 1e-11 and res["counters_equal"])
        e1.close()
    return res


def ranks_roofline(pkg, torch, dist, dev, rank, world, eng, cnt, walk_ms, walk_launches, n_own, ghosts, ic):
    """Every rank's walk against the two ceilings (collective; the same arithmetic as walk_roofline): fp64 vector issue from the
    rank's own counters and HIP events, and HBM - `hbm_measured_frac` where a PMC summary of this build exists: the bytes per list
    entry measured on one GPU (profiles/walk_traffic.json) x the rank's list entries / its walk time / 8 TB/s (PMC counters cannot be
    collected inside a timed multi-process run; the per-entry figure is what the one-GPU passes measured for the same kernels)."""
    t = walk_ms / max(walk_launches, 1) * 1e-3
    flops = (cnt["pp"] + cnt["nodes_used"]) * float(FLOP_PER_INTERACTION)
    entries = float(cnt["int_steps"] + cnt["int_lanes"])
    row = torch.tensor([t * 1e3, flops, entries, float(n_own), float(ghosts), float(torch.cuda.current_device()),
                        float(cnt["pp"]), float(cnt["nodes_used"])], dtype=torch.float64, device=dev)
    rows = [torch.zeros_like(row) for _ in range(world)]
    dist.all_gather(rows, row)
    rows = torch.stack(rows).cpu().numpy()
    per_entry, note = None, "no PMC summary of this library build: hbm_measured_frac null"
    tpath = os.path.join(ROOT, "profiles", "walk_traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        e = tj.get("by_ic", {}).get(ic, {}).get("6")
        if tj.get("build_stamp") == pkg.engine.load_library().mpg_build_stamp().decode() and e and e.get("list_entries_per_launch"):
            per_entry = e["hbm_bytes_per_launch"] / e["list_entries_per_launch"]
            note = "HBM bytes per list entry (%.1f B) from the one-GPU PMC passes of this build x the rank's entries" % per_entry
    tt = np.maximum(rows[:, 0] * 1e-3, 1e-12)
    return {"walk_ms": [round(float(x), 3) for x in rows[:, 0]],
            "frac": [round(float(x), 4) for x in rows[:, 1] / tt / 1e12 / FP64_VALU_PEAK_TF],
            "hbm_measured_frac": [round(float(x), 4) for x in rows[:, 2] * per_entry / tt / 1e9 / HBM_PEAK_GBS] if per_entry else None,
            "hbm_note": note,
            "own_particles": [int(x) for x in rows[:, 3]], "ghosts": [int(x) for x in rows[:, 4]], "device": [int(x) for x in rows[:, 5]],
            "pp_interactions": [int(x) for x in rows[:, 6]], "nodes_used": [int(x) for x in rows[:, 7]]}


def make_comm(pkg, torch, dist, args, dev, eng):
    """The communicator the multi-rank step runs on: the library's native RCCL communicator (csrc/rccl_comm.hip: ncclSend / ncclRecv /
    ncclAllReduce on the engine's stream, no Python in any collective), bootstrapped through the launcher's torch.distributed group and
    checked once with its self-test.  Should it fail on ANY rank, all ranks fall back to the torch.distributed callbacks and the line says so."""
    note = "torch.distributed callbacks (dist.py::TorchComm)"
    if args.comm == "rccl" and dist.get_backend() == "nccl":
        comm, ok, err = None, 1, ""
        try:
            comm = pkg.dist.RcclComm(eng.lib, dev, selftest_bytes=1 << 20)
        except Exception as e:      # noqa: BLE001 - whatever went wrong, every rank must learn of it
            ok, err = 0, repr(e)
        t = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) == 1:
            return comm, "native RCCL communicator of the library (mpg_rccl_*: ncclGroupStart + ncclSend / ncclRecv per peer, ncclAllReduce, " \
                         "on the engine's stream; RCCL %d)" % comm.stats()["rccl_version"]
        if comm is not None:
            comm.close()
        note = "torch.distributed callbacks (the native RCCL communicator failed on some rank: %s)" % (err or "another rank")
    return pkg.dist.TorchComm(dev), note


def gravity_bench_ranks(pkg, torch, dist, args, dev, rank, world, local_rank):
    """N > 1 (or MPG_FORCE_MGPU: the same code in a one-rank group): weak scaling, ~256^3 particles per GPU, particles on the owners of
    their Peano-Hilbert TopLeaves, everything through the library's choreography (mpg_dist_*, csrc/dist.hip) with the collectives on
    RCCL (dist.py::TorchComm)."""
    n = args.n or {1: 256, 2: 320, 4: 400, 8: 512}.get(world, int(round(256 * world ** (1. / 3) / (8 * world))) * 8 * world)
    nmesh = 2 * n
    pos, mass, box = getattr(pkg.ics, args.ic)(n)
    N = len(pos)
    eng = pkg.Engine(local_rank)
    configure_gravity(eng, args, box, n, nmesh)
    comm, comm_note = make_comm(pkg, torch, dist, args, dev, eng)
    dforce = pkg.dist.DistForce(eng, comm)
    rcut = 6.0 * 1.5 * box / nmesh                        # margin = Rcut * Asmth * cell size (gravshort-tree.c:102)
    # domain_decompose_full + domain_exchange (untimed): every rank starts from a contiguous share of the set
    lo, hi = (N * rank) // world, (N * (rank + 1)) // world
    sp = torch.from_numpy(pos[lo:hi]).to(dev)
    sm = torch.from_numpy(mass[lo:hi]).to(dev)
    sid = torch.arange(lo, hi, dtype=torch.int64, device=dev)
    ntn, ntl = dforce.domain_decompose(sp, box, overdecomposition=args.overdecomp)
    own_pos, own_mass, own_ids = dforce.domain_exchange(sp, sm, sid)
    del sp, sm, sid
    dforce.use_decomposition(box, rcut)
    n_own = int(own_pos.shape[0])
    tot = torch.tensor([n_own], dtype=torch.int64, device=dev)
    dist.all_reduce(tot)
    assert int(tot.item()) == N, "the ranks' own sets do not add up to the particle set (%d of %d)" % (int(tot.item()), N)
    if rank != 0 or args.no_parity_check:
        del pos                                            # (rank 0 keeps the host copy for the parity tail)
    z3 = lambda: torch.zeros(n_own, 3, dtype=torch.float64, device=dev)
    loc = dict(pos=own_pos, mass=own_mass, acc=z3(), prev=z3(), gravpm=z3(), pot=torch.zeros(n_own, dtype=torch.float64, device=dev), steps=0)

    def rank_sums(x):
        """max over ranks / mean over ranks of a per-rank number"""
        t = torch.zeros(world, dtype=torch.float64, device=dev)
        t[rank] = float(x)
        dist.all_reduce(t)
        return float(t.max() / t.mean())

    def step():
        loc["prev"], loc["acc"] = loc["acc"], loc["prev"]
        # the first step has no previous acceleration: Barnes-Hut opening (TreeUseBH = 2), as the reference's first step
        dforce.gravity_step(loc["pos"], loc["mass"], loc["acc"], loc["gravpm"], potential=loc["pot"], prev_accel=loc["prev"] if loc["steps"] else None)
        loc["steps"] += 1

    def rebalance():
        """domain_decompose_full again, now with the work the last walk measured per particle as the cost the TopLeaves are balanced
        by (domain.c:611), and the exchange of the particles with their last acceleration, which the relative criterion needs"""
        nonlocal own_ids, n_own
        cost = dforce.walk_cost(n_own)
        loc["work_before"] = rank_sums(cost.sum().item())
        loc["count_before"] = rank_sums(n_own)
        dforce.domain_decompose(loc["pos"], box, overdecomposition=args.overdecomp, cost=cost)
        p, m, ids, pa, pg = dforce.domain_exchange(loc["pos"], loc["mass"], own_ids, loc["acc"], loc["gravpm"])
        own_ids, n_own = ids, int(p.shape[0])
        dforce.use_decomposition(box, rcut)
        loc.update(pos=p, mass=m, acc=pa.contiguous(), prev=torch.zeros_like(pa), gravpm=pg.contiguous(),
                   pot=torch.zeros(n_own, dtype=torch.float64, device=dev))
        step()
        loc["work_after"] = rank_sums(dforce.walk_cost(n_own).sum().item())
        loc["count_after"] = rank_sums(n_own)

    def sync():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    step()          # set-up, not a step: first-use allocations, FFT plans, the list-capacity adaptation of the walk
    if not args.no_rebalance:
        step()      # (a walk with the relative criterion: its per-particle work is what the domains are balanced by)
        rebalance()
    for _ in range(args.warmup):
        step()
    sync()
    eng.walk_events_collect()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    t1 = time.perf_counter()
    walk_ms, walk_launches = eng.walk_events_collect()
    dt = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    elapsed = float(dt.item())
    # ---- untimed: phase times of one more step, interaction counters of another, then the self-check
    tm_sum = None
    eng.set_instrumentation(True, False)
    step()
    sync()
    ph = eng.phase_times()
    st, tm = dforce.stats(), dforce.times()
    eng.set_instrumentation(False, True)
    step()
    sync()
    cnt = eng.walk_counters()
    cnt["f32_fallback"] = eng.walk_f32_stats()[0] if os.environ.get("MPG_LISTS_F32") == "1" else None
    eng.set_instrumentation(False, False)
    eng.walk_events_collect()
    per_rank = ranks_roofline(pkg, torch, dist, dev, rank, world, eng, cnt, walk_ms, walk_launches, n_own, st["ghosts"], args.ic)
    rccl = comm.info() if hasattr(comm, "info") else None
    if rccl is not None and rccl["nranks"] != world:
        raise SystemExit("bench.py: RCCL reports %d ranks in the communicator, the launcher started %d" % (rccl["nranks"], world))
    parity = None
    if not args.no_parity_check:
        parity = parity_check_ranks(pkg, torch, dist, args, dev, rank, world, pos if rank == 0 else None, mass, box, n, nmesh, own_ids, loc, dforce, eng)
    out = None
    if rank == 0:
        value = N * args.steps / elapsed
        traffic, traffic_note = None, "per-rank walk; no PMC summary for multi-rank runs"
        out = {
            "metric": "particle-updates/sec (gravity force step: PM + tree build + short-range walk)",
            "value": value, "unit": "particles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d^3 DM-only TreePM force step, Nmesh=%d, %s ICs, all particles active, relative opening "
                                   "criterion (ErrTolForceAcc 0.002), TreeRcut 6, Asmth 1.5" % (n, nmesh, args.ic),
                       "particles": N, "nmesh": nmesh,
                       "parallelism": "%d GPUs: particles on the owners of their Peano-Hilbert TopLeaves (domain_decompose_full, %d TopLeaves); per step "
                                      "{Pos, Mass} shipped to the x-slab PM (2 all-to-all transposes + neighbour planes) and {GravPM, Potential} back, "
                                      "ghosts imported in whole level-La tree cells within Rcut of the rank's TopLeaves, top of the tree from an "
                                      "all-reduce; choreography in the library (mpg_dist_*), collectives on RCCL" % (world, ntl),
                       "communicator": comm_note,
                       "ghost_fraction_rank0": round(st["ghosts"] / max(n_own, 1), 3), "decomposition_level_La": st["La"]},
            "roofline": walk_roofline(eng, cnt, walk_ms, walk_launches, traffic, traffic_note),
            "phases_ms": {k: round(v, 3) for k, v in ph.items()},
        }
        out["roofline"]["note"] += "; rank 0's walk over its own particles (counters and events of rank 0); per_rank = the same of every rank"
        out["roofline"]["per_rank"] = per_rank
        out["config"]["rccl_ranks"] = rccl["nranks"] if rccl is not None else None     # ncclCommCount of the library's communicator
        out["config"]["devices"] = per_rank["device"]
        if "work_after" in loc:
            out["config"]["load_balance"] = {
                "walk_work_max_over_mean": round(loc["work_after"], 4), "particles_max_over_mean": round(loc["count_after"], 4),
                "by_particle_number": {"walk_work_max_over_mean": round(loc["work_before"], 4),
                                       "particles_max_over_mean": round(loc["count_before"], 4)},
                "note": "TopLeaves dealt to the ranks by the walk work measured per particle; by_particle_number = the same step on the "
                        "decomposition balanced by particle counts"}
        out["phases_ms"].update({"dist_pm_ms": round(tm["pm"], 3), "dist_ghost_import_ms": round(tm["ghosts"], 3),
                                 "dist_tree_and_top_ms": round(tm["tree"], 3), "dist_walk_ms": round(tm["walk"], 3),
                                 "dist_tree_build_beside_pm_ms": round(tm["tree_beside_pm"], 3),
                                 "dist_exchange_bytes": st["exchange_bytes"], "dist_transpose_bytes": st["transpose_bytes"]})
        if parity is not None:
            out["parity_check"] = parity
    if out is not None and hasattr(comm, "stats"):
        out["config"]["communicator_calls"] = comm.stats()
    dist.barrier()
    dist.destroy_process_group()
    dforce.close()
    if hasattr(comm, "close"):
        comm.close()
    eng.close()
    if out is not None:
        emit(out)
        if parity is not None and not parity["ok"]:
            print("bench.py: the multi-rank forces FAILED the self-check against the one-GPU path: %s" % json.dumps(parity), file=sys.stderr, flush=True)
            sys.exit(3)
    return out


def gravity_bench_single(pkg, torch, args, dev, local_rank):
    """N = 1: BASELINE.json configs[1], the configuration the metric is quoted on."""
    n = args.n or 256
    nmesh = 2 * n
    pos, mass, box = getattr(pkg.ics, args.ic)(n)
    N = len(pos)
    d_pos = torch.from_numpy(pos).to(dev)
    d_mass = torch.from_numpy(mass).to(dev)
    del pos
    eng = pkg.Engine(local_rank)
    configure_gravity(eng, args, box, n, nmesh)
    eng.dev_bind_particles(d_pos, d_mass, box)
    z3 = lambda: torch.zeros(N, 3, dtype=torch.float64, device=dev)
    gravpm, acc, prev, pot = z3(), z3(), z3(), torch.zeros(N, dtype=torch.float64, device=dev)

    def step():
        nonlocal acc, prev
        eng.dev_gravpm_force(gravpm, pot)
        eng.dev_force_tree_build()
        prev, acc = acc, prev
        eng.dev_grav_short_tree(acc, prev_accel=prev, gravpm=gravpm, potential=pot)

    step()          # set-up, not a step: first-use allocations, FFT plans, the list-capacity adaptation of the walk and the deposit's timing trial
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    eng.walk_events_collect()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    walk_ms, walk_launches, lists_ms, eval_ms, nsplit = eng.walk_events_collect_split()
    if args.traffic_child:      # (a child of live_walk_traffic under rocprofv3: the walks above are what it wanted)
        eng.close()
        return None
    # ---- untimed diagnostic passes: phase times of one more step, interaction counters of another (the counting builds of the
    # walk kernels are slower: kept out of the phase times)
    eng.set_instrumentation(True, False)
    step()
    torch.cuda.synchronize()
    ph = eng.phase_times()
    eng.set_instrumentation(False, True)
    step()
    torch.cuda.synchronize()
    cnt = eng.walk_counters()
    cnt["f32_fallback"] = eng.walk_f32_stats()[0] if os.environ.get("MPG_LISTS_F32") == "1" else None
    eng.set_instrumentation(False, False)
    eng.walk_events_collect()
    variant = eng.walk_choice()[0]
    traffic, traffic_note = walk_traffic(pkg, N, args.ic, variant)
    if variant == 6 and not args.no_extras and not args.no_live_traffic:
        # the HBM traffic of the walk measured in THIS run (two short child processes under rocprofv3 --pmc); the committed summary of
        # tools/prof.sh is the fall-back
        t0 = time.perf_counter()
        live, live_note = live_walk_traffic(args)
        if live is not None:
            traffic, traffic_note = live, live_note + " (%.0f s)" % (time.perf_counter() - t0)
        else:
            traffic_note += "; live measurement not available: " + live_note
    out = {
        "metric": "particle-updates/sec (gravity force step: PM + tree build + short-range walk)",
        "value": N * args.steps / elapsed, "unit": "particles/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%d^3 DM-only TreePM force step, Nmesh=%d, %s ICs, all particles active, relative opening "
                               "criterion (ErrTolForceAcc 0.002), TreeRcut 6, Asmth 1.5" % (n, nmesh, args.ic),
                   "particles": N, "nmesh": nmesh, "parallelism": "1 GPU"},
        "roofline": walk_roofline(eng, cnt, walk_ms, walk_launches, traffic, traffic_note),
        "phases_ms": {k: round(v, 3) for k, v in ph.items()},
    }
    if nsplit:      # the walk's two kernels one by one (an event between them on the engine stream, inside the timed region)
        out["roofline"]["kernels_ms"] = {"k_walk_lists8": round(lists_ms / nsplit, 3), "k_walk_eval": round(eval_ms / nsplit, 3), "launches_timed": nsplit,
                                         "frac_of_k_walk_eval_alone": round(out["roofline"]["frac"] * (walk_ms / walk_launches) / (eval_ms / nsplit), 4)}
    if os.environ.get("MPG_BENCH_TREEUSEBH") or os.environ.get("MPG_BENCH_BHANGLE"):   # (a diagnostic run: not BASELINE's configuration)
        out["config"]["workload"] += " -- DIAGNOSTIC: TreeUseBH=%s BHOpeningAngle=%s from the environment" % (
            os.environ.get("MPG_BENCH_TREEUSEBH", "2"), os.environ.get("MPG_BENCH_BHANGLE", "0.175"))
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pkg, d_pos.cpu().numpy(), mass, box, n, nmesh, prev.cpu().numpy() + gravpm.cpu().numpy(), args.cpu_sample)
    if not args.no_extras:
        # untimed legs after `value`: the short-range-only sub-steps, the PCIe-inclusive drop-in path, the other input sets of
        # SURVEY 8(d) at the same size, and BASELINE configs[2] (the gas configuration)
        out["substeps"] = substep_measure(pkg, torch, eng, N, acc, prev, gravpm, pot, dev)
        host_pos = d_pos.cpu().numpy()
        out["host_path"] = host_path_steps(pkg, eng, host_pos, mass, box)
        out["resident_path"] = resident_path_steps(pkg, eng, host_pos, mass, box)
        del gravpm, acc, prev, pot, d_pos, d_mass, host_pos
        torch.cuda.empty_cache()
        out["other_inputs"] = {ic: quick_gravity_steps(pkg, torch, eng, ic, n, nmesh, dev)
                               for ic in ("s_grid", "s_zel", "s_clust") if ic != args.ic}
    eng.close()
    if not args.no_extras:
        torch.cuda.empty_cache()
        out["hydro"] = hydro_measure(pkg, torch, args, dev, n=128 if n >= 128 else max(n // 2, 16))
    emit(out)
    return out


def substep_bench(pkg, torch, args, dev, local_rank):
    """--workload substep: SURVEY 8(d) metric (ii) as its own line (value = active particles per second at --active-frac on the tree of
    all particles)."""
    n = args.n or 256
    nmesh = 2 * n
    pos, mass, box = getattr(pkg.ics, args.ic)(n)
    N = len(pos)
    d_pos, d_mass = torch.from_numpy(pos).to(dev), torch.from_numpy(mass).to(dev)
    eng = pkg.Engine(local_rank)
    configure_gravity(eng, args, box, n, nmesh)
    eng.dev_bind_particles(d_pos, d_mass, box)
    z3 = lambda: torch.zeros(N, 3, dtype=torch.float64, device=dev)
    gravpm, acc, prev, pot = z3(), z3(), z3(), torch.zeros(N, dtype=torch.float64, device=dev)
    for _ in range(3):      # Barnes-Hut first walk, list-capacity adaptation, one relative-criterion walk: OldAcc for the sub-steps
        eng.dev_gravpm_force(gravpm, pot)
        eng.dev_force_tree_build()
        prev, acc = acc, prev
        eng.dev_grav_short_tree(acc, prev_accel=prev, gravpm=gravpm, potential=pot)
    fr = sorted(set([args.active_frac, 1. / 8, 1. / 64, 1. / 512]), reverse=True)
    res = substep_measure(pkg, torch, eng, N, acc, prev, gravpm, pot, dev, fracs=fr, reps=max(args.steps, 1))
    key = "1/%d" % round(1 / args.active_frac)
    r = res[key]["all_particle_tree"]
    out = {"metric": "active particle-updates/sec (short-range-only sub-step: tree build + walk of the ActiveParticle list)",
           "value": r["active_per_s"], "unit": "particles/s", "n_gpus": 1, "steps": max(args.steps, 1), "warmup": 1,
           "ms_per_step": r["ms_per_substep"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "%d^3 DM-only, sub-step with %s of the particles active (random), tree of all particles, relative criterion" % (n, key),
                      "particles": N, "active": res[key]["active"]},
           "roofline": {"bound": "fp64_valu", "kernel": {1: "k_grav_walk", 6: "k_walk_lists8 + k_walk_eval"}.get(r["walk_kernel"], "?"), "achieved": None,
                        "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": None, "traffic": None, "avg_launch_ms": r["walk_ms"],
                        "note": "sub-steps are bound by the tree build and the launch of a small walk, not by a roofline: see `substeps`"},
           "substeps": res}
    eng.close()
    emit(out)
    return out


def self_launch(ngpus, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks from here (torch.distributed.run, one process
    per GPU, rendezvous on 127.0.0.1 at a free port) and pass their exit code on.  Refuses - exit code 2, nothing printed on stdout -
    when the box shows fewer than N GPUs: an N-GPU command must never produce a line measured on fewer.  (MPG_DIST_BACKEND=gloo, the
    tests' way of putting several ranks on one GPU, lifts that check: the line then says `communicator: torch.distributed`.)"""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()
    backend = os.environ.get("MPG_DIST_BACKEND", "nccl")
    if ndev < 1 or (backend == "nccl" and ndev < ngpus):
        print("bench.py: --gpus %d needs %d visible GPUs, this box shows %d (HIP_VISIBLE_DEVICES=%s): refusing to run - no line is "
              "printed rather than one measured on fewer GPUs" % (ngpus, ngpus, ndev, os.environ.get("HIP_VISIBLE_DEVICES", "unset")),
              file=sys.stderr, flush=True)
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL's peer mappings need it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("bench.py: --gpus %d without a launcher: starting %d ranks (%s backend, %d GPUs visible)" % (ngpus, ngpus, backend, ndev),
          file=sys.stderr, flush=True)
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", "--size", dest="n", type=int, default=int(os.environ.get("MPG_BENCH_N", "0")),
                    help="particles per dimension (default: 256 per GPU, weak scaling)")
    ap.add_argument("--ic", default="s_zel", choices=["s_grid", "s_zel", "s_clust"],
                    help="synthetic input set (SURVEY 8(d)); s_zel, the Zel'dovich-displaced grid, is the headline set")
    ap.add_argument("--no-extras", action="store_true",
                    help="N = 1: skip the untimed legs (substeps, host_path, resident_path, other_inputs, hydro)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="N = 1: take roofline.traffic from profiles/walk_traffic.json instead of two "
                                                                   "rocprofv3 --pmc child runs of this command")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-parity-check", action="store_true", help="N > 1: skip the untimed self-check of the forces against the one-GPU path")
    ap.add_argument("--cpu-sample", type=int, default=1 << 22, help="targets walked by the CPU baseline")
    ap.add_argument("--thresh", type=int, default=16)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--mgpu", default="peano", choices=["peano"], help="(kept for the command lines of round 2; the x-slab / replicated forms were retired)")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "torch"],
                    help="N > 1: the library's native RCCL communicator (default) or the torch.distributed callbacks of rounds 2-3")
    ap.add_argument("--overdecomp", type=int, default=8, help="DomainOverDecompositionFactor (TopLeaves per rank and policy)")
    ap.add_argument("--no-rebalance", action="store_true",
                    help="N > 1: keep the decomposition by particle number (default: after two set-up steps the TopLeaves are dealt out "
                         "again by the measured work per particle, domain.c:611)")
    ap.add_argument("--sph", default="auto", choices=["auto", "de", "pe"],
                    help="hydro workload: density-entropy (BASELINE configs[2]) or pressure-entropy SPH (configs[4]); auto: de on one GPU, pe on several")
    ap.add_argument("--active-frac", type=float, default=1. / 64, help="substep workload: fraction of the particles that is active")
    ap.add_argument("--workload", default="gravity", choices=["gravity", "hydro", "integrate", "fof", "domain", "substep"],
                    help="gravity: BASELINE.json configs[1] (default, the headline metric; its line also carries the sub-steps and configs[2]); "
                         "hydro: configs[2] / [4] as their own line; substep: the short-range-only step; integrate / fof / domain: SURVEY 8(f) rows")
    args = ap.parse_args()
    if args.traffic_child:
        args.steps, args.warmup, args.no_extras, args.no_cpu_baseline, args.gpus = 2, 1, True, True, 1

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain process (`python bench.py --gpus N`, no launcher): the ranks are launched from here, one per GPU -
        # never a one-GPU line under an N-GPU command
        return self_launch(args.gpus, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE %d != --gpus %d (the launcher's rank count and --gpus must agree)" % (world, args.gpus))

    pkg = importlib.import_module("mp-gadget_amd")
    import torch
    import torch.distributed as dist

    ndev = torch.cuda.device_count()
    if os.environ.get("MPG_DIST_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % max(ndev, 1)
    elif local_rank >= ndev:
        raise SystemExit("bench.py: rank %d (LOCAL_RANK %d) has no GPU of its own: %d device(s) visible; RCCL needs one GPU per rank"
                         % (rank, local_rank, ndev))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # MPG_FORCE_MGPU=1: run the multi-GPU code path (collectives included) in a one-rank group - a single-GPU box can then
    # exercise exactly what the ranks of an N-GPU run execute
    multi = world > 1 or bool(os.environ.get("MPG_FORCE_MGPU"))
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("MPG_DIST_BACKEND", "nccl")   # "gloo" lets several ranks share one GPU in tests
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.workload in ("integrate", "fof", "domain", "substep") and world != 1:
        raise SystemExit("--workload %s is a one-GPU line" % args.workload)
    if args.workload == "integrate":
        return integrate_bench(pkg, torch, args, dev)
    if args.workload == "fof":
        return fof_bench(pkg, torch, args, dev)
    if args.workload == "domain":
        return domain_bench(pkg, torch, args, dev)
    if args.workload == "substep":
        return substep_bench(pkg, torch, args, dev, local_rank)
    if args.workload == "hydro":
        if multi:
            return hydro_bench_peano(pkg, torch, dist, args, dev, rank, world)
        return hydro_bench(pkg, torch, args, dev)
    if multi:
        return gravity_bench_ranks(pkg, torch, dist, args, dev, rank, world, local_rank)
    return gravity_bench_single(pkg, torch, args, dev, local_rank)


def integrate_bench(pkg, torch, args, dev):
    """the streaming loops between force steps on device-resident arrays - apply_PM_half_kick,
    apply_half_kick (gravity + hydro kick), drift_all_particles - for 2 x n^3 particles (half gas).  One step = the three
    loops once.  Pure HBM streaming: the roofline is bytes moved / time against the HBM peak."""
    n = args.n or 256
    N = 2 * n ** 3
    g = torch.Generator(device=dev).manual_seed(1)
    f8 = torch.float64
    box = 1000.0 * n
    r3 = lambda s: torch.randn(N, 3, dtype=f8, device=dev, generator=g) * s
    pos = torch.rand(N, 3, dtype=f8, device=dev, generator=g) * box
    pos.clamp_(min=1e-9)
    vel, gpm, gacc, hacc = r3(100.0), r3(1.0), r3(1.0), r3(1.0)
    typ = torch.cat([torch.zeros(N // 2, dtype=torch.uint8, device=dev), torch.ones(N // 2, dtype=torch.uint8, device=dev)])
    flags = torch.zeros(N, dtype=torch.uint8, device=dev)
    tb = torch.randint(0, 4, (N,), dtype=torch.uint8, device=dev, generator=g)
    hsml = torch.full((N,), box / n, dtype=f8, device=dev)
    dthsml = torch.zeros(N, dtype=f8, device=dev)
    ent = torch.ones(N, dtype=f8, device=dev)
    dte = torch.zeros(N, dtype=f8, device=dev)
    K = pkg.KickFactors()
    for b in range(4):
        K.gravkick[b], K.hydrokick[b], K.dt_entr[b], K.bin_active[b] = 1e-3, 1e-3, 1e-3, 1
    K.atime, K.MaxGasVel = 0.5, 3e5
    eng = pkg.Engine(dev.index or 0)
    eng.use_torch_stream()

    def step():
        eng.dev_apply_pm_half_kick(vel, gpm, 1e-3, flags=flags)
        eng.dev_apply_half_kick(vel, gacc, K, type=typ, flags=flags, tb_grav=tb, tb_hydro=tb, hydroaccel=hacc, entropy=ent, dtentropy=dte)
        eng.dev_drift_all_particles(pos, vel, 1e-3, box, (0.0, 0.0, 0.0), type=typ, flags=flags, hsml=hsml, dthsml=dthsml)

    for _ in range(args.warmup + 1):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # bytes per particle: PM kick 24+24+1 read, 24 written; half kick 24+24+1+1+2 read (+24+16 gas), 24 (+8) written;
    # drift 24+24+1+1 read (+16 gas), 24 (+8) written
    b_alg = N * (49 + 24 + 52 + 24 + 50 + 24) + (N // 2) * (40 + 8 + 16 + 8)
    ach = b_alg * args.steps / el / 1e9
    out = {"metric": "particle-updates/sec (PM half kick + half kick + drift)", "value": N * args.steps / el, "unit": "particles/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "time integration of 2x%d^3 particles (half gas) on device-resident arrays" % n, "particles": N},
           "roofline": {"bound": "hbm", "kernel": "k_pm_half_kick + k_half_kick + k_drift", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": b_alg,
                        "avg_launch_ms": 1e3 * el / args.steps,
                        "note": "three launches per step; each call also reads back an error word (one stream synchronisation per call)"}}
    eng.close()
    emit(out)
    return out


def fof_bench(pkg, torch, args, dev):
    """fof_fof (tree of the dark matter, primary linking, group table, P[].GrNr) on n^3 particles: a uniform
    background (60 %) plus Gaussian clumps of 20 .. 20000 members, linking length 0.2 mean separations, FOFHaloMinLength 32.
    One step = one fof_fof.  Reported against the HBM peak with the compulsory bytes of the passes (positions, IDs, labels, sort)."""
    n = args.n or 256
    N = n ** 3
    box = 1000.0 * n
    g = torch.Generator(device=dev).manual_seed(7)
    f8 = torch.float64
    nback = int(0.6 * N)
    parts = [torch.rand(nback, 3, dtype=f8, device=dev, generator=g) * box]
    left = N - nback
    LL = 0.2 * box / n
    cpu = torch.Generator().manual_seed(3)
    while left > 0:
        m = min(left, int(torch.exp(torch.empty(1).uniform_(math.log(20.), math.log(20000.), generator=cpu)).item()))
        c = torch.rand(3, dtype=f8, device=dev, generator=g) * box
        parts.append(torch.remainder(c + torch.randn(m, 3, dtype=f8, device=dev, generator=g) * (0.25 * LL * m ** (1. / 3)), box))
        left -= m
    pos = torch.cat(parts).contiguous()
    pos.clamp_(min=1e-9)
    mass = torch.ones(N, dtype=torch.float32, device=dev)
    ids = torch.randperm(N, device=dev, generator=g).to(torch.int64)
    vel = torch.randn(N, 3, dtype=f8, device=dev, generator=g)
    grnr = torch.zeros(N, dtype=torch.int64, device=dev)
    eng = pkg.Engine(dev.index or 0)
    eng.use_torch_stream()
    eng.dev_bind_particles(pos, mass, box)
    ng = 0
    for _ in range(args.warmup + 1):
        ng = eng.dev_fof_fof(ids, LL, 32, vel=vel, grnr=grnr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ng = eng.dev_fof_fof(ids, LL, 32, vel=vel, grnr=grnr)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    in_groups = int((grnr >= 0).sum().item())
    # compulsory bytes per particle: tree build (keys, sort, gather: ~200), link walk (32 source + 4 parent), flatten / labels (4+4+8+8+8),
    # label sort (8 passes x 24), accumulate (4+24+24+4+8), GrNr write 8
    b_alg = N * (200 + 36 + 32 + 192 + 64 + 8)
    ach = b_alg * args.steps / el / 1e9
    out = {"metric": "particles/sec through fof_fof (primary linking + group catalogue)", "value": N * args.steps / el, "unit": "particles/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "friends-of-friends groups of %d^3 dark-matter particles (60%% uniform + Gaussian clumps), LL = 0.2" % n,
                      "particles": N, "groups": ng, "particles_in_groups": in_groups},
           "roofline": {"bound": "hbm", "kernel": "fof_fof (tree build + k_fof_walk + sorts + k_fof_accumulate)", "achieved": ach,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": b_alg,
                        "avg_launch_ms": 1e3 * el / args.steps,
                        "note": "whole fof_fof call (about 30 launches, three host synchronisations for counts); the link walk is a "
                                "neighbour search bound by instruction issue and latency, not by HBM"}}
    eng.close()
    emit(out)
    return out


def domain_bench(pkg, torch, args, dev):
    """one Peano-Hilbert domain decomposition (domain_decompose_full up to the exchange, domain.c:153-225) of n^3
    particles for 8 tasks, as ONE rank sees it: key sample, top-tree arithmetic, the count pass, the balanced assignment and the pass
    that gives every particle its TopLeaf and destination task (mp-gadget_amd/domain_peano.py; the collectives of a real run are
    sums of a few integers and one small broadcast).  One step = one decomposition."""
    DP = importlib.import_module("mp-gadget_amd.domain_peano")
    n = args.n or 256
    ntask = 8
    ic = args.ic if args.ic != "s_grid" else "s_zel"      # (the default set of the force bench is a jittered lattice: use the displaced one)
    pos, mass, box = getattr(pkg.ics, ic)(n)
    N = len(pos)
    d_pos = torch.from_numpy(pos).to(dev)
    eng = pkg.Engine(dev.index or 0)
    eng.use_torch_stream()

    class EightTasks(DP.PeanoDomain):   # one process stands for rank 0 of 8: the sums over ranks see this rank's share only
        def _sum(self, *vals):
            return [int(v) for v in vals]

        def _recv_tree(self, src):      # the other seven hold no particles: empty trees
            import numpy as np
            return np.zeros(0, DP.TOPNODE_DTYPE), 0

        def _bcast_bytes(self, arr, src, n=None):
            return arr

    dom = EightTasks(eng, box, 0, 1, global_sorting=False)
    dom.world = ntask
    dom._cdev = torch.device("cpu")

    def one():
        policy = DP.DomainPolicy(0, ntask)
        tree, size = dom._global_toptree(d_pos, None, policy, max(int(0.5 * (N + 1)), 1))
        import ctypes as C
        import numpy as np
        ltn = np.zeros(size, np.int32)
        nl = C.c_int(0)
        P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
        assert eng.lib.mpg_domain_create_topleaves(P(tree, DP.TopNode), size, P(ltn, C.c_int), C.byref(nl)) == 0
        counts = dom._topleaves(d_pos, None, tree, size, nl.value, None)[0]
        lt, st, en = np.zeros(nl.value, np.int32), np.zeros(ntask, np.int32), np.zeros(ntask, np.int32)
        assert eng.lib.mpg_domain_assign_topleaves_balanced(P(tree, DP.TopNode), size, P(ltn, C.c_int), nl.value, P(counts, C.c_int64), ntask, 1,
                                                            P(lt, C.c_int), P(st, C.c_int), P(en, C.c_int)) == 0
        fc, tc, topleaf, task = dom._topleaves(d_pos, None, tree, size, nl.value, lt)
        return size, nl.value, tc

    for _ in range(args.warmup + 1):
        size, nleaves, tc = one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        size, nleaves, tc = one()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    b_alg = N * (32 + 24 + 24 + 8)      # sample pass: positions + keys (32); count pass: positions (24); layout pass: positions + TopLeaf + Task (32)
    ach = b_alg * args.steps / el / 1e9
    out = {"metric": "particles/sec through one Peano-Hilbert domain decomposition (8 tasks)", "value": N * args.steps / el, "unit": "particles/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
           "config": {"workload": "domain decomposition of %d^3 particles (%s) for %d tasks: TopNodes %d, TopLeaves %d" % (n, ic, ntask, size, nleaves),
                      "particles": N, "max_load_over_mean": float(tc.max() / tc.mean())},
           "roofline": {"bound": "hbm", "kernel": "k_peano_keys + k_topleaf (three passes over the positions)", "achieved": ach, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": b_alg,
                        "avg_launch_ms": 1e3 * el / args.steps,
                        "note": "whole decomposition incl. the host tree arithmetic on the 1/256 sample (about half of the time); the key of a "
                                "particle is 21 dependent table steps, which bounds the passes rather than HBM"}}
    eng.close()
    emit(out)
    return out


def hydro_ics(pkg, n):
    return pkg.ics.hydro_pair(n)


def parity_check_hydro_ranks(pkg, torch, dist, args, dev, rank, world, host, box, n, PE, own_ids, o_pos, o_mass, o_typ, a, t, df, eng, nsample=2048):
    """The multi-rank SPH step checks itself (BASELINE configs[4]: "per-step force tolerance check"; the reference's own gate is
    check_densities, test_density.c:87-142).  One more density() + hydro_force() on the ranks from the smoothing lengths they hold;
    rank 0 then repeats the two loops on ONE GPU over the WHOLE particle set from the same starting smoothing lengths, through the
    single-GPU path the parity tests pin to the oracle, and compares nsample gas targets spread over the ranks: Hsml (1e-12), Density and
    HydroAccel / DtEntropy (1e-10, for >= 99.9 % of the sample: a target whose neighbour number sits within rounding of the edge of the
    accepted window may take one pass more or fewer), and the loop counters summed over the ranks (target visits,
    neighbour interactions).  Untimed; collective."""
    f8 = dict(dtype=torch.float64, device=dev)
    n_own = int(own_ids.shape[0])
    hs_in = a["hsml"].clone()
    df.force_tree_build(o_pos, o_mass)
    df.density(o_typ, a, t, DoEgyDensity=PE)
    sd = eng.sph_stats()
    df.hydro_force(n_own, a, t)
    sh = eng.sph_stats()
    c_n = torch.tensor([sd["iterations"], sd["targets"], sd["interactions"], sh["targets"], sh["interactions"]], dtype=torch.int64, device=dev)
    it_max = c_n[:1].clone()
    dist.all_reduce(it_max, op=dist.ReduceOp.MAX)
    dist.all_reduce(c_n)
    gas = torch.nonzero(o_typ == 0).squeeze(1)
    k = max(nsample // world, 1)
    sel = gas[torch.linspace(0, max(int(gas.shape[0]) - 1, 0), steps=min(k, int(gas.shape[0])), device=dev).long().unique()] if gas.shape[0] else gas
    ns = int(sel.shape[0])
    rows = torch.cat([own_ids[sel].double()[:, None], a["hsml"][sel][:, None], a["density"][sel][:, None], a["hydroacc_out"][sel],
                      a["dtentropy_out"][sel][:, None]], dim=1) if ns else torch.zeros(0, 7, **f8)
    pad = torch.zeros(k, 7, **f8)
    pad[:ns] = rows
    nmax = torch.tensor([n_own], dtype=torch.int64, device=dev)
    dist.all_reduce(nmax, op=dist.ReduceOp.MAX)
    hin = torch.zeros(int(nmax.item()), 2, **f8)                     # (id, starting Hsml) of every own particle
    hin[:n_own, 0] = own_ids.double()
    hin[:n_own, 1] = hs_in
    cntv = torch.tensor([ns, n_own], dtype=torch.int64, device=dev)
    allc = [torch.zeros_like(cntv) for _ in range(world)]
    allr = [torch.zeros_like(pad) for _ in range(world)]
    allh = [torch.zeros_like(hin) for _ in range(world)]
    dist.all_gather(allc, cntv)
    dist.all_gather(allr, pad)
    dist.all_gather(allh, hin)
    res = None
    if rank == 0:
        pos, mass, typ = host
        N = len(pos)
        got = torch.cat([allr[r][:int(allc[r][0].item())] for r in range(world)])
        ids = got[:, 0].long()
        h0 = torch.zeros(N, **f8)
        for r in range(world):
            m = int(allc[r][1].item())
            h0[allh[r][:m, 0].long()] = allh[r][:m, 1]
        del allh
        e1 = pkg.Engine(dev.index or 0)
        e1.use_torch_stream()
        e1.gravshort_fill_ntab(0, 1.5)
        e1.gravpm_init_periodic(box, 1.5, 2 * n, G)
        e1.gravshort_set_softenings(box / n)
        e1.set_densitypar(1.0, 2.0, 2.0, 99999., pkg.engine.DENSITY_KERNEL_QUINTIC_SPLINE, 0.006)
        e1.set_hydropar(PE, 100.0, 0.75)
        p1, m1, t1 = torch.from_numpy(pos).to(dev), torch.from_numpy(mass).to(dev), torch.from_numpy(typ).to(dev)
        e1.dev_bind_particles(p1, m1, box, type=t1)
        z1, z3 = (lambda: torch.zeros(N, **f8)), (lambda: torch.zeros(N, 3, **f8))
        a1 = dict(hsml=h0, dthsml=z1(), vel=z3(), entropy=torch.ones(N, **f8), density=z1(), egywtdensity=z1(), dhsmlegyfac=z1(), divvel=z1(),
                  curlvel=z1(), hydroacc_out=z3(), dtentropy_out=z1(), maxsignalvel=z1())
        e1.dev_force_tree_rebuild_mask(pkg.engine.GASMASK)
        e1.dev_density(a1, t, DoEgyDensity=PE)
        s1d = e1.sph_stats()
        e1.dev_force_tree_calc_hmax()
        e1.dev_hydro_force(a1, t)
        s1h = e1.sph_stats()
        torch.cuda.synchronize()
        rel = lambda x, y, sc: ((x - y).abs() / sc) if x.dim() == 1 else ((x - y).norm(dim=1) / sc)
        d_h = rel(a1["hsml"][ids], got[:, 1], a1["hsml"][ids])
        d_rho = rel(a1["density"][ids], got[:, 2], a1["density"][ids])
        d_acc = rel(a1["hydroacc_out"][ids], got[:, 3:6], a1["hydroacc_out"][ids].norm(dim=1).mean().clamp_min(1e-300))
        d_dte = rel(a1["dtentropy_out"][ids], got[:, 6], a1["dtentropy_out"][ids].abs().mean().clamp_min(1e-300))
        frac = lambda d, tol: float((d <= tol).double().mean())
        cn = [int(x) for x in c_n.cpu()]
        c1 = [s1d["targets"], s1d["interactions"], s1h["targets"], s1h["interactions"]]
        close = lambda x, y: abs(x - y) <= 1e-5 * max(abs(y), 1)
        res = {"n": int(ids.shape[0]), "hsml_frac_within_1e-12": frac(d_h, 1e-12), "hsml_max_rel": float(d_h.max()),
               "density_frac_within_1e-10": frac(d_rho, 1e-10), "hydroaccel_frac_within_1e-10": frac(d_acc, 1e-10),
               "dtentropy_frac_within_1e-10": frac(d_dte, 1e-10), "density_passes_ranks_max": int(it_max.item()), "density_passes_one_gpu": s1d["iterations"],
               "counters_ranks": cn[1:], "counters_one_gpu": c1, "counters_equal": cn[1:] == c1,
               "counters": "[density target visits, density neighbours, hydro targets, hydro pairs], summed over the ranks",
               "gate": "Hsml within 1e-12, Density / HydroAccel / DtEntropy within 1e-10 for >= 99.9 % of the sample, Hsml of every sampled "
                       "target within the reference's bound (1e-3, test_density.c:203), counters within 1e-5 of the one-GPU loops'",
               "reference": "the same gas targets recomputed on one GPU from all %d particles and the same starting smoothing lengths through the "
                            "single-GPU loops (pinned to the oracle by tests/test_gpu_sph.py)" % N}
        res["ok"] = bool(res["hsml_frac_within_1e-12"] >= 0.999 and res["density_frac_within_1e-10"] >= 0.999 and
                         res["hydroaccel_frac_within_1e-10"] >= 0.999 and res["dtentropy_frac_within_1e-10"] >= 0.999 and res["hsml_max_rel"] <= 1e-3 and
                         all(close(x, y) for x, y in zip(cn[1:], c1)))
        e1.close()
    return res


def hydro_bench_peano(pkg, torch, dist, args, dev, rank, world):
    """configs[2] / [4] weak-scaled over GPUs on the reference's decomposition, everything through the library's choreography
    (mpg_dist_*, csrc/dist.hip): domain_decompose_full + exchange (untimed), then per step gravity (PM by particle shipping, ghost
    import, global top, walk) and the SPH loops (ghost columns along the ghost plan, density, the ghosts' fields from their owners,
    hmax, hydro force)."""
    n = args.n or {2: 160, 4: 200, 8: 256}.get(world, int(round(128 * world ** (1. / 3) / (4 * world))) * 4 * world)
    nmesh = 2 * n
    PE = 0 if args.sph == "de" else 1                        # configs[4]: pressure-entropy SPH on several GPUs
    pos, mass, typ, box = hydro_ics(pkg, n)
    N = len(pos)
    f8 = dict(dtype=torch.float64, device=dev)
    eng = pkg.Engine(dev.index or 0)
    eng.use_torch_stream()
    eng.set_walk_variant(args.variant)
    eng.gravshort_fill_ntab(0, 1.5)
    eng.gravpm_init_periodic(box, 1.5, nmesh, G)
    eng.set_gravshort_treepar(TreeUseBH=2)
    eng.gravshort_set_softenings(box / n)
    eng.set_densitypar(1.0, 2.0, 2.0, 99999., pkg.engine.DENSITY_KERNEL_QUINTIC_SPLINE, 0.006)
    eng.set_hydropar(PE, 100.0, 0.75)
    comm, comm_note = make_comm(pkg, torch, dist, args, dev, eng)
    df = pkg.dist.DistForce(eng, comm)
    share = slice((N * rank) // world, (N * (rank + 1)) // world)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a[share])).to(dev)
    s_pos = T(pos)
    df.domain_decompose(s_pos, box, overdecomposition=args.overdecomp)
    o_pos, o_mass, o_typ, own_ids = df.domain_exchange(s_pos, T(mass), T(typ), torch.arange(N, dtype=torch.int64, device=dev)[share].contiguous())
    n_own = int(o_pos.shape[0])
    host = (pos, mass, typ) if (rank == 0 and not args.no_parity_check) else None
    del pos
    df.use_decomposition(box, max(6.0 * 1.5 * box / nmesh, 6.0 * box / n))       # margin: Rcut and the largest smoothing length
    z1, z3 = (lambda: torch.zeros(n_own, **f8)), (lambda: torch.zeros(n_own, 3, **f8))
    a = dict(hsml=torch.full((n_own,), 2.0 * box / n, **f8), dthsml=z1(), vel=z3(), entropy=torch.ones(n_own, **f8), density=z1(), egywtdensity=z1(),
             dhsmlegyfac=z1(), divvel=z1(), curlvel=z1(), hydroacc_out=z3(), dtentropy_out=z1(), maxsignalvel=z1())
    acc, prev, gravpm, pot = z3(), z3(), z3(), z1()
    t = pkg.SphTimes()
    t.atime, t.hubble = 0.1, 0.1
    for i in range(47):
        t.dloga_bin[i] = 0.01
    state = dict(steps=0)

    def step():
        nonlocal acc, prev
        prev, acc = acc, prev
        df.gravity_step(o_pos, o_mass, acc, gravpm, potential=pot, prev_accel=prev if state["steps"] else None)
        df.density(o_typ, a, t, DoEgyDensity=PE)
        df.hydro_force(n_own, a, t)
        state["steps"] += 1

    def sync():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup + 1):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = torch.tensor([time.perf_counter() - t0], **f8)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    el = float(dt.item())
    parity = None
    if not args.no_parity_check:
        parity = parity_check_hydro_ranks(pkg, torch, dist, args, dev, rank, world, host, box, n, PE, own_ids, o_pos, o_mass, o_typ, a, t, df, eng)
    out = None
    if rank == 0:
        st = df.stats()
        out = {"metric": "particle-updates/sec (gravity + SPH force step)", "value": N * args.steps / el, "unit": "particles/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "2x%d^3 DM+gas TreePM + %s SPH force step, Nmesh=%d, s_zel ICs, quintic kernel" % (n, "pressure-entropy" if PE else "density-entropy", nmesh),
                          "particles": N, "parallelism": "%d GPUs: particles on the owners of their Peano-Hilbert TopLeaves, choreography in the library "
                                                         "(mpg_dist_*), collectives on RCCL" % world,
                          "communicator": comm_note, "rccl_ranks": comm.info()["nranks"] if hasattr(comm, "info") else None,
                          "ghost_fraction_rank0": round(st["ghosts"] / max(n_own, 1), 3), "density_iterations_last": eng.sph_stats()["iterations"]}}
    dist.barrier()
    dist.destroy_process_group()
    df.close()
    if hasattr(comm, "close"):
        comm.close()
    eng.close()
    if out is not None:
        if parity is not None:
            out["parity_check"] = parity
        emit(out)
        if parity is not None and not parity["ok"]:
            print("bench.py: the multi-rank SPH loops FAILED the self-check against the one-GPU path: %s" % json.dumps(parity), file=sys.stderr, flush=True)
            sys.exit(3)
    return out


def host_cpu_info():
    """(physical cores, logical cpus, model name) of this box from /proc/cpuinfo."""
    cores, model, logical = set(), "unknown", 0
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                logical += 1
            elif k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
                cores.add((phys, core))
    except OSError:
        pass
    return (len(cores) or os.cpu_count() or 1), (logical or os.cpu_count() or 1), model


def cgroup_cpu_limit():
    """CPUs this container may use at once (cgroup v2 cpu.max / v1 cfs quota), or None.  A GPU slot of a shared box comes with a
    share of its host cores: 128 threads under a 16-CPU quota run at an eighth of their speed each (measured in round 3: 8.1e7 pair
    interactions/s/thread with up to 16 threads, 8.2e6 with 128 - round 2's "12 % per-thread efficiency")."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(int(q) / int(p)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, int(q / p))
    except (OSError, ValueError):
        pass
    return None


def _numa_core_groups(nproc_hint=0):
    """Physical cores of this box grouped for one worker process each: (first logical cpu of every physical core), sorted by socket /
    NUMA node, cut into groups of ~16 cores that never straddle a socket (the reference runs as ranks x threads, README.rst:121)."""
    cpus = {}
    try:
        proc = phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                proc = int(v)
            elif k == "physical id":
                phys = int(v)
            elif k == "core id":
                core = int(v)
                cpus.setdefault((phys, core), proc)
    except OSError:
        pass
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    cores = sorted((k, c) for k, c in cpus.items() if c in allowed)
    if not cores:
        cores = [((0, i), c) for i, c in enumerate(sorted(allowed))]
    lim = cgroup_cpu_limit()
    if os.environ.get("MPG_CPU_THREADS"):
        lim = int(os.environ["MPG_CPU_THREADS"])
    if lim and lim < len(cores):
        cores = cores[:lim]           # (the quota does not say WHICH cpus: the first ones of one socket)
    by_socket = {}
    for (ph, _), c in cores:
        by_socket.setdefault(ph, []).append(c)
    groups = []
    for ph in sorted(by_socket):
        cs = by_socket[ph]
        k = max(1, round(len(cs) / 16)) if not nproc_hint else max(1, nproc_hint // len(by_socket))
        for i in range(k):
            g = cs[len(cs) * i // k:len(cs) * (i + 1) // k]
            if g:
                groups.append(g)
    return groups


def _cpu_worker(rank, nproc, cpus, path, box, n, nmesh, lo, hi, q):
    """One "rank" of the CPU baseline: pinned to its cores, it loads the particle set (first touch by its own threads: the pages land on
    its NUMA node), builds the oracle's tree, and walks its contiguous share [lo, hi) of the Morton-ordered sample.  Also timed: the tree
    of its own 1/nproc of the particles, which is what a rank of the reference builds (forcetree.c: local particles + top tree)."""
    try:
        os.sched_setaffinity(0, set(cpus))
    except (AttributeError, OSError):
        pass
    os.environ["OMP_NUM_THREADS"] = str(len(cpus))
    os.environ["OMP_PROC_BIND"] = "close"
    os.environ["OMP_PLACES"] = "cores"
    os.environ["OMP_WAIT_POLICY"] = "active"
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    orc = O.Oracle(fast=True)
    orc.fill_ntab(0, 1.5)
    d = np.load(path, mmap_mode="r")
    pos = np.array(d["pos"])            # copies: first touch in this process
    mass = np.array(d["mass"])
    old = np.array(d["old"])
    order = np.array(d["order"])
    N = len(pos)
    # the tree a rank of the reference builds: its own share of the particles (a contiguous Morton range)
    own = np.sort(order[N * rank // nproc:N * (rank + 1) // nproc])
    t0 = time.perf_counter()
    tr_own = orc.tree(np.ascontiguousarray(pos[own]), np.ascontiguousarray(mass[own]), box, father=False)
    t_tree_own = time.perf_counter() - t0
    del tr_own
    t0 = time.perf_counter()
    tr = orc.tree(pos, mass, box, father=False)
    t_tree_full = time.perf_counter() - t0
    par = O.make_grav_params(box, nmesh, npart_cbrt=n, G=G)
    par.TreeUseBH = 0
    act = np.ascontiguousarray(order[lo:hi])
    tr.grav_short_tree(par, oldacc=old, active=act[:min(len(act), 16384)])   # warm-up
    q.put(("ready", rank))
    walks, pp = [], 0
    for _ in range(3):
        t0 = time.perf_counter()
        _, _, c, _ = tr.grav_short_tree(par, oldacc=old, active=act)
        walks.append(time.perf_counter() - t0)
        pp = int(c[0])
    q.put(("done", rank, walks, pp, t_tree_own, t_tree_full, orc.num_threads()))


def _cpu_pm_worker(pos, mass, box, nmesh, threads, q):
    """The long-range step of the CPU baseline in a process of its own (the -ffast-math build of the oracle switches the thread that loads it to
    flush-to-zero arithmetic: kept out of the bench process).  Two runs, the second is reported (the first touches the mesh pages)."""
    try:
        os.environ["OMP_NUM_THREADS"] = str(threads)
        sys.path.insert(0, ROOT)
        from oracle import oracle as O
        orc = O.Oracle(fast=True)
        parts = {}
        for _ in range(2):
            parts = {}
            O.gravpm_force_c(orc, pos, mass, box, nmesh, 1.5, G, want_potential=True, workers=threads, timings=parts)
        q.put(("pm", parts))
    except Exception as e:      # noqa: BLE001 - reported by the parent
        q.put(("pm_failed", repr(e)))


def cpu_baseline(pkg, pos, mass, box, n, nmesh, aold_vec, sample):
    """The CPU "port" (SURVEY 8(d) "CPU baseline timing"): the oracle built with the reference's flags (-O3 -ffast-math -fopenmp) on the
    host cores of this box, as the reference runs on such a box: P processes x T threads (README.rst:121), each process pinned to a
    group of cores of one socket with its data first-touched there.  Bounded sample: `sample` targets CONTIGUOUS IN MORTON ORDER
    (the reference walks its particles in Peano-Hilbert order), cut into P contiguous shares; every process holds the tree of all
    particles (so that the walk of its share is the reference's walk) and also times the tree of its own 1/P of the particles, the
    tree a reference rank builds.  value = N / (tree of an own share + median walk time scaled from the sample to N + one whole
    gravpm_force on the same cores: oracle/pm_oracle.c's loops + pocketfft).  Round 2 ran ONE process over all cores with the arrays
    first-touched by one thread: 12 % of the per-thread rate of the 8-thread calibration."""
    import multiprocessing as mp
    import tempfile
    phys, logical, model = host_cpu_info()
    groups = _numa_core_groups()
    P = len(groups)
    N = len(pos)
    old = np.sqrt((aold_vec ** 2).sum(1)) / G
    sample = min(sample, N)
    q10 = np.minimum((pos / box * 1024).astype(np.int64), 1023)     # Morton order (10 bits per axis make consecutive targets neighbours)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249
    morton = (spread(q10[:, 0]) << 2) | (spread(q10[:, 1]) << 1) | spread(q10[:, 2])
    order = np.argsort(morton, kind="stable").astype(np.int32)
    del q10, morton
    start = (N - sample) // 2
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    path = os.path.join(shm, "mpg_cpu_baseline_%d.npz" % os.getpid())
    np.savez(path, pos=pos, mass=mass, old=old, order=order)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = []
    for r, cpus in enumerate(groups):
        lo, hi = start + sample * r // P, start + sample * (r + 1) // P
        procs.append(ctx.Process(target=_cpu_worker, args=(r, P, cpus, path, box, n, nmesh, lo, hi, q)))
    t_all0 = time.perf_counter()
    for p in procs:
        p.start()
    res, ready = {}, 0
    try:
        import queue as _queue
        deadline = time.perf_counter() + 1800
        while len(res) < P:
            try:
                m = q.get(timeout=5)
            except _queue.Empty:
                if any(p.exitcode not in (None, 0) for p in procs) or time.perf_counter() > deadline:
                    raise RuntimeError("cpu_baseline: a worker process died (exit codes %s)" % [p.exitcode for p in procs])
                continue
            if m[0] == "done":
                res[m[1]] = m[2:]
    finally:
        for p in procs:
            p.join(timeout=60)
        try:
            os.remove(path)
        except OSError:
            pass
    t_total = time.perf_counter() - t_all0
    walks = np.array([res[r][0] for r in range(P)])          # [P, 3] seconds
    t_walk = float(np.median(walks.max(0)))                  # per repetition the slowest process; median over the three
    pp = sum(res[r][1] for r in range(P))
    threads = sum(res[r][4] for r in range(P))
    t_tree_own = max(res[r][2] for r in range(P))
    t_tree_full = max(res[r][3] for r in range(P))
    # the long-range part on the same cores: one whole gravpm_force (gravpm.c:61-119) on the Nmesh^3 mesh - CIC deposit, potential
    # transfer, the three force transfers and the four read-outs as OpenMP loops (oracle/pm_oracle.c, the reference's flags), the five
    # 3-D transforms (1 r2c + 4 c2r, petapm.c:319-357) by pocketfft (C++, scipy.fft) with one worker per core the container may use.
    # The reference runs the transforms through PFFT on MPI ranks with pencil exchanges either side; those exchanges are not part of
    # a one-process run, which favours the CPU.  In a process of its own (_cpu_pm_worker).
    t_pm, pm_parts, t_pm_fft = None, None, None
    try:
        qpm = ctx.Queue()
        ppm = ctx.Process(target=_cpu_pm_worker, args=(pos, mass, box, nmesh, threads, qpm))
        ppm.start()
        m = qpm.get(timeout=900)
        ppm.join(timeout=60)
        if m[0] != "pm":
            raise RuntimeError(m[1])
        pm_parts = m[1]
        t_pm = float(sum(pm_parts.values()))
        t_pm_fft = pm_parts["fft"]
    except Exception as e:      # noqa: BLE001 - no scipy / a failed worker: the leg is reported as absent, the value then excludes the PM as in rounds 1-3
        sys.stderr.write("cpu_baseline: PM leg failed (%s)\n" % e)
        t_pm = None
    t_full = t_tree_own + t_walk * N / sample + (t_pm or 0.0)
    out = {"value": N / t_full, "pm_s": None if t_pm is None else round(t_pm, 3), "pm_fft_s": None if t_pm_fft is None else round(t_pm_fft, 3),
           "pm_parts_s": None if pm_parts is None else {k: round(v, 3) for k, v in pm_parts.items()},
           "pm_note": "value includes one whole gravpm_force on the CPU side: CIC deposit, transfer functions and the four read-outs as OpenMP "
                      "loops (oracle/pm_oracle.c, %d threads) + the five 3-D FFTs by pocketfft (%d workers)" % (threads, threads),
           "unit": "particles/s", "cores": threads, "kind": "port", "processes": P, "cgroup_cpu_limit": cgroup_cpu_limit(),
           "threads_per_process": [len(g) for g in groups], "cpu_model": model, "physical_cores": phys, "logical_cpus": logical,
           "omp": "per process: sched_setaffinity to its cores, OMP_PROC_BIND=close OMP_PLACES=cores", "walk_s_median_of_3": round(t_walk, 3),
           "walk_s_all": [round(float(w), 3) for w in walks.max(0)], "tree_build_own_share_s": round(t_tree_own, 3),
           "tree_build_all_particles_s": round(t_tree_full, 3), "wall_s": round(t_total, 1),
           "pairs_per_s_per_thread": pp / t_walk / threads if pp else None,
           "calibration_pairs_per_s_per_thread": 5.1e7,
           # what the whole box would give if the walk scaled linearly from the cores this container may use to all physical cores (an
           # upper bound for the CPU: shared L3 and memory bandwidth only make it less)
           "value_if_all_physical_cores": (N / t_full) * phys / max(threads, 1),
           "sample": "oracle (gcc -O3 -ffast-math -fopenmp), %d processes x %s threads pinned per socket: each builds the tree of all %d particles "
                     "(%.2f s, set-up) and walks its share of %d Morton-ordered targets (median of 3 of the slowest process: %.2f s), scaled to N, "
                     "plus the tree of an own 1/%d share (%.2f s), plus one whole PM step (pm_s)" % (P, "/".join(str(len(g)) for g in groups[:2]) + ("/..." if P > 2 else ""),
                                                                                      N, t_tree_full, sample, t_walk, P, t_tree_own)}
    return out


if __name__ == "__main__":
    main()
