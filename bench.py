#!/usr/bin/env python3
"""bench.py -- particle-updates/s of one TreePM gravity force step on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over the synchronised particle set, all particles active (a PM step of
run.c:522-548): gravpm_force (CIC deposit, 5 FFTs, transfers, readout) + force_tree_full (tree build + moments)
+ grav_short_tree (short-range walk with the relative opening criterion, OldAcc from the previous step).
Inputs are resident in HBM when the timed region starts.

N = 1 : 256^3 dark-matter particles, Nmesh = 512 (BASELINE.json configs[1]), S-grid synthetic ICs.
N > 1 : weak scaling, ~256^3 particles per GPU (n = 320 / 400 / 512 per dimension for N = 2 / 4 / 8, Nmesh = 2n).
        Every rank holds the particle set; targets are sharded over ranks as contiguous tree-order (Morton) ranges
        and the accelerations are exchanged with one RCCL all-gather per step.

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
    kernels = {1: "k_grav_walk", 4: "k_grav_walk_coop", 5: "k_grav_walk_shared", 6: {"0": "k_walk_lists", "1": "k_walk_lists2"}.get(os.environ.get("MPG_LISTS_MODE", "2"), "k_walk_lists8") + " + k_walk_eval",
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
    host memory, results written back into them - PCIe transfers and AoS packing included.  Not `value`."""
    P = pkg.make_particles(pos, mass)
    N = len(pos)
    ts = []
    for it in range(steps + 2):
        t0 = time.perf_counter()
        eng.gravpm_force(P)
        t1 = time.perf_counter()
        eng.force_tree_full(P, box)
        t2 = time.perf_counter()
        eng.grav_short_tree(P)
        t3 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2))
    ts = np.array(ts[2:])      # the first two steps allocate the pinned staging and run the Barnes-Hut walk
    tot = ts.sum(1).mean()
    return {"ms_per_step": round(1e3 * tot, 2), "particles_per_s": N / tot,
            "calls_ms": {"gravpm_force": round(1e3 * ts[:, 0].mean(), 2), "force_tree_full": round(1e3 * ts[:, 1].mean(), 2),
                         "grav_short_tree": round(1e3 * ts[:, 2].mean(), 2)},
            "note": "mpg_gravpm_force + mpg_force_tree_full + mpg_grav_short_tree on %d 160-byte particle_data records in pageable host "
                    "memory (H2D of Pos/Mass, D2H of GravPM/FullTreeGravAccel/Potential, packing on host threads); the device-resident "
                    "rate is `value`" % N}


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
                    help="N = 1: skip the other_inputs (s_grid, s_clust) and host_path (PCIe-inclusive drop-in calls) legs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1 << 22, help="targets walked by the CPU baseline")
    ap.add_argument("--thresh", type=int, default=16)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--mgpu", choices=["peano", "domain", "slab", "replicated"], default="peano",
                    help="N > 1: peano = particles on the owners of their Peano-Hilbert TopLeaves (the reference's domain_decompose_full), "
                         "force step through the library's own choreography (mpg_dist_*, csrc/dist.hip); domain = x-slab domains with "
                         "ghost import driven from Python (round 1); slab = particles replicated, slab PM and slab targets; replicated = "
                         "everything but the walk targets replicated")
    ap.add_argument("--overdecomp", type=int, default=8, help="peano: DomainOverDecompositionFactor (TopLeaves per rank and policy)")
    ap.add_argument("--no-rebalance", action="store_true",
                    help="peano: keep the decomposition by particle number (default: after two set-up steps the TopLeaves are dealt out "
                         "again by the measured work per particle, domain.c:611)")
    ap.add_argument("--sph", default="auto", choices=["auto", "de", "pe"],
                    help="hydro workload: density-entropy (BASELINE configs[2]) or pressure-entropy SPH (configs[4]); auto: de on one GPU, pe on several")
    ap.add_argument("--workload", default="gravity", choices=["gravity", "hydro", "integrate", "fof", "domain"],
                    help="gravity: BASELINE.json configs[1] (default, the headline metric); hydro: configs[2], 2 x n^3 DM+gas, "
                         "adds gas tree + density + hmax + hydro force (single GPU, diagnostic line)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))

    pkg = importlib.import_module("mp-gadget_amd")
    import torch
    import torch.distributed as dist

    if os.environ.get("MPG_DIST_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # MPG_FORCE_MGPU=1: run the multi-GPU code path (collectives included) in a one-rank group - a single-GPU box can then
    # exercise exactly what the ranks of an N-GPU run execute
    multi = world > 1 or bool(os.environ.get("MPG_FORCE_MGPU"))
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("MPG_DIST_BACKEND", "nccl")   # "gloo" lets two ranks share one GPU in tests
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.workload == "integrate":
        if world != 1:
            raise SystemExit("--workload integrate is single-GPU")
        return integrate_bench(pkg, torch, args, dev)
    if args.workload == "fof":
        if world != 1:
            raise SystemExit("--workload fof is single-GPU")
        return fof_bench(pkg, torch, args, dev)
    if args.workload == "domain":
        if world != 1:
            raise SystemExit("--workload domain is a one-rank line (tests/test_gpu_domain.py runs the decomposition on several ranks)")
        return domain_bench(pkg, torch, args, dev)
    if multi and world == 1:
        pkg.pm_slab.FORCE_COLLECTIVES = True
    if args.workload == "hydro":
        if multi and args.mgpu == "peano":
            return hydro_bench_peano(pkg, torch, dist, args, dev, rank, world)
        if multi:
            return hydro_bench_domain(pkg, torch, dist, args, dev, rank, world)
        return hydro_bench(pkg, torch, args, dev)
    # weak scaling: about 256^3 particles per GPU; Nmesh = 2 n must be a multiple of the number of GPUs (x-slab PM)
    n = args.n or {1: 256, 2: 320, 4: 400, 8: 512}.get(world, int(round(256 * world ** (1. / 3) / (8 * world))) * 8 * world)
    nmesh = 2 * n
    gen = getattr(pkg.ics, args.ic)
    slabwise = multi and args.mgpu == "domain" and args.ic == "s_grid" and world > 1 and n % world == 0
    if slabwise:
        # every rank generates only the grid planes its slab can own: its n/world planes and one more on either side (the
        # +-0.15 spacing jitter moves no particle further); the union over ranks is exactly s_grid(n)
        lo, hi = rank * (n // world), (rank + 1) * (n // world)
        parts = [pkg.ics.s_grid_planes(n, max(lo - 1, 0), min(hi + 1, n))]
        if lo == 0:
            parts.append(pkg.ics.s_grid_planes(n, n - 1, n))
        if hi == n:
            parts.append(pkg.ics.s_grid_planes(n, 0, 1))
        pos = np.concatenate([p[0] for p in parts])
        mass = np.concatenate([p[1] for p in parts])
        box = parts[0][2]
        N = n ** 3
    else:
        pos, mass, box = gen(n)
        N = len(pos)
    d_pos = torch.from_numpy(pos).to(dev)
    d_mass = torch.from_numpy(mass).to(dev)
    del pos
    eng = pkg.Engine(local_rank)
    eng.use_torch_stream()
    eng.set_walk_threshold(args.thresh)
    eng.set_walk_variant(args.variant)
    eng.gravshort_fill_ntab(0, 1.5)
    eng.gravpm_init_periodic(box, 1.5, nmesh, G)
    eng.set_gravshort_treepar(ErrTolForceAcc=0.002, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=2, Rcut=6.0,
                              FractionalGravitySoftening=1. / 30.)
    eng.gravshort_set_softenings(box / n)
    eng.dev_bind_particles(d_pos, d_mass, box)

    NB = int(d_pos.shape[0])    # particles bound at set-up (all of them, except in the slab-wise domain mode)
    gravpm = torch.zeros(NB, 3, dtype=torch.float64, device=dev)
    acc = torch.zeros(NB, 3, dtype=torch.float64, device=dev)
    prev = torch.zeros(NB, 3, dtype=torch.float64, device=dev)
    pot = torch.zeros(NB, dtype=torch.float64, device=dev)
    # N > 1: "slab" = x-slab PM (two all-to-all transposes per step) with the particles of the slab as
    # PM-readout and walk targets; "replicated" = every rank does the whole PM, targets are contiguous tree-slot ranges
    pm_ms = [0.0, 0]
    if multi and args.mgpu == "domain":
        rcut = 6.0 * 1.5 * box / nmesh                       # Rcut * Asmth * cell size (gravshort-tree.c:102)
        dom = pkg.domain.SlabDomain(eng, box, nmesh, rank, world, dev, rcut)
        own = dom.select_own(d_pos)
        own_pos, own_mass = d_pos[own].contiguous(), d_mass[own].contiguous()
        n_own = int(own.shape[0])
        tot = torch.tensor([n_own], dtype=torch.int64, device=dev)
        dist.all_reduce(tot)
        assert int(tot.item()) == N, "the ranks' own sets do not add up to the particle set (%d of %d)" % (int(tot.item()), N)
        del d_pos, d_mass, own, gravpm, acc, prev, pot      # from here on this rank holds its own particles and their ghosts only
        torch.cuda.empty_cache()
        spm = pkg.pm_slab.SlabPM(eng, box, nmesh, rank, world, dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        loc = {}                                             # arrays over [own | ghosts], sized on first use
    elif multi and args.mgpu == "peano":
        # domain_decompose_full + domain_exchange (untimed): every rank starts from a contiguous share of the set
        share = slice((N * rank) // world, (N * (rank + 1)) // world)
        pdom = pkg.domain_peano.PeanoDomain(eng, box, rank, world, overdecomposition=args.overdecomp)
        sp, sm = d_pos[share].contiguous(), d_mass[share].contiguous()
        pdom.decompose(sp)
        own_pos, own_mass = pdom.exchange(sp, sm)
        n_own = int(own_pos.shape[0])
        tot = torch.tensor([n_own], dtype=torch.int64, device=dev)
        dist.all_reduce(tot)
        assert int(tot.item()) == N, "the ranks' own sets do not add up to the particle set (%d of %d)" % (int(tot.item()), N)
        del d_pos, d_mass, sp, sm, gravpm, acc, prev, pot
        torch.cuda.empty_cache()
        comm = pkg.dist.TorchComm(dev)
        dforce = pkg.dist.DistForce(eng, comm)
        dforce.set_domain(pdom, 6.0 * 1.5 * box / nmesh)      # margin = Rcut * Asmth * cell size (gravshort-tree.c:102)
        z3 = lambda: torch.zeros(n_own, 3, dtype=torch.float64, device=dev)
        loc = dict(acc=z3(), prev=z3(), gravpm=z3(), pot=torch.zeros(n_own, dtype=torch.float64, device=dev), steps=0)
    elif multi and args.mgpu == "slab":
        spm = pkg.pm_slab.SlabPM(eng, box, nmesh, rank, world, dev)
        tex = pkg.pm_slab.TargetExchange(world, dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    elif multi:
        lo, hi = pkg.shard.slot_range(N, rank, world)
        chunk = pkg.shard.chunk_size(N, world)
        gbuf = torch.zeros(world * chunk, 3, dtype=torch.float64, device=dev)
        sbuf = torch.zeros(chunk, 3, dtype=torch.float64, device=dev)

    def step_domain():
        lpos, lmass = dom.import_ghosts(own_pos, own_mass)
        nl = int(lpos.shape[0])
        if loc.get("n", -1) < nl:
            z3 = lambda: torch.zeros(int(nl * 1.02) + 1024, 3, dtype=torch.float64, device=dev)
            loc.update(n=int(nl * 1.02) + 1024, acc=z3(), prev=z3(), gravpm=z3(), pot=torch.zeros(int(nl * 1.02) + 1024, dtype=torch.float64, device=dev))
        loc["pos"], loc["mass"] = lpos, lmass               # keep the bound arrays alive
        eng.dev_bind_particles(lpos, lmass, box)
        eng.dev_force_tree_build()
        dom.set_global_top(n_own)
        tg = dom.own_targets(n_own, nl)
        ev0.record()
        spm.force(tg, loc["gravpm"], loc["pot"])
        ev1.record()
        loc["prev"], loc["acc"] = loc["acc"], loc["prev"]
        eng.dev_grav_short_tree(loc["acc"], prev_accel=loc["prev"], gravpm=loc["gravpm"], potential=loc["pot"], active=tg)
        loc["ghost_fraction"] = nl / n_own - 1
        eng.synchronize()
        pm_ms[0] += ev0.elapsed_time(ev1)
        pm_ms[1] += 1

    def rank_sums(x):
        """max over ranks / mean over ranks of a per-rank number"""
        t = torch.zeros(world, dtype=torch.float64, device=dev)
        t[rank] = float(x)
        dist.all_reduce(t)
        return float(t.max() / t.mean())

    def rebalance_peano():
        """domain_decompose_full again, now with the work the last walk measured per particle as the cost the TopLeaves are balanced
        by, and the exchange of the particles (with their last acceleration, which the relative opening criterion needs)"""
        nonlocal own_pos, own_mass, n_own
        cost = dforce.walk_cost(n_own)
        loc["work_before"] = rank_sums(cost.sum().item())
        loc["count_before"] = rank_sums(n_own)
        # what equal-volume x-slabs (the domains of round 1) would carry of the same work
        slab = pkg.pm_slab.slab_of_cells(own_pos[:, 0], box / nmesh, nmesh, world)
        w = torch.zeros(world, dtype=torch.float64, device=dev).index_add_(0, slab, cost.double())
        dist.all_reduce(w)
        loc["work_xslab"] = float(w.max() / w.mean())
        pdom.decompose(own_pos, cost=cost)
        own_pos, own_mass, pa, pg = pdom.exchange(own_pos, own_mass, loc["acc"], loc["gravpm"])
        n_own = int(own_pos.shape[0])
        dforce.set_domain(pdom, 6.0 * 1.5 * box / nmesh)
        z3 = lambda: torch.zeros(n_own, 3, dtype=torch.float64, device=dev)
        loc.update(acc=pa.contiguous(), prev=z3(), gravpm=pg.contiguous(), pot=torch.zeros(n_own, dtype=torch.float64, device=dev))
        step_peano()
        loc["work_after"] = rank_sums(dforce.walk_cost(n_own).sum().item())
        loc["count_after"] = rank_sums(n_own)

    def step_peano():
        loc["prev"], loc["acc"] = loc["acc"], loc["prev"]
        # the first step has no previous acceleration: Barnes-Hut opening (TreeUseBH = 2), as the reference's first step
        dforce.gravity_step(own_pos, own_mass, loc["acc"], loc["gravpm"], potential=loc["pot"], prev_accel=loc["prev"] if loc["steps"] else None)
        loc["steps"] += 1

    def step():
        nonlocal acc, prev
        if multi and args.mgpu == "peano":
            return step_peano()
        if multi and args.mgpu == "domain":
            return step_domain()
        if not multi:
            eng.dev_gravpm_force(gravpm, pot)
            eng.dev_force_tree_build()
            prev, acc = acc, prev
            eng.dev_grav_short_tree(acc, prev_accel=prev, gravpm=gravpm, potential=pot)
        elif args.mgpu == "slab":
            eng.dev_force_tree_build()
            tg = spm.targets(d_pos, eng.dev_tree_order(N, dev))
            ev0.record()
            spm.force(tg, gravpm, pot)
            ev1.record()
            prev, acc = acc, prev
            eng.dev_grav_short_tree(acc, prev_accel=prev, gravpm=gravpm, potential=pot, active=tg)
            tex.exchange(acc, tg)
            pm_ms[0] += ev0.elapsed_time(ev1)   # both events are complete: exchange() synchronised on the target counts
            pm_ms[1] += 1
        else:
            eng.dev_gravpm_force(gravpm, pot)
            eng.dev_force_tree_build()
            prev, acc = acc, prev
            optr = eng.dev_tree_order_ptr()
            eng.dev_grav_short_tree(acc, prev_accel=prev, gravpm=gravpm, potential=pot, active=optr + 4 * lo, nactive=hi - lo)
            pkg.shard.exchange_results(acc, eng.dev_tree_order(N, dev), rank, world, sbuf, gbuf)

    def sync():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    step()          # set-up, not a step: first-use allocations, FFT plans, the list-capacity adaptation of the walk and the deposit's timing trial (engine.hip, pm.hip)
    if multi and args.mgpu == "peano" and not args.no_rebalance:
        step()      # (a walk with the relative criterion: its per-particle work is what the domains are balanced by)
        rebalance_peano()
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
    if multi:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    elapsed = float(dt.item())

    # ---- untimed diagnostic passes: phase times of one more step, interaction counters of another (the counting builds of the
    # walk kernels are slower: kept out of the phase times)
    eng.set_instrumentation(True, False)
    step()
    sync()
    ph = eng.phase_times()
    eng.set_instrumentation(False, True)
    step()
    sync()
    cnt = eng.walk_counters()
    eng.set_instrumentation(False, False)
    eng.walk_events_collect()
    if os.environ.get("MPG_BENCH_DEBUG") and rank == 0:
        if multi and args.mgpu in ("domain", "peano"):
            a, b, g = loc["acc"][:n_own], loc["prev"][:n_own], loc["gravpm"][:n_own]
        else:
            a, b, g = acc, prev, gravpm
        nrm = lambda t: float(t.norm(dim=1).mean())
        print("debug: mean |acc| %.6e |prev| %.6e |gravpm| %.6e |prev+gravpm| %.6e |acc+gravpm| %.6e" % (nrm(a), nrm(b), nrm(g), nrm(b + g), nrm(a + g)), flush=True)

    out = None
    if rank == 0:
        value = N * args.steps / elapsed
        traffic, traffic_note = None, "no PMC summary committed for this configuration"
        tpath = os.path.join(ROOT, "profiles", "walk_traffic.json")
        variant = eng.walk_choice()[0]
        if os.path.exists(tpath) and N == 256 ** 3 and world == 1:
            tj = json.load(open(tpath)).get("by_ic", {}).get(args.ic, {}).get(str(variant))
            if tj:
                traffic = tj["hbm_bytes_per_launch"]
                traffic_note = "bytes per walk (%s) from %s" % (tj["kernel"], tj["method"])
        out = {
            "metric": "particle-updates/sec (gravity force step: PM + tree build + short-range walk)",
            "value": value, "unit": "particles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d^3 DM-only TreePM force step, Nmesh=%d, %s ICs, all particles active, relative opening "
                                   "criterion (ErrTolForceAcc 0.002), TreeRcut 6, Asmth 1.5" % (n, nmesh, args.ic),
                       "particles": N, "nmesh": nmesh, "parallelism": "1 GPU" if world == 1 else
                       ("%d GPUs: particles on the owners of their Peano-Hilbert TopLeaves (domain_decompose_full, %d TopLeaves); per step "
                        "{Pos, Mass} shipped to the x-slab PM (2 all-to-all transposes + neighbour planes) and {GravPM, Potential} back, "
                        "ghosts imported in whole level-La tree cells within Rcut of the rank's TopLeaves, top of the tree from an "
                        "all-reduce; choreography in the library (mpg_dist_*), collectives on RCCL" % (world, pdom.NTopLeaves)
                        if args.mgpu == "peano" else
                        "%d GPUs: particles distributed in x-slab domains, ghosts imported in whole tree-cell columns within Rcut "
                        "(one personalised exchange per step), top of the tree from an all-reduce, x-slab PM (2 all-to-all "
                        "transposes + ghost planes per step); nothing replicated or all-gathered" % world if args.mgpu == "domain" else
                        "%d GPUs: x-slab PM (2 all-to-all transposes + ghost planes per step), slab particles as targets, tree "
                        "replicated, one all-gather of accelerations" % world if args.mgpu == "slab" else
                        "targets sharded over %d GPUs (tree-order ranges), PM and tree replicated, all-gather of accelerations" % world)},
            "roofline": walk_roofline(eng, cnt, walk_ms, walk_launches, traffic, traffic_note),
            "phases_ms": {k: round(v, 3) for k, v in ph.items()},
        }
        if pm_ms[1]:
            out["phases_ms"]["pm_slab_total_incl_collectives"] = round(pm_ms[0] / pm_ms[1], 3)
        if multi and args.mgpu == "domain":
            out["config"]["ghost_fraction_rank0"] = round(loc["ghost_fraction"], 3)
        if multi and args.mgpu == "peano":
            st, tm = dforce.stats(), dforce.times()
            out["config"]["ghost_fraction_rank0"] = round(st["ghosts"] / max(n_own, 1), 3)
            out["config"]["decomposition_level_La"] = st["La"]
            out["config"]["own_particles_max_over_mean"] = round(float(pdom.task_loads.max() / pdom.task_loads.mean()), 4)
            if "work_after" in loc:
                out["config"]["load_balance"] = {
                    "walk_work_max_over_mean": round(loc["work_after"], 4), "particles_max_over_mean": round(loc["count_after"], 4),
                    "by_particle_number": {"walk_work_max_over_mean": round(loc["work_before"], 4),
                                           "particles_max_over_mean": round(loc["count_before"], 4)},
                    "x_slab_domains_walk_work_max_over_mean": round(loc["work_xslab"], 4),
                    "note": "TopLeaves dealt to the ranks by measured walk work per particle (8 x leaf entries + nodes used + 8 x "
                            "traversal steps); by_particle_number = the same step on the decomposition balanced by particle counts"}
            out["phases_ms"].update({"dist_pm_ms": round(tm["pm"], 3), "dist_ghost_import_ms": round(tm["ghosts"], 3),
                                     "dist_tree_and_top_ms": round(tm["tree"], 3), "dist_walk_ms": round(tm["walk"], 3),
                                     "dist_exchange_bytes": st["exchange_bytes"], "dist_transpose_bytes": st["transpose_bytes"]})
        if world == 1 and not multi and not args.no_cpu_baseline:    # (MPG_FORCE_MGPU frees the full arrays: no baseline leg)
            out["cpu_baseline"] = cpu_baseline(pkg, d_pos.cpu().numpy(), mass, box, n, nmesh, prev.cpu().numpy() + gravpm.cpu().numpy(),
                                               args.cpu_sample)
        if world == 1 and not multi and not args.no_extras:
            # the other input sets of SURVEY 8(d) at the same size, and the PCIe-inclusive drop-in path (untimed legs: after `value`)
            host_pos = d_pos.cpu().numpy()
            del gravpm, acc, prev, pot
            out["host_path"] = host_path_steps(pkg, eng, host_pos, mass, box)
            del d_pos, d_mass, host_pos
            torch.cuda.empty_cache()
            out["other_inputs"] = {ic: quick_gravity_steps(pkg, torch, eng, ic, n, nmesh, dev)
                                   for ic in ("s_grid", "s_zel", "s_clust") if ic != args.ic}
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    if out is not None:
        emit(out)
    return out


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


def hydro_bench(pkg, torch, args, dev):
    """BASELINE.json configs[2]: 2 x n^3 (dark matter + gas), density-entropy SPH: one force step =
    gravpm_force + force_tree_full + grav_short_tree (all particles) + force_tree_rebuild_mask(GAS) + density
    + force_tree_calc_moments (hmax) + hydro_force  (run.c:466-548)."""
    n = args.n or 128
    nmesh = 2 * n
    PE = 1 if args.sph == "pe" else 0
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

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # ---- untimed diagnostic pass: per-phase times (events on the engine's stream) and the SPH kernels against SURVEY 8(d)'s bytes
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

    def roof(kernel, b, t_ms, note):
        ach = b / (t_ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": None, "algorithmic_bytes_per_launch": b, "avg_launch_ms": t_ms, "note": note}
    out = {"metric": "particle-updates/sec (gravity + SPH force step)", "value": N * args.steps / el, "unit": "particles/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "2x%d^3 DM+gas TreePM + %s SPH force step, Nmesh=%d, s_zel ICs, quintic kernel" % (n, "pressure-entropy" if PE else "density-entropy", nmesh),
                      "particles": N, "density_iterations": iters[-args.steps:]},
           "roofline": roof("k_density", b_dens, ms[2], "one density pass incl. queue set-up and predictions; B_dens = N_tgt*128 + N_cand*28 + "
                            "N_ngb*32 (SURVEY 8(d)): %d targets, %d candidates, %d neighbours" % (sd["targets"], sd["candidates"], sd["interactions"])),
           "roofline_hydro": roof("k_hydro", b_hyd, ms[4], "B_hyd = N_tgt*176 + N_cand*36 + N_pair*100: %d candidates, %d pairs"
                                  % (sh["candidates"], sh["interactions"])),
           "phases_ms": {"gravity_pm_tree_walk": round(ms[0], 3), "gas_tree": round(ms[1], 3), "density": round(ms[2], 3),
                         "hmax": round(ms[3], 3), "hydro": round(ms[4], 3)}}
    eng.close()
    emit(out)
    return out


def hydro_ics(pkg, n):
    return pkg.ics.hydro_pair(n)


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
    df = pkg.dist.DistForce(eng, pkg.dist.TorchComm(dev))
    share = slice((N * rank) // world, (N * (rank + 1)) // world)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a[share])).to(dev)
    s_pos = T(pos)
    df.domain_decompose(s_pos, box, overdecomposition=args.overdecomp)
    o_pos, o_mass, o_typ = df.domain_exchange(s_pos, T(mass), T(typ))
    n_own = int(o_pos.shape[0])
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
    out = None
    if rank == 0:
        st = df.stats()
        out = {"metric": "particle-updates/sec (gravity + SPH force step)", "value": N * args.steps / el, "unit": "particles/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "2x%d^3 DM+gas TreePM + %s SPH force step, Nmesh=%d, s_zel ICs, quintic kernel" % (n, "pressure-entropy" if PE else "density-entropy", nmesh),
                          "particles": N, "parallelism": "%d GPUs: particles on the owners of their Peano-Hilbert TopLeaves, choreography in the library "
                                                         "(mpg_dist_*), collectives on RCCL" % world,
                          "ghost_fraction_rank0": round(st["ghosts"] / max(n_own, 1), 3), "density_iterations_last": eng.sph_stats()["iterations"]}}
    dist.barrier()
    dist.destroy_process_group()
    df.close()
    eng.close()
    if out is not None:
        emit(out)
    return out


def hydro_bench_domain(pkg, torch, dist, args, dev, rank, world):
    """configs[2] weak-scaled over GPUs with the particles distributed: x-slab domains, ghosts within
    max(Rcut, largest Hsml), gravity as in the DM-only bench, then gas tree -> density (own gas) -> the ghosts' SPH fields from
    their owners -> hmax -> hydro_force (own gas)."""
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
    rcut = 6.0 * 1.5 * box / nmesh
    dom = pkg.domain.SlabDomain(eng, box, nmesh, rank, world, dev, rcut, margin=6.0 * box / n)
    g_pos = torch.from_numpy(pos).to(dev)
    own = dom.select_own(g_pos)
    n_own = int(own.shape[0])
    o_pos, o_mass, o_typ = g_pos[own].contiguous(), torch.from_numpy(mass).to(dev)[own].contiguous(), torch.from_numpy(typ).to(dev)[own].contiguous()
    del g_pos
    o_vel, o_ent = torch.zeros(n_own, 3, **f8), torch.ones(n_own, **f8)
    o_hsml = torch.full((n_own,), 2.0 * box / n, **f8)          # first pass converges it (untimed set-up step)
    o_prev = torch.zeros(n_own, 3, **f8)
    spm = pkg.pm_slab.SlabPM(eng, box, nmesh, rank, world, dev)
    t = pkg.SphTimes()
    t.atime, t.hubble = 0.1, 0.1
    for i in range(47):
        t.dloga_bin[i] = 0.01
    FIELDS = ("hsml", "density", "egywtdensity", "dhsmlegyfac", "divvel", "curlvel")
    keep = {}

    def step():
        nonlocal o_hsml, o_prev
        lpos, lmass, ltyp, lvel, lent, lhsml = dom.import_ghosts(o_pos, o_mass, (o_typ, o_vel, o_ent, o_hsml))
        nl = int(lpos.shape[0])
        z1, z3 = (lambda: torch.zeros(nl, **f8)), (lambda: torch.zeros(nl, 3, **f8))
        gravpm, acc, prev, pot = z3(), z3(), z3(), z1()
        prev[:n_own] = o_prev
        eng.dev_bind_particles(lpos, lmass, box, type=ltyp)
        eng.dev_force_tree_build()
        dom.set_global_top(n_own)
        tg = dom.own_targets(n_own, nl)
        spm.force(tg, gravpm, pot)
        eng.dev_grav_short_tree(acc, prev_accel=prev, gravpm=gravpm, potential=pot, active=tg)
        a = dict(hsml=lhsml, dthsml=z1(), vel=lvel, entropy=lent, density=z1(), egywtdensity=z1(), dhsmlegyfac=z1(), divvel=z1(), curlvel=z1(),
                 hydroacc_out=z3(), dtentropy_out=z1(), maxsignalvel=z1())
        eng.dev_force_tree_rebuild_mask(pkg.engine.GASMASK)
        act = torch.nonzero(ltyp[:n_own] == 0).squeeze(1).to(torch.int32).contiguous()
        eng.dev_density(a, t, active=act, DoEgyDensity=PE)
        dom.check_hsml_margin(a["hsml"][:n_own])
        for k, g in zip(FIELDS, dom.ghost_update_many([a[k][:n_own] for k in FIELDS])):
            a[k][n_own:] = g
        eng.dev_force_tree_calc_hmax()
        eng.dev_hydro_force(a, t, active=act)
        eng.synchronize()
        o_hsml, o_prev = a["hsml"][:n_own].clone(), acc[:n_own].clone()
        keep.update(arrays=(lpos, lmass, ltyp, a, acc), ghost_fraction=nl / n_own - 1, it=eng.sph_stats()["iterations"])

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
    out = None
    if rank == 0:
        out = {"metric": "particle-updates/sec (gravity + SPH force step)", "value": N * args.steps / el, "unit": "particles/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "2x%d^3 DM+gas TreePM + %s SPH force step, Nmesh=%d, s_zel ICs, quintic kernel" % (n, "pressure-entropy" if PE else "density-entropy", nmesh),
                          "particles": N, "parallelism": "%d GPUs: particles distributed in x-slab domains with ghost import" % world,
                          "ghost_fraction_rank0": round(keep["ghost_fraction"], 3), "density_iterations_last": keep["it"]}}
    dist.barrier()
    dist.destroy_process_group()
    eng.close()
    if out is not None:
        emit(out)
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


def cpu_baseline(pkg, pos, mass, box, n, nmesh, aold_vec, sample):
    """The CPU "port" (SURVEY 8(d) "CPU baseline timing"): the oracle built with the reference's flags (-O3 -ffast-math -fopenmp),
    one process, OMP_NUM_THREADS = the physical cores this process may use, OMP_PROC_BIND=spread.  Bounded sample: the full tree
    build + the short-range walk for `sample` targets that are CONTIGUOUS IN TREE (Morton) ORDER (the reference walks its
    particles in Peano-Hilbert order, so neighbouring threads share nodes in cache), median of 3 walks after a warm-up, scaled to N;
    the PM part (about 10 % of a reference step) is left out, which favours the CPU."""
    phys, logical, model = host_cpu_info()
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = logical
    threads = max(1, min(phys, usable))
    # libgomp reads these when it is loaded (the oracle library is the first OpenMP user in this process)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    os.environ["OMP_PROC_BIND"] = "spread"
    os.environ["OMP_PLACES"] = "cores"
    from oracle import oracle as O
    orc = O.Oracle(fast=True)
    orc.fill_ntab(0, 1.5)
    N = len(pos)
    t0 = time.perf_counter()
    tr = orc.tree(pos, mass, box, father=False)
    t_tree = time.perf_counter() - t0
    par = O.make_grav_params(box, nmesh, npart_cbrt=n, G=G)
    par.TreeUseBH = 0
    old = np.sqrt((aold_vec ** 2).sum(1)) / G
    sample = min(sample, N)
    # Morton order of the particles (10 bits per axis are enough to make consecutive targets neighbours)
    q = np.minimum((pos / box * 1024).astype(np.int64), 1023)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249
    morton = (spread(q[:, 0]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 2])
    order = np.argsort(morton, kind="stable").astype(np.int32)
    del q, morton
    start = (N - sample) // 2
    act = np.ascontiguousarray(order[start:start + sample])
    tr.grav_short_tree(par, oldacc=old, active=act[:65536])   # warm-up
    walks, pp = [], 0
    for _ in range(3):
        t0 = time.perf_counter()
        _, _, c, _ = tr.grav_short_tree(par, oldacc=old, active=act)
        walks.append(time.perf_counter() - t0)
        pp = int(c[0])     # counters: (pair interactions, nodes visited, nodes used)
    t_walk = float(np.median(walks))
    t_full = t_tree + t_walk * N / sample
    out = {"value": N / t_full, "unit": "particles/s", "cores": orc.num_threads(), "kind": "port",
           "cpu_model": model, "physical_cores": phys, "logical_cpus": logical, "omp": "OMP_PROC_BIND=spread OMP_PLACES=cores",
           "walk_s_median_of_3": round(t_walk, 3), "walk_s_all": [round(w, 3) for w in walks], "tree_build_s": round(t_tree, 3),
           "sample": "oracle (gcc -O3 -ffast-math -fopenmp): tree build of all %d particles (%.2f s) + short-range walk of %d "
                     "tree-ordered targets (median of 3: %.2f s, %d threads) scaled to N; PM excluded" % (N, t_tree, sample, t_walk, orc.num_threads())}
    if pp:
        out["pairs_per_s_per_thread"] = pp / t_walk / orc.num_threads()
    return out


if __name__ == "__main__":
    main()
