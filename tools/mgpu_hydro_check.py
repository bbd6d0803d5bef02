import os
"""One density -> hmax -> hydro_force pass with the particles distributed over ranks (x-slab domains, ghost import), saving rank
0's view of the global results; world == 1 without MPG_MGPU_MODE=domain is the plain single-GPU pass.  Used by
tests/test_gpu_sph.py::test_sph_ranks_match_one.  MPG_DIST_BACKEND=gloo lets the ranks share one GPU."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mp-gadget_amd")
import torch
import torch.distributed as dist

out, n = sys.argv[1], int(sys.argv[2])
rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
lr = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
torch.cuda.set_device(lr)
dev = torch.device("cuda", lr)
grouped = world > 1 or bool(os.environ.get("MPG_FORCE_COLLECTIVES"))
if grouped:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29555")
    dist.init_process_group(os.environ.get("MPG_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
mode = os.environ.get("MPG_MGPU_MODE", "single")
pos, mass, box = pkg.ics.s_zel(n, box=8.0)
N = len(pos)
rng = np.random.RandomState(3)
typ = np.zeros(N, np.uint8)
typ[rng.choice(N, N // 4, replace=False)] = 1                 # dark matter mixed in
vel = rng.standard_normal((N, 3))
ent = 1.0 + 0.5 * rng.random_sample(N)
f8 = dict(dtype=torch.float64, device=dev)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
g_pos, g_mass, g_typ, g_vel, g_ent = T(pos), T(mass), T(typ), T(vel), T(ent)
g_hsml = torch.full((N,), 2.2 * box / n, **f8)
eng = pkg.Engine(lr)
eng.set_walk_variant(int(os.environ.get("MPG_WALK_VARIANT", "6")))   # one kernel everywhere: the comparisons are of summation-order-identical results
eng.use_torch_stream()
eng.set_gravshort_treepar()
eng.gravshort_set_softenings(box / n)
eng.set_densitypar(1.0, 2.0, 2.0, 99999., pkg.engine.DENSITY_KERNEL_QUINTIC_SPLINE, 0.006)
eng.set_hydropar(0, 100.0, 0.75)
t = pkg.SphTimes()
t.atime, t.hubble = 0.5, 0.3
for i in range(47):
    t.dloga_bin[i] = 0.01
FIELDS = ("hsml", "density", "egywtdensity", "dhsmlegyfac", "divvel", "curlvel")


def arrays(m, hsml, v, e):
    z1 = lambda: torch.zeros(m, **f8)
    return dict(hsml=hsml, dthsml=z1(), vel=v, entropy=e, density=z1(), egywtdensity=z1(), dhsmlegyfac=z1(), divvel=z1(), curlvel=z1(),
                hydroacc_out=torch.zeros(m, 3, **f8), dtentropy_out=z1(), maxsignalvel=z1())


every = int(os.environ.get("MPG_ACTIVE_EVERY", "0"))   # > 0: after the full step a sub-step in which every `every`-th particle is active


def kick(a):
    """what changes between the full step and the sub-step (stands for the kicks / drifts in between)"""
    a["vel"] *= 1.1
    a["entropy"] *= 1.05


if mode == "single":
    a = arrays(N, g_hsml.clone(), g_vel, g_ent)
    eng.dev_bind_particles(g_pos, g_mass, box, type=g_typ)
    eng.dev_force_tree_rebuild_mask(pkg.engine.GASMASK)
    eng.dev_density(a, t)
    eng.dev_force_tree_calc_hmax()
    eng.dev_hydro_force(a, t)
    if every:
        kick(a)
        ids = torch.arange(N, device=dev)
        act = torch.nonzero((ids % every == 0) & (g_typ == 0)).squeeze(1).to(torch.int32).contiguous()
        eng.dev_density(a, t, active=act)
        eng.dev_force_tree_calc_hmax()
        eng.dev_hydro_force(a, t, active=act)
    res = {k: a[k] for k in FIELDS + ("hydroacc_out", "dtentropy_out", "maxsignalvel")}
elif mode == "peano":
    # the library's choreography on the reference's decomposition (csrc/dist.hip): particles on the owners of their TopLeaves,
    # ghosts of every tree cell within the margin (>= the largest smoothing length), SPH columns along the ghost plan
    DP = pkg.domain_peano
    share = slice((N * rank) // world, (N * (rank + 1)) // world)
    ids = torch.arange(N, dtype=torch.int64, device=dev)[share]
    dom = DP.PeanoDomain(eng, box, rank, world)
    dom.decompose(g_pos[share].contiguous())
    o_pos, o_mass, o_typ, o_vel, o_ent, o_hsml, o_ids = dom.exchange(g_pos[share].contiguous(), g_mass[share].contiguous(), g_typ[share].contiguous(),
                                                                     g_vel[share].contiguous(), g_ent[share].contiguous(),
                                                                     g_hsml[share].contiguous(), ids)
    n_own = int(o_pos.shape[0])
    eng.gravshort_fill_ntab(0, 1.5)
    eng.gravpm_init_periodic(box, 1.5, 2 * n, 43.0071)
    comm = pkg.dist.TorchComm(dev) if grouped else pkg.dist.LocalComm()
    df = pkg.dist.DistForce(eng, comm)
    df.set_domain(dom, 6.0 * box / n)
    a = arrays(n_own, o_hsml.contiguous(), o_vel.contiguous(), o_ent.contiguous())
    if os.environ.get("MPG_SPH_HOST"):
        # the drop-in forms: the rank's particle_data records and host arrays, as shim/sph-hip.c hands them over
        Prec = pkg.make_particles(o_pos.cpu().numpy(), o_mass.cpu().numpy(), type=o_typ.cpu().numpy())
        ha = {k: np.ascontiguousarray(v.cpu().numpy()) for k, v in a.items()}
        df.host_force_tree_full(Prec)
        df.host_density(Prec, ha, t)
        df.host_hydro_force(Prec, ha, t)
        if every:
            kick(ha)
            act = np.flatnonzero(o_ids.cpu().numpy() % every == 0).astype(np.int32)     # (dark matter among them: the library skips it)
            df.host_density(Prec, ha, t, ActiveParticle=act)
            df.host_hydro_force(Prec, ha, t, ActiveParticle=act)
        a = {k: torch.from_numpy(v).to(dev) for k, v in ha.items()}
    else:
        df.force_tree_build(o_pos, o_mass)
        df.density(o_typ.contiguous(), a, t)
        df.hydro_force(n_own, a, t)
        if every:
            kick(a)
            act = torch.nonzero(o_ids % every == 0).squeeze(1).to(torch.int32).contiguous()
            df.density(o_typ.contiguous(), a, t, active=act)
            df.hydro_force(n_own, a, t, active=act)
    res = {}
    for k in FIELDS + ("hydroacc_out", "dtentropy_out", "maxsignalvel"):
        full = torch.zeros((N,) + tuple(a[k].shape[1:]), **f8)
        full[o_ids] = a[k]
        if grouped:
            pkg.rows.TargetExchange(world, dev).exchange(full.reshape(N, -1), o_ids.to(torch.int32))
        res[k] = full
    if rank == 0:
        print("hydro peano: %s own %d" % (df.stats(), n_own), flush=True)
    df.close()
else:
    raise SystemExit("mode must be single or peano (the x-slab domains of round 1 were retired)")
torch.cuda.synchronize()
if rank == 0:
    np.savez(out, typ=typ, **{k: v.cpu().numpy() for k, v in res.items()})
if grouped:
    dist.barrier()
    dist.destroy_process_group()
eng.close()
