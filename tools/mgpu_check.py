"""Run one TreePM force step - on one GPU (mode "single") or with the particles on the owners of their Peano-Hilbert TopLeaves through the
library's choreography (mode "peano", csrc/dist.hip) - and save the whole set's accelerations, GravPM and potential from rank 0 (used by
tests/test_gpu_gravity.py).  Launch with torch.distributed.run; MPG_DIST_BACKEND=gloo lets the ranks share one GPU."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mp-gadget_amd")
import torch
import torch.distributed as dist
sys.path.insert(0, ROOT)

out = sys.argv[1]
n = int(sys.argv[2])
rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
lr = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
torch.cuda.set_device(lr)
dev = torch.device("cuda", lr)
grouped = world > 1 or bool(os.environ.get("MPG_FORCE_COLLECTIVES"))
if grouped:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29555")
    dist.init_process_group(os.environ.get("MPG_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
G = 43.0071
ic = os.environ.get("MPG_MGPU_IC", "s_zel")
pos, mass, box = pkg.ics.s_clust(n, seed=5) if ic == "s_clust" else getattr(pkg.ics, ic)(n)
N = len(pos)
d_pos, d_mass = torch.from_numpy(pos).to(dev), torch.from_numpy(mass).to(dev)
eng = pkg.Engine(lr)
eng.set_walk_variant(int(os.environ.get("MPG_WALK_VARIANT", "6")))   # one kernel everywhere: the comparisons are of summation-order-identical results
eng.use_torch_stream()
eng.gravshort_fill_ntab(0, 1.5)
eng.gravpm_init_periodic(box, 1.5, 2 * n, G)
eng.set_gravshort_treepar(TreeUseBH=0)
eng.gravshort_set_softenings(box / n)
eng.dev_bind_particles(d_pos, d_mass, box)
gravpm = torch.zeros(N, 3, dtype=torch.float64, device=dev)
acc = torch.zeros_like(gravpm)
old = torch.full((N,), 1e-7, dtype=torch.float64, device=dev)
mode = os.environ.get("MPG_MGPU_MODE", "peano")
assert mode in ("single", "peano", "peano1"), mode
pot = torch.zeros(N, dtype=torch.float64, device=dev)
every = int(os.environ.get("MPG_ACTIVE_EVERY", "0"))     # > 0: a sub-step - only the particles with index % every == 0 are walked
if world == 1 and mode != "peano1":
    eng.dev_gravpm_force(gravpm, pot)
    eng.dev_force_tree_build()
    act = torch.arange(0, N, every, dtype=torch.int32, device=dev) if every else None
    if every and os.environ.get("MPG_ACTIVE_TREE"):      # hierarchical gravity: the tree holds the active particles only
        eng.dev_force_tree_active_moments(act)
    eng.dev_grav_short_tree(acc, oldacc=old, active=act)
else:
    # the library's own choreography (csrc/dist.hip): particles on their Peano-Hilbert owners (domain_decompose_full + exchange),
    # PM by shipping particles to the x-slabs, ghosts in whole level-La cells around the rank's TopLeaves, global top of the tree
    DP = pkg.domain_peano
    share = slice((N * rank) // world, (N * (rank + 1)) // world)          # what this rank holds before the decomposition
    ids = torch.arange(N, dtype=torch.int64, device=dev)[share]
    dom = DP.PeanoDomain(eng, box, rank, world, overdecomposition=int(os.environ.get("MPG_OVERDECOMP", "4")))
    dom.decompose(d_pos[share].contiguous())
    opos, omass, oids = dom.exchange(d_pos[share].contiguous(), d_mass[share].contiguous(), ids)
    n_own = int(opos.shape[0])
    comm = pkg.dist.TorchComm(dev) if grouped else pkg.dist.LocalComm()
    df = pkg.dist.DistForce(eng, comm)
    rcut = 6.0 * 1.5 * box / (2 * n)
    df.set_domain(dom, rcut)
    f8 = dict(dtype=torch.float64, device=dev)
    ga, gg, gp = torch.zeros(n_own, 3, **f8), torch.zeros(n_own, 3, **f8), torch.zeros(n_own, **f8)
    if every and os.environ.get("MPG_ACTIVE_TREE") == "host":
        # the drop-in form on the rank's particle_data records: OldAcc from P[].FullTreeGravAccel + GravPM, results in AccelStore
        df.gravpm_force(opos, omass, gg, gp)
        act = torch.nonzero(oids % every == 0).squeeze(1)
        Prec = pkg.make_particles(opos.cpu().numpy(), omass.cpu().numpy())
        Prec["FullTreeGravAccel"][:, 0] = 1e-7 * G         # |FullTreeGravAccel + GravPM| / G = 1e-7: the OldAcc of the other forms
        store = np.zeros((n_own, 3))
        df.host_grav_short_tree_active_tree(Prec, store, ActiveParticle=act.cpu().numpy())
        ga[:] = torch.from_numpy(store).to(dev)
    elif every and os.environ.get("MPG_ACTIVE_TREE"):
        df.gravpm_force(opos, omass, gg, gp)
        act = torch.nonzero(oids % every == 0).squeeze(1)
        aa = torch.zeros(act.shape[0], 3, **f8)
        df.grav_short_tree_active_tree(opos[act].contiguous(), omass[act].contiguous(), aa, oldacc=torch.full((act.shape[0],), 1e-7, **f8))
        ga[act] = aa
    elif every:     # the three calls of a sub-step: the tree holds every particle, the active ones are walked
        df.gravpm_force(opos, omass, gg, gp)
        df.force_tree_build(opos, omass)
        act = torch.nonzero(oids % every == 0).squeeze(1).to(torch.int32).contiguous()
        df.grav_short_tree(ga, oldacc=torch.full((n_own,), 1e-7, **f8), active=act)
    else:
        df.gravity_step(opos, omass, ga, gg, potential=gp, oldacc=torch.full((n_own,), 1e-7, **f8))
        if os.environ.get("MPG_INPLACE"):
            # a second step with the relative criterion's input taken from the last acceleration, once into a separate array and once
            # IN PLACE (prev_accel aliased to accel, as a resident caller keeps FullTreeGravAccel): the same values.  The engine is bound
            # to own + ghost rows here while these arrays hold the own rows only (the in-place OldAcc pass read eng->n rows).
            sep, gp2 = torch.zeros_like(ga), torch.zeros_like(gp)
            df.gravity_step(opos, omass, sep, gg, potential=gp2, prev_accel=ga)
            inp = ga.clone()
            df.gravity_step(opos, omass, inp, gg, potential=gp2, prev_accel=inp)
            torch.cuda.synchronize()
            err = float((inp - sep).abs().max() / sep.abs().mean())
            print("rank %d in-place walk vs separate arrays: %.3e" % (rank, err), flush=True)
            assert err <= 1e-12, err
            ga = inp
    both = torch.zeros(N, 7, **f8)
    both[oids] = torch.cat([ga, gg, gp[:, None]], dim=1)
    if grouped:
        pkg.rows.TargetExchange(world, dev).exchange(both, oids.to(torch.int32))
    acc, gravpm, pot = both[:, 0:3].contiguous(), both[:, 3:6].contiguous(), both[:, 6].contiguous()
    if rank == 0:
        print("peano: %s own %d times %s" % (df.stats(), n_own, df.times()), flush=True)
    df.close()
# the matter power spectrum measured on the way (gravpm.c:331-382): rows of (k, P, Nmodes)
mpc = box / 1000.0
ps = eng.gravpm_get_powerspectrum(2 * n, mpc) if (world == 1 and mode != "peano1") else None
torch.cuda.synchronize()
if rank == 0:
    if ps is not None:
        np.save(out + ".ps.npy", np.stack([ps[0], ps[1], ps[2].astype(np.float64)], axis=1))
    np.save(out, np.concatenate([acc.cpu().numpy(), gravpm.cpu().numpy(), pot.cpu().numpy()[:, None]], axis=1))
if grouped:
    dist.barrier()
    dist.destroy_process_group()
eng.close()
