"""fof_fof with the particles distributed over ranks (groups may span domain boundaries): mpg_dist_dev_fof_fof on the Peano-Hilbert
decomposition (csrc/dist.hip) against the single-GPU finder; rank 0 saves P[].GrNr of all particles and the gathered group table.
Used by tests/test_gpu_fof.py::test_fof_groups_spanning_ranks.  MPG_DIST_BACKEND=gloo lets the ranks share one GPU."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("mp-gadget_amd")
import torch
import torch.distributed as dist

out = sys.argv[1]
rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
lr = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
torch.cuda.set_device(lr)
dev = torch.device("cuda", lr)
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29557")
    dist.init_process_group(os.environ.get("MPG_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
mode = os.environ.get("MPG_MGPU_MODE", "single")


def particle_set(seed=3, nclump=120, nback=20000, box=100.0):
    """Gaussian clumps of 5 .. 3000 members (the large ones many linking lengths across: they straddle domain boundaries) on a uniform
    background; 35 % gas that attaches to the nearest dark-matter particle"""
    rng = np.random.RandomState(seed)
    N0 = nback + 150 * nclump
    LL = 0.2 * box / np.cbrt(N0)
    parts = [rng.random_sample((nback, 3)) * box]
    for _ in range(nclump):
        m = int(np.exp(rng.uniform(np.log(5), np.log(3000))))
        c = rng.random_sample(3) * box
        if rng.random_sample() < 0.2:
            c[rng.randint(3)] = rng.choice([0.02, box - 0.02])
        parts.append(np.mod(c + rng.standard_normal((m, 3)) * 0.25 * LL * np.cbrt(m), box))
    pos = np.vstack(parts)
    pos = pos[rng.permutation(len(pos))]
    N = len(pos)
    typ = np.where(rng.random_sample(N) < 0.35, 0, 1).astype(np.uint8)
    ids = rng.permutation(N).astype(np.int64) + 1000
    vel = rng.standard_normal((N, 3)) * 30.0
    mass = np.where(typ == 0, 0.19, 0.81).astype(np.float32)
    return pos, mass, typ, ids, vel, box, LL


pos, mass, typ, ids, vel, box, LL = particle_set()
N = len(pos)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
eng = pkg.Engine(lr)
eng.use_torch_stream()
minlen = 8
if mode == "single":
    d_pos, d_mass, d_typ, d_ids, d_vel = T(pos), T(mass), T(typ), T(ids), T(vel)
    eng.dev_bind_particles(d_pos, d_mass, box, type=d_typ)
    grnr = torch.zeros(N, dtype=torch.int64, device=dev)
    ng = eng.dev_fof_fof(d_ids, LL, minlen, vel=d_vel, grnr=grnr)
    G = {k: v.cpu().numpy() for k, v in eng.dev_fof_groups(ng).items()}
    tot, rounds = ng, 0
else:
    comm = pkg.dist.TorchComm(dev) if world > 1 else pkg.dist.LocalComm()
    df = pkg.dist.DistForce(eng, comm)
    share = slice((N * rank) // world, (N * (rank + 1)) // world)
    s_pos = T(pos)[share].contiguous()
    s_idx = torch.arange(N, dtype=torch.int64, device=dev)[share].contiguous()
    df.domain_decompose(s_pos, box)
    o_pos, o_mass, o_typ, o_ids, o_vel, o_idx = df.domain_exchange(s_pos, T(mass)[share].contiguous(), T(typ)[share].contiguous(),
                                                                   T(ids)[share].contiguous(), T(vel)[share].contiguous(), s_idx)
    df.use_decomposition(box, 7.0 * LL)        # the secondary attachment searches out to 6.4 linking lengths (fof.c:1235-1239)
    g_own, tot, Gh = df.fof_fof(o_pos, o_mass, o_ids, LL, minlen, type=o_typ, vel=o_vel)
    rounds = df.stats_raw()[6]
    grnr = torch.full((N,), -2, dtype=torch.int64, device=dev)
    grnr[o_idx] = g_own
    if world > 1:
        full = grnr.to(torch.float64).reshape(N, 1).contiguous()
        pkg.rows.TargetExchange(world, dev).exchange(full, o_idx.to(torch.int32))
        grnr = full[:, 0].to(torch.int64)
        tabs = [None] * world
        dist.all_gather_object(tabs, Gh)
    else:
        tabs = [Gh]
    G = {k: np.concatenate([t[k] for t in tabs]) for k in tabs[0]}
    o = np.argsort(G["MinID"], kind="stable")
    G = {k: v[o] for k, v in G.items()}
    df.close()
torch.cuda.synchronize()
if rank == 0:
    print("fof %s: %d groups, %d label rounds" % (mode, tot, rounds), flush=True)
    np.savez(out, grnr=grnr.cpu().numpy(), total=tot, rounds=rounds, **{"G_" + k: v for k, v in G.items()})
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
eng.close()
