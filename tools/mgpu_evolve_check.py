import os
"""Three force / kick / drift steps with the particles distributed over ranks (x-slab domains, ghost import every step, migration
after every drift), saving rank 0's gather of the final state by particle id; world == 1 without MPG_MGPU_MODE=domain runs the
same steps on one GPU.  Used by tests/test_gpu_timestep.py::test_distributed_evolution_matches_one_gpu."""
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
    os.environ.setdefault("MASTER_PORT", "29556")
    dist.init_process_group(os.environ.get("MPG_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
mode = os.environ.get("MPG_MGPU_MODE", "single")
G = 43.0071
nmesh = 2 * n
pos, mass, box = pkg.ics.s_zel(n)
N = len(pos)
dt = 2e-4 * box / np.sqrt(G)
vel = np.random.RandomState(4).standard_normal((N, 3)) * 0.02 * box / n / dt
f8 = dict(dtype=torch.float64, device=dev)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
eng = pkg.Engine(lr)
eng.set_walk_variant(int(os.environ.get("MPG_WALK_VARIANT", "6")))   # one kernel everywhere: the comparisons are of summation-order-identical results
eng.use_torch_stream()
eng.gravshort_fill_ntab(0, 1.5)
eng.gravpm_init_periodic(box, 1.5, nmesh, G)
eng.set_gravshort_treepar(TreeUseBH=0)
eng.gravshort_set_softenings(box / n)
K = pkg.KickFactors()
K.gravkick[0], K.bin_active[0], K.atime, K.MaxGasVel = dt, 1, 1.0, 1e30
if mode == "single":
    d_pos, d_mass, d_vel = T(pos), T(mass), T(vel)
    acc, prev, gpm = torch.zeros(N, 3, **f8), torch.zeros(N, 3, **f8), torch.zeros(N, 3, **f8)
    for step in range(3):
        eng.dev_bind_particles(d_pos, d_mass, box)
        eng.dev_gravpm_force(gpm, None)
        eng.dev_force_tree_build()
        prev, acc = acc, prev
        eng.dev_grav_short_tree(acc, prev_accel=prev, gravpm=gpm)
        eng.dev_apply_pm_half_kick(d_vel, gpm, dt)
        eng.dev_apply_half_kick(d_vel, acc, K)
        eng.dev_drift_all_particles(d_pos, d_vel, dt, box)
    final = torch.cat([d_pos, d_vel, acc], dim=1)
elif mode == "peano":
    # the library's choreography end to end (csrc/dist.hip): domain_decompose_full + exchange, then per step the force on the
    # Peano-Hilbert domains, the kicks and the drift, domain_maintain + exchange of the particles that left their owner's TopLeaves
    comm = pkg.dist.TorchComm(dev) if grouped else pkg.dist.LocalComm()
    df = pkg.dist.DistForce(eng, comm)
    share = slice((N * rank) // world, (N * (rank + 1)) // world)
    s_pos = T(pos)[share].contiguous()
    s_id = torch.arange(N, dtype=torch.int64, device=dev)[share].contiguous()
    df.domain_decompose(s_pos, box)
    o_pos, o_mass, o_vel, o_id = df.domain_exchange(s_pos, T(mass)[share].contiguous(), T(vel)[share].contiguous(), s_id)
    df.use_decomposition(box, 6.0 * 1.5 * box / nmesh)
    o_acc = torch.zeros(o_pos.shape[0], 3, **f8)
    moved = 0
    for step in range(3):
        n_own = o_pos.shape[0]
        acc, gpm = torch.zeros(n_own, 3, **f8), torch.zeros(n_own, 3, **f8)
        df.gravity_step(o_pos, o_mass, acc, gpm, prev_accel=o_acc)
        o_acc = acc
        eng.dev_apply_pm_half_kick(o_vel, gpm, dt)
        eng.dev_apply_half_kick(o_vel, o_acc, K)
        eng.dev_drift_all_particles(o_pos, o_vel, dt, box)
        eng.synchronize()
        moved += df.domain_maintain(o_pos, box)
        o_pos, o_mass, o_vel, o_acc, o_id = df.domain_exchange(o_pos, o_mass, o_vel, o_acc, o_id)
    final = torch.zeros(N, 9, **f8)
    final[o_id] = torch.cat([o_pos, o_vel, o_acc], dim=1)
    cnt = torch.tensor([o_pos.shape[0], moved], dtype=torch.int64, device=dev)
    if grouped:
        pkg.rows.TargetExchange(world, dev).exchange(final, o_id.to(torch.int32))
        dist.all_reduce(cnt)
    assert int(cnt[0].item()) == N, "particles lost or duplicated in the exchange"
    if rank == 0:
        print("peano evolution: %d particles changed owner over 3 steps" % int(cnt[1].item()), flush=True)
    df.close()
else:
    raise SystemExit("mode must be single or peano (the x-slab domains of round 1 were retired)")
torch.cuda.synchronize()
if rank == 0:
    np.save(out, final.cpu().numpy())
if grouped:
    dist.barrier()
    dist.destroy_process_group()
eng.close()
