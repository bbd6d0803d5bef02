"""One Peano-Hilbert domain decomposition + particle exchange with the particles spread over the ranks (mp-gadget_amd/domain_peano.py);
every rank saves what it ends up with.  Used by tests/test_gpu_domain.py.  MPG_DIST_BACKEND=gloo lets the ranks share one GPU."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def particle_set(n, box=100.0, seed=21):
    """half uniform, half in five clumps; 3 % garbage; unequal shares per rank"""
    rng = np.random.RandomState(seed)
    p = rng.random_sample((n, 3)) * box
    m = n // 2
    c = rng.random_sample((5, 3)) * box
    p[:m] = (c[rng.randint(0, 5, m)] + rng.standard_normal((m, 3)) * box * 0.01) % box
    p = p[rng.permutation(n)]
    garbage = (rng.random_sample(n) < 0.03).astype(np.uint8)
    return p, garbage, box


def shares(n, world):
    w = np.arange(1, world + 1, dtype=np.float64) ** 0.5
    return np.concatenate([[0], np.round(np.cumsum(w) / w.sum() * n).astype(np.int64)])


if __name__ == "__main__":
    pkg = importlib.import_module("mp-gadget_amd")
    DP = importlib.import_module("mp-gadget_amd.domain_peano")
    import torch
    import torch.distributed as dist

    out, n = sys.argv[1], int(sys.argv[2])
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    lr = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(lr)
    dev = torch.device("cuda", lr)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29556")
        dist.init_process_group(os.environ.get("MPG_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    pos, garbage, box = particle_set(n)
    cut = shares(n, world)
    lo, hi = int(cut[rank]), int(cut[rank + 1])
    ids = torch.arange(lo, hi, dtype=torch.int64, device=dev)
    d_pos = torch.from_numpy(pos[lo:hi]).to(dev)
    d_garb = torch.from_numpy(garbage[lo:hi]).to(dev)
    eng = pkg.Engine(lr)
    eng.use_torch_stream()
    dom = DP.PeanoDomain(eng, box, rank, world, overdecomposition=4, global_sorting=bool(int(os.environ.get("MPG_GLOBAL_SORT", "1"))))
    dom.decompose(d_pos, d_garb)
    got_ids, got_pos = dom.exchange(ids, d_pos)
    perm = dom.peano_order(got_pos)              # slots_gc_sorted: the rank's particles in Peano-Hilbert order
    torch.cuda.synchronize()
    # the same decomposition and exchange through the library's own choreography (csrc/dist.hip, mpg_dist_domain_*)
    lib_res = {}
    if os.environ.get("MPG_CHECK_LIB_DOMAIN", "1") != "0":
        comm = pkg.dist.TorchComm(dev) if world > 1 else pkg.dist.LocalComm()
        df = pkg.dist.DistForce(eng, comm)
        df.domain_decompose(d_pos, box, garbage=d_garb, overdecomposition=4, global_sorting=bool(int(os.environ.get("MPG_GLOBAL_SORT", "1"))))
        g = df.domain_get()
        l_ids, l_pos = df.domain_exchange(ids, d_pos)
        lib_res = dict(lib_TopNodes=g["TopNodes"], lib_leaf_task=g["leaf_task"], lib_StartLeaf=g["StartLeaf"], lib_EndLeaf=g["EndLeaf"],
                       lib_TopLeafCount=g["TopLeafCount"], lib_ids=l_ids.cpu().numpy(), lib_pos=l_pos.cpu().numpy())
        df.close()
    np.savez(out + ".%d.npz" % rank, **lib_res, TopNodes=dom.TopNodes, leaf_task=dom.leaf_task, leaf_topnode=dom.leaf_topnode, StartLeaf=dom.StartLeaf,
             EndLeaf=dom.EndLeaf, TopLeafCount=dom.TopLeafCount, topleaf=dom.topleaf.cpu().numpy(), task=dom.task.cpu().numpy(),
             ids=got_ids.cpu().numpy(), pos=got_pos.cpu().numpy(), perm=perm.cpu().numpy(), policy=np.array([dom.last_policy, dom.policy.SubSampleDistance]),
             alloc_factor=dom.alloc_factor)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
