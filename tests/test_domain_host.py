"""Host side of the Peano-Hilbert domain decomposition (the mpg_domain_* entry points of the C-ABI, csrc/domain.hip; no GPU needed)
against its restatement oracle/domain_oracle.py (libgadget/domain.c): local refinement, truncation, pairwise merge, global
refinement, leaves and their balanced assignment - node by node, the numbering included."""
import ctypes as C
import importlib

import numpy as np
import pytest

from oracle import domain_oracle as D
from test_peano import keys_from_tables

pkg = importlib.import_module("mp-gadget_amd")
DP = importlib.import_module("mp-gadget_amd.domain_peano")


def keys_of(p, box):
    fac = 1.0 / (box * 1.001) * float(1 << 21)
    return keys_from_tables(((p + box / 2000) * fac).astype(np.int32))


def clumpy(n, box, seed):
    rng = np.random.RandomState(seed)
    p = rng.random_sample((n, 3)) * box
    m = n // 2
    c = rng.random_sample((5, 3)) * box
    p[:m] = (c[rng.randint(0, 5, m)] + rng.standard_normal((m, 3)) * box * 0.01) % box
    return p


def tree_arrays(t):
    return dict(StartKey=np.array(t.StartKey, np.uint64), Shift=np.array(t.Shift), Daughter=np.array(t.Daughter), Parent=np.array(t.Parent),
                Count=np.array(t.Count), Cost=np.array(t.Cost))


def assert_tree_equal(tree, size, t, leaf=None):
    assert size == t.size()
    a = tree_arrays(t)
    for k in a:
        assert np.array_equal(tree[k][:size], a[k]), k
    if leaf is not None:
        assert np.array_equal(tree["Leaf"][:size], np.array(leaf))


def lib_decompose(lib, ranks, ntopleaves, subsample, global_sort, alloc_factor=0.5):
    """the sequence of mp-gadget_amd/domain_peano.py with all ranks in this process and the samples taken on the host"""
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    ck = lambda rc: (_ for _ in ()).throw(RuntimeError(lib.mpg_last_error().decode())) if rc else None
    ntask = len(ranks)
    while True:
        maxns = [int(alloc_factor * (len(k) + 1)) for k in ranks]
        samples = [D.sample_keys(k, None, 0, subsample)[0] for k in ranks]
        if global_sort:
            allk = np.sort(np.concatenate(samples), kind="stable")
            off = np.cumsum([0] + [len(s) for s in samples])
            samples = [np.ascontiguousarray(allk[off[r]:off[r + 1]]) for r in range(ntask)]
        else:
            samples = [np.sort(s, kind="stable") for s in samples]
        trees, sizes, bad = [], [], 0
        for s, m in zip(samples, maxns):
            tree = np.zeros(m + 8, DP.TOPNODE_DTYPE)
            size, failed = C.c_int(0), C.c_int(0)
            ck(lib.mpg_domain_local_refine(P(s, C.c_uint64), None, C.c_int64(len(s)), P(tree, DP.TopNode), C.byref(size), m, C.byref(failed)))
            bad |= failed.value
            trees.append(tree)
            sizes.append(size)
        if not bad:
            totcost = sum(int(t[0]["Cost"]) for t in trees)
            totcount = sum(int(t[0]["Count"]) for t in trees)
            costlimit, countlimit = totcost // ntopleaves, totcount // ntopleaves
            for t, s in zip(trees, sizes):
                ck(lib.mpg_domain_toptree_truncate(P(t, DP.TopNode), C.byref(s), C.c_int64(countlimit), C.c_int64(costlimit)))
            alive = [True] * ntask
            sep = 1
            failed = C.c_int(0)
            while sep < ntask:
                for r in range(0, ntask, 2 * sep):
                    if r + sep < ntask:
                        ck(lib.mpg_domain_toptree_merge(P(trees[r], DP.TopNode), C.byref(sizes[r]), P(trees[r + sep], DP.TopNode), sizes[r + sep].value,
                                                        maxns[r], C.byref(failed)))
                        bad |= failed.value
                sep *= 2
            if sizes[0].value >= min(maxns):
                bad = 1
            if not bad:
                m = min(maxns)
                tree = np.zeros(m + 8, DP.TOPNODE_DTYPE)
                tree[:sizes[0].value] = trees[0][:sizes[0].value]
                size = C.c_int(sizes[0].value)
                ck(lib.mpg_domain_global_refine(P(tree, DP.TopNode), C.byref(size), m, C.c_int64(countlimit), C.c_int64(costlimit), C.byref(failed)))
                bad |= failed.value
        if bad:
            alloc_factor *= 1.2
            continue
        return tree, size.value, alloc_factor


@pytest.mark.parametrize("ntask,subsample,global_sort", [(1, 16, True), (3, 16, True), (4, 8, False), (8, 4, True), (5, 64, False)])
def test_host_functions_match_the_oracle(ntask, subsample, global_sort):
    lib = pkg.engine.load_library()
    box = 100.0
    n = 60000
    keys = keys_of(clumpy(n, box, 7 + ntask), box)
    cuts = np.linspace(0, n, ntask + 1).astype(int)
    ranks = [keys[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    ntop = 4 * ntask
    ref = D.decompose(ranks, ntop, presort=0, subsample=subsample, global_sort=global_sort)
    tree, size, af = lib_decompose(lib, ranks, ntop, subsample, global_sort)
    assert af == ref["alloc_factor"]
    # the global tree before leaves are numbered
    rt = ref["tree"]
    assert_tree_equal(tree, size, rt)
    # leaves, counts, assignment
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    leaf_topnode = np.zeros(size, np.int32)
    nl = C.c_int(0)
    assert lib.mpg_domain_create_topleaves(P(tree, DP.TopNode), size, P(leaf_topnode, C.c_int), C.byref(nl)) == 0
    leaf_of, topnode = D.create_topleaves(rt)
    assert nl.value == len(topnode) and np.array_equal(leaf_topnode[:nl.value], topnode)
    assert np.array_equal(tree["Leaf"][:size], leaf_of)
    counts = np.zeros(nl.value, np.int64)
    for k in ranks:
        counts += np.bincount(D.topleaf_of_keys(rt, leaf_of, k), minlength=nl.value)
    leaf_task = np.zeros(nl.value, np.int32)
    start, end = np.zeros(ntask, np.int32), np.zeros(ntask, np.int32)
    assert lib.mpg_domain_assign_topleaves_balanced(P(tree, DP.TopNode), size, P(leaf_topnode, C.c_int), nl.value, P(counts, C.c_int64), ntask, 1,
                                                    P(leaf_task, C.c_int), P(start, C.c_int), P(end, C.c_int)) == 0, lib.mpg_last_error()
    assert np.array_equal(leaf_task, ref["Task"])
    assert np.array_equal(leaf_topnode[:nl.value], ref["topnode"])
    assert np.array_equal(tree["Leaf"][:size], ref["Leaf"])
    assert np.array_equal(start, ref["StartLeaf"]) and np.array_equal(end, ref["EndLeaf"])
    # what the decomposition is for: contiguous key segments of balanced load, every key in exactly one leaf
    loads = np.array([ref["TopLeafCount"][s:e].sum() for s, e in zip(start, end)])
    assert loads.sum() == n and loads.max() <= 1.5 * n / ntask + 1
    sk = rt_start = np.array(rt.StartKey, np.uint64)[np.array(ref["topnode"])]
    for s, e in zip(start, end):
        assert np.all(np.diff(sk[s:e].astype(np.float64)) > 0)


def test_local_refine_edge_cases():
    lib = pkg.engine.load_library()
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    for keys in (np.zeros(0, np.uint64), np.array([5], np.uint64), np.array([7, 7, 7], np.uint64), np.array([0, (1 << 63) - 1], np.uint64)):
        tree = np.zeros(4096, DP.TOPNODE_DTYPE)
        size, failed = C.c_int(0), C.c_int(0)
        assert lib.mpg_domain_local_refine(P(keys, C.c_uint64), None, C.c_int64(len(keys)), P(tree, DP.TopNode), C.byref(size), 4000, C.byref(failed)) == 0
        t, f = D.local_refine(keys, np.ones(len(keys), np.int64), 4000)
        assert failed.value == f == 0
        assert_tree_equal(tree, size.value, t)
    # out of nodes is reported, not fatal
    keys = np.sort(np.random.RandomState(1).randint(0, 1 << 62, 1000).astype(np.uint64))
    tree = np.zeros(64, DP.TOPNODE_DTYPE)
    assert lib.mpg_domain_local_refine(P(keys, C.c_uint64), None, C.c_int64(len(keys)), P(tree, DP.TopNode), C.byref(size), 17, C.byref(failed)) == 0
    assert failed.value == 1 == D.local_refine(keys, np.ones(len(keys), np.int64), 17)[1]
    # an unsorted sample is an error
    bad = np.array([1 << 60, 5 << 60, (1 << 60) + 1], np.uint64)      # the third key falls into a leaf that was left behind
    assert lib.mpg_domain_local_refine(P(bad, C.c_uint64), None, C.c_int64(3), P(tree, DP.TopNode), C.byref(size), 60, C.byref(failed)) != 0


# ---- N > 1 on CPU: the collective choreography of mp-gadget_amd/domain_peano.py with two gloo ranks ------------------------------
# The two passes over the particles need the GPU; here they are replaced by host stand-ins built from the oracle (test-side
# only), so that everything else - sample all-gather, sums, pairwise tree hand-over, broadcast, count all-reduce, the
# C-ABI host functions and the all-to-all of particle rows - runs as in a multi-GPU job.
def _domain_worker(rank, world, port, n, global_sort, q):
    import os
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    sys.path.insert(0, os.path.join(root, "tools"))
    import mgpu_domain_check as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    box = 100.0

    class HostPasses(DP.PeanoDomain):
        def _sample(self, pos, garbage, policy):
            k, _ = D.sample_keys(keys_of(pos.numpy(), box), None if garbage is None else garbage.numpy(), policy.PreSort, policy.SubSampleDistance)
            return np.sort(k, kind="stable")

        def _topleaves(self, pos, garbage, tree, size, nleaves, leaf_task):
            t = D.TopTree()
            t.resize(size)
            for f in ("StartKey", "Shift", "Daughter"):
                getattr(t, f)[:] = [int(v) for v in tree[f][:size]]
            tl = D.topleaf_of_keys(t, [int(v) for v in tree["Leaf"][:size]], keys_of(pos.numpy(), box))
            if garbage is not None:
                tl[garbage.numpy() != 0] = -1
            counts = np.bincount(tl[tl >= 0], minlength=nleaves).astype(np.int64)
            if leaf_task is None:
                return counts, None, None, None
            task = np.where(tl >= 0, leaf_task[np.maximum(tl, 0)], -1)
            tc = np.bincount(task[task >= 0], minlength=self.world).astype(np.int64)
            return counts, tc, torch.from_numpy(tl.astype(np.int32)), torch.from_numpy(task.astype(np.int32))

    class Eng:
        lib, h = pkg.engine.load_library(), None

    pos, garbage, _ = T.particle_set(n, box)
    cut = T.shares(n, world)
    sl = [slice(int(a), int(b)) for a, b in zip(cut[:-1], cut[1:])]
    mine = sl[rank]
    dom = HostPasses(Eng(), box, rank, world, overdecomposition=4, global_sorting=global_sort)
    dom.decompose(torch.from_numpy(pos[mine]), torch.from_numpy(garbage[mine]))
    ids, got_pos = dom.exchange(torch.arange(mine.start, mine.stop, dtype=torch.int64), torch.from_numpy(pos[mine]))
    keys = keys_of(pos, box)
    ref = D.decompose([keys[s] for s in sl], 4 * world, presort=0, subsample=256, global_sort=global_sort, garbage=[garbage[s] for s in sl])
    ok = True
    try:
        assert_tree_equal(dom.TopNodes, dom.NTopNodes, ref["tree"], ref["Leaf"])
        assert np.array_equal(dom.leaf_task, ref["Task"]) and np.array_equal(dom.StartLeaf, ref["StartLeaf"]) and np.array_equal(dom.EndLeaf, ref["EndLeaf"])
        assert np.array_equal(dom.TopLeafCount, ref["TopLeafCount"])
        task_of = np.full(n, -1, np.int64)
        for s, tl in zip(sl, ref["TopLeaf"]):
            task_of[s] = np.where(garbage[s] == 0, np.array(ref["Task"])[tl], -1)
        assert np.array_equal(ids.numpy(), np.nonzero(task_of == rank)[0])
        assert np.array_equal(got_pos.numpy(), pos[ids.numpy()])
    except AssertionError as e:
        ok = repr(e)[:500]
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,global_sort", [(2, True), (3, False)])
def test_decomposition_collectives_gloo(world, global_sort):
    import multiprocessing as mp
    import os
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_domain_worker, args=(r, world, port, 150000, global_sort, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)], res
