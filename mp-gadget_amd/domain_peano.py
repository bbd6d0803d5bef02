"""Peano-Hilbert domain decomposition across the GPUs of a run (libgadget/domain.c:domain_decompose_full), one process per GPU.

The arithmetic is the engine's (csrc/domain.hip through the C-ABI: device passes over the particles, host functions on the top
tree); this module is the part the reference writes with MPI calls - the sums over ranks, the pairwise hand-over of trees
(domain_nonrecursively_combine_topTree, domain.c:1189-1270), the broadcast of the merged tree and the all-to-all of particle
records (exchange.c) - over torch.distributed (RCCL, or gloo in the CPU-launched tests)."""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import engine as E
from . import rows


class TopNode(C.Structure):
    """mpg_topnode, include/mpgadget_hip.h"""
    _fields_ = [("StartKey", C.c_uint64), ("Shift", C.c_int32), ("Daughter", C.c_int32), ("Parent", C.c_int32), ("Leaf", C.c_int32),
                ("Count", C.c_int64), ("Cost", C.c_int64)]


TOPNODE_DTYPE = np.dtype([("StartKey", "<u8"), ("Shift", "<i4"), ("Daughter", "<i4"), ("Parent", "<i4"), ("Leaf", "<i4"),
                          ("Count", "<i8"), ("Cost", "<i8")])
assert TOPNODE_DTYPE.itemsize == C.sizeof(TopNode)


def _ck(lib, rc):
    if rc:
        raise E.EngineError(lib.mpg_last_error().decode())


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class DomainPolicy:
    """DomainDecompositionPolicy, domain.c:50-57, as domain_policies_init fills it (domain.c:351-375)"""

    def __init__(self, i, ntask, overdecomposition=4):
        self.PreSort = 1 if i >= 2 else 0
        d = 256                                  # policies[k].SubSampleDistance for k = 0 .. i: 256 unless k > 4 and the previous
        for k in range(1, i + 1):                # policy's distance is above 2 (then half of it): 256 x5, 128 ... 2, 256, 128, 64, 32
            d = d // 2 if (k > 4 and d > 2) else 256
        self.SubSampleDistance = d
        self.NTopLeaves = overdecomposition * ntask * (i + 1)


NPOLICY = 16                                     # domain.c:48


class PeanoDomain:
    """DomainDecomp (domain.h:31-43) of one decomposition: TopNodes (structured array, TOPNODE_DTYPE), per leaf its Task and
    topnode, per task StartLeaf / EndLeaf, the global particle count per leaf."""

    def __init__(self, eng, box, rank=0, world=1, group=None, overdecomposition=4, alloc_factor=0.5, global_sorting=True, set_aside=None):
        self.eng, self.lib, self.box = eng, eng.lib, float(box)
        self.rank, self.world, self.group = rank, world, group
        self.overdecomposition, self.alloc_factor, self.global_sorting = overdecomposition, alloc_factor, global_sorting
        self.max_part = set_aside                # MaxPart * SetAsideFactor of domain_check_memory_bound (domain.c:549); None: no bound
        self.last_policy = 0
        self.TopNodes = None

    # ---------------------------------------------------------------- small collectives (host data)
    def _sum(self, *vals):
        v = torch.tensor(vals, dtype=torch.int64)
        if self.world > 1:
            v = v.to(self._cdev)
            dist.all_reduce(v, group=self.group)
        return [int(x) for x in v.cpu()]

    def _any(self, flag):
        return self._sum(1 if flag else 0)[0] > 0

    def _bcast_bytes(self, arr, src, n=None):
        """broadcast of a numpy array (bytes) from rank src; n = number of bytes (known to everybody)"""
        if self.world == 1:
            return arr
        t = torch.from_numpy(arr.view(np.uint8).copy() if arr is not None else np.zeros(n, np.uint8)).to(self._cdev)
        dist.broadcast(t, src, group=self.group)
        return t.cpu().numpy()

    def _send_tree(self, tree, size, dst):
        dist.send(torch.tensor([size], dtype=torch.int64).to(self._cdev), dst, group=self.group)
        dist.send(torch.from_numpy(tree[:size].view(np.uint8).copy()).to(self._cdev), dst, group=self.group)

    def _recv_tree(self, src):
        s = torch.zeros(1, dtype=torch.int64).to(self._cdev)
        dist.recv(s, src, group=self.group)
        n = int(s.item())
        b = torch.zeros(n * TOPNODE_DTYPE.itemsize, dtype=torch.uint8).to(self._cdev)
        dist.recv(b, src, group=self.group)
        return b.cpu().numpy().view(TOPNODE_DTYPE), n

    # ---------------------------------------------------------------- the two passes over the particles (device)
    def _sample(self, pos, garbage, policy):
        """the rank's sorted sample of keys (mpg_dev_domain_sample)"""
        n = int(pos.shape[0])
        cap = n // policy.SubSampleDistance + 2
        keys = np.zeros(cap, np.uint64)
        ns = C.c_int64(0)
        _ck(self.lib, self.lib.mpg_dev_domain_sample(self.eng.h, C.c_int64(n), E._ptr(pos), E._ptr(garbage), C.c_double(self.box), policy.PreSort,
                                                     policy.SubSampleDistance, _p(keys, C.c_uint64), C.c_int64(cap), C.byref(ns)))
        return keys[:ns.value]

    def _topleaves(self, pos, garbage, tree, size, nleaves, leaf_task):
        """(particles per leaf, per destination task, TopLeaf and Task of every particle) of this rank (mpg_dev_domain_topleaves);
        without leaf_task only the first"""
        n = int(pos.shape[0])
        counts = np.zeros(nleaves, np.int64)
        if leaf_task is None:
            _ck(self.lib, self.lib.mpg_dev_domain_topleaves(self.eng.h, C.c_int64(n), E._ptr(pos), E._ptr(garbage), C.c_double(self.box), _p(tree, TopNode),
                                                            size, nleaves, None, self.world, None, None, _p(counts, C.c_int64), None))
            return counts, None, None, None
        topleaf = torch.zeros(n, dtype=torch.int32, device=pos.device)
        task = torch.zeros(n, dtype=torch.int32, device=pos.device)
        tcounts = np.zeros(self.world, np.int64)
        _ck(self.lib, self.lib.mpg_dev_domain_topleaves(self.eng.h, C.c_int64(n), E._ptr(pos), E._ptr(garbage), C.c_double(self.box), _p(tree, TopNode), size,
                                                        nleaves, _p(leaf_task, C.c_int), self.world, E._ptr(topleaf), E._ptr(task), _p(counts, C.c_int64),
                                                        _p(tcounts, C.c_int64)))
        return counts, tcounts, topleaf, task

    # ---------------------------------------------------------------- the decomposition
    def _global_toptree(self, pos, garbage, policy, maxn):
        """domain_determine_global_toptree, domain.c:1280-1341: (tree, size) or None when out of top nodes"""
        lib = self.lib
        keys = self._sample(pos, garbage, policy)
        if self.global_sorting and self.world > 1:
            # mpsort_mpi (domain.c:1076-1077): the samples sorted over all ranks, every rank keeps as many as it had
            cnt = torch.tensor([len(keys)], dtype=torch.int64).to(self._cdev)
            allc = [torch.zeros_like(cnt) for _ in range(self.world)]
            dist.all_gather(allc, cnt, group=self.group)
            allc = [int(c.item()) for c in allc]
            pad = torch.zeros(max(max(allc), 1), dtype=torch.int64)
            pad[:len(keys)] = torch.from_numpy(keys.view(np.int64))
            parts = [torch.zeros_like(pad).to(self._cdev) for _ in range(self.world)]
            dist.all_gather(parts, pad.to(self._cdev), group=self.group)
            allk = np.sort(np.concatenate([p.cpu().numpy()[:c] for p, c in zip(parts, allc)]).view(np.uint64), kind="stable")
            off = sum(allc[:self.rank])
            keys = np.ascontiguousarray(allk[off:off + allc[self.rank]])
        tree = np.zeros(maxn + 8, TOPNODE_DTYPE)
        size, failed = C.c_int(0), C.c_int(0)
        _ck(lib, lib.mpg_domain_local_refine(_p(keys, C.c_uint64), None, C.c_int64(len(keys)), _p(tree, TopNode), C.byref(size), maxn, C.byref(failed)))
        if self._any(failed.value):
            return None
        totcost, totcount = self._sum(int(tree[0]["Cost"]), int(tree[0]["Count"]))
        costlimit, countlimit = totcost // policy.NTopLeaves, totcount // policy.NTopLeaves
        _ck(lib, lib.mpg_domain_toptree_truncate(_p(tree, TopNode), C.byref(size), C.c_int64(countlimit), C.c_int64(costlimit)))
        # pairwise combination up to rank 0 (domain.c:1206-1259)
        err, sep, mine = 0, 1, True
        while sep < self.world:
            if mine and self.rank % sep == 0:
                if (self.rank // sep) % 2 == 0:
                    src = self.rank + sep
                    if src < self.world:
                        other, nother = self._recv_tree(src)
                        _ck(lib, lib.mpg_domain_toptree_merge(_p(tree, TopNode), C.byref(size), _p(np.ascontiguousarray(other), TopNode), nother, maxn,
                                                              C.byref(failed)))
                        err |= failed.value
                else:
                    self._send_tree(tree, size.value, self.rank - sep)
                    mine = False
            sep *= 2
        nfinal = np.array([size.value if self.rank == 0 else 0], np.int64)
        nfinal = int(self._bcast_bytes(nfinal, 0).view(np.int64)[0])
        if nfinal >= maxn:
            err = 1
        if self._any(err):
            return None
        if self.world > 1:
            tree[:nfinal] = self._bcast_bytes(tree[:nfinal] if self.rank == 0 else None, 0, nfinal * TOPNODE_DTYPE.itemsize).view(TOPNODE_DTYPE)
        size = C.c_int(nfinal)
        _ck(lib, lib.mpg_domain_global_refine(_p(tree, TopNode), C.byref(size), maxn, C.c_int64(countlimit), C.c_int64(costlimit), C.byref(failed)))
        if self._any(failed.value):
            return None
        return tree, size.value

    def decompose(self, pos, garbage=None, cost=None):
        """domain_decompose_full up to the exchange (domain.c:153-225).  pos: [n, 3] float64 device tensor of this rank's particles,
        garbage: uint8 device tensor (IsGarbage) or None.  cost: per-particle work (float device tensor, e.g. DistForce.walk_cost())
        - the TopLeaves are then dealt to the tasks by equal WORK instead of equal particle numbers (domain_assign_balanced,
        domain.c:611: "cost").  Afterwards: self.TopNodes, .leaf_task, .leaf_topnode, .StartLeaf, .EndLeaf, .TopLeafCount (global),
        .task_loads / .task_costs, and per particle .topleaf / .task (int32 device tensors), .send_counts."""
        lib, n = self.lib, int(pos.shape[0])
        self._cdev = pos.device if (self.world > 1 and dist.get_backend(self.group) == "nccl") else torch.device("cpu")
        for i in range(self.last_policy, NPOLICY):
            policy = DomainPolicy(i, self.world, self.overdecomposition)
            while True:
                maxn = max(int(self.alloc_factor * (n + 1)), 1)      # domain_allocate, domain.c:384
                got = self._global_toptree(pos, garbage, policy, maxn)
                if self._any(got is None):
                    self.alloc_factor *= 1.2
                    if self.alloc_factor > 10:
                        raise E.EngineError("TopNodeAllocFactor unreasonably large")
                    continue
                break
            tree, size = got
            leaf_topnode = np.zeros(size, np.int32)
            nl = C.c_int(0)
            _ck(lib, lib.mpg_domain_create_topleaves(_p(tree, TopNode), size, _p(leaf_topnode, C.c_int), C.byref(nl)))
            nleaves = nl.value
            # domain_balance, domain.c:481-500
            counts = self._topleaves(pos, garbage, tree, size, nleaves, None)[0]
            if self.world > 1:
                t = torch.from_numpy(counts).to(self._cdev)
                dist.all_reduce(t, group=self.group)
                counts = t.cpu().numpy()
            leaf_cost = counts
            if cost is not None:
                # work per TopLeaf (leaves still in key order here): the sum of its particles' costs, over all ranks
                tl = self._topleaves(pos, garbage, tree, size, nleaves, np.zeros(nleaves, np.int32))[2].long()
                lc = torch.zeros(nleaves, dtype=torch.float64, device=pos.device)
                if n:
                    live = tl >= 0
                    lc.index_add_(0, tl[live], cost.to(torch.float64)[live])
                lc = lc.to(self._cdev)
                if self.world > 1:
                    dist.all_reduce(lc, group=self.group)
                leaf_cost = np.maximum(np.rint(lc.cpu().numpy()), 1).astype(np.int64)
            leaf_task = np.zeros(nleaves, np.int32)
            start, end = np.zeros(self.world, np.int32), np.zeros(self.world, np.int32)
            _ck(lib, lib.mpg_domain_assign_topleaves_balanced(_p(tree, TopNode), size, _p(leaf_topnode, C.c_int), nleaves, _p(leaf_cost, C.c_int64), self.world, 1,
                                                              _p(leaf_task, C.c_int), _p(start, C.c_int), _p(end, C.c_int)))
            # (counts were per leaf in key order; the assignment renumbers the leaves by (Task, Key): count again in the final order)
            fcounts, tcounts, self.topleaf, self.task = self._topleaves(pos, garbage, tree, size, nleaves, leaf_task)
            if self.world > 1:
                t = torch.from_numpy(fcounts).to(self._cdev)
                dist.all_reduce(t, group=self.group)
                fcounts = t.cpu().numpy()
            loads = np.array([fcounts[s:e].sum() for s, e in zip(start, end)])
            if self.max_part is not None and loads.max() > self.max_part and i < NPOLICY - 1:   # domain_check_memory_bound, domain.c:549
                continue
            self.last_policy = i
            self.policy = policy
            self.TopNodes, self.NTopNodes, self.NTopLeaves = tree[:size].copy(), size, nleaves
            self.leaf_task, self.leaf_topnode, self.StartLeaf, self.EndLeaf = leaf_task, leaf_topnode[:nleaves].copy(), start, end
            self.TopLeafCount, self.task_loads, self.send_counts = fcounts, loads, tcounts
            return self
        raise E.EngineError("No suitable domain decomposition policy worked for this particle distribution")

    def exchange(self, *columns, garbage=None):
        """domain_exchange with domain_layoutfunc (exchange.c, domain.c:794-802): every live particle goes to the task of its
        TopLeaf; garbage is dropped.  columns: device tensors with one row per particle; returns the rows this rank holds
        afterwards (its own that stay + received), in source-rank order."""
        task = self.task.to(torch.int64)
        live = task >= 0
        order = torch.argsort(torch.where(live, task, torch.full_like(task, self.world)), stable=True)
        nlive = int(live.sum().item())
        order = order[:nlive]
        counts = [int(c) for c in self.send_counts]
        assert sum(counts) == nlive, "send counts %r sum to %d, live rows %d of %d, rows per task %r" % (
            counts, sum(counts), nlive, int(task.shape[0]), torch.bincount(task[live], minlength=self.world).tolist())
        if self.world == 1:
            return [c[order] for c in columns]
        allc = rows.count_matrix(counts, self.world, columns[0].device if dist.get_backend(self.group) == "nccl" else torch.device("cpu"), self.group)
        return [rows.exchange_rows(c[order].contiguous(), counts, self.world, self.group, allc) for c in columns]

    def peano_order(self, pos, type=None):
        """The last step of domain_decompose_full (slots_gc_sorted, domain.c:238-241, slotsmanager.c:404-452): the permutation that
        puts this rank's particles (after the exchange) in (Type, Peano-Hilbert key) order; apply it to every column."""
        n = int(pos.shape[0])
        keys = torch.zeros(n, dtype=torch.int64, device=pos.device)
        perm = torch.zeros(n, dtype=torch.int32, device=pos.device)
        if n:
            self.eng.dev_peano_keys(pos, self.box, keys)
            self.eng.dev_order_by_type_and_key(keys, perm, type=type)
        return perm.long()

