"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's Peano-Hilbert domain decomposition (libgadget/domain.c), used by
tests/ as the checker of the engine's domain code (mp-gadget_amd/csrc/domain.hip).  Never imported by the product.

PARITY UNPINNED: the reference holds no test or golden vector for domain.c (libgadget/tests has none; test_exchange.c covers the
particle exchange with a hand-made layout function, not the decomposition), and domain.c does not build here without MPI.  This
file restates the algorithm function by function (citations below); the keys it consumes ARE pinned (tests/test_peano.py).

All ranks of a run are simulated in one process: `decompose(ranks, ...)` takes the per-rank key arrays.
"""
import numpy as np

BITS_PER_DIMENSION = 21
PEANOCELLS = 1 << (3 * BITS_PER_DIMENSION)      # peano.h:9-10 (key of a garbage particle in the presorted sample, domain.c:1045)


class TopTree:
    """struct local_topnode_data[] (domain.c:60-70) as parallel lists"""

    def __init__(self):
        # the root: domain.c:1085-1091
        self.StartKey, self.Shift, self.Daughter, self.Parent, self.Count, self.Cost = [0], [3 * BITS_PER_DIMENSION], [-1], [-1], [0], [0]

    def size(self):
        return len(self.StartKey)

    def resize(self, n):
        for a in (self.StartKey, self.Shift, self.Daughter, self.Parent, self.Count, self.Cost):
            del a[n:]
            a.extend([0] * (n - len(a)))

    def copy(self):
        t = TopTree()
        for k in ("StartKey", "Shift", "Daughter", "Parent", "Count", "Cost"):
            setattr(t, k, list(getattr(self, k)))
        return t

    def get_subnode(self, key):                 # domain_toptree_get_subnode, domain.c:827-836
        no = 0
        while self.Daughter[no] >= 0:
            no = self.Daughter[no] + ((key - self.StartKey[no]) >> (self.Shift[no] - 3))
        return no

    def insert(self, key, cost):                # domain_toptree_insert, domain.c:838-847
        leaf = self.get_subnode(key)
        self.Count[leaf] += 1
        self.Cost[leaf] += cost
        return leaf

    def new_daughters(self, i, counts, costs):
        d = self.size()
        self.Daughter[i] = d
        self.resize(d + 8)
        for j in range(8):
            s = d + j
            self.Daughter[s], self.Parent[s] = -1, i
            self.Shift[s] = self.Shift[i] - 3
            self.StartKey[s] = self.StartKey[i] + j * (1 << self.Shift[s])
            self.Count[s], self.Cost[s] = counts[j], costs[j]

    def split(self, i, maxn):                   # domain_toptree_split, domain.c:849-883
        if self.size() + 8 > maxn:
            return 1
        assert self.Shift[i] >= 3, "Failed to build a TopTree -- particles overly clustered."
        self.new_daughters(i, [0] * 8, [0] * 8)
        return 0

    def update_cost(self, start=0):             # domain_toptree_update_cost, domain.c:885-897
        if self.Daughter[start] == -1:
            return
        for j in range(8):
            sub = self.Daughter[start] + j
            self.update_cost(sub)
            self.Count[start] += self.Count[sub]
            self.Cost[start] += self.Cost[sub]

    def truncate(self, countlimit, costlimit):  # domain_toptree_truncate(_r) + garbage_collection, domain.c:899-967
        def cut(start):
            if self.Daughter[start] == -1:
                return
            if self.Count[start] < countlimit and self.Cost[start] < costlimit:
                self.Daughter[start] = -1
                return
            for j in range(8):
                cut(self.Daughter[start] + j)
        cut(0)
        # compaction in depth-first order (domain.c:928-952 does it in place; the skeleton's nodes are created in key order, so
        # a block never moves to a higher index and copying out of a snapshot gives the same tree)
        old = self.copy()
        last = [1]

        def gc(start):
            if self.Daughter[start] == -1:
                return
            oldd, newd = self.Daughter[start], last[0]
            self.Daughter[start] = newd
            last[0] += 8
            for j in range(8):
                for k in ("StartKey", "Shift", "Daughter", "Count", "Cost"):
                    getattr(self, k)[newd + j] = getattr(old, k)[oldd + j]
                self.Parent[newd + j] = start
            for j in range(8):
                gc(newd + j)
        gc(0)
        self.resize(last[0])


def local_refine(keys_sorted, costs, maxn):
    """domain_check_for_local_refine_subsample, domain.c:1085-1180, from the sorted sample; returns (tree, failed)"""
    t = TopTree()
    last_key, last_leaf, i, n = None, -1, 0, len(keys_sorted)
    while i < n:
        leaf = t.get_subnode(int(keys_sorted[i]))
        if leaf == last_leaf and t.Shift[leaf] >= 3:
            if t.split(leaf, maxn):
                return t, 1
            t.Count[leaf] = 0
            last_leaf = t.insert(last_key, 0)
            continue
        assert not (t.Count[leaf] != 0 and leaf != last_leaf), "sample not sorted"
        last_key = int(keys_sorted[i])
        last_leaf = t.insert(last_key, 0)
        i += 1
    for k in range(t.size()):
        t.Count[k] = 0
    for k in range(n):
        t.insert(int(keys_sorted[k]), int(costs[k]))
    t.update_cost(0)
    return t, 0


def merge(A, B, noA, noB, maxn):
    """domain_toptree_merge, domain.c:1474-1577"""
    if B.Shift[noB] < A.Shift[noA]:
        if A.Daughter[noA] < 0:
            assert A.size() + 8 < maxn, "Too many Topnodes"
            count = A.Count[noA] - B.Count[B.Parent[noB]]
            cost = A.Cost[noA] - B.Cost[B.Parent[noB]]
            cdiv = lambda v, j: _cdiv((j + 1) * v, 8) - _cdiv(j * v, 8)
            A.new_daughters(noA, [cdiv(count, j) for j in range(8)], [cdiv(cost, j) for j in range(8)])
        sub = A.Daughter[noA] + ((B.StartKey[noB] - A.StartKey[noA]) >> (A.Shift[noA] - 3))
        merge(A, B, sub, noB, maxn)
    elif B.Shift[noB] == A.Shift[noA]:
        A.Count[noA] += B.Count[noB]
        A.Cost[noA] += B.Cost[noB]
        if B.Daughter[noB] >= 0:
            for j in range(8):
                merge(A, B, noA, B.Daughter[noB] + j, maxn)
        elif A.Daughter[noA] >= 0:
            for j in range(8):
                merge(A, B, A.Daughter[noA] + j, noB, maxn)
    else:
        d = B.Shift[noB] - A.Shift[noA]
        if d > 60:
            return
        n = 1 << d
        A.Count[noA] += _cdiv(B.Count[noB], n)
        A.Cost[noA] += _cdiv(B.Cost[noB], n)
        if A.Daughter[noA] >= 0:
            for j in range(8):
                merge(A, B, A.Daughter[noA] + j, noB, maxn)


def _cdiv(a, b):
    """C integer division (truncation towards zero; the difference of two counts can be negative)"""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b > 0) else -q


def global_refine(t, maxn, countlimit, costlimit):
    """domain_global_refine, domain.c:1344-1395"""
    i = 0
    while i < t.size():
        if not (t.Daughter[i] >= 0 or t.Shift[i] <= 0) and not (t.Count[i] < countlimit and t.Cost[i] < costlimit):
            if t.size() + 8 > maxn:
                return 1
            t.new_daughters(i, [_cdiv(t.Count[i], 8)] * 8, [_cdiv(t.Cost[i], 8)] * 8)
        i += 1
    return 0


def sample_keys(keys, garbage, presort, subsample):
    """the sample of one rank before its sort, domain.c:1031-1073: (keys, costs)"""
    n = len(keys)
    if presort:
        k = keys.astype(np.uint64).copy()
        cost = np.ones(n, np.int64)
        if garbage is not None:
            k[garbage != 0] = PEANOCELLS
            cost[garbage != 0] = 0
        o = np.argsort(k, kind="stable")
        k, cost = k[o], cost[o]
        ngarb = 0 if garbage is None else int((garbage != 0).sum())
        ns = (n - ngarb) // subsample
        if ns == 0 and n > ngarb:
            ns = 1
        idx = np.arange(ns) * subsample
        return k[idx], cost[idx]
    ns = n // subsample
    if ns == 0 and n != 0:
        ns = 1
    idx = np.arange(ns) * subsample
    return keys.astype(np.uint64)[idx], np.ones(ns, np.int64)


def combine(trees, maxns):
    """domain_nonrecursively_combine_topTree, domain.c:1189-1270: returns (tree of rank 0, failed)"""
    ntask = len(trees)
    trees = list(trees)
    sep, err = 1, 0
    while sep < ntask:
        for r in range(0, ntask, 2 * sep):
            if r + sep < ntask and trees[r] is not None:
                B = trees[r + sep]
                if trees[r].size() + B.size() > maxns[r]:
                    err = 1
                elif B.size() > 0:
                    merge(trees[r], B, 0, 0, maxns[r])
                trees[r + sep] = None
        sep *= 2
    if trees[0].size() >= min(maxns):
        err = 1
    return trees[0], err


def create_topleaves(t):
    """domain_create_topleaves, domain.c:810-824: (Leaf per node, topnode per leaf)"""
    leaf_of = [-1] * t.size()
    topnode = []
    stack = [0]
    while stack:
        no = stack.pop()
        if t.Daughter[no] == -1:
            leaf_of[no] = len(topnode)
            topnode.append(no)
        else:
            stack.extend(t.Daughter[no] + j for j in range(7, -1, -1))
    return leaf_of, topnode


def topleaf_of_keys(t, leaf_of, keys):
    """domain_get_topleaf, domain.h:71-78, for an array of keys"""
    start = np.array(t.StartKey, np.uint64)
    shift = np.array(t.Shift, np.int64)
    dau = np.array(t.Daughter, np.int64)
    no = np.zeros(len(keys), np.int64)
    keys = keys.astype(np.uint64)
    while True:
        act = dau[no] >= 0
        if not act.any():
            break
        a = no[act]
        no[act] = dau[a] + ((keys[act] - start[a]) >> (shift[a] - 3).astype(np.uint64)).astype(np.int64)
    return np.array(leaf_of, np.int64)[no]


def assign_topleaves_balanced(t, topnode, cost, ntask, nseg_per_task=1):
    """domain_assign_topleaves_balanced, domain.c:610-752: returns (Task per final leaf, topnode per final leaf, Leaf per node)"""
    nleaf = len(topnode)
    nseg = ntask * nseg_per_task
    order = sorted(range(nleaf), key=lambda i: t.StartKey[topnode[i]])
    ext_node = [topnode[i] for i in order]
    ext_cost = [int(cost[i]) for i in order]
    ext_task = [-1] * nleaf
    total = sum(ext_cost)
    left = total
    mean_expected, mean_task = 1.0 * total / nseg, 1.0 * total / ntask
    curleaf = curseg = curtask = nrounds = 0
    curload = curtaskload = 0
    while nrounds < nleaf:
        append = advance = False
        if curleaf == nleaf:
            advance = True
        elif nleaf - curleaf == nseg - curseg:
            append = advance = True
        else:
            assigned = (total - left) + curload
            if (mean_expected * (curseg + 1) - assigned > 0.5 * ext_cost[curleaf]) or curload == 0:
                append = True
            else:
                advance = True
        if append:
            curload += ext_cost[curleaf]
            ext_task[curleaf] = curtask
            curleaf += 1
        if advance:
            curtaskload += curload
            if (mean_task - curtaskload < 0.5 * mean_expected) or (nseg - curseg <= ntask - curtask):
                curtaskload = 0
                curtask += 1
            left -= curload
            curload = 0
            curseg += 1
            if curtask == ntask:
                curtask = 0
                mean_expected, mean_task = 1.0 * left / nseg, 1.0 * left / ntask
                nrounds += 1
            if curleaf == nleaf:
                break
    assert curseg >= nseg and left == 0
    final = sorted(range(nleaf), key=lambda i: (ext_task[i], t.StartKey[ext_node[i]]))
    leaf_of = [-1] * t.size()
    task, node = [], []
    for i, k in enumerate(final):
        leaf_of[ext_node[k]] = i
        task.append(ext_task[k])
        node.append(ext_node[k])
    return task, node, leaf_of


def task_leafs(task, ntask):
    """domain_set_task_leafs, domain.c:756-786: (StartLeaf, EndLeaf) per task"""
    tl = list(task) + [ntask]
    start, end = [0] * (ntask + 1), [0] * (ntask + 1)
    ta = 0
    for i in range(len(tl)):
        if tl[i] == ta:
            continue
        end[ta] = i
        ta += 1
        while ta < tl[i]:
            end[ta] = start[ta] = i
            ta += 1
        start[ta] = i
    assert ta == ntask
    return start[:ntask], end[:ntask]


def decompose(ranks, ntopleaves, presort=0, subsample=256, global_sort=True, alloc_factor=0.5, garbage=None, ntask=None):
    """One policy of domain_decompose_full (domain.c:153-214 with domain_attempt_decompose :427-477, domain_determine_global_toptree
    :1280-1341, domain_balance :481-500), all ranks in this process.  ranks: list of key arrays.  Returns a dict with the global
    TopNodes (tree), per-final-leaf Task / topnode, Leaf per node, Tasks (StartLeaf, EndLeaf), TopLeafCount and per-rank TopLeaf
    of every (non-garbage) particle."""
    ntask = ntask or len(ranks)
    garbage = garbage or [None] * len(ranks)
    while True:
        maxns = [int(alloc_factor * (len(k) + 1)) for k in ranks]
        samples = [sample_keys(k, g, presort, subsample) for k, g in zip(ranks, garbage)]
        if global_sort:         # mpsort_mpi, domain.c:1076-1077: globally sorted, every rank keeps its number of items
            allk = np.concatenate([s[0] for s in samples])
            allc = np.concatenate([s[1] for s in samples])
            o = np.argsort(allk, kind="stable")
            allk, allc = allk[o], allc[o]
            off = np.cumsum([0] + [len(s[0]) for s in samples])
            samples = [(allk[off[r]:off[r + 1]], allc[off[r]:off[r + 1]]) for r in range(len(ranks))]
        else:
            samples = [tuple(a[np.argsort(s[0], kind="stable")] for a in s) for s in samples]
        trees, failed = [], 0
        for (k, c), m in zip(samples, maxns):
            t, f = local_refine(k, c, m)
            trees.append(t)
            failed |= f
        if not failed:
            totcost = sum(t.Cost[0] for t in trees)
            totcount = sum(t.Count[0] for t in trees)
            costlimit, countlimit = _cdiv(totcost, ntopleaves), _cdiv(totcount, ntopleaves)
            for t in trees:
                t.truncate(countlimit, costlimit)
            tree, failed = combine(trees, maxns)
            if not failed:
                failed = global_refine(tree, min(maxns), countlimit, costlimit)
        if failed:
            alloc_factor *= 1.2
            assert alloc_factor <= 10
            continue
        break
    leaf_of, topnode = create_topleaves(tree)
    assert len(topnode) >= ntask, "Number of Topleaves is less than NTask"
    counts = np.zeros(len(topnode), np.int64)
    for k, g in zip(ranks, garbage):
        live = k if g is None else k[g == 0]
        counts += np.bincount(topleaf_of_keys(tree, leaf_of, live), minlength=len(topnode))
    task, node, leaf_of = assign_topleaves_balanced(tree, topnode, counts, ntask)
    start, end = task_leafs(task, ntask)
    final_counts = np.zeros(len(node), np.int64)
    topleaf = []
    for k, g in zip(ranks, garbage):
        tl = topleaf_of_keys(tree, leaf_of, k)
        topleaf.append(tl)
        final_counts += np.bincount(tl if g is None else tl[g == 0], minlength=len(node))
    return dict(tree=tree, Task=task, topnode=node, Leaf=leaf_of, StartLeaf=start, EndLeaf=end, TopLeafCount=final_counts, TopLeaf=topleaf,
                alloc_factor=alloc_factor)
