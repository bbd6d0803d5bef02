// domain.hip -- Peano-Hilbert domain decomposition (libgadget/domain.c).
//
// What the reference does per decomposition (domain_decompose_full, domain.c:153-258): every rank samples the keys of its
// particles, builds a local top-level tree over the sample, the trees are truncated, merged pairwise up to rank 0 and refined
// until no leaf holds more than 1/NTopLeaves of the particles; the leaves are then counted over all particles and assigned
// to tasks in contiguous key segments of equal load; finally every particle is sent to the task of its leaf.
//
// Split here:
//   device   keys of all particles, the (optionally pre-sorted) strided sample and its sort        domain_sample
//            TopLeaf / Task of every particle and the particle counts per leaf and task            domain_topleaves
//            (both stream the positions once: 24 B read per particle, HBM-bound; the counts go through an LDS
//            histogram per block -- one global atomic per block and occupied leaf instead of one per particle)
//   host     the tree arithmetic on the sample (a few thousand nodes, sequential by construction)   toptree_*
//   caller   the collectives between ranks (sums of two integers, the pairwise tree merge, the all-to-all of the
//            particle records): mp-gadget_amd/domain_peano.py over torch.distributed, where the reference has MPI
// The node numbering of every tree equals the reference's (same creation order), so TopNodes / TopLeaves can be compared
// array by array with the CPU restatement the tests hold.
#include "domain.h"
#include "peano.h"
#include <algorithm>
#include <cstring>
#include <rocprim/rocprim.hpp>

namespace mpg {

namespace {
constexpr int BITS_PER_DIMENSION = 21;                               // peano.h:9
constexpr uint64_t PEANOCELLS = 1ull << (3 * BITS_PER_DIMENSION);    // peano.h:10

inline int64_t cdiv8(int64_t v, int j) { return (j + 1) * v / 8 - j * v / 8; }

// appends the 8 daughters of node i (equal eighths of its key range); counts / costs by the callback
template <typename F>
void add_daughters(TopNode *t, int *size, int i, F fill)
{
    const int d = *size;
    t[i].Daughter = d;
    for(int j = 0; j < 8; j++) {
        TopNode &s = t[d + j];
        s.Shift = t[i].Shift - 3;
        s.StartKey = t[i].StartKey + (uint64_t)j * (1ull << s.Shift);
        s.Daughter = -1;
        s.Parent = i;
        s.Leaf = -1;
        fill(j, s);
    }
    *size = d + 8;
}

inline int descend(const TopNode *t, uint64_t key)
{
    int no = 0;
    while(t[no].Daughter >= 0)
        no = t[no].Daughter + (int)((key - t[no].StartKey) >> (t[no].Shift - 3));
    return no;
}
} // namespace

// domain_check_for_local_refine_subsample, domain.c:1085-1180 (from the sorted sample on)
bool toptree_local_refine(const uint64_t *keys, const int64_t *costs, int64_t nsample, TopNode *t, int *size, int MaxTopNodes)
{
    MPG_CHECK(MaxTopNodes >= 1, "local_refine: no room for the root");
    t[0] = TopNode{0, 3 * BITS_PER_DIMENSION, -1, -1, -1, 0, 0};
    *size = 1;
    // pass 1, the skeleton: scanning sorted keys, a key either opens a fresh leaf or meets the leaf of its predecessor, which
    // is then split (the predecessor re-inserted) until the two separate or the key space is exhausted.  Count is only a
    // "visited" mark here.
    uint64_t last_key = 0;
    int last_leaf = -1;
    for(int64_t i = 0; i < nsample;) {
        const int leaf = descend(t, keys[i]);
        if(leaf == last_leaf && t[leaf].Shift >= 3) {
            if(*size + 8 > MaxTopNodes)
                return false;
            add_daughters(t, size, leaf, [](int, TopNode &s) { s.Count = 0; s.Cost = 0; });
            t[leaf].Count = 0;
            last_leaf = descend(t, last_key);
            t[last_leaf].Count++;
            continue;
        }
        MPG_CHECK(!(t[leaf].Count != 0 && leaf != last_leaf), "local_refine: the sample is not sorted by key");
        last_key = keys[i];
        last_leaf = leaf;
        t[leaf].Count++;
        i++;
    }
    // pass 2: the real counts and costs of the leaves, then of the internal nodes (children before parents: daughters are
    // stored behind their parent, so one backward sweep adds every node into its parent after its own subtree is complete)
    for(int k = 0; k < *size; k++)
        t[k].Count = t[k].Cost = 0;
    for(int64_t i = 0; i < nsample; i++) {
        TopNode &l = t[descend(t, keys[i])];
        l.Count++;
        l.Cost += costs ? costs[i] : 1;
    }
    for(int k = *size - 1; k > 0; k--) {
        t[t[k].Parent].Count += t[k].Count;
        t[t[k].Parent].Cost += t[k].Cost;
    }
    return true;
}

// domain_toptree_truncate, domain.c:899-967: branches cheaper than both limits become leaves; the survivors are renumbered
// depth first
void toptree_truncate(TopNode *t, int *size, int64_t countlimit, int64_t costlimit)
{
    const std::vector<TopNode> old(t, t + *size);
    std::vector<std::pair<int, int>> todo; // (index in the new tree, index in the old tree)
    int n = 1;
    todo.push_back({0, 0});
    while(!todo.empty()) {
        const auto [now, was] = todo.back();
        todo.pop_back();
        const TopNode &o = old[was];
        if(o.Daughter < 0 || (o.Count < countlimit && o.Cost < costlimit)) {
            t[now].Daughter = -1;
            continue;
        }
        t[now].Daughter = n;
        for(int j = 0; j < 8; j++) {
            t[n + j] = old[o.Daughter + j];
            t[n + j].Parent = now;
        }
        for(int j = 7; j >= 0; j--)
            todo.push_back({n + j, o.Daughter + j});
        n += 8;
    }
    *size = n;
}

namespace {
// domain_toptree_merge, domain.c:1474-1577
void merge_node(TopNode *A, const TopNode *B, int a, int b, int *sizeA, int MaxTopNodes)
{
    if(B[b].Shift < A[a].Shift) {
        // B is finer: descend in A, creating daughters that share what A holds beyond B's parent
        if(A[a].Daughter < 0) {
            MPG_CHECK(*sizeA + 8 < MaxTopNodes, "toptree merge: out of top nodes");
            const int64_t count = A[a].Count - B[B[b].Parent].Count, cost = A[a].Cost - B[B[b].Parent].Cost;
            add_daughters(A, sizeA, a, [&](int j, TopNode &s) {
                s.Count = cdiv8(count, j);
                s.Cost = cdiv8(cost, j);
            });
        }
        const int sub = A[a].Daughter + (int)((B[b].StartKey - A[a].StartKey) >> (A[a].Shift - 3));
        merge_node(A, B, sub, b, sizeA, MaxTopNodes);
    }
    else if(B[b].Shift == A[a].Shift) {
        A[a].Count += B[b].Count;
        A[a].Cost += B[b].Cost;
        if(B[b].Daughter >= 0) {
            for(int j = 0; j < 8; j++)
                merge_node(A, B, a, B[b].Daughter + j, sizeA, MaxTopNodes);
        }
        else if(A[a].Daughter >= 0) {
            for(int j = 0; j < 8; j++)
                merge_node(A, B, A[a].Daughter + j, b, sizeA, MaxTopNodes);
        }
    }
    else {
        // B is coarser: its content is spread evenly over the 2^d cells of A's size
        const int d = B[b].Shift - A[a].Shift;
        if(d > 60)
            return;
        const int64_t n = (int64_t)1 << d;
        A[a].Count += B[b].Count / n;
        A[a].Cost += B[b].Cost / n;
        if(A[a].Daughter >= 0)
            for(int j = 0; j < 8; j++)
                merge_node(A, B, A[a].Daughter + j, b, sizeA, MaxTopNodes);
    }
}
} // namespace

// one step of domain_nonrecursively_combine_topTree (domain.c:1232-1247): false if A has no room for B
bool toptree_merge(TopNode *A, int *sizeA, const TopNode *B, int sizeB, int MaxTopNodes)
{
    if(*sizeA + sizeB > MaxTopNodes)
        return false;
    if(sizeB > 0)
        merge_node(A, B, 0, 0, sizeA, MaxTopNodes);
    return true;
}

// domain_global_refine, domain.c:1344-1395: leaves of the merged tree above a limit are cut into eighths of the key range
bool toptree_global_refine(TopNode *t, int *size, int MaxTopNodes, int64_t countlimit, int64_t costlimit)
{
    for(int i = 0; i < *size; i++) {
        if(t[i].Daughter >= 0 || t[i].Shift <= 0)
            continue;
        if(t[i].Count < countlimit && t[i].Cost < costlimit)
            continue;
        if(*size + 8 > MaxTopNodes)
            return false;
        const int64_t c = t[i].Count / 8, w = t[i].Cost / 8;
        add_daughters(t, size, i, [&](int, TopNode &s) {
            s.Count = c;
            s.Cost = w;
        });
    }
    return true;
}

// domain_create_topleaves, domain.c:810-824: the leaves in depth-first (= key) order
int toptree_create_leaves(TopNode *t, int size, int *leaf_topnode)
{
    int nleaves = 0;
    std::vector<int> todo{0};
    while(!todo.empty()) {
        const int no = todo.back();
        todo.pop_back();
        MPG_CHECK(no >= 0 && no < size, "create_leaves: corrupt tree");
        if(t[no].Daughter == -1) {
            t[no].Leaf = nleaves;
            leaf_topnode[nleaves++] = no;
        }
        else
            for(int j = 7; j >= 0; j--)
                todo.push_back(t[no].Daughter + j);
    }
    return nleaves;
}

// domain_assign_topleaves_balanced + domain_set_task_leafs, domain.c:610-786
void toptree_assign_balanced(TopNode *t, int size, int *leaf_topnode, int nleaves, const int64_t *cost, int NTask, int NsegmentPerTask,
                             int *leaf_task, int *StartLeaf, int *EndLeaf)
{
    MPG_CHECK(nleaves >= NTask, "Number of Topleaves is less than NTask");
    struct Ext {
        uint64_t Key;
        int Task, topnode;
        int64_t cost;
    };
    std::vector<Ext> ext(nleaves);
    int64_t totalcost = 0;
    for(int i = 0; i < nleaves; i++) {
        ext[i] = Ext{t[leaf_topnode[i]].StartKey, -1, leaf_topnode[i], cost[i]};
        totalcost += cost[i];
    }
    std::stable_sort(ext.begin(), ext.end(), [](const Ext &a, const Ext &b) { return a.Key < b.Key; });
    const int Nsegment = NTask * NsegmentPerTask;
    int64_t left = totalcost, curload = 0, curtaskload = 0;
    double mean_expected = 1.0 * totalcost / Nsegment, mean_task = 1.0 * totalcost / NTask;
    int curleaf = 0, curseg = 0, curtask = 0, nrounds = 0;
    while(nrounds < nleaves) {
        bool append = false, advance = false;
        if(curleaf == nleaves)
            advance = true;
        else if(nleaves - curleaf == Nsegment - curseg)
            append = advance = true; // one leaf per remaining segment
        else {
            const int64_t assigned = (totalcost - left) + curload;
            if(mean_expected * (curseg + 1) - assigned > 0.5 * ext[curleaf].cost || curload == 0)
                append = true;
            else
                advance = true;
        }
        if(append) {
            curload += ext[curleaf].cost;
            ext[curleaf].Task = curtask;
            curleaf++;
        }
        if(advance) {
            curtaskload += curload;
            if(mean_task - curtaskload < 0.5 * mean_expected || Nsegment - curseg <= NTask - curtask) {
                curtaskload = 0;
                curtask++;
            }
            left -= curload;
            curload = 0;
            curseg++;
            if(curtask == NTask) {
                curtask = 0;
                mean_expected = 1.0 * left / Nsegment;
                mean_task = 1.0 * left / NTask;
                nrounds++;
            }
            if(curleaf == nleaves)
                break;
        }
    }
    MPG_CHECK(curseg >= Nsegment, "Not enough segments were created");
    MPG_CHECK(left == 0, "Total cost is not fully assigned to all ranks");
    std::stable_sort(ext.begin(), ext.end(), [](const Ext &a, const Ext &b) { return a.Task != b.Task ? a.Task < b.Task : a.Key < b.Key; });
    for(int i = 0; i < nleaves; i++) {
        t[ext[i].topnode].Leaf = i;
        leaf_task[i] = ext[i].Task;
        leaf_topnode[i] = ext[i].topnode;
    }
    (void)size;
    // the leaf range of every task (tasks without leaves get an empty range at the next task's start)
    int ta = 0;
    StartLeaf[0] = 0;
    for(int i = 0; i <= nleaves; i++) {
        const int task_i = i < nleaves ? leaf_task[i] : NTask;
        if(task_i == ta)
            continue;
        EndLeaf[ta++] = i;
        while(ta < task_i) {
            StartLeaf[ta] = EndLeaf[ta] = i;
            ta++;
        }
        if(ta < NTask)
            StartLeaf[ta] = i;
    }
    MPG_CHECK(ta == NTask, "domain entries found for a wrong number of tasks");
}

// ---- device passes -----------------------------------------------------------------------------------------------------
namespace {
static inline int nblk(int64_t n, int b = 256) { return (int)((n + b - 1) / b); }

__global__ void __launch_bounds__(256) k_garbage_keys(int64_t n, const uint8_t *__restrict__ garbage, uint64_t *__restrict__ keys,
                                                      unsigned long long *__restrict__ ngarbage)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool g = i < n && garbage[i] != 0;
    if(g)
        keys[i] = PEANOCELLS; // sorts behind every live key (domain.c:1044-1046)
    const unsigned long long m = __builtin_amdgcn_ballot_w64(g);
    if(m != 0 && (threadIdx.x & 63) == 0)
        atomicAdd(ngarbage, (unsigned long long)__popcll(m));
}

__global__ void __launch_bounds__(256) k_stride_gather(int64_t ns, int64_t stride, const uint64_t *__restrict__ keys, uint64_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < ns)
        out[i] = keys[i * stride];
}

// TopLeaf of every particle (domain_get_topleaf, domain.h:71-78) and the counts per leaf / task.  HIST: the block counts in
// LDS first (the particles of a block mostly share a few leaves) and adds its non-zero bins to the global arrays.
template <bool HIST>
__global__ void __launch_bounds__(256) k_topleaf(int64_t n, const uint64_t *__restrict__ keys, const uint8_t *__restrict__ garbage,
                                                 const uint64_t *__restrict__ StartKey, const int *__restrict__ Shift,
                                                 const int *__restrict__ Daughter, const int *__restrict__ Leaf, const int *__restrict__ leaf_task,
                                                 int nleaves, int *__restrict__ topleaf, int *__restrict__ task,
                                                 unsigned long long *__restrict__ counts)
{
    extern __shared__ unsigned s_hist[];
    if(HIST) {
        for(int k = threadIdx.x; k < nleaves; k += blockDim.x)
            s_hist[k] = 0;
        __syncthreads();
    }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int leaf = -1;
    if(i < n && !(garbage && garbage[i] != 0)) {
        const uint64_t key = keys[i];
        int no = 0;
        for(int d = Daughter[0]; d >= 0; d = Daughter[no])
            no = d + (int)((key - StartKey[no]) >> (Shift[no] - 3));
        leaf = Leaf[no];
    }
    if(i < n) {
        if(topleaf)
            topleaf[i] = leaf;
        if(task)
            task[i] = leaf >= 0 && leaf_task ? leaf_task[leaf] : -1;
    }
    if(HIST) {
        if(leaf >= 0)
            atomicAdd(&s_hist[leaf], 1u);
        __syncthreads();
        for(int k = threadIdx.x; k < nleaves; k += blockDim.x)
            if(s_hist[k])
                atomicAdd(&counts[k], (unsigned long long)s_hist[k]);
    }
    else if(leaf >= 0)
        atomicAdd(&counts[leaf], 1ull);
}
} // namespace

int64_t domain_sample(int64_t n, const double *pos, const uint8_t *garbage, double box, int presort, int subsample, uint64_t *h_keys, int64_t cap,
                      DomainScratch &ws, hipStream_t st)
{
    MPG_CHECK(subsample >= 1, "domain_sample: SubSampleDistance < 1");
    if(n == 0)
        return 0;
    ws.keys_a.reserve(n + 1);
    launch_peano_keys(n, pos, box, ws.keys_a.p, st);
    const uint64_t *src = ws.keys_a.p;
    int64_t ns = n / subsample;
    if(ns == 0)
        ns = 1;
    size_t tb = 0;
    if(presort) {
        // sorted by key, garbage last and left out of the sample (domain.c:1034-1062)
        ws.d_counts.reserve(1);
        MPG_HIP(hipMemsetAsync(ws.d_counts.p, 0, sizeof(unsigned long long), st));
        if(garbage)
            hipLaunchKernelGGL(k_garbage_keys, dim3(nblk(n)), dim3(256), 0, st, n, garbage, ws.keys_a.p, ws.d_counts.p);
        ws.keys_b.reserve(n + 1);
        MPG_HIP(rocprim::radix_sort_keys(nullptr, tb, ws.keys_a.p, ws.keys_b.p, (size_t)n, 0, 64, st));
        ws.tmp.reserve(tb + 16);
        MPG_HIP(rocprim::radix_sort_keys((void *)ws.tmp.p, tb, ws.keys_a.p, ws.keys_b.p, (size_t)n, 0, 64, st));
        unsigned long long ng = 0;
        MPG_HIP(hipMemcpyAsync(&ng, ws.d_counts.p, sizeof(ng), hipMemcpyDeviceToHost, st));
        MPG_HIP(hipStreamSynchronize(st));
        ns = (n - (int64_t)ng) / subsample;
        if(ns == 0 && n > (int64_t)ng)
            ns = 1;
        src = ws.keys_b.p;
    }
    MPG_CHECK(ns <= cap, "domain_sample: the output array is too small for the sample");
    if(ns == 0)
        return 0;
    ws.smp_a.reserve(ns + 1);
    ws.smp_b.reserve(ns + 1);
    hipLaunchKernelGGL(k_stride_gather, dim3(nblk(ns)), dim3(256), 0, st, ns, (int64_t)subsample, src, ws.smp_a.p);
    // the local sort of the sample (qsort_openmp(LP ...), domain.c:1079); a globally sorted sample is the caller's all-gather
    MPG_HIP(rocprim::radix_sort_keys(nullptr, tb, ws.smp_a.p, ws.smp_b.p, (size_t)ns, 0, 64, st));
    ws.tmp.reserve(tb + 16);
    MPG_HIP(rocprim::radix_sort_keys((void *)ws.tmp.p, tb, ws.smp_a.p, ws.smp_b.p, (size_t)ns, 0, 64, st));
    MPG_HIP(hipMemcpyAsync(h_keys, ws.smp_b.p, (size_t)ns * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    MPG_HIP(hipStreamSynchronize(st));
    return ns;
}

void domain_topleaves(int64_t n, const double *pos, const uint8_t *garbage, double box, const TopNode *tree, int size, int nleaves,
                      const int *leaf_task, int NTask, int *d_topleaf, int *d_task, int64_t *h_leaf_counts, int64_t *h_task_counts,
                      DomainScratch &ws, hipStream_t st)
{
    MPG_CHECK(size >= 1 && nleaves >= 1, "domain_topleaves: empty tree");
    std::vector<uint64_t> sk(size);
    std::vector<int> sh(size), da(size), lf(size);
    for(int i = 0; i < size; i++) {
        sk[i] = tree[i].StartKey;
        sh[i] = tree[i].Shift;
        da[i] = tree[i].Daughter;
        lf[i] = tree[i].Leaf;
        MPG_CHECK(da[i] >= 0 || (lf[i] >= 0 && lf[i] < nleaves), "domain_topleaves: a leaf of the tree has no TopLeaf index");
    }
    ws.d_start.reserve(size);
    ws.d_shift.reserve(size);
    ws.d_daughter.reserve(size);
    ws.d_leaf.reserve(size);
    ws.d_counts.reserve(nleaves + 1);
    MPG_HIP(hipMemcpyAsync(ws.d_start.p, sk.data(), size * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    MPG_HIP(hipMemcpyAsync(ws.d_shift.p, sh.data(), size * sizeof(int), hipMemcpyHostToDevice, st));
    MPG_HIP(hipMemcpyAsync(ws.d_daughter.p, da.data(), size * sizeof(int), hipMemcpyHostToDevice, st));
    MPG_HIP(hipMemcpyAsync(ws.d_leaf.p, lf.data(), size * sizeof(int), hipMemcpyHostToDevice, st));
    if(leaf_task) {
        ws.d_leaf_task.reserve(nleaves);
        MPG_HIP(hipMemcpyAsync(ws.d_leaf_task.p, leaf_task, nleaves * sizeof(int), hipMemcpyHostToDevice, st));
    }
    MPG_HIP(hipMemsetAsync(ws.d_counts.p, 0, (size_t)(nleaves + 1) * sizeof(unsigned long long), st));
    MPG_HIP(hipStreamSynchronize(st)); // (the host vectors above go out of scope)
    if(n > 0) {
        ws.keys_a.reserve(n + 1);
        launch_peano_keys(n, pos, box, ws.keys_a.p, st);
        const int *lt = leaf_task ? ws.d_leaf_task.p : nullptr;
        if(nleaves <= 8192)
            hipLaunchKernelGGL(k_topleaf<true>, dim3(nblk(n)), dim3(256), (size_t)nleaves * sizeof(unsigned), st, n, ws.keys_a.p, garbage, ws.d_start.p,
                               ws.d_shift.p, ws.d_daughter.p, ws.d_leaf.p, lt, nleaves, d_topleaf, d_task, ws.d_counts.p);
        else
            hipLaunchKernelGGL(k_topleaf<false>, dim3(nblk(n)), dim3(256), 0, st, n, ws.keys_a.p, garbage, ws.d_start.p, ws.d_shift.p,
                               ws.d_daughter.p, ws.d_leaf.p, lt, nleaves, d_topleaf, d_task, ws.d_counts.p);
        MPG_HIP(hipGetLastError());
    }
    std::vector<unsigned long long> c(nleaves);
    MPG_HIP(hipMemcpyAsync(c.data(), ws.d_counts.p, (size_t)nleaves * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    MPG_HIP(hipStreamSynchronize(st));
    if(h_task_counts)
        for(int k = 0; k < NTask; k++)
            h_task_counts[k] = 0;
    for(int k = 0; k < nleaves; k++) {
        if(h_leaf_counts)
            h_leaf_counts[k] = (int64_t)c[k];
        if(h_task_counts && leaf_task) {
            MPG_CHECK(leaf_task[k] >= 0 && leaf_task[k] < NTask, "domain_topleaves: a leaf is assigned to no task");
            h_task_counts[leaf_task[k]] += (int64_t)c[k];
        }
    }
}

} // namespace mpg
