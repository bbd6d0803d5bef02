// domain.h -- Peano-Hilbert domain decomposition (see domain.hip): the top-level tree over the key space, its leaves and
// their assignment to tasks (libgadget/domain.c), and the device passes over the particles that feed and apply it
#pragma once
#include "mpg_common.h"
#include <vector>

namespace mpg {

// one node of the top-level tree: struct local_topnode_data (domain.c:60-70) plus the Leaf of struct topnode_data (domain.h:12-18)
struct TopNode {
    uint64_t StartKey;
    int32_t Shift, Daughter, Parent, Leaf;
    int64_t Count, Cost;
};

// ---- host side: the tree over the sample (a few thousand nodes; every rank runs the same arithmetic on the same data)
// tree built from a rank's sorted sample of keys; costs may be null (every sample then costs 1).  false: MaxTopNodes too small
bool toptree_local_refine(const uint64_t *keys, const int64_t *costs, int64_t nsample, TopNode *tree, int *size, int MaxTopNodes);
void toptree_truncate(TopNode *tree, int *size, int64_t countlimit, int64_t costlimit);
bool toptree_merge(TopNode *A, int *sizeA, const TopNode *B, int sizeB, int MaxTopNodes);
bool toptree_global_refine(TopNode *tree, int *size, int MaxTopNodes, int64_t countlimit, int64_t costlimit);
int toptree_create_leaves(TopNode *tree, int size, int *leaf_topnode);
void toptree_assign_balanced(TopNode *tree, int size, int *leaf_topnode, int nleaves, const int64_t *cost, int NTask, int NsegmentPerTask,
                             int *leaf_task, int *StartLeaf, int *EndLeaf);

// ---- device side
struct DomainScratch {
    DevBuf<uint64_t> keys_a, keys_b, smp_a, smp_b, d_start;
    DevBuf<int> d_shift, d_daughter, d_leaf, d_leaf_task;
    DevBuf<unsigned long long> d_counts;
    DevBuf<char> tmp;
};
// the rank's sorted sample of keys (domain.c:1031-1083, local sort) -> host array; returns the number of samples
int64_t domain_sample(int64_t n, const double *pos, const uint8_t *garbage, double box, int presort, int subsample, uint64_t *h_keys, int64_t cap,
                      DomainScratch &ws, hipStream_t st);
// TopLeaf (and Task) of every particle, particles per leaf and per task (domain_compute_costs :1398-1457, the TopLeaf pass of
// domain_decompose_full :216-225, domain_layoutfunc :794-802)
void domain_topleaves(int64_t n, const double *pos, const uint8_t *garbage, double box, const TopNode *tree, int size, int nleaves,
                      const int *leaf_task, int NTask, int *d_topleaf, int *d_task, int64_t *h_leaf_counts, int64_t *h_task_counts,
                      DomainScratch &ws, hipStream_t st);
} // namespace mpg
