// ngb_walk.h -- group-cooperative neighbour search over the level-ordered tree (treewalk_visit_ngbiter / _nolist_ngbiter,
// treewalk.c:930-1265), shared by the SPH loops (sph.hip) and the pair-wise short-range gravity check (grav_pair_walk.hip).
#pragma once
#include "mpg_common.h"

namespace mpg {

#define FACT1 0.366025403785      // treewalk.c:19

// wave-wide ballot of a predicate.  (HIP's ballot64(int) compares an integer with zero: a boolean that exists only as a lane mask is
// first materialised as 0 / 1 in a VGPR and compared again - two vector instructions per ballot that this form does not need.)
__device__ __forceinline__ unsigned long long ballot64(const bool b) { return __builtin_amdgcn_ballot_w64(b); }

__device__ __forceinline__ double nearest_img(double x, double box, double invbox) { return x - box * rint(x * invbox); }

// cull_node, treewalk.c:1015-1042 (hm = 0: asymmetric search radius Hsml; symmetric: max(node hmax, Hsml))
__device__ __forceinline__ bool cull_node(const NodeGeo &g, double hm, double hsml, double px, double py, double pz, double box, double invbox)
{
    double dist = fmax(hm, hsml) + 0.5 * g.len;
    const double dx = nearest_img(g.cx - px, box, invbox);
    if(dx > dist || dx < -dist)
        return true;
    const double dy = nearest_img(g.cy - py, box, invbox);
    if(dy > dist || dy < -dist)
        return true;
    const double dz = nearest_img(g.cz - pz, box, invbox);
    if(dz > dist || dz < -dist)
        return true;
    const double r2 = dx * dx + dy * dy + dz * dz;
    dist += FACT1 * g.len;
    return r2 > dist * dist;
}

// ---------------------------------------------------------------------------------------------------------------------
// Group-cooperative neighbour search (both SPH loops).  A wave is 8 groups of 8 lanes; a group owns ONE target and walks
// the level-ordered copy of the tree (children of a node contiguous): one step pops a child range from the group's LIFO in
// LDS, the 8 lanes cull the <= 8 children (treewalk.c:1015-1042) with one coalesced read each, internal survivors push their
// own child range, and the surviving leaves are evaluated at once, lane s <-> particle s of the leaf (one coalesced read
// per group).  The visited set is the reference's; only the order of the sums differs.  (The first form, one lane per
// target walking the depth-first arrays, spent its time in dependent, uncoalesced 48-byte node reads: 27 ms per density
// pass over 2.1 M targets against the figures in DESIGN.md section 3.4.)
constexpr int SPH_STK = 160; // pending child ranges per group: <= 7 per level + 8, 21 levels

// The walk and the leaf work are separated in time so that the 8 groups of a wave stay in step: phase A walks (walk_step) and
// only records the opened leaves in a per-group list in LDS; phase B lets every group take its next leaf per iteration.
// (Interleaving them made every group wait while one group tested the leaves it had just opened.)
constexpr int SPH_LCAP = 120; // leaf entries per group; phase A pauses when a group may not fit 8 more

__device__ __forceinline__ int llist_push(unsigned *llist, int nl, const unsigned gm_leaf, const int lps, const int lpc, const int s)
{
    if(lpc > 0)
        llist[nl + __popc(gm_leaf & ((1u << s) - 1u))] = ((unsigned)lps << 4) | (unsigned)lpc;
    return nl + __popc(gm_leaf);
}

__device__ __forceinline__ double group_sum(double v)
{
    for(int off = 1; off < 8; off <<= 1)
        v += __shfl_xor(v, off);
    return v;
}

// One cooperative walk step shared by both loops: pops a child range, culls, pushes; returns in (leaf_ps, leaf_pc) the leaf
// this lane opened (pc = 0: none) and the group's mask of lanes that opened one.  SYM: symmetric search radius
// max(node hmax, Hsml) (hydro); otherwise Hsml (density).
template <bool SYM>
__device__ __forceinline__ unsigned walk_step(const TreeView &tv, unsigned *stack, int &sp, const bool valid_more, const int s, const int gshift,
                                              const double hsml, const double px, const double py, const double pz, int &leaf_ps, int &leaf_pc,
                                              bool &overflow)
{
    const bool can = valid_more;
    const unsigned range = can ? stack[sp - 1] : 0u;
    const int first = (int)(range >> 4), nch = (int)(range & 15u);
    int act = 0;
    unsigned pushval = 0;
    leaf_pc = 0;
    leaf_ps = 0;
    if(can && s < nch) {
        const int my = first + s;
        const NodeGeo g = tv.geoB[my];
        const NodeLinkB lk = tv.linkB[my];
        const double hm = SYM ? tv.hmaxB[my] : 0.0;
        if(!cull_node(g, hm, hsml, px, py, pz, tv.box, 1.0 / tv.box)) {
            if(lk.pcount > 0) {
                act = 1;
                leaf_ps = lk.pstart;
                leaf_pc = lk.pcount;
            }
            else if(lk.nchild > 0) {
                act = 3;
                pushval = ((unsigned)lk.firstchild << 4) | (unsigned)lk.nchild;
            }
        }
    }
    const unsigned gm_leaf = (unsigned)((ballot64(act == 1) >> gshift) & 0xffull);
    const unsigned gm_push = (unsigned)((ballot64(act == 3) >> gshift) & 0xffull);
    const unsigned below = (1u << s) - 1u;
    if(can && sp - 1 + __popc(gm_push) > SPH_STK)
        overflow = true;
    else if(act == 3)
        stack[sp - 1 + __popc(gm_push & below)] = pushval;
    if(can)
        sp += __popc(gm_push) - 1;
    return can ? gm_leaf : 0u;
}

} // namespace mpg
