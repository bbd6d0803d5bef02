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

// Two child ranges per step (round 3, the SPH loops): the search is a chain of dependent steps - LDS pop -> node loads -> cull ->
// ballots -> LDS push, ~57 per wave of 8 targets in the hydro loop and 45 % of that kernel's time (cycle counters) - whose
// latency 4 waves per SIMD do not hide; taking the two topmost ranges of the LIFO at once halves the number of sequential steps for
// the same node tests.  Lane s tests child s of both ranges.  The leaves opened go to the group's list, first those of the upper range.
// Returns the new number of list entries.
template <bool SYM>
__device__ __forceinline__ int walk_step2(const TreeView &tv, unsigned *stack, int &sp, const bool valid_more, const int s, const int gshift,
                                          const double hsml, const double px, const double py, const double pz, unsigned *llist, int nl, bool &overflow)
{
    const bool can = valid_more;
    const bool can2 = can && sp > 1;
    const unsigned r1 = can ? stack[sp - 1] : 0u, r2 = can2 ? stack[sp - 2] : 0u;
    const int n1 = (int)(r1 & 15u), n2 = (int)(r2 & 15u);
    const bool t1 = s < n1, t2 = s < n2;
    // (lanes without a child read node 0: no exec-mask regions around the loads, both nodes' loads are issued together)
    const int my1 = t1 ? (int)(r1 >> 4) + s : 0, my2 = t2 ? (int)(r2 >> 4) + s : 0;
    const NodeGeo g1 = tv.geoB[my1], g2 = tv.geoB[my2];
    const NodeLinkB k1 = tv.linkB[my1], k2 = tv.linkB[my2];
    const double h1 = SYM ? tv.hmaxB[my1] : 0.0, h2 = SYM ? tv.hmaxB[my2] : 0.0;
    const double invbox = 1.0 / tv.box;
    const bool in1 = t1 && !cull_node(g1, h1, hsml, px, py, pz, tv.box, invbox);
    const bool in2 = t2 && !cull_node(g2, h2, hsml, px, py, pz, tv.box, invbox);
    const bool leaf1 = in1 && k1.pcount > 0, leaf2 = in2 && k2.pcount > 0;
    const bool push1 = in1 && k1.pcount <= 0 && k1.nchild > 0, push2 = in2 && k2.pcount <= 0 && k2.nchild > 0;
    const unsigned gl1 = (unsigned)((ballot64(leaf1) >> gshift) & 0xffull), gl2 = (unsigned)((ballot64(leaf2) >> gshift) & 0xffull);
    const unsigned gp1 = (unsigned)((ballot64(push1) >> gshift) & 0xffull), gp2 = (unsigned)((ballot64(push2) >> gshift) & 0xffull);
    const unsigned below = (1u << s) - 1u;
    const int base = sp - (can ? 1 : 0) - (can2 ? 1 : 0);
    const int np1 = __popc(gp1), np2 = __popc(gp2);
    if(can && base + np1 + np2 > SPH_STK)
        overflow = true;
    else {
        // the lower range's children below the upper range's: the search stays depth-first in the upper range
        if(push2)
            stack[base + __popc(gp2 & below)] = ((unsigned)k2.firstchild << 4) | (unsigned)k2.nchild;
        if(push1)
            stack[base + np2 + __popc(gp1 & below)] = ((unsigned)k1.firstchild << 4) | (unsigned)k1.nchild;
    }
    if(can)
        sp = base + np1 + np2;
    if(leaf1)
        llist[nl + __popc(gl1 & below)] = ((unsigned)k1.pstart << 4) | (unsigned)k1.pcount;
    const int nl1 = nl + (can ? __popc(gl1) : 0);
    if(leaf2)
        llist[nl1 + __popc(gl2 & below)] = ((unsigned)k2.pstart << 4) | (unsigned)k2.pcount;
    return nl1 + (can ? __popc(gl2) : 0);
}

} // namespace mpg
