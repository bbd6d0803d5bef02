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
// WRAP = false: plain differences, for the targets of a wave that all lie farther from every face of the box than their search radius
// (interior_wave below).  Exact, not approximate: a node / particle that the nearest-image form keeps has |d| <= radius + len < Box / 2 per
// axis on the unwrapped image for such a target (rint(d / Box) = 0: NEAREST() returns d itself), and one whose unwrapped |d| exceeds Box / 2
// on an axis lies, on its nearest image, beyond the face the target is `radius` away from, plus its own half length: culled by both forms.
template <bool WRAP> __device__ __forceinline__ double near_img(double x, double box, double invbox) { return WRAP ? x - box * rint(x * invbox) : x; }

// Do all targets of this wave (lanes with `valid`) lie farther than their search radius from every face of the box?  (Box / 500 of margin
// as in the gravity walk's MODE 2, and radius < Box / 4 so that "beyond Box / 2" implies "culled" for every node below the root.)
__device__ __forceinline__ bool interior_wave(const bool valid, const double px, const double py, const double pz, const double radius, const double box)
{
    const double face = radius + 0.002 * box;
    const bool out = valid && (!(radius < 0.25 * box) || fmin(fmin(px, py), pz) < face || fmax(fmax(px, py), pz) > box - face);
    return ballot64(out) == 0ull;
}

// cull_node, treewalk.c:1015-1042 (hm = 0: asymmetric search radius Hsml; symmetric: max(node hmax, Hsml))
template <bool WRAP = true>
__device__ __forceinline__ bool cull_node(const NodeGeo &g, double hm, double hsml, double px, double py, double pz, double box, double invbox)
{
#ifdef NGB_CULL_BRANCHY // (the reference's form, with its early returns: experiment switch)
    double dist = fmax(hm, hsml) + 0.5 * g.len;
    const double dx = near_img<WRAP>(g.cx - px, box, invbox);
    if(dx > dist || dx < -dist)
        return true;
    const double dy = near_img<WRAP>(g.cy - py, box, invbox);
    if(dy > dist || dy < -dist)
        return true;
    const double dz = near_img<WRAP>(g.cz - pz, box, invbox);
    if(dz > dist || dz < -dist)
        return true;
    const double r2 = dx * dx + dy * dy + dz * dz;
    dist += FACT1 * g.len;
    return r2 > dist * dist;
#else
    // (without the early returns of the reference: |d| > dist on any axis is max |d| > dist, and the wave's lanes never agree on an exit)
    const double dist = fmax(hm, hsml) + 0.5 * g.len;
    const double dx = near_img<WRAP>(g.cx - px, box, invbox);
    const double dy = near_img<WRAP>(g.cy - py, box, invbox);
    const double dz = near_img<WRAP>(g.cz - pz, box, invbox);
    const double cmax = fmax(fmax(fabs(dx), fabs(dy)), fabs(dz));
    const double r2 = dx * dx + dy * dy + dz * dz;
    const double d2 = dist + FACT1 * g.len;
    return (cmax > dist) | (r2 > d2 * d2);
#endif
}

// The same test with its two comparisons taken as lane masks (walk_stepk): the ballot of a bare comparison is the comparison's own result
// register, and the boolean algebra of the step runs once per wave on the scalar unit (as in k_walk_lists8).  Written with per-lane
// booleans, hipcc materialised every `a && b` that went into a ballot as v_cndmask 0/1 + v_cmp again.
template <bool WRAP>
__device__ __forceinline__ unsigned long long cull_mask(const NodeGeo &g, double hm, double hsml, double px, double py, double pz, double box, double invbox)
{
    const double dist = fmax(hm, hsml) + 0.5 * g.len;
    const double dx = near_img<WRAP>(g.cx - px, box, invbox);
    const double dy = near_img<WRAP>(g.cy - py, box, invbox);
    const double dz = near_img<WRAP>(g.cz - pz, box, invbox);
    const double cmax = fmax(fmax(fabs(dx), fabs(dy)), fabs(dz));
    const double r2 = dx * dx + dy * dy + dz * dz;
    const double d2 = dist + FACT1 * g.len;
    return __builtin_amdgcn_ballot_w64(cmax > dist) | __builtin_amdgcn_ballot_w64(r2 > d2 * d2);
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

// The walk and the leaf work are separated in time so that the 8 groups of a wave stay in step: phase A walks (walk_stepk) and
// only records the opened leaves in a per-group list in LDS; phase B lets every group take its next leaf per iteration.
// (Interleaving them made every group wait while one group tested the leaves it had just opened.)
constexpr int SPH_LCAP = 120; // leaf entries per group; phase A pauses when a group may not fit 8 more per child range of a step

__device__ __forceinline__ double group_sum(double v)
{
    for(int off = 1; off < 8; off <<= 1)
        v += __shfl_xor(v, off);
    return v;
}

// One cooperative search step: pops child ranges, culls, pushes; SYM: symmetric search radius max(node hmax, Hsml) (hydro);
// otherwise Hsml (density, FOF, the pair-wise gravity check).  K child ranges per step (round 3; rounds 1-2 took one): the search is a
// chain of dependent steps - LDS pop -> node loads -> cull ->
// ballots -> LDS push, ~57 per wave of 8 targets in the hydro loop and 45 % of that kernel's time (cycle counters) - whose
// latency 4 waves per SIMD do not hide; taking the K topmost ranges of the LIFO at once divides the number of sequential steps
// for the same node tests.  Lane s tests child s of every range.  The leaves opened go to the group's list, first those of the upper ranges.
// Returns the new number of list entries.
// MERGE (round 4): opened leaves of one child range whose particles are contiguous in tree order (siblings: the children of a split cell
// hold one or two particles each, gas leaves 2.7 on average) are joined into one list entry of <= 8 particles, so that the candidate tests
// of phase B run on fuller lanes (k_density 8.4 -> 7.5 ms, k_hydro 9.5 -> 8.8 at 2 x 128^3).  Which sets of siblings fit one entry is a
// property of the tree (a leaf's NodeLinkB::firstchild, written with the level-ordered copy): all children, a quad or a pair; a set is joined when every
// existing child of it was opened by this target, its first lane emitting the run.  The candidates are those of the opened leaves and no
// others, so the reference's counters are unchanged.  (First form: the lanes compared start / count / contiguity with their neighbours
// s ^ 1, s ^ 2, s ^ 4 at run time - 65 vector instructions per child range against ~20.)
// sgeo / shm: the geometry the cull tests and, for SYM, the radius per node - the nodes' cells and `hmax` as in the reference (tv.geoB,
// tv.hmaxB: FOF, the pair-wise gravity check), or the cubes around the nodes' particles and their largest Hsml (tv.geoS, tv.hsmaxS: the
// SPH loops; TreeBuilder::calc_search_boxes).
// slink / NE (round 5): the links the search follows - tv.linkB, or the SEARCH links (tv.linkS, TreeBuilder::calc_search_links) in which an
// internal node of <= 8 NE particles is a leaf of its whole particle range; such a leaf goes to the list as up to NE runs of <= 8.
template <bool SYM, int K, bool MERGE = false, bool WRAP = true, int NE = 1>
__device__ __forceinline__ int walk_stepk(const TreeView &tv, const NodeGeo *__restrict__ sgeo, const double *__restrict__ shm, unsigned *stack, int &sp,
                                          const bool valid_more, const int s, const int gshift, const double hsml, const double px, const double py,
                                          const double pz, unsigned *llist, int nl, bool &overflow, const NodeLinkB *__restrict__ slink = nullptr)
{
    static_assert(NE == 1 || NE == 2 || NE == 4, "runs per search leaf");
    if(NE == 1 || slink == nullptr)
        slink = tv.linkB;
    const bool can = valid_more;
    const double invbox = 1.0 / tv.box;
    const unsigned below = (1u << s) - 1u;
    unsigned r[K];
    int my[K];
    bool tst[K];
    // (more than one range only while the LIFO has room for all their children: a depth-first search with one range per step needs at
    // most 7 entries per tree level + 8, which is what SPH_STK is sized for; K ranges at once could need up to K times that)
    const int take = (SPH_STK - sp >= 7 * K) ? K : 1;
#pragma unroll
    for(int k = 0; k < K; k++) { // range k: the k-th entry from the top of the LIFO
        r[k] = (can && sp > k && k < take) ? stack[sp - 1 - k] : 0u;
        tst[k] = s < (int)(r[k] & 15u);
        my[k] = tst[k] ? (int)(r[k] >> 4) + s : 0; // (lanes without a child read node 0: no exec-mask regions, all loads issued together)
    }
    NodeGeo g[K];
    NodeLinkB lk[K];
    double hm[K];
#pragma unroll
    for(int k = 0; k < K; k++) {
        g[k] = sgeo[my[k]];
        lk[k] = slink[my[k]];
        hm[k] = SYM ? shm[my[k]] : 0.0;
    }
    unsigned gl[K], gp[K], ent[K];
    unsigned pcn[K], gx[K][NE > 1 ? NE - 1 : 1]; // NE > 1: particles this lane lists; the group's lanes that list >= 2, >= 3, >= 4 runs
    unsigned long long m_x[K][NE > 1 ? NE - 1 : 1];
    unsigned long long m_leaf[K], m_push[K]; // lane masks: the children opened as leaves / whose own children are pushed
#pragma unroll
    for(int k = 0; k < K; k++) {
#ifndef NGB_NO_MASKS
        const unsigned long long m_in = __builtin_amdgcn_ballot_w64(tst[k]) & ~cull_mask<WRAP>(g[k], hm[k], hsml, px, py, pz, tv.box, invbox);
        const unsigned long long m_pc = __builtin_amdgcn_ballot_w64(lk[k].pcount > 0);
        m_leaf[k] = m_in & m_pc;
        m_push[k] = m_in & ~m_pc & __builtin_amdgcn_ballot_w64(lk[k].nchild > 0);
#else // (experiment switch: per-lane booleans, as until round 4)
        const bool in = tst[k] && !cull_node<WRAP>(g[k], hm[k], hsml, px, py, pz, tv.box, invbox);
        m_leaf[k] = ballot64(in && lk[k].pcount > 0);
        m_push[k] = ballot64(in && lk[k].pcount <= 0 && lk[k].nchild > 0);
#endif
        ent[k] = ((unsigned)lk[k].pstart << 4) | (unsigned)lk[k].pcount;
        if(MERGE) {
            const bool lf = __builtin_amdgcn_inverse_ballot_w64(m_leaf[k]);
            unsigned pcm = lf ? (unsigned)lk[k].pcount : 0u;
#ifdef NGB_MERGE_ANY_OPENED
            // Round 5 experiment (measured, no gain - profiles/r05a_experiments): a set is joined as soon as the target opened ANY leaf of it, its first lane emitting the whole run - whether or not
            // that lane's own leaf was opened.  One list entry is one test iteration of 8 lanes whatever it holds, so the particles of the
            // set's other leaves ride along on lanes that would idle (they fail the distance test: their leaf was culled), while a set
            // opened in part is one entry instead of one per opened leaf.  (The default joins a set only when ALL its existing leaves were
            // opened, which the tighter cull on the particles' cubes makes rarer.)
            const bool isleaf = tst[k] && lk[k].pcount > 0;
            const unsigned h = isleaf ? (unsigned)lk[k].firstchild : 0u;              // (a leaf's merge hints: NodeLinkB)
            const unsigned m = (unsigned)((m_leaf[k] >> gshift) & 0xffull);           // the children this target opened as leaves
            const unsigned sq = (h >> 4) & 15u, sp = h & 15u;
            pcm = (sp != 0u && (m & (3u << (s & 6))) != 0u) ? ((s & 1) == 0 ? sp : 0u) : pcm;
            pcm = (sq != 0u && (m & (15u << (s & 4))) != 0u) ? ((s & 3) == 0 ? sq : 0u) : pcm;
#else
            const unsigned h = lf ? (unsigned)lk[k].firstchild : 0u;                  // (a leaf's merge hints: NodeLinkB)
            const unsigned m = (unsigned)((m_leaf[k] >> gshift) & 0xffull);           // the children this target opened as leaves
            const unsigned x = m ^ ((1u << (r[k] & 15u)) - 1u);                       // existing children that are not among them
            const unsigned sq = (h >> 4) & 15u, sp = h & 15u;
            pcm = (sp != 0u && (x & (3u << (s & 6))) == 0u) ? ((s & 1) == 0 ? sp : 0u) : pcm;
            pcm = (sq != 0u && (x & (15u << (s & 4))) == 0u) ? ((s & 3) == 0 ? sq : 0u) : pcm;
#ifdef NGB_MERGE_OCT // (all children as one run: cannot occur in a tree whose cells are split at their 9th particle)
            pcm = (((h >> 8) & 15u) != 0u && x == 0u) ? (s == 0 ? ((h >> 8) & 15u) : 0u) : pcm;
#endif
#endif
            ent[k] = ((unsigned)lk[k].pstart << 4) | (NE > 1 ? min(pcm, 8u) : pcm);
            m_leaf[k] = __builtin_amdgcn_ballot_w64(pcm != 0u);
            pcn[k] = pcm;
        }
        else
            pcn[k] = (unsigned)lk[k].pcount;
        gl[k] = (unsigned)((m_leaf[k] >> gshift) & 0xffull);
        gp[k] = (unsigned)((m_push[k] >> gshift) & 0xffull);
        if(NE > 1) {
            static_assert(NE == 1 || MERGE, "runs per search leaf: the merging form only");
#pragma unroll
            for(int e = 1; e < NE; e++) { // (pcn is 0 on the lanes that list nothing)
                m_x[k][e - 1] = __builtin_amdgcn_ballot_w64(pcn[k] > 8u * (unsigned)e);
                gx[k][e - 1] = (unsigned)((m_x[k][e - 1] >> gshift) & 0xffull);
            }
        }
    }
    const int taken = can ? (sp < take ? sp : take) : 0;
    const int base = sp - taken;
    int npush = 0;
#pragma unroll
    for(int k = 0; k < K; k++)
        npush += __popc(gp[k]);
    if(can && base + npush > SPH_STK)
        overflow = true;
    else {
        // the children of the lower ranges below those of the upper ones: the search stays depth-first in the topmost range
        int at = base;
#pragma unroll
        for(int k = K - 1; k >= 0; k--) {
            if(__builtin_amdgcn_inverse_ballot_w64(m_push[k]))
                stack[at + __popc(gp[k] & below)] = ((unsigned)lk[k].firstchild << 4) | (unsigned)lk[k].nchild;
            at += __popc(gp[k]);
        }
    }
    if(can)
        sp = base + npush;
#pragma unroll
    for(int k = 0; k < K; k++) {
        if(NE == 1) {
            if(__builtin_amdgcn_inverse_ballot_w64(m_leaf[k]))
                llist[nl + __popc(gl[k] & below)] = ent[k];
            nl += can ? __popc(gl[k]) : 0;
        }
        else {
            // a lane's runs are consecutive list entries; its first one sits behind all runs of the lanes below it
            int pos = nl + __popc(gl[k] & below), tot = __popc(gl[k]);
#pragma unroll
            for(int e = 1; e < NE; e++) {
                pos += __popc(gx[k][e - 1] & below);
                tot += __popc(gx[k][e - 1]);
            }
            if(__builtin_amdgcn_inverse_ballot_w64(m_leaf[k]))
                llist[pos] = ent[k];
#pragma unroll
            for(int e = 1; e < NE; e++)
                if(__builtin_amdgcn_inverse_ballot_w64(m_x[k][e - 1]))
                    llist[pos + e] = (((unsigned)lk[k].pstart + 8u * (unsigned)e) << 4) | min(pcn[k] - 8u * (unsigned)e, 8u);
            nl += can ? tot : 0;
        }
    }
    return nl;
}

} // namespace mpg
