// tree_build.hip -- GPU construction of the reference's bucket oct-tree (libgadget/forcetree.c) for gfx950.
//
// The reference inserts particles one by one (forcetree.c:481-520) and merges per-thread sub-trees
// (:526-650).  The resulting topology is canonical: a cell is an internal node iff it
// holds more than NMAXCHILD=8 particles, children are the geometric octants (x | y<<1 | z<<2,
// forcetree.c:278-284) with centre = parent centre +- len/4 and len/2 (:302-320), empty children are
// pruned when moments are computed (:1032-1049), root len = 1.001*BoxSize centred on BoxSize/2 (:662-664).
// This file builds exactly that node set without insertion:
//   1. k_keys       per particle: 21-level octant path, obtained by replaying the reference's own
//                   floating-point descent (Pos > centre; centre +- 0.25*len; len *= 0.5) so that every
//                   octant decision is bit-identical to get_subnode();
//   2. rocPRIM radix sort of (key, index);
//   3. k_leaflevel  per sorted particle: depth of its leaf = the shallowest cell around it with <= 8
//                   particles, from the common-prefix lengths with its +-8 neighbours;
//   4. scan         depth-first pre-order node numbering: a leaf head emits the internal nodes that start
//                   at it (levels common+1 .. leaf-1) followed by the leaf;
//   5. k_fill_nodes geometry, particle ranges and `sibling` (= first node after the sub-tree);
//   6. k_leaf_moments / k_internal_moments (bottom-up by level): mass, centre of mass, hmax
//                   (forcetree.c:947-966, :985-1104).
// Node numbering and in-leaf particle order differ from the reference (they also differ between two
// reference runs with different thread counts); the node *set*, geometry and moments are the same.
#include "tree_build.h"
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>

namespace mpg {

__device__ __forceinline__ int cpl_levels(uint64_t a, uint64_t b)
{
    // number of leading 3-bit octant digits two 63-bit keys share (bit 63 is always clear)
    const uint64_t x = a ^ b;
    const int lz = x ? __clzll((long long)x) : 64;
    const int l = (lz - 1) / 3;
    return l > MAXLEVEL ? MAXLEVEL : l;
}

// include: optional byte per particle, 0 = leave out (the active-particle trees of force_tree_active_moments)
// the particles a tree holds: the types of the mask (forcetree.c:357-365), and of those the ones flagged in `include` if given
struct TreeMember {
    const uint8_t *type, *include;
    int mask;
    __device__ bool operator()(const uint32_t i) const
    {
        const int ty = type ? (type[i] & 7) : 1;
        return (((1 << ty) & mask) != 0) && (!include || include[i]);
    }
};

// octant path of particle idx[k] (all particles in caller order when idx is null, in which case idx is written too)
__global__ void __launch_bounds__(256) k_keys(int64_t n, const double *__restrict__ pos, const uint32_t *__restrict__ members, double box,
                                              uint64_t *__restrict__ keys, uint32_t *__restrict__ idx)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= n)
        return;
    const int64_t i = members ? (int64_t)members[k] : k;
    const double x = pos[3 * i + 0], y = pos[3 * i + 1], z = pos[3 * i + 2];
    double cx = box / 2., cy = box / 2., cz = box / 2.;
    double len = box * 1.001;
    uint64_t key = 0;
#pragma unroll 1
    for(int l = 0; l < MAXLEVEL; l++) {
        const double q = 0.25 * len;
        const int bx = x > cx, by = y > cy, bz = z > cz;
        key = (key << 3) | (uint64_t)(bx | (by << 1) | (bz << 2));
        cx += bx ? q : -q;
        cy += by ? q : -q;
        cz += bz ? q : -q;
        len *= 0.5;
    }
    keys[k] = key;
    if(!members)
        idx[k] = (uint32_t)i;
}

// the top bits of the keys as 32-bit keys, and the positions they are sorted with
__global__ void __launch_bounds__(256) k_key_tops(int64_t n, const uint64_t *__restrict__ keys, int shift, uint32_t *__restrict__ hi, uint32_t *__restrict__ pos)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= n)
        return;
    hi[k] = (uint32_t)(keys[k] >> shift);
    pos[k] = (uint32_t)k;
}

__global__ void __launch_bounds__(256) k_apply_order(int64_t n, const uint32_t *__restrict__ pos, const uint64_t *__restrict__ keys_in,
                                                     const uint32_t *__restrict__ idx_in, uint64_t *__restrict__ keys_out, uint32_t *__restrict__ idx_out)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= n)
        return;
    const uint32_t p = pos[k];
    keys_out[k] = keys_in[p];
    idx_out[k] = idx_in[p];
}

__global__ void __launch_bounds__(256) k_gather_src(int64_t n, const uint32_t *__restrict__ order, const double *__restrict__ pos,
                                                    const float *__restrict__ mass, Src4 *__restrict__ src)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= n)
        return;
    const int64_t i = order[k];
    Src4 s;
    s.x = pos[3 * i + 0];
    s.y = pos[3 * i + 1];
    s.z = pos[3 * i + 2];
    s.m = (double)mass[i];
    src[k] = s;
}

// flags[0]: error (too many coincident particles), flags[1]: max leaf level, flags[2]: MAXLEVEL - min leaf level
// Each block stages its 256 keys plus an 8-key halo on either side in LDS (one coalesced read instead of 17 per thread).
__global__ void __launch_bounds__(256) k_leaflevel(int64_t n, const uint64_t *__restrict__ keys, uint8_t *__restrict__ leaflevel,
                                                   uint32_t *__restrict__ cnt, int *__restrict__ flags, int *__restrict__ wave_ext, int minlevel)
{
    __shared__ uint64_t sk[256 + 16];
    const int64_t base = (int64_t)blockIdx.x * blockDim.x;
    const int64_t i = base + threadIdx.x;
    for(int t = threadIdx.x; t < 256 + 16; t += 256) {
        const int64_t j = base - 8 + t;
        sk[t] = (j >= 0 && j < n) ? keys[j] : 0;
    }
    __syncthreads();
    const bool inb = i < n;
    int L = 0;
    bool head = false;
    if(inb) {
        const int c0 = threadIdx.x + 8;
        const uint64_t ki = sk[c0];
        int cl[9], cr[9];
        cl[0] = cr[0] = MAXLEVEL + 1;
#pragma unroll
        for(int a = 1; a <= 8; a++) {
            cl[a] = (i - a >= 0) ? cpl_levels(ki, sk[c0 - a]) : -1;
            cr[a] = (i + a < n) ? cpl_levels(ki, sk[c0 + a]) : -1;
        }
        int best = -1;
#pragma unroll
        for(int a = 0; a <= 8; a++) {
            const int m = cl[a] < cr[8 - a] ? cl[a] : cr[8 - a];
            best = m > best ? m : best;
        }
        L = best + 1; // shallowest level whose cell holds <= 8 particles
        L = L < minlevel ? minlevel : L; // (domain-decomposed runs: cells above `minlevel` stay internal, see TreeBuilder::top_set)
        if(L > MAXLEVEL) {
            flags[0] = 1;
            L = MAXLEVEL;
        }
        leaflevel[i] = (uint8_t)L;
        const int c = (i == 0) ? -1 : cl[1];
        head = (i == 0) || (c < L);
        cnt[i] = head ? (uint32_t)(L - c) : 0u;
    }
    // level extrema: one word per wave, reduced by k_level_extrema.  (Same-address traffic from every wave - atomics at ~90
    // per microsecond, or 2 x 262144 uncached reads of one word to avoid them - cost this kernel 2-4 ms at 256^3.)
    int lmax = head ? L : 0, lmin = head ? (MAXLEVEL - L) : 0;
    for(int off = 32; off > 0; off >>= 1) {
        lmax = max(lmax, __shfl_down(lmax, off));
        lmin = max(lmin, __shfl_down(lmin, off));
    }
    if((threadIdx.x & 63) == 0)
        wave_ext[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = lmax | (lmin << 8);
}

// flags[1] = deepest leaf level, flags[2] = MAXLEVEL - shallowest leaf level
__global__ void __launch_bounds__(256) k_level_extrema(int64_t nwaves, const int *__restrict__ wave_ext, int *__restrict__ flags)
{
    int lmax = 0, lmin = 0;
    for(int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwaves; w += (int64_t)gridDim.x * blockDim.x) {
        const int v = wave_ext[w];
        lmax = max(lmax, v & 255);
        lmin = max(lmin, v >> 8);
    }
    for(int off = 32; off > 0; off >>= 1) {
        lmax = max(lmax, __shfl_down(lmax, off));
        lmin = max(lmin, __shfl_down(lmin, off));
    }
    if((threadIdx.x & 63) == 0) {
        atomicMax(&flags[1], lmax);
        atomicMax(&flags[2], lmin);
    }
}

__global__ void __launch_bounds__(256) k_fill_nodes(int64_t n, const uint64_t *__restrict__ keys, const uint8_t *__restrict__ leaflevel,
                                                    const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ base, double box,
                                                    NodeGeo *__restrict__ geo, NodeLink *__restrict__ link)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n || cnt[i] == 0)
        return;
    const uint64_t ki = keys[i];
    const int L = leaflevel[i];
    const int first = L - (int)cnt[i] + 1; // shallowest new level
    // size of the leaf: particles up to the next head
    int pc = 1;
    while(i + pc < n && cnt[i + pc] == 0)
        pc++;
    double cx = box / 2., cy = box / 2., cz = box / 2.;
    double len = box * 1.001;
    for(int l = 0; l <= L; l++) {
        if(l >= first) {
            const int64_t j = (int64_t)base[i] + (l - first);
            geo[j] = NodeGeo{cx, cy, cz, len};
            NodeLink lk;
            lk.level = l;
            lk.pstart = (int)i;
            int64_t e;
            if(l == L) {
                lk.pcount = pc;
                e = i + pc;
            }
            else {
                lk.pcount = 0;
                // first particle after i that leaves this level-l cell: keys are sorted.  Nine internal nodes in ten are parents of leaves and
                // hold 9 .. 64 particles, so the end is looked for by doubling steps from the leaf's end first (probes within a few hundred
                // bytes of keys[i]) and by bisection inside the bracket that finds (round 6: a bisection over [i, n) from the start took
                // ~24 dependent loads, the first dozen of them megabytes apart - 0.90 ms of the 3.5 ms tree build at 256^3)
                const int shift = 3 * (MAXLEVEL - l);
                const uint64_t pref = (l == 0) ? 0 : (ki >> shift);
                int64_t lo = i + pc, hi = n; // everything in the leaf shares the prefix
                if(l > 0)
                    for(int64_t step = 8;; step <<= 1) {
                        const int64_t p = lo - 1 + step;
                        if(p >= n)
                            break;
                        if((keys[p] >> shift) == pref)
                            lo = p + 1;
                        else {
                            hi = p;
                            break;
                        }
                    }
                else
                    lo = n; // (the root holds everything)
                while(lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    const uint64_t km = keys[mid];
                    const bool same = (l == 0) ? true : ((km >> shift) == pref);
                    if(same)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                e = lo;
            }
            lk.sibling = (e < n) ? (int)base[e] : -1;
            link[j] = lk;
        }
        if(l < L) {
            const int d = (int)((ki >> (3 * (MAXLEVEL - 1 - l))) & 7);
            const double q = 0.25 * len;
            cx += (d & 1) ? q : -q;
            cy += (d & 2) ? q : -q;
            cz += (d & 4) ? q : -q;
            len *= 0.5;
        }
    }
}

// Round 6: the same records, one thread per NODE.  k_fill_nodes above runs one thread per particle of which only the leaf heads (one in
// eight) have work, and of those one in eight a bracket search: 0.55 ms at 256^3 after the doubling steps.  k_node_heads scatters every
// head's index to the nodes it starts; k_fill_nodes_n then lets node j descend from the root to its own level along its head's key - the
// reference's arithmetic (centre +- len / 4, len / 2) in the same order, so the geometry is bit-identical - and finds its end.
__global__ void __launch_bounds__(256) k_node_heads(int64_t n, const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ base,
                                                    uint32_t *__restrict__ head)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    const uint32_t c = cnt[i];
    const uint32_t b = c ? base[i] : 0u;
    for(uint32_t k = 0; k < c; k++)
        head[b + k] = (uint32_t)i;
}

__global__ void __launch_bounds__(256) k_fill_nodes_n(int64_t nnodes, int64_t n, const uint32_t *__restrict__ head, const uint64_t *__restrict__ keys,
                                                      const uint8_t *__restrict__ leaflevel, const uint32_t *__restrict__ cnt,
                                                      const uint32_t *__restrict__ base, double box, NodeGeo *__restrict__ geo,
                                                      NodeLink *__restrict__ link)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= nnodes)
        return;
    const int64_t i = head[j];
    const uint64_t ki = keys[i];
    const int L = leaflevel[i];
    const int first = L - (int)cnt[i] + 1;
    const int l = first + (int)(j - (int64_t)base[i]); // this node's level
    double cx = box / 2., cy = box / 2., cz = box / 2.;
    double len = box * 1.001;
    for(int a = 0; a < l; a++) {
        const int d = (int)((ki >> (3 * (MAXLEVEL - 1 - a))) & 7);
        const double q = 0.25 * len;
        cx += (d & 1) ? q : -q;
        cy += (d & 2) ? q : -q;
        cz += (d & 4) ? q : -q;
        len *= 0.5;
    }
    geo[j] = NodeGeo{cx, cy, cz, len};
    int pc = 1;
    while(i + pc < n && cnt[i + pc] == 0)
        pc++;
    NodeLink lk;
    lk.level = l;
    lk.pstart = (int)i;
    int64_t e;
    if(l == L) {
        lk.pcount = pc;
        e = i + pc;
    }
    else {
        lk.pcount = 0;
        const int shift = 3 * (MAXLEVEL - l);
        const uint64_t pref = (l == 0) ? 0 : (ki >> shift);
        int64_t lo = i + pc, hi = n;
        if(l > 0)
            for(int64_t step = 8;; step <<= 1) {
                const int64_t p = lo - 1 + step;
                if(p >= n)
                    break;
                if((keys[p] >> shift) == pref)
                    lo = p + 1;
                else {
                    hi = p;
                    break;
                }
            }
        else
            lo = n;
        while(lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if((keys[mid] >> shift) == pref)
                lo = mid + 1;
            else
                hi = mid;
        }
        e = lo;
    }
    lk.sibling = (e < n) ? (int)base[e] : -1;
    link[j] = lk;
}

// Leaf moments: add_particle_moment_to_node + force_update_particle_node (forcetree.c:947-966, :985-1004).
// hsml/hact are in TREE order; hmax only counts gas/BH particles that are NOT hydro-active.
__global__ void __launch_bounds__(256) k_leaf_moments(int64_t nnodes, int64_t npart, const NodeLink *__restrict__ link,
                                                      const NodeGeo *__restrict__ geo, Src4 *__restrict__ src,
                                                      const double *__restrict__ hsml_gasbh, double *__restrict__ hmax)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= nnodes)
        return;
    const NodeLink lk = link[j];
    if(lk.pcount == 0)
        return;
    double m = 0, sx = 0, sy = 0, sz = 0, hm = 0;
    const NodeGeo g = geo[j];
    for(int k = 0; k < lk.pcount; k++) {
        const Src4 p = src[lk.pstart + k];
        m += p.m;
        sx += p.m * p.x;
        sy += p.m * p.y;
        sz += p.m * p.z;
        if(hsml_gasbh) {
            const double h = hsml_gasbh[lk.pstart + k]; // < 0: particle does not contribute (not gas/BH, or active)
            if(h >= 0) {
                hm = fmax(hm, fabs(p.x - g.cx) + h - g.len / 2.);
                hm = fmax(hm, fabs(p.y - g.cy) + h - g.len / 2.);
                hm = fmax(hm, fabs(p.z - g.cz) + h - g.len / 2.);
            }
        }
    }
    Src4 o;
    if(m > 0) {
        o.x = sx / m;
        o.y = sy / m;
        o.z = sz / m;
    }
    else {
        o.x = g.cx;
        o.y = g.cy;
        o.z = g.cz;
    }
    o.m = m;
    src[npart + j] = o;
    if(hmax)
        hmax[j] = hm;
}

// Internal moments of one level: force_update_node_recursive (forcetree.c:1081-1101): children in octant
// order, mass-weighted centre of mass of the children's centres of mass, max of hmax.
__global__ void __launch_bounds__(256) k_internal_moments(int64_t nnodes, int64_t npart, int level, const NodeLink *__restrict__ link,
                                                          Src4 *__restrict__ src, double *__restrict__ hmax)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= nnodes)
        return;
    const NodeLink lk = link[j];
    if(lk.pcount != 0 || lk.level != level)
        return;
    double m = 0, sx = 0, sy = 0, sz = 0, hm = 0;
    int c = (int)j + 1;
    do {
        const Src4 s = src[npart + c];
        m += s.m;
        sx += s.m * s.x;
        sy += s.m * s.y;
        sz += s.m * s.z;
        if(hmax)
            hm = fmax(hm, hmax[c]);
        c = link[c].sibling;
    } while(c != lk.sibling);
    Src4 o;
    o.m = m;
    if(m > 0) {
        o.x = sx / m;
        o.y = sy / m;
        o.z = sz / m;
    }
    else {
        o.x = sx;
        o.y = sy;
        o.z = sz;
    }
    src[npart + j] = o;
    if(hmax)
        hmax[j] = hm;
}

__global__ void __launch_bounds__(256) k_only_hmax_leaf(int64_t nnodes, const NodeLink *__restrict__ link, const NodeGeo *__restrict__ geo,
                                                        const Src4 *__restrict__ src, const double *__restrict__ hsml_gasbh,
                                                        double *__restrict__ hmax)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= nnodes)
        return;
    const NodeLink lk = link[j];
    if(lk.pcount == 0)
        return;
    const NodeGeo g = geo[j];
    double hm = 0;
    for(int k = 0; k < lk.pcount; k++) {
        const double h = hsml_gasbh[lk.pstart + k];
        if(h >= 0) {
            const Src4 p = src[lk.pstart + k];
            hm = fmax(hm, fabs(p.x - g.cx) + h - g.len / 2.);
            hm = fmax(hm, fabs(p.y - g.cy) + h - g.len / 2.);
            hm = fmax(hm, fabs(p.z - g.cz) + h - g.len / 2.);
        }
    }
    hmax[j] = hm;
}

__global__ void __launch_bounds__(256) k_only_hmax_internal(int64_t nnodes, int level, const NodeLink *__restrict__ link,
                                                            double *__restrict__ hmax)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= nnodes)
        return;
    const NodeLink lk = link[j];
    if(lk.pcount != 0 || lk.level != level)
        return;
    double hm = 0;
    int c = (int)j + 1;
    do {
        hm = fmax(hm, hmax[c]);
        c = link[c].sibling;
    } while(c != lk.sibling);
    hmax[j] = hm;
}

__global__ void __launch_bounds__(256) k_node_levels(int64_t nnodes, const NodeLink *__restrict__ link, uint32_t *__restrict__ lvl,
                                                     uint32_t *__restrict__ nid)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= nnodes)
        return;
    lvl[j] = (uint32_t)link[j].level;
    nid[j] = (uint32_t)j;
}

__global__ void __launch_bounds__(256) k_invert_perm(int64_t n, const uint32_t *__restrict__ dfs_of_bfs, uint32_t *__restrict__ bfs_of_dfs)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    bfs_of_dfs[dfs_of_bfs[i]] = (uint32_t)i;
}

// level-ordered copy: a stable sort of the depth-first nodes by level keeps key order inside a level, so the children of
// a node (consecutive keys one level down) become contiguous.  Children are counted along the sibling chain.
__global__ void __launch_bounds__(256) k_build_level_order(int64_t nnodes, int64_t npart, const uint32_t *__restrict__ dfs_of_bfs,
                                                           const uint32_t *__restrict__ bfs_of_dfs, const NodeGeo *__restrict__ geo,
                                                           const NodeLink *__restrict__ link, const Src4 *__restrict__ src,
                                                           const double *__restrict__ hmax, NodeGeo *__restrict__ geoB,
                                                           Src4 *__restrict__ momB, NodeLinkB *__restrict__ linkB, double *__restrict__ hmaxB)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= nnodes)
        return;
    const int64_t j = dfs_of_bfs[i];
    const NodeLink lk = link[j];
    geoB[i] = geo[j];
    momB[i] = src[npart + j];
    NodeLinkB o;
    o.pstart = lk.pstart;
    o.pcount = lk.pcount;
    o.firstchild = -1;
    o.nchild = 0;
    if(lk.pcount == 0 && j + 1 < nnodes) {
        o.firstchild = (int)bfs_of_dfs[j + 1];
        int c = (int)j + 1, n = 0;
        int pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // particles of child n if it is a leaf
        do {
            pc[n] = link[c].pcount;
            n++;
            c = link[c].sibling;
        } while(c != lk.sibling && n < 8);
        o.nchild = n;
        // the merge hints of the leaves among the children (NodeLinkB::firstchild of a leaf): sets of sibling leaves that fit one list
        // entry of the neighbour search.  (A leaf's own thread below writes its other three words only.)
        auto run = [&](const int c0, const int c1) -> unsigned { // children c0 .. c1-1: their particle count if they can be joined
            int sum = 0, have = 0;
            for(int k = c0; k < c1 && k < n; k++) {
                if(pc[k] <= 0)
                    return 0u;
                sum += pc[k];
                have++;
            }
            return (have >= 2 && sum <= 8) ? (unsigned)sum : 0u;
        };
        const unsigned oct = run(0, 8);
        for(int k = 0; k < n; k++)
            if(pc[k] > 0)
                linkB[o.firstchild + k].firstchild = (int)(run(k & 6, (k & 6) + 2) | (run(k & 4, (k & 4) + 4) << 4) | (oct << 8));
    }
    if(lk.pcount > 0 && i > 0) { // a leaf below the root: its first word is its parent's to write
        linkB[i].nchild = 0;
        linkB[i].pstart = o.pstart;
        linkB[i].pcount = o.pcount;
    }
    else {
        if(lk.pcount > 0)
            o.firstchild = 0; // (the root as a leaf: no siblings)
        linkB[i] = o;
    }
    if(hmaxB)
        hmaxB[i] = hmax ? hmax[j] : 0.0;
}

__global__ void __launch_bounds__(256) k_permute_hmax(int64_t nnodes, const uint32_t *__restrict__ dfs_of_bfs, const double *__restrict__ hmax,
                                                      double *__restrict__ hmaxB)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < nnodes)
        hmaxB[i] = hmax[dfs_of_bfs[i]];
}

// ---- search geometry of the SPH loops: the box around a node's particles, bottom-up; then a cube around the box, in level order
__global__ void __launch_bounds__(256) k_aabb_leaf(int64_t nnodes, const NodeLink *__restrict__ link, const NodeGeo *__restrict__ geo,
                                                   const Src4 *__restrict__ src, double *__restrict__ aabb)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= nnodes)
        return;
    const NodeLink lk = link[j];
    if(lk.pcount == 0 && !(j == 0 && nnodes == 1))
        return;
    double lo[3], hi[3];
    if(lk.pcount == 0) { // the lone root of an empty tree: its cell's centre
        const NodeGeo g = geo[j];
        lo[0] = hi[0] = g.cx;
        lo[1] = hi[1] = g.cy;
        lo[2] = hi[2] = g.cz;
    }
    else {
        const Src4 p0 = src[lk.pstart];
        lo[0] = hi[0] = p0.x;
        lo[1] = hi[1] = p0.y;
        lo[2] = hi[2] = p0.z;
        for(int k = 1; k < lk.pcount; k++) {
            const Src4 p = src[lk.pstart + k];
            lo[0] = fmin(lo[0], p.x);
            hi[0] = fmax(hi[0], p.x);
            lo[1] = fmin(lo[1], p.y);
            hi[1] = fmax(hi[1], p.y);
            lo[2] = fmin(lo[2], p.z);
            hi[2] = fmax(hi[2], p.z);
        }
    }
    for(int a = 0; a < 3; a++) {
        aabb[6 * j + a] = lo[a];
        aabb[6 * j + 3 + a] = hi[a];
    }
}

__global__ void __launch_bounds__(256) k_aabb_internal(int64_t nnodes, int level, const NodeLink *__restrict__ link, double *__restrict__ aabb)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= nnodes)
        return;
    const NodeLink lk = link[j];
    if(lk.pcount != 0 || lk.level != level || j + 1 >= nnodes)
        return;
    double b[6];
    int c = (int)j + 1;
    for(int a = 0; a < 6; a++)
        b[a] = aabb[6 * (int64_t)c + a];
    c = link[c].sibling;
    while(c != lk.sibling) {
        for(int a = 0; a < 3; a++) {
            b[a] = fmin(b[a], aabb[6 * (int64_t)c + a]);
            b[3 + a] = fmax(b[3 + a], aabb[6 * (int64_t)c + 3 + a]);
        }
        c = link[c].sibling;
    }
    for(int a = 0; a < 6; a++)
        aabb[6 * j + a] = b[a];
}

// the cube of the cull test (centre, side) around the box; `pad` (a rounding's worth of the box size) keeps a particle ON the box's
// face inside the cube whatever the rounding of the centre
__global__ void __launch_bounds__(256) k_aabb_cubes(int64_t nnodes, const uint32_t *__restrict__ dfs_of_bfs, const double *__restrict__ aabb, double pad,
                                                    NodeGeo *__restrict__ geoS)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= nnodes)
        return;
    const int64_t j = dfs_of_bfs[i];
    const double *b = aabb + 6 * j;
    NodeGeo g;
    g.cx = 0.5 * (b[0] + b[3]);
    g.cy = 0.5 * (b[1] + b[4]);
    g.cz = 0.5 * (b[2] + b[5]);
    g.len = fmax(fmax(b[3] - b[0], b[4] - b[1]), b[5] - b[2]) + pad;
    geoS[i] = g;
}

__global__ void __launch_bounds__(256) k_hsmax_leaf(int64_t nnodes, const NodeLink *__restrict__ link, const double *__restrict__ hsml, double *__restrict__ hsmax)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= nnodes)
        return;
    const NodeLink lk = link[j];
    if(lk.pcount == 0 && !(j == 0 && nnodes == 1))
        return;
    double hm = 0;
    for(int k = 0; k < lk.pcount; k++)
        hm = fmax(hm, hsml[lk.pstart + k]); // (negative: a particle that is no SPH source)
    hsmax[j] = hm;
}

static inline int nblk(int64_t n, int b = 256) { return (int)((n + b - 1) / b); }

void TreeBuilder::calc_search_boxes(hipStream_t st)
{
    ensure_level_order(st);
    aabb.reserve(6 * (size_t)nnodes + 6);
    geoS.reserve(nnodes + 16);
    hipLaunchKernelGGL(k_aabb_leaf, dim3(nblk(nnodes)), dim3(256), 0, st, nnodes, link.p, geo.p, src.p, aabb.p);
    for(int l = maxlevel - 1; l >= 0; l--)
        hipLaunchKernelGGL(k_aabb_internal, dim3(nblk(nnodes)), dim3(256), 0, st, nnodes, l, link.p, aabb.p);
    hipLaunchKernelGGL(k_aabb_cubes, dim3(nblk(nnodes)), dim3(256), 0, st, nnodes, nid_b.p, aabb.p, 1e-13 * box, geoS.p);
    MPG_HIP(hipGetLastError());
    has_boxes = true;
}

void TreeBuilder::calc_search_hsmax(const double *d_hsml_treeorder, hipStream_t st)
{
    ensure_level_order(st);
    hsmax.reserve(nnodes + 1);
    hsmaxS.reserve(nnodes + 16);
    hipLaunchKernelGGL(k_hsmax_leaf, dim3(nblk(nnodes)), dim3(256), 0, st, nnodes, link.p, d_hsml_treeorder, hsmax.p);
    for(int l = maxlevel - 1; l >= 0; l--)
        hipLaunchKernelGGL(k_only_hmax_internal, dim3(nblk(nnodes)), dim3(256), 0, st, nnodes, l, link.p, hsmax.p);
    hipLaunchKernelGGL(k_permute_hmax, dim3(nblk(nnodes)), dim3(256), 0, st, nnodes, nid_b.p, hsmax.p, hsmaxS.p);
    MPG_HIP(hipGetLastError());
    has_hsmax = true;
}

// linkB with the internal nodes of <= cap particles as leaves of their whole particle range (tree order is depth-first: the particles
// below node j end where those of the node after its subtree begin)
__global__ void __launch_bounds__(256) k_search_links(int64_t nnodes, int64_t npart, int cap, const uint32_t *__restrict__ dfs_of_bfs,
                                                      const NodeLink *__restrict__ link, const NodeLinkB *__restrict__ linkB, NodeLinkB *__restrict__ linkS)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= nnodes)
        return;
    NodeLinkB o = linkB[i];
    if(o.pcount == 0 && o.nchild > 0) {
        const NodeLink lk = link[dfs_of_bfs[i]];
        const int64_t end = lk.sibling >= 0 ? (int64_t)link[lk.sibling].pstart : npart;
        const int64_t total = end - (int64_t)lk.pstart;
        if(total <= cap) {
            o.pstart = lk.pstart;
            o.pcount = (int)total;
            o.nchild = 0;
            o.firstchild = 0; // (no merge hints: its parent's children are not all leaves of <= 8)
        }
    }
    linkS[i] = o;
}

void TreeBuilder::calc_search_links(int cap, hipStream_t st)
{
    ensure_level_order(st);
    linkS.reserve(nnodes + 16);
    hipLaunchKernelGGL(k_search_links, dim3(nblk(nnodes)), dim3(256), 0, st, nnodes, npart, cap, nid_b.p, link.p, linkB.p, linkS.p);
    MPG_HIP(hipGetLastError());
    has_slinks = true;
    slink_cap = cap;
}

// the particles of leaf i (level order) as the 8 records srcL[8 i ..]: one thread per record; a slot beyond the leaf's count holds a
// zero-mass record at the leaf's first particle (any finite position will do: its pair evaluates to exactly zero)
__global__ void __launch_bounds__(256) k_pad_leaves(int64_t nnodes, const NodeLinkB *__restrict__ linkB, const Src4 *__restrict__ src, Src4 *__restrict__ srcL)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t >> 3;
    const int s = (int)(t & 7);
    if(i > nnodes)
        return;
    Src4 r{0, 0, 0, 0};
    if(i < nnodes) {
        const NodeLinkB lk = linkB[i];
        if(lk.pcount <= 0)
            return; // (an internal node's block is never read)
        r = src[lk.pstart + (s < lk.pcount ? s : 0)];
        if(s >= lk.pcount)
            r.m = 0;
    }
    srcL[t] = r;
}

void TreeBuilder::ensure_leaf_pad(hipStream_t st)
{
    ensure_level_order(st);
    if(has_leaf_pad)
        return;
    srcL.reserve(((size_t)nnodes + 1) * 8 + 8);
    hipLaunchKernelGGL(k_pad_leaves, dim3(nblk((nnodes + 1) * 8)), dim3(256), 0, st, nnodes, linkB.p, src.p, srcL.p);
    MPG_HIP(hipGetLastError());
    has_leaf_pad = true;
}

void TreeBuilder::ensure_level_order(hipStream_t st)
{
    if(!has_bfs)
        make_level_order(st);
}

void TreeBuilder::make_level_order(hipStream_t st)
{
    const int64_t M = nnodes;
    lvl_a.reserve(M + 1);
    lvl_b.reserve(M + 1);
    nid_a.reserve(M + 1);
    nid_b.reserve(M + 1);
    bfs_of_dfs.reserve(M + 1);
    geoB.reserve(M + 16);
    momB.reserve(M + 16);
    linkB.reserve(M + 16);
    if(has_hmax)
        hmaxB.reserve(M + 16);
    // zero-mass padding records behind the moments (the evaluation kernel points idle lanes of its node loop at them)
    MPG_HIP(hipMemsetAsync(momB.p + M, 0, 16 * sizeof(Src4), st));
    hipLaunchKernelGGL(k_node_levels, dim3(nblk(M)), dim3(256), 0, st, M, link.p, lvl_a.p, nid_a.p);
    size_t tb = 0;
    MPG_HIP(rocprim::radix_sort_pairs(nullptr, tb, lvl_a.p, lvl_b.p, nid_a.p, nid_b.p, (size_t)M, 0, 5, st));
    tmp.reserve(tb + 16);
    MPG_HIP(rocprim::radix_sort_pairs((void *)tmp.p, tb, lvl_a.p, lvl_b.p, nid_a.p, nid_b.p, (size_t)M, 0, 5, st));
    hipLaunchKernelGGL(k_invert_perm, dim3(nblk(M)), dim3(256), 0, st, M, nid_b.p, bfs_of_dfs.p);
    hipLaunchKernelGGL(k_build_level_order, dim3(nblk(M)), dim3(256), 0, st, M, npart, nid_b.p, bfs_of_dfs.p, geo.p, link.p, src.p,
                       has_hmax ? hmax.p : (const double *)nullptr, geoB.p, momB.p, linkB.p, has_hmax ? hmaxB.p : (double *)nullptr);
    MPG_HIP(hipGetLastError());
    has_bfs = true;
}

void TreeBuilder::build(int64_t n, const double *d_pos, const float *d_mass, const uint8_t *d_type, int mask, double box,
                        hipStream_t st, EventTimer *tm, const uint8_t *d_include)
{
    MPG_CHECK(n < (int64_t)1 << 31, "tree build: more than 2^31 particles on one GPU");
    this->box = box;
    this->ncaller = n;
    has_moments = false;
    has_hmax = false;
    has_bfs = false;
    has_boxes = false;
    has_hsmax = false;
    has_slinks = false;
    has_leaf_pad = false;
    if(tm)
        tm->start(st);
    keys_a.reserve(n + 1);
    keys_b.reserve(n + 1);
    idx_a.reserve(n + 1);
    idx_b.reserve(n + 1);
    flags.reserve(8);
    MPG_HIP(hipMemsetAsync(flags.p, 0, 8 * sizeof(int64_t), st));
    unsigned long long *d_nmemb = (unsigned long long *)(flags.p + 4);
    int *d_flags = (int *)flags.p;
    // A tree of a subset (a type mask, an include list: the gas tree of the SPH loops, the trees of the active particles of the
    // hierarchical gravity levels) first compacts its members' indices and computes and sorts keys for those alone.  (Rounds 1-2 gave the
    // other particles the key ~0 and counted them with one same-address atomic each: 3.2 ms for the keys of a 32 768-particle tree in a
    // 256^3 table, plus the sort of all 2^24 keys.)
    npart = n;
    // (without a type array every particle counts as type 1, as TreeMember does: a mask without that bit selects nothing)
    if(!d_type && (mask & 2) == 0)
        npart = n = 0;
    const bool subset = n > 0 && (d_type != nullptr || d_include != nullptr);
    if(subset) {
        rocprim::counting_iterator<uint32_t> iota(0);
        const TreeMember member{d_type, d_include, mask};
        size_t sb = 0;
        MPG_HIP(rocprim::select(nullptr, sb, iota, idx_a.p, d_nmemb, (size_t)n, member, st));
        tmp.reserve(sb + 16);
        MPG_HIP(rocprim::select((void *)tmp.p, sb, iota, idx_a.p, d_nmemb, (size_t)n, member, st));
        unsigned long long nm = 0;
        MPG_HIP(hipMemcpyAsync(&nm, d_nmemb, sizeof(nm), hipMemcpyDeviceToHost, st));
        MPG_HIP(hipStreamSynchronize(st));
        npart = (int64_t)nm;
    }
    if(npart > 0)
        hipLaunchKernelGGL(k_keys, dim3(nblk(npart)), dim3(256), 0, st, npart, d_pos, subset ? (const uint32_t *)idx_a.p : (const uint32_t *)nullptr, box, keys_a.p,
                           idx_a.p);
    if(tm)
        tm->lap(st, &tm->t.tree_keys);
    // --- sort (stable: particles with equal keys stay in caller order), leaf levels, node numbering.
    // A tree whose leaves all lie at level <= 10 is fully determined by the top 30 bits of the keys: those bits are sorted first (4 radix
    // passes instead of 8) and the leaf levels found are the check - a level-10 cell with more than 8 particles shows up as a leaf level
    // > 10, and the sort is then done again on all bits (a clustered set: 4 passes lost).  Within a leaf of a shallow tree the particles
    // are in caller order instead of key order: the node set and every decision are the same, sums differ by rounding.  The attempt is
    // made for EVERY tree, whatever the builder built before: the particle order is a function of the particle set alone.
    static const bool full_sort_only = getenv("MPG_TREE_FULL_SORT") != nullptr;
    bool short_sort = !full_sort_only;
    constexpr int SHORT_LEVELS = 10;
    leaflevel.reserve(npart + 1);
    cnt.reserve(npart + 1);
    base.reserve(npart + 1);
    int hflags[3] = {0, 0, 0};
    uint32_t lastbase = 0, lastcnt = 0;
    for(;;) {
        if(npart == 0)
            break;
        size_t tmpbytes = 0;
        if(short_sort) {
            // (the top 30 bits as 32-bit keys of their own, sorted with their position: rocprim 4.2's radix_sort_pairs returns wrong
            // results - mismatched pairs - for begin_bit > 0 below a few million elements, measured with tools/sort_bits_check.hip)
            hi_a.reserve(npart + 1);
            hi_b.reserve(npart + 1);
            pos_a.reserve(npart + 1);
            pos_b.reserve(npart + 1);
            hipLaunchKernelGGL(k_key_tops, dim3(nblk(npart)), dim3(256), 0, st, npart, keys_a.p, 3 * (MAXLEVEL - SHORT_LEVELS), hi_a.p, pos_a.p);
            MPG_HIP(rocprim::radix_sort_pairs(nullptr, tmpbytes, hi_a.p, hi_b.p, pos_a.p, pos_b.p, (size_t)npart, 0, 3 * SHORT_LEVELS, st));
            tmp.reserve(tmpbytes + 16);
            MPG_HIP(rocprim::radix_sort_pairs((void *)tmp.p, tmpbytes, hi_a.p, hi_b.p, pos_a.p, pos_b.p, (size_t)npart, 0, 3 * SHORT_LEVELS, st));
            hipLaunchKernelGGL(k_apply_order, dim3(nblk(npart)), dim3(256), 0, st, npart, pos_b.p, keys_a.p, idx_a.p, keys_b.p, idx_b.p);
        }
        else {
            MPG_HIP(rocprim::radix_sort_pairs(nullptr, tmpbytes, keys_a.p, keys_b.p, idx_a.p, idx_b.p, (size_t)npart, 0, 64, st));
            tmp.reserve(tmpbytes + 16);
            MPG_HIP(rocprim::radix_sort_pairs((void *)tmp.p, tmpbytes, keys_a.p, keys_b.p, idx_a.p, idx_b.p, (size_t)npart, 0, 64, st));
        }
        if(tm)
            tm->lap(st, &tm->t.tree_sort);
        const int64_t nwaves = (int64_t)nblk(npart) * 4;
        wave_ext.reserve((size_t)nwaves);
        hipLaunchKernelGGL(k_leaflevel, dim3(nblk(npart)), dim3(256), 0, st, npart, keys_b.p, leaflevel.p, cnt.p, d_flags, wave_ext.p, force_internal_above);
        hipLaunchKernelGGL(k_level_extrema, dim3(64), dim3(256), 0, st, nwaves, wave_ext.p, d_flags);
        size_t sb = 0;
        MPG_HIP(rocprim::exclusive_scan(nullptr, sb, cnt.p, base.p, 0u, (size_t)npart, rocprim::plus<uint32_t>(), st));
        tmp.reserve(sb + 16);
        MPG_HIP(rocprim::exclusive_scan((void *)tmp.p, sb, cnt.p, base.p, 0u, (size_t)npart, rocprim::plus<uint32_t>(), st));
        MPG_HIP(hipMemcpyAsync(hflags, d_flags, sizeof(hflags), hipMemcpyDeviceToHost, st));
        MPG_HIP(hipMemcpyAsync(&lastbase, base.p + (npart - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        MPG_HIP(hipMemcpyAsync(&lastcnt, cnt.p + (npart - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        MPG_HIP(hipStreamSynchronize(st));
        if(short_sort && (hflags[1] > SHORT_LEVELS || hflags[0] != 0)) { // deeper than the bits sorted: all bits
            short_sort = false;
            MPG_HIP(hipMemsetAsync(d_flags, 0, 4 * sizeof(int), st));
            continue;
        }
        break;
    }
    if(npart > 0) {
        MPG_CHECK(hflags[0] == 0, "tree build: more than 8 particles share one 2^-21 cell (coincident particles; the reference "
                                  "aborts here too, forcetree.c:393-412)");
        nnodes = (int64_t)lastbase + lastcnt;
        maxlevel = hflags[1];
        minleaflevel = MAXLEVEL - hflags[2];
    }
    else {
        // empty tree: a lone root leaf with no particles (forcetree.c:657-678)
        nnodes = 1;
        maxlevel = 0;
        minleaflevel = 0;
    }
    // padding records behind each array: the cooperative walk reads nodes no .. no+7 (those past the end are ignored)
    src.reserve(npart + nnodes + 16);
    geo.reserve(nnodes + 16);
    link.reserve(nnodes + 16);
    // the padding records of the source array are zero-mass sources at the origin (grav_walk_split.hip points idle lanes at them)
    MPG_HIP(hipMemsetAsync(src.p + npart + nnodes, 0, 16 * sizeof(Src4), st));
    if(npart > 0) {
        hipLaunchKernelGGL(k_gather_src, dim3(nblk(npart)), dim3(256), 0, st, npart, idx_b.p, d_pos, d_mass, src.p);
        static const bool per_particle = getenv("MPG_TREE_FILL_PER_PARTICLE") != nullptr; // (rounds 1-5: one thread per particle)
        if(per_particle)
            hipLaunchKernelGGL(k_fill_nodes, dim3(nblk(npart)), dim3(256), 0, st, npart, keys_b.p, leaflevel.p, cnt.p, base.p, box, geo.p, link.p);
        else {
            node_head.reserve((size_t)nnodes + 1);
            hipLaunchKernelGGL(k_node_heads, dim3(nblk(npart)), dim3(256), 0, st, npart, cnt.p, base.p, node_head.p);
            hipLaunchKernelGGL(k_fill_nodes_n, dim3(nblk(nnodes)), dim3(256), 0, st, nnodes, npart, node_head.p, keys_b.p, leaflevel.p, cnt.p, base.p,
                               box, geo.p, link.p);
        }
    }
    else {
        NodeGeo g{box / 2., box / 2., box / 2., box * 1.001};
        NodeLink l{-1, 0, 0, 0};
        Src4 s{box / 2., box / 2., box / 2., 0.0};
        MPG_HIP(hipMemcpyAsync(geo.p, &g, sizeof(g), hipMemcpyHostToDevice, st));
        MPG_HIP(hipMemcpyAsync(link.p, &l, sizeof(l), hipMemcpyHostToDevice, st));
        MPG_HIP(hipMemcpyAsync(src.p, &s, sizeof(s), hipMemcpyHostToDevice, st));
        MPG_HIP(hipStreamSynchronize(st));
    }
    if(tm)
        tm->lap(st, &tm->t.tree_nodes);
}

void TreeBuilder::calc_moments(const double *d_hsml_gasbh_treeorder, hipStream_t st, EventTimer *tm)
{
    if(tm)
        tm->start(st);
    double *hm = nullptr;
    if(d_hsml_gasbh_treeorder) {
        hmax.reserve(nnodes + 1);
        hm = hmax.p;
    }
    if(npart > 0) {
        hipLaunchKernelGGL(k_leaf_moments, dim3(nblk(nnodes)), dim3(256), 0, st, nnodes, npart, link.p, geo.p, src.p,
                           d_hsml_gasbh_treeorder, hm);
        for(int l = maxlevel - 1; l >= 0; l--)
            hipLaunchKernelGGL(k_internal_moments, dim3(nblk(nnodes)), dim3(256), 0, st, nnodes, npart, l, link.p, src.p, hm);
    }
    has_moments = true;
    has_hmax = hm != nullptr;
    has_bfs = false; // the level-ordered copy is made on demand (ensure_level_order) by the kernels that walk it
    if(tm)
        tm->lap(st, &tm->t.tree_moments);
}

void TreeBuilder::calc_hmax(const double *d_hsml_gasbh_treeorder, hipStream_t st)
{
    hmax.reserve(nnodes + 1);
    if(npart > 0) {
        hipLaunchKernelGGL(k_only_hmax_leaf, dim3(nblk(nnodes)), dim3(256), 0, st, nnodes, link.p, geo.p, src.p, d_hsml_gasbh_treeorder,
                           hmax.p);
        for(int l = maxlevel - 1; l >= 0; l--)
            hipLaunchKernelGGL(k_only_hmax_internal, dim3(nblk(nnodes)), dim3(256), 0, st, nnodes, l, link.p, hmax.p);
    }
    else
        MPG_HIP(hipMemsetAsync(hmax.p, 0, sizeof(double), st));
    has_hmax = true;
    if(has_bfs) { // keep the level-ordered copy in step
        hmaxB.reserve(nnodes + 16);
        hipLaunchKernelGGL(k_permute_hmax, dim3(nblk(nnodes)), dim3(256), 0, st, nnodes, nid_b.p, hmax.p, hmaxB.p);
    }
}

// ---- domain-decomposed runs: the top of the tree from global sums ------------------------------------
// With particles distributed over ranks, the local tree holds the rank's own particles plus ghosts in whole columns of
// level-La cells, so cells at levels >= La are complete and equal to the global tree's; cells above (levels < La) also
// hold remote particles.  Their moments come from sums over ALL ranks: top_partial() adds (m, m x, m y, m z) of the rank's
// OWN particles (caller index < n_own) into the cells of level La-1; the caller all-reduces and builds the coarser levels;
// top_set() writes the moments of every local node above level La from those sums.
__global__ void __launch_bounds__(256) k_top_partial(int64_t npart, const uint64_t *__restrict__ keys, const uint32_t *__restrict__ order,
                                                     const Src4 *__restrict__ src, int64_t n_own, int shift, double *__restrict__ out)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool own = k < npart && (int64_t)order[k] < n_own;
    const unsigned long long cell = own ? (unsigned long long)(keys[k] >> shift) : ~0ull;
    Src4 p{0, 0, 0, 0};
    if(own)
        p = src[k];
    double v[4] = {p.m, p.m * p.x, p.m * p.y, p.m * p.z};
    // sorted keys: the lanes of a wave mostly share one cell -> one atomic per wave and component (same-address atomics
    // serialise at ~90 per microsecond); mixed waves fall back to per-lane atomics
    const unsigned long long c0 = __shfl(cell, 0);
    if(__ballot(cell != c0) == 0) {
        if(c0 == ~0ull)
            return;
#pragma unroll
        for(int j = 0; j < 4; j++) {
            double t = v[j];
            for(int off = 32; off > 0; off >>= 1)
                t += __shfl_down(t, off);
            if((threadIdx.x & 63) == 0)
                unsafeAtomicAdd(&out[4 * c0 + j], t);
        }
    }
    else if(own) {
#pragma unroll
        for(int j = 0; j < 4; j++)
            unsafeAtomicAdd(&out[4 * cell + j], v[j]);
    }
}

// sums[]: levels 0 .. La-1 concatenated (level l starts at (8^l - 1) / 7 cells), 4 doubles per cell
__global__ void __launch_bounds__(256) k_top_set(int64_t nnodes, int64_t npart, int La, const NodeLink *__restrict__ link,
                                                 const uint64_t *__restrict__ keys, const double *__restrict__ sums, Src4 *__restrict__ src,
                                                 int *__restrict__ flag)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= nnodes)
        return;
    const NodeLink lk = link[j];
    if(lk.level >= La)
        return;
    if(lk.pcount > 0) { // a leaf above the decomposition level: its particle set may be incomplete on this rank
        atomicExch(flag, 1);
        return;
    }
    const unsigned long long cell = lk.level == 0 ? 0ull : (unsigned long long)(keys[lk.pstart] >> (3 * (MAXLEVEL - lk.level)));
    unsigned long long base = 0, p8 = 1;
    for(int l = 0; l < lk.level; l++) {
        base += p8;
        p8 *= 8;
    }
    const double *q = sums + 4 * (base + cell);
    Src4 o;
    o.m = q[0];
    o.x = q[1] / q[0];
    o.y = q[2] / q[0];
    o.z = q[3] / q[0];
    src[npart + j] = o;
}

void TreeBuilder::top_partial(int La, int64_t n_own, double *d_out, hipStream_t st)
{
    MPG_CHECK(La >= 1 && La <= 8, "decomposition level must be in [1, 8]");
    const size_t ncell = (size_t)1 << (3 * (La - 1));
    MPG_HIP(hipMemsetAsync(d_out, 0, ncell * 4 * sizeof(double), st));
    if(npart > 0)
        hipLaunchKernelGGL(k_top_partial, dim3(nblk(npart)), dim3(256), 0, st, npart, keys_b.p, idx_b.p, src.p, n_own, 3 * (MAXLEVEL - (La - 1)),
                           d_out);
    MPG_HIP(hipGetLastError());
}

void TreeBuilder::top_set(int La, const double *d_sums, hipStream_t st, int *d_flag_later)
{
    MPG_CHECK(La >= 1 && La <= 8 && has_moments, "top_set: needs a tree with moments and a level in [1, 8]");
    if(d_flag_later) { // the caller reads the flag (already zeroed) with its next read-back: nothing waits here
        hipLaunchKernelGGL(k_top_set, dim3(nblk(nnodes)), dim3(256), 0, st, nnodes, npart, La, link.p, keys_b.p, d_sums, src.p, d_flag_later);
        MPG_HIP(hipGetLastError());
        has_bfs = false;
        return;
    }
    int *d_flag = (int *)flags.p;
    MPG_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_top_set, dim3(nblk(nnodes)), dim3(256), 0, st, nnodes, npart, La, link.p, keys_b.p, d_sums, src.p, d_flag);
    int f = 0;
    MPG_HIP(hipMemcpyAsync(&f, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
    MPG_HIP(hipStreamSynchronize(st));
    MPG_CHECK(f == 0, "domain decomposition: a cell above the decomposition level holds <= 8 local particles (use a coarser level)");
    has_bfs = false; // the level-ordered copy must pick up the new moments
}

TreeView TreeBuilder::view() const
{
    TreeView v;
    v.npart = npart;
    v.nnodes = nnodes;
    v.src = src.p;
    v.geo = geo.p;
    v.link = link.p;
    v.hmax = has_hmax ? hmax.p : nullptr;
    v.order = (const int *)idx_b.p;
    if(has_bfs) {
        v.geoB = geoB.p;
        v.momB = momB.p;
        v.linkB = linkB.p;
        v.hmaxB = has_hmax ? hmaxB.p : nullptr;
        v.geoS = has_boxes ? geoS.p : nullptr;
        v.hsmaxS = (has_boxes && has_hsmax) ? hsmaxS.p : nullptr;
        v.linkS = has_slinks ? linkS.p : nullptr;
        v.srcL = has_leaf_pad ? srcL.p : nullptr;
    }
    v.box = box;
    return v;
}

} // namespace mpg
