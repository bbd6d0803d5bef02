// fof.hip -- friends-of-friends group finder on the device (libgadget/fof.c).
//
//   fof_label_primary    fof.c:366-478, 480-579   particles of the primary link types closer than the linking length belong to one
//                                                 group; a group's label is the smallest particle ID in it (MinID)
//   fof_label_secondary  fof.c:1175-1327          a particle of a secondary link type takes the label of the NEAREST primary particle
//                                                 found within the last radius of the reference's doubling search (0.4 LL or half its
//                                                 Hsml, doubled while < 4 LL); otherwise it stays a group of its own
//   fof_compile_base / fof_assign_grnr / fof_compile_catalogue / add_particle_to_group / fof_finish_group_properties
//                        fof.c:631-755, 758-812, 874-903, 1106-1155     groups of at least FOFHaloMinLength members, numbered by
//                                                 (Length descending, MinID), with Length, LenType, Mass, MassType, CM, Vel, Imom, Jmom
//
// The reference labels groups by iterating a locked union over tree-walk neighbours until no MinID changes; the partition it
// converges to is the set of connected components of the "closer than LL" graph, which is what is computed here directly:
//   k_fof_link      the group-cooperative neighbour search of ngb_walk.h (8 lanes per target) over the tree of the primary types;
//                   every pair (q < j, r^2 <= LL^2, treewalk.c:984-991) is united in a lock-free union-find over tree slots
//                   (hook the larger root under the smaller with atomicCAS; parents are read at device scope)
//   k_fof_flatten / k_fof_minid / k_fof_labels     roots, the smallest ID per root, the label of every particle
//   k_fof_secondary the same neighbour search around each secondary particle, keeping the nearest primary
//   catalogue       radix sort of (label, particle), run lengths = group lengths, the groups long enough are selected in label
//                   order, ranked by a stable sort on length, and their properties are summed with wave-aggregated atomics
// Not carried: the sub-grid group properties (star formation rate, metals, black-hole masses, MaxDens / seeding), ghosts of
// other ranks (a group is complete on one GPU).
#include "fof.h"
#include "ngb_walk.h"
#include "timestep.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

#ifndef FOF_MERGE
#define FOF_MERGE true // sibling leaves that a target opens together as one list entry (walk_stepk<MERGE>, ngb_walk.h)
#endif

namespace mpg {

namespace {

__device__ __forceinline__ int ld_parent(const int *parent, int x) { return __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int find_root(const int *parent, int x)
{
    int p = ld_parent(parent, x);
    while(p != x) {
        x = p;
        p = ld_parent(parent, x);
    }
    return x;
}

// lock-free union: the larger root is hooked under the smaller one; a failed CAS means another lane re-rooted it meanwhile
__device__ __forceinline__ void unite(int *parent, int a, int b)
{
    for(;;) {
        a = find_root(parent, a);
        b = find_root(parent, b);
        if(a == b)
            return;
        if(a > b) {
            const int t = a;
            a = b;
            b = t;
        }
        if(atomicCAS(&parent[b], b, a) == b)
            return;
    }
}

__global__ void __launch_bounds__(256) k_fof_init(int64_t np, int *__restrict__ parent, unsigned long long *__restrict__ minid)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(q < np) {
        parent[q] = (int)q;
        minid[q] = ~0ull;
    }
}

// MODE 0: link the primary particles (targets = tree slots).  MODE 1: nearest primary of each listed particle.
template <int MODE>
__global__ void __launch_bounds__(256) k_fof_walk(const TreeView tv, const double LL, int *__restrict__ parent, const int *__restrict__ list,
                                                  const int64_t ntargets, const double *__restrict__ pos, const double *__restrict__ hsml,
                                                  const uint8_t *__restrict__ type, const int *__restrict__ root_of,
                                                  const unsigned long long *__restrict__ minid, unsigned long long *__restrict__ label,
                                                  unsigned *__restrict__ err)
{
    __shared__ unsigned s_stack[4 * 8 * SPH_STK];
    __shared__ unsigned s_llist[4 * 8 * SPH_LCAP];
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 3, s = lane & 7, gshift = grp * 8;
    unsigned *stack = s_stack + ((threadIdx.x >> 6) * 8 + grp) * SPH_STK;
    unsigned *llist = s_llist + ((threadIdx.x >> 6) * 8 + grp) * SPH_LCAP;
    const int64_t q = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + grp;
    const bool valid = q < ntargets;
    int ci = 0;
    double px = 0, py = 0, pz = 0, radius = LL;
    if(valid) {
        if(MODE == 0) {
            const Src4 me = tv.src[q];
            px = me.x;
            py = me.y;
            pz = me.z;
        }
        else {
            ci = list[q];
            px = pos[3 * (int64_t)ci];
            py = pos[3 * (int64_t)ci + 1];
            pz = pos[3 * (int64_t)ci + 2];
            // the last radius of the doubling search (fof.c:1228-1250, 1285-1293): float arithmetic as there
            float h = (float)(0.4 * LL);
            if(hsml && type && (type[ci] == 0 || type[ci] == 4 || type[ci] == 5) && (double)h < 0.5 * hsml[ci])
                h = (float)(0.5 * hsml[ci]);
            while((double)h < 4 * LL)
                h *= 2.0f;
            radius = (double)h;
        }
    }
    const double h2 = radius * radius;
    double best_r2 = 1e300;
    int best_j = -1;
    int sp = 0;
    if(valid) {
        if(s == 0)
            stack[0] = (0u << 4) | 1u; // the root
        sp = 1;
    }
    bool overflow = false;
    for(;;) {
        int nl = 0;
        for(;;) { // phase A: walk; opened leaves go to the group's list
            const bool go = sp > 0 && nl + 16 <= SPH_LCAP;
            if(ballot64(go) == 0)
                break;
            nl = walk_stepk<false, 2, FOF_MERGE>(tv, tv.geoB, nullptr, stack, sp, go, s, gshift, radius, px, py, pz, llist, nl, overflow); // (two child ranges per step, sibling leaves joined: ngb_walk.h)
            if(ballot64(overflow) != 0)
                break;
        }
        if(ballot64(overflow) != 0)
            break;
        // phase B: every group takes its next leaf; lane s <-> particle s.  (The particle of the NEXT leaf is requested before this one
        // is tested, as in the SPH loops: sph.hip.  Lanes beyond a leaf's count read its first particle.)
        unsigned e = (0 < nl) ? llist[0] : 0u;
        int ps = (int)(e >> 4), pc = (int)(e & 15u);
        Src4 o = tv.src[ps + (s < pc ? s : 0)];
        for(int it = 0;; it++) {
            const bool has = it < nl;
            if(ballot64(has) == 0)
                break;
            const unsigned e_n = (it + 1 < nl) ? llist[it + 1] : 0u;
            const int ps_n = (int)(e_n >> 4), pc_n = (int)(e_n & 15u);
            const Src4 o_n = tv.src[ps_n + (s < pc_n ? s : 0)];
            if(s < pc) {
                const int j = ps + s;
                const double d0 = nearest_img(px - o.x, tv.box, 1.0 / tv.box);
                const double d1 = nearest_img(py - o.y, tv.box, 1.0 / tv.box);
                const double d2 = nearest_img(pz - o.z, tv.box, 1.0 / tv.box);
                const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
                if(r2 <= h2) { // treewalk.c:984-991
                    if(MODE == 0) {
                        if(j > (int)q) // each pair once (fof.c:559: target <= other)
                            unite(parent, (int)q, j);
                    }
                    else if(r2 < best_r2) {
                        best_r2 = r2;
                        best_j = j;
                    }
                }
            }
            o = o_n;
            ps = ps_n;
            pc = pc_n;
        }
        if(ballot64(sp > 0) == 0)
            break;
    }
    if(ballot64(overflow) != 0) {
        if(lane == 0)
            atomicExch(err, 1u);
        return;
    }
    if(MODE == 1) {
        for(int off = 1; off < 8; off <<= 1) { // nearest over the 8 lanes of the group (ties: the lower tree slot)
            const double r2o = __shfl_xor(best_r2, off);
            const int jo = __shfl_xor(best_j, off);
            if(r2o < best_r2 || (r2o == best_r2 && jo >= 0 && (best_j < 0 || jo < best_j))) {
                best_r2 = r2o;
                best_j = jo;
            }
        }
        if(valid && s == 0 && best_j >= 0)
            label[ci] = minid[root_of[best_j]];
    }
}

__global__ void __launch_bounds__(256) k_fof_flatten(int64_t np, const int *__restrict__ parent, int *__restrict__ root_of,
                                                     const int *__restrict__ order, const unsigned long long *__restrict__ id,
                                                     unsigned long long *__restrict__ minid)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(q >= np)
        return;
    const int r = find_root(parent, (int)q);
    root_of[q] = r;
    atomicMin(&minid[r], id[order[q]]);
}

__global__ void __launch_bounds__(256) k_fof_own_labels(int64_t n, const unsigned long long *__restrict__ id, unsigned long long *__restrict__ label)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        label[i] = id[i]; // HaloLabel[i].MinID = P[i].ID (fof.c:409)
}

__global__ void __launch_bounds__(256) k_fof_primary_labels(int64_t np, const int *__restrict__ root_of, const int *__restrict__ order,
                                                            const unsigned long long *__restrict__ minid, unsigned long long *__restrict__ label)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(q < np)
        label[order[q]] = minid[root_of[q]];
}

__global__ void __launch_bounds__(256) k_fof_secondary_flags(int64_t n, const uint8_t *__restrict__ type, const uint8_t *__restrict__ flags,
                                                             int mask, int *__restrict__ value, uint8_t *__restrict__ keep)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    const int t = type ? type[i] : 1;
    value[i] = (int)i;
    keep[i] = (!(flags && (flags[i] & 3)) && ((1 << t) & mask)) ? 1 : 0; // fof_secondary_haswork, fof.c:1182-1189
}

__global__ void __launch_bounds__(256) k_iota(int64_t n, int *__restrict__ idx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        idx[i] = (int)i;
}

__global__ void __launch_bounds__(256) k_fof_keep_runs(int64_t nruns, const unsigned *__restrict__ counts, int minlen, uint8_t *__restrict__ keep,
                                                       const unsigned long long *__restrict__ run_label, const unsigned long long *__restrict__ also,
                                                       int64_t nalso)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(r >= nruns)
        return;
    bool k = counts[r] >= (unsigned)minlen; // fof.c:800-808
    if(!k && nalso > 0) { // a part of a group that continues on another rank: reported whatever its size (binary search of the label)
        const unsigned long long l = run_label[r];
        int64_t lo = 0, hi = nalso;
        while(lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if(also[mid] < l)
                lo = mid + 1;
            else
                hi = mid;
        }
        k = lo < nalso && also[lo] == l;
    }
    keep[r] = k ? 1 : 0;
}

__global__ void __launch_bounds__(256) k_fof_lenkeys(int64_t ng, const unsigned *__restrict__ length, unsigned *__restrict__ key)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(g < ng)
        key[g] = 0xffffffffu - length[g]; // (Length descending; the stable sort keeps MinID ascending within a length, fof.c:1495-1501)
}

__global__ void __launch_bounds__(256) k_fof_grnr(int64_t ng, const int *__restrict__ rank_order, int *__restrict__ grnr)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(r < ng)
        grnr[rank_order[r]] = (int)(r + 1); // fof.c:1127-1134: numbers start at 1
}

__global__ void __launch_bounds__(256) k_fof_firstpos(int64_t ng, const unsigned *__restrict__ start, const int *__restrict__ sidx,
                                                      const double *__restrict__ pos, float *__restrict__ firstpos)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(g >= ng)
        return;
    const int i = sidx[start[g]];
    for(int d = 0; d < 3; d++)
        firstpos[3 * g + d] = (float)pos[3 * (int64_t)i + d]; // BaseGroup.FirstPos is a float (fof.h:25, fof.c:773-775)
}

constexpr int NQ = 27; // Mass, MassType[6], CM[3], Vel[3], Jmom[3], Imom[9] + 2 spare

// add_particle_to_group (fof.c:631-703) for the particles in label order; P[].GrNr (fof.c:218-236)
__global__ void __launch_bounds__(256) k_fof_accumulate(int64_t n, int64_t ng, const unsigned *__restrict__ start, const unsigned *__restrict__ length,
                                                        const int *__restrict__ grnr, const int *__restrict__ sidx, const double *__restrict__ pos,
                                                        const double *__restrict__ vel, const float *__restrict__ mass,
                                                        const uint8_t *__restrict__ type, const float *__restrict__ firstpos, double box,
                                                        long long *__restrict__ p_grnr, double *__restrict__ acc, int *__restrict__ lentype)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int g = -1;
    int i = 0;
    if(k < n) {
        i = sidx[k];
        // the last group whose run starts at or before k
        int lo = 0, hi = (int)ng;
        while(lo < hi) {
            const int mid = (lo + hi) >> 1;
            if(start[mid] <= (unsigned)k)
                lo = mid + 1;
            else
                hi = mid;
        }
        g = lo - 1;
        if(g >= 0 && (unsigned)k >= start[g] + length[g])
            g = -1;
        p_grnr[i] = g >= 0 ? (long long)grnr[g] : -1ll;
    }
    double v[NQ];
    for(int c = 0; c < NQ; c++)
        v[c] = 0;
    int t = 0;
    if(g >= 0) {
        const double m = (double)mass[i];
        t = type ? type[i] : 1;
        double rel[3], xyz[3], vv[3];
        for(int d = 0; d < 3; d++) {
            const double first = (double)firstpos[3 * g + d];
            double x = pos[3 * (int64_t)i + d] - first; // NEAREST, partmanager.h:99
            if(x > 0.5 * box)
                x -= box;
            else if(x < -0.5 * box)
                x += box;
            rel[d] = x;
            xyz[d] = x + first;
            vv[d] = vel ? vel[3 * (int64_t)i + d] : 0.0;
        }
        const double jm[3] = {rel[1] * vv[2] - rel[2] * vv[1], rel[2] * vv[0] - rel[0] * vv[2], rel[0] * vv[1] - rel[1] * vv[0]};
        v[0] = m;
        v[1 + t] = m;
        for(int d = 0; d < 3; d++) {
            v[7 + d] = m * xyz[d];
            v[10 + d] = m * vv[d];
            v[13 + d] = m * jm[d];
            for(int e = 0; e < 3; e++)
                v[16 + 3 * d + e] = m * rel[d] * rel[e];
        }
    }
    // lanes of one wave mostly belong to one group (label order): sum over the wave, one atomic per quantity
    const int g0 = __builtin_amdgcn_readfirstlane(g);
    const bool uniform = __builtin_amdgcn_ballot_w64(g != g0) == 0;
    if(uniform) {
        if(g0 < 0)
            return;
        for(int c = 0; c < 25; c++) {
            double x = v[c];
            for(int off = 32; off > 0; off >>= 1)
                x += __shfl_xor(x, off);
            if((threadIdx.x & 63) == 0 && x != 0.0)
                unsafeAtomicAdd(&acc[(size_t)g0 * NQ + c], x);
        }
        for(int tt = 0; tt < 6; tt++) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(t == tt);
            if((threadIdx.x & 63) == 0 && m)
                atomicAdd(&lentype[(size_t)g0 * 6 + tt], (int)__popcll(m));
        }
    }
    else if(g >= 0) {
        for(int c = 0; c < 25; c++)
            if(v[c] != 0.0)
                unsafeAtomicAdd(&acc[(size_t)g * NQ + c], v[c]);
        atomicAdd(&lentype[(size_t)g * 6 + t], 1);
    }
}

// fof_finish_group_properties, fof.c:705-755
__global__ void __launch_bounds__(256) k_fof_finish(int64_t ng, const float *__restrict__ firstpos, double box, double *__restrict__ acc)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(g >= ng)
        return;
    double *a = acc + (size_t)g * NQ;
    const double M = a[0];
    double cm[3], rel[3], vcm[3];
    for(int d = 0; d < 3; d++) {
        a[10 + d] /= M;
        vcm[d] = a[10 + d];
        cm[d] = a[7 + d] / M;
        double x = cm[d] - (double)firstpos[3 * g + d];
        if(x > 0.5 * box)
            x -= box;
        else if(x < -0.5 * box)
            x += box;
        rel[d] = x;
        while(cm[d] >= box) // fof_periodic_wrap, fof.c:88-95
            cm[d] -= box;
        while(cm[d] < 0)
            cm[d] += box;
        a[7 + d] = cm[d];
    }
    const double jcm[3] = {rel[1] * vcm[2] - rel[2] * vcm[1], rel[2] * vcm[0] - rel[0] * vcm[2], rel[0] * vcm[1] - rel[1] * vcm[0]};
    for(int d = 0; d < 3; d++)
        a[13 + d] -= jcm[d] * M;
    for(int d = 0; d < 3; d++)
        for(int e = 0; e < 3; e++)
            a[16 + 3 * d + e] -= M * rel[d] * rel[e];
}

__global__ void __launch_bounds__(256) k_fof_export(int64_t ng, const unsigned long long *__restrict__ minid, const unsigned *__restrict__ length,
                                                    const int *__restrict__ grnr, const int *__restrict__ lentype, const double *__restrict__ acc,
                                                    const float *__restrict__ firstpos, FofTable out)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(g >= ng)
        return;
    const double *a = acc + (size_t)g * NQ;
    if(out.MinID)
        out.MinID[g] = minid[g];
    if(out.Length)
        out.Length[g] = (int)length[g];
    if(out.GrNr)
        out.GrNr[g] = grnr[g];
    if(out.Mass)
        out.Mass[g] = a[0];
    for(int t = 0; t < 6; t++) {
        if(out.LenType)
            out.LenType[6 * g + t] = lentype[6 * g + t];
        if(out.MassType)
            out.MassType[6 * g + t] = a[1 + t];
    }
    for(int d = 0; d < 3; d++) {
        if(out.CM)
            out.CM[3 * g + d] = a[7 + d];
        if(out.Vel)
            out.Vel[3 * g + d] = a[10 + d];
        if(out.Jmom)
            out.Jmom[3 * g + d] = a[13 + d];
        if(out.FirstPos)
            out.FirstPos[3 * g + d] = firstpos[3 * g + d];
    }
    if(out.Imom)
        for(int c = 0; c < 9; c++)
            out.Imom[9 * g + c] = a[16 + c];
}

inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

} // namespace

int64_t FofEngine::run(TreeBuilder &tree, const FofInput &in, hipStream_t st)
{
    compute_labels(tree, in, st);
    return catalogue(in, in.n, nullptr, 0, true, st);
}

// primary linking + secondary attachment: label[i] = the smallest ID of the particles linked with i (its own ID if none)
void FofEngine::compute_labels(TreeBuilder &tree, const FofInput &in, hipStream_t st)
{
    const int64_t n = in.n;
    const int64_t np = tree.npart;
    ngroups = 0;
    err.reserve(4);
    MPG_HIP(hipMemsetAsync(err.p, 0, sizeof(unsigned), st));
    label.reserve((size_t)n + 1);
    if(n == 0)
        return;
    hipLaunchKernelGGL(k_fof_own_labels, dim3(nblk(n)), dim3(256), 0, st, n, in.id, label.p);
    const TreeView tv = tree.view();
    if(np > 0) {
        tree.ensure_level_order(st);
        const TreeView tvb = tree.view();
        parent.reserve((size_t)np + 1);
        root_of.reserve((size_t)np + 1);
        minid.reserve((size_t)np + 1);
        hipLaunchKernelGGL(k_fof_init, dim3(nblk(np)), dim3(256), 0, st, np, parent.p, minid.p);
        hipLaunchKernelGGL(k_fof_walk<0>, dim3((unsigned)((np + 31) / 32)), dim3(256), 0, st, tvb, in.LL, parent.p, (const int *)nullptr, np,
                           (const double *)nullptr, (const double *)nullptr, (const uint8_t *)nullptr, (const int *)nullptr,
                           (const unsigned long long *)nullptr, (unsigned long long *)nullptr, err.p);
        hipLaunchKernelGGL(k_fof_flatten, dim3(nblk(np)), dim3(256), 0, st, np, parent.p, root_of.p, tvb.order, in.id, minid.p);
        hipLaunchKernelGGL(k_fof_primary_labels, dim3(nblk(np)), dim3(256), 0, st, np, root_of.p, tvb.order, minid.p, label.p);
        // secondary particles: nearest primary
        if(in.secondary_mask) {
            val.reserve((size_t)n + 1);
            keep.reserve((size_t)n + 1);
            list.reserve((size_t)n + 1);
            cnt.reserve(8);
            hipLaunchKernelGGL(k_fof_secondary_flags, dim3(nblk(n)), dim3(256), 0, st, n, in.type, in.flags, in.secondary_mask, val.p, keep.p);
            compact_flagged(val.p, keep.p, n, list.p, cnt.p, tmp, st);
            unsigned long long ns = 0;
            MPG_HIP(hipMemcpyAsync(&ns, cnt.p, sizeof(ns), hipMemcpyDeviceToHost, st));
            MPG_HIP(hipStreamSynchronize(st));
            if(ns > 0)
                hipLaunchKernelGGL(k_fof_walk<1>, dim3((unsigned)((ns + 31) / 32)), dim3(256), 0, st, tvb, in.LL, parent.p, list.p, (int64_t)ns, in.pos,
                                   in.hsml, in.type, root_of.p, minid.p, label.p, err.p);
        }
    }
    (void)tv;
    unsigned e = 0;
    MPG_HIP(hipMemcpyAsync(&e, err.p, sizeof(e), hipMemcpyDeviceToHost, st));
    MPG_HIP(hipStreamSynchronize(st));
    MPG_CHECK(e == 0, "fof: the neighbour walk overflowed its stack (corrupt tree?)");
}

// The catalogue of the first `n` particles (n <= in.n: with ghosts behind them, only the own ones) from label[]: particles in label
// order, runs = groups.  A run is kept when it has at least in.minlen members or its label is in `also_keep` (sorted, unique: the
// labels shared with other ranks, whose parts must all be reported).  finish: fof_finish_group_properties; without it the sums stay
// raw (about FirstPos) so that parts of a group can be added up.
int64_t FofEngine::catalogue(const FofInput &in, int64_t n, const unsigned long long *also_keep, int64_t nalso, bool finish, hipStream_t st)
{
    ngroups = 0;
    if(n == 0)
        return 0;
    size_t b = 0;
    // ---- catalogue: particles in label order, runs = groups
    slabel.reserve((size_t)n + 1);
    sidx.reserve((size_t)n + 1);
    val.reserve((size_t)n + 1);
    hipLaunchKernelGGL(k_iota, dim3(nblk(n)), dim3(256), 0, st, n, val.p);
    MPG_HIP(rocprim::radix_sort_pairs(nullptr, b, label.p, slabel.p, val.p, sidx.p, (size_t)n, 0, 64, st));
    tmp.reserve(b + 16);
    MPG_HIP(rocprim::radix_sort_pairs((void *)tmp.p, b, label.p, slabel.p, val.p, sidx.p, (size_t)n, 0, 64, st));
    run_label.reserve((size_t)n + 1);
    run_count.reserve((size_t)n + 1);
    run_start.reserve((size_t)n + 1);
    cnt.reserve(8);
    tmp.reserve(64);
    b = 0;
    MPG_HIP(rocprim::run_length_encode(nullptr, b, slabel.p, (unsigned)n, run_label.p, run_count.p, cnt.p + 1, st));
    tmp.reserve(b + 16);
    MPG_HIP(rocprim::run_length_encode((void *)tmp.p, b, slabel.p, (unsigned)n, run_label.p, run_count.p, cnt.p + 1, st));
    unsigned long long nruns = 0;
    MPG_HIP(hipMemcpyAsync(&nruns, cnt.p + 1, sizeof(nruns), hipMemcpyDeviceToHost, st));
    MPG_HIP(hipStreamSynchronize(st));
    b = 0;
    MPG_HIP(rocprim::exclusive_scan(nullptr, b, run_count.p, run_start.p, 0u, (size_t)nruns, rocprim::plus<unsigned>(), st));
    tmp.reserve(b + 16);
    MPG_HIP(rocprim::exclusive_scan((void *)tmp.p, b, run_count.p, run_start.p, 0u, (size_t)nruns, rocprim::plus<unsigned>(), st));
    keep.reserve((size_t)nruns + 1);
    hipLaunchKernelGGL(k_fof_keep_runs, dim3(nblk((int64_t)nruns)), dim3(256), 0, st, (int64_t)nruns, run_count.p, in.minlen, keep.p, run_label.p,
                       also_keep, nalso);
    g_minid.reserve((size_t)nruns + 1);
    g_len.reserve((size_t)nruns + 1);
    g_start.reserve((size_t)nruns + 1);
    auto select3 = [&](auto *src, auto *dst) {
        size_t bb = 0;
        MPG_HIP(rocprim::select(nullptr, bb, src, keep.p, dst, cnt.p + 2, (size_t)nruns, st));
        tmp.reserve(bb + 16);
        MPG_HIP(rocprim::select((void *)tmp.p, bb, src, keep.p, dst, cnt.p + 2, (size_t)nruns, st));
    };
    select3(run_label.p, g_minid.p);
    select3(run_count.p, g_len.p);
    select3(run_start.p, g_start.p);
    unsigned long long ng = 0;
    MPG_HIP(hipMemcpyAsync(&ng, cnt.p + 2, sizeof(ng), hipMemcpyDeviceToHost, st));
    MPG_HIP(hipStreamSynchronize(st));
    ngroups = (int64_t)ng;
    // group numbers: by (Length descending, MinID ascending)
    g_grnr.reserve((size_t)ng + 1);
    g_first.reserve(3 * (size_t)ng + 3);
    g_acc.reserve((size_t)ng * NQ + NQ);
    g_lentype.reserve((size_t)ng * 6 + 6);
    p_grnr.reserve((size_t)n + 1);
    if(ng > 0) {
        lenkey_a.reserve((size_t)ng + 1);
        lenkey_b.reserve((size_t)ng + 1);
        ord_a.reserve((size_t)ng + 1);
        ord_b.reserve((size_t)ng + 1);
        hipLaunchKernelGGL(k_fof_lenkeys, dim3(nblk((int64_t)ng)), dim3(256), 0, st, (int64_t)ng, g_len.p, lenkey_a.p);
        hipLaunchKernelGGL(k_iota, dim3(nblk((int64_t)ng)), dim3(256), 0, st, (int64_t)ng, ord_a.p);
        b = 0;
        MPG_HIP(rocprim::radix_sort_pairs(nullptr, b, lenkey_a.p, lenkey_b.p, ord_a.p, ord_b.p, (size_t)ng, 0, 32, st));
        tmp.reserve(b + 16);
        MPG_HIP(rocprim::radix_sort_pairs((void *)tmp.p, b, lenkey_a.p, lenkey_b.p, ord_a.p, ord_b.p, (size_t)ng, 0, 32, st));
        hipLaunchKernelGGL(k_fof_grnr, dim3(nblk((int64_t)ng)), dim3(256), 0, st, (int64_t)ng, ord_b.p, g_grnr.p);
        hipLaunchKernelGGL(k_fof_firstpos, dim3(nblk((int64_t)ng)), dim3(256), 0, st, (int64_t)ng, g_start.p, sidx.p, in.pos, g_first.p);
        MPG_HIP(hipMemsetAsync(g_acc.p, 0, (size_t)ng * NQ * sizeof(double), st));
        MPG_HIP(hipMemsetAsync(g_lentype.p, 0, (size_t)ng * 6 * sizeof(int), st));
    }
    hipLaunchKernelGGL(k_fof_accumulate, dim3(nblk(n)), dim3(256), 0, st, n, (int64_t)ng, g_start.p, g_len.p, g_grnr.p, sidx.p, in.pos, in.vel, in.mass,
                       in.type, g_first.p, in.box, p_grnr.p, g_acc.p, g_lentype.p);
    if(ng > 0 && finish)
        hipLaunchKernelGGL(k_fof_finish, dim3(nblk((int64_t)ng)), dim3(256), 0, st, (int64_t)ng, g_first.p, in.box, g_acc.p);
    MPG_HIP(hipStreamSynchronize(st));
    MPG_HIP(hipGetLastError());
    return ngroups;
}

void FofEngine::export_groups(const FofTable &out, hipStream_t st)
{
    if(ngroups > 0)
        hipLaunchKernelGGL(k_fof_export, dim3(nblk(ngroups)), dim3(256), 0, st, ngroups, (const unsigned long long *)g_minid.p, g_len.p, g_grnr.p,
                           g_lentype.p, g_acc.p, g_first.p, out);
    MPG_HIP(hipGetLastError());
}

} // namespace mpg
