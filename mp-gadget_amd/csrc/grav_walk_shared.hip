// grav_walk_shared.hip -- short-range gravity walk with ONE tree traversal shared by the 8 targets of a wave (variant 5).
//
// Per-target semantics are the reference's (force_treeev_shortrange, gravshort-tree.c:253-379): every (target, node) pair is
// discarded, used unopened or opened by exactly the reference's tests for that target, so the interaction set of every
// particle is the reference's; only the summation order differs (tests assert equal interaction counters).
//
// rocprof on the lane-per-target kernel (profiles/r01a_first) shows 2.7e9 wave-level vector loads per launch: every lane
// fetches its own 32-byte source and 80-byte node records, and the vector memory pipe (64 B/clk/CU), not fp64 issue, sets
// the time.  Eight consecutive targets in tree order sit in (or next to) one leaf and walk almost the same nodes, so here
// a wave of 64 lanes = 8 targets x 8 slots shares the traversal:
//   phase A  a wave-uniform LIFO of child ranges (level-ordered tree: the <= 8 children of a node are contiguous).  One
//            step pops a range with an 8-bit mask of the targets that opened the parent; lane (g, s) tests child s for
//            target g.  Per child the decisions of the 8 targets are gathered into bit masks (one 8x8 bit-matrix column
//            extraction per ballot): children opened by any target are pushed / appended to the leaf list with that mask,
//            children used unopened by any target go to the node list with theirs.
//   phase B  per leaf entry, lane (g, s) evaluates source s for target g if g is in the entry's mask (one 256-byte read
//            serves all 8 targets); node entries are taken 8 at a time.  Sums are reduced over s at the end.
// The lists live in a per-wave scratch area and are read with wave-uniform addresses.  Kernel is persistent; the waves of
// one XCD take chunks of 8 tree-ordered targets round-robin from that XCD's contiguous part of the target range.
#include "grav_walk.h"
#include "grav_pair.h"

namespace mpg {

constexpr int STK5 = 192; // pending child ranges per wave

// ballot bit index = lane = g*8 + s: view the 64 bits as an 8x8 matrix (row g, column s) and extract column s as a byte
__device__ __forceinline__ unsigned column_mask(unsigned long long b, int s)
{
    const unsigned long long t = (b >> s) & 0x0101010101010101ULL;
    return (unsigned)((t * 0x0102040810204080ULL) >> 56);
}

// counters (COUNT builds): [0] pair interactions [1] nodes visited [2] nodes used unopened
//   [3] phase-A wave steps [4] (target, child) tests done in them [5] phase-B lane-steps issued [6] of which active
template <bool POT, bool COUNT>
__global__ void __launch_bounds__(256) k_grav_walk_shared(const TreeView tv, const GravParams gp, const WalkIO io, int2 *__restrict__ scratch,
                                                          const int cap, unsigned *__restrict__ err)
{
    __shared__ WTab s_wf[NTAB];
    __shared__ float2 s_wp[POT ? NTAB : 1];
    __shared__ uint2 s_stack[4 * STK5];
    for(int i = threadIdx.x; i < NTAB - 1; i += blockDim.x) {
        s_wf[i] = WTab{(double)io.tab_force[i], (double)io.tab_force[i + 1]};
        if(POT)
            s_wp[i] = make_float2(io.tab_pot[i], io.tab_pot[i + 1]);
    }
    if(threadIdx.x == 0) {
        s_wf[NTAB - 1] = WTab{0, 0};
        if(POT)
            s_wp[NTAB - 1] = make_float2(0.f, 0.f);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int g = lane >> 3, s = lane & 7;
    const int wib = threadIdx.x >> 6;
    const int64_t gwave = (int64_t)blockIdx.x * (blockDim.x >> 6) + wib;
    // per-wave lists (wave-uniform addresses): leaf entries grow up from 0, node entries down from cap-1
    int2 *__restrict__ list = scratch + gwave * (int64_t)cap;
    uint2 *stack = s_stack + wib * STK5;

    const unsigned nchunks = (unsigned)((io.ntargets + 7) / 8);
    const unsigned xcd = blockIdx.x & 7;
    const unsigned waves_per_block = blockDim.x >> 6;
    const unsigned wave_in_xcd = (blockIdx.x >> 3) * waves_per_block + wib;
    const unsigned waves_in_xcd = (gridDim.x >> 3) * waves_per_block;
    const unsigned lo = (unsigned)(((uint64_t)nchunks * xcd) >> 3), hi = (unsigned)(((uint64_t)nchunks * (xcd + 1)) >> 3);
    unsigned long long guard = 0;
    const unsigned long long guard_max = 64ull * (unsigned long long)(tv.nnodes + tv.npart + 1024);

    for(unsigned chunk = lo + wave_in_xcd; chunk < hi; chunk += waves_in_xcd) {
        const int64_t slot = (int64_t)chunk * 8 + g;
        const bool valid = slot < io.ntargets;
        int ci = -1;
        double px = 0, py = 0, pz = 0, aold = 0;
        if(valid) {
            ci = io.targets ? io.targets[slot] : tv.order[slot];
            px = io.pos[3 * (int64_t)ci + 0];
            py = io.pos[3 * (int64_t)ci + 1];
            pz = io.pos[3 * (int64_t)ci + 2];
            double old = 0;
            if(io.oldacc)
                old = io.oldacc[ci];
            else if(io.prev_accel) { // grav_get_abs_accel, gravshort.h:70-80
                double s2 = 0;
                for(int j = 0; j < 3; j++) {
                    const double a = io.prev_accel[3 * (int64_t)ci + j] + (io.gravpm ? io.gravpm[3 * (int64_t)ci + j] : 0.0);
                    s2 += a * a;
                }
                old = sqrt(s2) / gp.G;
            }
            aold = gp.errtol * old;
        }
        const unsigned validmask = column_mask(__ballot(valid), 0); // bit g: target g exists

        int sp = 0; // wave-uniform
        if(validmask) {
            if(lane == 0)
                stack[0] = make_uint2((0u << 4) | 1u, validmask); // the root, for every valid target
            sp = 1;
        }
        double ax = 0, ay = 0, az = 0, pot = 0;
        unsigned n_pp = 0, n_vis = 0, n_used = 0, st_a = 0, st_al = 0, st_b = 0, st_bl = 0;

        do {
            // ------------------------------------------------------------------ phase A: shared cooperative walk
            int nleaf = 0, nuse = 0; // wave-uniform list fill
            while(sp > 0 && nleaf + nuse + 16 <= cap) {
                if(++guard > guard_max || sp + 8 > STK5) {
                    if(lane == 0)
                        atomicExch(&err[0], (guard > guard_max) ? 1u : 4u);
                    return;
                }
                const uint2 range = stack[sp - 1];
                const int first = (int)(range.x >> 4), nch = (int)(range.x & 15u);
                const unsigned rmask = range.y;
                const bool active = (s < nch) && ((rmask >> g) & 1u);
                int act = 0; // 0 discard, 1 leaf opened, 2 node used unopened, 3 internal node opened
                int pstart = 0, pcount = 0, fchild = 0, nchild = 0;
                if(active) {
                    const int my = first + s;
                    const NodeGeo ng = tv.geoB[my];
                    const Src4 mom = tv.momB[my];
                    const NodeLinkB lk = tv.linkB[my];
                    pstart = lk.pstart;
                    pcount = lk.pcount;
                    fchild = lk.firstchild;
                    nchild = lk.nchild;
                    // NEAREST(cofm - pos) and NEAREST(center - pos), gravshort-tree.c:299-300, 211, 234-236
                    double dx = mom.x - px, dy = mom.y - py, dz = mom.z - pz;
                    dx = fma(-rint(dx * gp.invbox), gp.box, dx);
                    dy = fma(-rint(dy * gp.invbox), gp.box, dy);
                    dz = fma(-rint(dz * gp.invbox), gp.box, dz);
                    double cdx = ng.cx - px, cdy = ng.cy - py, cdz = ng.cz - pz;
                    cdx = fabs(fma(-rint(cdx * gp.invbox), gp.box, cdx));
                    cdy = fabs(fma(-rint(cdy * gp.invbox), gp.box, cdy));
                    cdz = fabs(fma(-rint(cdz * gp.invbox), gp.box, cdz));
                    const double r2 = dx * dx + dy * dy + dz * dz;
                    // shall_we_discard_node, gravshort-tree.c:198-215
                    const double eff = fma(0.5, ng.len, gp.rcut);
                    const bool discard = (r2 > gp.rcut2) && (cdx > eff || cdy > eff || cdz > eff);
                    if(!discard) {
                        // shall_we_open_node, gravshort-tree.c:220-241
                        const double l2 = ng.len * ng.len;
                        const double inside = 0.6 * ng.len;
                        const bool open = ((!gp.use_bh) && (mom.m * l2 > r2 * r2 * aold)) || (l2 > r2 * gp.bhangle2) ||
                                          (cdx < inside && cdy < inside && cdz < inside);
                        if(!open)
                            act = 2;
                        else if(pcount > 0)
                            act = 1;
                        else if(nchild > 0)
                            act = 3;
                    }
                    if(COUNT) {
                        n_vis++;
                        if(act == 2)
                            n_used++;
                        if(act == 1)
                            n_pp += pcount;
                    }
                }
                // per child s: which targets opened it as a leaf / use it unopened / opened it as an internal node
                const unsigned m_leaf = column_mask(__ballot(act == 1), s);
                const unsigned m_use = column_mask(__ballot(act == 2), s);
                const unsigned m_push = column_mask(__ballot(act == 3), s);
                // the 8 lanes of group 0 (lane == s) own the children; node data of child s is target independent
                const bool owner = (g == 0) && (s < nch);
                // broadcast the child's link data from any lane that loaded it (lane with smallest g in rmask)
                const int srcg = __ffs((int)rmask) - 1;
                const int b_pstart = __shfl(pstart, srcg * 8 + s), b_pcount = __shfl(pcount, srcg * 8 + s);
                const int b_fchild = __shfl(fchild, srcg * 8 + s), b_nchild = __shfl(nchild, srcg * 8 + s);
                const unsigned long long bl = __ballot(owner && m_leaf != 0), bu = __ballot(owner && m_use != 0),
                                         bp = __ballot(owner && m_push != 0);
                const unsigned long long below = (1ull << lane) - 1ull;
                if(owner && m_leaf != 0)
                    list[nleaf + __popcll(bl & below)] = make_int2(b_pstart, b_pcount | (int)(m_leaf << 4));
                if(owner && m_use != 0)
                    list[cap - 1 - (nuse + __popcll(bu & below))] = make_int2(first + s, (int)m_use);
                if(owner && m_push != 0)
                    stack[sp - 1 + __popcll(bp & below)] = make_uint2(((unsigned)b_fchild << 4) | (unsigned)b_nchild, m_push);
                nleaf += __popcll(bl);
                nuse += __popcll(bu);
                sp += __popcll(bp) - 1;
                if(COUNT && lane == 0) {
                    st_a++;
                    st_al += nch * __popc(rmask);
                }
            }
            // ------------------------------------------------------------------ phase B1: leaf entries
#pragma unroll 1
            for(int r = 0; r < nleaf; r++) {
                if(++guard > guard_max) {
                    if(lane == 0)
                        atomicExch(&err[0], 2u);
                    return;
                }
                const int2 e = list[r];
                const int cnt = e.y & 15;
                const bool has = (s < cnt) && ((e.y >> (4 + g)) & 1);
                if(COUNT) {
                    st_b++;
                    st_bl += has ? 1 : 0;
                }
                if(has) {
                    const Src4 sc = tv.src[e.x + s];
                    double dx = sc.x - px, dy = sc.y - py, dz = sc.z - pz;
                    dx = fma(-rint(dx * gp.invbox), gp.box, dx);
                    dy = fma(-rint(dy * gp.invbox), gp.box, dy);
                    dz = fma(-rint(dz * gp.invbox), gp.box, dz);
                    pair_force<POT>(sc, dx, dy, dz, gp, s_wf, s_wp, ax, ay, az, pot);
                }
            }
            // ------------------------------------------------------------------ phase B2: node entries, 8 per step
#pragma unroll 1
            for(int r0 = 0; r0 < nuse; r0 += 8) {
                if(++guard > guard_max) {
                    if(lane == 0)
                        atomicExch(&err[0], 3u);
                    return;
                }
                const int r = r0 + s;
                bool has = r < nuse;
                int2 e = make_int2(0, 0);
                if(has) {
                    e = list[cap - 1 - r];
                    has = (e.y >> g) & 1;
                }
                if(COUNT) {
                    st_b++;
                    st_bl += has ? 1 : 0;
                }
                if(has) {
                    const Src4 sc = tv.momB[e.x];
                    double dx = sc.x - px, dy = sc.y - py, dz = sc.z - pz;
                    dx = fma(-rint(dx * gp.invbox), gp.box, dx);
                    dy = fma(-rint(dy * gp.invbox), gp.box, dy);
                    dz = fma(-rint(dz * gp.invbox), gp.box, dz);
                    pair_force<POT>(sc, dx, dy, dz, gp, s_wf, s_wp, ax, ay, az, pot);
                }
            }
            guard = 0;
        } while(sp > 0); // a list filled up: keep walking

        // reduce the partial sums over the 8 slots of each target
        for(int off = 1; off < 8; off <<= 1) {
            ax += __shfl_xor(ax, off);
            ay += __shfl_xor(ay, off);
            az += __shfl_xor(az, off);
            if(POT)
                pot += __shfl_xor(pot, off);
        }
        if(valid && s == 0) {
            // grav_short_reduce (assign) + grav_short_postprocess, gravshort.h:47-67,88-96
            io.accel[3 * (int64_t)ci + 0] = ax * gp.G;
            io.accel[3 * (int64_t)ci + 1] = ay * gp.G;
            io.accel[3 * (int64_t)ci + 2] = az * gp.G;
            if(POT && io.potential) {
                const double m = (double)io.mass[ci];
                double p = pot;
                p += m / (gp.h / 2.8);
                p -= 2.8372975 * pow(m, 2.0 / 3) * gp.cbrtrho0;
                p *= gp.G;
                io.potential[ci] = p;
            }
        }
        if(COUNT) {
            unsigned long long c0 = n_pp, c1 = n_vis, c2 = n_used, c3 = st_a, c4 = st_al, c5 = st_b, c6 = st_bl;
            for(int off = 32; off > 0; off >>= 1) {
                c0 += __shfl_down(c0, off);
                c1 += __shfl_down(c1, off);
                c2 += __shfl_down(c2, off);
                c3 += __shfl_down(c3, off);
                c4 += __shfl_down(c4, off);
                c5 += __shfl_down(c5, off);
                c6 += __shfl_down(c6, off);
            }
            if(lane == 0) {
                atomicAdd(&io.counters[0], c0);
                atomicAdd(&io.counters[1], c1);
                atomicAdd(&io.counters[2], c2);
                atomicAdd(&io.counters[3], c3);
                atomicAdd(&io.counters[4], c4);
                atomicAdd(&io.counters[5], c5);
                atomicAdd(&io.counters[6], c6);
            }
        }
    }
}

template <bool POT, bool COUNT> static void launch_shared_t(const TreeView &tv, const GravParams &gp, const WalkIO &io, WalkScratch &ws, hipStream_t st)
{
    if(io.ntargets == 0)
        return;
    auto kern = k_grav_walk_shared<POT, COUNT>;
    if(ws.num_cu == 0) {
        int dev = 0;
        MPG_HIP(hipGetDevice(&dev));
        MPG_HIP(hipDeviceGetAttribute(&ws.num_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    int occ = 0;
    MPG_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, 0));
    if(occ < 1)
        occ = 1;
    if(occ > 8)
        occ = 8;
    const int64_t nchunks = (io.ntargets + 7) / 8;
    int64_t nblocks = (int64_t)ws.num_cu * occ;
    const int64_t need = (nchunks + 3) / 4;
    if(nblocks > need)
        nblocks = need;
    nblocks = (nblocks + 7) / 8 * 8;
    const int cap = ws.cap < 64 ? 64 : ws.cap;
    ws.list.reserve((size_t)nblocks * 4 * 8 * ws.cap + (size_t)nblocks * 4 * cap);
    ws.ctr.reserve(16);
    MPG_HIP(hipMemsetAsync(ws.ctr.p, 0, 16 * sizeof(unsigned), st));
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(256), 0, st, tv, gp, io, ws.list.p, cap, ws.ctr.p + 8);
    MPG_HIP(hipGetLastError());
}

void launch_grav_walk_shared(const TreeView &tv, const GravParams &gp, const WalkIO &io, bool want_pot, bool count, WalkScratch &ws,
                             hipStream_t st)
{
    if(want_pot) {
        if(count)
            launch_shared_t<true, true>(tv, gp, io, ws, st);
        else
            launch_shared_t<true, false>(tv, gp, io, ws, st);
    }
    else {
        if(count)
            launch_shared_t<false, true>(tv, gp, io, ws, st);
        else
            launch_shared_t<false, false>(tv, gp, io, ws, st);
    }
}

} // namespace mpg
