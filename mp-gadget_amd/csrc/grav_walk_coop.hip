// grav_walk_coop.hip -- short-range gravity walk, group-cooperative list form (the default walk kernel).
//
// Per-target semantics are the reference's (force_treeev_shortrange, gravshort-tree.c:253-379): every node a target
// visits is discarded, used unopened or opened by exactly the reference's tests for THAT target, so the interaction set
// per particle is the reference's; only the summation order differs.  Like the reference, which first collects the
// particles of the opened leaves in `ngblist` and evaluates them afterwards (gravshort-tree.c:346-374), the walk has two
// phases.  Mapping (MI355X, 64-lane waves): a wave is split into 8 groups of 8 lanes; each group owns ONE target.
//
//   phase A  the walk uses a level-ordered copy of the tree in which the children of a node are contiguous.  The group
//            keeps a small LIFO of pending child ranges in LDS; one step pops a range, the 8 lanes test the <= 8
//            children (one coalesced read each of geometry / moments / links), and every opened internal child pushes
//            its own child range.  Opened leaves go to the group's leaf list, nodes used unopened to its node list
//            (both in a per-wave scratch area, 8 bytes/entry).  Steps per target = 1 + number of opened internal nodes,
//            instead of one dependent memory access per visited node.
//   phase B  for every leaf entry, lane s evaluates source s of the leaf (one coalesced 256-byte read per group);
//            node entries are spread over the lanes 8 at a time.  Partial sums are reduced over the 8 lanes at the end.
//
// Why this shape: rocprof on the first kernel (one target per lane, profiles/r01a_first) showed fp64 VALU issue as the
// limiter with ~48 % of the lanes active; a one-target-per-lane list kernel fixed the utilisation but became bound by
// the vector L1 instead (64 lanes gathering 32-byte records from 64 different leaves, profiles/r01b_walk3_unpipelined).
// Here a wave touches at most 8 leaves per step, 2 cache lines each, and all loads are software-pipelined one step ahead.
//
// Per-pair arithmetic (fp64): 3 sub, r2 (3), v_rsq_f64 + one Newton-cubic step (6), window lookup = one ds_read_b128 of
// (T[t], T[t+1]) + 4 flops, Newtonian factor (3), 3 fma into the acceleration (+ 5 for the potential).  The minimum-image
// wrap is hoisted out of the pair loop when the box is large enough that all sources of one list entry share the image
// of their node (FASTWRAP, decided on the host); for the unwrapped image (all pairs away from the box faces) the
// arithmetic is bit-identical to the per-pair NEAREST of partmanager.h:99.
#include "grav_walk.h"
#include "grav_pair.h"

namespace mpg {

constexpr int STK = 160; // pending child ranges per group (LIFO): <= 7 per tree level + 8, 21 levels

__device__ __forceinline__ void image_shift(const int code, const GravParams &gp, const double px, const double py, const double pz,
                                            double &spx, double &spy, double &spz)
{
    spx = fma((double)((code & 3) - 1), gp.box, px);
    spy = fma((double)(((code >> 2) & 3) - 1), gp.box, py);
    spz = fma((double)(((code >> 4) & 3) - 1), gp.box, pz);
}

// counters (COUNT builds): [0] pair interactions [1] nodes visited [2] nodes used unopened
//   [3] phase-A group steps [4] nodes consumed by them (of 8 tested) [5] phase-B lane-steps issued (8 per group step) [6] of which active
template <bool POT, bool COUNT, bool FASTWRAP>
__global__ void __launch_bounds__(256, 3) k_grav_walk_coop(const TreeView tv, const GravParams gp, const WalkIO io, int2 *__restrict__ scratch,
                                                        const int cap, unsigned *__restrict__ err)
{
    __shared__ WTab s_wf[NTAB];
    __shared__ float2 s_wp[POT ? NTAB : 1];
    __shared__ unsigned s_stack[4 * 8 * STK];
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
    const int grp = lane >> 3, s = lane & 7;
    const int gshift = grp * 8; // first lane of this group
    const int64_t gwave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int2 *__restrict__ list = scratch + (gwave * 8 + grp) * (int64_t)cap; // leaf entries grow up from 0, node entries down from cap-1
    unsigned *stack = s_stack + ((threadIdx.x >> 6) * 8 + grp) * STK;            // pending child ranges of this group: (first << 4) | count

    // chunks of 8 targets (one per group); XCD x (= blockIdx % 8, where the hardware places this block) owns a contiguous
    // part of the tree-ordered targets and its waves take chunks round-robin, so each private L2 serves one region of the tree
    const unsigned nchunks = (unsigned)((io.ntargets + 7) / 8);
    const unsigned xcd = blockIdx.x & 7;
    const unsigned waves_per_block = blockDim.x >> 6;
    const unsigned wave_in_xcd = (blockIdx.x >> 3) * waves_per_block + (threadIdx.x >> 6);
    const unsigned waves_in_xcd = (gridDim.x >> 3) * waves_per_block;
    const unsigned lo = (unsigned)(((uint64_t)nchunks * xcd) >> 3), hi = (unsigned)(((uint64_t)nchunks * (xcd + 1)) >> 3);
    unsigned long long guard = 0; // bound on loop iterations: a runaway loop reports through err[] instead of hanging the GPU
    const unsigned long long guard_max = 64ull * (unsigned long long)(tv.nnodes + tv.npart + 1024);

    for(unsigned chunk = lo + wave_in_xcd; chunk < hi; chunk += waves_in_xcd) {
        const int64_t slot = (int64_t)chunk * 8 + grp;
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

        int sp = 0; // stack pointer (group-uniform)
        if(valid) {
            if(s == 0)
                stack[0] = (0u << 4) | 1u; // the root
            sp = 1;
        }
        double ax = 0, ay = 0, az = 0, pot = 0;
        double bx = 0, by = 0, bz = 0, potb = 0; // second accumulator chain (two entries per iteration in phase B1)
        unsigned n_pp = 0, n_vis = 0, n_used = 0, st_a = 0, st_al = 0, st_b = 0, st_bl = 0;
        long long cyc_a = 0, cyc_b = 0, cyc_t0 = 0;

        do {
            // ------------------------------------------------------------------ phase A: cooperative walk
            // The group keeps a LIFO of pending child ranges in LDS.  One step pops a range (the <= 8 children of an
            // opened node, contiguous in the level-ordered tree), lane s tests child s, and every opened internal child
            // pushes its own child range.  Steps per target = 1 + number of opened internal nodes.
            int nleaf = 0, nnode = 0; // entries in the two lists (group-uniform)
            if(COUNT)
                cyc_t0 = clock64();
            for(;;) {
                const bool can = (sp > 0) && (nleaf + nnode + 8 <= cap);
                if(__ballot(can) == 0)
                    break;
                if(++guard > guard_max || __ballot(can && sp + 8 > STK) != 0) {
                    if(lane == 0)
                        atomicExch(&err[0], (guard > guard_max) ? 1u : 4u);
                    return;
                }
                const unsigned range = can ? stack[sp - 1] : 0u;
                const int first = (int)(range >> 4), nch = (int)(range & 15u);
                int act = 0; // 0 nothing, 1 leaf opened (list), 2 node used unopened (list), 3 internal node opened (push)
                unsigned pushval = 0;
                int2 entry = make_int2(0, 0);
                if(can && s < nch) {
                    const int my = first + s;
                    const NodeGeo g = tv.geoB[my];
                    const Src4 mom = tv.momB[my];
                    const NodeLinkB lk = tv.linkB[my];
                    // periodic image of this node relative to the target: k = rint((c - p)/Box) per axis
                    const double kx = rint((g.cx - px) * gp.invbox);
                    const double ky = rint((g.cy - py) * gp.invbox);
                    const double kz = rint((g.cz - pz) * gp.invbox);
                    double dx, dy, dz, cdx, cdy, cdz;
                    int code;
                    if(FASTWRAP) {
                        const double qx = fma(kx, gp.box, px), qy = fma(ky, gp.box, py), qz = fma(kz, gp.box, pz);
                        cdx = fabs(g.cx - qx);
                        cdy = fabs(g.cy - qy);
                        cdz = fabs(g.cz - qz);
                        dx = mom.x - qx;
                        dy = mom.y - qy;
                        dz = mom.z - qz;
                        code = ((int)kx + 1) | (((int)ky + 1) << 2) | (((int)kz + 1) << 4);
                        if(g.len * 4.0 > gp.box) {
                            // top levels only: centre of mass and geometric centre may sit on different periodic
                            // images; take NEAREST(cofm - pos) exactly as gravshort-tree.c:299-300 does
                            const double jx = rint((mom.x - px) * gp.invbox), jy = rint((mom.y - py) * gp.invbox),
                                         jz = rint((mom.z - pz) * gp.invbox);
                            dx = fma(-jx, gp.box, mom.x - px);
                            dy = fma(-jy, gp.box, mom.y - py);
                            dz = fma(-jz, gp.box, mom.z - pz);
                            code = ((int)jx + 1) | (((int)jy + 1) << 2) | (((int)jz + 1) << 4);
                        }
                    }
                    else {
                        cdx = fabs(fma(-kx, gp.box, g.cx - px));
                        cdy = fabs(fma(-ky, gp.box, g.cy - py));
                        cdz = fabs(fma(-kz, gp.box, g.cz - pz));
                        dx = mom.x - px;
                        dy = mom.y - py;
                        dz = mom.z - pz;
                        dx = fma(-rint(dx * gp.invbox), gp.box, dx);
                        dy = fma(-rint(dy * gp.invbox), gp.box, dy);
                        dz = fma(-rint(dz * gp.invbox), gp.box, dz);
                        code = 21;
                    }
                    const double r2 = dx * dx + dy * dy + dz * dz;
                    // shall_we_discard_node, gravshort-tree.c:198-215
                    const double eff = fma(0.5, g.len, gp.rcut);
                    const bool discard = (r2 > gp.rcut2) && (cdx > eff || cdy > eff || cdz > eff);
                    if(!discard) {
                        // shall_we_open_node, gravshort-tree.c:220-241
                        const double l2 = g.len * g.len;
                        const double inside = 0.6 * g.len;
                        const bool open = ((!gp.use_bh) && (mom.m * l2 > r2 * r2 * aold)) || (l2 > r2 * gp.bhangle2) ||
                                          (cdx < inside && cdy < inside && cdz < inside);
                        if(!open) {
                            act = 2; // node used unopened: its moments are a 1-element source
                            entry = make_int2(my, 1 | (code << 4));
                        }
                        else if(lk.pcount > 0) {
                            act = 1;
                            entry = make_int2(lk.pstart, lk.pcount | (code << 4));
                        }
                        else if(lk.nchild > 0) {
                            act = 3;
                            pushval = ((unsigned)lk.firstchild << 4) | (unsigned)lk.nchild;
                        }
                    }
                    if(COUNT) {
                        n_vis++;
                        if(act == 2)
                            n_used++;
                        if(act == 1)
                            n_pp += lk.pcount;
                    }
                }
                const unsigned gm_leaf = (unsigned)((__ballot(act == 1) >> gshift) & 0xffull);
                const unsigned gm_node = (unsigned)((__ballot(act == 2) >> gshift) & 0xffull);
                const unsigned gm_push = (unsigned)((__ballot(act == 3) >> gshift) & 0xffull);
                const unsigned below = (1u << s) - 1u;
                if(act == 1)
                    list[nleaf + __popc(gm_leaf & below)] = entry;
                if(act == 2)
                    list[cap - 1 - (nnode + __popc(gm_node & below))] = entry;
                if(act == 3)
                    stack[sp - 1 + __popc(gm_push & below)] = pushval;
                if(can) {
                    nleaf += __popc(gm_leaf);
                    nnode += __popc(gm_node);
                    sp += __popc(gm_push) - 1;
                    if(COUNT && s == 0) {
                        st_a++;
                        st_al += nch;
                    }
                }
            }
            if(COUNT) {
                const long long t1 = clock64();
                cyc_a += t1 - cyc_t0;
                cyc_t0 = t1;
            }
            // ------------------------------------------------------------------ phase B1: leaf entries, lane s <-> source s
            // Two entries per iteration (two independent dependency chains per lane) and a software pipeline: the entries
            // and sources of iteration k+1 are requested before the pairs of iteration k are evaluated.  The explicit
            // s_waitcnt keeps hipcc from hoisting the new requests above the wait for the old ones (its own loop-carried
            // scoreboard merge would otherwise expose the full latency of the newest load every iteration).
            {
                const int npair = (nleaf + 1) >> 1;
                int2 ea_n = make_int2(0, 0), eb_n = make_int2(0, 0);
                Src4 sa_n{}, sb_n{};
                if(0 < nleaf)
                    ea_n = list[0];
                if(1 < nleaf)
                    eb_n = list[1];
                if(s < (ea_n.y & 15))
                    sa_n = tv.src[ea_n.x + s];
                if(s < (eb_n.y & 15))
                    sb_n = tv.src[eb_n.x + s];
                for(int k = 0;; k++) {
                    const bool more = k < npair;
                    if(__ballot(more) == 0)
                        break;
                    if(++guard > guard_max) {
                        if(lane == 0)
                            atomicExch(&err[0], 2u);
                        return;
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const int2 ea = ea_n, eb = eb_n;
                    const Src4 sa = sa_n, sb = sb_n;
                    const bool hasa = more && s < (ea.y & 15), hasb = more && s < (eb.y & 15);
                    // request iteration k+1
                    const int r2 = 2 * k + 2;
                    ea_n = make_int2(0, 0);
                    eb_n = make_int2(0, 0);
                    if(r2 < nleaf)
                        ea_n = list[r2];
                    if(r2 + 1 < nleaf)
                        eb_n = list[r2 + 1];
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // list entries are L2-resident; sources are the long pole
                    if(s < (ea_n.y & 15))
                        sa_n = tv.src[ea_n.x + s];
                    if(s < (eb_n.y & 15))
                        sb_n = tv.src[eb_n.x + s];
                    if(COUNT) {
                        if(more) {
                            st_b += 1 + ((2 * k + 1 < nleaf) ? 1 : 0);
                            st_bl += (hasa ? 1 : 0) + (hasb ? 1 : 0);
                        }
                    }
                    double dxa = 0, dya = 0, dza = 0, dxb = 0, dyb = 0, dzb = 0;
                    if(FASTWRAP) {
                        double spx, spy, spz;
                        image_shift(ea.y >> 4, gp, px, py, pz, spx, spy, spz);
                        dxa = sa.x - spx;
                        dya = sa.y - spy;
                        dza = sa.z - spz;
                        image_shift(eb.y >> 4, gp, px, py, pz, spx, spy, spz);
                        dxb = sb.x - spx;
                        dyb = sb.y - spy;
                        dzb = sb.z - spz;
                    }
                    else {
                        dxa = sa.x - px;
                        dya = sa.y - py;
                        dza = sa.z - pz;
                        dxa = fma(-rint(dxa * gp.invbox), gp.box, dxa);
                        dya = fma(-rint(dya * gp.invbox), gp.box, dya);
                        dza = fma(-rint(dza * gp.invbox), gp.box, dza);
                        dxb = sb.x - px;
                        dyb = sb.y - py;
                        dzb = sb.z - pz;
                        dxb = fma(-rint(dxb * gp.invbox), gp.box, dxb);
                        dyb = fma(-rint(dyb * gp.invbox), gp.box, dyb);
                        dzb = fma(-rint(dzb * gp.invbox), gp.box, dzb);
                    }
                    if(hasa)
                        pair_force<POT>(sa, dxa, dya, dza, gp, s_wf, s_wp, ax, ay, az, pot);
                    if(hasb)
                        pair_force<POT>(sb, dxb, dyb, dzb, gp, s_wf, s_wp, bx, by, bz, potb);
                }
            }
            // ------------------------------------------------------------------ phase B2: node entries, 8 per group step
            for(int r = s;; r += 8) {
                const bool has = r < nnode;
                if(__ballot(has) == 0)
                    break;
                if(++guard > guard_max) {
                    if(lane == 0)
                        atomicExch(&err[0], 3u);
                    return;
                }
                if(COUNT) {
                    st_b++;
                    st_bl += has ? 1 : 0;
                }
                if(has) {
                    const int2 it = list[cap - 1 - r];
                    const Src4 sc = tv.momB[it.x];
                    double dx, dy, dz;
                    if(FASTWRAP) {
                        double spx, spy, spz;
                        image_shift(it.y >> 4, gp, px, py, pz, spx, spy, spz);
                        dx = sc.x - spx;
                        dy = sc.y - spy;
                        dz = sc.z - spz;
                    }
                    else {
                        dx = sc.x - px;
                        dy = sc.y - py;
                        dz = sc.z - pz;
                        dx = fma(-rint(dx * gp.invbox), gp.box, dx);
                        dy = fma(-rint(dy * gp.invbox), gp.box, dy);
                        dz = fma(-rint(dz * gp.invbox), gp.box, dz);
                    }
                    pair_force<POT>(sc, dx, dy, dz, gp, s_wf, s_wp, ax, ay, az, pot);
                }
            }
            if(COUNT)
                cyc_b += clock64() - cyc_t0;
            guard = 0;
        } while(__ballot(sp > 0) != 0); // a list filled up: keep walking

        ax += bx;
        ay += by;
        az += bz;
        pot += potb;
        // reduce the partial sums over the 8 lanes of the group
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
                atomicAdd(&io.counters[7], (unsigned long long)cyc_a);
                atomicAdd(&io.counters[8], (unsigned long long)cyc_b);
            }
        }
    }
}

template <bool POT, bool COUNT, bool FASTWRAP>
static void launch_coop_t(const TreeView &tv, const GravParams &gp, const WalkIO &io, WalkScratch &ws, hipStream_t st)
{
    if(io.ntargets == 0)
        return;
    auto kern = k_grav_walk_coop<POT, COUNT, FASTWRAP>;
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
    ws.list.reserve((size_t)nblocks * 4 * 8 * ws.cap);
    ws.ctr.reserve(16);
    MPG_HIP(hipMemsetAsync(ws.ctr.p, 0, 16 * sizeof(unsigned), st));
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(256), 0, st, tv, gp, io, ws.list.p, ws.cap, ws.ctr.p + 8);
    MPG_HIP(hipGetLastError());
}

unsigned walk_coop_error(WalkScratch &ws, hipStream_t st)
{
    if(!ws.ctr.p)
        return 0;
    unsigned e = 0;
    MPG_HIP(hipMemcpyAsync(&e, ws.ctr.p + 8, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    MPG_HIP(hipStreamSynchronize(st));
    return e;
}

void launch_grav_walk_coop(const TreeView &tv, const GravParams &gp, const WalkIO &io, bool want_pot, bool count, bool fastwrap,
                           WalkScratch &ws, hipStream_t st)
{
#define MPG_WC(P, C)                                       \
    do {                                                   \
        if(fastwrap)                                       \
            launch_coop_t<P, C, true>(tv, gp, io, ws, st); \
        else                                               \
            launch_coop_t<P, C, false>(tv, gp, io, ws, st);\
    } while(0)
    if(want_pot) {
        if(count)
            MPG_WC(true, true);
        else
            MPG_WC(true, false);
    }
    else {
        if(count)
            MPG_WC(false, true);
        else
            MPG_WC(false, false);
    }
#undef MPG_WC
}

} // namespace mpg
