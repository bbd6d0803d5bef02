// grav_walk.hip -- short-range TreePM gravity walk (force_treeev_shortrange, libgadget/gravshort-tree.c:253-379)
// for gfx950, fp64.
//
// Semantics are the reference's, per target: walk the node list through `sibling` / first child; a node is
// discarded (gravshort-tree.c:198-215), used unopened (apply_accn_to_output with the node's mass at its centre of
// mass, :158-193) or opened (:220-241); the particles of every opened leaf interact with the target through the
// Gadget-2 softening spline and the tabulated short-range window (gravity.c:54-66).  Every lane takes exactly the
// decisions the reference takes for its target, so the set of interactions per particle is the reference's; only
// the summation order differs.
//
// Mapping to the hardware (MI355X: 64-wide waves, 4 SIMD/CU, fp64 FMA 16 lanes/clk/SIMD):
//   * one lane = one target, targets in tree (Morton) order, so the 64 lanes of a wave walk almost the same nodes
//     and their loads of node / particle records (32-byte Src4, 32-byte NodeGeo, 16-byte NodeLink) collapse onto a
//     few cache lines served by L1/L2; the tree of a 256^3 run (0.4 GB) lives in HBM + the 256 MB Infinity Cache;
//   * "while-while" traversal: a NODE phase advances every lane that has no pending work until it either opens a
//     leaf or accepts a node -- both become a pending source range [ps, ps+pc) of Src4 records (a node used
//     unopened is the 1-element range holding its moments) -- and an INTERACTION phase then evaluates the
//     pending ranges of all lanes together.  The expensive phase (>= 45 fp64 instructions per pair) therefore runs
//     with nearly all lanes busy, instead of idling the lanes whose opening decision differs;
//   * the two window tables live in LDS as (T[t], T[t+1]) double pairs: one ds_read_b128 per lookup;
//   * blockIdx is remapped so that each XCD (private 4 MB L2) owns one contiguous eighth of the Morton-ordered
//     targets.
// The kernel is bounded by fp64 VALU issue, not by HBM; MFMA is not applicable (pairwise
// 1/r^2 with a per-pair table lookup is not a contraction).
#include "grav_walk.h"
#include "grav_pair.h"

namespace mpg {

__device__ __forceinline__ double nearest_img(double x, double box, double invbox)
{
    // NEAREST(x, Box) of partmanager.h:99 for |x| <= Box: x - Box*rint(x/Box) (x -+ Box is exact there).
    return x - box * rint(x * invbox);
}

template <bool POT, bool COUNT, bool FASTWRAP, int THRESH>
__global__ void __launch_bounds__(256, 6) k_grav_walk(const TreeView tv, const GravParams gp, const WalkIO io)
{
    __shared__ WTab s_wf[NTAB];
    __shared__ WTab s_wp[POT ? NTAB : 1];
    for(int i = threadIdx.x; i < NTAB - 1; i += blockDim.x) {
        s_wf[i] = WTab{(double)io.tab_force[i], (double)io.tab_force[i + 1]};
        if(POT)
            s_wp[i] = WTab{(double)io.tab_pot[i], (double)io.tab_pot[i + 1]};
    }
    if(threadIdx.x == 0) {
        s_wf[NTAB - 1] = WTab{0, 0};
        if(POT)
            s_wp[NTAB - 1] = WTab{0, 0};
    }
    __syncthreads();

    // XCD-aware remap: hardware places block b on XCD b % 8; give each XCD a contiguous range of targets.
    const unsigned nb = gridDim.x;
    const unsigned per = (nb + 7) / 8;
    const unsigned lb = (blockIdx.x % 8) * per + blockIdx.x / 8;
    const int64_t slot = (int64_t)lb * blockDim.x + threadIdx.x;
    const bool valid = slot < io.ntargets;

    int ci = -1;
    double px = 0, py = 0, pz = 0, aold = 0;
    if(valid) {
        ci = io.targets ? io.targets[slot] : tv.order[slot];
        px = io.pos[3 * (int64_t)ci + 0];
        py = io.pos[3 * (int64_t)ci + 1];
        pz = io.pos[3 * (int64_t)ci + 2];
        double old;
        if(io.oldacc)
            old = io.oldacc[ci];
        else if(io.prev_accel) {
            // grav_get_abs_accel, gravshort.h:70-80
            double s2 = 0;
            for(int j = 0; j < 3; j++) {
                const double a = io.prev_accel[3 * (int64_t)ci + j] + (io.gravpm ? io.gravpm[3 * (int64_t)ci + j] : 0.0);
                s2 += a * a;
            }
            old = sqrt(s2) / gp.G;
        }
        else
            old = 0;
        aold = gp.errtol * old;
    }

    const int64_t npart = tv.npart;
    int no = valid ? 0 : -1;
    int ps = 0, pc = 0;
    double spx = px, spy = py, spz = pz; // target shifted to the periodic image of the pending range (FASTWRAP)
    double ax = 0, ay = 0, az = 0, pot = 0;
    unsigned n_pp = 0, n_vis = 0, n_used = 0;

    for(;;) {
        // ---- node phase: advance the lanes that have nothing pending
        for(;;) {
            const bool walking = (pc == 0) && (no >= 0);
            const unsigned long long wm = __ballot(walking);
            if(wm == 0)
                break;
            if(__popcll(wm) < THRESH && __ballot(pc > 0) != 0)
                break;
            if(walking) {
                const NodeGeo g = tv.geo[no];
                const Src4 mom = tv.src[npart + no];
                const NodeLink lk = tv.link[no];
                if(COUNT)
                    n_vis++;
                // periodic image of this node relative to the target: k = rint((c - p)/Box) per axis
                const double kx = rint((g.cx - px) * gp.invbox), ky = rint((g.cy - py) * gp.invbox), kz = rint((g.cz - pz) * gp.invbox);
                double qx = fma(kx, gp.box, px), qy = fma(ky, gp.box, py), qz = fma(kz, gp.box, pz);
                const double cdx = fabs(g.cx - qx), cdy = fabs(g.cy - qy), cdz = fabs(g.cz - qz);
                double dx = mom.x - qx, dy = mom.y - qy, dz = mom.z - qz;
                if(!FASTWRAP || g.len * 4.0 > gp.box) {
                    // (top levels only when FASTWRAP) centre of mass and geometric centre may sit on different periodic
                    // images: take NEAREST(cofm - pos) exactly as gravshort-tree.c:299-300 does
                    const double jx = rint((mom.x - px) * gp.invbox), jy = rint((mom.y - py) * gp.invbox), jz = rint((mom.z - pz) * gp.invbox);
                    qx = fma(jx, gp.box, px);
                    qy = fma(jy, gp.box, py);
                    qz = fma(jz, gp.box, pz);
                    dx = fma(-jx, gp.box, mom.x - px);
                    dy = fma(-jy, gp.box, mom.y - py);
                    dz = fma(-jz, gp.box, mom.z - pz);
                }
                const double r2 = dx * dx + dy * dy + dz * dz;
                // shall_we_discard_node, gravshort-tree.c:198-215
                const double eff = gp.rcut + 0.5 * g.len;
                const bool discard = (r2 > gp.rcut2) && (cdx > eff || cdy > eff || cdz > eff);
                if(discard) {
                    no = lk.sibling;
                }
                else {
                    // shall_we_open_node, gravshort-tree.c:220-241
                    const double l2 = g.len * g.len;
                    const double inside = 0.6 * g.len;
                    const bool open = ((!gp.use_bh) && (mom.m * l2 > r2 * r2 * aold)) || (l2 > r2 * gp.bhangle2) ||
                                      (cdx < inside && cdy < inside && cdz < inside);
                    if(!open) {
                        // node used unopened: its moments are a 1-element source range
                        ps = (int)(npart + no);
                        pc = 1;
                        spx = qx;
                        spy = qy;
                        spz = qz;
                        no = lk.sibling;
                        if(COUNT)
                            n_used++;
                    }
                    else if(lk.pcount > 0) {
                        ps = lk.pstart;
                        pc = lk.pcount;
                        spx = qx;
                        spy = qy;
                        spz = qz;
                        no = lk.sibling;
                        if(COUNT)
                            n_pp += lk.pcount;
                    }
                    else
                        no = no + 1; // first child (depth-first pre-order layout)
                }
            }
        }
        // ---- interaction phase
        if(__ballot(pc > 0) == 0)
            break;
#pragma unroll 1
        for(int k = 0; k < NMAXCHILD; k++) {
            const bool has = k < pc;
            if(__ballot(has) == 0)
                break;
            if(has) {
                const Src4 sc = tv.src[ps + k];
                double dx, dy, dz;
                if(FASTWRAP) {
                    dx = sc.x - spx;
                    dy = sc.y - spy;
                    dz = sc.z - spz;
                }
                else {
                    dx = nearest_img(sc.x - px, gp.box, gp.invbox);
                    dy = nearest_img(sc.y - py, gp.box, gp.invbox);
                    dz = nearest_img(sc.z - pz, gp.box, gp.invbox);
                }
                pair_force<POT>(sc, dx, dy, dz, gp, s_wf, s_wp, ax, ay, az, pot);
            }
        }
        pc = 0;
    }

    if(valid) {
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
        unsigned long long c0 = n_pp, c1 = n_vis, c2 = n_used;
        for(int off = 32; off > 0; off >>= 1) {
            c0 += __shfl_down(c0, off);
            c1 += __shfl_down(c1, off);
            c2 += __shfl_down(c2, off);
        }
        if((threadIdx.x & 63) == 0) {
            atomicAdd(&io.counters[0], c0);
            atomicAdd(&io.counters[1], c1);
            atomicAdd(&io.counters[2], c2);
        }
    }
}

template <bool POT, bool COUNT, bool FASTWRAP> static void launch_t(const TreeView &tv, const GravParams &gp, const WalkIO &io, int thresh, hipStream_t st)
{
    const int64_t nb = (io.ntargets + 255) / 256;
    if(nb == 0)
        return;
    // grid rounded up to a multiple of 8 so the XCD remap is a bijection onto [0, 8*per)
    const unsigned per = (unsigned)((nb + 7) / 8);
    dim3 grid(per * 8), block(256);
    switch(thresh) {
    case 1: hipLaunchKernelGGL((k_grav_walk<POT, COUNT, FASTWRAP, 1>), grid, block, 0, st, tv, gp, io); break;
    case 8: hipLaunchKernelGGL((k_grav_walk<POT, COUNT, FASTWRAP, 8>), grid, block, 0, st, tv, gp, io); break;
    case 16: hipLaunchKernelGGL((k_grav_walk<POT, COUNT, FASTWRAP, 16>), grid, block, 0, st, tv, gp, io); break;
    case 32: hipLaunchKernelGGL((k_grav_walk<POT, COUNT, FASTWRAP, 32>), grid, block, 0, st, tv, gp, io); break;
    case 48: hipLaunchKernelGGL((k_grav_walk<POT, COUNT, FASTWRAP, 48>), grid, block, 0, st, tv, gp, io); break;
    default: hipLaunchKernelGGL((k_grav_walk<POT, COUNT, FASTWRAP, 24>), grid, block, 0, st, tv, gp, io); break;
    }
    MPG_HIP(hipGetLastError());
}

void launch_grav_walk(const TreeView &tv, const GravParams &gp, const WalkIO &io, bool want_pot, bool count, bool fastwrap, int thresh,
                      hipStream_t st)
{
#define MPG_W1(P, C)                                          \
    do {                                                      \
        if(fastwrap)                                          \
            launch_t<P, C, true>(tv, gp, io, thresh, st);     \
        else                                                  \
            launch_t<P, C, false>(tv, gp, io, thresh, st);    \
    } while(0)
    if(want_pot) {
        if(count)
            MPG_W1(true, true);
        else
            MPG_W1(true, false);
    }
    else {
        if(count)
            MPG_W1(false, true);
        else
            MPG_W1(false, false);
    }
#undef MPG_W1
}

} // namespace mpg
