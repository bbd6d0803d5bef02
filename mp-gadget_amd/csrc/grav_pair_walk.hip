// grav_pair_walk.hip -- grav_short_pair (libgadget/gravshort-pair.c:21-120): the exact pair-wise short-range force within the
// sphere of radius Rcut, which runtests.c (:131) compares the tree force against.  The reference runs it as a neighbour
// iteration (treewalk_visit_ngbiter with Hsml = Rcut, all particle types, asymmetric); here the group-cooperative neighbour
// search of ngb_walk.h finds the leaves, lane s takes particle s of a leaf, and a particle with r^2 <= Rcut^2 contributes
// through the same softening spline and tabulated window as the tree kernel (apply_accn_to_output and
// grav_short_pair_ngbiter are the same arithmetic; Acc += -dist * fac with dist = I.Pos - P[other].Pos).
#include "grav_walk.h"
#include "grav_pair.h"
#include "ngb_walk.h"

namespace mpg {

template <bool POT>
__global__ void __launch_bounds__(256) k_grav_short_pair(const TreeView tv, const GravParams gp, const WalkIO io, const double rcut_abs,
                                                         unsigned *__restrict__ err)
{
    __shared__ WTab s_wf[NTAB];
    __shared__ WTab s_wp[POT ? NTAB : 1];
    __shared__ unsigned s_stack[4 * 8 * SPH_STK];
    __shared__ unsigned s_llist[4 * 8 * SPH_LCAP];
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
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 3, s = lane & 7, gshift = grp * 8;
    unsigned *stack = s_stack + ((threadIdx.x >> 6) * 8 + grp) * SPH_STK;
    unsigned *llist = s_llist + ((threadIdx.x >> 6) * 8 + grp) * SPH_LCAP;
    const int64_t q = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + grp;
    const bool valid = q < io.ntargets;
    int ci = 0;
    double px = 0, py = 0, pz = 0;
    if(valid) {
        ci = io.targets ? io.targets[q] : tv.order[q];
        px = io.pos[3 * (int64_t)ci];
        py = io.pos[3 * (int64_t)ci + 1];
        pz = io.pos[3 * (int64_t)ci + 2];
    }
    const double rcut2 = rcut_abs * rcut_abs;
    double ax = 0, ay = 0, az = 0, pot = 0;
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
            nl = walk_stepk<false, 2, true>(tv, tv.geoB, nullptr, stack, sp, go, s, gshift, rcut_abs, px, py, pz, llist, nl, overflow); // (two child ranges per step: ngb_walk.h)
            if(ballot64(overflow) != 0)
                break;
        }
        if(ballot64(overflow) != 0)
            break;
        for(int it = 0;; it++) { // phase B: every group takes its next leaf; lane s <-> particle s
            const bool has = it < nl;
            if(ballot64(has) == 0)
                break;
            const unsigned e = has ? llist[it] : 0u;
            const int ps = (int)(e >> 4), pc = (int)(e & 15u);
            if(s < pc) {
                const Src4 o = tv.src[ps + s];
                // dist = I.Pos - P[other].Pos (treewalk.c:968-975); pair_force takes source - target = -dist
                const double d0 = nearest_img(px - o.x, tv.box, 1.0 / tv.box);
                const double d1 = nearest_img(py - o.y, tv.box, 1.0 / tv.box);
                const double d2 = nearest_img(pz - o.z, tv.box, 1.0 / tv.box);
                if(d0 * d0 + d1 * d1 + d2 * d2 <= rcut2)
                    pair_force<POT>(o, -d0, -d1, -d2, gp, s_wf, s_wp, ax, ay, az, pot);
            }
        }
        if(ballot64(sp > 0) == 0)
            break;
    }
    if(ballot64(overflow) != 0) {
        if(lane == 0)
            atomicExch(err, 1u);
        return;
    }
    ax = group_sum(ax);
    ay = group_sum(ay);
    az = group_sum(az);
    if(POT)
        pot = group_sum(pot);
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
}

void launch_grav_short_pair(const TreeView &tv, const GravParams &gp, const WalkIO &io, double rcut_abs, bool want_pot, unsigned *d_err,
                            hipStream_t st)
{
    if(io.ntargets == 0)
        return;
    const dim3 grid((unsigned)((io.ntargets + 31) / 32)), block(256);
    if(want_pot)
        hipLaunchKernelGGL(k_grav_short_pair<true>, grid, block, 0, st, tv, gp, io, rcut_abs, d_err);
    else
        hipLaunchKernelGGL(k_grav_short_pair<false>, grid, block, 0, st, tv, gp, io, rcut_abs, d_err);
    MPG_HIP(hipGetLastError());
}

} // namespace mpg
