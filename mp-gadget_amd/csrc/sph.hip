// sph.hip -- SPH density (with the smoothing-length iteration) and hydro force for gfx950, fp64.
//
// Reference: libgadget/density.c (density_ngbiter :424-519, density_postprocess :532-586, density_check_neighbours
// :589-689, set_init_hsml :691-749), libgadget/hydra.c (hydro_copy :247-277, hydro_ngbiter :318-506, hydro_postprocess
// :514-528), libgadget/densitykernel.c, and the neighbour visitors of libgadget/treewalk.c (cull_node :1015-1042,
// treewalk_visit_nolist_ngbiter :1152-1265, treewalk_visit_ngbiter :930-1007, treewalk_do_hsml_loop :1269-1367).
//
// Mapping: one lane per target, targets in tree (Morton) order, walking the depth-first node arrays of the gas tree
// (sibling / first child = node+1).  A target needs ~100 neighbours from ~10 leaves, four orders of magnitude less work
// than the gravity walk, so the simple mapping is kept: neighbouring lanes visit the same leaves and their loads
// coalesce.  All per-source quantities that the reference predicts lazily per neighbour (VelPred, EntVarPred,
// PressurePred, density prediction, the Balsara factor f2; hydra.c:379-416) are deterministic functions of the source
// alone and are precomputed once per call into tree-ordered records.
// Kernels are HBM/L2-streaming with ~150 flop per neighbour; MFMA does not apply.
#include "sph.h"
#include "ngb_walk.h"
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace mpg {

#define NUMDIMS 3
#define NORM_COEFF 4.188790204786 // densitykernel.h:6
#define SPH_GAMMA (5.0 / 3.0)     // physconst.h:35
#define SPH_GAMMA_MINUS1 (SPH_GAMMA - 1)


struct DKernel { // DensityKernel, densitykernel.h:23-33
    double H, HH, Hinv, Wknorm, dWknorm, support;
};

__device__ __forceinline__ double p2(double x) { return x * x; }
__device__ __forceinline__ double p3(double x) { return x * x * x; }
__device__ __forceinline__ double p4(double x) { return (x * x) * (x * x); }
__device__ __forceinline__ double p5(double x) { return (x * x) * (x * x) * x; }

__device__ __forceinline__ double ksupport(int type) { return type == 0 ? 2. : (type == 1 ? 3. : 2.5); }
__device__ __forceinline__ double ksigma3(int type) { return type == 0 ? 1 / M_PI : (type == 1 ? 1 / (120 * M_PI) : 1 / (20 * M_PI)); }

__device__ __forceinline__ DKernel kernel_init(double H, int type) // densitykernel.c:136-153
{
    DKernel k;
    k.H = H;
    k.HH = H * H;
    k.Hinv = 1. / H;
    k.support = ksupport(type);
    const double hinv = k.Hinv * k.support;
    k.Wknorm = ksigma3(type) * p3(hinv);
    k.dWknorm = k.Wknorm * hinv;
    return k;
}

// The kernel polynomials of densitykernel.c:24-90 without branches on q: every term (c - q)^n of the reference's piecewise form is taken
// of max(c - q, 0), which is the term where the reference has it and an exact zero (added or subtracted last, in the reference's order:
// the sums' bits do not change) where it has not.  The lanes of a wave hold neighbours at all distances: with three branches the wave ran
// all three bodies one after the other (round 4: the SPH kernels are issue-bound, profiles/r04a_experiments).
__device__ __forceinline__ double pos_part(double x) { return fmax(x, 0.0); }
template <int TYPE> __device__ __forceinline__ double wk_q(double q) // densitykernel.c:24-90
{
    if(TYPE == 0)
        return 0.25 * p3(pos_part(2 - q)) - p3(pos_part(1 - q));
    else if(TYPE == 1)
        return p5(pos_part(3 - q)) - 6 * p5(pos_part(2 - q)) + 15 * p5(pos_part(1 - q));
    else
        return p4(pos_part(2.5 - q)) - 5 * p4(pos_part(1.5 - q)) + 10 * p4(pos_part(0.5 - q));
}
template <int TYPE> __device__ __forceinline__ double dwk_q(double q)
{
    if(TYPE == 0)
        return -0.25 * 3 * p2(pos_part(2 - q)) + 3 * p2(pos_part(1 - q));
    else if(TYPE == 1)
        return -5 * p4(pos_part(3 - q)) + 30 * p4(pos_part(2 - q)) - 75 * p4(pos_part(1 - q));
    else
        return -4 * p3(pos_part(2.5 - q)) + 20 * p3(pos_part(1.5 - q)) - 40 * p3(pos_part(0.5 - q));
}
__device__ __forceinline__ double kernel_wk(const DKernel &k, int type, double u)
{
    const double q = u * k.support;
    return k.Wknorm * (type == 0 ? wk_q<0>(q) : (type == 1 ? wk_q<1>(q) : wk_q<2>(q)));
}
__device__ __forceinline__ double kernel_dwk(const DKernel &k, int type, double u)
{
    const double q = u * k.support;
    return k.dWknorm * (type == 0 ? dwk_q<0>(q) : (type == 1 ? dwk_q<1>(q) : dwk_q<2>(q)));
}

#ifndef SPH_EXACT_DIV
#define SPH_FAST_DIV // round 4: the quotients and the square root of the pair evaluations by reciprocals (k_hydro 10.1 -> 9.6 ms)
#endif
// 1 / x and 1 / sqrt(x) to within an ulp: v_rcp_f64 / v_rsq_f64 and Newton steps instead of the ~30-instruction IEEE division and
// square-root expansions (the reference itself is built with -ffast-math).  x > 0 and finite.
__device__ __forceinline__ double rcp_fast(const double x)
{
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    return fma(fma(-x, y, 1.0), y, y);
}
__device__ __forceinline__ double rsqrt_fast(const double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);
}

// SPH_VelPred, density.c:91-100
__device__ __forceinline__ void vel_pred(const SphView &A, const mpg_sph_times &T, int64_t i, double v[3])
{
    const int bg = A.tb_grav ? A.tb_grav[i] : 0, bh = A.tb_hydro ? A.tb_hydro[i] : 0;
    for(int j = 0; j < 3; j++)
        v[j] = A.vel[3 * i + j] + T.gravkicks[bg] * (A.gacc ? A.gacc[3 * i + j] : 0.0) + (A.gpm ? A.gpm[3 * i + j] : 0.0) * T.FgravkickB +
               T.hydrokicks[bh] * (A.hydroacc_in ? A.hydroacc_in[3 * i + j] : 0.0);
}

// SPH_EntVarPred, density.c:69-86
__device__ __forceinline__ double ent_var_pred(const SphView &A, const mpg_sph_times &T, int64_t i)
{
    const int bin = A.tb_hydro ? A.tb_hydro[i] : 0;
    double e = A.entropy[i] + (A.dtentropy_in ? A.dtentropy_in[i] : 0.0) * T.dloga_kick[bin];
    if(e < 0.05 * A.entropy[i])
        e = 0.05 * A.entropy[i];
    if(e <= 0)
        return 0;
    return exp(1. / SPH_GAMMA * log(e));
}

// tree-ordered per-source record of the density loop: predicted velocity + predicted entropy
__global__ void __launch_bounds__(256) k_sph_predict(int64_t npart, const int *__restrict__ order, const SphView A, const mpg_sph_times T,
                                                     Aux4 *__restrict__ aux, double *__restrict__ entvarpred_caller)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= npart)
        return;
    const int64_t ci = order[k];
    double v[3];
    vel_pred(A, T, ci, v);
    const double e = ent_var_pred(A, T, ci);
    aux[k] = Aux4{v[0], v[1], v[2], e};
    entvarpred_caller[ci] = e;
}



#ifndef SPH_WALK_K
#define SPH_WALK_K 2 // child ranges per search step (walk_stepk, ngb_walk.h)
#endif

#ifndef SPH_MERGE
#define SPH_MERGE true // contiguous opened leaves of a child range joined into one list entry (walk_stepk, ngb_walk.h)
#endif

#ifndef SPH_NE
// runs of <= 8 particles per search leaf: with -DSPH_NE=2 (or 4) the searches stop at nodes of <= 16 (32) particles and list their whole
// particle range (TreeBuilder::calc_search_links; MPG_SPH_LEAF_CAP picks a smaller capacity at run time).  Measured at 2 x 128^3 on the
// Zel'dovich set (profiles/r05a_experiments): fewer search steps, but 20 - 34 % more candidates and as many test iterations - density
// 7.27 -> 7.32 ms, hydro 7.51 -> 7.41 ms - so the default stays 1 (the reference's leaves).
#define SPH_NE 1
#endif

struct DensAcc {
    double EgyRho = 0, DhsmlEgy = 0, Rho = 0, DhsmlDensity = 0, Ngb = 0, Div = 0, Rot0 = 0, Rot1 = 0, Rot2 = 0, G0 = 0, G1 = 0, G2 = 0;
};

// Candidate handling is split in two so that the expensive part runs with full lanes: every candidate of an opened leaf
// gets the distance test (treewalk.c:1218-1232; cheap, ~1/3 pass), the survivors are compacted into a small per-group
// buffer in LDS, and the kernel evaluation (density_ngbiter / hydro_ngbiter) is run 8 survivors at a time.
#ifndef SPH_TRIG
// survivors in some group of the wave that trigger an evaluation (<= SPH_CBUF - 8: the next append must fit).  24 since round 4 (16 before): the
// later the trigger, the more of the wave's 8 groups hold a full 8 survivors when it comes (k_hydro 8.4 -> 7.9 ms; k_density unchanged; 8: 10.2)
#define SPH_TRIG 24
#endif
constexpr int SPH_CBUF = 32; // survivor slots per group (a ring: a power of two)

// distance test of density: returns whether the kernel evaluation is needed; counts the reference's "ninteractions"
template <bool WRAP>
__device__ __forceinline__ bool density_test(const Src4 s, const double px, const double py, const double pz, const double h2, const double HH,
                                             const double box, unsigned &n_int)
{
    // the distance vector points to 'other': I.Pos - P[other].Pos (treewalk.c:1218-1225)
    const double d0 = near_img<WRAP>(px - s.x, box, 1.0 / box);
    const double d1 = near_img<WRAP>(py - s.y, box, 1.0 / box);
    const double d2 = near_img<WRAP>(pz - s.z, box, 1.0 / box);
    const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
    if(r2 > h2)
        return false;
    n_int++;
    return r2 < HH;
}

// density_ngbiter for one neighbour inside the kernel, density.c:451-518
template <bool WRAP>
__device__ __forceinline__ void density_eval(const Src4 s, const Aux4 o, const double px, const double py, const double pz, const DKernel &kern,
                                             const double kvol, const double *ivel, const DensityCtl &C, const double box, DensAcc &a)
{
    const double d0 = near_img<WRAP>(px - s.x, box, 1.0 / box);
    const double d1 = near_img<WRAP>(py - s.y, box, 1.0 / box);
    const double d2 = near_img<WRAP>(pz - s.z, box, 1.0 / box);
    const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
#ifdef SPH_FAST_DIV
    const double rinv_d = r2 > 0 ? rsqrt_fast(r2) : 0.0;
    const double r = r2 * rinv_d;
#else
    const double r = sqrt(r2);
#endif
    const double u = r * kern.Hinv;
    const double wk = kernel_wk(kern, C.ktype, u);
    a.Ngb += wk * kvol;
    const double dwk = kernel_dwk(kern, C.ktype, u);
    const double mass_j = s.m;
    a.Rho += mass_j * wk;
    const double density_dW = -(NUMDIMS * kern.Hinv * wk + u * dwk);
    a.DhsmlDensity += mass_j * density_dW;
    if(C.DoEgyDensity) {
        a.EgyRho += mass_j * o.w * wk;
        a.DhsmlEgy += mass_j * o.w * density_dW;
    }
    if(r > 0) {
#ifdef SPH_FAST_DIV
        const double fac = mass_j * dwk * rinv_d;
#else
        const double fac = mass_j * dwk / r;
#endif
        const double dv0 = ivel[0] - o.x, dv1 = ivel[1] - o.y, dv2 = ivel[2] - o.z;
        a.Div += -fac * (d0 * dv0 + d1 * dv1 + d2 * dv2);
        a.Rot0 += fac * (dv1 * d2 - d1 * dv2); // crossproduct(dv, dist), densitykernel.h:63-76
        a.Rot1 += fac * (dv2 * d0 - d2 * dv0);
        a.Rot2 += fac * (dv0 * d1 - d0 * dv1);
        a.G0 += fac * d0;
        a.G1 += fac * d1;
        a.G2 += fac * d2;
    }
}

// The survivor buffer of a group is a ring of SPH_CBUF slots (head, cnt group-uniform: no entries are moved after an evaluation).
// Appends this lane's survivor (if any); returns the new fill level.
__device__ __forceinline__ int cbuf_push(int *cbuf, const int head, int cnt, const bool keep, const int sidx, const int s, const int gshift)
{
    const unsigned gm = (unsigned)((ballot64(keep) >> gshift) & 0xffull);
    if(keep)
        cbuf[(head + cnt + __popc(gm & ((1u << s) - 1u))) & (SPH_CBUF - 1)] = sidx;
    return cnt + __popc(gm);
}

// The end of a density pass for one target (the 8 lanes of its group hold partial sums): density_reduce + density_postprocess +
// density_check_neighbours (density.c:374-409, 532-689), the append of an unfinished target to `redo`, the pass statistics.
__device__ __forceinline__ void density_finish(const TreeView &tv, const SphView &A, const mpg_density_params &P, const DensityCtl &C, DensAcc &a,
                                               const bool valid, const int s, const int lane, const int i, const int ty, const double hsml,
                                               int *__restrict__ redo, unsigned *__restrict__ nredo, unsigned long long *__restrict__ stats,
                                               const unsigned n_int, const unsigned n_cand)
{
    // sum over the 8 lanes of the group
    a.EgyRho = group_sum(a.EgyRho);
    a.DhsmlEgy = group_sum(a.DhsmlEgy);
    a.Rho = group_sum(a.Rho);
    a.DhsmlDensity = group_sum(a.DhsmlDensity);
    a.Ngb = group_sum(a.Ngb);
    a.Div = group_sum(a.Div);
    a.Rot0 = group_sum(a.Rot0);
    a.Rot1 = group_sum(a.Rot1);
    a.Rot2 = group_sum(a.Rot2);
    a.G0 = group_sum(a.G0);
    a.G1 = group_sum(a.G1);
    a.G2 = group_sum(a.G2);
    bool notdone = false;
    if(valid && s == 0) {
        const double EgyRho = a.EgyRho, DhsmlEgy = a.DhsmlEgy, Rho = a.Rho, DhsmlDensity = a.DhsmlDensity, Ngb = a.Ngb, Div = a.Div;
        const double Rot0 = a.Rot0, Rot1 = a.Rot1, Rot2 = a.Rot2, G0 = a.G0, G1 = a.G1, G2 = a.G2;
        // ---- density_reduce (PRIMARY: assign), density.c:374-409
        C.NumNgb[i] = Ngb;
        double dhsmlfac = DhsmlDensity;
        A.density[i] = Rho;
        double divvel = Div;
        if(ty == 0 && A.gradrho) {
            A.gradrho[3 * (int64_t)i] = G0;
            A.gradrho[3 * (int64_t)i + 1] = G1;
            A.gradrho[3 * (int64_t)i + 2] = G2;
        }
        // ---- density_postprocess, density.c:532-586
        dhsmlfac *= hsml / (NUMDIMS * Rho);
        dhsmlfac = 1 / (1 + dhsmlfac);
        double newh = hsml;
        bool done = true;
        if(C.update_hsml) {
            // density_check_neighbours, density.c:589-689
            double desnumngb = C.DesNumNgb;
            if(C.BlackHoleOn && ty == 5)
                desnumngb = desnumngb * P.BlackHoleNgbFactor;
            double L = C.Left[i], R = C.Right[i];
            if(Ngb < (desnumngb - P.MaxNumNgbDeviation) || (Ngb > (desnumngb + P.MaxNumNgbDeviation))) {
                done = false;
                if((R - L) < 1.0e-5 * L) {
                    newh = R;
                    done = true;
                }
                else {
                    if(Ngb < desnumngb)
                        L = hsml;
                    else
                        R = hsml;
                    if((R < tv.box && L > 0) || (hsml * 1.26 > 0.99 * tv.box))
                        newh = cbrt(0.5 * (p3(L) + p3(R)));
                    else {
                        double fac = 1.26;
                        if(Ngb > 0)
                            fac = 1 - (Ngb - desnumngb) / (NUMDIMS * Ngb) * dhsmlfac;
                        if(R > 0.99 * tv.box && L > 0)
                            if(dhsmlfac <= 0 || fabs(Ngb - desnumngb) >= 0.5 * desnumngb || fac > 1.26)
                                fac = 1.26;
                        if(R < 0.99 * tv.box && L == 0)
                            if(dhsmlfac <= 0 || fac < 1. / 3)
                                fac = 1. / 3;
                        newh = hsml * fac;
                    }
                    if(C.BlackHoleOn && ty == 5 && L > P.BlackHoleMaxAccretionRadius) {
                        newh = P.BlackHoleMaxAccretionRadius;
                        done = true;
                    }
                    else if(R < C.MinGasHsml) {
                        newh = C.MinGasHsml;
                        done = true;
                    }
                }
                C.Left[i] = L;
                C.Right[i] = R;
            }
            else {
                if(C.BlackHoleOn && ty == 5 && newh > P.BlackHoleMaxAccretionRadius)
                    newh = P.BlackHoleMaxAccretionRadius;
                if(newh < C.MinGasHsml)
                    newh = C.MinGasHsml;
            }
            A.hsml[i] = newh;
            notdone = !done;
        }
        if(ty == 0) {
            if(C.DoEgyDensity) {
                const double EntPred = C.entvarpred[i];
                double egyfac = DhsmlEgy;
                egyfac *= newh / (NUMDIMS * EgyRho);
                egyfac *= -dhsmlfac;
                A.dhsmlegyfac[i] = egyfac;
                A.egywtdensity[i] = EgyRho / EntPred;
            }
            else
                A.dhsmlegyfac[i] = dhsmlfac;
            A.curlvel[i] = sqrt(Rot0 * Rot0 + Rot1 * Rot1 + Rot2 * Rot2) / Rho;
            divvel /= Rho;
            A.divvel[i] = divvel;
            if(A.dthsml)
                A.dthsml[i] = (1.0 / NUMDIMS) * divvel * newh;
        }
        else {
            divvel /= Rho;
            A.divvel[i] = divvel;
            if(A.dthsml)
                A.dthsml[i] = (1.0 / NUMDIMS) * divvel * newh;
        }
    }
    // wave-aggregated append of the unfinished targets (one atomic per wave)
    {
        const unsigned long long m = ballot64(notdone);
        if(m != 0) {
            unsigned basepos = 0;
            const int leader = __ffsll((long long)m) - 1;
            if(lane == leader)
                basepos = atomicAdd(nredo, (unsigned)__popcll(m));
            basepos = __shfl(basepos, leader);
            if(notdone)
                redo[basepos + __popcll(m & ((1ull << lane) - 1ull))] = i;
        }
    }
    // statistics: successful distance tests (the reference's ninteractions) and candidates tested
    unsigned long long c_int = n_int, c_cand = n_cand;
    for(int off = 32; off > 0; off >>= 1) {
        c_int += __shfl_down(c_int, off);
        c_cand += __shfl_down(c_cand, off);
    }
    if(lane == 0 && stats) {
        atomicAdd(&stats[0], c_int);
        atomicAdd(&stats[1], c_cand);
    }
}

// One density pass over the current queue: treewalk_visit_nolist_ngbiter + density_ngbiter + density_reduce +
// density_postprocess + density_check_neighbours.  Targets that are not done are appended to `redo`.
__global__ void __launch_bounds__(256, 4) k_density(const TreeView tv, const SphView A, const mpg_sph_times T, const mpg_density_params P,
                                                 const DensityCtl C, const Aux4 *__restrict__ aux, const int *__restrict__ queue,
                                                 int64_t nqueue, int *__restrict__ redo, unsigned *__restrict__ nredo,
                                                 unsigned long long *__restrict__ stats, unsigned *__restrict__ err)
{
    __shared__ unsigned s_stack[4 * 8 * SPH_STK];
    __shared__ int s_cbuf[4 * 8 * SPH_CBUF];
    __shared__ unsigned s_llist[4 * 8 * SPH_LCAP];
#ifdef MPG_EXP_LDSPAD_SPH // timing experiment: fewer resident blocks per CU with the same code (bytes of unused LDS)
    __shared__ unsigned s_pad[MPG_EXP_LDSPAD_SPH / 4];
    if(tv.box < 0)
        s_pad[threadIdx.x] = 1u, s_llist[0] = s_pad[(threadIdx.x + 1) & 255];
#endif
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 3, s = lane & 7, gshift = grp * 8;
    unsigned *stack = s_stack + ((threadIdx.x >> 6) * 8 + grp) * SPH_STK;
    int *cbuf = s_cbuf + ((threadIdx.x >> 6) * 8 + grp) * SPH_CBUF;
    unsigned *llist = s_llist + ((threadIdx.x >> 6) * 8 + grp) * SPH_LCAP;
    int cnt = 0, head = 0; // survivors waiting in the group's ring buffer, its first slot (group-uniform)
    const int64_t q = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + grp;
    const bool valid = q < nqueue;
    unsigned n_int = 0, n_cand = 0;
    int i = 0, ty = 0;
    double px = 0, py = 0, pz = 0, hsml = 0;
    double ivel[3] = {0, 0, 0};
    if(valid) {
        i = queue[q];
        ty = A.type ? (A.type[i] & 7) : 0;
        px = A.pos[3 * (int64_t)i];
        py = A.pos[3 * (int64_t)i + 1];
        pz = A.pos[3 * (int64_t)i + 2];
        if(ty != 0) { // density_copy, density.c:357-372
            ivel[0] = A.vel[3 * (int64_t)i];
            ivel[1] = A.vel[3 * (int64_t)i + 1];
            ivel[2] = A.vel[3 * (int64_t)i + 2];
        }
        else
            vel_pred(A, T, i, ivel);
        hsml = A.hsml[i];
    }
    // the search's geometry: the cubes around the nodes' particles when the tree carries them (TreeBuilder::calc_search_boxes), else the cells
    const NodeGeo *__restrict__ sgeo = tv.geoS ? tv.geoS : tv.geoB;
    const DKernel kern = kernel_init(valid ? hsml : 1.0, C.ktype);
    const double kvol = NORM_COEFF * p3(kern.H);
    const double h2 = hsml * hsml;
    DensAcc a;
    int sp = 0;
    if(valid) {
        if(s == 0)
            stack[0] = (0u << 4) | 1u; // the root
        sp = 1;
    }
    bool overflow = false;
    // the search and the pair loops, with (WRAP) or without NEAREST(): see interior_wave, ngb_walk.h
    auto loops = [&](auto wrap_tag) {
    constexpr bool WRAP = decltype(wrap_tag)::value;
    for(;;) {
        // ---- phase A: walk; opened leaves go to the group's list
        int nl = 0;
        for(;;) {
            const bool go = sp > 0 && nl + 8 * SPH_WALK_K * SPH_NE <= SPH_LCAP;
            if(ballot64(go) == 0)
                break;
            nl = walk_stepk<false, SPH_WALK_K, SPH_MERGE, WRAP, SPH_NE>(tv, sgeo, nullptr, stack, sp, go, s, gshift, hsml, px, py, pz, llist, nl, overflow, tv.linkS);
            if(ballot64(overflow) != 0)
                break;
        }
        if(ballot64(overflow) != 0)
            break;
        // ---- phase B: every group takes its next leaf; lane s <-> particle s
        // (the candidate of the NEXT leaf is requested before this one is tested: an iteration is a dependent LDS read -> gather -> test ->
        // LDS append chain, and 4 waves per SIMD do not hide the gather's latency.  Lanes beyond the leaf's count read its first particle.)
        unsigned e = (0 < nl) ? llist[0] : 0u;
        int ps = (int)(e >> 4), pc = (int)(e & 15u);
        Src4 cand = tv.src[ps + (s < pc ? s : 0)];
        for(int it = 0;; it++) {
            const bool has = it < nl;
            if(ballot64(has) == 0)
                break;
            const unsigned e_n = (it + 1 < nl) ? llist[it + 1] : 0u;
            const int ps_n = (int)(e_n >> 4), pc_n = (int)(e_n & 15u);
            const Src4 cand_n = tv.src[ps_n + (s < pc_n ? s : 0)];
            bool keep = false;
            if(s < pc) {
                n_cand++;
                keep = density_test<WRAP>(cand, px, py, pz, h2, kern.HH, tv.box, n_int);
            }
            cnt = cbuf_push(cbuf, head, cnt, keep, ps + s, s, gshift);
            if(ballot64(cnt >= SPH_TRIG) != 0) {
                if(cnt >= 8) {
                    const int sidx = cbuf[(head + s) & (SPH_CBUF - 1)];
                    density_eval<WRAP>(tv.src[sidx], aux[sidx], px, py, pz, kern, kvol, ivel, C, tv.box, a);
                    head = (head + 8) & (SPH_CBUF - 1);
                    cnt -= 8;
                }
            }
            cand = cand_n;
            ps = ps_n;
            pc = pc_n;
        }
        if(ballot64(sp > 0) == 0)
            break;
    }
    if(ballot64(overflow) != 0)
        return;
    while(ballot64(cnt > 0) != 0) { // drain the survivor buffers
        if(s < cnt) {
            const int sidx = cbuf[(head + s) & (SPH_CBUF - 1)];
            density_eval<WRAP>(tv.src[sidx], aux[sidx], px, py, pz, kern, kvol, ivel, C, tv.box, a);
        }
        head = (head + 8) & (SPH_CBUF - 1);
        cnt = cnt > 8 ? cnt - 8 : 0;
    }
    };
    if(interior_wave(valid, px, py, pz, hsml, tv.box))
        loops(std::false_type{});
    else
        loops(std::true_type{});
    if(ballot64(overflow) != 0) {
        if(lane == 0)
            atomicExch(err, 1u);
        return;
    }
    density_finish(tv, A, P, C, a, valid, s, lane, i, ty, hsml, redo, nredo, stats, n_int, n_cand);
}

// marks the active particles (caller indices) in a byte map
__global__ void __launch_bounds__(256) k_mark_active(int64_t nact, const int *__restrict__ active, uint8_t *__restrict__ flags)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k < nact)
        flags[active[k]] = 1;
}

// Work queue of a loop in TREE ORDER (consecutive entries are neighbours in space, so the 8 targets of a wave walk the same
// nodes and leaves): the active gas / black-hole particles of the tree.  INIT: also the per-target initialisation of
// density() (density.c:277-285).  flags == null: every particle is active.
template <bool INIT>
__global__ void __launch_bounds__(256) k_queue_treeorder(int64_t npart, const int *__restrict__ order, const uint8_t *__restrict__ flags,
                                                         const SphView A, const DensityCtl C, double box, bool gas_only,
                                                         int *__restrict__ queue, unsigned *__restrict__ nqueue)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool inb = k < npart;
    const int i = inb ? order[k] : 0;
    const bool act = inb && (!flags || flags[i]);
    if(INIT && act) {
        C.Right[i] = box;
        C.NumNgb[i] = 0;
        C.Left[i] = 0;
    }
    const int ty = (act && A.type) ? (A.type[i] & 7) : 0;
    const bool work = act && (ty == 0 || (!gas_only && ty == 5)); // density_haswork (density.c:521-530) / hydro_haswork
    // wave-aggregated append: one atomic per wave (same-address atomics serialise)
    const unsigned long long m = ballot64(work);
    unsigned basepos = 0;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    if(work && lane == leader)
        basepos = atomicAdd(nqueue, (unsigned)__popcll(m));
    basepos = __shfl(basepos, leader < 0 ? 0 : leader);
    if(work)
        queue[basepos + __popcll(m & ((1ull << lane) - 1ull))] = i;
}

// The black holes are targets of the density loop (density_haswork, density.c:521-530) but no neighbours: the search takes gas only
// (density.c:438).  When the tree was built for gas alone (run.c:466 builds GASMASK | BHMASK only to reuse the tree for the mergers)
// they are not among its particles, so they are appended to the queue from the particle table.  Type 7 = garbage / swallowed.
__global__ void __launch_bounds__(256) k_queue_blackholes(int64_t n, const uint8_t *__restrict__ flags, const SphView A, const DensityCtl C, double box,
                                                          int *__restrict__ queue, unsigned *__restrict__ nqueue)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool work = i < n && (!flags || flags[i]) && A.type && (A.type[i] & 7) == 5;
    if(work) {
        C.Right[i] = box;
        C.NumNgb[i] = 0;
        C.Left[i] = 0;
    }
    const unsigned long long m = ballot64(work);
    unsigned basepos = 0;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    if(work && lane == leader)
        basepos = atomicAdd(nqueue, (unsigned)__popcll(m));
    basepos = __shfl(basepos, leader < 0 ? 0 : leader);
    if(work)
        queue[basepos + __popcll(m & ((1ull << lane) - 1ull))] = (int)i;
}

// hsml of the gas particles of the tree in tree order (negative: does not contribute), for force_tree hmax
__global__ void __launch_bounds__(256) k_hsml_treeorder(int64_t npart, const int *__restrict__ order, const SphView A, double *__restrict__ out)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= npart)
        return;
    const int64_t ci = order[k];
    const int ty = A.type ? (A.type[ci] & 7) : 0;
    out[k] = (ty == 0 || ty == 5) ? A.hsml[ci] : -1.0;
}

// set_init_hsml, density.c:691-749: climb from the particle's leaf until the node holds 10 DesNumNgb particle masses.
// The device tree stores no father links; the ancestors of tree slot k are the nodes whose particle range contains k,
// found by descending from the root (first child = node+1, then the sibling chain).
__global__ void __launch_bounds__(256) k_set_init_hsml(const TreeView tv, const SphView A, double DesNumNgb, double MeanGasSeparation)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= tv.npart)
        return;
    const int64_t ci = tv.order[k];
    const int ty = A.type ? (A.type[ci] & 7) : 0;
    if(ty != 0 && ty != 5)
        return;
    const double m = (double)A.mass[ci];
    // deepest ancestor (the leaf's father chain upward stops at the first node with mass >= 10 DesNumNgb m; the leaf
    // itself is never the answer because the climb starts at the particle: first candidate is its leaf)
    int no = 0, best = -1;
    while(true) {
        const NodeLink lk = tv.link[no];
        const Src4 mom = tv.src[tv.npart + no];
        if(!(10 * DesNumNgb * m > mom.m))
            best = no; // this node satisfies the stop condition; a deeper one that does is preferred
        if(lk.pcount > 0)
            break;
        int c = no + 1; // find the child containing slot k
        while(true) {
            const NodeLink cl = tv.link[c];
            const int cend = (cl.sibling >= 0 && cl.sibling != lk.sibling) ? tv.link[cl.sibling].pstart : -1;
            if(cl.sibling == lk.sibling || k < cend)
                break;
            c = cl.sibling;
        }
        no = c;
    }
    double h = MeanGasSeparation;
    const int use = best >= 0 ? best : 0; // the climb ends at the root at the latest
    {
        const NodeGeo g = tv.geo[use];
        const Src4 mom = tv.src[tv.npart + use];
        const double testhsml = g.len * pow(3.0 / (4 * M_PI) * DesNumNgb * m / mom.m, 1.0 / 3);
        if(testhsml < 500. * MeanGasSeparation)
            h = testhsml;
    }
    A.hsml[ci] = h;
}

// ---------------------------------------------------------------- hydro
__device__ __forceinline__ double density_pred(double Density, double DivVel, double dtdrift) // SPH_DensityPred, hydra.c:300-312
{
    const double p = Density - DivVel * Density * dtdrift;
    return (p >= 1e-6 * Density) ? p : 1e-6 * Density;
}
__device__ __forceinline__ double pressure_pred(double eom, double entvar) // PressurePred, hydra.c:62-76
{
    if(entvar * eom <= 0)
        return 0;
    return exp(SPH_GAMMA * log(entvar * eom));
}

// per-source record of the hydro loop (tree order); every field is a function of the source particle alone
__global__ void __launch_bounds__(256) k_hydro_prepare(int64_t npart, const int *__restrict__ order, const SphView A, const mpg_sph_times T,
                                                       const mpg_hydro_params HP, const double *__restrict__ entvarpred, double fac_mu,
                                                       HydroSrc *__restrict__ hs, double *__restrict__ hsml_t)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= npart)
        return;
    const int64_t ci = order[k];
    const int bin = A.tb_hydro ? A.tb_hydro[ci] : 0;
    HydroSrc o;
    double v[3];
    vel_pred(A, T, ci, v);
    o.vx = v[0];
    o.vy = v[1];
    o.vz = v[2];
    o.hsml = A.hsml[ci];
    o.entvarpred = entvarpred[ci];
    o.density = density_pred(A.density[ci], A.divvel[ci], T.drifts[bin]);
    o.eomdensity = density_pred(HP.DensityIndependentSphOn ? A.egywtdensity[ci] : A.density[ci], A.divvel[ci], T.drifts[bin]);
    o.pressure = (o.entvarpred == 0) ? 0.0 : pressure_pred(o.eomdensity, o.entvarpred);
    o.soundspeed = sqrt(SPH_GAMMA * o.pressure / o.eomdensity);
    o.f2 = fabs(A.divvel[ci]) / (fabs(A.divvel[ci]) + A.curlvel[ci] + 0.0001 * o.soundspeed / fac_mu / o.hsml); // hydra.c:447-448
    o.dhsml = A.dhsmlegyfac[ci];
    o.dloga = T.dloga_bin[bin];
    hs[k] = o;
    hsml_t[k] = o.hsml; // (the distance tests read the smoothing lengths alone: 8 bytes per candidate instead of a line of the 96-byte records)
}

// the target-side constants of hydro_ngbiter
struct HydroTarget {
    double px, py, pz, IMass, IDensity, IEgyRho, IF1, soundspeed_i, p_over_rho2_i;
    HydroSrc me;
};
struct HydroAcc {
    double Acc0 = 0, Acc1 = 0, Acc2 = 0, DtEntropy = 0, MaxSignalVel = 0;
};

// symmetric distance test (treewalk.c:1218-1232) and the pair condition of hydro_ngbiter (hydra.c:330-338): is this a pair?
template <bool WRAP>
__device__ __forceinline__ bool hydro_test(const Src4 s, const double hsml_j, const HydroTarget &t, const DKernel &kernel_i, const HydroCtl &C,
                                           const double box)
{
    const double hh = fmax(hsml_j, t.me.hsml);
    const double d0 = near_img<WRAP>(t.px - s.x, box, 1.0 / box);
    const double d1 = near_img<WRAP>(t.py - s.y, box, 1.0 / box);
    const double d2 = near_img<WRAP>(t.pz - s.z, box, 1.0 / box);
    const double rsq = d0 * d0 + d1 * d1 + d2 * d2;
    if(rsq > hh * hh)
        return false;
    const double HHj = hsml_j * hsml_j; // kernel_init(hsml_j).HH
    return !(rsq <= 0 || !(rsq < kernel_i.HH || rsq < HHj));
}

// hydro_ngbiter for one pair, hydra.c:296-512
template <bool WRAP>
__device__ __forceinline__ void hydro_eval(const Src4 s, const HydroSrc &o, const HydroTarget &t, const DKernel &kernel_i, const HydroCtl &C,
                                           const mpg_hydro_params &HP, const double box, HydroAcc &a)
{
    const HydroSrc &me = t.me;
    const double d0 = near_img<WRAP>(t.px - s.x, box, 1.0 / box);
    const double d1 = near_img<WRAP>(t.py - s.y, box, 1.0 / box);
    const double d2 = near_img<WRAP>(t.pz - s.z, box, 1.0 / box);
    const double rsq = d0 * d0 + d1 * d1 + d2 * d2;
#ifdef SPH_FAST_DIV
    // (rsq > 0: hydro_test; the quotients of this function by reciprocals - see rcp_fast)
    const double rinv = rsqrt_fast(rsq);
    const double r = rsq * rinv;
    DKernel kernel_j;
    kernel_j.H = o.hsml;
    kernel_j.HH = o.hsml * o.hsml;
    kernel_j.Hinv = rcp_fast(o.hsml);
    kernel_j.support = ksupport(C.ktype);
    {
        const double hinv = kernel_j.Hinv * kernel_j.support;
        kernel_j.Wknorm = ksigma3(C.ktype) * p3(hinv);
        kernel_j.dWknorm = kernel_j.Wknorm * hinv;
    }
    const double inv_eom_j = rcp_fast(o.eomdensity);
    const double p_over_rho2_j = o.pressure * inv_eom_j * inv_eom_j;
#define SPH_DIV_R(x) ((x) * rinv)
#define SPH_DIV(x, y) ((x) * rcp_fast(y))
#else
    const DKernel kernel_j = kernel_init(o.hsml, C.ktype);
    const double r = sqrt(rsq);
    const double p_over_rho2_j = o.pressure / (o.eomdensity * o.eomdensity);
#define SPH_DIV_R(x) ((x) / r)
#define SPH_DIV(x, y) ((x) / (y))
#endif
    const double soundspeed_j = o.soundspeed;
    const double vsig = t.soundspeed_i + soundspeed_j;
    if(vsig > a.MaxSignalVel)
        a.MaxSignalVel = vsig;
    const double dv0 = me.vx - o.vx, dv1 = me.vy - o.vy, dv2 = me.vz - o.vz;
    const double vdotr = d0 * dv0 + d1 * dv1 + d2 * dv2;
    const double vdotr2 = vdotr + C.hubble_a2 * rsq;
    const double dwk_i = kernel_dwk(kernel_i, C.ktype, r * kernel_i.Hinv);
    const double dwk_j = kernel_dwk(kernel_j, C.ktype, r * kernel_j.Hinv);
    double visc = 0;
    if(vdotr2 < 0) { // Gadget-2 eqs. 13-14, hydra.c:435-462
        const double mu_ij = SPH_DIV_R(C.fac_mu * vdotr2);
        const double rho_ij = 0.5 * (t.IDensity + o.density);
        double vs = t.soundspeed_i + soundspeed_j;
        vs -= 3 * mu_ij;
        if(vs > a.MaxSignalVel)
            a.MaxSignalVel = vs;
        visc = SPH_DIV(0.25 * HP.ArtBulkViscConst * vs * (-mu_ij), rho_ij) * (t.IF1 + o.f2);
        const double dloga = 2 * fmax(me.dloga, o.dloga);
        if(dloga > 0 && (dwk_i + dwk_j) < 0) {
            if((t.IMass + s.m) > 0)
                visc = fmin(visc, SPH_DIV(0.5 * C.fac_vsic_fix * vdotr2, 0.5 * (t.IMass + s.m) * (dwk_i + dwk_j) * r * dloga));
        }
    }
    const double hfc_visc = SPH_DIV_R(0.5 * s.m * visc * (dwk_i + dwk_j));
    double hfc = hfc_visc;
    double rr1 = 1, rr2 = 1;
    if(HP.DensityIndependentSphOn) {
        rr1 = 0, rr2 = 0;
        hfc += SPH_DIV_R(s.m * (SPH_DIV(dwk_i * t.p_over_rho2_i * o.entvarpred, me.entvarpred) + SPH_DIV(dwk_j * p_over_rho2_j * me.entvarpred, o.entvarpred)));
        if(HP.DensityContrastLimit >= 0) {
            rr1 = SPH_DIV(t.IEgyRho, t.IDensity);
            rr2 = SPH_DIV(o.eomdensity, o.density);
            if(HP.DensityContrastLimit > 0) {
                rr1 = fmin(rr1, HP.DensityContrastLimit);
                rr2 = fmin(rr2, HP.DensityContrastLimit);
            }
        }
    }
    hfc += SPH_DIV_R(s.m * (t.p_over_rho2_i * me.dhsml * dwk_i * rr1 + p_over_rho2_j * o.dhsml * dwk_j * rr2));
    a.Acc0 += -hfc * d0;
    a.Acc1 += -hfc * d1;
    a.Acc2 += -hfc * d2;
    a.DtEntropy += 0.5 * hfc_visc * vdotr2;
}

// The end of the hydro loop for one target: hydro_reduce (assign) + hydro_postprocess (hydra.c:279-294, 514-528) and the statistics.
__device__ __forceinline__ void hydro_finish(const SphView &A, const HydroCtl &C, HydroAcc &a, const HydroTarget &t, const bool valid, const int s,
                                             const int lane, const int i, unsigned long long *__restrict__ stats, const unsigned n_cand,
                                             const unsigned n_pair)
{
    a.Acc0 = group_sum(a.Acc0);
    a.Acc1 = group_sum(a.Acc1);
    a.Acc2 = group_sum(a.Acc2);
    a.DtEntropy = group_sum(a.DtEntropy);
    for(int off = 1; off < 8; off <<= 1)
        a.MaxSignalVel = fmax(a.MaxSignalVel, __shfl_xor(a.MaxSignalVel, off));
    if(valid && s == 0) {
        // hydro_reduce (assign) + hydro_postprocess, hydra.c:279-294, 514-528
        A.hydroacc_out[3 * (int64_t)i] = a.Acc0;
        A.hydroacc_out[3 * (int64_t)i + 1] = a.Acc1;
        A.hydroacc_out[3 * (int64_t)i + 2] = a.Acc2;
        A.maxsignalvel[i] = a.MaxSignalVel;
        A.dtentropy_out[i] = a.DtEntropy * (SPH_GAMMA_MINUS1 / (C.hubble_a2 * pow(t.IDensity, SPH_GAMMA_MINUS1)));
    }
    unsigned long long c_cand = n_cand, c_pair = n_pair;
    for(int off = 32; off > 0; off >>= 1) {
        c_cand += __shfl_down(c_cand, off);
        c_pair += __shfl_down(c_pair, off);
    }
    if(lane == 0 && stats) {
        atomicAdd(&stats[0], c_cand);
        atomicAdd(&stats[1], c_pair);
    }
}

// hydro_force loop: group-cooperative walk with the symmetric cull (see k_density)
__global__ void __launch_bounds__(256, 4) k_hydro(const TreeView tv, const SphView A, const mpg_sph_times T, const mpg_hydro_params HP,
                                               const HydroCtl C, const HydroSrc *__restrict__ hs, const double *__restrict__ hsml_t,
                                               const int *__restrict__ slot_of,
                                               const int *__restrict__ targets, int64_t ntargets, unsigned long long *__restrict__ stats,
                                               unsigned *__restrict__ err)
{
    __shared__ unsigned s_stack[4 * 8 * SPH_STK];
    __shared__ int s_cbuf[4 * 8 * SPH_CBUF];
    __shared__ unsigned s_llist[4 * 8 * SPH_LCAP];
#ifdef MPG_EXP_LDSPAD_SPH // timing experiment: fewer resident blocks per CU with the same code (bytes of unused LDS)
    __shared__ unsigned s_pad[MPG_EXP_LDSPAD_SPH / 4];
    if(tv.box < 0)
        s_pad[threadIdx.x] = 1u, s_llist[0] = s_pad[(threadIdx.x + 1) & 255];
#endif
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 3, s = lane & 7, gshift = grp * 8;
    unsigned *stack = s_stack + ((threadIdx.x >> 6) * 8 + grp) * SPH_STK;
    int *cbuf = s_cbuf + ((threadIdx.x >> 6) * 8 + grp) * SPH_CBUF;
    unsigned *llist = s_llist + ((threadIdx.x >> 6) * 8 + grp) * SPH_LCAP;
    int cnt = 0, head = 0; // survivors waiting in the group's ring buffer, its first slot (group-uniform)
    const int64_t q = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + grp;
    const bool valid = q < ntargets; // the queue holds gas particles only (hydro_haswork)
    unsigned n_cand = 0, n_pair = 0;
    int i = 0;
    HydroTarget t{};
    t.me.hsml = 1.0;
    if(valid) {
        i = targets[q];
        t.me = hs[slot_of[i]];
        t.px = A.pos[3 * (int64_t)i];
        t.py = A.pos[3 * (int64_t)i + 1];
        t.pz = A.pos[3 * (int64_t)i + 2];
        // hydro_copy, hydra.c:247-277
        t.IMass = (double)A.mass[i];
        t.IDensity = A.density[i];
        t.IEgyRho = A.egywtdensity ? A.egywtdensity[i] : 0.0;
        const double eomdensity_i = HP.DensityIndependentSphOn ? t.IEgyRho : t.IDensity;
        // the target's own pressure: PressurePred[PI] = predicted from the drifted EOM density (hydra.c:206-212)
        const double IPressure = t.me.pressure;
        const double soundspeed_c = sqrt(SPH_GAMMA * IPressure / eomdensity_i);
        t.IF1 = fabs(A.divvel[i]) / (fabs(A.divvel[i]) + A.curlvel[i] + 0.0001 * soundspeed_c / t.me.hsml / C.fac_mu);
        if(HP.DensityIndependentSphOn) {
            t.soundspeed_i = sqrt(SPH_GAMMA * IPressure / t.IEgyRho);
            t.p_over_rho2_i = IPressure / (t.IEgyRho * t.IEgyRho);
        }
        else {
            t.soundspeed_i = sqrt(SPH_GAMMA * IPressure / t.IDensity);
            t.p_over_rho2_i = IPressure / (t.IDensity * t.IDensity);
        }
    }
    // the symmetric search on the cubes around the nodes' particles with the largest Hsml below each node as its radius, when the tree carries
    // both (calc_search_boxes, calc_search_hsmax); else on the cells with the reference's hmax
    const bool tight = tv.geoS && tv.hsmaxS;
    const NodeGeo *__restrict__ sgeo = tight ? tv.geoS : tv.geoB;
    const double *__restrict__ shm = tight ? tv.hsmaxS : tv.hmaxB;
    const DKernel kernel_i = kernel_init(t.me.hsml, C.ktype);
    HydroAcc a;
    a.MaxSignalVel = t.soundspeed_i;
    int sp = 0;
    if(valid) {
        if(s == 0)
            stack[0] = (0u << 4) | 1u; // the root
        sp = 1;
    }
    bool overflow = false;
    auto loops = [&](auto wrap_tag) { // (see k_density)
    constexpr bool WRAP = decltype(wrap_tag)::value;
    for(;;) {
        // ---- phase A: walk; opened leaves go to the group's list
        int nl = 0;
        for(;;) {
            const bool go = sp > 0 && nl + 8 * SPH_WALK_K * SPH_NE <= SPH_LCAP;
            if(ballot64(go) == 0)
                break;
            nl = walk_stepk<true, SPH_WALK_K, SPH_MERGE, WRAP, SPH_NE>(tv, sgeo, shm, stack, sp, go, s, gshift, t.me.hsml, t.px, t.py, t.pz, llist, nl, overflow, tv.linkS);
#ifdef SPH_HIST
            if(lane == 0)
                atomicAdd(&stats[7], 1ull);
#endif
            if(ballot64(overflow) != 0)
                break;
        }
        if(ballot64(overflow) != 0)
            break;
        // ---- phase B: every group takes its next leaf; lane s <-> particle s
        // (the candidate of the next leaf is requested before this one is tested: see k_density)
        unsigned e = (0 < nl) ? llist[0] : 0u;
        int ps = (int)(e >> 4), pc = (int)(e & 15u);
        Src4 cand = tv.src[ps + (s < pc ? s : 0)];
        double cand_h = hsml_t[ps + (s < pc ? s : 0)];
        for(int it = 0;; it++) {
            const bool has = it < nl;
            if(ballot64(has) == 0)
                break;
            const unsigned e_n = (it + 1 < nl) ? llist[it + 1] : 0u;
            const int ps_n = (int)(e_n >> 4), pc_n = (int)(e_n & 15u);
            const Src4 cand_n = tv.src[ps_n + (s < pc_n ? s : 0)];
            const double cand_hn = hsml_t[ps_n + (s < pc_n ? s : 0)];
            bool keep = false;
#ifdef SPH_HIST // experiment: list entries by particle count (1-2, 3-4, 5-6, 7-8), phase-B iterations and walk steps per wave
            if(s == 0 && has)
                atomicAdd(&stats[2 + (pc - 1) / 2], 1ull);
            if(lane == 0)
                atomicAdd(&stats[6], 1ull);
#endif
            if(s < pc) {
                n_cand++;
                keep = hydro_test<WRAP>(cand, cand_h, t, kernel_i, C, tv.box);
                n_pair += keep ? 1u : 0u;
            }
            cnt = cbuf_push(cbuf, head, cnt, keep, ps + s, s, gshift);
            if(ballot64(cnt >= SPH_TRIG) != 0) {
                if(cnt >= 8) {
                    const int sidx = cbuf[(head + s) & (SPH_CBUF - 1)];
                    hydro_eval<WRAP>(tv.src[sidx], hs[sidx], t, kernel_i, C, HP, tv.box, a);
                    head = (head + 8) & (SPH_CBUF - 1);
                    cnt -= 8;
                }
            }
            cand = cand_n;
            cand_h = cand_hn;
            ps = ps_n;
            pc = pc_n;
        }
        if(ballot64(sp > 0) == 0)
            break;
    }
    if(ballot64(overflow) != 0)
        return;
    while(ballot64(cnt > 0) != 0) { // drain the survivor buffers
        if(s < cnt) {
            const int sidx = cbuf[(head + s) & (SPH_CBUF - 1)];
            hydro_eval<WRAP>(tv.src[sidx], hs[sidx], t, kernel_i, C, HP, tv.box, a);
        }
        head = (head + 8) & (SPH_CBUF - 1);
        cnt = cnt > 8 ? cnt - 8 : 0;
    }
    };
    // symmetric search: the radius of a cull is max(node hmax, Hsml) <= max(root hmax, Hsml)
    if(interior_wave(valid, t.px, t.py, t.pz, fmax(t.me.hsml, shm[0]), tv.box))
        loops(std::false_type{});
    else
        loops(std::true_type{});
    if(ballot64(overflow) != 0) {
        if(lane == 0)
            atomicExch(err, 1u);
        return;
    }
    hydro_finish(A, C, a, t, valid, s, lane, i, stats, n_cand, n_pair);
}

__global__ void __launch_bounds__(256) k_slot_of(int64_t npart, const int *__restrict__ order, int *__restrict__ slot_of)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k < npart)
        slot_of[order[k]] = (int)k;
}

static inline unsigned nblk(int64_t n, int b = 256) { return (unsigned)((n + b - 1) / b); }

static int kernel_index(int enumtype)
{
    // enum DensityKernelType (densitykernel.h:17-21): 1 cubic, 2 quintic, 4 quartic -> KERNELS[] index (densitykernel.c:155-171)
    MPG_CHECK(enumtype == 1 || enumtype == 2 || enumtype == 4, "Density Kernel type is unknown");
    return enumtype == 1 ? 0 : (enumtype == 2 ? 1 : 2);
}

// The SPH searches stop at nodes of <= this many particles and list their whole range (TreeBuilder::calc_search_links): 8 = the
// reference's leaves only (also under MPG_SPH_CELL_CULL=1, where the candidates are the reference's one for one).
static int search_leaf_cap()
{
    static const int cap = [] {
        if(getenv("MPG_SPH_CELL_CULL"))
            return 8;
        const char *e = getenv("MPG_SPH_LEAF_CAP");
        const int c = e ? atoi(e) : 8 * SPH_NE;
        return c < 8 ? 8 : (c > 8 * SPH_NE ? 8 * SPH_NE : c);
    }();
    return cap;
}

double sph_desnumngb(const mpg_density_params &P)
{
    const int t = kernel_index(P.DensityKernelType);
    const double support = t == 0 ? 2. : (t == 1 ? 3. : 2.5);
    return NORM_COEFF * pow(support * P.DensityResolutionEta, NUMDIMS); // density_kernel_desnumngb, densitykernel.c:124-131
}

void SphEngine::density(TreeBuilder &tree, const SphView &A, const mpg_sph_times &T, const mpg_density_params &P, double force_softening,
                        const int *d_active, int64_t nactive, int64_t n, int update_hsml, int DoEgyDensity, int BlackHoleOn, bool bh_in_tree,
                        hipStream_t st)
{
    tree.ensure_level_order(st); // the cooperative walk uses the level-ordered copy of the tree
    // The asymmetric search of the density loop keeps the reference's CELL test (cull_node): on the cubes around the nodes' particles it
    // tests 29 % fewer candidates (2 x 128^3: 686 M -> 487 M for 231 M neighbours) but runs 3 % longer - a candidate is a lane of a test
    // iteration that runs anyway, and leaves dropped from a set of siblings break the join of the rest (profiles/r05a_experiments).  The
    // hydro loop's symmetric search gains from the cubes (hydro_force below).  MPG_SPH_DENSITY_CUBES=1 switches them on here too.
    static const bool density_cubes = getenv("MPG_SPH_DENSITY_CUBES") != nullptr && getenv("MPG_SPH_CELL_CULL") == nullptr;
    if(density_cubes && !tree.has_boxes)
        tree.calc_search_boxes(st);
    const int leaf_cap = search_leaf_cap();
    if(leaf_cap > 8 && !(tree.has_slinks && tree.slink_cap == leaf_cap))
        tree.calc_search_links(leaf_cap, st);
    TreeView tv = tree.view();
    if(!density_cubes)
        tv.geoS = nullptr;
    if(leaf_cap <= 8)
        tv.linkS = nullptr;
    MPG_CHECK(tv.npart > 0 || n == 0, "density: the tree holds no gas particles");
    const int64_t nact = d_active ? nactive : n;
    left.reserve(n + 1);
    right.reserve(n + 1);
    numngb.reserve(n + 1);
    entvarpred.reserve(n + 1);
    // (targets: the gas of the tree, and the black holes, which a gas-only tree does not hold; BlackHoleOn only changes their
    // neighbour number and caps their radius, density.c:598-600, 667-670)
    const int64_t qcap = !bh_in_tree ? tv.npart + n : tv.npart;
    queue_a.reserve(qcap + 1);
    queue_b.reserve(qcap + 1);
    aux.reserve(tv.npart + 1);
    ctr.reserve(8);
    stats.reserve(8);
    DensityCtl C;
    C.ktype = kernel_index(P.DensityKernelType);
    C.DesNumNgb = sph_desnumngb(P);
    C.MinGasHsml = P.MinGasHsmlFractional * (force_softening / 2.8); // density.c:268
    C.update_hsml = update_hsml;
    C.DoEgyDensity = DoEgyDensity;
    C.BlackHoleOn = BlackHoleOn;
    C.Left = left.p;
    C.Right = right.p;
    C.NumNgb = numngb.p;
    C.entvarpred = entvarpred.p;
    MPG_HIP(hipMemsetAsync(ctr.p, 0, 8 * sizeof(unsigned), st));
    MPG_HIP(hipMemsetAsync(stats.p, 0, 8 * sizeof(unsigned long long), st));
    if(tv.npart > 0)
        hipLaunchKernelGGL(k_sph_predict, dim3(nblk(tv.npart)), dim3(256), 0, st, tv.npart, tv.order, A, T, aux.p, entvarpred.p);
    const uint8_t *flags = mark_active(d_active, nact, n, st);
    if(tv.npart > 0 && nact > 0)
        hipLaunchKernelGGL(k_queue_treeorder<true>, dim3(nblk(tv.npart)), dim3(256), 0, st, tv.npart, tv.order, flags, A, C, tv.box, false,
                           queue_a.p, ctr.p);
    if(!bh_in_tree && A.type && n > 0 && nact > 0)
        hipLaunchKernelGGL(k_queue_blackholes, dim3(nblk(n)), dim3(256), 0, st, n, flags, A, C, tv.box, queue_a.p, ctr.p);
    unsigned nq = 0;
    MPG_HIP(hipMemcpyAsync(&nq, ctr.p, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    MPG_HIP(hipStreamSynchronize(st));
    last_iterations = 0;
    last_targets = 0;
    int *qa = queue_a.p, *qb = queue_b.p;
    while(nq > 0) {
        last_iterations++;
        last_targets += nq;
        MPG_HIP(hipMemsetAsync(ctr.p + 1, 0, sizeof(unsigned), st));
        hipLaunchKernelGGL(k_density, dim3(nblk(nq, 32)), dim3(256), 0, st, tv, A, T, P, C, aux.p, qa, (int64_t)nq, qb, ctr.p + 1, stats.p,
                           ctr.p + 7);
        MPG_HIP(hipGetLastError());
        if(!update_hsml)
            break;
        unsigned nr[7] = {0, 0, 0, 0, 0, 0, 0};
        MPG_HIP(hipMemcpyAsync(nr, ctr.p + 1, sizeof(nr), hipMemcpyDeviceToHost, st));
        MPG_HIP(hipStreamSynchronize(st));
        MPG_CHECK(nr[6] == 0, "density: neighbour-search stack overflow (tree deeper than the walk supports)");
        nq = nr[0];
        int *t = qa;
        qa = qb;
        qb = t;
        if(nq > 0 && last_iterations > 400) // MAXITER, treewalk.c:1362-1364
            fail(__FILE__, __LINE__, "failed to converge density for " + std::to_string(nq) + " particles");
    }
    if(update_hsml && tv.npart > 0) {
        // update_tree_hmax_father for every finished particle (density.c:551-553) == leaf hmax from the final Hsml
        // (gathered in calc_hmax, from the array as it is then: with distributed particles the smoothing lengths of the ghost
        // particles are refreshed from their owners between density() and the hmax pass)
        hsml_view = A;
        hmax_pending = true;
    }
    unsigned long long hs[2] = {0, 0};
    unsigned e = 0;
    MPG_HIP(hipMemcpyAsync(hs, stats.p, sizeof(hs), hipMemcpyDeviceToHost, st));
    MPG_HIP(hipMemcpyAsync(&e, ctr.p + 7, sizeof(e), hipMemcpyDeviceToHost, st));
    MPG_HIP(hipStreamSynchronize(st));
    MPG_CHECK(e == 0, "density: neighbour-search stack overflow (tree deeper than the walk supports)");
    last_interactions = (int64_t)hs[0];
    last_candidates = (int64_t)hs[1];
}

const uint8_t *SphEngine::mark_active(const int *d_active, int64_t nactive, int64_t n, hipStream_t st)
{
    if(!d_active)
        return nullptr;
    active_flags.reserve(n + 1);
    MPG_HIP(hipMemsetAsync(active_flags.p, 0, (size_t)n, st));
    if(nactive > 0)
        hipLaunchKernelGGL(k_mark_active, dim3(nblk(nactive)), dim3(256), 0, st, nactive, d_active, active_flags.p);
    return active_flags.p;
}

void SphEngine::set_init_hsml(TreeBuilder &tree, const SphView &A, const mpg_density_params &P, double MeanGasSeparation, hipStream_t st)
{
    const TreeView tv = tree.view();
    MPG_CHECK(tree.has_moments, "set_init_hsml needs tree moments (force_tree_calc_moments, density.c:695)");
    if(tv.npart > 0)
        hipLaunchKernelGGL(k_set_init_hsml, dim3(nblk(tv.npart)), dim3(256), 0, st, tv, A, sph_desnumngb(P), MeanGasSeparation);
    MPG_HIP(hipGetLastError());
}

void SphEngine::calc_hmax(TreeBuilder &tree, hipStream_t st)
{
    MPG_CHECK(hmax_pending, "force_tree_calc_moments for hmax called before density()");
    const TreeView tv = tree.view();
    hsml_tree.reserve(tv.npart + 1);
    if(tv.npart > 0)
        hipLaunchKernelGGL(k_hsml_treeorder, dim3(nblk(tv.npart)), dim3(256), 0, st, tv.npart, tv.order, hsml_view, hsml_tree.p);
    tree.calc_hmax(hsml_tree.p, st);
    hmax_pending = false;
}

void SphEngine::hydro_force(TreeBuilder &tree, const SphView &A, const mpg_sph_times &T, const mpg_density_params &P, const mpg_hydro_params &HP,
                            const int *d_active, int64_t nactive, int64_t n, hipStream_t st)
{
    tree.ensure_level_order(st);
    TreeView tv = tree.view();
    MPG_CHECK(tree.has_hmax && tv.hmax && tv.hmaxB, "Hydro called before hmax computed"); // hydra.c:172-173
    MPG_CHECK(entvarpred.p != nullptr, "hydro_force needs the predicted entropies of density()");
    hsrc.reserve(tv.npart + 1);
    hsml_t.reserve(tv.npart + 1);
    slot_of.reserve(n + 1);
    stats.reserve(8);
    HydroCtl C;
    C.ktype = kernel_index(P.DensityKernelType);
    const double atime = T.atime, hubble = T.hubble;
    C.fac_mu = pow(atime, 3 * (SPH_GAMMA - 1) / 2) / atime; // hydra.c:219-223
    C.fac_vsic_fix = hubble * pow(atime, 3 * SPH_GAMMA_MINUS1);
    C.hubble_a2 = hubble * atime * atime;
    MPG_HIP(hipMemsetAsync(stats.p, 0, 8 * sizeof(unsigned long long), st));
    if(tv.npart > 0) {
        hipLaunchKernelGGL(k_slot_of, dim3(nblk(tv.npart)), dim3(256), 0, st, tv.npart, tv.order, slot_of.p);
        hipLaunchKernelGGL(k_hydro_prepare, dim3(nblk(tv.npart)), dim3(256), 0, st, tv.npart, tv.order, A, T, HP, entvarpred.p, C.fac_mu, hsrc.p, hsml_t.p);
        // the symmetric search runs on the cubes around the nodes' particles; its radius per node is the largest of the smoothing lengths
        // the pair tests below read (hsml_t), where the reference has the reach beyond the cell's faces (hmax)
        static const bool cell_cull = getenv("MPG_SPH_CELL_CULL") != nullptr;
        if(!cell_cull) {
            if(!tree.has_boxes)
                tree.calc_search_boxes(st);
            tree.calc_search_hsmax(hsml_t.p, st);
            const int leaf_cap = search_leaf_cap();
            if(leaf_cap > 8 && !(tree.has_slinks && tree.slink_cap == leaf_cap))
                tree.calc_search_links(leaf_cap, st);
            tv = tree.view();
            if(leaf_cap <= 8)
                tv.linkS = nullptr;
        }
        else
            tv.linkS = nullptr;
    }
    // work queue: the active gas particles in tree order
    queue_a.reserve(tv.npart + 1);
    ctr.reserve(8);
    MPG_HIP(hipMemsetAsync(ctr.p, 0, 8 * sizeof(unsigned), st));
    const uint8_t *flags = mark_active(d_active, nactive, n, st);
    unsigned nt = 0;
    if(tv.npart > 0 && (!d_active || nactive > 0)) {
        hipLaunchKernelGGL(k_queue_treeorder<false>, dim3(nblk(tv.npart)), dim3(256), 0, st, tv.npart, tv.order, flags, A, DensityCtl{}, tv.box, true,
                           queue_a.p, ctr.p);
        MPG_HIP(hipMemcpyAsync(&nt, ctr.p, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        MPG_HIP(hipStreamSynchronize(st));
    }
    if(nt > 0)
        hipLaunchKernelGGL(k_hydro, dim3(nblk(nt, 32)), dim3(256), 0, st, tv, A, T, HP, C, hsrc.p, hsml_t.p, slot_of.p, queue_a.p, (int64_t)nt, stats.p,
                           ctr.p + 7);
    MPG_HIP(hipGetLastError());
    unsigned long long hs[2] = {0, 0};
    unsigned e = 0;
    MPG_HIP(hipMemcpyAsync(hs, stats.p, sizeof(hs), hipMemcpyDeviceToHost, st));
    MPG_HIP(hipMemcpyAsync(&e, ctr.p + 7, sizeof(e), hipMemcpyDeviceToHost, st));
    MPG_HIP(hipStreamSynchronize(st));
#ifdef SPH_HIST
    {
        unsigned long long h8[8];
        MPG_HIP(hipMemcpy(h8, stats.p, sizeof(h8), hipMemcpyDeviceToHost));
        fprintf(stderr, "SPH_HIST hydro: targets %lld cand %llu pairs %llu entries[1-2,3-4,5-6,7-8] %llu %llu %llu %llu waveiters %llu wavesteps %llu\n", (long long)nt,
                h8[0], h8[1], h8[2], h8[3], h8[4], h8[5], h8[6], h8[7]);
    }
#endif
    MPG_CHECK(e == 0, "hydro_force: neighbour-search stack overflow (tree deeper than the walk supports)");
    last_candidates = (int64_t)hs[0];
    last_interactions = (int64_t)hs[1];
}

} // namespace mpg
