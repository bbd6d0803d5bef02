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
#include <cmath>

namespace mpg {

#define NUMDIMS 3
#define NORM_COEFF 4.188790204786 // densitykernel.h:6
#define SPH_GAMMA (5.0 / 3.0)     // physconst.h:35
#define SPH_GAMMA_MINUS1 (SPH_GAMMA - 1)
#define FACT1 0.366025403785      // treewalk.c:19

__device__ __forceinline__ double nearest_img(double x, double box, double invbox) { return x - box * rint(x * invbox); }

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

template <int TYPE> __device__ __forceinline__ double wk_q(double q) // densitykernel.c:24-90
{
    if(TYPE == 0) {
        if(q < 1.0)
            return 0.25 * p3(2 - q) - p3(1 - q);
        if(q < 2.0)
            return 0.25 * p3(2 - q);
        return 0.0;
    }
    else if(TYPE == 1) {
        if(q < 1.0)
            return p5(3 - q) - 6 * p5(2 - q) + 15 * p5(1 - q);
        if(q < 2.0)
            return p5(3 - q) - 6 * p5(2 - q);
        if(q < 3.0)
            return p5(3 - q);
        return 0.0;
    }
    else {
        if(q < 0.5)
            return p4(2.5 - q) - 5 * p4(1.5 - q) + 10 * p4(0.5 - q);
        if(q < 1.5)
            return p4(2.5 - q) - 5 * p4(1.5 - q);
        if(q < 2.5)
            return p4(2.5 - q);
        return 0.0;
    }
}
template <int TYPE> __device__ __forceinline__ double dwk_q(double q)
{
    if(TYPE == 0) {
        if(q < 1.0)
            return -0.25 * 3 * p2(2 - q) + 3 * p2(1 - q);
        if(q < 2.0)
            return -0.25 * 3 * p2(2 - q);
        return 0.0;
    }
    else if(TYPE == 1) {
        if(q < 1.0)
            return -5 * p4(3 - q) + 30 * p4(2 - q) - 75 * p4(1 - q);
        if(q < 2.0)
            return -5 * p4(3 - q) + 30 * p4(2 - q);
        if(q < 3.0)
            return -5 * p4(3 - q);
        return 0.0;
    }
    else {
        if(q < 0.5)
            return -4 * p3(2.5 - q) + 20 * p3(1.5 - q) - 40 * p3(0.5 - q);
        if(q < 1.5)
            return -4 * p3(2.5 - q) + 20 * p3(1.5 - q);
        if(q < 2.5)
            return -4 * p3(2.5 - q);
        return 0.0;
    }
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

// One density pass over the current queue: treewalk_visit_nolist_ngbiter + density_ngbiter + density_reduce +
// density_postprocess + density_check_neighbours.  Targets that are not done are appended to `redo`.
__global__ void __launch_bounds__(256) k_density(const TreeView tv, const SphView A, const mpg_sph_times T, const mpg_density_params P,
                                                 const DensityCtl C, const Aux4 *__restrict__ aux, const int *__restrict__ queue,
                                                 int64_t nqueue, int *__restrict__ redo, unsigned *__restrict__ nredo,
                                                 unsigned long long *__restrict__ stats)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n_int = 0, n_cand = 0;
    if(q < nqueue) {
        const int i = queue[q];
        const int ty = A.type ? (A.type[i] & 7) : 0;
        const double px = A.pos[3 * (int64_t)i], py = A.pos[3 * (int64_t)i + 1], pz = A.pos[3 * (int64_t)i + 2];
        double ivel[3];
        if(ty != 0) { // density_copy, density.c:357-372
            ivel[0] = A.vel[3 * (int64_t)i];
            ivel[1] = A.vel[3 * (int64_t)i + 1];
            ivel[2] = A.vel[3 * (int64_t)i + 2];
        }
        else
            vel_pred(A, T, i, ivel);
        const double hsml = A.hsml[i];
        const DKernel kern = kernel_init(hsml, C.ktype);
        const double kvol = NORM_COEFF * p3(kern.H);
        const double h2 = hsml * hsml;
        double EgyRho = 0, DhsmlEgy = 0, Rho = 0, DhsmlDensity = 0, Ngb = 0, Div = 0, Rot0 = 0, Rot1 = 0, Rot2 = 0, G0 = 0, G1 = 0, G2 = 0;
        int no = 0;
        while(no >= 0) {
            const NodeGeo g = tv.geo[no];
            const NodeLink lk = tv.link[no];
            if(cull_node(g, 0.0, hsml, px, py, pz, tv.box, 1.0 / tv.box)) {
                no = lk.sibling;
                continue;
            }
            if(lk.pcount > 0) {
                for(int k = 0; k < lk.pcount; k++) {
                    const int sidx = lk.pstart + k;
                    const Src4 s = tv.src[sidx];
                    n_cand++;
                    // the distance vector points to 'other': I.Pos - P[other].Pos (treewalk.c:1218-1225)
                    const double d0 = nearest_img(px - s.x, tv.box, 1.0 / tv.box);
                    const double d1 = nearest_img(py - s.y, tv.box, 1.0 / tv.box);
                    const double d2 = nearest_img(pz - s.z, tv.box, 1.0 / tv.box);
                    const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
                    if(r2 > h2)
                        continue;
                    n_int++;
                    if(r2 < kern.HH) { // density_ngbiter, density.c:451-518
                        const double r = sqrt(r2);
                        const double u = r * kern.Hinv;
                        const double wk = kernel_wk(kern, C.ktype, u);
                        Ngb += wk * kvol;
                        const double dwk = kernel_dwk(kern, C.ktype, u);
                        const double mass_j = s.m;
                        Rho += mass_j * wk;
                        const double density_dW = -(NUMDIMS * kern.Hinv * wk + u * dwk);
                        DhsmlDensity += mass_j * density_dW;
                        const Aux4 a = aux[sidx];
                        if(C.DoEgyDensity) {
                            EgyRho += mass_j * a.w * wk;
                            DhsmlEgy += mass_j * a.w * density_dW;
                        }
                        if(r > 0) {
                            const double fac = mass_j * dwk / r;
                            const double dv0 = ivel[0] - a.x, dv1 = ivel[1] - a.y, dv2 = ivel[2] - a.z;
                            Div += -fac * (d0 * dv0 + d1 * dv1 + d2 * dv2);
                            Rot0 += fac * (dv1 * d2 - d1 * dv2); // crossproduct(dv, dist), densitykernel.h:63-76
                            Rot1 += fac * (dv2 * d0 - d2 * dv0);
                            Rot2 += fac * (dv0 * d1 - d0 * dv1);
                            G0 += fac * d0;
                            G1 += fac * d1;
                            G2 += fac * d2;
                        }
                    }
                }
                no = lk.sibling;
                continue;
            }
            no = no + 1;
        }
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
            if(!done)
                redo[atomicAdd(nredo, 1u)] = i;
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
    // statistics: successful distance tests (the reference's ninteractions) and candidates tested
    for(int off = 32; off > 0; off >>= 1) {
        n_int += __shfl_down(n_int, off);
        n_cand += __shfl_down(n_cand, off);
    }
    if((threadIdx.x & 63) == 0 && stats) {
        atomicAdd(&stats[0], n_int);
        atomicAdd(&stats[1], n_cand);
    }
}

__global__ void __launch_bounds__(256) k_density_init(int64_t nact, const int *__restrict__ active, const SphView A, const DensityCtl C,
                                                      double box, int *__restrict__ queue, unsigned *__restrict__ nqueue)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool inb = k < nact;
    const int i = inb ? (active ? active[k] : (int)k) : 0;
    if(inb) {
        C.Right[i] = box; // density.c:277-285
        C.NumNgb[i] = 0;
        C.Left[i] = 0;
    }
    const int ty = (inb && A.type) ? (A.type[i] & 7) : 0;
    const bool work = inb && (ty == 0 || ty == 5); // density_haswork, density.c:521-530
    // wave-aggregated append: one atomic per wave (same-address atomics serialise)
    const unsigned long long m = __ballot(work);
    unsigned basepos = 0;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    if(work && lane == leader)
        basepos = atomicAdd(nqueue, (unsigned)__popcll(m));
    basepos = __shfl(basepos, leader < 0 ? 0 : leader);
    if(work)
        queue[basepos + __popcll(m & ((1ull << lane) - 1ull))] = i;
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
                                                       HydroSrc *__restrict__ hs)
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
}

__global__ void __launch_bounds__(256) k_hydro(const TreeView tv, const SphView A, const mpg_sph_times T, const mpg_hydro_params HP,
                                               const HydroCtl C, const HydroSrc *__restrict__ hs, const int *__restrict__ slot_of,
                                               const int *__restrict__ targets, int64_t ntargets, unsigned long long *__restrict__ stats)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n_cand = 0, n_pair = 0;
    if(q < ntargets) {
        const int i = targets ? targets[q] : tv.order[q];
        const int ty = A.type ? (A.type[i] & 7) : 0;
        if(ty == 0) { // hydro_haswork
            const int myslot = slot_of[i];
            const HydroSrc me = hs[myslot];
            const double px = A.pos[3 * (int64_t)i], py = A.pos[3 * (int64_t)i + 1], pz = A.pos[3 * (int64_t)i + 2];
            // hydro_copy, hydra.c:247-277
            const double IMass = (double)A.mass[i];
            const double IDensity = A.density[i];
            const double IEgyRho = A.egywtdensity ? A.egywtdensity[i] : 0.0;
            const double eomdensity_i = HP.DensityIndependentSphOn ? IEgyRho : IDensity;
            // the target's own pressure: PressurePred[PI] = predicted from the drifted EOM density (hydra.c:206-212)
            const double IPressure = me.pressure;
            const double soundspeed_c = sqrt(SPH_GAMMA * IPressure / eomdensity_i);
            const double IF1 = fabs(A.divvel[i]) / (fabs(A.divvel[i]) + A.curlvel[i] + 0.0001 * soundspeed_c / me.hsml / C.fac_mu);
            double soundspeed_i, p_over_rho2_i;
            if(HP.DensityIndependentSphOn) {
                soundspeed_i = sqrt(SPH_GAMMA * IPressure / IEgyRho);
                p_over_rho2_i = IPressure / (IEgyRho * IEgyRho);
            }
            else {
                soundspeed_i = sqrt(SPH_GAMMA * IPressure / IDensity);
                p_over_rho2_i = IPressure / (IDensity * IDensity);
            }
            const DKernel kernel_i = kernel_init(me.hsml, C.ktype);
            double Acc0 = 0, Acc1 = 0, Acc2 = 0, DtEntropy = 0, MaxSignalVel = soundspeed_i;
            int no = 0;
            while(no >= 0) {
                const NodeGeo g = tv.geo[no];
                const NodeLink lk = tv.link[no];
                if(cull_node(g, tv.hmax[no], me.hsml, px, py, pz, tv.box, 1.0 / tv.box)) {
                    no = lk.sibling;
                    continue;
                }
                if(lk.pcount > 0) {
                    for(int k = 0; k < lk.pcount; k++) {
                        const int sidx = lk.pstart + k;
                        const Src4 s = tv.src[sidx];
                        const HydroSrc o = hs[sidx];
                        n_cand++;
                        const double hh = fmax(o.hsml, me.hsml);
                        const double d0 = nearest_img(px - s.x, tv.box, 1.0 / tv.box);
                        const double d1 = nearest_img(py - s.y, tv.box, 1.0 / tv.box);
                        const double d2 = nearest_img(pz - s.z, tv.box, 1.0 / tv.box);
                        const double rsq = d0 * d0 + d1 * d1 + d2 * d2;
                        if(rsq > hh * hh)
                            continue;
                        const DKernel kernel_j = kernel_init(o.hsml, C.ktype);
                        if(rsq <= 0 || !(rsq < kernel_i.HH || rsq < kernel_j.HH))
                            continue;
                        n_pair++;
                        const double r = sqrt(rsq);
                        const double p_over_rho2_j = o.pressure / (o.eomdensity * o.eomdensity);
                        const double soundspeed_j = o.soundspeed;
                        double vsig = soundspeed_i + soundspeed_j;
                        if(vsig > MaxSignalVel)
                            MaxSignalVel = vsig;
                        const double dv0 = me.vx - o.vx, dv1 = me.vy - o.vy, dv2 = me.vz - o.vz;
                        const double vdotr = d0 * dv0 + d1 * dv1 + d2 * dv2;
                        const double vdotr2 = vdotr + C.hubble_a2 * rsq;
                        const double dwk_i = kernel_dwk(kernel_i, C.ktype, r * kernel_i.Hinv);
                        const double dwk_j = kernel_dwk(kernel_j, C.ktype, r * kernel_j.Hinv);
                        double visc = 0;
                        if(vdotr2 < 0) { // Gadget-2 eqs. 13-14, hydra.c:435-462
                            const double mu_ij = C.fac_mu * vdotr2 / r;
                            const double rho_ij = 0.5 * (IDensity + o.density);
                            double vs = soundspeed_i + soundspeed_j;
                            vs -= 3 * mu_ij;
                            if(vs > MaxSignalVel)
                                MaxSignalVel = vs;
                            visc = 0.25 * HP.ArtBulkViscConst * vs * (-mu_ij) / rho_ij * (IF1 + o.f2);
                            const double dloga = 2 * fmax(me.dloga, o.dloga);
                            if(dloga > 0 && (dwk_i + dwk_j) < 0) {
                                if((IMass + s.m) > 0)
                                    visc = fmin(visc, 0.5 * C.fac_vsic_fix * vdotr2 / (0.5 * (IMass + s.m) * (dwk_i + dwk_j) * r * dloga));
                            }
                        }
                        const double hfc_visc = 0.5 * s.m * visc * (dwk_i + dwk_j) / r;
                        double hfc = hfc_visc;
                        double rr1 = 1, rr2 = 1;
                        if(HP.DensityIndependentSphOn) {
                            rr1 = 0, rr2 = 0;
                            hfc += s.m * (dwk_i * p_over_rho2_i * o.entvarpred / me.entvarpred + dwk_j * p_over_rho2_j * me.entvarpred / o.entvarpred) / r;
                            if(HP.DensityContrastLimit >= 0) {
                                rr1 = IEgyRho / IDensity;
                                rr2 = o.eomdensity / o.density;
                                if(HP.DensityContrastLimit > 0) {
                                    rr1 = fmin(rr1, HP.DensityContrastLimit);
                                    rr2 = fmin(rr2, HP.DensityContrastLimit);
                                }
                            }
                        }
                        hfc += s.m * (p_over_rho2_i * me.dhsml * dwk_i * rr1 + p_over_rho2_j * o.dhsml * dwk_j * rr2) / r;
                        Acc0 += -hfc * d0;
                        Acc1 += -hfc * d1;
                        Acc2 += -hfc * d2;
                        DtEntropy += 0.5 * hfc_visc * vdotr2;
                    }
                    no = lk.sibling;
                    continue;
                }
                no = no + 1;
            }
            // hydro_reduce (assign) + hydro_postprocess, hydra.c:279-294, 514-528
            A.hydroacc_out[3 * (int64_t)i] = Acc0;
            A.hydroacc_out[3 * (int64_t)i + 1] = Acc1;
            A.hydroacc_out[3 * (int64_t)i + 2] = Acc2;
            A.maxsignalvel[i] = MaxSignalVel;
            A.dtentropy_out[i] = DtEntropy * (SPH_GAMMA_MINUS1 / (C.hubble_a2 * pow(IDensity, SPH_GAMMA_MINUS1)));
        }
    }
    for(int off = 32; off > 0; off >>= 1) {
        n_cand += __shfl_down(n_cand, off);
        n_pair += __shfl_down(n_pair, off);
    }
    if((threadIdx.x & 63) == 0 && stats) {
        atomicAdd(&stats[0], n_cand);
        atomicAdd(&stats[1], n_pair);
    }
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

double sph_desnumngb(const mpg_density_params &P)
{
    const int t = kernel_index(P.DensityKernelType);
    const double support = t == 0 ? 2. : (t == 1 ? 3. : 2.5);
    return NORM_COEFF * pow(support * P.DensityResolutionEta, NUMDIMS); // density_kernel_desnumngb, densitykernel.c:124-131
}

void SphEngine::density(TreeBuilder &tree, const SphView &A, const mpg_sph_times &T, const mpg_density_params &P, double force_softening,
                        const int *d_active, int64_t nactive, int64_t n, int update_hsml, int DoEgyDensity, int BlackHoleOn, hipStream_t st)
{
    const TreeView tv = tree.view();
    MPG_CHECK(tv.npart > 0 || n == 0, "density: the tree holds no gas particles");
    const int64_t nact = d_active ? nactive : n;
    left.reserve(n + 1);
    right.reserve(n + 1);
    numngb.reserve(n + 1);
    entvarpred.reserve(n + 1);
    queue_a.reserve(nact + 1);
    queue_b.reserve(nact + 1);
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
    if(nact > 0)
        hipLaunchKernelGGL(k_density_init, dim3(nblk(nact)), dim3(256), 0, st, nact, d_active, A, C, tv.box, queue_a.p, ctr.p);
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
        hipLaunchKernelGGL(k_density, dim3(nblk(nq)), dim3(256), 0, st, tv, A, T, P, C, aux.p, qa, (int64_t)nq, qb, ctr.p + 1, stats.p);
        MPG_HIP(hipGetLastError());
        if(!update_hsml)
            break;
        unsigned nr = 0;
        MPG_HIP(hipMemcpyAsync(&nr, ctr.p + 1, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        MPG_HIP(hipStreamSynchronize(st));
        nq = nr;
        int *t = qa;
        qa = qb;
        qb = t;
        if(nq > 0 && last_iterations > 400) // MAXITER, treewalk.c:1362-1364
            fail(__FILE__, __LINE__, "failed to converge density for " + std::to_string(nq) + " particles");
    }
    if(update_hsml && tv.npart > 0) {
        // update_tree_hmax_father for every finished particle (density.c:551-553) == leaf hmax from the final Hsml
        hsml_tree.reserve(tv.npart + 1);
        hipLaunchKernelGGL(k_hsml_treeorder, dim3(nblk(tv.npart)), dim3(256), 0, st, tv.npart, tv.order, A, hsml_tree.p);
        hmax_pending = true;
    }
    unsigned long long hs[2] = {0, 0};
    MPG_HIP(hipMemcpyAsync(hs, stats.p, sizeof(hs), hipMemcpyDeviceToHost, st));
    MPG_HIP(hipStreamSynchronize(st));
    last_interactions = (int64_t)hs[0];
    last_candidates = (int64_t)hs[1];
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
    tree.calc_hmax(hsml_tree.p, st);
    hmax_pending = false;
}

void SphEngine::hydro_force(TreeBuilder &tree, const SphView &A, const mpg_sph_times &T, const mpg_density_params &P, const mpg_hydro_params &HP,
                            const int *d_active, int64_t nactive, int64_t n, hipStream_t st)
{
    const TreeView tv = tree.view();
    MPG_CHECK(tree.has_hmax && tv.hmax, "Hydro called before hmax computed"); // hydra.c:172-173
    MPG_CHECK(entvarpred.p != nullptr, "hydro_force needs the predicted entropies of density()");
    hsrc.reserve(tv.npart + 1);
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
        hipLaunchKernelGGL(k_hydro_prepare, dim3(nblk(tv.npart)), dim3(256), 0, st, tv.npart, tv.order, A, T, HP, entvarpred.p, C.fac_mu, hsrc.p);
    }
    const int64_t nt = d_active ? nactive : tv.npart;
    if(nt > 0)
        hipLaunchKernelGGL(k_hydro, dim3(nblk(nt)), dim3(256), 0, st, tv, A, T, HP, C, hsrc.p, slot_of.p, d_active, nt, stats.p);
    MPG_HIP(hipGetLastError());
    unsigned long long hs[2] = {0, 0};
    MPG_HIP(hipMemcpyAsync(hs, stats.p, sizeof(hs), hipMemcpyDeviceToHost, st));
    MPG_HIP(hipStreamSynchronize(st));
    last_candidates = (int64_t)hs[0];
    last_interactions = (int64_t)hs[1];
}

} // namespace mpg
