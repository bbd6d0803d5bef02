/* oracle/sph_oracle.c
 *
 * TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, fp64) of MP-Gadget's SPH density and hydro-force loops:
 * libgadget/densitykernel.c, density.c, hydra.c and the neighbour visitors of treewalk.c.  It is the checker for the
 * HIP SPH kernels; nothing in the product path may use it.
 *
 * Pinning: the kernel functions are checked bit-for-bit against oracle/_ref (densitykernel.c compiled in place);
 * the density loop against the reference's known answers of libgadget/tests/test_density.c (mean Hsml 0.501747 +- 1e-4
 * on the 32^3 grid, cubic spline, eta = 1, MaxNumNgbDeviation = 2; stability under MaxNumNgbDeviation 0.5, :126-147).
 * hydra.c has no known answer in the reference's tests and cannot be built here; since round 3 os_hydro_force (and the derived
 * fields of os_density) are pinned by PHYSICS instead: tests/sph_paper.py states the SPH equations from the publications and the
 * comoving-variable physics (not from hydra.c) and evaluates them over all pairs; tests/test_hydro_physics.py requires agreement to
 * 2e-10 in both SPH formulations, at a = 1 and at a cosmological epoch, with and without the bound on the viscous force, adds
 * closed-form states and an energy balance, and shows that seven deliberate mutations of this file each fail a gate.
 *
 * Single rank, trivial domain: export / ghost machinery (treewalk.c:325-793) is not restated.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "oracle_tree.h"

#define NUMDIMS 3
#define NORM_COEFF 4.188790204786 /* densitykernel.h:6 */
#define GAMMA (5.0 / 3.0)         /* physconst.h:35-36 */
#define GAMMA_MINUS1 (GAMMA - 1)
#define FACT1 0.366025403785      /* treewalk.c:19 */
#define MAXITER 400               /* treewalk.h */
#define TIMEBINS 46
#define DMIN(a, b) (((a) < (b)) ? (a) : (b))

/* ------------------------------ densitykernel.c ------------------------------ */
typedef struct {
    double H, HH, Hinv;
    int type; /* 0 cubic, 1 quintic, 2 quartic (index into KERNELS, densitykernel.c:91-105) */
    double support, Wknorm, dWknorm;
} okernel;

static double wk_cs(double q) /* densitykernel.c:24-32 */
{
    if(q < 1.0)
        return 0.25 * pow(2 - q, 3) - pow(1 - q, 3);
    if(q < 2.0)
        return 0.25 * pow(2 - q, 3);
    return 0.0;
}
static double dwk_cs(double q) /* :33-41 */
{
    if(q < 1.0)
        return -0.25 * 3 * pow(2 - q, 2) + 3 * pow(1 - q, 2);
    if(q < 2.0)
        return -0.25 * 3 * pow(2 - q, 2);
    return 0.0;
}
static double wk_qus(double q) /* :42-53 */
{
    if(q < 0.5)
        return pow(2.5 - q, 4) - 5 * pow(1.5 - q, 4) + 10 * pow(0.5 - q, 4);
    if(q < 1.5)
        return pow(2.5 - q, 4) - 5 * pow(1.5 - q, 4);
    if(q < 2.5)
        return pow(2.5 - q, 4);
    return 0.0;
}
static double dwk_qus(double q) /* :54-65 */
{
    if(q < 0.5)
        return -4 * pow(2.5 - q, 3) + 20 * pow(1.5 - q, 3) - 40 * pow(0.5 - q, 3);
    if(q < 1.5)
        return -4 * pow(2.5 - q, 3) + 20 * pow(1.5 - q, 3);
    if(q < 2.5)
        return -4 * pow(2.5 - q, 3);
    return 0.0;
}
static double wk_qs(double q) /* :66-77 */
{
    if(q < 1.0)
        return pow(3 - q, 5) - 6 * pow(2 - q, 5) + 15 * pow(1 - q, 5);
    if(q < 2.0)
        return pow(3 - q, 5) - 6 * pow(2 - q, 5);
    if(q < 3.0)
        return pow(3 - q, 5);
    return 0.0;
}
static double dwk_qs(double q) /* :78-90 */
{
    if(q < 1.0)
        return -5 * pow(3 - q, 4) + 30 * pow(2 - q, 4) - 75 * pow(1 - q, 4);
    if(q < 2.0)
        return -5 * pow(3 - q, 4) + 30 * pow(2 - q, 4);
    if(q < 3.0)
        return -5 * pow(3 - q, 4);
    return 0.0;
}

static const double K_SUPPORT[3] = {2., 3., 2.5};
static const double K_SIGMA3[3] = {1 / M_PI, 1 / (120 * M_PI), 1 / (20 * M_PI)}; /* sigma[NUMDIMS-1] */

/* enum DensityKernelType (densitykernel.h:17-21): 1 cubic, 2 quintic, 4 quartic -> table index (densitykernel.c:155-171) */
int os_kernel_index(int enumtype) { return enumtype == 1 ? 0 : (enumtype == 2 ? 1 : (enumtype == 4 ? 2 : -1)); }

void os_kernel_init(okernel *k, double H, int type) /* densitykernel.c:136-153 */
{
    k->H = H;
    k->HH = H * H;
    k->Hinv = 1. / H;
    k->type = type;
    k->support = K_SUPPORT[type];
    const double hinv = k->Hinv * k->support;
    k->Wknorm = K_SIGMA3[type] * pow(hinv, NUMDIMS);
    k->dWknorm = k->Wknorm * hinv;
}
double os_kernel_wk(const okernel *k, double u) /* :116-122 */
{
    const double q = u * K_SUPPORT[k->type];
    return k->Wknorm * (k->type == 0 ? wk_cs(q) : (k->type == 1 ? wk_qs(q) : wk_qus(q)));
}
double os_kernel_dwk(const okernel *k, double u) /* :108-114 */
{
    const double q = u * K_SUPPORT[k->type];
    return k->dWknorm * (k->type == 0 ? dwk_cs(q) : (k->type == 1 ? dwk_qs(q) : dwk_qus(q)));
}
double os_kernel_desnumngb(int type, double eta) { return NORM_COEFF * pow(K_SUPPORT[type] * eta, NUMDIMS); } /* :124-131 */
static double kernel_volume(const okernel *k) { return NORM_COEFF * pow(k->H, NUMDIMS); }                  /* :133-137 */
static double kernel_dW(const okernel *k, double u, double wk, double dwk) { return -(NUMDIMS * k->Hinv * wk + u * dwk); } /* densitykernel.h:46-50 */

/* ------------------------------ inputs ------------------------------ */
typedef struct { /* struct density_params, density.h:10-25 (same order) */
    double DensityResolutionEta, MaxNumNgbDeviation, BlackHoleNgbFactor, BlackHoleMaxAccretionRadius;
    int DensityKernelType; /* enum value 1 / 2 / 4 */
    double MinGasHsmlFractional;
} odens_params;

typedef struct { /* struct hydro_params, hydra.c:26-34 */
    int DensityIndependentSphOn;
    double DensityContrastLimit, ArtBulkViscConst;
} ohydro_params;

/* time-dependent scalars the callers derive from DriftKickTimes / Cosmology */
typedef struct {
    double FgravkickB;              /* kick_factor_data, density.h:34-39 */
    double gravkicks[TIMEBINS + 1];
    double hydrokicks[TIMEBINS + 1];
    double drifts[TIMEBINS + 1];    /* hydra.c:178-186 */
    double dloga_kick[TIMEBINS + 1];/* dloga_from_dti(Ti_Current - Ti_kick[bin]) of SPH_EntVarPred, density.c:75 */
    double dloga_bin[TIMEBINS + 1]; /* get_dloga_for_bin(bin, Ti_Current), hydra.c:271,463 */
    double atime, hubble;
} osph_times;

/* particle table in caller order (n entries each); SPH slot fields are indexed by particle, not by PI */
typedef struct {
    int64_t n;
    const double *pos;
    const float *mass;
    const int *type;
    double *hsml, *dthsml;
    const double *vel, *gacc, *gpm, *hydroacc_in;
    const unsigned char *tb_hydro, *tb_grav;
    const double *entropy, *dtentropy_in;
    double *density, *egywtdensity, *dhsmlegyfac, *divvel, *curlvel; /* density outputs */
    double *numngb;                                                  /* scratch / diagnostic, n */
    double *gradrho;                                                 /* optional n x 3 */
    double *hydroacc_out, *dtentropy_out, *maxsignalvel;             /* hydro outputs */
    double *entvarpred;                                              /* n: filled by density, reused by hydro */
} osph_arrays;

static double FORCE_SOFTENING_G = 0; /* set through os_set_softening: MinGasHsml = frac * FORCE_SOFTENING()/2.8, density.c:268 */
void os_set_softening(double force_softening) { FORCE_SOFTENING_G = force_softening; }

static void sph_velpred(const osph_arrays *A, const osph_times *T, int64_t i, double *v) /* SPH_VelPred, density.c:91-100 */
{
    for(int j = 0; j < 3; j++)
        v[j] = A->vel[3 * i + j] + T->gravkicks[A->tb_grav ? A->tb_grav[i] : 0] * (A->gacc ? A->gacc[3 * i + j] : 0.0) +
               (A->gpm ? A->gpm[3 * i + j] : 0.0) * T->FgravkickB +
               T->hydrokicks[A->tb_hydro ? A->tb_hydro[i] : 0] * (A->hydroacc_in ? A->hydroacc_in[3 * i + j] : 0.0);
}

static double sph_entvarpred(const osph_arrays *A, const osph_times *T, int64_t i) /* SPH_EntVarPred, density.c:69-86 */
{
    const int bin = A->tb_hydro ? A->tb_hydro[i] : 0;
    const double dloga = T->dloga_kick[bin];
    double e = A->entropy[i] + (A->dtentropy_in ? A->dtentropy_in[i] : 0.0) * dloga;
    if(e < 0.05 * A->entropy[i])
        e = 0.05 * A->entropy[i];
    if(e <= 0)
        return 0;
    return exp(1. / GAMMA * log(e));
}

/* ------------------------------ density ------------------------------ */
typedef struct {
    double EgyRho, DhsmlEgyDensity, Rho, DhsmlDensity, Ngb, Div, Rot[3], GradRho[3];
} odens_result;

/* treewalk_visit_nolist_ngbiter (treewalk.c:1152-1265) + cull_node asymmetric (:1015-1042) + density_ngbiter
 * (density.c:424-519) for one target.  Returns the number of particles distance-tested with success (ninteractions). */
static int64_t density_visit(const otree *tree, const osph_arrays *A, const osph_times *T, int ktype, int DoEgy, int want_grad,
                             int target_type, const double *ipos, const double *ivel, double hsml, odens_result *O, int64_t *ncand)
{
    okernel kernel;
    os_kernel_init(&kernel, hsml, ktype);
    const double kvol = kernel_volume(&kernel);
    const double Box = tree->box;
    int64_t nint = 0, ncnd = 0;
    int no = (int)tree->firstnode;
    while(no >= 0) {
        const onode *cur = &tree->nodes[no];
        { /* cull_node */
            double dist = hsml + 0.5 * cur->len;
            double r2 = 0;
            int culled = 0;
            for(int d = 0; d < 3; d++) {
                const double dx = NEAREST(cur->center[d] - ipos[d], Box);
                if(dx > dist || dx < -dist) {
                    culled = 1;
                    break;
                }
                r2 += dx * dx;
            }
            if(!culled) {
                dist += FACT1 * cur->len;
                if(r2 > dist * dist)
                    culled = 1;
            }
            if(culled) {
                no = cur->sibling;
                continue;
            }
        }
        if(cur->ChildType == PARTICLE_NODE_TYPE) {
            for(int k = 0; k < cur->noccupied; k++) {
                const int other = cur->suns[k];
                ncnd++;
                double dist[3], r2 = 0;
                const double h2 = hsml * hsml;
                int d;
                for(d = 0; d < 3; d++) {
                    dist[d] = NEAREST(ipos[d] - tree->pos[3 * (size_t)other + d], Box);
                    r2 += dist[d] * dist[d];
                    if(r2 > h2)
                        break;
                }
                if(r2 > h2)
                    continue;
                nint++;
                const double r = sqrt(r2);
                if(r2 < kernel.HH) { /* density_ngbiter */
                    const double u = r * kernel.Hinv;
                    const double wk = os_kernel_wk(&kernel, u);
                    O->Ngb += wk * kvol;
                    const double dwk = os_kernel_dwk(&kernel, u);
                    const double mass_j = A->mass[other];
                    O->Rho += mass_j * wk;
                    const double density_dW = kernel_dW(&kernel, u, wk, dwk);
                    O->DhsmlDensity += mass_j * density_dW;
                    double VelPred[3];
                    sph_velpred(A, T, other, VelPred);
                    if(DoEgy) {
                        const double EntVarPred = A->entvarpred[other];
                        O->EgyRho += mass_j * EntVarPred * wk;
                        O->DhsmlEgyDensity += mass_j * EntVarPred * density_dW;
                    }
                    if(r > 0) {
                        const double fac = mass_j * dwk / r;
                        double dv[3], rot[3];
                        for(d = 0; d < 3; d++)
                            dv[d] = ivel[d] - VelPred[d];
                        O->Div += -fac * (dist[0] * dv[0] + dist[1] * dv[1] + dist[2] * dv[2]);
                        rot[0] = dv[1] * dist[2] - dist[1] * dv[2]; /* crossproduct(dv, dist), densitykernel.h:63-76 */
                        rot[1] = dv[2] * dist[0] - dist[2] * dv[0];
                        rot[2] = dv[0] * dist[1] - dist[0] * dv[1];
                        for(d = 0; d < 3; d++)
                            O->Rot[d] += fac * rot[d];
                        if(want_grad)
                            for(d = 0; d < 3; d++)
                                O->GradRho[d] += fac * dist[d];
                    }
                }
            }
            no = cur->sibling;
            continue;
        }
        else if(cur->ChildType == PSEUDO_NODE_TYPE) {
            no = cur->sibling;
            continue;
        }
        no = cur->suns[0];
    }
    (void)target_type;
    if(ncand)
        *ncand += ncnd;
    return nint;
}

/* update_tree_hmax_father, forcetree.c:1286-1315 */
static void update_hmax_father(otree *tree, int i, const double *pos, double hsml)
{
    const int no = tree->father[i];
    if(no < 0)
        return;
    onode *node = &tree->nodes[no];
    double newhmax = 0;
    for(int j = 0; j < 3; j++)
        newhmax = DMAX(newhmax, fabs(pos[j] - node->center[j]) + hsml - node->len / 2.);
#pragma omp critical(hmax)
    {
        if(newhmax > node->hmax)
            node->hmax = newhmax;
    }
}

/* density_check_neighbours, density.c:589-689.  Returns 1 when done. */
static int check_neighbours(int i, const otree *tree, osph_arrays *A, const odens_params *P, double desnumngb0, double MinGasHsml,
                            int BlackHoleOn, double *Left, double *Right, const double *NumNgb, const double *DhsmlDensityFactor)
{
    double desnumngb = desnumngb0;
    const int ty = A->type ? A->type[i] : 0;
    if(BlackHoleOn && ty == 5)
        desnumngb = desnumngb * P->BlackHoleNgbFactor;
    if(NumNgb[i] < (desnumngb - P->MaxNumNgbDeviation) || (NumNgb[i] > (desnumngb + P->MaxNumNgbDeviation))) {
        if((Right[i] - Left[i]) < 1.0e-5 * Left[i]) {
            A->hsml[i] = Right[i];
            return 1;
        }
        if(NumNgb[i] < desnumngb)
            Left[i] = A->hsml[i];
        else
            Right[i] = A->hsml[i];
        if((Right[i] < tree->box && Left[i] > 0) || (A->hsml[i] * 1.26 > 0.99 * tree->box))
            A->hsml[i] = cbrt(0.5 * (pow(Left[i], 3) + pow(Right[i], 3)));
        else {
            const double DensFac = DhsmlDensityFactor[i];
            double fac = 1.26;
            if(NumNgb[i] > 0)
                fac = 1 - (NumNgb[i] - desnumngb) / (NUMDIMS * NumNgb[i]) * DensFac;
            if(Right[i] > 0.99 * tree->box && Left[i] > 0)
                if(DensFac <= 0 || fabs(NumNgb[i] - desnumngb) >= 0.5 * desnumngb || fac > 1.26)
                    fac = 1.26;
            if(Right[i] < 0.99 * tree->box && Left[i] == 0)
                if(DensFac <= 0 || fac < 1. / 3)
                    fac = 1. / 3;
            A->hsml[i] *= fac;
        }
        if(BlackHoleOn && ty == 5)
            if(Left[i] > P->BlackHoleMaxAccretionRadius) {
                A->hsml[i] = P->BlackHoleMaxAccretionRadius;
                return 1;
            }
        if(Right[i] < MinGasHsml) {
            A->hsml[i] = MinGasHsml;
            return 1;
        }
        return 0;
    }
    else {
        if(BlackHoleOn && ty == 5)
            if(A->hsml[i] > P->BlackHoleMaxAccretionRadius)
                A->hsml[i] = P->BlackHoleMaxAccretionRadius;
        if(A->hsml[i] < MinGasHsml)
            A->hsml[i] = MinGasHsml;
        return 1;
    }
}

/* density(), density.c:234-355, with treewalk_do_hsml_loop (treewalk.c:1269-1367).
 * stats: [0] iterations, [1] sum of targets over iterations, [2] successful distance tests, [3] candidates tested.
 * Returns 0, or -1 if MAXITER was exceeded. */
int os_density(otree *tree, const odens_params *P, osph_arrays *A, const osph_times *T, int64_t nactive, const int *active,
               int update_hsml, int DoEgyDensity, int BlackHoleOn, int64_t *stats)
{
    const int64_t n = A->n;
    const int ktype = os_kernel_index(P->DensityKernelType);
    const double DesNumNgb = os_kernel_desnumngb(ktype, P->DensityResolutionEta);
    const double MinGasHsml = P->MinGasHsmlFractional * (FORCE_SOFTENING_G / 2.8);
    double *Left = (double *)malloc(sizeof(double) * n), *Right = (double *)malloc(sizeof(double) * n);
    double *DhsmlDensityFactor = (double *)malloc(sizeof(double) * n), *Rot = (double *)malloc(sizeof(double) * 3 * n);
    double *NumNgb = A->numngb ? A->numngb : (double *)malloc(sizeof(double) * n);
    const int64_t nact = active ? nactive : n;
    /* queue: haswork = gas or BH (density.c:521-530) */
    int *queue = (int *)malloc(sizeof(int) * (nact > 0 ? nact : 1));
    int64_t size = 0;
    for(int64_t k = 0; k < nact; k++) {
        const int i = active ? active[k] : (int)k;
        Right[i] = tree->box;
        NumNgb[i] = 0;
        Left[i] = 0;
        const int ty = A->type ? A->type[i] : 0;
        if(ty == 0 || ty == 5)
            queue[size++] = i;
    }
    /* EntVarPred for all gas (density.c:296-301) */
    for(int64_t i = 0; i < n; i++)
        if(!A->type || A->type[i] == 0)
            A->entvarpred[i] = sph_entvarpred(A, T, i);
    int64_t s_iter = 0, s_targets = 0, s_int = 0, s_cand = 0;
    int rc = 0;
    int *redo = (int *)malloc(sizeof(int) * (size > 0 ? size : 1));
    while(size > 0) {
        s_iter++;
        s_targets += size;
        int64_t nredo = 0, it_int = 0, it_cand = 0;
#pragma omp parallel for schedule(dynamic, 32) reduction(+ : it_int, it_cand)
        for(int64_t k = 0; k < size; k++) {
            const int i = queue[k];
            const int ty = A->type ? A->type[i] : 0;
            double ivel[3];
            if(ty != 0) {
                for(int j = 0; j < 3; j++)
                    ivel[j] = A->vel[3 * (size_t)i + j];
            }
            else
                sph_velpred(A, T, i, ivel); /* density_copy, density.c:357-372 */
            odens_result O;
            memset(&O, 0, sizeof(O));
            int64_t nc = 0;
            it_int += density_visit(tree, A, T, ktype, DoEgyDensity, A->gradrho != NULL, ty, &A->pos[3 * (size_t)i], ivel, A->hsml[i], &O, &nc);
            it_cand += nc;
            /* density_reduce (PRIMARY: assign), density.c:374-409 */
            NumNgb[i] = O.Ngb;
            DhsmlDensityFactor[i] = O.DhsmlDensity;
            A->density[i] = O.Rho;
            A->divvel[i] = O.Div;
            if(ty == 0) {
                Rot[3 * (size_t)i + 0] = O.Rot[0];
                Rot[3 * (size_t)i + 1] = O.Rot[1];
                Rot[3 * (size_t)i + 2] = O.Rot[2];
                if(A->gradrho)
                    for(int d = 0; d < 3; d++)
                        A->gradrho[3 * (size_t)i + d] = O.GradRho[d];
                if(DoEgyDensity) {
                    A->egywtdensity[i] = O.EgyRho;
                    A->dhsmlegyfac[i] = O.DhsmlEgyDensity;
                }
            }
            /* density_postprocess, density.c:532-586 */
            double *DhsmlDens = &DhsmlDensityFactor[i];
            const double density = A->density[i];
            *DhsmlDens *= A->hsml[i] / (NUMDIMS * density);
            *DhsmlDens = 1 / (1 + *DhsmlDens);
            const double hsml_used = A->hsml[i];
            int done = 1;
            if(update_hsml) {
                done = check_neighbours(i, tree, A, P, DesNumNgb, MinGasHsml, BlackHoleOn, Left, Right, NumNgb, DhsmlDensityFactor);
                if(done && tree->father && ty == 0)
                    update_hmax_father(tree, i, &A->pos[3 * (size_t)i], A->hsml[i]);
                if(!done) {
                    int64_t slot;
#pragma omp atomic capture
                    slot = nredo++;
                    redo[slot] = i;
                }
            }
            (void)hsml_used;
            if(ty == 0) {
                if(DoEgyDensity) {
                    const double EntPred = A->entvarpred[i];
                    A->dhsmlegyfac[i] *= A->hsml[i] / (NUMDIMS * A->egywtdensity[i]);
                    A->dhsmlegyfac[i] *= -(*DhsmlDens);
                    A->egywtdensity[i] /= EntPred;
                }
                else
                    A->dhsmlegyfac[i] = *DhsmlDens;
                const double *R = &Rot[3 * (size_t)i];
                A->curlvel[i] = sqrt(R[0] * R[0] + R[1] * R[1] + R[2] * R[2]) / A->density[i];
                A->divvel[i] /= A->density[i];
                if(A->dthsml)
                    A->dthsml[i] = (1.0 / NUMDIMS) * A->divvel[i] * A->hsml[i];
            }
            else if(ty == 5) {
                A->divvel[i] /= A->density[i];
                if(A->dthsml)
                    A->dthsml[i] = (1.0 / NUMDIMS) * A->divvel[i] * A->hsml[i];
            }
        }
        s_int += it_int;
        s_cand += it_cand;
        if(!update_hsml)
            break;
        /* the reference's redo queue order depends on threads; sort for reproducibility (order does not affect results) */
        size = nredo;
        for(int64_t k = 0; k < size; k++)
            queue[k] = redo[k];
        if(size > 0 && s_iter > MAXITER) {
            rc = -1;
            break;
        }
    }
    if(stats) {
        stats[0] = s_iter;
        stats[1] = s_targets;
        stats[2] = s_int;
        stats[3] = s_cand;
    }
    free(redo);
    free(queue);
    free(Left);
    free(Right);
    free(DhsmlDensityFactor);
    free(Rot);
    if(!A->numngb)
        free(NumNgb);
    return rc;
}

/* set_init_hsml, density.c:691-749 (after force_tree_calc_moments, which the caller has run). */
void os_set_init_hsml(const otree *tree, const odens_params *P, osph_arrays *A, double MeanGasSeparation)
{
    const int ktype = os_kernel_index(P->DensityKernelType);
    const double DesNumNgb = os_kernel_desnumngb(ktype, P->DensityResolutionEta);
    for(int64_t i = 0; i < A->n; i++) {
        const int ty = A->type ? A->type[i] : 0;
        if(ty != 0 && ty != 5)
            continue;
        int no = (int)i;
        do {
            const int p = (no >= tree->firstnode) ? tree->nodes[no].father : tree->father[no];
            if(p < tree->firstnode)
                break;
            no = p;
        } while(10 * DesNumNgb * A->mass[i] > tree->nodes[no].mass);
        A->hsml[i] = MeanGasSeparation;
        if(no >= tree->firstnode) {
            const double testhsml = tree->nodes[no].len * pow(3.0 / (4 * M_PI) * DesNumNgb * A->mass[i] / tree->nodes[no].mass, 1.0 / 3);
            if(testhsml < 500. * MeanGasSeparation)
                A->hsml[i] = testhsml;
        }
    }
}

/* ------------------------------ hydro ------------------------------ */
static double sph_density_pred(double Density, double DivVel, double dtdrift) /* SPH_DensityPred, hydra.c:300-312 */
{
    const double DensityPred = Density - DivVel * Density * dtdrift;
    if(DensityPred >= 1e-6 * Density)
        return DensityPred;
    return 1e-6 * Density;
}
static double pressure_pred(double EOMDensityPred, double EntVarPred) /* PressurePred, hydra.c:62-76 */
{
    if(EntVarPred * EOMDensityPred <= 0)
        return 0;
    return exp(GAMMA * log(EntVarPred * EOMDensityPred));
}

/* hydro_force(), hydra.c:153-245: treewalk_visit_ngbiter (treewalk.c:930-1007) with the symmetric cull_node (:1015-1042),
 * hydro_copy (:247-277), hydro_ngbiter (:318-506), hydro_reduce / hydro_postprocess (:279-294, :514-528).
 * tree must carry hmax (ot_calc_moments after the density loop, run.c:477).  stats: [0] candidates, [1] pairs evaluated. */
void os_hydro_force(const otree *tree, const odens_params *DP, const ohydro_params *HP, osph_arrays *A, const osph_times *T,
                    int64_t nactive, const int *active, int64_t *stats)
{
    const int64_t n = A->n;
    const int ktype = os_kernel_index(DP->DensityKernelType);
    const double Box = tree->box;
    const double atime = T->atime, hubble = T->hubble;
    const double fac_mu = pow(atime, 3 * (GAMMA - 1) / 2) / atime;
    const double fac_vsic_fix = hubble * pow(atime, 3 * GAMMA_MINUS1);
    const double hubble_a2 = hubble * atime * atime;
    /* PressurePred for all gas (hydra.c:195-214); EntVarPred comes from density() */
    double *Pressure = (double *)malloc(sizeof(double) * n);
    for(int64_t i = 0; i < n; i++) {
        Pressure[i] = 0;
        if(A->type && A->type[i] != 0)
            continue;
        if(A->entvarpred[i] == 0)
            continue;
        const int bin = A->tb_hydro ? A->tb_hydro[i] : 0;
        const double eom = sph_density_pred(HP->DensityIndependentSphOn ? A->egywtdensity[i] : A->density[i], A->divvel[i], T->drifts[bin]);
        Pressure[i] = pressure_pred(eom, A->entvarpred[i]);
    }
    const int64_t nact = active ? nactive : n;
    int64_t s_cand = 0, s_pair = 0;
#pragma omp parallel reduction(+ : s_cand, s_pair)
    {
        int *ngblist = (int *)malloc(sizeof(int) * (tree->ninserted > 0 ? tree->ninserted : 1));
#pragma omp for schedule(dynamic, 32)
        for(int64_t kk = 0; kk < nact; kk++) {
            const int i = active ? active[kk] : (int)kk;
            if(A->type && A->type[i] != 0)
                continue; /* hydro_haswork */
            /* hydro_copy */
            double IVel[3];
            sph_velpred(A, T, i, IVel);
            const double IHsml = A->hsml[i], IMass = A->mass[i], IDensity = A->density[i], IEgyRho = A->egywtdensity ? A->egywtdensity[i] : 0;
            const double IEntVarPred = A->entvarpred[i];
            const double IDhsml = A->dhsmlegyfac[i];
            const double eomdensity_i = HP->DensityIndependentSphOn ? IEgyRho : IDensity;
            const double IPressure = Pressure[i];
            const double Idloga = T->dloga_bin[A->tb_hydro ? A->tb_hydro[i] : 0];
            const double soundspeed_c = sqrt(GAMMA * IPressure / eomdensity_i);
            const double IF1 = fabs(A->divvel[i]) / (fabs(A->divvel[i]) + A->curlvel[i] + 0.0001 * soundspeed_c / IHsml / fac_mu);
            /* iterator start (other == -1) */
            double soundspeed_i, p_over_rho2_i;
            if(HP->DensityIndependentSphOn) {
                soundspeed_i = sqrt(GAMMA * IPressure / IEgyRho);
                p_over_rho2_i = IPressure / (IEgyRho * IEgyRho);
            }
            else {
                soundspeed_i = sqrt(GAMMA * IPressure / IDensity);
                p_over_rho2_i = IPressure / (IDensity * IDensity);
            }
            okernel kernel_i;
            os_kernel_init(&kernel_i, IHsml, ktype);
            double Acc[3] = {0, 0, 0}, DtEntropy = 0, MaxSignalVel = soundspeed_i;
            const double *ipos = &A->pos[3 * (size_t)i];
            /* ngb_treefind_threads with the symmetric cull */
            int numcand = 0;
            int no = (int)tree->firstnode;
            while(no >= 0) {
                const onode *cur = &tree->nodes[no];
                double dist = DMAX(cur->hmax, IHsml) + 0.5 * cur->len;
                double r2 = 0;
                int culled = 0;
                for(int d = 0; d < 3; d++) {
                    const double dx = NEAREST(cur->center[d] - ipos[d], Box);
                    if(dx > dist || dx < -dist) {
                        culled = 1;
                        break;
                    }
                    r2 += dx * dx;
                }
                if(!culled) {
                    dist += FACT1 * cur->len;
                    if(r2 > dist * dist)
                        culled = 1;
                }
                if(culled) {
                    no = cur->sibling;
                    continue;
                }
                if(cur->ChildType == PARTICLE_NODE_TYPE) {
                    for(int k = 0; k < cur->noccupied; k++)
                        ngblist[numcand++] = cur->suns[k];
                    no = cur->sibling;
                    continue;
                }
                else if(cur->ChildType == PSEUDO_NODE_TYPE) {
                    no = cur->sibling;
                    continue;
                }
                no = cur->suns[0];
            }
            s_cand += numcand;
            for(int c = 0; c < numcand; c++) {
                const int other = ngblist[c];
                const double hh = DMAX(A->hsml[other], IHsml);
                const double h2 = hh * hh;
                double dist[3], rsq = 0;
                int d;
                for(d = 0; d < 3; d++) {
                    dist[d] = NEAREST(ipos[d] - A->pos[3 * (size_t)other + d], Box);
                    rsq += dist[d] * dist[d];
                    if(rsq > h2)
                        break;
                }
                if(rsq > h2)
                    continue;
                const double r = sqrt(rsq);
                /* hydro_ngbiter */
                okernel kernel_j;
                os_kernel_init(&kernel_j, A->hsml[other], ktype);
                if(rsq <= 0 || !(rsq < kernel_i.HH || rsq < kernel_j.HH))
                    continue;
                s_pair++;
                double VelPred[3];
                sph_velpred(A, T, other, VelPred);
                const double EntVarPred = A->entvarpred[other];
                const int bin = A->tb_hydro ? A->tb_hydro[other] : 0;
                const double density_j = sph_density_pred(A->density[other], A->divvel[other], T->drifts[bin]);
                const double eomdensity = sph_density_pred(HP->DensityIndependentSphOn ? A->egywtdensity[other] : A->density[other],
                                                           A->divvel[other], T->drifts[bin]);
                const double Pressure_j = Pressure[other];
                const double p_over_rho2_j = Pressure_j / (eomdensity * eomdensity);
                const double soundspeed_j = sqrt(GAMMA * Pressure_j / eomdensity);
                double vsig = soundspeed_i + soundspeed_j;
                if(vsig > MaxSignalVel)
                    MaxSignalVel = vsig;
                double dv[3];
                for(d = 0; d < 3; d++)
                    dv[d] = IVel[d] - VelPred[d];
                const double vdotr = dist[0] * dv[0] + dist[1] * dv[1] + dist[2] * dv[2];
                const double vdotr2 = vdotr + hubble_a2 * rsq;
                const double dwk_i = os_kernel_dwk(&kernel_i, r * kernel_i.Hinv);
                const double dwk_j = os_kernel_dwk(&kernel_j, r * kernel_j.Hinv);
                double visc = 0;
                if(vdotr2 < 0) {
                    const double mu_ij = fac_mu * vdotr2 / r;
                    const double rho_ij = 0.5 * (IDensity + density_j);
                    double vs = soundspeed_i + soundspeed_j;
                    vs -= 3 * mu_ij;
                    if(vs > MaxSignalVel)
                        MaxSignalVel = vs;
                    const double f2 = fabs(A->divvel[other]) /
                                      (fabs(A->divvel[other]) + A->curlvel[other] + 0.0001 * soundspeed_j / fac_mu / A->hsml[other]);
                    visc = 0.25 * HP->ArtBulkViscConst * vs * (-mu_ij) / rho_ij * (IF1 + f2);
                    const double dloga = 2 * DMAX(Idloga, T->dloga_bin[bin]);
                    if(dloga > 0 && (dwk_i + dwk_j) < 0) {
                        if((IMass + A->mass[other]) > 0)
                            visc = DMIN(visc, 0.5 * fac_vsic_fix * vdotr2 / (0.5 * (IMass + A->mass[other]) * (dwk_i + dwk_j) * r * dloga));
                    }
                }
                const double mj = A->mass[other];
                const double hfc_visc = 0.5 * mj * visc * (dwk_i + dwk_j) / r;
                double hfc = hfc_visc;
                double rr1 = 1, rr2 = 1;
                if(HP->DensityIndependentSphOn) {
                    rr1 = 0, rr2 = 0;
                    hfc += mj * (dwk_i * p_over_rho2_i * EntVarPred / IEntVarPred + dwk_j * p_over_rho2_j * IEntVarPred / EntVarPred) / r;
                    if(HP->DensityContrastLimit >= 0) {
                        rr1 = IEgyRho / IDensity;
                        rr2 = eomdensity / density_j;
                        if(HP->DensityContrastLimit > 0) {
                            rr1 = DMIN(rr1, HP->DensityContrastLimit);
                            rr2 = DMIN(rr2, HP->DensityContrastLimit);
                        }
                    }
                }
                hfc += mj * (p_over_rho2_i * IDhsml * dwk_i * rr1 + p_over_rho2_j * A->dhsmlegyfac[other] * dwk_j * rr2) / r;
                for(d = 0; d < 3; d++)
                    Acc[d] += (-hfc * dist[d]);
                DtEntropy += (0.5 * hfc_visc * vdotr2);
            }
            /* reduce (assign) + postprocess */
            for(int d = 0; d < 3; d++)
                A->hydroacc_out[3 * (size_t)i + d] = Acc[d];
            A->maxsignalvel[i] = MaxSignalVel;
            A->dtentropy_out[i] = DtEntropy * (GAMMA_MINUS1 / (hubble_a2 * pow(A->density[i], GAMMA_MINUS1)));
        }
        free(ngblist);
    }
    if(stats) {
        stats[0] = s_cand;
        stats[1] = s_pair;
    }
    free(Pressure);
}
