/* libgadget/sph-hip.c -- density() and hydro_force() of the reference forwarded to libmpgadget_hip.so.
 *
 * Replaces the three LOOPS of density.c and hydra.c - density(), set_init_hsml(), hydro_force() - whose definitions the maintainer
 * puts under `#ifndef MPGADGET_HIP` there; the small host functions other modules call (SPH_EntVarPred, SPH_VelPred,
 * DM_VelPred, init_kick_factor_data, GetNumNgb, ..., density.c:22-132) stay where they are.  Two one-line hooks hand the module
 * parameters over: `mpg_shim_set_densitypar(&DensityParams);` at the end of set_densitypar() (density.c:22-27) and
 * `mpg_shim_set_hydropar(HydroParams.DensityIndependentSphOn, HydroParams.DensityContrastLimit, HydroParams.ArtBulkViscConst);`
 * at the end of set_hydro_params() (hydra.c:36-48).  The reference keeps the SPH fields in
 * slots, SphP[P[i].PI] (slotsmanager.h:93-129); the C-ABI takes plain arrays in particle order (mpg_sph_arrays), so this file
 * gathers the slot fields before the call and scatters the results after it.  The time-dependent scalars the reference derives
 * before its loops come from the reference's own functions and go over as one mpg_sph_times.
 * Compiled inside the reference tree (see gravity-hip.c).  One rank per GPU.  NTask > 1: mpg_dist_density / mpg_dist_hydro_force on the
 * table gravity-hip.c last handed to mpg_dist_force_tree_full (ghost columns travel inside the library), with the
 * step's ActiveParticle list. */
#include <mpi.h>
#include <math.h>
#include <string.h>
#include "density.h"
#include "hydra.h"
#include "forcetree.h"
#include "partmanager.h"
#include "slotsmanager.h"
#include "timestep.h"
#include "timefac.h"
#include "cosmology.h"
#include "walltime.h"
#include "utils/endrun.h"
#include "utils/mymalloc.h"
#include <mpgadget_hip.h>
#include "mpg_shim.h" /* the rank's engine / multi-rank state, mpg_shim_sync, mpg_shim_ck (gravity-hip.c) */

#define ck mpg_shim_ck

/* the largest smoothing length of the density loop's targets on this rank: the ghosts must cover it (mpg_shim_sync takes the
 * maximum over the ranks).  The loop may still grow Hsml beyond the margin chosen from it (void gas): density() then repeats
 * the call with the margin the library reports (mpg_dist_last_max_hsml). */
static double max_target_hsml(void)
{
    double h = 0;
    int64_t i;
    #pragma omp parallel for reduction(max : h)
    for(i = 0; i < PartManager->NumPart; i++)
        if(!P[i].IsGarbage && !P[i].Swallowed && (P[i].Type == 0 || P[i].Type == 5) && P[i].Hsml > h)
            h = P[i].Hsml;
    return h;
}

void mpg_shim_set_densitypar(const struct density_params *dp)
{
    _Static_assert(sizeof(struct density_params) == sizeof(mpg_density_params), "density_params (density.h:10-25)");
    ck(mpg_set_densitypar(mpg_shim_engine(), (const mpg_density_params *)dp));
}

void mpg_shim_set_hydropar(int DensityIndependentSphOn, double DensityContrastLimit, double ArtBulkViscConst)
{
    mpg_hydro_params hp = {DensityIndependentSphOn, DensityContrastLimit, ArtBulkViscConst};
    ck(mpg_set_hydropar(mpg_shim_engine(), &hp));
}

/* kick_factor_data (density.c:115-132), the drift factors of hydra.c:178-186, dloga of the kick and of the bin */
static void fill_times(mpg_sph_times *t, const DriftKickTimes *times, Cosmology *CP, double atime)
{
    struct kick_factor_data kf;
    int b;
    init_kick_factor_data(&kf, times, CP);
    memset(t, 0, sizeof(*t));
    t->FgravkickB = kf.FgravkickB;
    for(b = 0; b <= TIMEBINS; b++) {
        t->gravkicks[b] = kf.gravkicks[b];
        t->hydrokicks[b] = kf.hydrokicks[b];
        t->drifts[b] = is_timebin_active(b, times->Ti_Current) ? 0 : get_exact_drift_factor(CP, times->Ti_lastactivedrift[b], times->Ti_Current);
        t->dloga_kick[b] = dloga_from_dti(times->Ti_Current - times->Ti_kick[b], times->Ti_Current);
        t->dloga_bin[b] = get_dloga_for_bin(b, times->Ti_Current);
    }
    t->atime = atime;
    t->hubble = atime > 0 ? hubble_function(CP, atime) : 0;
}

/* arrays in particle order, from the arena (freed in reverse order: utils/memory.c enforces LIFO) */
struct sph_host {
    mpg_sph_arrays A;
    double *block;
    uint8_t *tb;
};

static void gather(struct sph_host *H)
{
    const int64_t n = PartManager->NumPart;
    int64_t i;
    H->block = (double *)mymalloc("mpg_sph", (size_t)n * 31 * sizeof(double));
    H->tb = (uint8_t *)mymalloc("mpg_sph_tb", (size_t)n * 2);
    double *q = H->block;
#define COL(w) (q += (size_t)n * (w), q - (size_t)n * (w))
    double *hsml = COL(1), *dthsml = COL(1), *vel = COL(3), *gacc = COL(3), *gpm = COL(3), *hin = COL(3), *ent = COL(1), *dte = COL(1);
    memset(&H->A, 0, sizeof(H->A));
    H->A.density = COL(1);
    H->A.egywtdensity = COL(1);
    H->A.dhsmlegyfac = COL(1);
    H->A.divvel = COL(1);
    H->A.curlvel = COL(1);
    H->A.gradrho = COL(3);
    H->A.hydroacc_out = COL(3);
    H->A.dtentropy_out = COL(1);
    H->A.maxsignalvel = COL(1);
#undef COL
    #pragma omp parallel for
    for(i = 0; i < n; i++) {
        int k;
        hsml[i] = P[i].Hsml;
        for(k = 0; k < 3; k++) {
            vel[3 * i + k] = P[i].Vel[k];
            gacc[3 * i + k] = P[i].FullTreeGravAccel[k];
            gpm[3 * i + k] = P[i].GravPM[k];
            hin[3 * i + k] = P[i].Type == 0 ? SPHP(i).HydroAccel[k] : 0;
        }
        ent[i] = P[i].Type == 0 ? SPHP(i).Entropy : 0;
        dte[i] = P[i].Type == 0 ? SPHP(i).DtEntropy : 0;
        H->tb[i] = P[i].TimeBinHydro;
        H->tb[n + i] = P[i].TimeBinGravity;
        if(P[i].Type == 0) { /* the hydro force reads the density fields the density loop left in the slots */
            H->A.density[i] = SPHP(i).Density;
            H->A.egywtdensity[i] = SPHP(i).EgyWtDensity;
            H->A.dhsmlegyfac[i] = SPHP(i).DhsmlEgyDensityFactor;
            H->A.divvel[i] = SPHP(i).DivVel;
            H->A.curlvel[i] = SPHP(i).CurlVel;
        }
    }
    H->A.hsml = hsml;
    H->A.dthsml = dthsml;
    H->A.vel = vel;
    H->A.gacc = gacc;
    H->A.gpm = gpm;
    H->A.hydroacc_in = hin;
    H->A.entropy = ent;
    H->A.dtentropy_in = dte;
    H->A.tb_hydro = H->tb;
    H->A.tb_grav = H->tb + n;
}

static void release(struct sph_host *H)
{
    myfree(H->tb);
    myfree(H->block);
}

#define view mpg_shim_view

void set_init_hsml(ForceTree *tree, DomainDecomp *ddecomp, const double MeanGasSeparation)
{
    struct sph_host H;
    int64_t i;
    mpg_shim_set_domain(ddecomp);
    /* init.c:setup_smoothinglengths calls this once at start-up, before any force: a fresh epoch.  The estimate is local arithmetic
     * on the node sizes of the rank's own gas tree (density.c:57-73: no neighbour search), so one rank's form serves any NTask */
    mpg_shim_particles_changed();
    mpg_shim_sync(-1, -1, tree->BoxSize, 0);
    mpg_particle_view v = view();
    gather(&H);
    ck(mpg_set_init_hsml(mpg_shim_engine(), &v, tree->BoxSize, &H.A, MeanGasSeparation));
    #pragma omp parallel for
    for(i = 0; i < PartManager->NumPart; i++)
        if(P[i].Type == 0 || P[i].Type == 5)
            P[i].Hsml = H.A.hsml[i];
    release(&H);
    mpg_shim_particles_changed(); /* (the engine now holds this table under an epoch that gravity must not reuse blindly: Hsml changed) */
}

void density(const ActiveParticles *act, int update_hsml, int DoEgyDensity, int BlackHoleOn, const DriftKickTimes times, Cosmology *CP,
             struct sph_pred_data *SPH_predicted, MyFloat *GradRho_mag, const ForceTree *const tree)
{
    (void)SPH_predicted; /* the engine keeps its own prediction cache (density.c:75-100) */
    struct sph_host H;
    mpg_sph_times t;
    int64_t i;
    walltime_measure("/Misc");
    /* density() is the FIRST force call of a step (run.c:472, and of the start-up: init.c): the table may just have been drifted
     * and exchanged, the decomposition rewritten */
    if(mpg_shim_resident()) {
        /* a resident stretch (timestep-hip.c): the slot fields were gathered once and live in HBM; the loop runs on them in place and
         * its results stay there (mpg_shim_resident_end scatters them).  |grad rho| is a host output of this call: not in this mode */
        if(GradRho_mag)
            endrun(5, "density(): GradRho_mag is not available inside a resident stretch (mpg_shim_resident_end first)\n");
        mpg_particle_view rv = view();
        fill_times(&t, &times, CP, 0);
        ck(mpg_density(mpg_shim_engine(), &rv, tree->BoxSize, mpg_shim_resident_sph(), &t, act->ActiveParticle, act->NumActiveParticle, update_hsml,
                       DoEgyDensity, BlackHoleOn));
        walltime_add("/SPH/Density/WalkPrim", walltime_measure(WALLTIME_IGNORE));
        return;
    }
    mpg_shim_sync(times.Ti_Current, 0, tree->BoxSize, mpg_shim_dist() ? 1.26 * max_target_hsml() : 0);
    mpg_particle_view v = view();
    gather(&H);
    fill_times(&t, &times, CP, 0);
    walltime_measure("/SPH/Density/Init");
    if(mpg_shim_dist()) {
        int attempt;
        ck(mpg_dist_set_sph_options(mpg_shim_dist(), BlackHoleOn));
        for(attempt = 0;; attempt++) {
            /* own particles + ghosts of this table: the local set the gas tree is built from (mpg_dist_force_tree_full once per epoch) */
            mpg_shim_dist_tree(&v);
            const int rc = mpg_dist_density(mpg_shim_dist(), &v, &H.A, &t, act->ActiveParticle, act->NumActiveParticle, update_hsml, DoEgyDensity);
            mpg_shim_dist_tree_replaced(); /* (the gas tree took the place of the gravity tree inside the library) */
            if(rc == 0)
                break;
            /* a smoothing length outgrew the ghost margin (collective: every rank sees the same all-reduced maximum): wider margin,
             * new ghosts, same inputs */
            const double hmax = mpg_dist_last_max_hsml(mpg_shim_dist());
            if(attempt >= 3 || !(hmax > mpg_shim_margin()))
                ck(rc); /* (another failure, or no convergence: endrun with the library's message) */
            mpg_shim_sync(times.Ti_Current, 0, tree->BoxSize, 1.26 * hmax);
        }
    }
    else
        ck(mpg_density(mpg_shim_engine(), &v, tree->BoxSize, &H.A, &t, act->ActiveParticle, act->NumActiveParticle, update_hsml,
                       DoEgyDensity, BlackHoleOn));
    #pragma omp parallel for
    for(i = 0; i < PartManager->NumPart; i++) {
        if(P[i].Type != 0 && P[i].Type != 5)
            continue;
        P[i].Hsml = H.A.hsml[i];
        if(P[i].Type == 0) {
            P[i].DtHsml = H.A.dthsml[i];
            SPHP(i).Density = H.A.density[i];
            SPHP(i).EgyWtDensity = H.A.egywtdensity[i];
            SPHP(i).DhsmlEgyDensityFactor = H.A.dhsmlegyfac[i];
            SPHP(i).DivVel = H.A.divvel[i];
            SPHP(i).CurlVel = H.A.curlvel[i];
            if(GradRho_mag) {
                const double *g = H.A.gradrho + 3 * i;
                GradRho_mag[P[i].PI] = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
            }
        }
        else if(BlackHoleOn) {
            BHP(i).Density = H.A.density[i];
            BHP(i).DivVel = H.A.divvel[i];
        }
    }
    release(&H);
    /* density.c:344-354: one device kernel stands where the reference has the top-tree, primary and secondary walks */
    int64_t st[4];
    ck(mpg_sph_get_stats(mpg_shim_engine(), st));
    const double timeall = walltime_measure(WALLTIME_IGNORE);
    walltime_add("/SPH/Density/WalkTop", 0);
    walltime_add("/SPH/Density/WalkPrim", timeall);
    walltime_add("/SPH/Density/WalkSec", 0);
    walltime_add("/SPH/Density/PostPre", 0);
    walltime_add("/SPH/Density/Wait", 0);
    walltime_add("/SPH/Density/Reduce", 0);
    walltime_add("/SPH/Density/Misc", 0);
}

void hydro_force(const ActiveParticles *act, const double atime, struct sph_pred_data *SPH_predicted, const DriftKickTimes times,
                 Cosmology *CP, const ForceTree *const tree)
{
    (void)SPH_predicted;
    struct sph_host H;
    mpg_sph_times t;
    int64_t i;
    walltime_measure("/Misc");
    /* hydro_force() follows density() of the same step on the same table (run.c:472-489): same epoch, and for several ranks the
     * local set, gas tree and ghost columns the density loop left in the library (mpg_dist_hydro_force checks that it is so) */
    if(mpg_shim_resident()) { /* (see density()) */
        mpg_particle_view rv = view();
        fill_times(&t, &times, CP, atime);
        ck(mpg_hydro_force(mpg_shim_engine(), &rv, mpg_shim_resident_sph(), &t, act->ActiveParticle, act->NumActiveParticle));
        walltime_add("/SPH/Hydro/WalkPrim", walltime_measure(WALLTIME_IGNORE));
        return;
    }
    mpg_shim_sync(times.Ti_Current, 0, tree->BoxSize, 0);
    mpg_particle_view v = view();
    gather(&H);
    fill_times(&t, &times, CP, atime);
    walltime_measure("/SPH/Hydro/Init");
    if(mpg_shim_dist()) {
        ck(mpg_dist_hydro_force(mpg_shim_dist(), &v, &H.A, &t, act->ActiveParticle, act->NumActiveParticle));
    }
    else
        ck(mpg_hydro_force(mpg_shim_engine(), &v, &H.A, &t, act->ActiveParticle, act->NumActiveParticle));
    #pragma omp parallel for
    for(i = 0; i < PartManager->NumPart; i++) {
        int k;
        if(P[i].Type != 0 || !is_timebin_active(P[i].TimeBinHydro, times.Ti_Current))
            continue;
        for(k = 0; k < 3; k++)
            SPHP(i).HydroAccel[k] = H.A.hydroacc_out[3 * i + k];
        SPHP(i).DtEntropy = H.A.dtentropy_out[i];
        SPHP(i).MaxSignalVel = H.A.maxsignalvel[i];
    }
    release(&H);
    const double timeall = walltime_measure(WALLTIME_IGNORE);
    walltime_add("/SPH/Hydro/WalkTop", 0);
    walltime_add("/SPH/Hydro/WalkPrim", timeall);
    walltime_add("/SPH/Hydro/WalkSec", 0);
    walltime_add("/SPH/Hydro/PostPre", 0);
    walltime_add("/SPH/Hydro/Wait", 0);
    walltime_add("/SPH/Hydro/Reduce", 0);
    walltime_add("/SPH/Hydro/Misc", 0);
}
