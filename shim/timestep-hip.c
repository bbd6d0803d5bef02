/* libgadget/timestep-hip.c -- a run whose particle table STAYS IN HBM between two domain decompositions (resident mode).
 *
 * With gravity-hip.c / sph-hip.c alone every force call of a step moves Pos / Mass up and its results down, because drift_all_particles,
 * apply_half_kick and find_timesteps run on the host in between (run.c:392-794): half of a 256^3 step (bench.py host_path).  This file
 * takes the integrator over as well.  The maintainer brackets the stretch of run.c in which nothing reorders P[] -
 *
 *     domain_decompose_full(...) / domain_maintain(...)          run.c:422-434   (host; reorders and exchanges P[])
 *     mpg_shim_resident_begin(PartManager->BoxSize);             <- one upload: P[] columns + the SPH slot fields in particle order
 *     ... density, hydro_force, gravpm_force, force_tree_full, grav_short_tree, find_hydro_timesteps, apply_half_kick,
 *         apply_PM_half_kick of this step, drift_all_particles at the top of the next ...      (all on the device copies)
 *     mpg_shim_resident_end();                                   <- one fetch, before the next domain_maintain, a snapshot, FOF, or any
 *                                                                   host module that reads P[] / SphP[] (cooling, star formation)
 *
 * - and compiles timestep.c and drift.c with the entry points below renamed (one line each in libgadget/Makefile, as for forcetree.o):
 *     timestep.o: CFLAGS += -Dapply_half_kick=cpu_apply_half_kick -Dapply_PM_half_kick=cpu_apply_PM_half_kick -Dfind_hydro_timesteps=cpu_find_hydro_timesteps
 *                           -Dfind_timesteps=cpu_find_timesteps -Dapply_hydro_half_kick=cpu_apply_hydro_half_kick
 *                           -Dhierarchical_gravity_and_timesteps=cpu_hierarchical_gravity_and_timesteps
 *                           -Dhierarchical_gravity_accelerations=cpu_hierarchical_gravity_accelerations
 *     drift.o:    CFLAGS += -Ddrift_all_particles=cpu_drift_all_particles
 * These are ALL the functions of run.c's step that read or write the integrated columns (run.c:420, 498-499, 536-565, 754-794): a function of
 * that set that stayed on the host inside a resident stretch would integrate the stale host copies and its work would be discarded by the
 * fetch at the end.  The resident stretch carries the branch WITHOUT SplitGravityTimestepsOn (run.c:553-565, 754-759:
 * find_timesteps + apply_half_kick); with All.HierarchicalGravity the three functions of that branch stop the run with a message instead
 * of kicking host copies (its level loop has a device form of its own, mpg_dev_hierarchical_*, which takes device arrays the resident
 * table does not hand out yet).
 * Outside a resident stretch the definitions here call those cpu_ originals, so a run that never calls mpg_shim_resident_begin behaves as
 * before.  Inside one, the factors come from the reference's own get_exact_*_factor / dloga_from_dti (host arithmetic on the integer
 * timeline) and the per-particle loops run as mpg_resident_* on the device (include/mpgadget_hip.h).  One rank per GPU; NTask > 1 keeps the
 * host path (the resident calls are one-rank forms: mpg_dist_* owns the multi-rank choreography).
 * Not covered (the shim stops with a message): black-hole particles inside a resident stretch (drift.c:33-55 repositioning, the
 * dynamic-friction kicks, timestep.c:1003-1010), hierarchical gravity (apply_hydro_half_kick, hierarchical_gravity_and_timesteps and
 * hierarchical_gravity_accelerations below call endrun inside a resident stretch), ForceEqualTimesteps. */
#include <mpi.h>
#include <math.h>
#include <string.h>
#include "timestep.h"
#include "drift.h"
#include "timefac.h"
#include "timebinmgr.h"
#include "cosmology.h"
#include "partmanager.h"
#include "slotsmanager.h"
#include "walltime.h"
#include "utils/endrun.h"
#include "utils/mymalloc.h"
#include <mpgadget_hip.h>
#include "mpg_shim.h"

#define ck mpg_shim_ck
#define view mpg_shim_view

/* the renamed originals (timestep.c:873-929, 964-985, 617-733; drift.c:84-102) */
void cpu_apply_half_kick(const ActiveParticles *act, Cosmology *CP, DriftKickTimes *times, const double atime);
void cpu_apply_PM_half_kick(Cosmology *CP, DriftKickTimes *times);
int cpu_find_hydro_timesteps(const ActiveParticles *act, DriftKickTimes *times, const double atime, const Cosmology *CP, const int isFirstTimeStep);
void cpu_drift_all_particles(inttime_t ti0, inttime_t ti1, Cosmology *CP, const double random_shift[3]);
/* (timestep.c:739-849, 930-968, 296-490, 502-599) */
int cpu_find_timesteps(const ActiveParticles *act, DriftKickTimes *times, const double atime, int FastParticleType, const Cosmology *CP, const double asmth,
                       const int isFirstTimeStep);
void cpu_apply_hydro_half_kick(const ActiveParticles *act, Cosmology *CP, DriftKickTimes *times, const double atime);
int cpu_hierarchical_gravity_and_timesteps(const ActiveParticles *act, PetaPM *pm, DomainDecomp *ddecomp, struct grav_accel_store StoredGravAccel,
                                           DriftKickTimes *times, const double atime, int HybridNuGrav, int FastParticleType, Cosmology *CP,
                                           const char *EmergencyOutputDir);
int cpu_hierarchical_gravity_accelerations(const ActiveParticles *act, PetaPM *pm, DomainDecomp *ddecomp, struct grav_accel_store StoredGravAccel,
                                           DriftKickTimes *times, int HybridNuGrav, Cosmology *CP, const char *EmergencyOutputDir);

/* ---- the resident stretch ---- */
static struct {
    int on;
    double BoxSize;
    mpg_sph_arrays A;  /* the SPH slot fields in particle order (host; the library keeps the device copies) */
    double *block;
    uint8_t *tb;
} R;

int mpg_shim_resident(void) { return R.on; }
const mpg_sph_arrays *mpg_shim_resident_sph(void) { return R.on ? &R.A : NULL; }

void mpg_shim_resident_begin(double BoxSize)
{
    const int64_t n = PartManager->NumPart;
    int64_t i;
    if(R.on)
        endrun(5, "mpg_shim_resident_begin: already resident\n");
    if(mpg_shim_ntask() > 1) /* (the host path stays in force) */
        return;
    for(i = 0; i < n; i++)
        if(P[i].Type == 5 && !P[i].IsGarbage && !P[i].Swallowed)
            endrun(5, "mpg_shim_resident_begin: black holes are not carried by the resident integrator (drift.c:33-55, timestep.c:1003-1010)\n");
    mpg_particle_view v = view();
    ck(mpg_resident_begin(mpg_shim_engine(), &v, BoxSize));
    /* the slot fields, gathered once: what sph-hip.c gathers per call */
    R.block = (double *)mymalloc("mpg_resident_sph", (size_t)n * 31 * sizeof(double));
    R.tb = (uint8_t *)mymalloc("mpg_resident_tb", (size_t)n * 2);
    double *q = R.block;
#define COL(w) (q += (size_t)n * (w), q - (size_t)n * (w))
    memset(&R.A, 0, sizeof(R.A));
    double *hsml = COL(1), *dthsml = COL(1), *vel = COL(3), *gacc = COL(3), *gpm = COL(3), *hin = COL(3), *ent = COL(1), *dte = COL(1);
    R.A.density = COL(1);
    R.A.egywtdensity = COL(1);
    R.A.dhsmlegyfac = COL(1);
    R.A.divvel = COL(1);
    R.A.curlvel = COL(1);
    R.A.gradrho = COL(3);
    R.A.hydroacc_out = COL(3);
    R.A.dtentropy_out = COL(1);
    R.A.maxsignalvel = COL(1);
#undef COL
    memset(R.block, 0, (size_t)n * 31 * sizeof(double));
    #pragma omp parallel for
    for(i = 0; i < n; i++) {
        int k;
        hsml[i] = P[i].Hsml;
        dthsml[i] = P[i].DtHsml;
        R.tb[i] = P[i].TimeBinHydro;
        R.tb[n + i] = P[i].TimeBinGravity;
        if(P[i].Type != 0)
            continue;
        ent[i] = SPHP(i).Entropy;
        R.A.density[i] = SPHP(i).Density;
        R.A.egywtdensity[i] = SPHP(i).EgyWtDensity;
        R.A.dhsmlegyfac[i] = SPHP(i).DhsmlEgyDensityFactor;
        R.A.divvel[i] = SPHP(i).DivVel;
        R.A.curlvel[i] = SPHP(i).CurlVel;
        R.A.dtentropy_out[i] = SPHP(i).DtEntropy;
        R.A.maxsignalvel[i] = SPHP(i).MaxSignalVel;
        for(k = 0; k < 3; k++)
            R.A.hydroacc_out[3 * i + k] = SPHP(i).HydroAccel[k];
    }
    R.A.hsml = hsml;
    R.A.dthsml = dthsml;
    R.A.vel = vel;              /* (aliases of the resident table's columns on the device: never read on the host) */
    R.A.gacc = gacc;
    R.A.gpm = gpm;
    R.A.hydroacc_in = hin;      /* (device: aliases hydroacc_out / dtentropy_out) */
    R.A.dtentropy_in = dte;
    R.A.entropy = ent;
    R.A.tb_hydro = R.tb;
    R.A.tb_grav = R.tb + n;
    ck(mpg_resident_sph_begin(mpg_shim_engine(), &v, &R.A));
    R.BoxSize = BoxSize;
    R.on = 1;
}

void mpg_shim_resident_end(void)
{
    const int64_t n = PartManager->NumPart;
    int64_t i;
    if(!R.on)
        return;
    mpg_particle_view v = view();
    ck(mpg_resident_sph_end(mpg_shim_engine(), &R.A));
    ck(mpg_resident_end(mpg_shim_engine(), &v)); /* Pos, Vel, FullTreeGravAccel, GravPM, Potential back into P[] */
    #pragma omp parallel for
    for(i = 0; i < n; i++) {
        int k;
        P[i].TimeBinHydro = R.tb[i];       /* (every type: find_timesteps sets both bins of every active particle) */
        P[i].TimeBinGravity = R.tb[n + i];
        if(P[i].Type != 0)
            continue;
        P[i].Hsml = R.A.hsml[i];
        P[i].DtHsml = R.A.dthsml[i];
        SPHP(i).Entropy = R.A.entropy[i];
        SPHP(i).Density = R.A.density[i];
        SPHP(i).EgyWtDensity = R.A.egywtdensity[i];
        SPHP(i).DhsmlEgyDensityFactor = R.A.dhsmlegyfac[i];
        SPHP(i).DivVel = R.A.divvel[i];
        SPHP(i).CurlVel = R.A.curlvel[i];
        SPHP(i).DtEntropy = R.A.dtentropy_out[i];
        SPHP(i).MaxSignalVel = R.A.maxsignalvel[i];
        for(k = 0; k < 3; k++)
            SPHP(i).HydroAccel[k] = R.A.hydroacc_out[3 * i + k];
    }
    myfree(R.tb);
    myfree(R.block);
    R.on = 0;
    mpg_shim_particles_changed(); /* (the host table is current again and may now be reordered) */
}

/* the device's new time bins into P[]: build_active_particles (timestep.c:1333-1420, run.c:436) and update_kick_times read them on the
 * host at the top of the next step.  2 bytes per particle; everything else of P[] stays stale until mpg_shim_resident_end. */
static void fetch_timebins(void)
{
    const int64_t n = PartManager->NumPart;
    int64_t i;
    ck(mpg_resident_fetch_timebins(mpg_shim_engine(), R.tb, R.tb + n));
    #pragma omp parallel for
    for(i = 0; i < n; i++) {
        P[i].TimeBinHydro = R.tb[i];
        P[i].TimeBinGravity = R.tb[n + i];
    }
}

/* ---- the reference's entry points ---- */

/* apply_half_kick, timestep.c:873-929: the kick factors per bin on the host as there, the particle loop on the device */
void apply_half_kick(const ActiveParticles *act, Cosmology *CP, DriftKickTimes *times, const double atime)
{
    if(!R.on) {
        cpu_apply_half_kick(act, CP, times, atime);
        return;
    }
    int bin;
    mpg_kick_factors K;
    walltime_measure("/Misc");
    memset(&K, 0, sizeof(K));
    for(bin = 0; bin <= TIMEBINS; bin++) {
        K.bin_active[bin] = (unsigned char)is_timebin_active(bin, times->Ti_Current);
        K.dt_entr[bin] = dloga_from_dti(dti_from_timebin(bin) / 2, times->Ti_Current); /* timestep.c:915-917 */
        if(bin < times->mintimebin || !K.bin_active[bin])
            continue;
        const inttime_t newkick = times->Ti_kick[bin] + dti_from_timebin(bin) / 2;
        K.gravkick[bin] = get_exact_gravkick_factor(CP, times->Ti_kick[bin], newkick);
        K.hydrokick[bin] = get_exact_hydrokick_factor(CP, times->Ti_kick[bin], newkick);
    }
    K.atime = atime;
    K.MaxGasVel = mpg_shim_max_gas_vel(); /* TimestepParams.MaxGasVel is static in timestep.c: one accessor added there */
    mpg_particle_view v = view();
    ck(mpg_resident_apply_half_kick(mpg_shim_engine(), &v, act->ActiveParticle, act->NumActiveParticle, &K));
    walltime_measure("/Timeline/HalfKick/Short");
}

/* apply_PM_half_kick, timestep.c:964-985 */
void apply_PM_half_kick(Cosmology *CP, DriftKickTimes *times)
{
    if(!R.on) {
        cpu_apply_PM_half_kick(CP, times);
        return;
    }
    const inttime_t tistart = times->PM_kick;
    const inttime_t tiend = tistart + times->PM_length / 2;
    const double Fgravkick = get_exact_gravkick_factor(CP, tistart, tiend);
    mpg_particle_view v = view();
    ck(mpg_resident_apply_pm_half_kick(mpg_shim_engine(), &v, Fgravkick));
    times->PM_kick = tiend;
    walltime_measure("/Timeline/HalfKick/Long");
}

/* drift_all_particles, drift.c:84-102 */
void drift_all_particles(inttime_t ti0, inttime_t ti1, Cosmology *CP, const double random_shift[3])
{
    if(!R.on) {
        cpu_drift_all_particles(ti0, ti1, CP, random_shift);
        mpg_shim_prefetch(ti1, PartManager->BoxSize); /* (round 6: the step's upload starts here instead of inside gravpm_force) */
        return;
    }
    if(ti1 < ti0)
        endrun(12, "Trying to reverse time: ti0=%ld ti1=%ld\n", ti0, ti1);
    const double ddrift = get_exact_drift_factor(CP, ti0, ti1);
    mpg_particle_view v = view();
    ck(mpg_resident_drift_all_particles(mpg_shim_engine(), &v, ddrift, random_shift));
    int64_t i;
    #pragma omp parallel for
    for(i = 0; i < PartManager->NumPart; i++)
        PartManager->Base[i].Ti_drift = ti1; /* (the host's record of where the particles are in time stays current) */
    walltime_measure("/Drift");
}

/* find_hydro_timesteps, timestep.c:617-733: the particle loop and the update of times->mintimebin on the device copies */
int find_hydro_timesteps(const ActiveParticles *act, DriftKickTimes *times, const double atime, const Cosmology *CP, const int isFirstTimeStep)
{
    if(!R.on)
        return cpu_find_hydro_timesteps(act, times, atime, CP, isFirstTimeStep);
    _Static_assert(TIMEBINS == MPG_TIMEBINS, "TIMEBINS (timebinmgr.h:8)");
    _Static_assert(sizeof(DriftKickTimes) == sizeof(mpg_drift_kick_times), "DriftKickTimes (timestep.h:10-27)");
    const double hubble = hubble_function(CP, atime);
    mpg_timeline tl;
    mpg_timestep_params par;
    mpg_hydrostep_result res;
    mpg_shim_timeline(&tl);                                 /* SyncPoints[].loga: static in timebinmgr.c, one accessor added there */
    par.ErrTolIntAccuracy = 0;                              /* (not read by the hydro criterion) */
    par.MinSizeTimestep = mpg_shim_min_size_timestep();     /* TimestepParams.MinSizeTimestep, the same accessor file */
    mpg_particle_view v = view();
    ck(mpg_resident_find_hydro_timesteps(mpg_shim_engine(), &v, act->ActiveParticle, act->NumActiveParticle, (mpg_drift_kick_times *)times, &tl, &par,
                                         mpg_shim_courant_fac(), atime, hubble, isFirstTimeStep, &res));
    fetch_timebins();
    message(0, "Hydro timesteps: Accel: %ld Soundspeed: %ld DivVel: %ld Accrete: %ld Neighbour: %ld\n", (long)res.ntitype[0], (long)res.ntitype[1],
            (long)res.ntitype[4], (long)res.ntitype[2], (long)res.ntitype[3]);
    walltime_measure("/Timeline/Hydro");
    message(0, "Min grav timebin: %d mintimebin %d\n", times->mingravtimebin, times->mintimebin);
    return (int)res.badstepsizecount;
}

/* find_timesteps, timestep.c:739-849 (run.c:756, the branch without SplitGravityTimestepsOn): the particle loop, the PM step's shrink and
 * times->mintimebin / maxtimebin on the device copies.  On a PM step the new PM length needs the rms velocities (get_PM_timestep_ti ->
 * get_long_range_timestep_dloga, timestep.c:1201-1300, static there and a loop over P[].Vel): the velocities are fetched for it - a PM step
 * is one step in many - and the reference's own function computes the length. */
int find_timesteps(const ActiveParticles *act, DriftKickTimes *times, const double atime, int FastParticleType, const Cosmology *CP, const double asmth,
                   const int isFirstTimeStep)
{
    if(!R.on)
        return cpu_find_timesteps(act, times, atime, FastParticleType, CP, asmth, isFirstTimeStep);
    if(mpg_shim_force_equal_timesteps())
        endrun(5, "find_timesteps: ForceEqualTimesteps is not carried inside a resident stretch (timestep.c:759-761)\n");
    walltime_measure("/Misc");
    mpg_particle_view v = view();
    inttime_t dti_max_pm = 0;
    if(is_PM_timestep(times)) {
        ck(mpg_resident_fetch(mpg_shim_engine(), &v, MPG_FIELD_VEL)); /* P[].Vel current for the rms velocities */
        dti_max_pm = mpg_shim_get_PM_timestep_ti(times, atime, CP, FastParticleType, asmth);
    }
    const double hubble = hubble_function(CP, atime);
    mpg_timeline tl;
    mpg_timestep_params par;
    mpg_timestep_result res;
    mpg_shim_timeline(&tl);
    par.ErrTolIntAccuracy = mpg_shim_err_tol_int_accuracy();
    par.MinSizeTimestep = mpg_shim_min_size_timestep();
    ck(mpg_resident_find_timesteps(mpg_shim_engine(), &v, act->ActiveParticle, act->NumActiveParticle, (mpg_drift_kick_times *)times, &tl, &par,
                                   mpg_shim_courant_fac(), atime, hubble, dti_max_pm, &res));
    fetch_timebins();
    message(0, "PM timebin: %lx (dloga: %g). Criteria: Accel: %ld Soundspeed: %ld DivVel: %ld Accrete: %ld Neighbour: %ld\n", times->PM_length,
            dloga_from_dti(times->PM_length, times->Ti_Current), (long)res.ntitype[0], (long)res.ntitype[1], (long)res.ntitype[4], (long)res.ntitype[2],
            (long)res.ntitype[3]);
    /* (set_bh_first_timestep, timestep.c:844-845: no black holes inside a resident stretch - mpg_shim_resident_begin refuses them) */
    walltime_measure("/Timeline");
    return (int)res.badstepsizecount;
}

/* The branch of run.c with All.HierarchicalGravity (run.c:498-499, 536-541, 766-775).  Outside a resident stretch: the originals.  Inside
 * one they must not run - they would kick and re-bin the stale host copies, and mpg_shim_resident_end would then overwrite their work. */
static void no_hierarchical_gravity(const char *fn)
{
    endrun(5, "%s inside mpg_shim_resident_begin / _end: hierarchical gravity (SplitGravityTimestepsOn) is not carried by the resident "
              "integrator - set SplitGravityTimestepsOn = 0 or leave the resident stretch (mpg_shim_resident_end) first\n", fn);
}

void apply_hydro_half_kick(const ActiveParticles *act, Cosmology *CP, DriftKickTimes *times, const double atime)
{
    if(R.on)
        no_hierarchical_gravity("apply_hydro_half_kick");
    cpu_apply_hydro_half_kick(act, CP, times, atime);
}

int hierarchical_gravity_and_timesteps(const ActiveParticles *act, PetaPM *pm, DomainDecomp *ddecomp, struct grav_accel_store StoredGravAccel,
                                       DriftKickTimes *times, const double atime, int HybridNuGrav, int FastParticleType, Cosmology *CP,
                                       const char *EmergencyOutputDir)
{
    if(R.on)
        no_hierarchical_gravity("hierarchical_gravity_and_timesteps");
    return cpu_hierarchical_gravity_and_timesteps(act, pm, ddecomp, StoredGravAccel, times, atime, HybridNuGrav, FastParticleType, CP, EmergencyOutputDir);
}

int hierarchical_gravity_accelerations(const ActiveParticles *act, PetaPM *pm, DomainDecomp *ddecomp, struct grav_accel_store StoredGravAccel,
                                       DriftKickTimes *times, int HybridNuGrav, Cosmology *CP, const char *EmergencyOutputDir)
{
    if(R.on)
        no_hierarchical_gravity("hierarchical_gravity_accelerations");
    return cpu_hierarchical_gravity_accelerations(act, pm, ddecomp, StoredGravAccel, times, HybridNuGrav, CP, EmergencyOutputDir);
}
