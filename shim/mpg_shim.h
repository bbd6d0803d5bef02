/* libgadget/mpg_shim.h -- what gravity-hip.c and sph-hip.c share: the rank's engine, its multi-rank state, the error check, and the
 * bookkeeping that tells the library WHEN the rank's particle table or the domain decomposition changed.
 *
 * The reference calls its force modules in a fixed order per step (run.c:392-548): drift + domain_decompose_full / domain_maintain,
 * then density() and hydro_force() (run.c:472,489), then gravpm_force() on PM steps (run.c:522) and force_tree_full() +
 * grav_short_tree() (run.c:546-547); at start-up set_init_hsml() and density() run before any gravity (init.c).  None of these
 * calls says "P[] moved since you last saw it", and ddecomp is rewritten IN PLACE by every domain_decompose_full (run.c:422,434),
 * usually with the same NTopLeaves.  So every entry point of the shim starts with mpg_shim_sync():
 *   - the particle table is identified by (Ti_Current, &P[0], NumPart): a new key is a new epoch for the engine's upload cache
 *     (mpg_set_particle_epoch) and invalidates the local tree / ghost plan of the multi-rank path;
 *   - the decomposition is identified by a hash over TopNodes (StartKey, Shift, Daughter, Leaf) and TopLeaves[].Task, recomputed on
 *     every call (a few thousand entries): a new hash, or a margin that grew, goes to mpg_dist_set_domain.
 * A maintainer who prefers explicit hooks calls mpg_shim_particles_changed() after anything else that moves or reorders P[]
 * within one Ti_Current (fof_fof's exchange, a restart) and mpg_shim_set_domain() once with the run's DomainDecomp. */
#ifndef MPG_SHIM_H
#define MPG_SHIM_H
#include "domain.h"
#include "timebinmgr.h"
#include "forcetree.h"
#include <mpgadget_hip.h>

mpg_engine *mpg_shim_engine(void);  /* the rank's engine (created on first use; one rank = one GPU) */
mpg_dist *mpg_shim_dist(void);      /* its multi-rank state, NULL with one rank */
int mpg_shim_ntask(void);
void mpg_shim_ck(int rc);           /* endrun(5, mpg_last_error()) on a non-zero return code */

/* the DomainDecomp of this run (run.c keeps one object for the whole run); gravpm_force() and set_init_hsml() also pass it */
void mpg_shim_set_domain(DomainDecomp *ddecomp);
/* P[] was moved / reordered / resized by something the shim cannot see */
void mpg_shim_particles_changed(void);
/* P[] is final for the step (the end of drift_all_particles): start the epoch's packing pass + uploads on a host thread (one rank, host path) */
void mpg_shim_prefetch(inttime_t Ti_Current, double BoxSize);
/* Start of every entry point.  Ti_Current < 0: not known to the caller (gravpm_force: matched through Time = get_atime(Ti)).
 * margin_want: the interaction range this call needs covered by ghosts (Rcut in length units; the largest smoothing length for the
 * SPH loops); the domain is (re-)set when the decomposition changed or the margin in force is smaller. */
void mpg_shim_sync(inttime_t Ti_Current, double Time, double BoxSize, double margin_want);
/* NTask > 1: the local tree + ghost plan of the current table (mpg_dist_force_tree_full once per epoch) */
void mpg_shim_dist_tree(const mpg_particle_view *v);
/* ... and its invalidation by the SPH loops / FOF, which replace the gravity tree inside the library */
void mpg_shim_dist_tree_replaced(void);
/* ---- trees that exist only on the device (forcetree-hip.c) ----
 * The constructors of forcetree.h record what was asked for; the consumers in gravity-hip.c / sph-hip.c build the device tree from the
 * record.  mpg_shim_deferred_tree: the record of a tree that has no host nodes (NULL: an ordinary host tree).  mpg_shim_host_tree: build
 * the host tree now (before the first host module that receives it allocates anything); mpg_shim_require_host_tree: the guard at the top
 * of treewalk_run. */
enum { MPG_TREE_FULL = 1, MPG_TREE_ACTIVE = 2, MPG_TREE_MASK = 3 };
struct mpg_deferred_tree {
    const ForceTree *tree;       /* the caller's object (key) */
    DomainDecomp *ddecomp;
    int kind, mask, HybridNuTracer, alloc_father, moments_wanted, materialised;
    const int *ActiveParticle;   /* MPG_TREE_ACTIVE: the list the tree is built from (NULL: all) */
    int64_t NumActiveParticle;
    const char *EmergencyOutputDir;
    struct NODE root;            /* what tree->Nodes[tree->firstnode] reads while the tree is deferred */
};
const struct mpg_deferred_tree *mpg_shim_deferred_tree(const ForceTree *tree);
void mpg_shim_host_tree(ForceTree *tree);
void mpg_shim_require_host_tree(const ForceTree *tree, const char *walk);
double mpg_shim_margin(void); /* the ghost margin in force (0: no domain handed over yet) */
mpg_particle_view mpg_shim_view(void);
/* ---- a run whose table stays in HBM between two domain decompositions (timestep-hip.c) ----
 * mpg_shim_resident_begin after the step's domain_decompose_full / domain_maintain, mpg_shim_resident_end before the next one and before
 * any host module that reads P[] / SphP[]; in between density(), hydro_force(), gravpm_force(), force_tree_full(), grav_short_tree(),
 * find_timesteps(), find_hydro_timesteps(), apply_half_kick(), apply_PM_half_kick() and drift_all_particles() run on the device copies;
 * the hierarchical-gravity functions (apply_hydro_half_kick, hierarchical_gravity_*) stop the run there. */
void mpg_shim_resident_begin(double BoxSize);
void mpg_shim_resident_end(void);
int mpg_shim_resident(void);
const mpg_sph_arrays *mpg_shim_resident_sph(void); /* the host set of the resident SPH arrays (sph-hip.c passes it instead of gathering) */
/* the accessors the maintainer adds next to the file-static parameters they read (two lines each):
 *   timestep.c:   double mpg_shim_max_gas_vel(void) { return TimestepParams.MaxGasVel; }            (timestep.c:40-60)
 *                 double mpg_shim_min_size_timestep(void) { return TimestepParams.MinSizeTimestep; }
 *                 double mpg_shim_courant_fac(void) { return TimestepParams.CourantFac; }
 *                 double mpg_shim_err_tol_int_accuracy(void) { return TimestepParams.ErrTolIntAccuracy; }
 *                 int mpg_shim_force_equal_timesteps(void) { return TimestepParams.ForceEqualTimesteps; }
 *                 inttime_t mpg_shim_get_PM_timestep_ti(const DriftKickTimes *times, double atime, const Cosmology *CP, int FastParticleType,
 *                     double asmth) { return get_PM_timestep_ti(times, atime, CP, FastParticleType, asmth); }      (static, timestep.c:1281-1300)
 *   timebinmgr.c: void mpg_shim_timeline(mpg_timeline *tl): tl->nsync = NSyncPoints and tl->loga = an array of SyncPoints[i].loga
 *                 (timebinmgr.c:18; kept alongside SyncPoints by setup_sync_points) */
double mpg_shim_max_gas_vel(void);
double mpg_shim_min_size_timestep(void);
double mpg_shim_courant_fac(void);
double mpg_shim_err_tol_int_accuracy(void);
int mpg_shim_force_equal_timesteps(void);
#include "timestep.h"
#include "cosmology.h"
inttime_t mpg_shim_get_PM_timestep_ti(const DriftKickTimes *times, double atime, const Cosmology *CP, int FastParticleType, double asmth);
void mpg_shim_timeline(mpg_timeline *tl);
#endif
