/* include/mpgadget_hip.h -- C-ABI of the MI355X (gfx950) TreePM + SPH force engine.
 *
 * The reference (MP-Gadget, plain C, statically linked) has no FFI: its "operator API" is the set of C
 * entry points run.c / timestep.c / init.c / runtests.c call.  Each function below replaces one of
 * those entry points (cited as libgadget/<file>:<line>) with plain pointers and sizes, so that a
 * 30-line shim compiled inside the reference tree can forward the reference symbols
 *   gravpm_init_periodic, gravpm_force, force_tree_full, force_tree_rebuild_mask, force_tree_free,
 *   grav_short_tree, gravshort_fill_ntab, gravshort_set_softenings, set_gravshort_treepar, FORCE_SOFTENING,
 *   density, hydro_force
 * to this library.  No torch / C++ types appear in any signature.
 * Around that path: drift / kicks / gravity time bins and the hierarchical gravity level loop, Peano-Hilbert keys
 * and the (type, key) particle order, friends-of-friends groups, the matter power spectrum of the PM step, and the snapshot / IC
 * wire format - each section below cites what it replaces.
 *
 * Conventions
 *   - every call returns 0 on success, non-zero on failure; mpg_last_error() gives the message
 *     (the shim maps it to endrun(), utils/endrun.h:4-7 -- the reference has no return codes on this path);
 *   - "host" calls take the reference's AoS particle table through an mpg_particle_view (base pointer,
 *     stride and byte offsets: the same idea as PetaPMParticleStruct, petapm.h:86-99), copy what they
 *     need to HBM, run, and scatter results back into the caller's structs;
 *   - "dev" calls take device pointers (SoA, fp64) already resident in HBM and leave results in HBM;
 *     they are what bench.py times;
 *   - all arithmetic on the path is fp64 (MyFloat = double, types.h:13-17); P.Mass is float
 *     (partmanager.h:15) and the short-range window tables are float (gravity.c:20).
 */
#ifndef MPGADGET_HIP_H
#define MPGADGET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpg_engine mpg_engine;

/* ---- life cycle ------------------------------------------------------------------------------ */
/* Create an engine bound to HIP device `device` (one engine per GPU / MPI rank). */
int mpg_engine_create(mpg_engine **out, int device);
void mpg_engine_destroy(mpg_engine *eng);
/* Last error message of the calling thread ("" if none). */
const char *mpg_last_error(void);
/* Library / build identification string. */
const char *mpg_version(void);
/* Hash over the sources, headers and compiler flags this library was built from (build.py); the Python host side refuses a library
 * whose stamp differs from the sources next to it. */
const char *mpg_build_stamp(void);
/* Use an existing HIP stream (hipStream_t passed as void*) for all engine work; NULL = engine's own. */
int mpg_engine_set_stream(mpg_engine *eng, void *hip_stream);
void *mpg_engine_get_stream(mpg_engine *eng);
/* Block until all engine work is complete. */
int mpg_engine_synchronize(mpg_engine *eng);

/* ---- module parameters ----------------------------------------------------------------------- */
/* struct gravshort_tree_params, libgadget/gravity.h:9-22 (same fields, same order). */
typedef struct mpg_gravshort_tree_params {
    double ErrTolForceAcc;
    double BHOpeningAngle;
    double MaxBHOpeningAngle;
    int TreeUseBH;
    double Rcut;
    double FractionalGravitySoftening;
} mpg_gravshort_tree_params;

/* set_gravshort_treepar / get_gravshort_treepar, libgadget/gravshort-tree.c:53-61 */
int mpg_set_gravshort_treepar(mpg_engine *eng, const mpg_gravshort_tree_params *par);
int mpg_get_gravshort_treepar(mpg_engine *eng, mpg_gravshort_tree_params *par);
/* gravshort_set_softenings, libgadget/gravshort-tree.c:44-50 */
int mpg_gravshort_set_softenings(mpg_engine *eng, double MeanSeparation);
/* FORCE_SOFTENING, libgadget/gravshort-tree.c:37-41 (returns 2.8 * GravitySoftening) */
double mpg_force_softening(mpg_engine *eng);
/* gravshort_fill_ntab, libgadget/gravity.c:22-51.  window_type 0 = exact (needs Asmth == 1.5), 1 = erfc.
 * `table` = the calibrated 512 x 5 float64 table of libgadget/shortrange-kernel.c (row-major; carried as
 * data in mp-gadget_amd/data/shortrange_force_kernels.f64). */
int mpg_gravshort_fill_ntab(mpg_engine *eng, int window_type, double Asmth, const double *table, int nrows);
/* gravpm_init_periodic, libgadget/gravpm.c:51-54 (-> petapm_init, petapm.c:105-223): creates the mesh and FFT plans. */
int mpg_gravpm_init_periodic(mpg_engine *eng, double BoxSize, double Asmth, int Nmesh, double G);
/* petapm_destroy, libgadget/petapm.c:225-232 */
int mpg_petapm_destroy(mpg_engine *eng);
/* init_forcetree_params, libgadget/forcetree.c:39-44 (node pool = factor * NumPart; the engine grows it on demand) */
int mpg_init_forcetree_params(mpg_engine *eng, double TreeAllocFactor);

/* ---- particle table view (host AoS) ---------------------------------------------------------- */
/* Byte offsets into the caller's AoS record; -1 = field absent.  For struct particle_data
 * (libgadget/partmanager.h:9-71; 160 bytes) use mpg_particle_view_reference_layout(). */
typedef struct mpg_particle_view {
    void *base;        /* &P[0] */
    int64_t n;         /* PartManager->NumPart */
    int64_t stride;    /* sizeof(struct particle_data) */
    int32_t off_pos;   /* double[3]  Pos */
    int32_t off_mass;  /* float      Mass */
    int32_t off_flags; /* uint8 holding the bit-fields of partmanager.h:17-21: bit 0 IsGarbage, bit 1 Swallowed */
    int32_t off_type;  /* uint8 Type (partmanager.h:30) */
    int32_t off_accel; /* double[3]  FullTreeGravAccel */
    int32_t off_gravpm;    /* double[3]  GravPM */
    int32_t off_potential; /* double     Potential */
    int32_t off_hsml;      /* double     Hsml */
    int32_t off_vel;       /* double[3]  Vel */
    int32_t off_pi;        /* int32      PI (slot index) */
} mpg_particle_view;
/* Fill the offsets for the reference's struct particle_data (partmanager.h:9-71, offsets verified in SURVEY 8(a)). */
void mpg_particle_view_reference_layout(mpg_particle_view *v, void *particles, int64_t NumPart);

/* ---- host (drop-in) entry points ------------------------------------------------------------- */
/* Every host entry point that needs positions packs Pos / Mass / Type of P[] and uploads them, because P may have moved or
 * changed between calls.  A caller that knows better declares an EPOCH of its particle table: calls made with the same non-zero
 * epoch, the same &P[0], NumPart and BoxSize as the last upload reuse it (gravpm_force -> force_tree_full -> grav_short_tree of
 * one step, run.c:522-548).  Bump the epoch after anything that moves or reorders particles (drift, exchange, garbage collection);
 * 0 (the default) switches the reuse off. */
int mpg_set_particle_epoch(mpg_engine *eng, int64_t epoch);
/* Overlap of the host path's transfers with the device's work inside one epoch (default off).  When on - and only while a non-zero epoch is
 * declared - (a) the first call of an epoch packs Pos / Mass / Type, Potential and FullTreeGravAccel in ONE pass over the records, and
 * mpg_grav_short_tree takes OldAcc = |FullTreeGravAccel + GravPM| / G on the device from that upload and the device's own GravPM instead of
 * reading P[] again; (b) mpg_gravpm_force RETURNS once its results are on their way: GravPM and Potential are copied down and written into
 * P[] by a host thread while force_tree_full and grav_short_tree run, and are complete when the next call on the table that needs them
 * returns (mpg_grav_short_tree at the latest) or when mpg_host_results_sync returns.  A caller whose host code reads P[].GravPM /
 * P[].Potential between gravpm_force and grav_short_tree (energy_statistics, run.c:527) calls mpg_host_results_sync first.  (c) A walk of all
 * particles on a tree of all particles runs in slices of the tree order (5 from 2^20 particles on, each 0.55 of the one before it so that
 * the write-back nothing hides is the smallest; `on` = 2 .. 8 forces that many at any size), the results of a slice travelling down and into
 * P[] while the next is walked; results bit-identical to the unsliced walk. */
int mpg_set_host_overlap(mpg_engine *eng, int on);
/* The epoch's packing pass + uploads started early (round 6): call after mpg_set_particle_epoch as soon as P[] is final for the step - the
 * end of drift_all_particles (drift.c:84-102) - and the pass (Pos, Mass, Type, Potential, FullTreeGravAccel) runs on a host thread while the
 * caller goes on (run.c:420-522: domain_maintain, the active list); mpg_gravpm_force / mpg_density / ... of the same epoch and table join it
 * instead of packing.  A new epoch in between (an exchange, a garbage collection) simply leaves the upload unused.  Needs
 * mpg_set_host_overlap; a no-op without it or in resident mode.  P[] must not be written until the epoch's first entry point has returned,
 * and that entry point (or mpg_set_particle_epoch / mpg_host_results_sync, which wait for the pass) is the next call on this engine: the
 * pass uses the engine's stream. */
int mpg_host_prefetch(mpg_engine *eng, const mpg_particle_view *pv, double BoxSize);
int mpg_host_results_sync(mpg_engine *eng);
/* gravpm_force, libgadget/gravpm.c:61-119: zero GravPM, CIC deposit, r2c, Green's function, 4 x (transfer, c2r,
 * CIC readout).  Writes P[i].GravPM[3] (=) and P[i].Potential (+=, as readout_potential does). */
int mpg_gravpm_force(mpg_engine *eng, const mpg_particle_view *pv);
/* force_tree_full, libgadget/forcetree.c:110-128: tree of all particles with moments.  `mask` as ALLMASK etc.
 * (forcetree.h:22-27); particles whose type bit is clear are left out. */
int mpg_force_tree_full(mpg_engine *eng, const mpg_particle_view *pv, double BoxSize);
/* force_tree_rebuild_mask, libgadget/forcetree.c:151-166 */
int mpg_force_tree_rebuild_mask(mpg_engine *eng, const mpg_particle_view *pv, double BoxSize, int mask);
/* force_tree_free, libgadget/forcetree.c:1403-1413 */
int mpg_force_tree_free(mpg_engine *eng);
/* force_tree_active_moments (forcetree.c:129-148): a tree of the active particles only (ActiveParticle == NULL: all), with
 * moments; HybridNuTracer != 0 leaves neutrinos (type 2) out.  A walk over it takes the same active list as its targets. */
int mpg_force_tree_active_moments(mpg_engine *eng, const mpg_particle_view *pv, double BoxSize, const int *ActiveParticle,
                                  int64_t NumActiveParticle, int HybridNuTracer);
int mpg_dev_force_tree_active_moments(mpg_engine *eng, const int *d_active, int64_t nactive, int HybridNuTracer);
/* grav_short_tree, libgadget/gravshort-tree.c:96-154.  ActiveParticle == NULL means all particles
 * (timestep.c:77-84).  AccelStore may be NULL.  When the tree holds all particles (full_particle_tree_flag)
 * P[i].FullTreeGravAccel and P[i].Potential are updated as grav_short_postprocess does (gravshort.h:47-67).
 * After the call TreeUseBH > 1 is reset to 0 (gravshort-tree.c:148-151). */
int mpg_grav_short_tree(mpg_engine *eng, const mpg_particle_view *pv, const int *ActiveParticle, int64_t NumActiveParticle,
                        double (*AccelStore)[3], double rho0);

/* ---- device-resident drop-in mode ------------------------------------------------------------------------------------------
 * The host-pointer calls above upload Pos / Mass and download their results on every call, because the caller may have touched P[]
 * in between (run.c:392-548 drifts, kicks and exchanges on the host).  A caller that lets the engine integrate as well
 * (mpg_dev_drift_all_particles / mpg_dev_apply_pm_half_kick / mpg_dev_apply_half_kick, drift.c:18-102, timestep.c:873-1036, on the
 * arrays of mpg_resident_arrays) declares its table RESIDENT: mpg_resident_begin uploads Pos, Mass, Type / flags, Vel,
 * FullTreeGravAccel, GravPM and Potential once; from then on mpg_gravpm_force, mpg_force_tree_full / _rebuild_mask and
 * mpg_grav_short_tree called with the same view run on the device copies and leave their results there (P[] on the host goes stale:
 * AccelStore, when given, still receives its copy).  mpg_resident_fetch writes the named columns back into P[] for the host modules
 * that read them, mpg_resident_push takes columns the host changed, mpg_resident_end fetches everything and leaves the mode.
 * Anything that reorders or resizes P[] (domain_exchange, slots_gc) goes between _end and a new _begin. */
#define MPG_FIELD_POS 1u
#define MPG_FIELD_VEL 2u
#define MPG_FIELD_ACCEL 4u     /* FullTreeGravAccel */
#define MPG_FIELD_GRAVPM 8u
#define MPG_FIELD_POTENTIAL 16u
typedef struct mpg_resident_view {
    int64_t n;
    double *d_pos;            /* [n][3] */
    float *d_mass;            /* [n] */
    unsigned char *d_type;    /* [n] (7 = garbage / swallowed) */
    double *d_vel;            /* [n][3] or NULL when the view had no Vel */
    double *d_fulltree_accel; /* [n][3] */
    double *d_gravpm;         /* [n][3] */
    double *d_potential;      /* [n] */
} mpg_resident_view;
int mpg_resident_begin(mpg_engine *eng, const mpg_particle_view *pv, double BoxSize);
int mpg_resident_arrays(mpg_engine *eng, mpg_resident_view *out);
int mpg_resident_fetch(mpg_engine *eng, const mpg_particle_view *pv, unsigned fields);
int mpg_resident_push(mpg_engine *eng, const mpg_particle_view *pv, unsigned fields);
int mpg_resident_end(mpg_engine *eng, const mpg_particle_view *pv);

/* (a gas run stays resident too: mpg_resident_sph_begin and the integrator calls further down, behind the types they take) */

/* ---- device-resident entry points (inputs and outputs stay in HBM) ---------------------------- */
/* Bind device arrays in the caller's particle order: pos[n][3] f64, mass[n] f32, type[n] u8 (NULL = all type 1).
 * The arrays must stay valid until the next bind. */
int mpg_dev_bind_particles(mpg_engine *eng, int64_t n, const double *d_pos, const float *d_mass, const uint8_t *d_type,
                           double BoxSize);
/* gravpm_force on bound particles: d_gravpm[n][3] (=), d_potential[n] (+=; may be NULL). */
int mpg_dev_gravpm_force(mpg_engine *eng, double *d_gravpm, double *d_potential);
/* force_tree_full / force_tree_rebuild_mask on bound particles.  Called right after mpg_dev_gravpm_force (the order of run.c:522-548)
 * the build runs on an engine-internal second stream next to the PM force - neither depends on the other - and everything queued
 * afterwards waits for both.  The bound positions must not change between the two calls other than through this API
 * (mpg_dev_bind_particles / mpg_dev_drift_all_particles switch the overlap off for the next build); MPG_NO_TREE_OVERLAP=1 disables it. */
int mpg_dev_force_tree_build(mpg_engine *eng, int mask);
/* grav_short_tree on bound particles.  d_oldacc[n] = |FullTreeGravAccel + GravPM| / G (grav_get_abs_accel,
 * gravshort.h:70-80) or NULL to have it computed from d_prev_accel[n][3] + d_gravpm[n][3].
 * d_active: int32 target indices or NULL (all).  d_accel[n][3] receives G * Acc for every target;
 * d_potential[n] (may be NULL) receives the post-processed potential. */
int mpg_dev_grav_short_tree(mpg_engine *eng, const double *d_oldacc, const double *d_prev_accel, const double *d_gravpm,
                            const int *d_active, int64_t nactive, double *d_accel, double *d_potential, double rho0);

/* Work measure for the domain decomposition (the reference balances TopLeaves by cost, domain.c:611): while d_cost is set, the
 * two-kernel walk writes d_cost[i] = 8 x leaf-list entries + nodes used + 8 x traversal steps for every target i (caller order;
 * other entries untouched).  NULL switches it off. */
int mpg_dev_set_walk_cost(mpg_engine *eng, float *d_cost);

/* ---- SPH: density (with the smoothing-length iteration) and hydro force ----------------------------- */
/* struct density_params, libgadget/density.h:10-25 (same fields, same order; DensityKernelType is the enum value
 * 1 cubic / 2 quintic / 4 quartic of densitykernel.h:17-21). */
typedef struct mpg_density_params {
    double DensityResolutionEta;
    double MaxNumNgbDeviation;
    double BlackHoleNgbFactor;
    double BlackHoleMaxAccretionRadius;
    int DensityKernelType;
    double MinGasHsmlFractional;
} mpg_density_params;
/* struct hydro_params, libgadget/hydra.c:26-34 */
typedef struct mpg_hydro_params {
    int DensityIndependentSphOn;
    double DensityContrastLimit;
    double ArtBulkViscConst;
} mpg_hydro_params;
/* set_densitypar (density.c:22-27) / the hydro_params of set_hydro_params (hydra.c:36-48) */
int mpg_set_densitypar(mpg_engine *eng, const mpg_density_params *dp);
int mpg_set_hydropar(mpg_engine *eng, const mpg_hydro_params *hp);
/* GetNumNgb (density.c:53-59) for the current parameters */
double mpg_get_numngb(mpg_engine *eng);

/* Time-dependent scalars the reference derives from DriftKickTimes / Cosmology before the loops:
 * kick_factor_data (density.h:34-39, filled by init_kick_factor_data density.c:115-132), the per-bin density drift
 * factors of hydra.c:178-186, dloga_from_dti(Ti_Current - Ti_kick[bin]) used by SPH_EntVarPred (density.c:75),
 * get_dloga_for_bin (hydra.c:271,463), and atime / hubble_function(atime) (hydra.c:219-223).  Index = time bin, 0..46. */
typedef struct mpg_sph_times {
    double FgravkickB;
    double gravkicks[47];
    double hydrokicks[47];
    double drifts[47];
    double dloga_kick[47];
    double dloga_bin[47];
    double atime, hubble;
} mpg_sph_times;

/* Device arrays of the particle table in caller order (n = bound particles).  SPH slot fields (SphP[P[i].PI].X,
 * slotsmanager.h:93-129) are indexed by PARTICLE here; entries of non-gas particles are ignored.  NULL = absent/zero
 * for the optional inputs (gacc, gpm, hydroacc_in, tb_*, dtentropy_in) and optional outputs (dthsml, gradrho). */
typedef struct mpg_sph_arrays {
    double *hsml;                 /* in/out  P.Hsml */
    double *dthsml;               /* out     P.DtHsml */
    const double *vel;            /* [n][3]  P.Vel */
    const double *gacc;           /* [n][3]  P.FullTreeGravAccel */
    const double *gpm;            /* [n][3]  P.GravPM */
    const double *hydroacc_in;    /* [n][3]  SphP.HydroAccel (previous step, for the velocity prediction) */
    const uint8_t *tb_hydro;      /* P.TimeBinHydro */
    const uint8_t *tb_grav;       /* P.TimeBinGravity */
    const double *entropy;        /* SphP.Entropy */
    const double *dtentropy_in;   /* SphP.DtEntropy (previous step, for the entropy prediction) */
    double *density;              /* out SphP.Density (BHP.Density for type 5 targets) */
    double *egywtdensity;         /* out SphP.EgyWtDensity */
    double *dhsmlegyfac;          /* out SphP.DhsmlEgyDensityFactor */
    double *divvel;               /* out SphP.DivVel */
    double *curlvel;              /* out SphP.CurlVel */
    double *gradrho;              /* out [n][3] or NULL */
    double *hydroacc_out;         /* out [n][3] SphP.HydroAccel */
    double *dtentropy_out;        /* out SphP.DtEntropy */
    double *maxsignalvel;         /* out SphP.MaxSignalVel */
} mpg_sph_arrays;

/* force_tree_rebuild_mask without moments (forcetree.c:151-166) on the bound particles; with_moments != 0 also runs
 * force_tree_calc_moments (needed by set_init_hsml). */
int mpg_dev_force_tree_rebuild_mask(mpg_engine *eng, int mask, int with_moments);
/* set_init_hsml, libgadget/density.c:691-749 (tree with moments, mask GASMASK+BHMASK in the reference) */
int mpg_dev_set_init_hsml(mpg_engine *eng, const mpg_sph_arrays *A, double MeanGasSeparation);
/* density, libgadget/density.c:234-355.  d_active NULL = all particles.  Targets: gas and (non-swallowed) black holes. */
int mpg_dev_density(mpg_engine *eng, const mpg_sph_arrays *A, const mpg_sph_times *T, const int *d_active, int64_t nactive,
                    int update_hsml, int DoEgyDensity, int BlackHoleOn);
/* force_tree_calc_moments after density (run.c:477): propagates the leaf hmax of the final Hsml up the gas tree */
int mpg_dev_force_tree_calc_hmax(mpg_engine *eng);
/* force_update_hmax (forcetree.h:113, forcetree.c:1290-1340): the hmax moments of the current (gas) tree from the smoothing lengths in
 * d_hsml[n] (caller order; only gas and black holes count), without a density loop before it (test_forcetree.c:257-292) */
int mpg_dev_force_update_hmax(mpg_engine *eng, const double *d_hsml);
/* hydro_force, libgadget/hydra.c:153-245 (needs density() and the hmax moments of the same tree) */
int mpg_dev_hydro_force(mpg_engine *eng, const mpg_sph_arrays *A, const mpg_sph_times *T, const int *d_active, int64_t nactive);
/* Host-pointer forms of the four calls above: `P` supplies Pos / Mass / Type / flags (the reference AoS table), `A` holds HOST
 * arrays in particle order (the in-tree shim gathers SphP[P[i].PI].X into them).  Inputs are copied to HBM,
 * outputs copied back; the gas tree is (re)built inside, as force_tree_rebuild_mask does in run.c:466. */
int mpg_set_init_hsml(mpg_engine *eng, const mpg_particle_view *pv, double BoxSize, const mpg_sph_arrays *A, double MeanGasSeparation);
int mpg_density(mpg_engine *eng, const mpg_particle_view *pv, double BoxSize, const mpg_sph_arrays *A, const mpg_sph_times *T,
                const int *ActiveParticle, int64_t NumActiveParticle, int update_hsml, int DoEgyDensity, int BlackHoleOn);
int mpg_hydro_force(mpg_engine *eng, const mpg_particle_view *pv, const mpg_sph_arrays *A, const mpg_sph_times *T,
                    const int *ActiveParticle, int64_t NumActiveParticle);
/* statistics of the last SPH call: [0] density iterations, [1] targets summed over iterations,
 * [2] successful distance tests / hydro pairs evaluated, [3] candidates distance-tested */
int mpg_sph_get_stats(mpg_engine *eng, int64_t stats[4]);

/* ---- introspection (tests, bench roofline accounting) ----------------------------------------- */
typedef struct mpg_tree_stats {
    int64_t NumParticles; /* particles in the tree */
    int64_t numnodes;     /* live nodes */
    int64_t numleaves;
    int32_t maxlevel;
    double root_mass, root_cofm[3], root_hmax;
} mpg_tree_stats;
int mpg_tree_get_stats(mpg_engine *eng, mpg_tree_stats *st);
/* Copy the node table to host arrays (each may be NULL): level[i], center[i][3], len[i], cofm[i][3], mass[i],
 * hmax[i], sibling[i] (-1 = none), pstart[i], pcount[i] (0 for internal nodes).  Nodes are in depth-first
 * pre-order, children in octant order x | y<<1 | z<<2 (forcetree.c:278-284). */
int mpg_tree_export(mpg_engine *eng, int32_t *level, double *center, double *len, double *cofm, double *mass, double *hmax,
                    int32_t *sibling, int32_t *pstart, int32_t *pcount);
/* Tree-order permutation: order[k] = caller index of the k-th particle in tree order (n entries). */
int mpg_tree_export_order(mpg_engine *eng, int32_t *order);
/* Walk counters of the last grav_short_tree: [0] particle-particle interactions (= the reference's Ninteractions,
 * treewalk.c:904-912), [1] nodes visited, [2] nodes used unopened, [3] targets, [4..7] phase statistics of the
 * cooperative kernel: phase-A group steps, nodes consumed by them (of 8 tested each), phase-B lane-steps issued, of which active; [8],[9] wave clock cycles spent in phase A / phase B (summed over waves). */
int mpg_walk_get_counters(mpg_engine *eng, int64_t counters[10]);
/* ... and of the list kernel's fp32 pre-classification of the node tests (MPG_LISTS_F32=1; counting walks only): [0] target passes that fell
 * back to the fp64 tests, [1] waves (8 targets) whose main loop ran the fp32 form */
int mpg_walk_get_f32_stats(mpg_engine *eng, int64_t out[2]);
/* Per-phase device times (ms, HIP events on the engine stream) of the last call of each phase. */
typedef struct mpg_phase_times {
    float pm_deposit, pm_fft, pm_transfer, pm_readout, pm_total;
    float tree_keys, tree_sort, tree_nodes, tree_moments, tree_total;
    float walk;
    int32_t walk_launches;
} mpg_phase_times;
int mpg_get_phase_times(mpg_engine *eng, mpg_phase_times *t);
/* Enable (1) / disable (0) event timing + walk counters (counters cost a few % in the walk kernel). */
int mpg_set_instrumentation(mpg_engine *eng, int timing, int counters);
/* HIP events are recorded (without synchronising) on the engine stream around every walk-kernel launch.  This call
 * synchronises the stream, returns the summed duration (ms) and number of launches since the last collect. */
int mpg_walk_events_collect(mpg_engine *eng, double *total_ms, int *count);
/* ... and, of the walks that ran as ONE list kernel + ONE evaluation kernel, the two kernels' times (an event between them) */
int mpg_walk_events_collect2(mpg_engine *eng, double *total_ms, int *count, double *lists_ms, double *eval_ms, int *count_split);
/* Device pointer to the tree-order permutation (int32 [NumParticles]: tree slot -> caller index) of the current tree;
 * a contiguous slice of it is a spatially compact active list (used to shard targets over GPUs). */
const int *mpg_dev_tree_order(mpg_engine *eng);
/* grav_short_pair (gravshort-pair.c:21-57): the exact pair-wise short-range force over all particles within Rcut * Asmth * cell
 * size of each target (a sphere, not the tree walk's cube), same softening spline and window; runtests.c:131 checks the tree force
 * against it.  Needs a tree (any walk order); results as for grav_short_tree: FullTreeGravAccel and Potential when the tree holds
 * every particle. */
int mpg_grav_short_pair(mpg_engine *eng, const mpg_particle_view *pv, const int *ActiveParticle, int64_t NumActiveParticle, double Rcut,
                        double rho0);
int mpg_dev_grav_short_pair(mpg_engine *eng, const int *d_active, int64_t nactive, double Rcut, double *d_accel, double *d_potential,
                            double rho0);

/* ---- time integration on device-resident arrays: the streaming loops the reference runs between force
 * steps.  Arrays are in particle order with n entries; flags[i] is the bit-field byte of struct particle_data (bit 0 IsGarbage,
 * bit 1 Swallowed; may be NULL).  Results are bit-identical to the reference's loops (no FMA contraction).
 * Not carried: black-hole repositioning (drift.c:33-55) and the dynamic-friction / drag kicks of type 5 (timestep.c:1003-1010). */
#define MPG_TIMEBINS 46 /* timebinmgr.h:8 */
typedef struct mpg_kick_factors_s {
    double gravkick[MPG_TIMEBINS + 1];  /* get_exact_gravkick_factor(Ti_kick[bin], Ti_kick[bin] + dti/2); 0 for inactive bins */
    double hydrokick[MPG_TIMEBINS + 1]; /* get_exact_hydrokick_factor, same interval (timestep.c:878-890) */
    double dt_entr[MPG_TIMEBINS + 1];   /* dloga_from_dti(dti_from_timebin(bin) / 2, Ti_Current)  (timestep.c:917) */
    unsigned char bin_active[MPG_TIMEBINS + 1]; /* is_timebin_active(bin, Ti_Current) */
    double atime, MaxGasVel;            /* TimestepParams.MaxGasVel (timestep.c:1026) */
} mpg_kick_factors;
/* drift_all_particles (drift.c:84-102): Pos += Vel * ddrift + random_shift, wrapped into (0, BoxSize]; gas Hsml += DtHsml *
 * ddrift, capped at BoxSize/2.  Returns non-zero (the reference's endrun(5)) for Hsml <= 0 or a non-finite position.
 * d_type, d_flags, d_hsml, d_dthsml may be NULL (then no smoothing lengths are drifted). */
int mpg_dev_drift_all_particles(mpg_engine *eng, int64_t n, double *d_pos, const double *d_vel, const unsigned char *d_type,
                                const unsigned char *d_flags, double *d_hsml, const double *d_dthsml, double ddrift, double BoxSize,
                                const double random_shift[3]);
/* apply_PM_half_kick (timestep.c:964-985): Vel += GravPM * Fgravkick */
int mpg_dev_apply_pm_half_kick(mpg_engine *eng, int64_t n, double *d_vel, const double *d_gravpm, const unsigned char *d_flags, double Fgravkick);
/* apply_half_kick (timestep.c:873-929): short-range gravity kick of the particles in active gravity bins and the hydro kick
 * (velocity, gas velocity limit, entropy) of gas.  d_active == NULL: all n particles.  d_tb_grav / d_tb_hydro NULL: bin 0. */
int mpg_dev_apply_half_kick(mpg_engine *eng, int64_t n, const int *d_active, int64_t nactive, double *d_vel, const double *d_gravaccel,
                            const unsigned char *d_type, const unsigned char *d_flags, const unsigned char *d_tb_grav,
                            const unsigned char *d_tb_hydro, const double *d_hydroaccel, double *d_entropy, const double *d_dtentropy,
                            const mpg_kick_factors *K);

/* get_timestep_gravity_dloga (timestep.c:1039-1074) for every particle: dloga = H * sqrt(2 ErrTolIntAccuracy a (FORCE_SOFTENING/2.8)
 * / |a_phys|), a_phys = (FullTreeGravAccel + GravPM) / a^2.  Uses the softening set by mpg_gravshort_set_softenings. */
int mpg_dev_timestep_gravity_dloga(mpg_engine *eng, int64_t n, const double *d_gravaccel, const double *d_gravpm, double atime, double hubble,
                                   double ErrTolIntAccuracy, double *d_dloga);

/* get_timestep_hydro_dloga (timestep.c:1076-1118) for every particle: gas takes the Courant criterion 2 CourantFac a Hsml / (fac3
 * MaxSignalVel), fac3 = a^(3 (1 - GAMMA) / 2), or, when shorter, the Gadget-4 criterion on the change of the smoothing length CourantFac
 * a^2 |Hsml / (DtHsml + 1e-20)|; a black hole the step of the bin above the shortest of its gas neighbours (d_bh_mintimebin = BHP().minTimeBin
 * per particle and dloga_for_bin[b] = get_dloga_for_bin(b, Ti_Current), a host array of MPG_TIMEBINS + 1 entries; both NULL: no limiter);
 * every other type dt = 1.  d_titype (may be NULL) receives enum TimeStepType (timestep.c:89-96: 0 ACCEL, 1 COURANT, 3 NEIGH, 4 HSML).
 * d_type NULL: no gas; d_dthsml NULL: zero. */
int mpg_dev_timestep_hydro_dloga(mpg_engine *eng, int64_t n, const unsigned char *d_type, const double *d_hsml, const double *d_dthsml,
                                 const double *d_maxsignalvel, const unsigned char *d_bh_mintimebin, const double *dloga_for_bin, double atime,
                                 double hubble, double CourantFac, double *d_dloga, unsigned char *d_titype);

/* ---- hierarchical gravity (SplitGravityTimestepsOn, the default): the level loop of timestep.c:239-599 on
 * device-resident arrays.  Per level it builds the tree of the particles active at that level (force_tree_active_moments),
 * walks it for exactly those particles into a separate acceleration array (grav_short_tree with AccelStore) and applies the
 * hierarchical half kick; the first half of a step also assigns the gravity time bins.  The integer timeline, the kick times
 * and the kick integrals stay with the caller: it passes its sync points, its DriftKickTimes and a callback for
 * get_exact_gravkick_factor.  Not carried here: get_PM_timestep_ti (the caller passes the PM step length it found), the hydro
 * bins (find_hydro_timesteps), star / black-hole particles created during the step. */
typedef struct {            /* the integer timeline: SyncPoints[i].loga (timebinmgr.c:18, 372-417); host array, nsync >= 2 */
    int64_t nsync;
    const double *loga;
} mpg_timeline;
typedef struct {            /* DriftKickTimes, timestep.h:10-27 (same fields, same order) */
    int mintimebin, maxtimebin, mingravtimebin;
    int64_t Ti_kick[MPG_TIMEBINS + 1];
    int64_t Ti_lastactivedrift[MPG_TIMEBINS + 1];
    int64_t Ti_Current;
    int64_t PM_length, PM_start, PM_kick;
} mpg_drift_kick_times;
typedef double (*mpg_gravkick_fn)(void *ctx, int64_t ti0, int64_t ti1); /* get_exact_gravkick_factor(CP, ti0, ti1), timefac.c:65-68 */
typedef struct {            /* device arrays in particle order (the particles bound with mpg_dev_bind_particles) */
    double *d_vel;                  /* P[].Vel              [n][3] */
    const double *d_gravpm;         /* P[].GravPM           [n][3] */
    double *d_fulltree_accel;       /* P[].FullTreeGravAccel[n][3] */
    double *d_potential;            /* P[].Potential        [n] (written by walks over a full tree; may be NULL) */
    unsigned char *d_tb_grav;       /* P[].TimeBinGravity   [n] */
    const unsigned char *d_flags;   /* bit 0 IsGarbage, bit 1 Swallowed; may be NULL */
    double *d_stored_accel;         /* StoredGravAccel.GravAccel [n][3], or NULL: FullTreeGravAccel plays its part */
} mpg_hiergrav_arrays;
typedef struct {            /* TimestepParams (timestep.c:40-60) as far as this path reads them */
    double ErrTolIntAccuracy, MinSizeTimestep;
} mpg_timestep_params;
/* hierarchical_gravity_and_timesteps (timestep.c:293-490).  d_active: the active list (NULL = all n particles), with
 * NumActiveGravity of them gravitationally active.  On a PM step (Ti_Current == PM_start + PM_length) dti_max_pm is the
 * caller's get_PM_timestep_ti and times->PM_length / PM_start are updated as the reference does.  rho0 = the mean density
 * of grav_short_tree.  Returns in *badstepsizecount the number of particles print_bad_timebin would have reported. */
int mpg_dev_hierarchical_gravity_and_timesteps(mpg_engine *eng, const mpg_hiergrav_arrays *A, const int *d_active, int64_t NumActiveParticle,
                                               int64_t NumActiveGravity, mpg_drift_kick_times *times, const mpg_timeline *timeline,
                                               const mpg_timestep_params *par, double atime, double hubble, int64_t dti_max_pm, double rho0,
                                               int HybridNuGrav, mpg_gravkick_fn gravkick, void *gravkick_ctx, int64_t *badstepsizecount);
/* hierarchical_gravity_accelerations (timestep.c:495-599): the second half of the step. */
int mpg_dev_hierarchical_gravity_accelerations(mpg_engine *eng, const mpg_hiergrav_arrays *A, const int *d_active, int64_t NumActiveParticle,
                                               int64_t NumActiveGravity, mpg_drift_kick_times *times, double rho0, int HybridNuGrav,
                                               mpg_gravkick_fn gravkick, void *gravkick_ctx);
/* find_hydro_timesteps (timestep.c:617-733), the assignment of the hydro time bins of the active gas / black-hole particles, in its two
 * halves: the particle loop on the device (new bin from get_timestep_hydro_dloga through convert_timestep_to_ti and get_timebin_from_dti,
 * never above the particle's gravity bin, written to TimeBinHydro when old and new bin are active), and - after the caller's MPI_Allreduce of
 * the smallest bin and the counts, where it has ranks - the update of times->mintimebin (host).  Not carried: the dynamic-friction bins of the
 * black holes (timestep.c:676-695).  Uses the particles bound with mpg_dev_bind_particles for n. */
typedef struct {
    const unsigned char *d_type;          /* P[].Type; NULL: no gas */
    const unsigned char *d_flags;         /* bit 0 IsGarbage, bit 1 Swallowed; may be NULL */
    const double *d_hsml, *d_dthsml;      /* P[].Hsml, P[].DtHsml (may be NULL: zero) */
    const double *d_maxsignalvel;         /* SPHP().MaxSignalVel (hydro_force) */
    const unsigned char *d_tb_grav;       /* P[].TimeBinGravity; NULL: no cap */
    unsigned char *d_tb_hydro;            /* P[].TimeBinHydro, updated */
    const unsigned char *d_bh_mintimebin; /* BHP().minTimeBin per particle, or NULL */
} mpg_hydrostep_arrays;
typedef struct {
    int mTimeBin;               /* the smallest new bin among this rank's particles (MPG_TIMEBINS if it has none): MPI_MIN over the ranks */
    int64_t ntitype[5];         /* particles by criterion: TI_ACCEL, TI_COURANT, TI_ACCRETE, TI_NEIGH, TI_HSML (MPI_SUM) */
    int64_t badstepsizecount;   /* bin_hydro < 1 (MPI_SUM; the function's return value in the reference) */
    int64_t badtimebins;        /* print_bad_timebin cases: dti <= 1 or > TIMEBASE */
} mpg_hydrostep_result;
int mpg_dev_find_hydro_timesteps(mpg_engine *eng, const mpg_hydrostep_arrays *A, const int *d_active, int64_t NumActiveParticle,
                                 const mpg_drift_kick_times *times, const mpg_timeline *timeline, const mpg_timestep_params *par, double CourantFac,
                                 double atime, double hubble, mpg_hydrostep_result *out);
/* the tail of find_hydro_timesteps (timestep.c:709-733) with the all-reduced smallest bin: the rule for a bin without active particles,
 * set_bh_first_timestep on the first step (d_type / d_tb_hydro of n particles; may be NULL when isFirstTimeStep == 0), times->mintimebin */
int mpg_dev_hydro_timesteps_finish(mpg_engine *eng, int mTimeBin_global, int isFirstTimeStep, int64_t n, const unsigned char *d_type,
                                   unsigned char *d_tb_hydro, mpg_drift_kick_times *times);
/* find_timesteps (timestep.c:739-849): the step assignment of a run WITHOUT SplitGravityTimestepsOn (run.c:756) - gravity step from
 * FullTreeGravAccel + GravPM (get_timestep_gravity_dloga), for gas / black holes the hydro step where it is shorter, TimeBinHydro and
 * TimeBinGravity both set to the new bin when old and new bin are active.  On a PM step (Ti_Current == PM_start + PM_length) dti_max_pm is the
 * caller's get_PM_timestep_ti and times->PM_length / PM_start are updated as the reference does; the finish call (after the caller's
 * MPI_Allreduce of mTimeBin (MIN), maxTimeBin (MAX) and the counts, where it has ranks) shrinks the PM step onto the longest tree step and
 * sets times->mintimebin / maxtimebin.  Not carried: ForceEqualTimesteps, set_bh_first_timestep (the caller keeps both). */
typedef struct {
    int mTimeBin, maxTimeBin;   /* smallest / largest new bin among this rank's active particles (MPG_TIMEBINS / 0 if it has none) */
    int isPM;                   /* this was a PM step */
    int64_t ntitype[5];         /* particles by criterion: TI_ACCEL, TI_COURANT, TI_ACCRETE, TI_NEIGH, TI_HSML */
    int64_t badstepsizecount;   /* bin < 1 (the function's return value in the reference) */
    int64_t badtimebins;        /* print_bad_timebin cases: dti <= 1 or > TIMEBASE */
} mpg_timestep_result;
int mpg_dev_find_timesteps(mpg_engine *eng, const mpg_hydrostep_arrays *A, const double *d_fulltree_accel, const double *d_gravpm,
                           unsigned char *d_tb_grav, const int *d_active, int64_t NumActiveParticle, mpg_drift_kick_times *times,
                           const mpg_timeline *timeline, const mpg_timestep_params *par, double CourantFac, double atime, double hubble,
                           int64_t dti_max_pm, mpg_timestep_result *out);
int mpg_find_timesteps_finish(int mTimeBin_global, int maxTimeBin_global, int isPM, mpg_drift_kick_times *times);
/* A GAS run stays resident too (round 5).  mpg_resident_sph_begin, on a resident table, uploads every array of `A` (host arrays in
 * particle order, as the host forms mpg_density / mpg_hydro_force take them) ONCE; its vel / gacc / gpm then alias the resident
 * P[].Vel / FullTreeGravAccel / GravPM, and the *_in arrays of the velocity / entropy prediction alias the *_out arrays (SphP.HydroAccel,
 * SphP.DtEntropy: one field each in the reference).  From then on mpg_density / mpg_hydro_force called with the same `A` run on the device
 * copies and leave their results there, and the integrator between the force steps runs on the device as well:
 *   mpg_resident_drift_all_particles   drift_all_particles (drift.c:84-102): Pos, and Hsml += DtHsml * ddrift for gas
 *   mpg_resident_apply_pm_half_kick    apply_PM_half_kick (timestep.c:964-985)
 *   mpg_resident_apply_half_kick       apply_half_kick (timestep.c:873-929): gravity kick, hydro kick, gas velocity limit, entropy
 *   mpg_resident_find_timesteps        find_timesteps (timestep.c:739-849): the step assignment of run.c:756 (no SplitGravityTimestepsOn)
 *   mpg_resident_find_hydro_timesteps  find_hydro_timesteps (timestep.c:617-733) on TimeBinHydro (both halves, one rank; several ranks:
 *                                      mpg_dev_find_hydro_timesteps + the caller's MPI_Allreduce + mpg_dev_hydro_timesteps_finish on
 *                                      mpg_resident_sph_arrays)
 * mpg_resident_sph_end writes every array back to `A` (Entropy and the time bins included: the kicks and the bin assignment changed them)
 * and leaves the mode; mpg_resident_end does the same for the table.  shim/timestep-hip.c forwards the reference's own entry points. */
int mpg_resident_sph_begin(mpg_engine *eng, const mpg_particle_view *pv, const mpg_sph_arrays *A);
int mpg_resident_sph_arrays(mpg_engine *eng, mpg_sph_arrays *device_arrays_out);
int mpg_resident_sph_end(mpg_engine *eng, const mpg_sph_arrays *A);
/* the resident time bins into host arrays of n bytes (either may be NULL): build_active_particles (timestep.c:1333-1420) reads
 * P[].TimeBinHydro / TimeBinGravity on the host at the top of every step, so the shim copies them into P[] after every bin assignment */
int mpg_resident_fetch_timebins(mpg_engine *eng, unsigned char *tb_hydro, unsigned char *tb_grav);
int mpg_resident_drift_all_particles(mpg_engine *eng, const mpg_particle_view *pv, double ddrift, const double random_shift[3]);
int mpg_resident_apply_pm_half_kick(mpg_engine *eng, const mpg_particle_view *pv, double Fgravkick);
int mpg_resident_apply_half_kick(mpg_engine *eng, const mpg_particle_view *pv, const int *ActiveParticle, int64_t NumActiveParticle,
                                 const mpg_kick_factors *K);

/* find_timesteps on a resident gas run, one rank (both halves): accelerations, Hsml, DtHsml, MaxSignalVel and the time bins are the resident ones */
int mpg_resident_find_timesteps(mpg_engine *eng, const mpg_particle_view *pv, const int *ActiveParticle, int64_t NumActiveParticle,
                                mpg_drift_kick_times *times, const mpg_timeline *timeline, const mpg_timestep_params *par, double CourantFac,
                                double atime, double hubble, int64_t dti_max_pm, mpg_timestep_result *out);
/* both halves on a resident gas run (mpg_resident_sph_begin), one rank: Hsml, DtHsml, MaxSignalVel and the time bins are the resident ones */
int mpg_resident_find_hydro_timesteps(mpg_engine *eng, const mpg_particle_view *pv, const int *ActiveParticle, int64_t NumActiveParticle,
                                      mpg_drift_kick_times *times, const mpg_timeline *timeline, const mpg_timestep_params *par, double CourantFac,
                                      double atime, double hubble, int isFirstTimeStep, mpg_hydrostep_result *out);

/* build_active_sublist (timestep.c:1435-1478): the entries of d_active (NULL = all n) that are not garbage, whose gravity bin
 * is <= maxtimebin and active at Ti_Current, order preserved.  d_out must hold NumActiveParticle entries. */
int mpg_dev_build_active_sublist(mpg_engine *eng, const int *d_active, int64_t NumActiveParticle, const unsigned char *d_tb_grav,
                                 const unsigned char *d_flags, int maxtimebin, int64_t Ti_Current, int *d_out, int64_t *n_out);

/* ---- particle order: Peano-Hilbert keys and the (type, key) sort the reference keeps its particles in
 * (slots_gc_sorted after every full domain decomposition, domain.c:247).  Keys are bit-identical to the reference's. */
/* ---- Peano-Hilbert domain decomposition (libgadget/domain.c) -------------------------------------------------------------
 * The reference's domain_decompose_full (domain.c:153-258) in the pieces its MPI code is made of: the device passes over a
 * rank's particles, the arithmetic on the top-level tree (host; no GPU needed), and - left to the caller, where the reference
 * calls MPI - the sums, the pairwise hand-over of trees and the all-to-all of particle records (INTEGRATION.md shows the
 * sequence; mp-gadget_amd/domain_peano.py is that caller over torch.distributed). */
typedef struct mpg_topnode {            /* struct local_topnode_data, domain.c:60-70, + Leaf of struct topnode_data, domain.h:12-18 */
    uint64_t StartKey;
    int32_t Shift, Daughter, Parent, Leaf;
    int64_t Count, Cost;
} mpg_topnode;
/* the rank's sample of keys for the local refinement, sorted (domain.c:1031-1083 with DomainUseGlobalSorting = 0; for the
 * global sort the caller gathers the ranks' samples and sorts them): every SubSampleDistance-th particle, after a key sort
 * that drops garbage if PreSort.  keys_out: HOST array of `cap` entries; d_garbage: device bytes (IsGarbage) or NULL */
int mpg_dev_domain_sample(mpg_engine *eng, int64_t n, const double *d_pos, const unsigned char *d_garbage, double BoxSize, int PreSort,
                          int SubSampleDistance, uint64_t *keys_out, int64_t cap, int64_t *nsample);
/* domain_check_for_local_refine_subsample from the sorted sample on (domain.c:1085-1180); costs NULL: 1 per sample.
 * *failed = 1: MaxTopNodes too small (the caller enlarges TopNodeAllocFactor and retries, domain.c:182-194) */
int mpg_domain_local_refine(const uint64_t *keys, const int64_t *costs, int64_t nsample, mpg_topnode *tree, int *size, int MaxTopNodes, int *failed);
/* domain_toptree_truncate, domain.c:954-967 */
int mpg_domain_toptree_truncate(mpg_topnode *tree, int *size, int64_t countlimit, int64_t costlimit);
/* one receive step of domain_nonrecursively_combine_topTree (domain.c:1232-1247): tree B of rank ThisTask + sep merged into A */
int mpg_domain_toptree_merge(mpg_topnode *A, int *sizeA, const mpg_topnode *B, int sizeB, int MaxTopNodes, int *failed);
/* domain_global_refine, domain.c:1344-1395 */
int mpg_domain_global_refine(mpg_topnode *tree, int *size, int MaxTopNodes, int64_t countlimit, int64_t costlimit, int *failed);
/* domain_create_topleaves, domain.c:810-824: sets tree[].Leaf, fills leaf_topnode[*nleaves] */
int mpg_domain_create_topleaves(mpg_topnode *tree, int size, int *leaf_topnode, int *nleaves);
/* domain_assign_topleaves_balanced + domain_set_task_leafs (domain.c:610-786): reorders the leaves by (Task, Key) - leaf_topnode
 * and tree[].Leaf are rewritten - and fills leaf_task[nleaves], StartLeaf[NTask], EndLeaf[NTask] */
int mpg_domain_assign_topleaves_balanced(mpg_topnode *tree, int size, int *leaf_topnode, int nleaves, const int64_t *cost, int NTask,
                                         int NsegmentPerTask, int *leaf_task, int *StartLeaf, int *EndLeaf);
/* one pass over the rank's particles: P[].TopLeaf (domain.c:216-225, -1 for garbage) into d_topleaf, the destination task
 * (domain_layoutfunc, domain.c:794-802) into d_task, particles per leaf (domain_compute_costs, domain.c:1398-1457, before the
 * Allreduce) into leaf_counts[nleaves] and per destination task into task_counts[NTask] (host arrays).  leaf_task, d_topleaf,
 * d_task, leaf_counts, task_counts may be NULL. */
int mpg_dev_domain_topleaves(mpg_engine *eng, int64_t n, const double *d_pos, const unsigned char *d_garbage, double BoxSize,
                             const mpg_topnode *tree, int size, int nleaves, const int *leaf_task, int NTask, int32_t *d_topleaf,
                             int32_t *d_task, int64_t *leaf_counts, int64_t *task_counts);

/* PEANO(Pos, BoxSize), libgadget/utils/peano.h:15-21 + peano_hilbert_key, peano.c:117-140: d_keys[n] */
int mpg_dev_peano_keys(mpg_engine *eng, int64_t n, const double *d_pos, double BoxSize, uint64_t *d_keys);
/* the order of slots_gc_sorted (slotsmanager.c:404-452): d_perm[k] = index of the k-th particle by (Type, Key), garbage
 * (flags bit 0) last; *n_live = the particles that are not garbage.  d_type NULL: one type.  Equal (type, key) keep their
 * input order (the reference's qsort leaves it unspecified). */
int mpg_dev_order_by_type_and_key(mpg_engine *eng, int64_t n, const unsigned char *d_type, const unsigned char *d_flags,
                                  const uint64_t *d_keys, int *d_perm, int64_t *n_live);

/* ---- friends-of-friends groups (libgadget/fof.c) on the bound particles: primary linking of the
 * FOFPrimaryLinkTypes at the comoving linking length, secondary particles attached to the nearest primary one, groups of at
 * least FOFHaloMinLength members numbered by (Length descending, MinID).  A group is labelled by its smallest particle ID.
 * Not carried: the sub-grid group properties (Sfr, metals, black-hole masses, MaxDens / seed index) and groups that span ranks. */
typedef struct mpg_fof_params {     /* struct FOFParams, fof.c:37-48, as far as this path reads it */
    int FOFPrimaryLinkTypes;        /* bit mask of particle types, default 2 (dark matter) */
    int FOFSecondaryLinkTypes;      /* default 1 + 16 + 32 (gas, stars, black holes); disjoint from the primary types */
    double FOFHaloComovingLinkingLength; /* FOFHaloLinkingLength * mean DM separation (fof_init, fof.c:83-86) */
    int FOFHaloMinLength;
} mpg_fof_params;
typedef struct mpg_fof_groups {     /* struct Group / BaseGroup, fof.h:14-49: device arrays of *ngroups entries; NULL = not wanted */
    uint64_t *MinID;
    int *Length, *GrNr, *LenType;   /* LenType[g][6] */
    double *Mass, *MassType;        /* MassType[g][6] */
    double *CM, *Vel, *Jmom;        /* [g][3] */
    double *Imom;                   /* [g][3][3] */
    float *FirstPos;                /* [g][3] */
} mpg_fof_groups;
/* fof_fof (fof.c:157-253) with StoreGrNr: d_id[n] = P[].ID, d_vel[n][3] (may be NULL: zero), d_hsml[n] (may be NULL: no radius hint
 * for gas), d_flags (may be NULL).  d_grnr[n] (may be NULL) receives P[].GrNr (-1: in no group); *ngroups = fof.TotNgroups.
 * Builds the tree of the primary types itself (force_tree_rebuild_mask) and leaves it in place. */
int mpg_dev_fof_fof(mpg_engine *eng, const mpg_fof_params *par, const uint64_t *d_id, const double *d_vel, const double *d_hsml,
                    const unsigned char *d_flags, int64_t *d_grnr, int64_t *ngroups);
/* the group table of the last mpg_dev_fof_fof, in MinID order (the order of fof.Group) */
int mpg_dev_fof_groups(mpg_engine *eng, const mpg_fof_groups *out);

/* ---- snapshot / IC wire format: the "bigfile" blocks petaio.c reads and writes (libgadget/petaio.c:986-1120;
 * format: depends/bigfile/src/bigfile.c).  `file` is the snapshot directory (PART_000, an MP-GenIC output), `block` a column such
 * as "1/Position" or "Header".  dtypes are bigfile's: "f8", "f4", "i8", "u8", "i4", "u4", "u1", "S1" (little-endian).  Host IO only;
 * files written here are read by the reference library and vice versa (tests/test_snapshot_io.py). */
typedef struct mpg_bigblock_info {
    char dtype[8];   /* normalised, e.g. "<f8" */
    int nmemb;       /* values per element (3 for Position) */
    int nfile;       /* physical files of the block */
    int64_t size;    /* elements */
} mpg_bigblock_info;
int mpg_bigfile_block_info(const char *file, const char *block, mpg_bigblock_info *info);            /* big_file_open_block */
/* big_block_read with a cast: `count` elements from element `start`, converted to want_dtype, into out */
int mpg_bigfile_read_block(const char *file, const char *block, int64_t start, int64_t count, const char *want_dtype, void *out);
/* big_file_create_block + big_block_write: `size` elements of `nmemb` values stored as `dtype`, split evenly over nfile files
 * (bigfile-mpi.c:106-111); data holds src_dtype values (NULL: dtype) */
int mpg_bigfile_write_block(const char *file, const char *block, const char *dtype, int nmemb, int nfile, int64_t size, const char *src_dtype,
                            const void *data);
/* big_block_get_attr / big_block_set_attr; get returns 2 when the attribute does not exist */
int mpg_bigfile_get_attr(const char *file, const char *block, const char *name, const char *want_dtype, void *out, int nmemb);
int mpg_bigfile_set_attr(const char *file, const char *block, const char *name, const char *dtype, const void *data, int nmemb);

/* ---- matter power spectrum of the PM density field: gravpm_force measures it on every PM step (measure_power_spectrum /
 * powerspectrum_add_mode, gravpm.c:331-382: |delta_k|^2 with the CIC window removed once, weight 2 off the kz = 0 / Nyquist planes,
 * Nmesh logarithmic bins) and writes powerspectrum-<a>.txt (powerspectrum_sum / powerspectrum_save, powerspectrum.c:55-122).
 * The engine accumulates the raw sums during mpg_(dev_)gravpm_force and mpg_dev_pm_slab_forward_b; the neutrino linear-response
 * correction (gravpm.c:307-327, 415-436) is not carried. */
int mpg_gravpm_measure_power(mpg_engine *eng, int on);   /* default on, as in the reference */
/* raw sums of the last PM step into caller device arrays: d_acc[2 Nmesh + 1] = Power[Nmesh], kk[Nmesh], Norm; d_modes[Nmesh].
 * One process per GPU: sum them over the ranks (the MPI_Allreduce of powerspectrum_sum) before mpg_powerspectrum_sum. */
int mpg_dev_gravpm_powerspectrum_raw(mpg_engine *eng, double *d_acc, int64_t *d_modes);
/* powerspectrum_sum (powerspectrum.c:55-91) on host arrays: averages, normalises by Norm, converts to Mpc/h units with BoxSize_in_MPC
 * and drops empty bins; kk, Power, Nmodes hold nbins entries, *nonzero of which are filled. */
int mpg_powerspectrum_sum(int nbins, const double *acc, const int64_t *modes, double BoxSize_in_MPC, double *kk, double *Power,
                          int64_t *Nmodes, int *nonzero);
/* both steps for one GPU */
int mpg_gravpm_get_powerspectrum(mpg_engine *eng, double BoxSize_in_MPC, double *kk, double *Power, int64_t *Nmodes, int *nonzero);
/* powerspectrum_save (powerspectrum.c:93-122): OutputDir/filename-<Time>.txt with the reference's columns "k P N P(z=0)" */
int mpg_powerspectrum_save(const char *OutputDir, const char *filename, double Time, double D1, int nonzero, const double *kk,
                           const double *Power, const int64_t *Nmodes);

/* ---- long-range PM over several GPUs, one process per GPU (petapm.c:584-885 exchanges region meshes with 2-D pencils and lets
 * PFFT transpose; here: x-slabs of Nmesh/world planes, two all-to-all transposes per PM step and one neighbour plane).  The
 * engine does the local stages; the caller (one rank per GPU) does the collectives between them on the engine's stream:
 *
 *   mpg_dev_pm_slab_init(rank, world)        after mpg_gravpm_init_periodic; Nmesh % world == 0.  Outputs the number of complex
 *                                            values per peer block (P * Py * (Nmesh/2+1)) and the doubles of one mesh plane.
 *   mpg_dev_pm_slab_forward_a(sendA)         deposit the bound particles onto this rank's planes, 2-D r2c, pack:
 *                                            sendA[world][cplx_per_peer] complex      -> all-to-all -> recvA
 *   mpg_dev_pm_slab_forward_b(recvA, sendB)  1-D c2c along x, potential_transfer, inverse 1-D c2c of the potential:
 *                                            sendB[world][cplx_per_peer] complex      -> all-to-all -> recvB
 *   mpg_dev_pm_slab_inverse_c(recvB, ghost_send)  2-D c2r: the potential on this rank's planes; ghost_send[5][plane] = its first 3
 *                                            planes (for the previous rank) and its last 2 (for the next rank)
 *   mpg_dev_pm_slab_readout(ghost_recv, targets, n, GravPM, Potential)   ghost_recv[5][plane] = the next rank's first 3 planes, then
 *                                            the previous rank's last 2.  Forces = 4-point differences of the potential (the
 *                                            real-space form of force_transfer, gravpm.c:456-489), then readout_force_* /
 *                                            readout_potential (gravpm.c:499-510) for the listed particles, whose base cell must
 *                                            lie in this rank's slab.
 * Every rank binds the same particle set (mpg_dev_bind_particles); a particle's CIC cloud is deposited by the owners of the
 * planes it touches, so no region exchange is needed.  world == 1 reproduces mpg_dev_gravpm_force to FFT round-off. */
int mpg_dev_pm_slab_init(mpg_engine *eng, int rank, int world, int64_t *cplx_per_peer, int64_t *plane_doubles);

/* ---- particles distributed over ranks (the reference: Peano-Hilbert domains, a replicated top-tree whose leaves carry the
 * moments of remote sub-trees, forcetree.c:1106-1290; here: x-slab domains, ghosts imported in whole columns of level-La
 * cells).  The bound particle set is [own particles | ghosts]; the tree built from it equals the global tree at levels >= La;
 * the nodes above get their moments from sums over all ranks:
 *   mpg_dev_tree_top_partial(La, n_own, out)  out[8^(La-1)][4] = (sum m, sum m x, sum m y, sum m z) of the OWN particles
 *                                             (caller index < n_own) per level-(La-1) cell, cells numbered by octant path
 *   -> all-reduce over ranks; coarser levels by summing 8 children
 *   mpg_dev_tree_top_set(La, sums)            sums[(8^La - 1)/7][4]: levels 0 .. La-1 concatenated; sets the moments of
 *                                             every local node above level La (error if such a node is a leaf) */
int mpg_dev_tree_top_partial(mpg_engine *eng, int La, int64_t n_own, double *d_out);
int mpg_dev_tree_top_set(mpg_engine *eng, int La, const double *d_sums);
int mpg_dev_pm_slab_forward_a(mpg_engine *eng, double *sendA);
int mpg_dev_pm_slab_forward_b(mpg_engine *eng, double *recvA, double *sendB);
int mpg_dev_pm_slab_inverse_c(mpg_engine *eng, const double *recvB, double *ghost_send);
int mpg_dev_pm_slab_readout(mpg_engine *eng, const double *ghost_recv, const int *d_targets, int64_t ntargets, double *d_gravpm,
                            double *d_potential);

/* ---- the force step on several ranks behind the CALLER's communicator ------------------------------------------------------
 * The reference's entry points are collective over MPI_COMM_WORLD (gravity.h:40,55; forcetree.h:115-148; treewalk.c:801-902;
 * petapm.c:584-885; domain.c:153-258; exchange.c).  This library links neither MPI nor RCCL: the caller hands it the three
 * collectives the path needs as callbacks over its own communicator (run.c: MPI_Allreduce / MPI_Alltoall / MPI_Alltoallv;
 * the Python host side: torch.distributed = RCCL over xGMI), and every rank calls the mpg_dist_* functions collectively, as it
 * calls gravpm_force / force_tree_full / grav_short_tree today.  All choreography (what is sent where, in which order) is in
 * the library (csrc/dist.hip); a callback only moves bytes.
 *
 * Particles live on the rank that owns their Peano-Hilbert TopLeaf (domain_decompose_full).  Per force step:
 *   PM     every particle's {Pos, Mass} goes to the rank(s) owning the x-planes its CIC cloud touches (32 B per particle: an
 *          eighth of what the reference's region meshes cost at Nmesh = 2 N^(1/3)); slab FFT with two all-to-all transposes;
 *          GravPM / Potential return to the owners.
 *   tree   a rank imports, as ghosts, the particles of every level-La tree cell within Rcut of its TopLeaves (whole cells: the
 *          local tree equals the global one at levels >= La); the nodes above get their moments from an all-reduce and are
 *          kept internal down to level La, the counterpart of the reference's replicated top-tree with pseudo nodes
 *          (forcetree.c:654-723, 1145-1284).  Ghost import replaces the export of walk targets (treewalk.c:325-793): the
 *          interaction sets are the same, there is no return trip.
 * Callback contract: return 0 on success; `on_device` says whether the buffers are device pointers (only if device_buffers
 * was set; otherwise the library stages through pinned host memory).  Counts and displacements are in BYTES. */
typedef struct mpg_comm {
    void *ctx;                 /* handed back to every callback (e.g. the MPI_Comm) */
    int ThisTask, NTask;       /* NTask <= 64 */
    int device_buffers;        /* 1: alltoallv / allreduce accept device pointers (RCCL, GPU-aware MPI) */
    /* MPI_Allreduce(MPI_IN_PLACE, buf, count, dtype ? MPI_INT64_T : MPI_DOUBLE, op ? MPI_MAX : MPI_SUM) */
    int (*allreduce)(void *ctx, void *buf, int64_t count, int dtype, int op, int on_device);
    /* MPI_Alltoall of ONE int64 per peer, host memory */
    int (*alltoall_i64)(void *ctx, const int64_t *send, int64_t *recv);
    /* MPI_Alltoallv of bytes */
    int (*alltoallv)(void *ctx, const void *send, const int64_t *sendbytes, const int64_t *sdispls, void *recv, const int64_t *recvbytes,
                     const int64_t *rdispls, int on_device);
    /* Optional (NULL: every call blocks until its data has arrived, as MPI does).  A communicator whose device-buffer collectives are
     * stream-ordered work (RCCL) sets bind_stream: mpg_dist_create calls it once with the engine's hipStream_t.  From then on a callback
     * that is handed device pointers ENQUEUES on that stream and returns at once, and the library does not synchronise around it:
     * kernels and collectives of a force step run back to back.  Calls with host pointers stay blocking. */
    int (*bind_stream)(void *ctx, void *hip_stream);
} mpg_comm;

/* ---- mpg_comm on a native RCCL communicator (csrc/rccl_comm.hip) --------------------------------------------------------------
 * Stands where the reference has MPI on this path: the query export / import of the tree walks (treewalk.c:586-655: MPI_Alltoall of
 * counts, MPI_Isend / MPI_Irecv per peer) and the PM mesh exchanges (petapm.c:751,815,869: MPI_Alltoallv) become ncclGroupStart +
 * ncclSend / ncclRecv per peer + ncclGroupEnd and ncclAllReduce on the engine's stream, device memory to device memory over xGMI.
 * librccl is opened at run time (the library does not link it).  Bootstrap: rank 0 obtains the 128-byte id, the CALLER hands it to
 * every rank (MPI_Bcast: shim/mpg_rccl_mpi.c; torch.distributed: bench.py; shared memory: tests/c/test_cabi.c), then every rank calls
 * mpg_rccl_create (collective: ncclCommInitRank) on ITS device and passes the callbacks of mpg_rccl_comm to mpg_dist_create.
 * mpg_rccl_selftest runs every collective once on a known pattern (collective) and fails if a byte differs. */
#define MPG_RCCL_ID_BYTES 128
typedef struct mpg_rccl mpg_rccl;
int mpg_rccl_available(void);                 /* 1 if librccl could be opened */
int mpg_rccl_get_unique_id(void *id128);
int mpg_rccl_create(mpg_rccl **out, int ThisTask, int NTask, const void *id128, int device);
int mpg_rccl_comm(mpg_rccl *r, mpg_comm *out);
int mpg_rccl_selftest(mpg_rccl *r, int64_t bytes_per_peer);
/* calls3: allreduce, alltoall_i64, alltoallv calls so far; bytes sent to OTHER ranks; the RCCL version code (any may be NULL) */
int mpg_rccl_stats(mpg_rccl *r, int64_t *calls3, int64_t *bytes_sent, int *version);
/* what RCCL reports for the communicator: ncclCommCount, ncclCommUserRank, and the HIP device it was created on (any may be NULL) */
int mpg_rccl_comm_info(mpg_rccl *r, int *nranks, int *rank, int *device);
const char *mpg_rccl_last_error(mpg_rccl *r); /* detail of the last failed callback */
void mpg_rccl_destroy(mpg_rccl *r);

typedef struct mpg_dist mpg_dist;
/* eng: configured as for one rank (mpg_gravpm_init_periodic with the GLOBAL Nmesh, tables, tree parameters, softening);
 * Nmesh % NTask == 0.  The comm struct is copied. */
int mpg_dist_create(mpg_dist **out, mpg_engine *eng, const mpg_comm *comm);
void mpg_dist_destroy(mpg_dist *d);
/* domain_decompose_full (domain.c:153-258) over the communicator: policies, the rank's key sample, local refinement, the pairwise
 * combination of the ranks' trees, global refinement, the TopLeaves and their balanced assignment (by particle number, or by the
 * per-particle work d_cost[n] when given: mpg_dist_walk_cost), P[].TopLeaf and the destination task of every particle.  d_garbage:
 * device bytes (IsGarbage) or NULL.  Then mpg_dist_domain_exchange moves the particles (domain_exchange, exchange.c): ncols <= 16
 * columns of col_bytes[j] bytes per particle each (device arrays over the n particles); on return *n_new particles live on this
 * rank and d_new_cols[j] point to their columns in library-owned device buffers (valid until the next exchange), the particles that
 * stayed and those that arrived in source-rank order; garbage is dropped.  mpg_dist_domain_get copies the decomposition out (arrays
 * sized by the counts mpg_dist_domain_decompose returned; any may be NULL); mpg_dist_use_decomposition hands it to the force step
 * (= mpg_dist_set_domain with it). */
int mpg_dist_domain_decompose(mpg_dist *d, int64_t n, const double *d_pos, const unsigned char *d_garbage, double BoxSize,
                              int DomainOverDecompositionFactor, int DomainUseGlobalSorting, const float *d_cost, int *NTopNodes,
                              int *NTopLeaves);
int mpg_dist_domain_get(mpg_dist *d, mpg_topnode *TopNodes, int *leaf_task, int *StartLeaf, int *EndLeaf, int64_t *TopLeafCount);
int mpg_dist_domain_exchange(mpg_dist *d, int64_t n, int ncols, const void *const *d_cols, const int *col_bytes, int64_t *n_new,
                             void **d_new_cols);
int mpg_dist_use_decomposition(mpg_dist *d, double BoxSize, double margin, int La);
/* domain_maintain (domain.c:262-319) after a drift: the decomposition is kept, TopLeaf and destination task of every particle are
 * found again; *n_leaving (may be NULL) = the particles that mpg_dist_domain_exchange will now move to other ranks */
int mpg_dist_domain_maintain(mpg_dist *d, int64_t n, const double *d_pos, const unsigned char *d_garbage, double BoxSize, int64_t *n_leaving);
/* The domain the ranks' particles were distributed by: the TopNodes of domain_decompose_full (same on every rank) and the Task
 * of every TopLeaf (DomainDecomp.TopNodes / TopLeaves, domain.h:12-43).  margin: at least Rcut in length units (the walk's
 * cut-off, gravshort-tree.c:102) and, for SPH, the largest smoothing length; La = 0 picks the level whose cells are
 * [margin, 2 margin) wide. */
int mpg_dist_set_domain(mpg_dist *d, double BoxSize, const mpg_topnode *TopNodes, int NTopNodes, const int *leaf_task, int NTopLeaves,
                        double margin, int La);
/* gravpm_force + force_tree_full + grav_short_tree for the rank's OWN particles (device arrays, n_own rows; all active).
 * d_prev_accel (may be NULL): last step's FullTreeGravAccel for the relative opening criterion, else d_oldacc (|a|/G per
 * particle, may be NULL -> Barnes-Hut walk if TreeUseBH says so).  Outputs: d_gravpm[n_own][3] assigned, d_accel[n_own][3]
 * assigned, d_potential[n_own] (may be NULL) = PM + tree potential as the reference leaves it in P[].Potential. */
int mpg_dist_gravity_step(mpg_dist *d, int64_t n_own, const double *d_pos, const float *d_mass, const double *d_oldacc,
                          const double *d_prev_accel, double *d_accel, double *d_gravpm, double *d_potential, double rho0);
/* The three entry points separately, in the reference's order (run.c:522-548) - gravity_step is exactly this sequence:
 *   gravpm_force (gravity.h:55)          -> GravPM assigned, Potential accumulated (readout_potential, gravpm.c:499-501)
 *   force_tree_full (forcetree.h:115)    -> ghost import, local tree with the global top
 *   grav_short_tree (gravity.h:40)       -> accelerations of all own particles (and the tree potential, assigned) */
int mpg_dist_dev_gravpm_force(mpg_dist *d, int64_t n_own, const double *d_pos, const float *d_mass, double *d_gravpm, double *d_potential);
/* Garbage and swallowed particles among the own rows (P[].IsGarbage, P[].Swallowed: star formation and black-hole mergers leave them in
 * the table until the next domain_decompose_full collects them).  The reference skips them in place in every loop of this path
 * (treewalk.c:234, forcetree.c:806, gravpm.c:176-179); so do the mpg_dist_dev_* calls that follow on a table of n_own rows: such rows are
 * shipped nowhere, are not in the local tree, are no targets; their GravPM is zeroed, their other outputs are left alone.
 * d_garbage: n_own device bytes (non-zero = skip) or NULL for none; stays in force until the next call.  The host forms
 * (mpg_dist_gravpm_force, ...) read the flag byte of the particle view themselves. */
int mpg_dist_dev_set_garbage(mpg_dist *d, int64_t n_own, const unsigned char *d_garbage);
int mpg_dist_dev_force_tree_build(mpg_dist *d, int64_t n_own, const double *d_pos, const float *d_mass);
int mpg_dist_dev_grav_short_tree(mpg_dist *d, const double *d_oldacc, const double *d_prev_accel, const double *d_gravpm, double *d_accel,
                                 double *d_potential, double rho0);
/* Hierarchical gravity (timestep.c: hierarchical_gravity_accelerations; force_tree_active_moments, forcetree.c:129-148): the short-range
 * force of the ACTIVE particles on each other, on the tree of the active particles only.  d_pos / d_mass / d_oldacc hold the rank's
 * n_act active particles (compacted by the caller), d_accel[n_act][3] is assigned; d_potential is not touched (the tree is not the full
 * particle tree: gravshort.h:57-67 leaves P[].Potential and FullTreeGravAccel alone then).  Such sets are sparse and small: every rank receives the whole set (one all-gather of
 * 28 bytes per particle), builds the tree one GPU would build and walks its own members, so the results are those of one GPU.  The
 * local tree of mpg_dist_dev_force_tree_build is replaced (build it again before the next walk on all particles). */
int mpg_dist_dev_grav_short_tree_active_tree(mpg_dist *d, int64_t n_act, const double *d_pos, const float *d_mass, const double *d_oldacc,
                                             double *d_accel, double *d_potential, double rho0);
/* ... as a drop-in call: the active particles are gathered from the rank's P[] (OldAcc from P[].FullTreeGravAccel + GravPM), their
 * AccelStore[i] is assigned, P[] is left alone.  ActiveParticle == NULL: all particles of the table. */
int mpg_dist_grav_short_tree_active_tree(mpg_dist *d, const mpg_particle_view *pv, const int *ActiveParticle, int64_t NumActiveParticle,
                                         double (*AccelStore)[3], double rho0);
/* the same for a subset of the own particles: d_active[nactive] = their indices (device array, no duplicates), the ActiveParticle list of a
 * sub-step (run.c:392-470: the tree holds every particle, the active ones are walked).  Only their entries of d_accel / d_potential are
 * written.  d_active == NULL: all own particles. */
int mpg_dist_dev_grav_short_tree_active(mpg_dist *d, const int *d_active, int64_t nactive, const double *d_oldacc, const double *d_prev_accel,
                                        const double *d_gravpm, double *d_accel, double *d_potential, double rho0);
/* density() and hydro_force() (density.h:42, hydra.h) for the rank's own gas on the particle set of the last
 * mpg_dist_dev_force_tree_build (own + ghosts; the domain margin must cover the largest smoothing length: checked).  d_type and the
 * arrays of A are device arrays over the n_own own particles (mpg_sph_arrays; optional inputs may be NULL); the ghosts' columns
 * travel along the ghost plan, and between the two loops the ghosts' Hsml / Density / EgyWtDensity / DhsmlEgyDensityFactor / DivVel /
 * CurlVel are refreshed from their owners. */
int mpg_dist_dev_density(mpg_dist *d, int64_t n_own, const uint8_t *d_type, const mpg_sph_arrays *A, const mpg_sph_times *T, int update_hsml,
                         int DoEgyDensity);
int mpg_dist_dev_hydro_force(mpg_dist *d, int64_t n_own, const mpg_sph_arrays *A, const mpg_sph_times *T);
/* ... for a sub-step: only the own gas among d_active[nactive] (device array of own-particle indices; NULL = all) is treated; the arrays
 * of A must then hold, for the other particles, the results of their last loops (they are read: the hydro loop and the ghosts need the
 * inactive particles' Density ... CurlVel) and those entries come back unchanged, as the reference leaves SphP of inactive particles */
int mpg_dist_dev_density_active(mpg_dist *d, int64_t n_own, const uint8_t *d_type, const mpg_sph_arrays *A, const mpg_sph_times *T,
                                const int *d_active, int64_t nactive, int update_hsml, int DoEgyDensity);
int mpg_dist_dev_hydro_force_active(mpg_dist *d, int64_t n_own, const mpg_sph_arrays *A, const mpg_sph_times *T, const int *d_active,
                                    int64_t nactive);
/* fof_fof (fof.c:157-253) with the particles on their owners: groups may span ranks.  Device arrays over the n_own own particles
 * (d_type NULL: all type 1; d_vel NULL: zero); the linking length must not exceed the domain margin.  d_grnr[n_own] (may be NULL)
 * receives P[].GrNr with GLOBAL group numbers (length descending, then MinID); *ngroups_total = fof.TotNgroups; this rank keeps the
 * *ngroups_here groups with MinID % NTask == ThisTask: mpg_dist_fof_groups copies their table into HOST arrays. */
int mpg_dist_dev_fof_fof(mpg_dist *d, int64_t n_own, const double *d_pos, const float *d_mass, const uint8_t *d_type, const uint64_t *d_id,
                         const double *d_vel, const mpg_fof_params *par, int64_t *d_grnr, int64_t *ngroups_total, int64_t *ngroups_here);
int mpg_dist_fof_groups(mpg_dist *d, const mpg_fof_groups *out);
/* ... and as drop-in calls on the rank's particle table in host memory (what libgadget's callers hand over; shim/gravity-hip.c):
 * Pos / Mass are read from P[], GravPM / FullTreeGravAccel / Potential (and AccelStore, may be NULL) are written as the reference's
 * functions write them; OldAcc of the walk comes from P[].FullTreeGravAccel + P[].GravPM.  All particles active (a PM step). */
int mpg_dist_gravpm_force(mpg_dist *d, const mpg_particle_view *pv);
int mpg_dist_force_tree_full(mpg_dist *d, const mpg_particle_view *pv);
int mpg_dist_grav_short_tree(mpg_dist *d, const mpg_particle_view *pv, double (*AccelStore)[3], double rho0);
/* ... for the sub-steps: ActiveParticle[NumActiveParticle] (host array of indices into P[], NULL = all) are walked, their
 * FullTreeGravAccel / Potential / AccelStore entries updated (the tree of mpg_dist_force_tree_full holds every particle) */
int mpg_dist_grav_short_tree_active(mpg_dist *d, const mpg_particle_view *pv, const int *ActiveParticle, int64_t NumActiveParticle,
                                    double (*AccelStore)[3], double rho0);
/* density() / hydro_force() as drop-in calls on the same table (after mpg_dist_force_tree_full on it): A holds HOST arrays in particle
 * order, as for mpg_density / mpg_hydro_force (the shim gathers SphP[P[i].PI].X into them); inputs are read, outputs written.
 * ActiveParticle[NumActiveParticle]: host array of indices into P[] (NULL = all), see mpg_dist_dev_density_active */
/* BlackHoleOn of density() (density.c:234) for the following mpg_dist_(dev_)density calls.  The own black holes that are not swallowed
 * are targets of the density loop next to the gas in any case (density_haswork, density.c:521-530: Hsml, Density and DivVel of BHP);
 * BlackHoleOn gives them BlackHoleNgbFactor times the neighbours and caps their radius (density.c:598-600, 667-670).  The smoothing
 * lengths of both must lie within the domain margin.  Swallowed black holes and garbage carry type 7 in d_type (the host form reads
 * the flag bits). */
int mpg_dist_set_sph_options(mpg_dist *d, int BlackHoleOn);
/* the largest smoothing length over all ranks at the end of the last mpg_dist_(dev_)density loop.  When that call failed because it
 * exceeds the domain margin (the neighbours of such a particle are not all local), set the domain again with a margin above this
 * value, rebuild the local tree and repeat the call: the inputs were not modified. */
double mpg_dist_last_max_hsml(mpg_dist *d);
/* PartManager->MaxPart for domain_check_memory_bound (domain.c:378-424): a decomposition that gives one task more particles is retried
 * with the next policy (domain.c:199-201).  0 (default): no bound. */
int mpg_dist_domain_set_maxpart(mpg_dist *d, int64_t MaxPart);
/* the matter power spectrum of the last mpg_dist_(dev_)gravpm_force over all ranks (gravpm.c:110-118; powerspectrum_sum's
 * MPI_Allreduce, powerspectrum.c:55-91).  Collective; arrays of Nmesh entries, *nonzero of which are filled on every rank. */
int mpg_dist_gravpm_get_powerspectrum(mpg_dist *d, double BoxSize_in_MPC, double *kk, double *Power, int64_t *Nmodes, int *nonzero);
int mpg_dist_density(mpg_dist *d, const mpg_particle_view *pv, const mpg_sph_arrays *A, const mpg_sph_times *T, const int *ActiveParticle,
                     int64_t NumActiveParticle, int update_hsml, int DoEgyDensity);
int mpg_dist_hydro_force(mpg_dist *d, const mpg_particle_view *pv, const mpg_sph_arrays *A, const mpg_sph_times *T, const int *ActiveParticle,
                         int64_t NumActiveParticle);
/* per-particle work of the last mpg_dist walk for the rank's own particles (device pointer, n_own floats, caller order):
 * the cost the next domain decomposition balances (mpg_dev_set_walk_cost) */
const float *mpg_dist_walk_cost(mpg_dist *d);
/* what the last step moved: [0] ghosts imported, [1] particles shipped to PM slabs, [2] local tree particles (own + ghosts),
 * [3] decomposition level La, [4] bytes sent in personalised exchanges, [5] bytes sent in transposes */
int mpg_dist_get_stats(mpg_dist *d, int64_t stats[8]);
/* phase times of the last step in ms (host clock around stream synchronisations): [0] PM shipping + slab PM, [1] ghost import,
 * [2] tree build + global top, [3] walk.  After mpg_dist_gravity_step, which builds the local tree on a second stream beside the PM
 * step (MPG_DIST_NO_OVERLAP=1: one after the other): [0] PM with the tree build beside it, [2] global top + targets, [4] the tree build
 * on its own stream */
int mpg_dist_get_times(mpg_dist *d, double ms[8]);
/* cells of the tree above this level are kept internal (never leaves) by the next tree builds; 0 restores forcetree.c's rule.
 * Used by the distributed step; exposed for tests. */
int mpg_dev_force_tree_set_min_leaf_level(mpg_engine *eng, int level);

/* Tuning knobs of the walk; results do not depend on any of them.
 *   variant   0 = auto (default): kernel 6 for 4096 targets or more, kernel 1 below that;
 *             1 = lane-per-target while-while kernel (grav_walk.hip); 4 = group-cooperative list kernel (grav_walk_coop.hip);
 *             6 = two kernels, list construction then evaluation (grav_walk_split.hip)
 *   threshold (kernel 1) the node phase keeps running while at least that many lanes of a wave still search (default 16)
 *   list capacity (kernels 4, 6) interaction-list entries per target (default 512): kernel 4 drains its lists when they are
 *             full; kernel 6 hands targets with longer lists to kernel 1 and doubles its capacity when > 2 % of them do */
int mpg_set_walk_threshold(mpg_engine *eng, int thresh);
/* kernel 6: overlap != 0 builds the lists of slice k+1 on a second stream while slice k is evaluated (default on);
 * chunks_per_wave = 0 uses persistent grids, > 0 that many 8-target chunks per wave (default 2) */
int mpg_set_walk_split_mode(mpg_engine *eng, int overlap, int chunks_per_wave);
int mpg_set_walk_list_capacity(mpg_engine *eng, int cap);
/* kernel 6: on != 0 takes the kernels' 64-bit-offset variants whatever the array sizes (default: only when the source and node arrays exceed
 * 4 GiB, i.e. from about 512^3 particles in one tree); results do not depend on it */
int mpg_set_walk_offsets64(mpg_engine *eng, int on);
int mpg_set_walk_variant(mpg_engine *eng, int variant);
/* kernel in use (the explicit variant, or the default policy's pick: 6 for >= 4096 targets, else 1; 0 = no walk yet), kernel 6's current list capacity and
 * the number of targets its last walk handed to the fallback kernel (either output may be NULL) */
int mpg_get_walk_choice(mpg_engine *eng, int *variant, int *list_capacity, unsigned *last_overflow);

#ifdef __cplusplus
}
#endif
#endif /* MPGADGET_HIP_H */
