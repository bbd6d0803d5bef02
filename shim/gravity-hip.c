/* libgadget/gravity-hip.c -- the reference's gravity entry points forwarded to libmpgadget_hip.so.
 *
 * Replaces gravpm.o, gravshort-tree.o and gravity.o in the link of libgadget; forcetree.o (with its five constructors
 * renamed: forcetree-hip.c takes their place), treewalk.o and petapm.o stay for the 14 other tree walks and MP-GenIC.  Compiled inside the reference tree with its own headers:
 *     $(CC) $(OPTIMIZE) -I$(MPGADGET_HIP)/include -c gravity-hip.c mpg_mpi_comm.c          link: -lmpgadget_hip
 * (This container cannot compile it - gravity.h pulls in pfft.h and GSL, SURVEY 8(c) - so everything that does not need a
 * reference type lives behind the C-ABI, where tests/c/test_cabi.c exercises it with the same call sequence.)
 *
 * One rank = one GPU.  NTask == 1: the host-pointer calls.  NTask > 1: the mpg_dist_* calls on the rank's own P[] with the
 * domain of ddecomp and MPI as the communicator (mpg_mpi_comm.c) - collective, as the functions they replace. */
#include <mpi.h>
#include <stdlib.h>
#include <string.h>
#include "gravity.h"     /* PetaPM, struct gravshort_tree_params, prototypes of everything defined here */
#include "forcetree.h"   /* ForceTree */
#include "domain.h"      /* DomainDecomp, struct topnode_data / topleaf_data */
#include "partmanager.h" /* P, PartManager */
#include "timestep.h"    /* ActiveParticles, get_atime */
#include "walltime.h"
#include "utils/endrun.h"
#include "utils/mymalloc.h"
#include <mpgadget_hip.h>
#include "mpg_mpi_comm.h"
#include "mpg_shim.h"
#include "mpg_shim_epoch.h"
#include <stddef.h>

_Static_assert(sizeof(struct particle_data) == 160 && __builtin_offsetof(struct particle_data, GravPM) == 88 &&
               __builtin_offsetof(struct particle_data, FullTreeGravAccel) == 64 && __builtin_offsetof(struct particle_data, Potential) == 152,
               "struct particle_data is not the layout mpg_particle_view_reference_layout assumes (partmanager.h:9-71)");

extern const double shortrange_force_kernels[][5]; /* libgadget/shortrange-kernel.c stays in the link as data */

static mpg_engine *E;
static mpg_dist *D;
static mpg_rccl *Rccl;
static MPI_Comm Comm;
static int NTask = 1;

/* ---- what the library was last told (mpg_shim.h) ---- */
static struct mpg_table_key Table = MPG_TABLE_KEY_INIT; /* the particle-table epoch handed to mpg_set_particle_epoch and the
                                                         * (Ti_Current, &P[0], NumPart, sample hash) it was declared for (mpg_shim_epoch.h) */
#define Epoch (Table.epoch)
static DomainDecomp *Domain;       /* the run's decomposition object */
static uint64_t DomainHash;        /* of what mpg_dist_set_domain last received ... */
static double DomainMargin;        /* ... with this margin */
static int64_t DistTreeEpoch = -1; /* epoch of the local tree + ghost plan inside the library (NTask > 1) */

#define ck mpg_shim_ck
void mpg_shim_ck(int rc)
{
    if(rc)
        endrun(5, "mpgadget_hip: %s\n", mpg_last_error());
}

static mpg_engine *eng(void)
{
    if(!E) {
        int rank, local;
        MPI_Comm shm;
        Comm = MPI_COMM_WORLD;
        MPI_Comm_rank(Comm, &rank);
        MPI_Comm_size(Comm, &NTask);
        MPI_Comm_split_type(Comm, MPI_COMM_TYPE_SHARED, rank, MPI_INFO_NULL, &shm); /* rank on this node -> GPU */
        MPI_Comm_rank(shm, &local);
        MPI_Comm_free(&shm);
        ck(mpg_engine_create(&E, local));
        if(NTask > 1) {
            /* the exchanges of the force step on RCCL over xGMI (mpg_rccl_mpi.c) unless MPG_SHIM_COMM=mpi or RCCL cannot be opened:
             * then MPI with host staging (mpg_mpi_comm.c) */
            mpg_comm c;
            const char *want = getenv("MPG_SHIM_COMM");
            if((want && !strcmp(want, "mpi")) || mpg_rccl_mpi_comm(Comm, local, &Rccl, &c) != 0)
                c = mpg_mpi_comm(&Comm);
            message(0, "mpgadget_hip: collectives of the force step on %s\n", Rccl ? "RCCL (device buffers, stream-ordered)" : "MPI (host staging)");
            ck(mpg_dist_create(&D, E, &c));
        }
    }
    return E;
}

mpg_engine *mpg_shim_engine(void) { return eng(); }
mpg_dist *mpg_shim_dist(void) { return eng(), D; }
int mpg_shim_ntask(void) { return eng(), NTask; }

mpg_particle_view mpg_shim_view(void)
{
    mpg_particle_view v;
    mpg_particle_view_reference_layout(&v, P, PartManager->NumPart);
    return v;
}
#define view mpg_shim_view

void mpg_shim_set_domain(DomainDecomp *ddecomp) { Domain = ddecomp; }
void mpg_shim_particles_changed(void)
{
    Table.dirty = 1;
    if(E) /* (a prefetch of the table - mpg_shim_prefetch - may still be reading P[]: it ends before the caller rewrites the records) */
        mpg_host_results_sync(E);
}

/* The end of drift_all_particles (timestep-hip.c calls this after the reference's own drift): P[] is final for the step, so the epoch's one
 * packing pass and its uploads can start now, on a host thread, while run.c goes on with domain_maintain and the active list (run.c:420-470);
 * the step's first force call joins it.  One rank, host path only. */
void mpg_shim_prefetch(inttime_t Ti_Current, double BoxSize)
{
    if(NTask > 1 || mpg_shim_resident())
        return;
    mpg_shim_sync(Ti_Current, 0, BoxSize, 0);
    mpg_particle_view v = mpg_shim_view();
    ck(mpg_host_prefetch(eng(), &v, BoxSize));
}
void mpg_shim_dist_tree_replaced(void) { DistTreeEpoch = -1; }
double mpg_shim_margin(void) { return DomainMargin; }

/* FNV-1a over what mpg_dist_set_domain reads: the decomposition is rewritten in place by every domain_decompose_full (run.c:422,434),
 * so neither the pointer nor NTopLeaves says whether it changed */
static uint64_t domain_hash(const DomainDecomp *dd)
{
    uint64_t h = 1469598103934665603ull;
    int i;
#define MIX(x) (h = (h ^ (uint64_t)(x)) * 1099511628211ull)
    MIX(dd->NTopNodes);
    MIX(dd->NTopLeaves);
    for(i = 0; i < dd->NTopNodes; i++) {
        MIX(dd->TopNodes[i].StartKey);
        MIX(dd->TopNodes[i].Shift);
        MIX(dd->TopNodes[i].Daughter);
        MIX(dd->TopNodes[i].Daughter < 0 ? dd->TopNodes[i].Leaf : -1);
    }
    for(i = 0; i < dd->NTopLeaves; i++)
        MIX(dd->TopLeaves[i].Task);
#undef MIX
    return h;
}

/* the sample hash of mpg_shim_epoch.h over P[]: a reorder or an exchange inside one Ti_Current that keeps &P[0] and NumPart
 * changes it; a caller that knows can still say so at once with mpg_shim_particles_changed() */
static uint64_t table_sample_hash(void)
{
    return mpg_table_sample_hash(P, sizeof(struct particle_data), PartManager->NumPart, offsetof(struct particle_data, ID),
                                 offsetof(struct particle_data, Pos));
}

static void push_domain(double BoxSize, double margin)
{
    const DomainDecomp *dd = Domain;
    mpg_topnode *tn = (mpg_topnode *)mymalloc("mpg_topnodes", dd->NTopNodes * sizeof(mpg_topnode));
    int *task = (int *)mymalloc("mpg_leaftask", dd->NTopLeaves * sizeof(int));
    int i;
    for(i = 0; i < dd->NTopNodes; i++) {
        memset(&tn[i], 0, sizeof(tn[i]));
        tn[i].StartKey = dd->TopNodes[i].StartKey;
        tn[i].Shift = dd->TopNodes[i].Shift;
        tn[i].Daughter = dd->TopNodes[i].Daughter;
        tn[i].Leaf = dd->TopNodes[i].Daughter < 0 ? dd->TopNodes[i].Leaf : -1;
    }
    for(i = 0; i < dd->NTopLeaves; i++)
        task[i] = dd->TopLeaves[i].Task;
    ck(mpg_dist_set_domain(D, BoxSize, tn, dd->NTopNodes, task, dd->NTopLeaves, margin, 0));
    myfree(task);
    myfree(tn);
}

void mpg_shim_sync(inttime_t Ti_Current, double Time, double BoxSize, double margin_want)
{
    eng();
    /* ---- the particle table ---- */
    if(Ti_Current < 0 && Table.ti >= 0 && Time == get_atime((inttime_t)Table.ti))
        Ti_Current = (inttime_t)Table.ti; /* gravpm_force of the step whose density() / grav_short_tree() already came by (run.c:356,522) */
    {
        const uint64_t sample = table_sample_hash();
        int changed = mpg_table_key_differs(&Table, Ti_Current, (const void *)P, PartManager->NumPart, sample);
        if(NTask > 1) /* (the ghost plan and the local trees are rebuilt collectively: every rank takes the same decision) */
            MPI_Allreduce(MPI_IN_PLACE, &changed, 1, MPI_INT, MPI_MAX, Comm);
        if(changed)
            mpg_table_key_take(&Table, Ti_Current, (const void *)P, PartManager->NumPart, sample);
    }
    ck(mpg_set_particle_epoch(E, Epoch));
    /* ---- the decomposition (several ranks) ---- */
    if(NTask > 1) {
        if(!Domain)
            endrun(5, "mpgadget_hip: no DomainDecomp known yet: call mpg_shim_set_domain(ddecomp) after domain_decompose_full\n");
        const uint64_t h = domain_hash(Domain);
        /* every rank must take the same decision: the margin wanted is the largest over the ranks */
        double m = margin_want;
        MPI_Allreduce(MPI_IN_PLACE, &m, 1, MPI_DOUBLE, MPI_MAX, Comm);
        if(m <= 0)
            m = DomainMargin; /* a caller without a range of its own (hydro_force after density, set_init_hsml) keeps what is in force */
        if(m > 0 && (h != DomainHash || m > DomainMargin)) {
            push_domain(BoxSize, m);
            DomainHash = h;
            DomainMargin = m;
            DistTreeEpoch = -1; /* the ghost plan belongs to the old need-map */
        }
    }
}

void mpg_shim_dist_tree(const mpg_particle_view *v)
{
    if(DistTreeEpoch == Epoch)
        return;
    ck(mpg_dist_force_tree_full(D, v));
    DistTreeEpoch = Epoch;
}

/* the clocks the reference charges on this path, fed from the engine's per-phase device times */
static void charge_pm_clocks(void)
{
    mpg_phase_times t;
    if(mpg_get_phase_times(eng(), &t))
        return;
    walltime_add("/PMgrav/init", 1e-3 * t.pm_deposit);     /* pm_init_regions + deposit (petapm.c:280) */
    walltime_add("/PMgrav/r2c", 0.2 * 1e-3 * t.pm_fft);    /* one forward of the five transforms (petapm.c:319) */
    walltime_add("/PMgrav/calc", 1e-3 * t.pm_transfer);    /* transfer functions (petapm.c:341) */
    walltime_add("/PMgrav/c2r", 0.8 * 1e-3 * t.pm_fft);    /* the inverse transforms (petapm.c:346) */
    walltime_add("/PMgrav/readout", 1e-3 * t.pm_readout);  /* petapm.c:355 */
}

/* gravpm_init_periodic -> petapm_init (gravpm.c:51-54, petapm.c:105-223).  The mesh, the plans and the pencil layout live in the
 * engine; the PetaPM object keeps what its other readers use (BoxSize, Asmth, Nmesh, G, CellSize: gravshort-tree.c:102, timestep.c,
 * run.c) and everything petapm_destroy (petapm.c:225-232: runtests.c:204,222 call it on this object) releases, so that it can stay as it
 * is in petapm.o (MP-GenIC and the reionisation PM use that file): the communicator it frees is a duplicate made here, Mesh2Task[0] is
 * allocated where petapm_init allocates it (the same place in the allocator's stack: petapm.c:122), the two plans are NULL
 * (pfft_destroy_plan returns on a null plan).  The regions describe the whole mesh on rank 0's terms: nothing on the device path reads
 * them (the x-slab layout of the multi-rank PM is the library's own). */
void gravpm_init_periodic(PetaPM *pm, double BoxSize, double Asmth, int Nmesh, double G)
{
    int i, NTaskHere;
    memset(pm, 0, sizeof(*pm));
    pm->BoxSize = BoxSize;
    pm->Asmth = Asmth;
    pm->Nmesh = Nmesh;
    pm->G = G;
    pm->CellSize = BoxSize / Nmesh;
    pm->comm = MPI_COMM_WORLD;
    MPI_Comm_size(MPI_COMM_WORLD, &NTaskHere);
    pm->Mesh2Task[0] = (int *)mymalloc2("Mesh2Task", 2 * sizeof(int) * Nmesh);
    pm->Mesh2Task[1] = pm->Mesh2Task[0] + Nmesh;
    for(i = 0; i < Nmesh; i++) { /* x-planes in NTask slabs (the library's decomposition), y undivided */
        pm->Mesh2Task[0][i] = (int)(((int64_t)i * NTaskHere) / Nmesh);
        pm->Mesh2Task[1][i] = 0;
    }
    MPI_Comm_dup(MPI_COMM_WORLD, &pm->priv->comm_cart_2d);
    pm->NTask2d[0] = NTaskHere;
    pm->NTask2d[1] = 1;
    MPI_Comm_rank(MPI_COMM_WORLD, &pm->ThisTask2d[0]);
    pm->ThisTask2d[1] = 0;
    pm->priv->plan_forw = NULL;
    pm->priv->plan_back = NULL;
    pm->priv->fftsize = 0;
    for(i = 0; i < 3; i++) {
        pm->real_space_region.offset[i] = 0;
        pm->real_space_region.size[i] = Nmesh;
        pm->fourier_space_region.offset[i] = 0;
        pm->fourier_space_region.size[i] = i == 2 ? Nmesh / 2 + 1 : Nmesh;
    }
    ck(mpg_gravpm_init_periodic(eng(), BoxSize, Asmth, Nmesh, G));
}

void gravpm_force(PetaPM *pm, DomainDecomp *ddecomp, Cosmology *CP, double Time, double UnitLength_in_cm, const char *PowerOutputDir,
                  double TimeIC)
{
    (void)CP;
    (void)TimeIC;
    walltime_measure("/Misc");
    mpg_shim_set_domain(ddecomp);
    const struct gravshort_tree_params tp = get_gravshort_treepar();
    /* one upload of Pos / Mass serves the calls of this step (run.c:472-548); Ti_Current is not an argument here: matched through Time */
    mpg_shim_sync(-1, Time, pm->BoxSize, tp.Rcut * pm->Asmth * pm->CellSize);
    mpg_particle_view v = view();
    if(NTask == 1)
        ck(mpg_gravpm_force(eng(), &v)); /* writes P[i].GravPM, accumulates P[i].Potential */
    else
        ck(mpg_dist_gravpm_force(D, &v));
    charge_pm_clocks();
    /* the matter power spectrum gravpm_force saves on every PM step (gravpm.c:110-118): summed over the ranks, written by rank 0 */
    if(PowerOutputDir) {
        double *kk = (double *)mymalloc("pk", 2 * pm->Nmesh * sizeof(double)), *pw = kk + pm->Nmesh;
        int64_t *nm = (int64_t *)mymalloc("pkn", pm->Nmesh * sizeof(int64_t));
        int nonzero = 0, rank;
        const double BoxSize_in_MPC = pm->BoxSize * UnitLength_in_cm / 3.085678e24; /* CM_PER_MPC, gravpm.c:112 */
        if(NTask == 1)
            ck(mpg_gravpm_get_powerspectrum(eng(), BoxSize_in_MPC, kk, pw, nm, &nonzero));
        else
            ck(mpg_dist_gravpm_get_powerspectrum(D, BoxSize_in_MPC, kk, pw, nm, &nonzero));
        MPI_Comm_rank(MPI_COMM_WORLD, &rank);
        if(rank == 0)
            ck(mpg_powerspectrum_save(PowerOutputDir, "powerspectrum", Time, 1.0, nonzero, kk, pw, nm));
        myfree(nm);
        myfree(kk);
    }
    walltime_measure("/PMgrav/PowerSpec");
}

void gravshort_fill_ntab(const enum ShortRangeForceWindowType t, const double Asmth)
{
    ck(mpg_gravshort_fill_ntab(eng(), (int)t, Asmth, &shortrange_force_kernels[0][0], 512));
}

void set_gravshort_treepar(struct gravshort_tree_params p)
{
    _Static_assert(sizeof(struct gravshort_tree_params) == sizeof(mpg_gravshort_tree_params), "gravshort_tree_params (gravity.h:9-22)");
    ck(mpg_set_gravshort_treepar(eng(), (const mpg_gravshort_tree_params *)&p));
}

struct gravshort_tree_params get_gravshort_treepar(void)
{
    struct gravshort_tree_params p;
    ck(mpg_get_gravshort_treepar(eng(), (mpg_gravshort_tree_params *)&p));
    return p;
}

void gravshort_set_softenings(double MeanSeparation) { ck(mpg_gravshort_set_softenings(eng(), MeanSeparation)); }
/* (blackhole.c, density.c, timestep.c and run.c read the softening through this) */
double FORCE_SOFTENING(void) { return mpg_force_softening(eng()); }

/* The ForceTree handed in is a descriptor: with forcetree-hip.c in the link force_tree_full() / force_tree_active_moments() build
 * nothing on the host (the reference's OpenMP build took 10 x this whole step), they record mask and active list; the device tree of
 * the gravity path is built here, where Ti_Current says whether this step's upload of the table is still valid (run.c:546-547). */
void grav_short_tree(const ActiveParticles *act, PetaPM *pm, ForceTree *tree, MyFloat (*AccelStore)[3], double rho0, inttime_t Ti_Current)
{
    if(!tree->moments_computed_flag)
        endrun(2, "Gravtree called before tree moments computed!\n");
    walltime_measure("/Misc");
    const struct gravshort_tree_params tp = get_gravshort_treepar();
    /* (run.c calls density() / hydro_force() and, on PM steps, gravpm_force() before this with the same Ti_Current: same epoch, one
     * upload; a redecomposition on a non-PM step - extradomain or needfull, run.c:417-435 - is seen here through the hash) */
    mpg_shim_sync(Ti_Current, 0, tree->BoxSize, tp.Rcut * pm->Asmth * pm->CellSize);
    mpg_particle_view v = view();
    if(NTask == 1) {
        /* the device tree this ForceTree stands for (forcetree-hip.c): the tree of the active particles of a hierarchical gravity level
         * (force_tree_active_moments, timestep.c:287), or the tree of all particles of the mask */
        const struct mpg_deferred_tree *dt = mpg_shim_deferred_tree(tree);
        if(dt && dt->kind == MPG_TREE_ACTIVE && dt->ActiveParticle)
            ck(mpg_force_tree_active_moments(eng(), &v, tree->BoxSize, dt->ActiveParticle, dt->NumActiveParticle, dt->HybridNuTracer));
        else
            ck(mpg_force_tree_rebuild_mask(eng(), &v, tree->BoxSize, tree->mask));
        ck(mpg_grav_short_tree(eng(), &v, act->ActiveParticle, act->NumActiveParticle, AccelStore, rho0));
    }
    else {
        if(!tree->full_particle_tree_flag) {
            /* hierarchical_gravity_accelerations: the tree of force_tree_active_moments holds the active particles only; the library
             * gathers that (small) set on every rank and builds its tree itself, results in AccelStore */
            if(!AccelStore)
                endrun(5, "mpgadget_hip: a walk on an active-only tree needs AccelStore\n");
            ck(mpg_dist_grav_short_tree_active_tree(D, &v, act->ActiveParticle, act->NumActiveParticle, AccelStore, rho0));
            mpg_shim_dist_tree_replaced();
        }
        else {
            /* the gravity tree of this table: built once per epoch unless an SPH loop has replaced it in the meantime */
            mpg_shim_dist_tree(&v);
            ck(mpg_dist_grav_short_tree_active(D, &v, act->ActiveParticle, act->NumActiveParticle, AccelStore, rho0));
        }
    }
    /* gravshort-tree.c:135-144: no export phase on this path (ghosts are imported before the walk), so the top-tree and
     * secondary walks cost nothing; the tree build of the device tree is charged to the reference's build clocks */
    mpg_phase_times t;
    ck(mpg_get_phase_times(eng(), &t));
    walltime_add("/Tree/Build/Nodes", 1e-3 * (t.tree_keys + t.tree_sort + t.tree_nodes));
    walltime_add("/Tree/Build/Moments", 1e-3 * t.tree_moments);
    walltime_add("/Tree/WalkTop", 0);
    walltime_add("/Tree/WalkPrim", 1e-3 * t.walk);
    walltime_add("/Tree/WalkSec", 0);
    walltime_add("/Tree/Reduce", 0);
    walltime_add("/Tree/PostPre", 0);
    walltime_add("/Tree/Wait", 0);
    const double timeall = walltime_measure(WALLTIME_IGNORE);
    walltime_add("/Tree/Misc", timeall - 1e-3 * (t.walk + t.tree_total));
}

/* grav_short_pair (gravshort-pair.c:21-57; runtests.c:131): the exact pair-wise force inside Rcut */
void grav_short_pair(const ActiveParticles *act, PetaPM *pm, ForceTree *tree, double Rcut, double rho0)
{
    (void)pm;
    if(mpg_shim_ntask() > 1)
        endrun(5, "mpgadget_hip: grav_short_pair is a single-rank test helper (runtests.c)\n");
    mpg_shim_particles_changed(); /* (no Ti_Current here: runtests.c moves nothing between its calls, but it may have read a snapshot) */
    mpg_shim_sync(-1, -1, tree->BoxSize, 0);
    mpg_particle_view v = view();
    ck(mpg_force_tree_rebuild_mask(eng(), &v, tree->BoxSize, tree->mask));
    ck(mpg_grav_short_pair(eng(), &v, act->ActiveParticle, act->NumActiveParticle, Rcut, rho0));
    walltime_measure("/Tree/Pairwise");
}
