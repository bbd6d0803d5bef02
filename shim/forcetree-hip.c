/* libgadget/forcetree-hip.c -- the tree CONSTRUCTORS of forcetree.h (forcetree.h:127-148) for a link in which the force loops run on
 * the GPU: the trees that only gravity-hip.c / sph-hip.c consume are never built on the host.
 *
 * What run.c does per step (run.c:462-548):
 *     force_tree_rebuild_mask(&gasTree, GASMASK|BHMASK); density(); force_tree_calc_moments(&gasTree); hydro_force();      [466-489]
 *     force_tree_full(&Tree); grav_short_tree(&Tree); force_tree_free(&Tree);                                             [546-548]
 *     hierarchical gravity: force_tree_active_moments(&Tree, subact); grav_short_tree(...); force_tree_free(&Tree);      [timestep.c:287-289]
 * With gravity-hip.c and sph-hip.c in the link every consumer named there is a device loop that builds its own tree in HBM (3 ms for
 * 256^3 particles) - the reference's OpenMP build (0.8 s for the same set on 16 cores) produced a struct NODE array nobody read.
 *
 * How it is linked ("Tree constructors"): forcetree.c stays in the link for its other 20 functions and for the trees the
 * CPU modules walk, compiled with its five constructors renamed:
 *     forcetree.o: CFLAGS += -Dforce_tree_full=cpu_force_tree_full -Dforce_tree_rebuild_mask=cpu_force_tree_rebuild_mask \
 *                            -Dforce_tree_active_moments=cpu_force_tree_active_moments \
 *                            -Dforce_tree_calc_moments=cpu_force_tree_calc_moments -Dforce_tree_free=cpu_force_tree_free
 * and this file defines the five public names.  A constructor here only RECORDS what was asked for (mask, active list, flags) in
 * the caller's ForceTree and in a small table; the device tree is built by the consumer (grav_short_tree, density, ...: they know
 * Ti_Current, i.e. whether the upload of this step's table is still valid).
 *
 * Which trees are deferred:
 *   force_tree_full, force_tree_active_moments   always: their only consumer is grav_short_tree (run.c:546-548, timestep.c:287-289,
 *                                                runtests.c:123,208; gravpm.c:97 is replaced by gravity-hip.c).
 *   force_tree_rebuild_mask(GASMASK | BHMASK)    the gas tree of run.c:466,608 and init.c:485, consumed by density(), set_init_hsml(),
 *                                                hydro_force().  When sub-grid modules follow (metal_return, blackhole,
 *                                                cooling_and_starformation: run.c:607-665) the maintainer adds ONE line in front of
 *                                                them, `mpg_shim_host_tree(&gasTree);`, which builds the host tree then (P[].Hsml is
 *                                                final by then, so its hmax moments are the ones run.c:477 would have computed).
 *   every other mask                             built on the host at once, as before: those trees belong to CPU modules that walk
 *                                                them immediately (fof.c:177, winds.c:238, veldisp.c:414, bhdynfric.c:295,372).
 * A host walk that is handed a tree which was never materialised stops with a message naming the walk (one guard line at the top of
 * treewalk_run, treewalk.c:802).  The host tree is not built from inside the walk: libgadget's allocator is a stack (utils/memory.c:
 * 376-400), and a tree allocated after a module's own scratch arrays would be in their way when they are freed. */
#include <string.h>
#include "forcetree.h"
#include "partmanager.h"
#include "timestep.h"
#include "walltime.h"
#include "utils/endrun.h"
#include <mpgadget_hip.h>
#include "mpg_shim.h"

/* forcetree.c's own constructors under the names the -D flags above give them */
void cpu_force_tree_full(ForceTree *tree, DomainDecomp *ddecomp, const int HybridNuTracer, const char *EmergencyOutputDir);
void cpu_force_tree_active_moments(ForceTree *tree, DomainDecomp *ddecomp, const ActiveParticles *act, const int HybridNuTracer,
                                   const int alloc_father, const char *EmergencyOutputDir);
void cpu_force_tree_rebuild_mask(ForceTree *tree, DomainDecomp *ddecomp, int mask, const char *EmergencyOutputDir);
void cpu_force_tree_calc_moments(ForceTree *tree, DomainDecomp *ddecomp);
void cpu_force_tree_free(ForceTree *tree);

#define MPG_MAX_DEFERRED 8 /* run.c holds two trees at a time (gasTree, Tree) */
static struct mpg_deferred_tree Deferred[MPG_MAX_DEFERRED];

static struct mpg_deferred_tree *find(const ForceTree *tree)
{
    int i;
    for(i = 0; i < MPG_MAX_DEFERRED; i++)
        if(Deferred[i].tree == tree)
            return &Deferred[i];
    return NULL;
}

const struct mpg_deferred_tree *mpg_shim_deferred_tree(const ForceTree *tree)
{
    const struct mpg_deferred_tree *r = find(tree);
    return (r && !r->materialised) ? r : NULL;
}

/* the descriptor a deferred tree leaves in the caller's ForceTree: every flag as the host constructor would set it, no node memory.
 * Nodes[firstnode] is a one-node stand-in so that run.c:481's "Root hmax" message reads a number (filled by force_tree_calc_moments). */
static struct mpg_deferred_tree *defer(ForceTree *tree, DomainDecomp *ddecomp, int kind, int mask, const ActiveParticles *act, int HybridNuTracer,
                                       int alloc_father, const char *dir)
{
    struct mpg_deferred_tree *r = find(NULL);
    if(!r)
        endrun(5, "mpgadget_hip: more than %d trees alive at once\n", MPG_MAX_DEFERRED);
    memset(r, 0, sizeof(*r));
    r->tree = tree;
    r->ddecomp = ddecomp;
    r->kind = kind;
    r->mask = mask;
    r->HybridNuTracer = HybridNuTracer;
    r->alloc_father = alloc_father;
    r->EmergencyOutputDir = dir;
    if(act && act->ActiveParticle) { /* (timestep.c:287-289 keeps subact alive until force_tree_free) */
        r->ActiveParticle = act->ActiveParticle;
        r->NumActiveParticle = act->NumActiveParticle;
    }
    memset(tree, 0, sizeof(*tree));
    tree->tree_allocated_flag = 1;
    tree->mask = mask;
    tree->BoxSize = PartManager->BoxSize;
    tree->firstnode = PartManager->MaxPart; /* forcetree.c:1388 */
    tree->lastnode = tree->firstnode + 1;
    tree->numnodes = 1;
    tree->NumParticles = r->ActiveParticle ? r->NumActiveParticle : PartManager->NumPart; /* (an upper bound until the device tree exists) */
    tree->Nodes_base = &r->root;
    tree->Nodes = tree->Nodes_base - tree->firstnode;
    tree->NTopLeaves = ddecomp->NTopLeaves;
    tree->TopLeaves = ddecomp->TopLeaves;
    MPI_Comm_rank(MPI_COMM_WORLD, &tree->ThisTask);
    r->root.len = PartManager->BoxSize * 1.001; /* force_create_node_for_topnode, forcetree.c:560-575 */
    r->root.center[0] = r->root.center[1] = r->root.center[2] = 0.5 * PartManager->BoxSize;
    r->root.sibling = r->root.father = -1;
    mpg_shim_set_domain(ddecomp);
    return r;
}

void force_tree_full(ForceTree *tree, DomainDecomp *ddecomp, const int HybridNuTracer, const char *EmergencyOutputDir)
{
    if(force_tree_allocated(tree))
        force_tree_free(tree);
    walltime_measure("/Misc");
    /* forcetree.c:118-127: all types, neutrinos left out while they are tracers; moments wanted; a tree of all particles */
    defer(tree, ddecomp, MPG_TREE_FULL, HybridNuTracer ? GASMASK + DMMASK + STARMASK + BHMASK : ALLMASK, NULL, HybridNuTracer, 1, EmergencyOutputDir);
    tree->moments_computed_flag = 1;
    tree->hmax_computed_flag = 1;
    tree->full_particle_tree_flag = 1;
}

void force_tree_active_moments(ForceTree *tree, DomainDecomp *ddecomp, const ActiveParticles *act, const int HybridNuTracer, const int alloc_father,
                               const char *EmergencyOutputDir)
{
    if(force_tree_allocated(tree))
        force_tree_free(tree);
    walltime_measure("/Misc");
    defer(tree, ddecomp, MPG_TREE_ACTIVE, HybridNuTracer ? GASMASK + DMMASK + STARMASK + BHMASK : ALLMASK, act, HybridNuTracer, alloc_father,
          EmergencyOutputDir);
    tree->moments_computed_flag = 1;
    tree->hmax_computed_flag = 1;
    if(!act->ActiveParticle) /* forcetree.c:146-148 */
        tree->full_particle_tree_flag = 1;
}

void force_tree_rebuild_mask(ForceTree *tree, DomainDecomp *ddecomp, int mask, const char *EmergencyOutputDir)
{
    if(mask != (GASMASK | BHMASK)) { /* a CPU module's own tree: it walks it next */
        cpu_force_tree_rebuild_mask(tree, ddecomp, mask, EmergencyOutputDir);
        return;
    }
    message(0, "Tree construction for types: %d (deferred to the device loops).\n", mask);
    if(force_tree_allocated(tree))
        force_tree_free(tree);
    defer(tree, ddecomp, MPG_TREE_MASK, mask, NULL, 0, 1, EmergencyOutputDir); /* forcetree.c:162: no moments yet, father array for hmax */
}

/* run.c:477 (after density(), before hydro_force()), blackhole.c:292, density.c:704 (set_init_hsml: replaced by sph-hip.c) */
void force_tree_calc_moments(ForceTree *tree, DomainDecomp *ddecomp)
{
    struct mpg_deferred_tree *r = find(tree);
    if(!r || r->materialised) {
        cpu_force_tree_calc_moments(tree, ddecomp);
        return;
    }
    /* the device gas tree carries mass moments from its build and the hmax moments of the final smoothing lengths from the end of the
     * density loop (mpg_density / mpg_dist_density); the root's value is what run.c:481 prints */
    r->moments_wanted = 1;
    tree->moments_computed_flag = 1;
    tree->hmax_computed_flag = 1;
    mpg_tree_stats st;
    if(mpg_tree_get_stats(mpg_shim_engine(), &st) == 0) {
        r->root.mom.hmax = st.root_hmax;
        r->root.mom.mass = st.root_mass;
        r->root.mom.cofm[0] = st.root_cofm[0];
        r->root.mom.cofm[1] = st.root_cofm[1];
        r->root.mom.cofm[2] = st.root_cofm[2];
        tree->NumParticles = st.NumParticles;
    }
}

void force_tree_free(ForceTree *tree)
{
    struct mpg_deferred_tree *r = find(tree);
    if(r && !r->materialised) {
        /* nothing was allocated on the host; the device tree stays in the engine's buffers for the next build to reuse */
        memset(tree, 0, sizeof(*tree));
        memset(r, 0, sizeof(*r));
        return;
    }
    if(r)
        memset(r, 0, sizeof(*r));
    cpu_force_tree_free(tree);
}

/* The host tree a deferred constructor stood for, built now by forcetree.c (a tree that is not deferred, or already built: nothing).
 * Call it where the FIRST host module receives the tree, before that module allocates (the allocator is a stack). */
void mpg_shim_host_tree(ForceTree *tree)
{
    struct mpg_deferred_tree *r = find(tree);
    if(!r || r->materialised)
        return;
    const struct mpg_deferred_tree rec = *r;
    ForceTree t;
    memset(&t, 0, sizeof(t));
    if(rec.kind == MPG_TREE_FULL)
        cpu_force_tree_full(&t, rec.ddecomp, rec.HybridNuTracer, rec.EmergencyOutputDir);
    else if(rec.kind == MPG_TREE_ACTIVE) {
        ActiveParticles act = init_empty_active_particles(PartManager);
        if(rec.ActiveParticle) {
            act.ActiveParticle = (int *)rec.ActiveParticle;
            act.NumActiveParticle = rec.NumActiveParticle;
        }
        cpu_force_tree_active_moments(&t, rec.ddecomp, &act, rec.HybridNuTracer, rec.alloc_father, rec.EmergencyOutputDir);
    }
    else {
        cpu_force_tree_rebuild_mask(&t, rec.ddecomp, rec.mask, rec.EmergencyOutputDir);
        if(rec.moments_wanted) /* run.c:477 came by while the tree was deferred; P[].Hsml is final, so these are the same moments */
            cpu_force_tree_calc_moments(&t, rec.ddecomp);
    }
    *tree = t;
    r->materialised = 1;
}

/* treewalk.c:802, first line of treewalk_run: `mpg_shim_require_host_tree(tw->tree, tw->ev_label);` */
void mpg_shim_require_host_tree(const ForceTree *tree, const char *walk)
{
    if(mpg_shim_deferred_tree(tree))
        endrun(5, "mpgadget_hip: host tree walk %s on a tree that exists only on the device: call mpg_shim_host_tree(tree) where the tree is "
                  "handed to this module (run.c:607), before the module allocates\n", walk);
}
