/* oracle/gravtree_oracle.c
 *
 * TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, fp64) of the reference
 * TreePM short-range path of MP-Gadget, written from the algorithm description
 * of the reference sources; it is the checker for the HIP engine and the
 * "port" CPU baseline of bench.py.  Nothing in the product path
 * (mp-gadget_amd/) may import, link or call this file.
 *
 * Parity pinning: the full reference path cannot be built in this image
 * (gravity.h -> petapm.h needs <pfft.h>, omega_nu_single.h needs GSL; both are
 * absent and may not be stubbed), so this oracle is pinned by
 *   (a) the reference's own known answers: libgadget/tests/test_gravity.c
 *       direct-sum bounds (:146-160, :259-260), test_forcetree.c invariants,
 *   (b) reference outputs recorded by the survey probe:
 *       mean |FullTreeGravAccel| and Ninteractions/N on the S-grid set,
 *   (c) oracle/_ref: the leaf files that DO compile stand-alone
 *       (shortrange-kernel.c table, densitykernel.c, utils/peano.c).
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference/libgadget).
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

/* ---------------- tree build: forcetree.c:278-520, 654-687 ---------------- */

static int get_subnode(const onode *node, const double *p) /* forcetree.c:278-284 */
{
    return (p[0] > node->center[0]) + ((p[1] > node->center[1]) << 1) + ((p[2] > node->center[2]) << 2);
}

static void init_internal_node(onode *nf, const onode *parent, int subnode) /* forcetree.c:302-327 */
{
    const double lenhalf = 0.25 * parent->len;
    nf->len = 0.5 * parent->len;
    nf->sibling = -10;
    nf->father = -10;
    nf->TopLevel = nf->InternalTopLevel = nf->DependsOnLocalMass = 0;
    nf->ChildType = PARTICLE_NODE_TYPE;
    for(int j = 0; j < 3; j++) {
        const int sign = (subnode & (1 << j)) ? 1 : -1;
        nf->center[j] = parent->center[j] + sign * lenhalf;
    }
    for(int j = 0; j < NMAXCHILD; j++)
        nf->suns[j] = -1;
    nf->noccupied = 0;
    nf->cofm[0] = nf->cofm[1] = nf->cofm[2] = 0;
    nf->mass = 0;
    nf->hmax = 0;
}

static void add_particle_moment_to_node(otree *t, onode *pn, int p) /* forcetree.c:947-966 */
{
    const double m = t->mass[p];
    pn->mass += m;
    for(int k = 0; k < 3; k++)
        pn->cofm[k] += m * t->pos[3 * (size_t)p + k];
    const int ty = t->type ? t->type[p] : 1;
    if((ty == 0 || ty == 5) && t->hsml && !(t->hydro_active && t->hydro_active[p])) {
        for(int j = 0; j < 3; j++)
            pn->hmax = DMAX(pn->hmax, fabs(t->pos[3 * (size_t)p + j] - pn->center[j]) + t->hsml[p] - pn->len / 2.);
    }
}

static void modify_internal_node(otree *t, int parent, int subnode, int p) /* forcetree.c:357-365 */
{
    if(t->father)
        t->father[p] = parent;
    t->nodes[parent].suns[subnode] = p;
    add_particle_moment_to_node(t, &t->nodes[parent], p);
}

/* forcetree.c:370-477. Serial restatement: the node cache just hands out 8 consecutive nodes. */
static int create_new_node_layer(otree *t, int firstparent, int p_toplace, int64_t *nnext)
{
    int parent = firstparent;
    for(;;) {
        onode *nprnt = &t->nodes[parent];
        int newsuns[NMAXCHILD], oldsuns[NMAXCHILD];
        memcpy(oldsuns, nprnt->suns, sizeof(oldsuns));
        if(*nnext + 8 >= t->lastnode)
            return 1; /* pool exhausted: > NMAXCHILD coincident particles or too small a pool */
        newsuns[0] = (int)*nnext;
        *nnext += 8;
        for(int i = 0; i < 8; i++) {
            newsuns[i] = newsuns[0] + i;
            onode *nf = &t->nodes[newsuns[i]];
            init_internal_node(nf, nprnt, i);
            nf->father = parent;
        }
        for(int i = 0; i < NMAXCHILD; i++) {
            const int subnode = get_subnode(nprnt, &t->pos[3 * (size_t)oldsuns[i]]);
            const int child = newsuns[subnode];
            onode *nchild = &t->nodes[child];
            modify_internal_node(t, child, nchild->noccupied, oldsuns[i]);
            nchild->noccupied++;
        }
        memcpy(nprnt->suns, newsuns, sizeof(newsuns));
        for(int i = 0; i < 7; i++)
            t->nodes[nprnt->suns[i]].sibling = nprnt->suns[i + 1];
        t->nodes[nprnt->suns[7]].sibling = nprnt->sibling;
        nprnt->cofm[0] = nprnt->cofm[1] = nprnt->cofm[2] = 0;
        nprnt->mass = 0;
        nprnt->hmax = 0;

        const int subnode = get_subnode(nprnt, &t->pos[3 * (size_t)p_toplace]);
        const int child = nprnt->suns[subnode];
        onode *nchild = &t->nodes[child];
        if(nchild->noccupied < NMAXCHILD) {
            modify_internal_node(t, child, nchild->noccupied, p_toplace);
            nchild->noccupied++;
            break;
        }
        nchild->ChildType = NODE_NODE_TYPE;
        nchild->noccupied = NODEFULL;
        parent = child;
    }
    t->nodes[firstparent].ChildType = NODE_NODE_TYPE;
    t->nodes[firstparent].noccupied = NODEFULL;
    return 0;
}

static int add_particle_to_tree(otree *t, int i, int cur_start, int64_t *nnext) /* forcetree.c:481-520 */
{
    int cur = cur_start, child;
    do {
        const int nocc = t->nodes[cur].noccupied;
        if(nocc < NODEFULL)
            break;
        const int subnode = get_subnode(&t->nodes[cur], &t->pos[3 * (size_t)i]);
        child = t->nodes[cur].suns[subnode];
        cur = child;
    } while(child >= t->firstnode);

    const int nocc = t->nodes[cur].noccupied;
    t->nodes[cur].noccupied++;
    if(nocc < NMAXCHILD)
        modify_internal_node(t, cur, nocc, i);
    else if(nocc < NODEFULL) {
        if(create_new_node_layer(t, cur, i, nnext))
            return -1;
    }
    else
        return -2;
    return cur;
}

/* forcetree.c:196-270 (+ :654-687 root, :727-860 insertion loop, single thread, trivial domain:
 * one TopNode with Daughter=-1, so the root is the only top-level node). */
otree *ot_build(int64_t N, const double *pos, const float *mass, const int *type, const double *hsml,
                const unsigned char *hydro_active, int mask, double box, double alloc_factor, int alloc_father)
{
    otree *t = (otree *)calloc(1, sizeof(otree));
    t->npart = N;
    t->pos = pos;
    t->mass = mass;
    t->type = type;
    t->hsml = hsml;
    t->hydro_active = hydro_active;
    t->box = box;
    for(;;) {
        int64_t maxnodes = (int64_t)(alloc_factor * N) + 1 + 16;
        t->firstnode = N;
        t->lastnode = N + maxnodes;
        t->nodes_base = (onode *)malloc((maxnodes + 1) * sizeof(onode));
        t->nodes = t->nodes_base - t->firstnode;
        t->father = alloc_father ? (int *)malloc(sizeof(int) * (N > 0 ? N : 1)) : NULL;
        if(t->father)
            memset(t->father, -1, sizeof(int) * (N > 0 ? N : 1));
        /* root: forcetree.c:657-678 */
        int64_t nnext = t->firstnode;
        onode *root = &t->nodes[nnext];
        root->len = box * 1.001;
        for(int i = 0; i < 3; i++)
            root->center[i] = box / 2.;
        for(int i = 0; i < NMAXCHILD; i++)
            root->suns[i] = -1;
        root->noccupied = 0;
        root->father = -1;
        root->sibling = -1;
        root->TopLevel = 1;
        root->InternalTopLevel = 0;
        root->DependsOnLocalMass = 0;
        root->ChildType = PARTICLE_NODE_TYPE;
        root->cofm[0] = root->cofm[1] = root->cofm[2] = 0;
        root->mass = 0;
        root->hmax = 0;
        nnext++;
        int failed = 0;
        int64_t nins = 0;
        for(int64_t i = 0; i < N; i++) { /* forcetree.c:792-832 */
            const int ty = type ? type[i] : 1;
            if(!((1 << ty) & mask))
                continue;
            if(mass[i] <= 0) {
                fprintf(stderr, "oracle: zero mass particle %ld\n", (long)i);
                abort();
            }
            nins++;
            if(add_particle_to_tree(t, (int)i, (int)t->firstnode, &nnext) < 0) {
                failed = 1;
                break;
            }
        }
        if(!failed) {
            t->ninserted = nins;
            t->numnodes = nnext - t->firstnode;
            break;
        }
        /* forcetree.c:215-229: retry with a 1.15x larger pool */
        free(t->nodes_base);
        free(t->father);
        alloc_factor *= 1.15;
        if(alloc_factor > 30.0) {
            fprintf(stderr, "oracle: TreeAllocFactor too large (coincident particles?)\n");
            abort();
        }
    }
    t->moments_computed = 0;
    return t;
}

static int force_get_sibling(int sib, int j, const int *suns) /* forcetree.c:969-982 */
{
    int nextsib = sib;
    for(int jj = j + 1; jj < 8; jj++)
        if(suns[jj] >= 0) {
            nextsib = suns[jj];
            break;
        }
    return nextsib;
}

static void force_update_particle_node(otree *t, int no) /* forcetree.c:985-1004 */
{
    onode *n = &t->nodes[no];
    if(n->mass > 0) {
        for(int j = 0; j < 3; j++)
            n->cofm[j] /= n->mass;
    }
    else
        for(int j = 0; j < 3; j++)
            n->cofm[j] = n->center[j];
}

static void force_update_node_recursive(otree *t, int no, int sib) /* forcetree.c:1017-1104 */
{
    int *suns = t->nodes[no].suns;
    int jj = 0;
    for(int j = 0; j < 8; j++, jj++) {
        while(jj < 8 && !t->nodes[suns[jj]].TopLevel && t->nodes[suns[jj]].ChildType == PARTICLE_NODE_TYPE &&
              t->nodes[suns[jj]].noccupied == 0)
            jj++;
        suns[j] = (jj < 8) ? suns[jj] : -1;
    }
    for(int j = 0; j < 8; j++) {
        const int p = suns[j];
        if(p < 0)
            continue;
        const int nextsib = force_get_sibling(sib, j, suns);
        t->nodes[p].sibling = nextsib;
        if(t->nodes[p].ChildType == PARTICLE_NODE_TYPE)
            force_update_particle_node(t, p);
        if(t->nodes[p].ChildType == NODE_NODE_TYPE)
            force_update_node_recursive(t, p, nextsib);
    }
    onode *n = &t->nodes[no];
    /* n->mass / cofm were zeroed when the node became internal (forcetree.c:449) */
    for(int j = 0; j < 8; j++) {
        const int p = suns[j];
        if(p < 0)
            continue;
        const onode *c = &t->nodes[p];
        n->mass += c->mass;
        n->cofm[0] += c->mass * c->cofm[0];
        n->cofm[1] += c->mass * c->cofm[1];
        n->cofm[2] += c->mass * c->cofm[2];
        if(c->hmax > n->hmax)
            n->hmax = c->hmax;
    }
    if(n->mass > 0) {
        n->cofm[0] /= n->mass;
        n->cofm[1] /= n->mass;
        n->cofm[2] /= n->mass;
    }
}

/* forcetree.c:170-183 with force_update_node_parallel :1119-1143 for the single (root) top leaf.
 * Like the reference it must run exactly once per tree (leaf cofm sums are normalised in place). */
void ot_calc_moments(otree *t)
{
    const int root = (int)t->firstnode;
    if(t->moments_computed) {
        fprintf(stderr, "oracle: ot_calc_moments called twice on one tree\n");
        abort();
    }
    t->nodes[root].DependsOnLocalMass = 1;
    if(t->nodes[root].ChildType == NODE_NODE_TYPE)
        force_update_node_recursive(t, root, t->nodes[root].sibling);
    else
        force_update_particle_node(t, root);
    t->moments_computed = 1;
}

void ot_free(otree *t)
{
    if(!t)
        return;
    free(t->nodes_base);
    free(t->father);
    free(t);
}

int64_t ot_numnodes(const otree *t) { return t->numnodes; }
int64_t ot_firstnode(const otree *t) { return t->firstnode; }
int64_t ot_ninserted(const otree *t) { return t->ninserted; }
const int *ot_father(const otree *t) { return t->father; }

/* Export the node table for structural comparison (test_forcetree.c-style checks and the
 * comparison with the HIP tree builder).  Arrays have numnodes entries; entries of pruned /
 * never-linked nodes are flagged live=0.  A node is "live" if reachable from the root. */
static void mark_live(const otree *t, int no, int level, int *live, int *levels)
{
    live[no - t->firstnode] = 1;
    levels[no - t->firstnode] = level;
    const onode *n = &t->nodes[no];
    if(n->ChildType == NODE_NODE_TYPE)
        for(int j = 0; j < 8; j++)
            if(n->suns[j] >= 0)
                mark_live(t, n->suns[j], level + 1, live, levels);
}

void ot_export(const otree *t, int *live, int *level, double *center, double *len, double *cofm, double *mass, double *hmax,
               int *childtype, int *noccupied, int *sibling, int *father, int *suns)
{
    memset(live, 0, sizeof(int) * t->numnodes);
    memset(level, 0, sizeof(int) * t->numnodes);
    mark_live(t, (int)t->firstnode, 0, live, level);
    for(int64_t i = 0; i < t->numnodes; i++) {
        const onode *n = &t->nodes_base[i];
        for(int k = 0; k < 3; k++) {
            center[3 * i + k] = n->center[k];
            cofm[3 * i + k] = n->cofm[k];
        }
        len[i] = n->len;
        mass[i] = n->mass;
        hmax[i] = n->hmax;
        childtype[i] = n->ChildType;
        noccupied[i] = n->noccupied;
        sibling[i] = n->sibling;
        father[i] = n->father;
        for(int k = 0; k < 8; k++)
            suns[8 * i + k] = n->suns[k];
    }
}

/* --------------- short-range window: gravity.c:20-66 ---------------------- */

#define NTAB 512
static float shortrange_table[NTAB], shortrange_table_potential[NTAB];
static double shortrange_dx = 0.02935420743639786; /* shortrange_force_kernels[1][0] */

/* gravity.c:22-51.  `table` = the 512x5 calibrated data (shortrange-kernel.c), type 0 exact / 1 erfc */
int og_fill_ntab(const double *table, int type, double Asmth)
{
    if(type == 0 && Asmth != 1.5)
        return -1; /* gravity.c:25-29 is a fatal error */
    shortrange_dx = table[5 * 1 + 0];
    for(int i = 0; i < NTAB; i++) {
        const double u = table[5 * i + 0] * 0.5 / Asmth;
        if(type == 0) {
            shortrange_table[i] = table[5 * i + 2];
            shortrange_table_potential[i] = table[5 * i + 1];
        }
        else {
            shortrange_table[i] = erfc(u) + 2.0 * u / sqrt(M_PI) * exp(-u * u);
            shortrange_table_potential[i] = erfc(u);
        }
    }
    return 0;
}

void og_get_ntab(float *force, float *pot)
{
    memcpy(force, shortrange_table, sizeof(shortrange_table));
    memcpy(pot, shortrange_table_potential, sizeof(shortrange_table_potential));
}

static inline int grav_apply_short_range_window(double r, double *fac, double *pot, const double cellsize) /* gravity.c:54-66 */
{
    const double dx = shortrange_dx;
    double i = (r / cellsize / dx);
    size_t tabindex = floor(i);
    if(tabindex >= NTAB - 1)
        return 1;
    *fac *= (tabindex + 1 - i) * shortrange_table[tabindex] + (i - tabindex) * shortrange_table[tabindex + 1];
    *pot *= (tabindex + 1 - i) * shortrange_table_potential[tabindex] + (i - tabindex) * shortrange_table_potential[tabindex + 1];
    return 0;
}

/* --------------- short-range tree walk: gravshort-tree.c ------------------- */

typedef struct {
    double ErrTolForceAcc, BHOpeningAngle, MaxBHOpeningAngle; /* gravity.h:9-22 */
    int TreeUseBH;
    double Rcut;      /* already TreeRcut*Asmth*cellsize (gravshort-tree.c:102) */
    double h;         /* FORCE_SOFTENING() = 2.8*GravitySoftening (gravshort-tree.c:37-41) */
    double cellsize;  /* BoxSize/Nmesh */
    double G;
    double cbrtrho0;
} ograv_params;

typedef struct {
    double Acc[3], Potential;
} oresult;

static inline void apply_accn_to_output(oresult *out, const double dx[3], const double r2, const double mass,
                                        const double cellsize, const double h) /* gravshort-tree.c:158-193 */
{
    const double r = sqrt(r2);
    double fac = mass / (r2 * r);
    double facpot = -mass / r;
    if(r2 < h * h) {
        double wp;
        const double h3_inv = 1.0 / h / h / h;
        const double u = r / h;
        if(u < 0.5) {
            fac = mass * h3_inv * (10.666666666667 + u * u * (32.0 * u - 38.4));
            wp = -2.8 + u * u * (5.333333333333 + u * u * (6.4 * u - 9.6));
        }
        else {
            fac = mass * h3_inv * (21.333333333333 - 48.0 * u + 38.4 * u * u - 10.666666666667 * u * u * u - 0.066666666667 / (u * u * u));
            wp = -3.2 + 0.066666666667 / u + u * u * (10.666666666667 + u * (-16.0 + u * (9.6 - 2.133333333333 * u)));
        }
        facpot = mass / h * wp;
    }
    if(0 == grav_apply_short_range_window(r, &fac, &facpot, cellsize)) {
        for(int i = 0; i < 3; i++)
            out->Acc[i] += dx[i] * fac;
        out->Potential += facpot;
    }
}

static inline int shall_we_discard_node(const double len, const double r2, const double center[3], const double inpos[3],
                                        const double Box, const double rcut, const double rcut2) /* gravshort-tree.c:198-215 */
{
    if(r2 > rcut2) {
        const double eff_dist = rcut + 0.5 * len;
        for(int i = 0; i < 3; i++)
            if(fabs(NEAREST(center[i] - inpos[i], Box)) > eff_dist)
                return 1;
    }
    return 0;
}

static inline int shall_we_open_node(const double len, const double mass, const double r2, const double center[3],
                                     const double inpos[3], const double Box, const double aold, const int TreeUseBH,
                                     const double BHOpeningAngle2) /* gravshort-tree.c:220-241 */
{
    if((TreeUseBH == 0) && (mass * len * len > r2 * r2 * aold))
        return 1;
    double bhangle = len * len / r2;
    if(bhangle > BHOpeningAngle2)
        return 1;
    const double inside = 0.6 * len;
    if(fabs(NEAREST(center[0] - inpos[0], Box)) < inside && fabs(NEAREST(center[1] - inpos[1], Box)) < inside &&
       fabs(NEAREST(center[2] - inpos[2], Box)) < inside)
        return 1;
    return 0;
}

/* force_treeev_shortrange, gravshort-tree.c:253-379, PRIMARY mode, NodeList = {root,-1,..}.
 * ngblist must hold at least ninserted ints. Returns #particle interactions (treewalk.c:904-912). */
static int64_t force_treeev_shortrange(const otree *tree, const ograv_params *par, const double *inpos, const double OldAcc,
                                       oresult *output, int *ngblist, int64_t *nnodes_visited, int64_t *nnodes_used)
{
    const double Box = tree->box;
    const double cellsize = par->cellsize;
    const double rcut = par->Rcut;
    const double rcut2 = rcut * rcut;
    const double aold = par->ErrTolForceAcc * OldAcc;
    const int TreeUseBH = par->TreeUseBH;
    double BHOpeningAngle2 = par->BHOpeningAngle * par->BHOpeningAngle;
    if(TreeUseBH == 0)
        BHOpeningAngle2 = par->MaxBHOpeningAngle * par->MaxBHOpeningAngle;
    int numcand = 0;
    int no = (int)tree->firstnode;
    int64_t nvis = 0, nused = 0;
    while(no >= 0) {
        const onode *nop = &tree->nodes[no];
        nvis++;
        double dx[3];
        for(int i = 0; i < 3; i++)
            dx[i] = NEAREST(nop->cofm[i] - inpos[i], Box);
        const double r2 = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];
        if(shall_we_discard_node(nop->len, r2, nop->center, inpos, Box, rcut, rcut2)) {
            no = nop->sibling;
            continue;
        }
        const int open_node = shall_we_open_node(nop->len, nop->mass, r2, nop->center, inpos, Box, aold, TreeUseBH, BHOpeningAngle2);
        if(!open_node) {
            no = nop->sibling;
            apply_accn_to_output(output, dx, r2, nop->mass, cellsize, par->h);
            nused++;
            continue;
        }
        if(nop->ChildType == PARTICLE_NODE_TYPE) {
            for(int i = 0; i < nop->noccupied; i++)
                ngblist[numcand++] = nop->suns[i];
            no = nop->sibling;
        }
        else if(nop->ChildType == PSEUDO_NODE_TYPE)
            no = nop->sibling;
        else
            no = nop->suns[0];
    }
    for(int i = 0; i < numcand; i++) {
        const int pp = ngblist[i];
        double dx[3];
        for(int j = 0; j < 3; j++)
            dx[j] = NEAREST(tree->pos[3 * (size_t)pp + j] - inpos[j], Box);
        const double r2 = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];
        apply_accn_to_output(output, dx, r2, tree->mass[pp], cellsize, par->h);
    }
    if(nnodes_visited)
        *nnodes_visited += nvis;
    if(nnodes_used)
        *nnodes_used += nused;
    return numcand;
}

/* grav_short_tree, gravshort-tree.c:96-154, with fill/reduce/postprocess of gravshort.h:47-96.
 *  active:   list of target particle indices or NULL (= all npart)
 *  oldacc:   per particle |FullTreeGravAccel + GravPM| / G   (grav_get_abs_accel, gravshort.h:70-80)
 *  accel:    [npart][3] output, multiplied by G (postprocess)
 *  pot:      if non-NULL (full particle tree): P.Potential <- result; += m/(h/2.8); -= 2.8372975 m^{2/3} cbrtrho0; *= G
 *  counters: [0]=sum particle interactions (Ninteractions), [1]=nodes visited, [2]=nodes used unopened */
void og_grav_short_tree(const otree *tree, const ograv_params *par, int64_t nactive, const int *active, const double *oldacc,
                        double *accel, double *pot, int64_t *counters, int64_t *ninter_per_particle)
{
    const int64_t nact = active ? nactive : tree->npart;
    int64_t s_pp = 0, s_vis = 0, s_used = 0;
#pragma omp parallel reduction(+ : s_pp, s_vis, s_used)
    {
        int *ngblist = (int *)malloc(sizeof(int) * (tree->ninserted > 0 ? tree->ninserted : 1));
#pragma omp for schedule(dynamic, 64)
        for(int64_t k = 0; k < nact; k++) {
            const int i = active ? active[k] : (int)k;
            oresult out = {{0, 0, 0}, 0};
            int64_t nv = 0, nu = 0;
            const int64_t npp = force_treeev_shortrange(tree, par, &tree->pos[3 * (size_t)i], oldacc ? oldacc[i] : 0.0, &out, ngblist, &nv, &nu);
            s_pp += npp;
            s_vis += nv;
            s_used += nu;
            if(ninter_per_particle)
                ninter_per_particle[i] = npp;
            /* reduce (assign, PRIMARY) + postprocess */
            accel[3 * (size_t)i + 0] = out.Acc[0] * par->G;
            accel[3 * (size_t)i + 1] = out.Acc[1] * par->G;
            accel[3 * (size_t)i + 2] = out.Acc[2] * par->G;
            if(pot) {
                double p = out.Potential;
                const double m = tree->mass[i];
                p += m / (par->h / 2.8);
                p -= 2.8372975 * pow(m, 2.0 / 3) * par->cbrtrho0;
                p *= par->G;
                pot[i] = p;
            }
        }
        free(ngblist);
    }
    if(counters) {
        counters[0] = s_pp;
        counters[1] = s_vis;
        counters[2] = s_used;
    }
}

/* grav_short_pair, gravshort-pair.c:21-120: exact pairwise short-range force inside the Rcut SPHERE
 * (asymmetric ngbiter with Hsml=Rcut: r2 <= Rcut^2 is accepted, treewalk.c:976-991).
 * O(N^2) brute force over the periodic minimum image; for small N only. */
void og_grav_short_pair(int64_t N, const double *pos, const float *mass, double Box, const ograv_params *par, double Rcut_abs,
                        double *accel)
{
#pragma omp parallel for schedule(dynamic, 16)
    for(int64_t i = 0; i < N; i++) {
        oresult out = {{0, 0, 0}, 0};
        for(int64_t j = 0; j < N; j++) {
            double dist[3];
            for(int d = 0; d < 3; d++)
                dist[d] = NEAREST(pos[3 * i + d] - pos[3 * j + d], Box); /* I.Pos - P[other].Pos, treewalk.c:968-975 */
            const double r2 = dist[0] * dist[0] + dist[1] * dist[1] + dist[2] * dist[2];
            if(r2 > Rcut_abs * Rcut_abs)
                continue;
            const double r = sqrt(r2);
            const double m = mass[j];
            const double h = par->h;
            double fac, potv;
            if(r >= h) {
                fac = m / (r2 * r);
                potv = -m / r;
            }
            else {
                const double h_inv = 1.0 / h, h3_inv = h_inv * h_inv * h_inv, u = r * h_inv;
                double wp;
                if(u < 0.5) {
                    fac = m * h3_inv * (10.666666666667 + u * u * (32.0 * u - 38.4));
                    wp = -2.8 + u * u * (5.333333333333 + u * u * (6.4 * u - 9.6));
                }
                else {
                    fac = m * h3_inv * (21.333333333333 - 48.0 * u + 38.4 * u * u - 10.666666666667 * u * u * u - 0.066666666667 / (u * u * u));
                    wp = -3.2 + 0.066666666667 / u + u * u * (10.666666666667 + u * (-16.0 + u * (9.6 - 2.133333333333 * u)));
                }
                potv = m * h_inv * wp;
            }
            if(grav_apply_short_range_window(r, &fac, &potv, par->cellsize) == 0)
                for(int d = 0; d < 3; d++)
                    out.Acc[d] += -dist[d] * fac;
        }
        for(int d = 0; d < 3; d++)
            accel[3 * i + d] = out.Acc[d] * par->G;
    }
}

/* Direct Newtonian(+spline) summation over the 27 nearest periodic images; the independent check
 * of libgadget/tests/test_gravity.c:38-71 (grav_force) and :125-144 (force_direct). */
void og_force_direct(int64_t N, const double *pos, const float *mass, double Box, double h, double G, double *accn)
{
    memset(accn, 0, sizeof(double) * 3 * N);
#pragma omp parallel for schedule(dynamic, 16)
    for(int64_t i = 0; i < N; i++) {
        double a[3] = {0, 0, 0};
        for(int xx = -1; xx <= 1; xx++)
            for(int yy = -1; yy <= 1; yy++)
                for(int zz = -1; zz <= 1; zz++) {
                    const double offset[3] = {Box * xx, Box * yy, Box * zz};
                    for(int64_t j = 0; j < N; j++) {
                        if(j == i)
                            continue; /* the reference only sums pairs i<j (no self / self-image term) */
                        double dist[3], r2 = 0;
                        for(int d = 0; d < 3; d++) {
                            dist[d] = offset[d] + pos[3 * i + d] - pos[3 * j + d];
                            r2 += dist[d] * dist[d];
                        }
                        const double r = sqrt(r2);
                        double fac = 1 / (r2 * r);
                        if(r < h) {
                            const double h_inv = 1.0 / h, h3_inv = h_inv * h_inv * h_inv, u = r * h_inv;
                            if(u < 0.5)
                                fac = 1. * h3_inv * (10.666666666667 + u * u * (32.0 * u - 38.4));
                            else
                                fac = 1. * h3_inv * (21.333333333333 - 48.0 * u + 38.4 * u * u - 10.666666666667 * u * u * u - 0.066666666667 / (u * u * u));
                        }
                        for(int d = 0; d < 3; d++)
                            a[d] += -dist[d] * fac * G * mass[j];
                    }
                }
        for(int d = 0; d < 3; d++)
            accn[3 * i + d] = a[d];
    }
}

int og_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
