/* oracle/oracle_tree.h -- node / tree types shared by the CPU oracle sources.  TEST INFRASTRUCTURE ONLY. */
#ifndef ORACLE_TREE_H
#define ORACLE_TREE_H
#include <stdint.h>

#define NMAXCHILD 8            /* forcetree.h:13 */
#define NODEFULL (1 << 16)     /* forcetree.h:14 */
#define PARTICLE_NODE_TYPE 0   /* forcetree.h:17-19 */
#define NODE_NODE_TYPE 1
#define PSEUDO_NODE_TYPE 2

/* partmanager.h:99 */
#define NEAREST(x, B) (((x) > 0.5 * (B)) ? ((x) - (B)) : (((x) < -0.5 * (B)) ? ((x) + (B)) : (x)))
#define DMAX(a, b) (((a) > (b)) ? (a) : (b))

/* struct NODE of forcetree.h:37-66, widened flags */
typedef struct {
    int sibling, father;
    double len, center[3];
    double cofm[3], mass, hmax;
    int suns[NMAXCHILD];
    int noccupied;
    int TopLevel, InternalTopLevel, DependsOnLocalMass, ChildType;
} onode;

typedef struct {
    int64_t npart;     /* particles offered (index space [0,npart)) */
    int64_t ninserted; /* particles actually in the tree (mask, garbage) */
    int64_t firstnode, lastnode, numnodes;
    onode *nodes_base, *nodes; /* nodes[firstnode] is the root */
    int *father;               /* per particle leaf node (forcetree.c:360) */
    double box;
    const double *pos;  /* N x 3 */
    const float *mass;  /* N (P.Mass is float, partmanager.h:15) */
    const int *type;    /* N or NULL (all type 1) */
    const double *hsml; /* N or NULL */
    const unsigned char *hydro_active; /* N or NULL: 1 = active (excluded from leaf hmax) */
    int moments_computed;
} otree;


#endif
