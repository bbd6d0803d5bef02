/* fof_oracle.c -- CPU restatement of the friends-of-friends labelling -- TEST INFRASTRUCTURE ONLY (see oracle.py).
 *
 * Follows libgadget/fof.c:
 *   fof_label_primary  :366-478, fofp_merge :480-540, fof_primary_ngbiter :543-579   every pair of primary-type particles with
 *                      r^2 <= LL^2 (the acceptance test of treewalk_visit_ngbiter, treewalk.c:984-991, NEAREST() per axis) is merged;
 *                      the label of a group is the smallest P[].ID in it.  The reference reaches that fixed point by iterating
 *                      locked merges over tree-walk neighbours; here the pairs come from a cell grid and a serial union-find.
 *   fof_label_secondary :1175-1327   the doubling search, literally: hsml = (float) 0.4 LL (or half the particle's Hsml for gas,
 *                      stars, black holes if that is larger); the nearest primary particle with r^2 <= hsml^2 gives its label;
 *                      nothing found and hsml < 4 LL: hsml *= 2 and search again; else the particle stays alone.
 * Particles that are garbage or swallowed (flags bits 0, 1) take no part.  Every particle starts as its own group (MinID = ID). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static double nearest(double x, double box) { return x > 0.5 * box ? x - box : (x < -0.5 * box ? x + box : x); } /* partmanager.h:99 */

static int find(int *head, int i)
{
    int r = i;
    while(head[r] != r)
        r = head[r];
    while(head[i] != r) {
        int t = head[i];
        head[i] = r;
        i = t;
    }
    return r;
}

static int cellof(double x, double box, int nc)
{
    int c = (int) floor(x / box * nc);
    if(c < 0)
        c += nc;
    if(c >= nc)
        c -= nc;
    if(c < 0)
        c = 0;
    if(c >= nc)
        c = nc - 1;
    return c;
}

int ofof_label(int64_t n, const double *pos, const uint8_t *type, const uint8_t *flags, const uint64_t *id, const double *hsml, double box,
               double LL, int primary_mask, int secondary_mask, uint64_t *label)
{
    int nc = (int) floor(box / LL);
    if(nc > 256)
        nc = 256;
    if(nc < 1)
        nc = 1;
    const double cs = box / nc; /* >= LL */
    int *first = (int *) malloc(sizeof(int) * (size_t) nc * nc * nc);
    int *next = (int *) malloc(sizeof(int) * (size_t) (n + 1));
    int *head = (int *) malloc(sizeof(int) * (size_t) (n + 1));
    for(int64_t c = 0; c < (int64_t) nc * nc * nc; c++)
        first[c] = -1;
    for(int64_t i = 0; i < n; i++) {
        label[i] = id[i];
        head[i] = (int) i;
        next[i] = -1;
    }
#define ISPRIM(i) (!(flags && (flags[i] & 3)) && ((1 << (type ? type[i] : 1)) & primary_mask))
    for(int64_t i = n - 1; i >= 0; i--) { /* (lists in increasing index order) */
        if(!ISPRIM(i))
            continue;
        const int64_t c = ((int64_t) cellof(pos[3 * i], box, nc) * nc + cellof(pos[3 * i + 1], box, nc)) * nc + cellof(pos[3 * i + 2], box, nc);
        next[i] = first[c];
        first[c] = (int) i;
    }
    /* primary linking */
    const double LL2 = LL * LL;
    const int reach = nc >= 3 ? 1 : 0; /* fewer than 3 cells per side: every cell is a neighbour of every cell */
    for(int64_t i = 0; i < n; i++) {
        if(!ISPRIM(i))
            continue;
        const int cx = cellof(pos[3 * i], box, nc), cy = cellof(pos[3 * i + 1], box, nc), cz = cellof(pos[3 * i + 2], box, nc);
        for(int ax = (reach ? -1 : 0); ax <= (reach ? 1 : nc - 1); ax++)
            for(int ay = (reach ? -1 : 0); ay <= (reach ? 1 : nc - 1); ay++)
                for(int az = (reach ? -1 : 0); az <= (reach ? 1 : nc - 1); az++) {
                    const int x = reach ? (cx + ax + nc) % nc : ax, y = reach ? (cy + ay + nc) % nc : ay, z = reach ? (cz + az + nc) % nc : az;
                    for(int j = first[((int64_t) x * nc + y) * nc + z]; j >= 0; j = next[j]) {
                        if(j <= i)
                            continue;
                        double r2 = 0;
                        for(int d = 0; d < 3; d++) {
                            const double dd = nearest(pos[3 * i + d] - pos[3 * j + d], box);
                            r2 += dd * dd;
                        }
                        if(r2 > LL2)
                            continue;
                        const int a = find(head, (int) i), b = find(head, j);
                        if(a != b) {
                            if(a < b)
                                head[b] = a;
                            else
                                head[a] = b;
                        }
                    }
                }
    }
    /* MinID of every group, handed to its members */
    for(int64_t i = 0; i < n; i++)
        if(ISPRIM(i)) {
            const int r = find(head, (int) i);
            if(id[i] < label[r])
                label[r] = id[i];
        }
    for(int64_t i = 0; i < n; i++)
        if(ISPRIM(i))
            label[i] = label[find(head, (int) i)];
    /* secondary: nearest primary particle, doubling search */
    for(int64_t i = 0; i < n; i++) {
        const int t = type ? type[i] : 1;
        if((flags && (flags[i] & 3)) || !((1 << t) & secondary_mask))
            continue;
        float h = (float) (0.4 * LL);
        if(hsml && (t == 0 || t == 4 || t == 5) && h < 0.5 * hsml[i])
            h = (float) (0.5 * hsml[i]);
        for(;;) {
            const double H = h, H2 = H * H;
            double best = 1e29; /* LARGE */
            int64_t bj = -1;
            const int span = (int) ceil(H / cs);
            const int full = 2 * span + 1 >= nc;
            const int cx = cellof(pos[3 * i], box, nc), cy = cellof(pos[3 * i + 1], box, nc), cz = cellof(pos[3 * i + 2], box, nc);
            for(int ax = (full ? 0 : -span); ax <= (full ? nc - 1 : span); ax++)
                for(int ay = (full ? 0 : -span); ay <= (full ? nc - 1 : span); ay++)
                    for(int az = (full ? 0 : -span); az <= (full ? nc - 1 : span); az++) {
                        const int x = full ? ax : ((cx + ax) % nc + nc) % nc, y = full ? ay : ((cy + ay) % nc + nc) % nc,
                                  z = full ? az : ((cz + az) % nc + nc) % nc;
                        for(int j = first[((int64_t) x * nc + y) * nc + z]; j >= 0; j = next[j]) {
                            double r2 = 0;
                            for(int d = 0; d < 3; d++) {
                                const double dd = nearest(pos[3 * i + d] - pos[3 * j + d], box);
                                r2 += dd * dd;
                            }
                            if(r2 > H2)
                                continue;
                            const double r = sqrt(r2);
                            if(r < best) {
                                best = r;
                                bj = j;
                            }
                        }
                    }
            if(bj >= 0) {
                label[i] = label[bj];
                break;
            }
            if(h < 4 * LL)
                h *= 2.0f;
            else
                break;
        }
    }
    free(head);
    free(next);
    free(first);
    return 0;
}
