/* test_shim_epoch.c -- shim/mpg_shim_epoch.h (when does mpg_shim_sync declare a new particle-table epoch?) on a table of 160-byte records
 * laid out as struct particle_data (partmanager.h:9-71: Pos at 0, ID at 136).  Plain C, no reference tree, no GPU; run by
 * tests/test_abi.py::test_shim_epoch_bookkeeping.  The sequence is run.c's: the calls of one step see one epoch; drift, exchange, a
 * reorder inside one Ti_Current, a changed NumPart, a moved table and an explicit mpg_shim_particles_changed() each open a new one. */
#include <stdio.h>
#include <stdlib.h>
#include "mpg_shim_epoch.h"

struct rec {
    double Pos[3];
    char pad0[112];
    uint64_t ID;
    char pad1[16];
};
_Static_assert(sizeof(struct rec) == 160 && offsetof(struct rec, ID) == 136, "record layout");

static struct mpg_table_key K = MPG_TABLE_KEY_INIT;
static int fails;

/* what mpg_shim_sync does with the table (one rank) */
static int64_t sync_(int64_t ti, const struct rec *P, int64_t n)
{
    const uint64_t s = mpg_table_sample_hash(P, sizeof(struct rec), n, offsetof(struct rec, ID), offsetof(struct rec, Pos));
    if(mpg_table_key_differs(&K, ti, P, n, s))
        mpg_table_key_take(&K, ti, P, n, s);
    return K.epoch;
}
#define EXPECT(cond, what)                                     \
    do {                                                       \
        if(!(cond)) {                                          \
            printf("FAIL %s (epoch %lld)\n", what, (long long)K.epoch); \
            fails++;                                           \
        }                                                      \
    } while(0)

int main(void)
{
    const int64_t n = 10000;
    struct rec *P = calloc(n + 1, sizeof(struct rec)), *Q = calloc(n + 1, sizeof(struct rec));
    for(int64_t i = 0; i < n; i++) {
        P[i].ID = 1000 + i;
        for(int k = 0; k < 3; k++)
            P[i].Pos[k] = 0.001 * i + k;
    }
    int64_t e = sync_(5, P, n);
    EXPECT(e == 1, "the first call opens epoch 1");
    EXPECT(sync_(5, P, n) == e && sync_(5, P, n) == e, "density(), hydro_force(), gravpm_force(), grav_short_tree() of one step share an epoch");
    EXPECT(sync_(6, P, n) == e + 1, "a new Ti_Current is a new epoch");
    e = K.epoch;
    /* drift inside the same Ti (not what run.c does, but what a reorder looks like): positions of sampled records change */
    for(int64_t i = 0; i < n; i++)
        P[i].Pos[0] += 1e-9;
    EXPECT(sync_(6, P, n) == e + 1, "moved particles inside one Ti_Current are seen through the sample hash");
    e = K.epoch;
    /* a reorder that keeps pointer and count (domain exchange / peano sort): swap two halves */
    for(int64_t i = 0; i < n / 2; i++) {
        struct rec t = P[i];
        P[i] = P[i + n / 2];
        P[i + n / 2] = t;
    }
    EXPECT(sync_(6, P, n) == e + 1, "a reorder with the same &P[0] and NumPart is a new epoch");
    e = K.epoch;
    EXPECT(sync_(6, P, n) == e, "... and the next call of the same step keeps it");
    EXPECT(sync_(6, P, n - 1) == e + 1, "a changed NumPart is a new epoch");
    e = K.epoch;
    memcpy(Q, P, (n + 1) * sizeof(struct rec));
    EXPECT(sync_(6, Q, n - 1) == e + 1, "the same records at another address are a new epoch");
    e = K.epoch;
    K.dirty = 1; /* mpg_shim_particles_changed() */
    EXPECT(sync_(6, Q, n - 1) == e + 1, "mpg_shim_particles_changed() forces one");
    e = K.epoch;
    EXPECT(sync_(6, Q, n - 1) == e, "... once");
    EXPECT(sync_(-1, Q, n - 1) == e + 1 && sync_(-1, Q, n - 1) == e + 2, "a caller that does not know Ti_Current never reuses an upload");
    e = K.epoch;
    /* an unsampled record changes: NOT seen (the documented limit of a 64-record sample; the explicit hook exists for that) */
    sync_(7, Q, n - 1);
    e = K.epoch;
    Q[1].Pos[2] += 1.0; /* stride is (n - 1) / 64 = 156: record 1 is not sampled */
    EXPECT(sync_(7, Q, n - 1) == e, "a change of one unsampled record inside a Ti_Current is not detected (documented)");
    /* small and empty tables */
    EXPECT(mpg_table_sample_hash(P, sizeof(struct rec), 0, 136, 0) == 1469598103934665603ull, "the hash of an empty table is the FNV offset");
    EXPECT(mpg_table_sample_hash(P, sizeof(struct rec), 3, 136, 0) != mpg_table_sample_hash(P, sizeof(struct rec), 2, 136, 0), "every record of a small table is sampled");
    if(fails)
        return 1;
    printf("PASS shim epoch bookkeeping (%lld epochs)\n", (long long)K.epoch);
    return 0;
}
