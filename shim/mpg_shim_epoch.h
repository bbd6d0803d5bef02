/* libgadget/mpg_shim_epoch.h -- the bookkeeping of mpg_shim_sync that needs no reference type: WHEN is the rank's particle table a new one
 * for the engine's upload cache (a new "epoch" for mpg_set_particle_epoch)?  Kept apart from gravity-hip.c so that it can be compiled and
 * run without the reference tree (tests/c/test_shim_epoch.c).
 *
 * The table is identified by (Ti_Current, &P[0], NumPart) plus a hash over a sample of the records' IDs and positions: a reorder or an
 * exchange inside one Ti_Current that keeps the pointer and the count (the second domain_decompose_full of the first step, run.c:422,434;
 * fof_fof's exchange) changes the sample.  A caller that knows says so at once (mpg_shim_particles_changed -> dirty). */
#ifndef MPG_SHIM_EPOCH_H
#define MPG_SHIM_EPOCH_H
#include <stddef.h>
#include <stdint.h>
#include <string.h>

struct mpg_table_key {
    int64_t epoch;       /* what mpg_set_particle_epoch receives */
    int64_t ti;          /* Ti_Current of the epoch (-1: none yet) */
    const void *base;    /* &P[0] */
    int64_t numpart;
    uint64_t sample;     /* mpg_table_sample_hash of the table the epoch was declared for */
    int dirty;           /* mpg_shim_particles_changed() since then */
};
#define MPG_TABLE_KEY_INIT {0, -1, NULL, -1, 0, 1}

/* FNV-1a over the IDs (8 bytes at off_id) and positions (3 doubles at off_pos) of <= 64..127 records spread evenly over a table of n records
 * of `stride` bytes */
static inline uint64_t mpg_table_sample_hash(const void *table, size_t stride, int64_t n, size_t off_id, size_t off_pos)
{
    uint64_t h = 1469598103934665603ull;
    const int64_t step = n > 64 ? n / 64 : 1;
    int64_t i;
    int k;
    for(i = 0; i < n; i += step) {
        const char *rec = (const char *)table + (size_t)i * stride;
        uint64_t w[4];
        memcpy(&w[0], rec + off_id, 8);
        memcpy(&w[1], rec + off_pos, 24);
        for(k = 0; k < 4; k++)
            h = (h ^ w[k]) * 1099511628211ull;
    }
    return h;
}

/* Does what the caller sees now differ from the table of the current epoch?  ti < 0 (a caller that does not know the time line position)
 * always does.  On several ranks the answers are combined (MPI_MAX) BEFORE mpg_table_key_take, so that every rank opens the epoch. */
static inline int mpg_table_key_differs(const struct mpg_table_key *k, int64_t ti, const void *base, int64_t numpart, uint64_t sample)
{
    return k->dirty || ti < 0 || ti != k->ti || base != k->base || numpart != k->numpart || sample != k->sample;
}

/* open a new epoch for the table seen now */
static inline void mpg_table_key_take(struct mpg_table_key *k, int64_t ti, const void *base, int64_t numpart, uint64_t sample)
{
    k->epoch++;
    k->ti = ti;
    k->base = base;
    k->numpart = numpart;
    k->sample = sample;
    k->dirty = 0;
}
#endif
