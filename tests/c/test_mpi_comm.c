/* test_mpi_comm.c -- shim/mpg_mpi_comm.c by itself, on host buffers, under a real MPI (no GPU, no library): the three collectives the
 * library asks its caller for (include/mpgadget_hip.h, mpg_comm) against known patterns, on ragged and empty blocks.
 * Built and started with mpiexec by tests/test_abi.py::test_shim_mpi_communicator_runs (CPU).  Exit 0 and "PASS" from rank 0. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mpg_mpi_comm.h"

#define FAIL(...)                                  \
    do {                                           \
        fprintf(stderr, "FAIL rank %d: ", me);     \
        fprintf(stderr, __VA_ARGS__);              \
        fprintf(stderr, "\n");                     \
        MPI_Abort(MPI_COMM_WORLD, 1);              \
    } while(0)

/* byte b of the block that rank s sends to rank d */
static unsigned char pat(int s, int d, int64_t b) { return (unsigned char)(31 * s + 7 * d + 3 * b + 1); }
/* bytes rank s sends to rank d: ragged, some empty, one not a multiple of 8 */
static int64_t blk(int s, int d, int nt) { return ((s + 2 * d) % 3 == 0) ? 0 : 1000 * (int64_t)(1 + (s * nt + d) % 5) + (s == 0 ? 3 : 0); }

int main(int argc, char **argv)
{
    MPI_Init(&argc, &argv);
    static MPI_Comm world;
    world = MPI_COMM_WORLD;
    mpg_comm c = mpg_mpi_comm(&world);
    const int me = c.ThisTask, nt = c.NTask;
    int rk, sz;
    MPI_Comm_rank(world, &rk);
    MPI_Comm_size(world, &sz);
    if(me != rk || nt != sz || c.device_buffers != 0 || c.bind_stream != NULL || c.ctx != &world)
        FAIL("descriptor: task %d of %d, device_buffers %d", me, nt, c.device_buffers);

    /* allreduce: doubles SUM, int64 SUM, int64 MAX, doubles MAX (dtype ? int64 : double, op ? MAX : SUM), in place */
    double d[3] = {1.0 + me, 0.5 * me, -2.0};
    if(c.allreduce(c.ctx, d, 3, 0, 0, 0))
        FAIL("allreduce double sum returned an error");
    if(d[0] != nt + 0.5 * nt * (nt - 1) || d[1] != 0.25 * nt * (nt - 1) || d[2] != -2.0 * nt)
        FAIL("allreduce double sum: %g %g %g", d[0], d[1], d[2]);
    int64_t k[2] = {(int64_t)1 << (33 + me % 2), 5 - me};
    int64_t ksum = 0;
    for(int r = 0; r < nt; r++)
        ksum += (int64_t)1 << (33 + r % 2);
    int64_t k2[2] = {k[0], k[1]};
    if(c.allreduce(c.ctx, k, 2, 1, 0, 0) || k[0] != ksum || k[1] != 5 * (int64_t)nt - (int64_t)nt * (nt - 1) / 2)
        FAIL("allreduce int64 sum: %lld %lld", (long long)k[0], (long long)k[1]);
    if(c.allreduce(c.ctx, k2, 2, 1, 1, 0) || k2[0] != ((int64_t)1 << (nt > 1 ? 34 : 33)) || k2[1] != 5)
        FAIL("allreduce int64 max: %lld %lld", (long long)k2[0], (long long)k2[1]);
    double dm = -1.5 * me;
    if(c.allreduce(c.ctx, &dm, 1, 0, 1, 0) || dm != 0.0)
        FAIL("allreduce double max: %g", dm);
    if(c.allreduce(c.ctx, &dm, (int64_t)1 << 32, 0, 0, 0) == 0)   /* a count MPI's int cannot hold is refused, not truncated */
        FAIL("allreduce accepted a count beyond INT_MAX");

    /* alltoall of one int64 per peer */
    int64_t *s1 = malloc(nt * sizeof(int64_t)), *r1 = malloc(nt * sizeof(int64_t));
    for(int r = 0; r < nt; r++)
        s1[r] = ((int64_t)me << 40) + r;
    if(c.alltoall_i64(c.ctx, s1, r1))
        FAIL("alltoall_i64 returned an error");
    for(int r = 0; r < nt; r++)
        if(r1[r] != ((int64_t)r << 40) + me)
            FAIL("alltoall_i64: from %d got %lld", r, (long long)r1[r]);

    /* alltoallv in bytes: ragged blocks, gaps between the blocks on both sides (displacements are not the running sums) */
    int64_t *sb = malloc(4 * nt * sizeof(int64_t)), *sd = sb + nt, *rb = sd + nt, *rd = rb + nt;
    int64_t so = 5, ro = 11;
    for(int r = 0; r < nt; r++) {
        sb[r] = blk(me, r, nt);
        sd[r] = so;
        so += sb[r] + 13;
        rb[r] = blk(r, me, nt);
        rd[r] = ro;
        ro += rb[r] + 7;
    }
    unsigned char *sbuf = malloc(so + 1), *rbuf = malloc(ro + 1);
    memset(sbuf, 0xEE, so + 1);
    memset(rbuf, 0xDD, ro + 1);
    for(int r = 0; r < nt; r++)
        for(int64_t b = 0; b < sb[r]; b++)
            sbuf[sd[r] + b] = pat(me, r, b);
    if(c.alltoallv(c.ctx, sbuf, sb, sd, rbuf, rb, rd, 0))
        FAIL("alltoallv returned an error");
    int64_t at = 0;
    for(int r = 0; r < nt; r++) {
        for(; at < rd[r]; at++)
            if(rbuf[at] != 0xDD)
                FAIL("alltoallv wrote into the gap in front of block %d", r);
        for(int64_t b = 0; b < rb[r]; b++, at++)
            if(rbuf[at] != pat(r, me, b))
                FAIL("alltoallv: byte %lld of the block from %d", (long long)b, r);
    }
    for(; at <= ro; at++)
        if(rbuf[at] != 0xDD)
            FAIL("alltoallv wrote behind the last block");
    /* all blocks empty */
    for(int r = 0; r < nt; r++)
        sb[r] = rb[r] = 0;
    if(c.alltoallv(c.ctx, sbuf, sb, sd, rbuf, rb, rd, 0))
        FAIL("alltoallv of empty blocks returned an error");

    /* a displacement beyond INT_MAX bytes on the receiving side: the shim agrees (one MPI_Allreduce) on 8-byte units for all ranks.
     * The last block of every rank lands 2^31 + 64 bytes into a buffer of which only the pages written are ever touched. */
    {
        const int64_t far = ((int64_t)1 << 31) + 64;
        unsigned char *big = malloc((size_t)far + 4096);
        if(!big)
            FAIL("malloc of the sparse 2 GiB receive buffer");
        so = 0;
        for(int r = 0; r < nt; r++) {
            sb[r] = 8 * (int64_t)(1 + (me + r) % 4);
            sd[r] = so;
            so += sb[r];
            rb[r] = 8 * (int64_t)(1 + (r + me) % 4);
            rd[r] = (r == nt - 1) ? far : 64 * (int64_t)r;
        }
        for(int r = 0; r < nt; r++)
            for(int64_t b = 0; b < sb[r]; b++)
                sbuf[sd[r] + b] = pat(me, r, b);
        if(c.alltoallv(c.ctx, sbuf, sb, sd, big, rb, rd, 0))
            FAIL("alltoallv with a displacement beyond INT_MAX returned an error");
        for(int r = 0; r < nt; r++)
            for(int64_t b = 0; b < rb[r]; b++)
                if(big[rd[r] + b] != pat(r, me, b))
                    FAIL("alltoallv (8-byte units): byte %lld of the block from %d", (long long)b, r);
        /* ... and a block that is not a multiple of 8 bytes cannot be expressed in those units: this rank reports an error */
        sb[me] = rb[me] = 12;
        const int rc = c.alltoallv(c.ctx, sbuf, sb, sd, big, rb, rd, 0);
        if(!rc)
            FAIL("alltoallv accepted a 12-byte block in 8-byte units");
        free(big);
    }

    int ok = 1, all = 0;
    MPI_Reduce(&ok, &all, 1, MPI_INT, MPI_SUM, 0, world);
    if(me == 0)
        printf("PASS mpg_mpi_comm on %d MPI processes\n", all);
    MPI_Finalize();
    return 0;
}
