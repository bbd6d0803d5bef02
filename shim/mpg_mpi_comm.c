/* mpg_mpi_comm.c -- mpg_comm over MPI (see mpg_mpi_comm.h).  Compiles against any MPI-3 <mpi.h>. */
#include "mpg_mpi_comm.h"
#include <limits.h>
#include <stdlib.h>
#include <string.h>

static int cb_allreduce(void *ctx, void *buf, int64_t count, int dtype, int op, int on_device)
{
    (void)on_device; /* a GPU-aware MPI takes the device pointer as it is */
    if(count > INT_MAX)
        return 1;
    return MPI_Allreduce(MPI_IN_PLACE, buf, (int)count, dtype ? MPI_INT64_T : MPI_DOUBLE, op ? MPI_MAX : MPI_SUM, *(MPI_Comm *)ctx) != MPI_SUCCESS;
}

static int cb_alltoall_i64(void *ctx, const int64_t *send, int64_t *recv)
{
    return MPI_Alltoall((void *)send, 1, MPI_INT64_T, recv, 1, MPI_INT64_T, *(MPI_Comm *)ctx) != MPI_SUCCESS;
}

/* counts arrive in bytes.  MPI counts are ints: an exchange whose blocks and displacements all stay below 2 GiB ON EVERY RANK goes in
 * bytes; otherwise in 8-byte units (the large messages of the library - 32 / 48 / 128-byte rows, complex doubles, mesh planes - are
 * multiples of 8 bytes), which reaches 16 GiB per peer.  The unit is agreed with one MPI_Allreduce so that all ranks use the same
 * datatype (the type signatures of sender and receiver must match). */
static int cb_alltoallv(void *ctx, const void *send, const int64_t *sb, const int64_t *sd, void *recv, const int64_t *rb, const int64_t *rd,
                        int on_device)
{
    (void)on_device;
    MPI_Comm comm = *(MPI_Comm *)ctx;
    int nt, rc = 0;
    MPI_Comm_size(comm, &nt);
    int small = 1, all_small = 0;
    for(int r = 0; r < nt; r++)
        if(sb[r] > INT_MAX || sd[r] > INT_MAX || rb[r] > INT_MAX || rd[r] > INT_MAX)
            small = 0;
    if(MPI_Allreduce(&small, &all_small, 1, MPI_INT, MPI_MIN, comm) != MPI_SUCCESS)
        return 1;
    const int64_t unit = all_small ? 1 : 8;
    int *c = (int *)malloc(4 * (size_t)nt * sizeof(int));
    if(!c)
        return 1;
    for(int r = 0; r < nt; r++) {
        if((sb[r] | sd[r] | rb[r] | rd[r]) % unit || sb[r] / unit > INT_MAX || sd[r] / unit > INT_MAX || rb[r] / unit > INT_MAX ||
           rd[r] / unit > INT_MAX)
            rc = 1;
        c[r] = (int)(sb[r] / unit);
        c[nt + r] = (int)(sd[r] / unit);
        c[2 * nt + r] = (int)(rb[r] / unit);
        c[3 * nt + r] = (int)(rd[r] / unit);
    }
    /* (a rank that cannot express its blocks still takes part, with empty ones, so that the others do not hang; it reports the error) */
    if(rc)
        for(int r = 0; r < 4 * nt; r++)
            c[r] = 0;
    const MPI_Datatype ty = all_small ? MPI_BYTE : MPI_INT64_T;
    if(MPI_Alltoallv((void *)send, c, c + nt, ty, recv, c + 2 * nt, c + 3 * nt, ty, comm) != MPI_SUCCESS)
        rc = 1;
    free(c);
    return rc;
}

mpg_comm mpg_mpi_comm(MPI_Comm *comm)
{
    mpg_comm m;
    memset(&m, 0, sizeof(m)); /* (bind_stream = NULL: MPI calls block) */
    m.ctx = comm;
    MPI_Comm_rank(*comm, &m.ThisTask);
    MPI_Comm_size(*comm, &m.NTask);
    m.device_buffers = 0;
    m.allreduce = cb_allreduce;
    m.alltoall_i64 = cb_alltoall_i64;
    m.alltoallv = cb_alltoallv;
    return m;
}
