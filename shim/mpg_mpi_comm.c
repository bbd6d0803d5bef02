/* mpg_mpi_comm.c -- mpg_comm over MPI (see mpg_mpi_comm.h).  Compiles against any MPI-3 <mpi.h>. */
#include "mpg_mpi_comm.h"
#include <limits.h>
#include <stdlib.h>

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

/* counts arrive in bytes; every message of the library is a multiple of 8 bytes (32-byte rows, complex doubles, mesh planes), so
 * the exchange runs in 8-byte units and int counts reach 16 GiB per peer */
static int cb_alltoallv(void *ctx, const void *send, const int64_t *sb, const int64_t *sd, void *recv, const int64_t *rb, const int64_t *rd,
                        int on_device)
{
    (void)on_device;
    MPI_Comm comm = *(MPI_Comm *)ctx;
    int nt, rc = 0;
    MPI_Comm_size(comm, &nt);
    int *c = (int *)malloc(4 * (size_t)nt * sizeof(int));
    if(!c)
        return 1;
    for(int r = 0; r < nt && !rc; r++) {
        if((sb[r] | sd[r] | rb[r] | rd[r]) & 7 || sb[r] / 8 > INT_MAX || sd[r] / 8 > INT_MAX || rb[r] / 8 > INT_MAX || rd[r] / 8 > INT_MAX)
            rc = 1;
        c[r] = (int)(sb[r] / 8);
        c[nt + r] = (int)(sd[r] / 8);
        c[2 * nt + r] = (int)(rb[r] / 8);
        c[3 * nt + r] = (int)(rd[r] / 8);
    }
    if(!rc)
        rc = MPI_Alltoallv((void *)send, c, c + nt, MPI_INT64_T, recv, c + 2 * nt, c + 3 * nt, MPI_INT64_T, comm) != MPI_SUCCESS;
    free(c);
    return rc;
}

mpg_comm mpg_mpi_comm(MPI_Comm *comm)
{
    mpg_comm m;
    m.ctx = comm;
    MPI_Comm_rank(*comm, &m.ThisTask);
    MPI_Comm_size(*comm, &m.NTask);
    m.device_buffers = 0;
    m.allreduce = cb_allreduce;
    m.alltoall_i64 = cb_alltoall_i64;
    m.alltoallv = cb_alltoallv;
    return m;
}
