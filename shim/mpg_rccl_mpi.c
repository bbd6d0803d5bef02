/* libgadget/mpg_rccl_mpi.c -- the library's native RCCL communicator (mpg_rccl_*, csrc/rccl_comm.hip) bootstrapped from MPI.
 *
 * MPI stays what it is in MP-Gadget - the launcher, the communicator of every host module - and carries ONE broadcast for this path:
 * rank 0's 128-byte ncclUniqueId.  From then on the exchanges of the force step (the ghost import that stands for the query export /
 * import of treewalk.c:586-655, the PM particle shipping and transposes that stand for petapm.c:751,815,869, the all-reduce of the top
 * of the tree that stands for force_exchange_pseudodata, forcetree.c:1145-1284) run as ncclSend / ncclRecv / ncclAllReduce on the
 * engine's stream between device buffers over xGMI: no host staging, no MPI in the step.  gravity-hip.c takes this communicator when
 * RCCL can be opened (MPG_SHIM_COMM=mpi keeps MPI with host staging, mpg_mpi_comm.c). */
#include <stdlib.h>
#include <string.h>
#include "mpg_mpi_comm.h"

int mpg_rccl_mpi_comm(MPI_Comm comm, int device, mpg_rccl **out, mpg_comm *c)
{
    char id[MPG_RCCL_ID_BYTES];
    int rank = 0, size = 1, ok = 1, all = 0;
    MPI_Comm_rank(comm, &rank);
    MPI_Comm_size(comm, &size);
    memset(id, 0, sizeof(id));
    *out = NULL;
    if(rank == 0)
        ok = mpg_rccl_available() && mpg_rccl_get_unique_id(id) == 0;
    MPI_Bcast(&ok, 1, MPI_INT, 0, comm);
    if(!ok)
        return 1; /* every rank returns: the caller falls back to mpg_mpi_comm on all of them */
    MPI_Bcast(id, MPG_RCCL_ID_BYTES, MPI_BYTE, 0, comm);
    ok = mpg_rccl_create(out, rank, size, id, device) == 0 && mpg_rccl_selftest(*out, 0) == 0 && mpg_rccl_comm(*out, c) == 0;
    MPI_Allreduce(&ok, &all, 1, MPI_INT, MPI_MIN, comm); /* one decision for all ranks */
    if(!all) {
        mpg_rccl_destroy(*out);
        *out = NULL;
        return 1;
    }
    return 0;
}
