/* mpg_mpi_comm.h -- the three collectives of mpg_comm (include/mpgadget_hip.h) on an MPI communicator: what the in-tree shim
 * (gravity-hip.c, sph-hip.c) hands to the mpg_dist_* calls.  Host buffers (device_buffers = 0): the library stages through pinned
 * memory, so any MPI works; set device_buffers after the call if the MPI is GPU-aware. */
#ifndef MPG_MPI_COMM_H
#define MPG_MPI_COMM_H
#include <mpi.h>
#include <mpgadget_hip.h>
/* `comm` must stay valid as long as the returned struct is in use (its address is the callback context) */
mpg_comm mpg_mpi_comm(MPI_Comm *comm);
/* mpg_rccl_mpi.c: the native RCCL communicator with MPI as its bootstrap (one MPI_Bcast of the unique id).  Collective over comm;
 * returns non-zero ON EVERY RANK when RCCL is unavailable or its self-test fails on any rank (the caller then uses mpg_mpi_comm). */
int mpg_rccl_mpi_comm(MPI_Comm comm, int device, mpg_rccl **out, mpg_comm *c);
#endif
