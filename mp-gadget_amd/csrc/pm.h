// pm.h -- single-GPU particle-mesh solver (see pm.hip)
#pragma once
#include "mpg_common.h"
#include "tree_build.h"
#include <hipfft/hipfft.h>
#include <vector>

namespace mpg {

struct PMesh {
    double box = 0, Asmth = 0, G = 0, cellsize = 0;
    int nmesh = 0;
    bool have_plans = false;
    hipfftHandle plan_r2c{}, plan_c2r{};
    DevBuf<double> real;    // Nmesh^3
    DevBuf<double> rho_k;   // 2 * Nmesh^2 (Nmesh/2+1): potential in Fourier space after the transfer
    DevBuf<double> work_k;  // same size: per-component work array (Z2D overwrites its input)
    DevBuf<double> invsinc2, difffac;

    // gravpm_init_periodic -> petapm_init (gravpm.c:51-54, petapm.c:105-223)
    void init(double BoxSize, double Asmth, int Nmesh, double G, hipStream_t st);
    // petapm_destroy (petapm.c:225-232)
    void destroy();
    // gravpm_force (gravpm.c:61-119): d_gravpm[n][3] is assigned, d_potential[n] (may be null) is incremented
    void force(int64_t n, const double *d_pos, const float *d_mass, const uint8_t *d_active, double *d_gravpm, double *d_potential,
               hipStream_t st, EventTimer *tm);
    ~PMesh() { destroy(); }

    // ---- slab-decomposed form for several GPUs (one process per GPU): rank r owns the x-planes [r P, (r+1) P), P = Nmesh / world,
    // of the real mesh and, between the two transposes, the ky-rows [r Py, (r+1) Py) of the Fourier mesh.  The transposes are
    // all-to-alls done by the caller (RCCL); see pm.hip.
    struct Slab {
        int rank = 0, world = 1, P = 0, Py = 0;
        bool ready = false;
        hipfftHandle p2d_r2c{}, p2d_c2r{}, p1d_fwd{};
        DevBuf<double> realF[4]; // Potential, ForceX, ForceY, ForceZ: (P + 1) planes of Nmesh^2 each, the last is the ghost plane
        DevBuf<double> C;        // 2 * P * Nmesh * (Nmesh/2+1): the slab after / before the 2-D transforms
        DevBuf<double> rho_k;    // 2 * Nmesh * Py * (Nmesh/2+1): potential in Fourier space, layout [ky local][kz][kx]
        DevBuf<double> work;     // same size: per-function work array
    } slab;
    DevBuf<unsigned> slab_err;
    size_t slab_cplx_per_peer() const { return (size_t)slab.P * slab.Py * (nmesh / 2 + 1); }
    void slab_init(int rank, int world);
    void slab_destroy();
    // deposit the particles whose CIC cloud touches this rank's planes, 2-D r2c, pack for the transpose: sendA[world][P][Py][Nz]
    void slab_forward_a(int64_t n, const double *d_pos, const float *d_mass, double *sendA, hipStream_t st);
    // recvA[Nmesh][Py][Nz] (x slowest): 1-D transform along x, potential transfer; then per function the force transfer and
    // the inverse 1-D transform straight into sendB[Nmesh][4][Py][Nz]
    void slab_forward_b(double *recvA, double *sendB, hipStream_t st);
    // recvB[world][P][4][Py][Nz] -> 4 real slabs (2-D c2r); ghost_send[4][Nmesh^2] = first plane of each
    void slab_inverse_c(const double *recvB, double *ghost_send, hipStream_t st);
    // ghost_recv[4][Nmesh^2] = first planes of the next rank; CIC readout for `nt` targets (caller indices) that lie in the slab
    void slab_readout(const double *ghost_recv, const int *targets, int64_t nt, const double *d_pos, double *d_gravpm, double *d_potential,
                      hipStream_t st);
};

} // namespace mpg
