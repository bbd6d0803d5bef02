// pm.h -- single-GPU particle-mesh solver (see pm.hip)
#pragma once
#include "mpg_common.h"
#include "tree_build.h"
#include <rocfft/rocfft.h>
#include <vector>

namespace mpg {

// One rocFFT plan with its execution info and work buffer (the plans of the PM solver: rocFFT's own API - rocfft_plan_create with a
// plan description where the layout is not the default, rocfft_execute on the engine's stream - not the hipFFT front end of rounds 1-4).
struct FftPlan {
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    DevBuf<char> work;
    // lengths: fastest dimension first (rocFFT's order); strides / distances in elements, 0 = contiguous default
    void create(rocfft_result_placement placement, rocfft_transform_type type, int dims, const size_t *lengths, size_t batch,
                const size_t *in_strides = nullptr, size_t in_dist = 0, const size_t *out_strides = nullptr, size_t out_dist = 0);
    void exec(void *in, void *out, hipStream_t st);
    void destroy();
    bool ready() const { return plan != nullptr; }
};

struct PMesh {
    double box = 0, Asmth = 0, G = 0, cellsize = 0;
    int nmesh = 0;
    bool have_plans = false;
    bool kspace_force = false; // true: forces by four inverse transforms as the reference does; false: by differencing the potential (pm.hip)
    FftPlan plan_r2c, plan_c2r;
    DevBuf<double> real;    // Nmesh^3
    DevBuf<double> rho_k;   // 2 * Nmesh^2 (Nmesh/2+1): potential in Fourier space after the transfer
    DevBuf<double> work_k;  // same size: per-component work array (Z2D overwrites its input)
    DevBuf<double> grad_z;  // Nmesh^3: the third force mesh of the one-pass gradient (the other two live in work_k / rho_k)
    DevBuf<double> invsinc2, difffac;
    // deposit: 0 not tuned yet, 1 plain atomics, 2 cell-sorted with wave-aggregated atomics (pm.hip)
    struct DepositState {
        int mode = 0, since_tune = 0;
    } dep_single, dep_slab;
    void deposit(int64_t n, const double *d_pos, const float *d_mass, const uint8_t *d_active, double *mesh, int x0, int P, DepositState &ds,
                 hipStream_t st, EventTimer *tm);
    DevBuf<unsigned long long> dep_keys_a, dep_keys_b;
    DevBuf<int> dep_idx_a, dep_idx_b;
    DevBuf<char> dep_tmp;
    // matter power spectrum of the PM density field (gravpm.c:331-382): raw sums of the last PM step
    bool measure_power = true, ps_valid = false;
    DevBuf<double> ps_acc;               // Power[Nmesh], kk[Nmesh], Norm
    DevBuf<unsigned long long> ps_modes; // Nmodes[Nmesh]
    void ps_zero(hipStream_t st);
    size_t ps_lds_bytes() const { return (size_t)nmesh * 3 * sizeof(double); }

    // gravpm_init_periodic -> petapm_init (gravpm.c:51-54, petapm.c:105-223)
    void init(double BoxSize, double Asmth, int Nmesh, double G, hipStream_t st);
    void ensure_single(); // meshes + 3-D plans of the single-GPU form, made on first use
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
        FftPlan p2d_r2c, p2d_c2r, p1d_fwd, p1d_inv, p1d_fwd_t, p1d_inv_t;
        bool strided = false;
        DevBuf<double> phi;      // the potential: planes -2 .. P+2 of Nmesh^2 each (2 + 3 ghost planes around the slab)
        DevBuf<double> force;    // one force component on planes 0 .. P (and the density slab before the forward transform)
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
    // recvA[Nmesh][Py][Nz] (x slowest): 1-D transform along x, potential transfer, inverse 1-D transform -> sendB[Nmesh][Py][Nz]
    void slab_forward_b(double *recvA, double *sendB, hipStream_t st);
    // recvB[world][P][Py][Nz] -> the potential slab (2-D c2r); ghost_send[5][Nmesh^2] = its first 3 and last 2 planes
    void slab_inverse_c(const double *recvB, double *ghost_send, hipStream_t st);
    // ghost_recv[5][Nmesh^2] = the next rank's first 3 planes, then the previous rank's last 2; forces by differencing the
    // potential, CIC readout for `nt` targets (caller indices) that lie in the slab
    void slab_readout_rows(const double *ghost_recv, int64_t nrows, const double *d_pos, double *d_gravpm, double *d_potential, hipStream_t st);
    void slab_readout(const double *ghost_recv, const int *targets, int64_t nt, const double *d_pos, double *d_gravpm, double *d_potential,
                      hipStream_t st);
};

} // namespace mpg
