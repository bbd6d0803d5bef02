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
};

} // namespace mpg
