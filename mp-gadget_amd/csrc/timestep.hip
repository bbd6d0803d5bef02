// timestep.hip -- the streaming time-integration loops either side of the force path, on device-resident
// arrays: with positions, velocities and accelerations kept in HBM between force steps, the per-step host <-> device traffic
// of the drop-in path is no longer needed.
//
//   k_drift          drift_all_particles / real_drift_particle   libgadget/drift.c:18-102
//   k_pm_half_kick   apply_PM_half_kick                          libgadget/timestep.c:964-985
//   k_half_kick      apply_half_kick + do_grav_short_range_kick + do_hydro_kick (gas part)   timestep.c:873-929, 988-1036
// The per-bin factors (get_exact_drift/gravkick/hydrokick_factor, dloga_from_dti) are computed by the caller with the
// reference's own functions and passed in (mpg_kick_factors), as for the SPH loops.  Not carried: black-hole repositioning
// and the dynamic-friction / drag kicks of type-5 particles (sub-grid physics, out of scope).
// Pure HBM streaming: per particle 73 B read + 32 B written (drift), 48 + 24 B (PM kick), up to 98 + 32 B (half kick).
// Floating-point contraction is off in this file: the reference's loops are compiled without FMA, and these results are
// required to be bit-identical to it (tests/test_gpu_timestep.py).
#include "mpg_common.h"
#include "../../include/mpgadget_hip.h"

#pragma clang fp contract(off)

namespace mpg {

__global__ void __launch_bounds__(256) k_drift(int64_t n, double *__restrict__ pos, const double *__restrict__ vel, const uint8_t *__restrict__ type,
                                               const uint8_t *__restrict__ flags, double *__restrict__ hsml, const double *__restrict__ dthsml,
                                               double ddrift, double box, double s0, double s1, double s2, unsigned *__restrict__ err)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    const double shift[3] = {s0, s1, s2};
    double p[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
    const bool gone = flags && (flags[i] & 3); // IsGarbage | Swallowed: only the random shift is kept up to date (drift.c:21-30)
    if(!gone) {
        if(type && type[i] == 0 && hsml) { // drift.c:56-70
            double h = hsml[i] + dthsml[i] * ddrift;
            if(h <= 0)
                atomicExch(err, 5u);
            const double maxh = box / 2.;
            if(h > maxh)
                h = maxh;
            hsml[i] = h;
        }
#pragma unroll
        for(int j = 0; j < 3; j++) {
            p[j] += vel[3 * i + j] * ddrift + shift[j];
            if(!isfinite(p[j]))
                atomicExch(err, 5u);
        }
    }
    else {
#pragma unroll
        for(int j = 0; j < 3; j++)
            p[j] += shift[j];
    }
#pragma unroll
    for(int j = 0; j < 3; j++) {
        if(isfinite(p[j])) { // (the reference aborts before wrapping a non-finite position)
            while(p[j] > box)
                p[j] -= box;
            while(p[j] <= 0)
                p[j] += box;
        }
        pos[3 * i + j] = p[j];
    }
}

__global__ void __launch_bounds__(256) k_pm_half_kick(int64_t n, double *__restrict__ vel, const double *__restrict__ gravpm,
                                                      const uint8_t *__restrict__ flags, double F)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n || (flags && (flags[i] & 3)))
        return;
#pragma unroll
    for(int j = 0; j < 3; j++)
        vel[3 * i + j] += gravpm[3 * i + j] * F;
}

__global__ void __launch_bounds__(256) k_half_kick(int64_t n, const int *__restrict__ active, int64_t nactive, double *__restrict__ vel,
                                                   const double *__restrict__ gacc, const uint8_t *__restrict__ type, const uint8_t *__restrict__ flags,
                                                   const uint8_t *__restrict__ tbg, const uint8_t *__restrict__ tbh, const double *__restrict__ hacc,
                                                   double *__restrict__ entropy, const double *__restrict__ dtentropy, const mpg_kick_factors K,
                                                   unsigned *__restrict__ err)
{
    const int64_t pa = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(pa >= (active ? nactive : n))
        return;
    const int64_t i = active ? active[pa] : pa;
    if(flags && (flags[i] & 3))
        return;
    const int bg = tbg ? tbg[i] : 0;
    if(bg > MPG_TIMEBINS) {
        atomicExch(err, 4u);
        return;
    }
    double v[3] = {vel[3 * i], vel[3 * i + 1], vel[3 * i + 2]};
    if(K.bin_active[bg]) {
#pragma unroll
        for(int j = 0; j < 3; j++)
            v[j] += gacc[3 * i + j] * K.gravkick[bg];
    }
    const int ty = type ? type[i] : 1;
    if(ty == 0) {
        const int bh = tbh ? tbh[i] : 0;
#pragma unroll
        for(int j = 0; j < 3; j++)
            v[j] += hacc[3 * i + j] * K.hydrokick[bh];
        double vv = 0;
#pragma unroll
        for(int j = 0; j < 3; j++)
            vv += v[j] * v[j];
        vv = sqrt(vv);
        if(vv > 0 && vv / K.atime > K.MaxGasVel) {
#pragma unroll
            for(int j = 0; j < 3; j++)
                v[j] *= K.MaxGasVel * K.atime / vv;
        }
        entropy[i] += dtentropy[i] * K.dt_entr[bh];
    }
#pragma unroll
    for(int j = 0; j < 3; j++)
        vel[3 * i + j] = v[j];
}

// get_timestep_gravity_dloga, timestep.c:1039-1074
__global__ void __launch_bounds__(256) k_timestep_gravity(int64_t n, const double *__restrict__ gacc, const double *__restrict__ gpm, double atime,
                                                          double hubble, double errtol, double soft, double *__restrict__ dloga)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    const double a2inv = 1 / (atime * atime);
    double ax = a2inv * gacc[3 * i + 0];
    double ay = a2inv * gacc[3 * i + 1];
    double az = a2inv * gacc[3 * i + 2];
    ay += a2inv * gpm[3 * i + 1];
    ax += a2inv * gpm[3 * i + 0];
    az += a2inv * gpm[3 * i + 2];
    double ac2 = ax * ax + ay * ay + az * az;
    if(ac2 == 0)
        ac2 = 1.0e-60;
    const double ac = sqrt(ac2);
    const double dt = sqrt(2 * errtol * atime * (soft / 2.8) / ac);
    dloga[i] = dt * hubble;
}

static inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

void launch_timestep_gravity(int64_t n, const double *gacc, const double *gpm, double atime, double hubble, double errtol, double soft,
                             double *dloga, hipStream_t st)
{
    if(n > 0)
        hipLaunchKernelGGL(k_timestep_gravity, dim3(nblk(n)), dim3(256), 0, st, n, gacc, gpm, atime, hubble, errtol, soft, dloga);
    MPG_HIP(hipGetLastError());
}

void launch_drift(int64_t n, double *pos, const double *vel, const uint8_t *type, const uint8_t *flags, double *hsml, const double *dthsml,
                  double ddrift, double box, const double shift[3], unsigned *err, hipStream_t st)
{
    if(n > 0)
        hipLaunchKernelGGL(k_drift, dim3(nblk(n)), dim3(256), 0, st, n, pos, vel, type, flags, hsml, dthsml, ddrift, box, shift[0], shift[1],
                           shift[2], err);
    MPG_HIP(hipGetLastError());
}

void launch_pm_half_kick(int64_t n, double *vel, const double *gravpm, const uint8_t *flags, double F, hipStream_t st)
{
    if(n > 0)
        hipLaunchKernelGGL(k_pm_half_kick, dim3(nblk(n)), dim3(256), 0, st, n, vel, gravpm, flags, F);
    MPG_HIP(hipGetLastError());
}

void launch_half_kick(int64_t n, const int *active, int64_t nactive, double *vel, const double *gacc, const uint8_t *type, const uint8_t *flags,
                      const uint8_t *tbg, const uint8_t *tbh, const double *hacc, double *entropy, const double *dtentropy,
                      const mpg_kick_factors &K, unsigned *err, hipStream_t st)
{
    const int64_t na = active ? nactive : n;
    if(na > 0)
        hipLaunchKernelGGL(k_half_kick, dim3(nblk(na)), dim3(256), 0, st, n, active, nactive, vel, gacc, type, flags, tbg, tbh, hacc, entropy,
                           dtentropy, K, err);
    MPG_HIP(hipGetLastError());
}

} // namespace mpg
