// pm.hip -- long-range particle-mesh gravity (libgadget/petapm.c + gravpm.c) on one gfx950 GPU.
//
// Reference pipeline (gravpm_force, gravpm.c:61-119 -> petapm_force, petapm.c:359-379):
//   CIC deposit into per-region buffers (petapm.c:955-1020,1138-1144) -> pencil exchange into the FFT layout
//   (:786-840) -> r2c (:305) -> potential_transfer (gravpm.c:383-454) -> for Potential, ForceX, ForceY, ForceZ:
//   force_transfer (gravpm.c:458-489) -> c2r (petapm.c:344) -> exchange back (:842-885) -> CIC readout (gravpm.c:499-510).
// On one rank the regions / pencils only relocate cells: here particles deposit straight into
// the global Nmesh^3 mesh with hardware fp64 atomics (global_atomic_add_f64) and read straight back from it.
// PFFT (third-party, not vendored) is an unnormalised DFT; hipFFT/rocFFT D2Z / Z2D are the same transform.
// All kernels are HBM-streaming: per PM step ~ N*(28+128) + 5*3*2*R + 5*2*R + N*(24+256+32) bytes, R = 8*Nmesh^3.
#include "pm.h"
#include <cmath>

namespace mpg {

#define MPG_FFT(expr)                                                                                          \
    do {                                                                                                       \
        hipfftResult _r = (expr);                                                                              \
        if(_r != HIPFFT_SUCCESS)                                                                               \
            ::mpg::fail(__FILE__, __LINE__, std::string("hipFFT error ") + std::to_string((int)_r) + " in " #expr); \
    } while(0)

__device__ __forceinline__ int wrap(int i, int n)
{
    // periodic wrap of petapm.c:903-918 (cells -1 .. Nmesh+1 can occur)
    i = (i >= n) ? i - n : i;
    i = (i < 0) ? i + n : i;
    return i;
}

// put_particle_to_mesh through pm_iterate_one (petapm.c:955-1020, :1138-1144)
__global__ void __launch_bounds__(256) k_cic_deposit(int64_t n, const double *__restrict__ pos, const float *__restrict__ mass,
                                                     const uint8_t *__restrict__ active, double cellsize, int nmesh,
                                                     double *__restrict__ mesh)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    if(active && !active[i])
        return;
    int ic[3];
    double res[3];
#pragma unroll
    for(int k = 0; k < 3; k++) {
        const double tmp = pos[3 * i + k] / cellsize;
        const double fl = floor(tmp);
        ic[k] = (int)fl;
        res[k] = tmp - fl;
    }
    const double m = (double)mass[i];
#pragma unroll
    for(int c = 0; c < 8; c++) {
        double w = 1.0;
        size_t lin = 0;
#pragma unroll
        for(int k = 0; k < 3; k++) {
            const int off = (c >> k) & 1;
            lin = lin * (size_t)nmesh + (size_t)wrap(ic[k] + off, nmesh);
            w *= off ? res[k] : (1 - res[k]);
        }
        unsafeAtomicAdd(&mesh[lin], w * m);
    }
}

__device__ __forceinline__ double sinc_unnormed(double x)
{
    // gravpm.c:295-302
    if(x < 1e-5 && x > -1e-5) {
        const double x2 = x * x;
        return 1.0 - x2 / 6. + x2 * x2 / 120.;
    }
    return sin(x) / x;
}

// potential_transfer, gravpm.c:383-454, swept as pm_apply_transfer_function does (petapm.c:1092-1132).
// Layout here: [kx][ky][kz], kz in [0, N/2].  k index -> signed mode: petapm_mesh_to_k, petapm.c:81-84.
__global__ void __launch_bounds__(256) k_potential_transfer(int nmesh, double asmth2, double pot_factor, const double *__restrict__ invsinc2,
                                                            double2 *__restrict__ cplx)
{
    const int nz = nmesh / 2 + 1;
    const size_t total = (size_t)nmesh * nmesh * nz;
    const size_t ip = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(ip >= total)
        return;
    const int iz = (int)(ip % nz);
    const size_t t = ip / nz;
    const int iy = (int)(t % nmesh);
    const int ix = (int)(t / nmesh);
    const int kx = ix <= nmesh / 2 ? ix : ix - nmesh;
    const int ky = iy <= nmesh / 2 ? iy : iy - nmesh;
    const int kz = iz;
    const long long k2 = (long long)kx * kx + (long long)ky * ky + (long long)kz * kz;
    double2 v = cplx[ip];
    if(k2 == 0) {
        v.x = 0.0;
        v.y = 0.0;
    }
    else {
        const double smth = exp(-(double)k2 * asmth2) / (double)k2;
        // f = prod 1/sinc^2 ; fac = pot_factor * smth * f * f
        const double f = invsinc2[ix] * invsinc2[iy] * invsinc2[iz];
        const double fac = pot_factor * smth * f * f;
        v.x *= fac;
        v.y *= fac;
    }
    cplx[ip] = v;
}

// force_transfer for one axis, gravpm.c:476-498: (re, im) <- (-im*fac, re*fac), fac = -diff_kernel(k 2pi/N) N/Box.
// axis < 0: plain copy (the Potential pass has no transfer function, gravpm.c:32-39).
__global__ void __launch_bounds__(256) k_force_transfer(int nmesh, int axis, const double *__restrict__ difffac,
                                                        const double2 *__restrict__ src, double2 *__restrict__ dst)
{
    const int nz = nmesh / 2 + 1;
    const size_t total = (size_t)nmesh * nmesh * nz;
    const size_t ip = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(ip >= total)
        return;
    double2 v = src[ip];
    if(axis >= 0) {
        const int iz = (int)(ip % nz);
        const size_t t = ip / nz;
        const int iy = (int)(t % nmesh);
        const int ix = (int)(t / nmesh);
        const int ii = axis == 0 ? ix : (axis == 1 ? iy : iz);
        const double fac = difffac[ii];
        const double t0 = -v.y * fac, t1 = v.x * fac;
        v.x = t0;
        v.y = t1;
    }
    dst[ip] = v;
}

// readout_potential / readout_force_{x,y,z} through pm_iterate_one (gravpm.c:499-510).
// comp < 3: out[3*i+comp] = sum (GravPM is zeroed before, gravpm.c:88-92); comp == 3: out[i] += sum (Potential).
__global__ void __launch_bounds__(256) k_cic_readout(int64_t n, const double *__restrict__ pos, const uint8_t *__restrict__ active,
                                                     double cellsize, int nmesh, const double *__restrict__ mesh, int comp,
                                                     double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    int ic[3];
    double res[3];
#pragma unroll
    for(int k = 0; k < 3; k++) {
        const double tmp = pos[3 * i + k] / cellsize;
        const double fl = floor(tmp);
        ic[k] = (int)fl;
        res[k] = tmp - fl;
    }
    double acc = 0;
#pragma unroll
    for(int c = 0; c < 8; c++) {
        double w = 1.0;
        size_t lin = 0;
#pragma unroll
        for(int k = 0; k < 3; k++) {
            const int off = (c >> k) & 1;
            lin = lin * (size_t)nmesh + (size_t)wrap(ic[k] + off, nmesh);
            w *= off ? res[k] : (1 - res[k]);
        }
        acc += w * mesh[lin];
    }
    if(comp < 3)
        out[3 * i + comp] = acc;
    else
        out[i] += acc;
}

static inline unsigned nblk(size_t n, int b = 256) { return (unsigned)((n + b - 1) / b); }

void PMesh::init(double BoxSize, double Asmth_, int Nmesh_, double G_, hipStream_t st)
{
    destroy();
    MPG_CHECK(Nmesh_ >= 2 && (Nmesh_ % 2) == 0, "gravpm_init_periodic: Nmesh must be even and >= 2");
    box = BoxSize;
    Asmth = Asmth_;
    nmesh = Nmesh_;
    G = G_;
    cellsize = box / nmesh; // petapm.c:112
    const size_t nreal = (size_t)nmesh * nmesh * nmesh;
    const size_t ncplx = (size_t)nmesh * nmesh * (nmesh / 2 + 1);
    real.reserve(nreal);
    rho_k.reserve(2 * ncplx);
    work_k.reserve(2 * ncplx);
    // per-index tables: 1/sinc^2(pi k / N) (gravpm.c:412-418) and the differencing factor (gravpm.c:482)
    std::vector<double> is2(nmesh), dff(nmesh);
    for(int i = 0; i < nmesh; i++) {
        const int k = i <= nmesh / 2 ? i : i - nmesh;
        double tmp = (k * M_PI) / nmesh;
        double s = (tmp < 1e-5 && tmp > -1e-5) ? 1.0 - tmp * tmp / 6. + tmp * tmp * tmp * tmp / 120. : sin(tmp) / tmp;
        is2[i] = 1. / (s * s);
        const double w = k * (2 * M_PI / nmesh);
        dff[i] = -1 * (1 / 6.0 * (8 * sin(w) - sin(2 * w))) * (nmesh / box);
    }
    invsinc2.reserve(nmesh);
    difffac.reserve(nmesh);
    MPG_HIP(hipMemcpyAsync(invsinc2.p, is2.data(), nmesh * sizeof(double), hipMemcpyHostToDevice, st));
    MPG_HIP(hipMemcpyAsync(difffac.p, dff.data(), nmesh * sizeof(double), hipMemcpyHostToDevice, st));
    MPG_HIP(hipStreamSynchronize(st));
    MPG_FFT(hipfftCreate(&plan_r2c));
    MPG_FFT(hipfftCreate(&plan_c2r));
    size_t ws1 = 0, ws2 = 0;
    MPG_FFT(hipfftMakePlan3d(plan_r2c, nmesh, nmesh, nmesh, HIPFFT_D2Z, &ws1));
    MPG_FFT(hipfftMakePlan3d(plan_c2r, nmesh, nmesh, nmesh, HIPFFT_Z2D, &ws2));
    have_plans = true;
}

void PMesh::destroy()
{
    if(have_plans) {
        (void)hipfftDestroy(plan_r2c);
        (void)hipfftDestroy(plan_c2r);
        have_plans = false;
    }
    real.release();
    rho_k.release();
    work_k.release();
    nmesh = 0;
}

void PMesh::force(int64_t n, const double *d_pos, const float *d_mass, const uint8_t *d_active, double *d_gravpm, double *d_potential,
                  hipStream_t st, EventTimer *tm)
{
    MPG_CHECK(have_plans, "gravpm_force called before gravpm_init_periodic");
    const size_t nreal = (size_t)nmesh * nmesh * nmesh;
    const size_t ncplx = (size_t)nmesh * nmesh * (nmesh / 2 + 1);
    MPG_FFT(hipfftSetStream(plan_r2c, st));
    MPG_FFT(hipfftSetStream(plan_c2r, st));
    float t_fft = 0, t_tr = 0, t_ro = 0, t;
    if(tm)
        tm->start(st);
    // pm_init_regions zeroes the mesh (petapm.c:932-952); deposit
    MPG_HIP(hipMemsetAsync(real.p, 0, nreal * sizeof(double), st));
    if(n > 0)
        hipLaunchKernelGGL(k_cic_deposit, dim3(nblk(n)), dim3(256), 0, st, n, d_pos, d_mass, d_active, cellsize, nmesh, real.p);
    if(tm)
        tm->lap(st, &tm->t.pm_deposit);
    MPG_FFT(hipfftExecD2Z(plan_r2c, real.p, (hipfftDoubleComplex *)rho_k.p));
    if(tm) {
        tm->lap(st, &t);
        t_fft += t;
    }
    const double asmth2 = pow((2 * M_PI) * Asmth / nmesh, 2);
    const double pot_factor = -G / (M_PI * box);
    hipLaunchKernelGGL(k_potential_transfer, dim3(nblk(ncplx)), dim3(256), 0, st, nmesh, asmth2, pot_factor, invsinc2.p, (double2 *)rho_k.p);
    if(tm) {
        tm->lap(st, &t);
        t_tr += t;
    }
    // functions[] = Potential, ForceX, ForceY, ForceZ (gravpm.c:32-39)
    for(int f = 0; f < 4; f++) {
        const int axis = f - 1;
        if(f == 0 && !d_potential)
            continue;
        hipLaunchKernelGGL(k_force_transfer, dim3(nblk(ncplx)), dim3(256), 0, st, nmesh, axis, difffac.p, (const double2 *)rho_k.p,
                           (double2 *)work_k.p);
        if(tm) {
            tm->lap(st, &t);
            t_tr += t;
        }
        MPG_FFT(hipfftExecZ2D(plan_c2r, (hipfftDoubleComplex *)work_k.p, real.p));
        if(tm) {
            tm->lap(st, &t);
            t_fft += t;
        }
        if(n > 0) {
            if(f == 0)
                hipLaunchKernelGGL(k_cic_readout, dim3(nblk(n)), dim3(256), 0, st, n, d_pos, d_active, cellsize, nmesh, real.p, 3, d_potential);
            else
                hipLaunchKernelGGL(k_cic_readout, dim3(nblk(n)), dim3(256), 0, st, n, d_pos, d_active, cellsize, nmesh, real.p, axis, d_gravpm);
        }
        if(tm) {
            tm->lap(st, &t);
            t_ro += t;
        }
    }
    MPG_HIP(hipGetLastError());
    if(tm && tm->enabled) {
        tm->t.pm_fft = t_fft;
        tm->t.pm_transfer = t_tr;
        tm->t.pm_readout = t_ro;
        tm->t.pm_total = tm->t.pm_deposit + t_fft + t_tr + t_ro;
    }
}

} // namespace mpg
