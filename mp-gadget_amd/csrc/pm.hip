// pm.hip -- long-range particle-mesh gravity (libgadget/petapm.c + gravpm.c) on one gfx950 GPU.
//
// Reference pipeline (gravpm_force, gravpm.c:61-119 -> petapm_force, petapm.c:359-379):
//   CIC deposit into per-region buffers (petapm.c:955-1020,1138-1144) -> pencil exchange into the FFT layout
//   (:786-840) -> r2c (:305) -> potential_transfer (gravpm.c:383-454) -> for Potential, ForceX, ForceY, ForceZ:
//   force_transfer (gravpm.c:458-489) -> c2r (petapm.c:344) -> exchange back (:842-885) -> CIC readout (gravpm.c:499-510).
// On one rank the regions / pencils only relocate cells: here particles deposit straight into
// the global Nmesh^3 mesh with hardware fp64 atomics (global_atomic_add_f64) and read straight back from it.
// PFFT (third-party, not vendored) is an unnormalised DFT; the real-to-complex / complex-to-real transforms of rocFFT are the same transform.
// All kernels are HBM-streaming: per PM step ~ N*(28+128) + 5*3*2*R + 5*2*R + N*(24+256+32) bytes, R = 8*Nmesh^3.
#include "pm.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <rocprim/rocprim.hpp>

namespace mpg {

#define MPG_FFT(expr)                                                                                          \
    do {                                                                                                       \
        rocfft_status _r = (expr);                                                                             \
        if(_r != rocfft_status_success)                                                                        \
            ::mpg::fail(__FILE__, __LINE__, std::string("rocFFT error ") + std::to_string((int)_r) + " in " #expr); \
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

// ---- deposit for clustered sets.  In a dense clump thousands of particles share a handful of mesh cells and the plain kernel's
// atomics serialise on those addresses (256^3 clustered set: 29.8 ms instead of 4.0).  Here the particles are first sorted by
// their base cell (one radix sort of cell index -> particle); a wave then holds RUNS of particles with the same 8 target cells,
// their corner weights are summed over each run with a segmented wave scan, and only the last lane of a run issues the 8
// atomics ("wavefront atomics": one per cell and wave instead of one per particle).  Slower than the plain kernel when cells hold
// less than one particle (the sort costs what the atomics do), so PMesh::force times both and keeps the faster one.
__global__ void __launch_bounds__(256) k_cell_keys(int64_t n, const double *__restrict__ pos, const uint8_t *__restrict__ active, double cellsize,
                                                   int nmesh, unsigned long long *__restrict__ keys, int *__restrict__ idx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    unsigned long long lin = 0;
#pragma unroll
    for(int k = 0; k < 3; k++)
        lin = lin * (unsigned long long)nmesh + (unsigned long long)wrap((int)floor(pos[3 * i + k] / cellsize), nmesh);
    keys[i] = (active && !active[i]) ? ~0ull : lin; // inactive particles sort to the end and are skipped
    idx[i] = (int)i;
}

__global__ void __launch_bounds__(256) k_cic_deposit_sorted(int64_t n, const unsigned long long *__restrict__ skeys, const int *__restrict__ sidx,
                                                            const double *__restrict__ pos, const float *__restrict__ mass, double cellsize,
                                                            int nmesh, int x0, int P, double *__restrict__ mesh)
{
    // mesh holds the x-planes [x0, x0 + P) (all of them on one GPU: x0 = 0, P = nmesh); corners on other planes are skipped
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    unsigned long long key = ~0ull;
    double w[8];
    int ic[3] = {0, 0, 0};
#pragma unroll
    for(int c = 0; c < 8; c++)
        w[c] = 0;
    if(k < n) {
        key = skeys[k];
        if(key != ~0ull) {
            const int i = sidx[k];
            double res[3];
#pragma unroll
            for(int d = 0; d < 3; d++) {
                const double tmp = pos[3 * (int64_t)i + d] / cellsize;
                const double fl = floor(tmp);
                ic[d] = (int)fl;
                res[d] = tmp - fl;
            }
            const double m = (double)mass[i];
#pragma unroll
            for(int c = 0; c < 8; c++) {
                double x = m;
#pragma unroll
                for(int d = 0; d < 3; d++)
                    x *= ((c >> d) & 1) ? res[d] : (1 - res[d]);
                w[c] = x;
            }
        }
    }
    // segmented inclusive scan over the lanes of the wave (segments = runs of equal keys)
    const unsigned long long prev = __shfl_up(key, 1);
    bool f = lane == 0 || prev != key; // head of a run (within this wave)
    const bool head_next = __shfl_down(f ? 1 : 0, 1) != 0;
    const bool tail = lane == 63 || head_next;
    for(int d = 1; d < 64; d <<= 1) {
        const bool f2 = __shfl_up(f ? 1 : 0, d) != 0;
        double v2[8];
#pragma unroll
        for(int c = 0; c < 8; c++)
            v2[c] = __shfl_up(w[c], d);
        if(lane >= d && !f) {
#pragma unroll
            for(int c = 0; c < 8; c++)
                w[c] += v2[c];
            f = f2;
        }
    }
    if(tail && key != ~0ull) {
#pragma unroll
        for(int c = 0; c < 8; c++) {
            const int px = wrap(ic[0] + (c & 1), nmesh) - x0;
            if(px < 0 || px >= P)
                continue;
            size_t lin = (size_t)px;
#pragma unroll
            for(int d = 1; d < 3; d++)
                lin = lin * (size_t)nmesh + (size_t)wrap(ic[d] + ((c >> d) & 1), nmesh);
            unsafeAtomicAdd(&mesh[lin], w[c]);
        }
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
// ny rows of ky starting at y0 are held (ny = nmesh, y0 = 0 on one GPU; a ky-slab in the slab-decomposed form).
// XLAST: the array is [ky local][kz][kx] (kx fastest: the slab form after its transpose) instead of [kx][ky local][kz].
template <bool XLAST>
__global__ void __launch_bounds__(256) k_potential_transfer(int nmesh, int ny, int y0, double asmth2, double pot_factor,
                                                            const double *__restrict__ invsinc2, double2 *__restrict__ cplx)
{
    const int nz = nmesh / 2 + 1;
    const size_t total = (size_t)nmesh * ny * nz;
    const size_t ip = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(ip >= total)
        return;
    int ix, iy, iz;
    if(XLAST) {
        ix = (int)(ip % nmesh);
        const size_t j = ip / nmesh;
        iz = (int)(j % nz);
        iy = y0 + (int)(j / nz);
    }
    else {
        iz = (int)(ip % nz);
        const size_t t = ip / nz;
        iy = y0 + (int)(t % ny);
        ix = (int)(t / ny);
    }
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

// measure_power_spectrum + powerspectrum_add_mode, gravpm.c:331-382: per Fourier cell (before the potential transfer touches
// it) m = |delta_k|^2 de-convolved with the CIC window once (invwindow^2), weight 2 except on the kz = 0 and Nyquist planes,
// logarithmic bins in |k| (Nmesh bins up to sqrt(3) Nmesh / 2); the k = 0 mode is the normalisation.  acc = [Power[nbins],
// kk[nbins], Norm], modes[nbins]; the reference's per-thread copies are per-block LDS histograms here.
// FUSE (round 6): the potential transfer of the same cell in the same pass (k_potential_transfer's arithmetic, after the mode has been
// measured: gravpm.c measures before it multiplies) - one read of the 1.07 GB of rho_k instead of two
template <bool XLAST, bool FUSE = false>
__global__ void __launch_bounds__(256) k_power_spectrum(int nmesh, int ny, int y0, const double *__restrict__ invsinc2,
                                                        double2 *__restrict__ cplx, double *__restrict__ acc,
                                                        unsigned long long *__restrict__ modes, double asmth2 = 0, double pot_factor = 0)
{
    extern __shared__ double s_ps[]; // Power[nbins], kk[nbins], then modes[nbins] (u64)
    const int nbins = nmesh;
    double *s_pow = s_ps, *s_kk = s_ps + nbins;
    unsigned long long *s_n = (unsigned long long *)(s_ps + 2 * nbins);
    for(int b = threadIdx.x; b < nbins; b += blockDim.x) {
        s_pow[b] = 0;
        s_kk[b] = 0;
        s_n[b] = 0;
    }
    __syncthreads();
    const int nz = nmesh / 2 + 1;
    const size_t total = (size_t)nmesh * ny * nz;
    const double binsperunit = (nbins - 1) / log(sqrt(3.0) * nmesh / 2.0);
    for(size_t ip = (size_t)blockIdx.x * blockDim.x + threadIdx.x; ip < total; ip += (size_t)gridDim.x * blockDim.x) {
        int ix, iy, iz;
        if(XLAST) {
            ix = (int)(ip % nmesh);
            const size_t j = ip / nmesh;
            iz = (int)(j % nz);
            iy = y0 + (int)(j / nz);
        }
        else {
            iz = (int)(ip % nz);
            const size_t t = ip / nz;
            iy = y0 + (int)(t % ny);
            ix = (int)(t / ny);
        }
        const int kx = ix <= nmesh / 2 ? ix : ix - nmesh;
        const int ky = iy <= nmesh / 2 ? iy : iy - nmesh;
        const int kz = iz;
        const long long k2 = (long long)kx * kx + (long long)ky * ky + (long long)kz * kz;
        const double2 v = cplx[ip];
        const double m = v.x * v.x + v.y * v.y;
        if(k2 == 0) {
            acc[2 * nbins] = m; // Norm
            if(FUSE)
                cplx[ip] = make_double2(0.0, 0.0);
            continue;
        }
        const double f = invsinc2[ix] * invsinc2[iy] * invsinc2[iz];
        if(FUSE) { // (k_potential_transfer, same expressions in the same order.  exp(-k2 asmth2) / k2 and the bin from tables indexed by k2 -
                   // 2.4 MB of them at Nmesh 512 - measured SLOWER than the exp and the log: 1.06 against 0.64 ms, the gathers miss)
            const double smth = exp(-(double)k2 * asmth2) / (double)k2;
            const double fac = pot_factor * smth * f * f;
            cplx[ip] = make_double2(v.x * fac, v.y * fac);
        }
        const int kint = (int)floor(binsperunit * log((double)k2) / 2.);
        if(kint >= nbins)
            continue;
        const int w = (kz == 0 || kz == nmesh / 2) ? 1 : 2;
        __hip_atomic_fetch_add(&s_pow[kint], w * m * f * f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&s_kk[kint], w * sqrt((double)k2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&s_n[kint], (unsigned long long)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    for(int b = threadIdx.x; b < nbins; b += blockDim.x)
        if(s_n[b]) {
            unsafeAtomicAdd(&acc[b], s_pow[b]);
            unsafeAtomicAdd(&acc[nbins + b], s_kk[b]);
            atomicAdd(&modes[b], s_n[b]);
        }
}

// force_transfer for one axis, gravpm.c:476-498: (re, im) <- (-im*fac, re*fac), fac = -diff_kernel(k 2pi/N) N/Box.
// axis < 0: plain copy (the Potential pass has no transfer function, gravpm.c:32-39).
// The destination element of source (ix, row, iz) is dst[(ix * xmul + xoff) * ny * nz + row * nz + iz]: xmul = 1, xoff = 0 on one
// GPU; (4, function) in the slab form, which interleaves the four functions so that each all-to-all block stays contiguous.
template <bool XLAST>
__global__ void __launch_bounds__(256) k_force_transfer(int nmesh, int ny, int y0, int axis, const double *__restrict__ difffac,
                                                        const double2 *__restrict__ src, double2 *__restrict__ dst, int xmul, int xoff)
{
    const int nz = nmesh / 2 + 1;
    const size_t total = (size_t)nmesh * ny * nz;
    const size_t ip = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(ip >= total)
        return;
    double2 v = src[ip];
    const size_t rowsz = (size_t)ny * nz;
    if(XLAST) { // same layout in and out
        if(axis >= 0) {
            const int ix = (int)(ip % nmesh);
            const size_t j = ip / nmesh;
            const int iz = (int)(j % nz), iy = y0 + (int)(j / nz);
            const double fac = difffac[axis == 0 ? ix : (axis == 1 ? iy : iz)];
            const double t0 = -v.y * fac, t1 = v.x * fac;
            v.x = t0;
            v.y = t1;
        }
        dst[ip] = v;
        return;
    }
    const int ix = (int)(ip / rowsz);
    if(axis >= 0) {
        const int iz = (int)(ip % nz);
        const int iy = y0 + (int)((ip / nz) % ny);
        const int ii = axis == 0 ? ix : (axis == 1 ? iy : iz);
        const double fac = difffac[ii];
        const double t0 = -v.y * fac, t1 = v.x * fac;
        v.x = t0;
        v.y = t1;
    }
    dst[((size_t)ix * xmul + xoff) * rowsz + (ip - (size_t)ix * rowsz)] = v;
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
    if(active && !active[i]) // garbage / swallowed: outside every region (gravpm.c:176-179), GravPM stays at the zero of gravpm.c:88-92
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

// Force component along `axis` from the potential mesh by the 4-point central difference
//     F = -[ 2/3 (phi[+1] - phi[-1]) - 1/12 (phi[+2] - phi[-2]) ] N / Box          (periodic)
// This IS the reference's force_transfer (gravpm.c:456-489): its Fourier-space factor i * (-diff_kernel(w)) N/Box,
// diff_kernel(w) = (8 sin w - sin 2w) / 6 ("the same as GADGET-2 but in fourier space: c1 = 2/3, c2 = 1/12"), is the symbol of
// exactly this stencil, so differencing the potential in real space replaces three of the four inverse transforms (and their
// transfer sweeps) by three streaming passes; the results differ from the Fourier-space form by rounding only.
// nplanes / plane0: the x-planes held (all of them on one GPU); for axis 0 in the slab form, planes -2..-1 and P..P+1 are the
// ghost planes stored around the slab (see slab_gradient).
__global__ void __launch_bounds__(256) k_gradient_axis(int nmesh, int nplanes, int axis, double scale, const double *__restrict__ phi,
                                                       double *__restrict__ out, int xghost)
{
    const size_t total = (size_t)nplanes * nmesh * nmesh;
    const size_t ip = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(ip >= total)
        return;
    const int iz = (int)(ip % nmesh);
    const size_t t = ip / nmesh;
    const int iy = (int)(t % nmesh);
    const int ix = (int)(t / nmesh);
    const size_t plane = (size_t)nmesh * nmesh;
    // phi is stored with `xghost` ghost planes below plane 0 (0 on one GPU, where x wraps periodically instead)
    const double *c = phi + ((size_t)(ix + xghost) * nmesh + iy) * nmesh + iz;
    double p1, m1, p2, m2;
    if(axis == 0) {
        if(xghost) {
            p1 = c[plane];
            m1 = c[-(ptrdiff_t)plane];
            p2 = c[2 * plane];
            m2 = c[-2 * (ptrdiff_t)plane];
        }
        else {
            const size_t row = (size_t)iy * nmesh + iz;
            p1 = phi[(size_t)wrap(ix + 1, nmesh) * plane + row];
            m1 = phi[(size_t)wrap(ix - 1, nmesh) * plane + row];
            p2 = phi[(size_t)wrap(ix + 2, nmesh) * plane + row];
            m2 = phi[(size_t)wrap(ix - 2, nmesh) * plane + row];
        }
    }
    else if(axis == 1) {
        const double *r = c - (size_t)iy * nmesh;
        p1 = r[(size_t)wrap(iy + 1, nmesh) * nmesh];
        m1 = r[(size_t)wrap(iy - 1, nmesh) * nmesh];
        p2 = r[(size_t)wrap(iy + 2, nmesh) * nmesh];
        m2 = r[(size_t)wrap(iy - 2, nmesh) * nmesh];
    }
    else {
        const double *r = c - iz;
        p1 = r[wrap(iz + 1, nmesh)];
        m1 = r[wrap(iz - 1, nmesh)];
        p2 = r[wrap(iz + 2, nmesh)];
        m2 = r[wrap(iz - 2, nmesh)];
    }
    out[ip] = -((2.0 / 3.0) * (p1 - m1) - (1.0 / 12.0) * (p2 - m2)) * scale;
}

// The three force components in ONE pass over the potential (single-GPU form: x wraps periodically): the twelve neighbours of a
// cell are read once (the z row from registers of neighbouring lanes' cache lines, the y and x neighbours from L2), three meshes
// are written - 4.3 GB at Nmesh = 512 instead of the 6.4 GB of three k_gradient_axis passes.  Same arithmetic per component.
__global__ void __launch_bounds__(256) k_gradient3(int nmesh, double scale, const double *__restrict__ phi, double *__restrict__ gx,
                                                   double *__restrict__ gy, double *__restrict__ gz)
{
    const size_t total = (size_t)nmesh * nmesh * nmesh;
    const size_t ip = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(ip >= total)
        return;
    const int iz = (int)(ip % nmesh);
    const size_t t = ip / nmesh;
    const int iy = (int)(t % nmesh);
    const int ix = (int)(t / nmesh);
    const size_t plane = (size_t)nmesh * nmesh;
    const size_t row = (size_t)iy * nmesh + iz;
    const double *px = phi + row;
    const double *py = phi + (size_t)ix * plane + iz;
    const double *pz = phi + (size_t)ix * plane + (size_t)iy * nmesh;
    const double c1 = 2.0 / 3.0, c2 = 1.0 / 12.0;
    gx[ip] = -(c1 * (px[(size_t)wrap(ix + 1, nmesh) * plane] - px[(size_t)wrap(ix - 1, nmesh) * plane]) -
               c2 * (px[(size_t)wrap(ix + 2, nmesh) * plane] - px[(size_t)wrap(ix - 2, nmesh) * plane])) * scale;
    gy[ip] = -(c1 * (py[(size_t)wrap(iy + 1, nmesh) * nmesh] - py[(size_t)wrap(iy - 1, nmesh) * nmesh]) -
               c2 * (py[(size_t)wrap(iy + 2, nmesh) * nmesh] - py[(size_t)wrap(iy - 2, nmesh) * nmesh])) * scale;
    gz[ip] = -(c1 * (pz[wrap(iz + 1, nmesh)] - pz[wrap(iz - 1, nmesh)]) - c2 * (pz[wrap(iz + 2, nmesh)] - pz[wrap(iz - 2, nmesh)])) * scale;
}

// readout_potential + readout_force_x/y/z (gravpm.c:491-510, petapm.c:1106-1144) in ONE pass without force meshes (round 4): the force at
// a CIC corner is the 4-point difference of the potential there (k_gradient_axis above: the reference's force_transfer in real space), so
// a particle gathers, per corner, the potential and its 12 stencil neighbours straight from the potential mesh and differences on the
// fly.  104 gathers per particle instead of 32 - but from ONE mesh, with the z neighbours in the same cache line and the lanes of a wave
// (particles come in some spatial order: Peano-Hilbert after a domain decomposition, lattice order in initial conditions) sharing lines -
// and the gradient pass with its 3 x Nmesh^3 stores (4.3 GB moved at Nmesh = 512) is gone: gradient 1.5 ms + four read-outs 1.6 ms ->
// 1.2 ms at 256^3 / 512^3, PM 9.9 -> 7.9 ms.  Same stencil expression, weights and corner order as k_gradient3 + k_cic_readout.
// (Measured against it and not kept, profiles/r04a_experiments: the potential staged in LDS tiles of 16^3 cells + halo, 74 KB per block,
// with the particles grouped by tile first - 2.0 ms + 0.5 ms for the grouping; the same gathers in tile order - 1.6 + 0.5 ms.)
__global__ void __launch_bounds__(256) k_cic_readout_stencil(int64_t n, const double *__restrict__ pos, const uint8_t *__restrict__ active,
                                                             double cellsize, int nmesh, double scale, const double *__restrict__ phi,
                                                             double *__restrict__ gravpm, double *__restrict__ potential)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    if(active && !active[i]) // garbage / swallowed: outside every region (gravpm.c:176-179)
        return;
    size_t wi[3][6]; // wrapped cell indices ic-2 .. ic+3 per axis, times the axis stride
    double res[3];
    const size_t stride[3] = {(size_t)nmesh * nmesh, (size_t)nmesh, 1};
#pragma unroll
    for(int k = 0; k < 3; k++) {
        const double tmp = pos[3 * i + k] / cellsize;
        const double fl = floor(tmp);
        res[k] = tmp - fl;
        const int c = wrap((int)fl, nmesh);
#pragma unroll
        for(int j = 0; j < 6; j++)
            wi[k][j] = (size_t)wrap(c - 2 + j, nmesh) * stride[k];
    }
    const double c1 = 2.0 / 3.0, c2 = 1.0 / 12.0;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
    for(int cc = 0; cc < 8; cc++) { // corner order and weight products as in k_cic_readout: bit 0 -> x, bit 1 -> y, bit 2 -> z
        const int ox = cc & 1, oy = (cc >> 1) & 1, oz = (cc >> 2) & 1;
        double w = 1.0;
        w *= ox ? res[0] : (1 - res[0]);
        w *= oy ? res[1] : (1 - res[1]);
        w *= oz ? res[2] : (1 - res[2]);
        const size_t bx = wi[0][2 + ox], by = wi[1][2 + oy], bz = wi[2][2 + oz];
        a0 += w * phi[bx + by + bz];
        a1 += w * (-(c1 * (phi[wi[0][3 + ox] + by + bz] - phi[wi[0][1 + ox] + by + bz]) - c2 * (phi[wi[0][4 + ox] + by + bz] - phi[wi[0][0 + ox] + by + bz])) * scale);
        a2 += w * (-(c1 * (phi[bx + wi[1][3 + oy] + bz] - phi[bx + wi[1][1 + oy] + bz]) - c2 * (phi[bx + wi[1][4 + oy] + bz] - phi[bx + wi[1][0 + oy] + bz])) * scale);
        a3 += w * (-(c1 * (phi[bx + by + wi[2][3 + oz]] - phi[bx + by + wi[2][1 + oz]]) - c2 * (phi[bx + by + wi[2][4 + oz]] - phi[bx + by + wi[2][0 + oz]])) * scale);
    }
    if(potential)
        potential[i] += a0;
    gravpm[3 * i + 0] = a1;
    gravpm[3 * i + 1] = a2;
    gravpm[3 * i + 2] = a3;
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
    dep_single = DepositState(); // (the faster deposit form is chosen again on the next particle set)
    dep_slab = DepositState();
    // the meshes and the 3-D plans are made by the first gravpm_force (ensure_single): the slab-decomposed form never needs them
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
}

// ---- rocFFT plans (petapm.c:284-357 builds its PFFT plans here) ------------------------------------------------------------------
void FftPlan::create(rocfft_result_placement placement, rocfft_transform_type type, int dims, const size_t *lengths, size_t batch,
                     const size_t *in_strides, size_t in_dist, const size_t *out_strides, size_t out_dist)
{
    static const bool once = (rocfft_setup(), true);
    (void)once;
    destroy();
    rocfft_plan_description desc = nullptr;
    if(in_strides || out_strides) {
        MPG_FFT(rocfft_plan_description_create(&desc));
        const bool real_fwd = type == rocfft_transform_type_real_forward, real_inv = type == rocfft_transform_type_real_inverse;
        const rocfft_array_type in_t = real_fwd ? rocfft_array_type_real : (real_inv ? rocfft_array_type_hermitian_interleaved : rocfft_array_type_complex_interleaved);
        const rocfft_array_type out_t = real_fwd ? rocfft_array_type_hermitian_interleaved : (real_inv ? rocfft_array_type_real : rocfft_array_type_complex_interleaved);
        MPG_FFT(rocfft_plan_description_set_data_layout(desc, in_t, out_t, nullptr, nullptr, in_strides ? (size_t)dims : 0, in_strides, in_dist,
                                                        out_strides ? (size_t)dims : 0, out_strides, out_dist));
    }
    MPG_FFT(rocfft_plan_create(&plan, placement, type, rocfft_precision_double, (size_t)dims, lengths, batch, desc));
    if(desc)
        MPG_FFT(rocfft_plan_description_destroy(desc));
    MPG_FFT(rocfft_execution_info_create(&info));
    size_t wb = 0;
    MPG_FFT(rocfft_plan_get_work_buffer_size(plan, &wb));
    if(wb > 0) {
        work.reserve(wb + 64);
        MPG_FFT(rocfft_execution_info_set_work_buffer(info, work.p, wb));
    }
}

void FftPlan::exec(void *in, void *out, hipStream_t st)
{
    MPG_FFT(rocfft_execution_info_set_stream(info, st));
    void *ib[1] = {in}, *ob[1] = {out};
    MPG_FFT(rocfft_execute(plan, ib, out == in ? nullptr : ob, info));
}

void FftPlan::destroy()
{
    if(info)
        (void)rocfft_execution_info_destroy(info);
    if(plan)
        (void)rocfft_plan_destroy(plan);
    info = nullptr;
    plan = nullptr;
    work.release();
}

void PMesh::ensure_single()
{
    if(have_plans)
        return;
    const size_t nreal = (size_t)nmesh * nmesh * nmesh;
    const size_t ncplx = (size_t)nmesh * nmesh * (nmesh / 2 + 1);
    real.reserve(nreal);
    rho_k.reserve(2 * ncplx);
    work_k.reserve(2 * ncplx);
    // x slowest, z fastest: rocFFT takes the lengths fastest first; contiguous real mesh <-> Nmesh^2 (Nmesh/2 + 1) Hermitian half
    const size_t len3[3] = {(size_t)nmesh, (size_t)nmesh, (size_t)nmesh};
    plan_r2c.create(rocfft_placement_notinplace, rocfft_transform_type_real_forward, 3, len3, 1);
    plan_c2r.create(rocfft_placement_notinplace, rocfft_transform_type_real_inverse, 3, len3, 1);
    have_plans = true;
}

void PMesh::ps_zero(hipStream_t st)
{
    ps_acc.reserve(2 * (size_t)nmesh + 8);
    ps_modes.reserve((size_t)nmesh + 8);
    MPG_HIP(hipMemsetAsync(ps_acc.p, 0, (2 * (size_t)nmesh + 1) * sizeof(double), st));
    MPG_HIP(hipMemsetAsync(ps_modes.p, 0, (size_t)nmesh * sizeof(unsigned long long), st));
    ps_valid = true;
}

void PMesh::destroy()
{
    if(have_plans) {
        plan_r2c.destroy();
        plan_c2r.destroy();
        have_plans = false;
    }
    slab_destroy();
    real.release();
    rho_k.release();
    work_k.release();
    grad_z.release();
    nmesh = 0;
}

__global__ void k_cic_deposit_slab(int64_t n, const double *__restrict__ pos, const float *__restrict__ mass, double cellsize, int nmesh, int x0,
                                   int P, double *__restrict__ slab);

// CIC deposit onto the x-planes [x0, x0 + P) of `mesh` (zeroed by the caller): plain atomics, or cell-sorted with wave-aggregated
// atomics; the first call and every 64th time both forms on the set at hand and keep the faster one.
void PMesh::deposit(int64_t n, const double *d_pos, const float *d_mass, const uint8_t *d_active, double *mesh, int x0, int P, DepositState &ds,
                    hipStream_t st, EventTimer *tm)
{
    const size_t ncell = (size_t)P * nmesh * nmesh, nreal = (size_t)nmesh * nmesh * nmesh;
    const bool whole = x0 == 0 && P == nmesh;
    auto plain = [&]() {
        if(whole)
            hipLaunchKernelGGL(k_cic_deposit, dim3(nblk(n)), dim3(256), 0, st, n, d_pos, d_mass, d_active, cellsize, nmesh, mesh);
        else
            hipLaunchKernelGGL(k_cic_deposit_slab, dim3(nblk(n)), dim3(256), 0, st, n, d_pos, d_mass, cellsize, nmesh, x0, P, mesh);
    };
    auto sorted = [&]() {
        dep_keys_a.reserve((size_t)n + 1);
        dep_keys_b.reserve((size_t)n + 1);
        dep_idx_a.reserve((size_t)n + 1);
        dep_idx_b.reserve((size_t)n + 1);
        hipLaunchKernelGGL(k_cell_keys, dim3(nblk(n)), dim3(256), 0, st, n, d_pos, d_active, cellsize, nmesh, dep_keys_a.p, dep_idx_a.p);
        int bits = 1;
        while(bits < 64 && ((unsigned long long)1 << bits) < (unsigned long long)nreal)
            bits++;
        size_t tb = 0;
        MPG_HIP(rocprim::radix_sort_pairs(nullptr, tb, dep_keys_a.p, dep_keys_b.p, dep_idx_a.p, dep_idx_b.p, (size_t)n, 0, 64, st));
        dep_tmp.reserve(tb + 16);
        // (all 64 bits when there are inactive particles: they carry the all-ones key)
        MPG_HIP(rocprim::radix_sort_pairs((void *)dep_tmp.p, tb, dep_keys_a.p, dep_keys_b.p, dep_idx_a.p, dep_idx_b.p, (size_t)n, 0,
                                          d_active ? 64 : bits, st));
        hipLaunchKernelGGL(k_cic_deposit_sorted, dim3(nblk(n)), dim3(256), 0, st, n, dep_keys_b.p, dep_idx_b.p, d_pos, d_mass, cellsize, nmesh, x0, P,
                           mesh);
    };
    if(const char *e = getenv("MPG_PM_DEPOSIT")) // experiment knob: "plain" / "sorted"
        ds.mode = !strcmp(e, "sorted") ? 2 : 1;
    if(ds.mode == 0 || ++ds.since_tune >= 64) {
        hipEvent_t e0, e1, e2;
        MPG_HIP(hipEventCreate(&e0));
        MPG_HIP(hipEventCreate(&e1));
        MPG_HIP(hipEventCreate(&e2));
        MPG_HIP(hipEventRecord(e0, st));
        sorted();
        MPG_HIP(hipEventRecord(e1, st));
        MPG_HIP(hipMemsetAsync(mesh, 0, ncell * sizeof(double), st));
        plain();
        MPG_HIP(hipEventRecord(e2, st));
        MPG_HIP(hipEventSynchronize(e2));
        float ts = 0, tp = 0;
        MPG_HIP(hipEventElapsedTime(&ts, e0, e1));
        MPG_HIP(hipEventElapsedTime(&tp, e1, e2));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipEventDestroy(e2);
        ds.mode = ts < tp ? 2 : 1;
        ds.since_tune = 0;
        if(tm)
            tm->start(st); // the trial is not the deposit's time
        MPG_HIP(hipMemsetAsync(mesh, 0, ncell * sizeof(double), st));
    }
    if(ds.mode == 2)
        sorted();
    else
        plain();
}

void PMesh::force(int64_t n, const double *d_pos, const float *d_mass, const uint8_t *d_active, double *d_gravpm, double *d_potential,
                  hipStream_t st, EventTimer *tm)
{
    MPG_CHECK(nmesh > 0, "gravpm_force called before gravpm_init_periodic");
    MPG_CHECK(!slab.ready, "gravpm_force: the mesh is in its slab-decomposed form (use the pm_slab calls)");
    ensure_single();
    const size_t nreal = (size_t)nmesh * nmesh * nmesh;
    const size_t ncplx = (size_t)nmesh * nmesh * (nmesh / 2 + 1);
    float t_fft = 0, t_tr = 0, t_ro = 0, t;
    if(tm)
        tm->start(st);
    // pm_init_regions zeroes the mesh (petapm.c:932-952); deposit
    MPG_HIP(hipMemsetAsync(real.p, 0, nreal * sizeof(double), st));
    if(n > 0)
        deposit(n, d_pos, d_mass, d_active, real.p, 0, nmesh, dep_single, st, tm);
    if(tm)
        tm->lap(st, &tm->t.pm_deposit);
    plan_r2c.exec(real.p, rho_k.p, st);
    if(tm) {
        tm->lap(st, &t);
        t_fft += t;
    }
    const double asmth2 = pow((2 * M_PI) * Asmth / nmesh, 2);
    const double pot_factor = -G / (M_PI * box);
    static const bool fuse_ps = !(getenv("MPG_PM_FUSE_PS") && getenv("MPG_PM_FUSE_PS")[0] == '0');
    if(measure_power && fuse_ps) { // P(k) and the potential transfer in one pass over rho_k (round 6)
        ps_zero(st);
        hipLaunchKernelGGL((k_power_spectrum<false, true>), dim3(2048), dim3(256), ps_lds_bytes(), st, nmesh, nmesh, 0, invsinc2.p, (double2 *)rho_k.p,
                           ps_acc.p, ps_modes.p, asmth2, pot_factor);
    }
    else {
        if(measure_power) {
            ps_zero(st);
            hipLaunchKernelGGL((k_power_spectrum<false, false>), dim3(2048), dim3(256), ps_lds_bytes(), st, nmesh, nmesh, 0, invsinc2.p,
                               (double2 *)rho_k.p, ps_acc.p, ps_modes.p, 0.0, 0.0);
        }
        hipLaunchKernelGGL(k_potential_transfer<false>, dim3(nblk(ncplx)), dim3(256), 0, st, nmesh, nmesh, 0, asmth2, pot_factor, invsinc2.p,
                           (double2 *)rho_k.p);
    }
    if(tm) {
        tm->lap(st, &t);
        t_tr += t;
    }
    // functions[] = Potential, ForceX, ForceY, ForceZ (gravpm.c:32-39).  Default: one inverse transform (the potential), the
    // forces by differencing it in real space (k_gradient_axis: the same operator as force_transfer); kspace_force restores
    // the reference's four inverse transforms.
    if(!kspace_force) {
        plan_c2r.exec(rho_k.p, real.p, st); // rho_k is consumed: it is not needed again
        if(tm) {
            tm->lap(st, &t);
            t_fft += t;
        }
        static const bool one_pass = !(getenv("MPG_PM_GRADIENT_PASSES") && getenv("MPG_PM_GRADIENT_PASSES")[0] == '3');
        static const bool stencil = one_pass && !(getenv("MPG_PM_STENCIL") && getenv("MPG_PM_STENCIL")[0] == '0');
        if(stencil) { // potential and forces in one read-out pass straight from the potential mesh (k_cic_readout_stencil)
            if(n > 0)
                hipLaunchKernelGGL(k_cic_readout_stencil, dim3(nblk(n)), dim3(256), 0, st, n, d_pos, d_active, cellsize, nmesh, (double)nmesh / box,
                                   (const double *)real.p, d_gravpm, d_potential);
            if(tm) {
                tm->lap(st, &t);
                t_ro += t;
            }
        }
        else {
        if(n > 0 && d_potential)
            hipLaunchKernelGGL(k_cic_readout, dim3(nblk(n)), dim3(256), 0, st, n, d_pos, d_active, cellsize, nmesh, real.p, 3, d_potential);
        if(tm) {
            tm->lap(st, &t);
            t_ro += t;
        }
        if(one_pass) { // the Fourier buffers are free now (Z2D consumed rho_k): they hold two of the three force meshes
            grad_z.reserve(nreal);
            double *g[3] = {work_k.p, rho_k.p, grad_z.p};
            hipLaunchKernelGGL(k_gradient3, dim3(nblk(nreal)), dim3(256), 0, st, nmesh, (double)nmesh / box, real.p, g[0], g[1], g[2]);
            if(tm) {
                tm->lap(st, &t);
                t_tr += t;
            }
            for(int axis = 0; axis < 3 && n > 0; axis++)
                hipLaunchKernelGGL(k_cic_readout, dim3(nblk(n)), dim3(256), 0, st, n, d_pos, d_active, cellsize, nmesh, g[axis], axis, d_gravpm);
            if(tm) {
                tm->lap(st, &t);
                t_ro += t;
            }
        }
        else
            for(int axis = 0; axis < 3; axis++) {
                hipLaunchKernelGGL(k_gradient_axis, dim3(nblk(nreal)), dim3(256), 0, st, nmesh, nmesh, axis, (double)nmesh / box, real.p, work_k.p, 0);
                if(tm) {
                    tm->lap(st, &t);
                    t_tr += t;
                }
                if(n > 0)
                    hipLaunchKernelGGL(k_cic_readout, dim3(nblk(n)), dim3(256), 0, st, n, d_pos, d_active, cellsize, nmesh, work_k.p, axis, d_gravpm);
                if(tm) {
                    tm->lap(st, &t);
                    t_ro += t;
                }
            }
        }
    }
    else
        for(int f = 0; f < 4; f++) {
        const int axis = f - 1;
        if(f == 0 && !d_potential)
            continue;
        hipLaunchKernelGGL(k_force_transfer<false>, dim3(nblk(ncplx)), dim3(256), 0, st, nmesh, nmesh, 0, axis, difffac.p,
                           (const double2 *)rho_k.p, (double2 *)work_k.p, 1, 0);
        if(tm) {
            tm->lap(st, &t);
            t_tr += t;
        }
        plan_c2r.exec(work_k.p, real.p, st);
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

// ================================================================================================ slab-decomposed form
// Reference: petapm.c lays the mesh out in 2-D pencils over all ranks and moves particles' "region" meshes to them and back
// (petapm.c:584-885); PFFT transposes between the three 1-D transform stages.  With <= 8 GPUs on one node a 1-D (slab)
// decomposition needs one transpose per 3-D transform and keeps every message large (xGMI is point-to-point: 7 peers x one
// contiguous block each).  Every rank holds all particle positions, so instead of exchanging region
// meshes each rank deposits, straight into its own planes, the part of every particle's CIC cloud that falls on them, and
// reads forces back for the particles whose base cell lies in its slab (one ghost plane from the next rank).
//
//   forward_a : deposit -> 2-D r2c over (y,z) of the P own planes -> pack by destination ky-slab     [sendA]
//   all-to-all (caller)                                                                               [recvA = [x][ky local][kz]]
//   forward_b : tiled transpose to [ky local][kz][kx] -> 1-D c2c along x (contiguous rows) -> potential transfer -> inverse 1-D
//               c2c -> tiled transpose back to [x][ky local][kz] (the block for rank d, its x-planes, is contiguous)  [sendB]
//   all-to-all (caller)                                                                               [recvB]
//   inverse_c : unpack to [x local][ky][kz] -> 2-D c2r -> the potential slab; its first 3 / last 2 planes out as ghosts [ghost_send]
//   neighbour exchange (caller) -> readout: forces by differencing the potential (k_gradient_axis), CIC readout.
// rocFFT transforms are unnormalised like PFFT's; the three 1-D stages compose to the same 3-D DFT.

__global__ void __launch_bounds__(256) k_cic_deposit_slab(int64_t n, const double *__restrict__ pos, const float *__restrict__ mass,
                                                          double cellsize, int nmesh, int x0, int P, double *__restrict__ slab)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    const double tx = pos[3 * i + 0] / cellsize;
    const double fx = floor(tx);
    const int ix = (int)fx;
    const int p0 = wrap(ix, nmesh) - x0, p1 = wrap(ix + 1, nmesh) - x0; // planes relative to the slab
    const bool in0 = p0 >= 0 && p0 < P, in1 = p1 >= 0 && p1 < P;
    if(!in0 && !in1)
        return;
    const double rx = tx - fx;
    int ic[2];
    double res[2];
#pragma unroll
    for(int k = 0; k < 2; k++) {
        const double tmp = pos[3 * i + 1 + k] / cellsize;
        const double fl = floor(tmp);
        ic[k] = (int)fl;
        res[k] = tmp - fl;
    }
    const double m = (double)mass[i];
#pragma unroll
    for(int c = 0; c < 8; c++) {
        const int offx = c & 1; // same corner order and weight product order as k_cic_deposit
        if(offx ? !in1 : !in0)
            continue;
        double w = offx ? rx : (1 - rx);
        size_t lin = (size_t)(offx ? p1 : p0);
#pragma unroll
        for(int k = 0; k < 2; k++) {
            const int off = (c >> (k + 1)) & 1;
            lin = lin * (size_t)nmesh + (size_t)wrap(ic[k] + off, nmesh);
            w *= off ? res[k] : (1 - res[k]);
        }
        unsafeAtomicAdd(&slab[lin], w * m);
    }
}

// C[xl][y][z] -> sendA[d][xl][yl][z], d = y / Py
__global__ void __launch_bounds__(256) k_slab_pack_a(int nmesh, int P, int Py, const double2 *__restrict__ C, double2 *__restrict__ sendA)
{
    const int nz = nmesh / 2 + 1;
    const size_t total = (size_t)P * nmesh * nz;
    const size_t ip = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(ip >= total)
        return;
    const int iz = (int)(ip % nz);
    const size_t t = ip / nz;
    const int y = (int)(t % nmesh);
    const int xl = (int)(t / nmesh);
    const int d = y / Py, yl = y - d * Py;
    sendA[(((size_t)d * P + xl) * Py + yl) * nz + iz] = C[ip];
}

// recvB[s][xl][yl][z] -> C[xl][y = s Py + yl][z]
__global__ void __launch_bounds__(256) k_slab_unpack_b(int nmesh, int P, int Py, const double2 *__restrict__ recvB, double2 *__restrict__ C)
{
    const int nz = nmesh / 2 + 1;
    const size_t total = (size_t)P * nmesh * nz;
    const size_t ip = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(ip >= total)
        return;
    const int iz = (int)(ip % nz);
    const size_t t = ip / nz;
    const int y = (int)(t % nmesh);
    const int xl = (int)(t / nmesh);
    const int s = y / Py, yl = y - s * Py;
    C[ip] = recvB[(((size_t)s * P + xl) * Py + yl) * nz + iz];
}

// readout for a list of targets whose base cell lies in the slab; plane P of the slab is the ghost (first plane of the next rank)
__global__ void __launch_bounds__(256) k_cic_readout_slab(int64_t nt, const int *__restrict__ targets, const double *__restrict__ pos,
                                                          double cellsize, int nmesh, int x0, int P, const double *__restrict__ slab, int comp,
                                                          double *__restrict__ out, unsigned *__restrict__ err)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= nt)
        return;
    const int64_t i = targets[t];
    const double tx = pos[3 * i + 0] / cellsize;
    const double fx = floor(tx);
    const int px = wrap((int)fx, nmesh) - x0;
    if(px < 0 || px >= P) { // the caller's target list is not this rank's slab
        atomicExch(err, 1u);
        return;
    }
    const double rx = tx - fx;
    int ic[2];
    double res[2];
#pragma unroll
    for(int k = 0; k < 2; k++) {
        const double tmp = pos[3 * i + 1 + k] / cellsize;
        const double fl = floor(tmp);
        ic[k] = (int)fl;
        res[k] = tmp - fl;
    }
    double acc = 0;
#pragma unroll
    for(int c = 0; c < 8; c++) {
        const int offx = c & 1;
        double w = offx ? rx : (1 - rx);
        size_t lin = (size_t)(px + offx);
#pragma unroll
        for(int k = 0; k < 2; k++) {
            const int off = (c >> (k + 1)) & 1;
            lin = lin * (size_t)nmesh + (size_t)wrap(ic[k] + off, nmesh);
            w *= off ? res[k] : (1 - res[k]);
        }
        acc += w * slab[lin];
    }
    if(comp < 3)
        out[3 * i + comp] = acc;
    else
        out[i] += acc;
}

// out[c * out_ld + r] = in[r * in_ld + c] for r < rows, c < cols (complex doubles), through a 32 x 32 LDS tile so that both the
// reads and the writes are coalesced.  The 1-D transforms along x then run on contiguous rows: rocFFT's strided plan for the
// same transform (stride Py*Nz, batch Py*Nz) measured 2.06 ms against 0.4 ms + 0.5 ms for transpose + contiguous transform.
__global__ void __launch_bounds__(256) k_transpose(int rows, int cols, const double2 *__restrict__ in, size_t in_ld, double2 *__restrict__ out,
                                                   size_t out_ld)
{
    __shared__ double2 tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for(int k = 0; k < 32; k += 8) {
        const int r = r0 + ty + k, c = c0 + tx;
        if(r < rows && c < cols)
            tile[ty + k][tx] = in[(size_t)r * in_ld + c];
    }
    __syncthreads();
#pragma unroll
    for(int k = 0; k < 32; k += 8) {
        const int c = c0 + ty + k, r = r0 + tx;
        if(r < rows && c < cols)
            out[(size_t)c * out_ld + r] = tile[tx][ty + k];
    }
}

void PMesh::slab_destroy()
{
    if(slab.ready) {
        slab.p2d_r2c.destroy();
        slab.p2d_c2r.destroy();
        slab.p1d_fwd.destroy();
        slab.p1d_inv.destroy();
        slab.p1d_fwd_t.destroy();
        slab.p1d_inv_t.destroy();
        slab.ready = false;
    }
    slab.phi.release();
    slab.force.release();
    slab.C.release();
    slab.rho_k.release();
    slab.work.release();
}

void PMesh::slab_init(int rank, int world)
{
    MPG_CHECK(nmesh > 0, "pm_slab_init called before gravpm_init_periodic");
    MPG_CHECK(world >= 1 && rank >= 0 && rank < world, "pm_slab_init: bad rank / world");
    MPG_CHECK(nmesh % world == 0, "pm_slab_init: Nmesh must be a multiple of the number of GPUs");
    slab_destroy();
    slab.rank = rank;
    slab.world = world;
    slab.P = slab.Py = nmesh / world;
    const int nz = nmesh / 2 + 1;
    const size_t S = (size_t)slab.Py * nz;
    MPG_CHECK(slab.P >= 3, "pm_slab_init: at least 3 mesh planes per GPU are needed");
    slab.phi.reserve((size_t)(slab.P + 5) * nmesh * nmesh);
    slab.force.reserve((size_t)(slab.P + 1) * nmesh * nmesh);
    slab.C.reserve(2 * (size_t)slab.P * nmesh * nz);
    slab.rho_k.reserve(2 * (size_t)nmesh * S);
    // the single-GPU meshes and plans are not needed in this form
    if(have_plans) {
        plan_r2c.destroy();
        plan_c2r.destroy();
        have_plans = false;
    }
    real.release();
    rho_k.release();
    work_k.release();
    const size_t len2[2] = {(size_t)nmesh, (size_t)nmesh}, len1[1] = {(size_t)nmesh};
    slab.p2d_r2c.create(rocfft_placement_notinplace, rocfft_transform_type_real_forward, 2, len2, (size_t)slab.P); // the slab's planes
    slab.p2d_c2r.create(rocfft_placement_notinplace, rocfft_transform_type_real_inverse, 2, len2, (size_t)slab.P);
    slab.p1d_fwd.create(rocfft_placement_inplace, rocfft_transform_type_complex_forward, 1, len1, S); // contiguous rows of kx
    slab.p1d_inv.create(rocfft_placement_inplace, rocfft_transform_type_complex_inverse, 1, len1, S);
    // the same transforms straight on the exchange buffers' [x][j] layout (element stride S along x, consecutive j one element apart):
    // transform and transpose in one rocFFT plan each way, instead of k_transpose + a contiguous transform (MPG_PM_STRIDED_FFT=1; an
    // experiment of round 5: profiles/r05a_experiments)
    slab.strided = getenv("MPG_PM_STRIDED_FFT") != nullptr;
    if(slab.strided) {
        const size_t sS[1] = {S}, s1[1] = {1};
        slab.p1d_fwd_t.create(rocfft_placement_notinplace, rocfft_transform_type_complex_forward, 1, len1, S, sS, 1, s1, (size_t)nmesh);
        slab.p1d_inv_t.create(rocfft_placement_notinplace, rocfft_transform_type_complex_inverse, 1, len1, S, s1, (size_t)nmesh, sS, 1);
    }
    slab.work.reserve(2 * (size_t)nmesh * S);
    slab.ready = true;
}

void PMesh::slab_forward_a(int64_t n, const double *d_pos, const float *d_mass, double *sendA, hipStream_t st)
{
    MPG_CHECK(slab.ready, "pm_slab: not initialised");
    const int nz = nmesh / 2 + 1;
    const size_t nreal = (size_t)slab.P * nmesh * nmesh;
    MPG_HIP(hipMemsetAsync(slab.force.p, 0, nreal * sizeof(double), st)); // (the force buffer doubles as the density slab)
    if(n > 0)
        deposit(n, d_pos, d_mass, nullptr, slab.force.p, slab.rank * slab.P, slab.P, dep_slab, st, nullptr);
    slab.p2d_r2c.exec(slab.force.p, slab.C.p, st);
    hipLaunchKernelGGL(k_slab_pack_a, dim3(nblk((size_t)slab.P * nmesh * nz)), dim3(256), 0, st, nmesh, slab.P, slab.Py, (const double2 *)slab.C.p,
                       (double2 *)sendA);
    MPG_HIP(hipGetLastError());
}

void PMesh::slab_forward_b(double *recvA, double *sendB, hipStream_t st)
{
    MPG_CHECK(slab.ready, "pm_slab: not initialised");
    const int nz = nmesh / 2 + 1;
    const size_t ncplx = (size_t)nmesh * slab.Py * nz;
    const int y0 = slab.rank * slab.Py;
    const size_t S = (size_t)slab.Py * nz;
    // [x][j] -> [j][x], j = (ky local, kz): the transforms along x run on contiguous rows
    const dim3 tgrid_f((unsigned)((S + 31) / 32), (unsigned)((nmesh + 31) / 32)), tgrid_b((unsigned)((nmesh + 31) / 32), (unsigned)((S + 31) / 32));
    if(slab.strided)
        slab.p1d_fwd_t.exec(recvA, slab.rho_k.p, st);
    else {
        hipLaunchKernelGGL(k_transpose, tgrid_f, dim3(256), 0, st, nmesh, (int)S, (const double2 *)recvA, S, (double2 *)slab.rho_k.p, (size_t)nmesh);
        slab.p1d_fwd.exec(slab.rho_k.p, slab.rho_k.p, st);
    }
    const double asmth2 = pow((2 * M_PI) * Asmth / nmesh, 2);
    const double pot_factor = -G / (M_PI * box);
    if(measure_power) { // this rank's ky rows: the caller sums the raw accumulators over the ranks (powerspectrum_sum's Allreduce)
        ps_zero(st);
        hipLaunchKernelGGL((k_power_spectrum<true, false>), dim3(1024), dim3(256), ps_lds_bytes(), st, nmesh, slab.Py, y0, invsinc2.p,
                           (double2 *)slab.rho_k.p, ps_acc.p, ps_modes.p, 0.0, 0.0);
    }
    hipLaunchKernelGGL(k_potential_transfer<true>, dim3(nblk(ncplx)), dim3(256), 0, st, nmesh, slab.Py, y0, asmth2, pot_factor, invsinc2.p,
                       (double2 *)slab.rho_k.p);
    // only the potential is transformed back: the forces are its real-space differences (k_gradient_axis), which also cuts
    // the inverse all-to-all to a quarter
    if(slab.strided)
        slab.p1d_inv_t.exec(slab.rho_k.p, sendB, st);
    else {
        slab.p1d_inv.exec(slab.rho_k.p, slab.rho_k.p, st);
        // [j][x] -> sendB[x][j]: the block for rank d (its x-planes) is contiguous
        hipLaunchKernelGGL(k_transpose, tgrid_b, dim3(256), 0, st, (int)S, nmesh, (const double2 *)slab.rho_k.p, (size_t)nmesh, (double2 *)sendB, S);
    }
    MPG_HIP(hipGetLastError());
}

void PMesh::slab_inverse_c(const double *recvB, double *ghost_send, hipStream_t st)
{
    MPG_CHECK(slab.ready, "pm_slab: not initialised");
    const int nz = nmesh / 2 + 1;
    const size_t plane = (size_t)nmesh * nmesh;
    hipLaunchKernelGGL(k_slab_unpack_b, dim3(nblk((size_t)slab.P * nmesh * nz)), dim3(256), 0, st, nmesh, slab.P, slab.Py, (const double2 *)recvB,
                       (double2 *)slab.C.p);
    double *phi0 = slab.phi.p + 2 * plane; // plane 0 of the slab; planes -2, -1 and P .. P+2 are ghosts
    slab.p2d_c2r.exec(slab.C.p, phi0, st);
    // ghosts the neighbours need: the first 3 planes go to the previous rank, the last 2 to the next
    MPG_HIP(hipMemcpyAsync(ghost_send, phi0, 3 * plane * sizeof(double), hipMemcpyDeviceToDevice, st));
    MPG_HIP(hipMemcpyAsync(ghost_send + 3 * plane, phi0 + (size_t)(slab.P - 2) * plane, 2 * plane * sizeof(double), hipMemcpyDeviceToDevice, st));
    MPG_HIP(hipGetLastError());
}

// k_cic_readout_stencil for a slab: potential and forces of the listed targets in one pass from the slab's potential, which is stored with
// two ghost planes below plane 0 and planes P .. P+2 above (x is not wrapped: the neighbours' planes are there; y and z wrap)
__global__ void __launch_bounds__(256) k_cic_readout_slab_stencil(int64_t nt, const int *__restrict__ targets, const double *__restrict__ pos,
                                                                  double cellsize, int nmesh, int x0, int P, double scale,
                                                                  const double *__restrict__ phi /* plane -2 first */, double *__restrict__ gravpm,
                                                                  double *__restrict__ potential, unsigned *__restrict__ err)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= nt)
        return;
    const int64_t i = targets ? targets[t] : t; // (no list: every row whose base cell lies in the slab is a target, the others are skipped)
    size_t wi[3][6];
    double res[3];
    const size_t stride[3] = {(size_t)nmesh * nmesh, (size_t)nmesh, 1};
    {
        const double tmp = pos[3 * i] / cellsize;
        const double fl = floor(tmp);
        res[0] = tmp - fl;
        const int px = wrap((int)fl, nmesh) - x0;
        if(px < 0 || px >= P) { // not this rank's slab: an error in a caller's target list
            if(targets)
                atomicExch(err, 1u);
            return;
        }
#pragma unroll
        for(int j = 0; j < 6; j++)
            wi[0][j] = (size_t)(px + j) * stride[0]; // (px - 2 + j) + 2 ghost planes
    }
#pragma unroll
    for(int k = 1; k < 3; k++) {
        const double tmp = pos[3 * i + k] / cellsize;
        const double fl = floor(tmp);
        res[k] = tmp - fl;
        const int c = wrap((int)fl, nmesh);
#pragma unroll
        for(int j = 0; j < 6; j++)
            wi[k][j] = (size_t)wrap(c - 2 + j, nmesh) * stride[k];
    }
    const double c1 = 2.0 / 3.0, c2 = 1.0 / 12.0;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
    for(int cc = 0; cc < 8; cc++) {
        const int ox = cc & 1, oy = (cc >> 1) & 1, oz = (cc >> 2) & 1;
        double w = ox ? res[0] : (1 - res[0]);
        w *= oy ? res[1] : (1 - res[1]);
        w *= oz ? res[2] : (1 - res[2]);
        const size_t bx = wi[0][2 + ox], by = wi[1][2 + oy], bz = wi[2][2 + oz];
        a0 += w * phi[bx + by + bz];
        a1 += w * (-(c1 * (phi[wi[0][3 + ox] + by + bz] - phi[wi[0][1 + ox] + by + bz]) - c2 * (phi[wi[0][4 + ox] + by + bz] - phi[wi[0][0 + ox] + by + bz])) * scale);
        a2 += w * (-(c1 * (phi[bx + wi[1][3 + oy] + bz] - phi[bx + wi[1][1 + oy] + bz]) - c2 * (phi[bx + wi[1][4 + oy] + bz] - phi[bx + wi[1][0 + oy] + bz])) * scale);
        a3 += w * (-(c1 * (phi[bx + by + wi[2][3 + oz]] - phi[bx + by + wi[2][1 + oz]]) - c2 * (phi[bx + by + wi[2][4 + oz]] - phi[bx + by + wi[2][0 + oz]])) * scale);
    }
    if(potential)
        potential[i] += a0;
    gravpm[3 * i + 0] = a1;
    gravpm[3 * i + 1] = a2;
    gravpm[3 * i + 2] = a3;
}

// slab_readout for ALL rows of d_pos whose base cell lies in the slab (the rows a rank received for its slab: a particle whose CIC cloud
// straddles two slabs was shipped to both, its base cell's owner reads it out); nothing is read back: no target list, no count
void PMesh::slab_readout_rows(const double *ghost_recv, int64_t nrows, const double *d_pos, double *d_gravpm, double *d_potential, hipStream_t st)
{
    MPG_CHECK(slab.ready, "pm_slab: not initialised");
    const size_t plane = (size_t)nmesh * nmesh;
    double *phi0 = slab.phi.p + 2 * plane;
    MPG_HIP(hipMemcpyAsync(phi0 + (size_t)slab.P * plane, ghost_recv, 3 * plane * sizeof(double), hipMemcpyDeviceToDevice, st));
    MPG_HIP(hipMemcpyAsync(slab.phi.p, ghost_recv + 3 * plane, 2 * plane * sizeof(double), hipMemcpyDeviceToDevice, st));
    if(nrows > 0)
        hipLaunchKernelGGL(k_cic_readout_slab_stencil, dim3(nblk(nrows)), dim3(256), 0, st, nrows, (const int *)nullptr, d_pos, cellsize, nmesh,
                           slab.rank * slab.P, slab.P, (double)nmesh / box, (const double *)slab.phi.p, d_gravpm, d_potential, (unsigned *)nullptr);
    MPG_HIP(hipGetLastError());
}

void PMesh::slab_readout(const double *ghost_recv, const int *targets, int64_t nt, const double *d_pos, double *d_gravpm, double *d_potential,
                         hipStream_t st)
{
    MPG_CHECK(slab.ready, "pm_slab: not initialised");
    const size_t plane = (size_t)nmesh * nmesh;
    DevBuf<unsigned> &flag = slab_err;
    flag.reserve(1);
    MPG_HIP(hipMemsetAsync(flag.p, 0, sizeof(unsigned), st));
    double *phi0 = slab.phi.p + 2 * plane;
    // ghost_recv: planes P, P+1, P+2 (the next rank's first three), then planes -2, -1 (the previous rank's last two)
    MPG_HIP(hipMemcpyAsync(phi0 + (size_t)slab.P * plane, ghost_recv, 3 * plane * sizeof(double), hipMemcpyDeviceToDevice, st));
    MPG_HIP(hipMemcpyAsync(slab.phi.p, ghost_recv + 3 * plane, 2 * plane * sizeof(double), hipMemcpyDeviceToDevice, st));
    const int x0 = slab.rank * slab.P;
    static const bool stencil = !(getenv("MPG_PM_STENCIL") && getenv("MPG_PM_STENCIL")[0] == '0');
    if(stencil) {
        if(nt > 0)
            hipLaunchKernelGGL(k_cic_readout_slab_stencil, dim3(nblk(nt)), dim3(256), 0, st, nt, targets, d_pos, cellsize, nmesh, x0, slab.P,
                               (double)nmesh / box, (const double *)slab.phi.p, d_gravpm, d_potential, flag.p);
    }
    else {
    if(nt > 0 && d_potential)
        hipLaunchKernelGGL(k_cic_readout_slab, dim3(nblk(nt)), dim3(256), 0, st, nt, targets, d_pos, cellsize, nmesh, x0, slab.P, phi0, 3, d_potential,
                           flag.p);
    }
    const size_t ncell = (size_t)(slab.P + 1) * plane; // planes 0 .. P: the CIC readout reaches one plane beyond the slab
    for(int axis = 0; axis < 3 && nt > 0 && !stencil; axis++) {
        hipLaunchKernelGGL(k_gradient_axis, dim3(nblk(ncell)), dim3(256), 0, st, nmesh, slab.P + 1, axis, (double)nmesh / box, slab.phi.p,
                           slab.force.p, 2);
        hipLaunchKernelGGL(k_cic_readout_slab, dim3(nblk(nt)), dim3(256), 0, st, nt, targets, d_pos, cellsize, nmesh, x0, slab.P, slab.force.p, axis,
                           d_gravpm, flag.p);
    }
    MPG_HIP(hipGetLastError());
    unsigned e = 0;
    MPG_HIP(hipMemcpyAsync(&e, flag.p, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    MPG_HIP(hipStreamSynchronize(st));
    MPG_CHECK(e == 0, "pm_slab_readout: a target's base cell is outside this rank's slab");
}

} // namespace mpg
