// rsq_acc.hip -- accuracy of v_rsq_f64 on gfx950 and of the Newton refinements the walk's pair evaluation could use.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/rsq_acc tools/rsq_acc.hip ; run on the GPU box.  Prints the largest
// relative error (in units of 2^-53) of: the raw instruction, one quadratic step (4 instructions), one cubic step (5).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

__global__ void k(const double *x, double *raw, double *quad, double *cub, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    const double v = x[i];
    const double y = __builtin_amdgcn_rsq(v);
    raw[i] = y;
    const double e = fma(-(v * y), y, 1.0);
    quad[i] = fma(y * e, 0.5, y);
    cub[i] = fma(y * e, fma(e, 0.375, 0.5), y);
}

int main()
{
    const int n = 1 << 24;
    std::vector<double> x(n);
    unsigned long long s = 88172645463325252ull;
    for(int i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;
        x[i] = ldexp(1.0 + u, (int)((s >> 3) % 80) - 40); // mantissas uniform in [1, 2), exponents -40 .. 39
    }
    double *dx, *d0, *d1, *d2;
    CHK(hipMalloc(&dx, n * 8)); CHK(hipMalloc(&d0, n * 8)); CHK(hipMalloc(&d1, n * 8)); CHK(hipMalloc(&d2, n * 8));
    CHK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    CHK(hipDeviceSynchronize());
    std::vector<double> r0(n), r1(n), r2(n);
    CHK(hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost));
    long double m0 = 0, m1 = 0, m2 = 0;
    for(int i = 0; i < n; i++) {
        const long double t = 1.0L / sqrtl((long double)x[i]);
        m0 = fmaxl(m0, fabsl(r0[i] - t) / t);
        m1 = fmaxl(m1, fabsl(r1[i] - t) / t);
        m2 = fmaxl(m2, fabsl(r2[i] - t) / t);
    }
    const long double u = ldexpl(1.0L, -53);
    printf("v_rsq_f64 max rel err %.3Le (%.1Lf x 2^-53 = 2^%.1Lf)\n", m0, m0 / u, log2l(m0));
    printf("quadratic step        %.3Le (%.2Lf x 2^-53)\n", m1, m1 / u);
    printf("cubic step            %.3Le (%.2Lf x 2^-53)\n", m2, m2 / u);
    return 0;
}
