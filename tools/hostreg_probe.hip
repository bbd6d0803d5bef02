// hostreg_probe.hip -- what does the drop-in host path cost on this box, per way of moving a 160-byte-record particle table?
//   hipcc --offload-arch=gfx950 -O3 -fopenmp tools/hostreg_probe.hip -o tools/_bin/hostreg_probe && tools/_bin/hostreg_probe [n]
// Measures, for n records of 160 bytes in PAGEABLE host memory (malloc, first touched by the host threads):
//   (a) hipHostRegister of the table (once per table address) and hipHostUnregister
//   (b) a device kernel GATHERING Pos / Mass / flags / Type straight from the registered records (zero copy, one thread per record)
//   (c) a device kernel SCATTERING GravPM + Potential (32 B) and FullTreeGravAccel + Potential straight into the records
//   (d) the staged way: host threads pack the columns into pinned buffers + hipMemcpyAsync (and back: copy + host threads unpack)
// so that csrc/engine.hip's host forms can take the cheaper one.
#include <hip/hip_runtime.h>
#include <omp.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
constexpr int STRIDE = 160, OFF_POS = 0, OFF_MASS = 28, OFF_FLAGS = 36, OFF_TYPE = 39, OFF_ACC = 64, OFF_GPM = 88, OFF_POT = 152;

__global__ void __launch_bounds__(256) k_gather(int64_t n, const char *__restrict__ base, double *__restrict__ pos, float *__restrict__ mass,
                                                uint8_t *__restrict__ type)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if(i >= n)
        return;
    const char *r = base + i * STRIDE;
    const double *p = (const double *)(r + OFF_POS);
    const double x = p[0], y = p[1], z = p[2];
    const uint2 w = *(const uint2 *)(r + 32); // PI, flags .. Type
    const float m = *(const float *)(r + OFF_MASS);
    pos[3 * i] = x;
    pos[3 * i + 1] = y;
    pos[3 * i + 2] = z;
    mass[i] = m;
    uint8_t ty = (w.y >> 24) & 7;
    if(w.y & 1)
        ty = 7;
    type[i] = ty;
}
// 8 lanes per record-pair?  The simple form: one thread per record, three 8-byte stores per vector
__global__ void __launch_bounds__(256) k_scatter(int64_t n, char *__restrict__ base, const double *__restrict__ v3, const double *__restrict__ pot, int off)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if(i >= n)
        return;
    char *r = base + i * STRIDE;
    double *a = (double *)(r + off);
    a[0] = v3[3 * i];
    a[1] = v3[3 * i + 1];
    a[2] = v3[3 * i + 2];
    *(double *)(r + OFF_POT) = pot[i];
}

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : (int64_t)256 * 256 * 256;
    const size_t bytes = (size_t)n * STRIDE;
    printf("records %lld x %d B = %.2f GB, host threads %d\n", (long long)n, STRIDE, bytes / 1e9, omp_get_max_threads());
    char *P = (char *)aligned_alloc(64, bytes);
#pragma omp parallel for schedule(static)
    for(int64_t i = 0; i < n; i++) {
        memset(P + i * STRIDE, 0, STRIDE);
        double *p = (double *)(P + i * STRIDE);
        p[0] = i;
        p[1] = 2 * i;
        p[2] = 3 * i;
        *(float *)(P + i * STRIDE + OFF_MASS) = 1.f;
        P[i * STRIDE + OFF_TYPE] = 1;
    }
    double *d_pos, *d_v3, *d_pot;
    float *d_mass;
    uint8_t *d_type;
    CK(hipMalloc(&d_pos, 24 * n));
    CK(hipMalloc(&d_v3, 24 * n));
    CK(hipMalloc(&d_pot, 8 * n));
    CK(hipMalloc(&d_mass, 4 * n));
    CK(hipMalloc(&d_type, n));
    CK(hipMemset(d_v3, 0, 24 * n));
    CK(hipMemset(d_pot, 0, 8 * n));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const unsigned nb = (unsigned)((n + 255) / 256);
    // (a)
    double t0 = now();
    hipError_t e = hipHostRegister(P, bytes, hipHostRegisterDefault);
    double t1 = now();
    printf("(a) hipHostRegister: %s, %.1f ms\n", hipGetErrorString(e), t1 - t0);
    if(e == hipSuccess) {
        char *dP = nullptr;
        CK(hipHostGetDevicePointer((void **)&dP, P, 0));
        for(int rep = 0; rep < 3; rep++) {
            t0 = now();
            hipLaunchKernelGGL(k_gather, dim3(nb), dim3(256), 0, st, n, dP, d_pos, d_mass, d_type);
            CK(hipStreamSynchronize(st));
            t1 = now();
            printf("(b) zero-copy gather of Pos/Mass/Type: %.2f ms (%.1f GB/s useful, 29 B per record)\n", t1 - t0, 29.0 * n / (t1 - t0) / 1e6);
        }
        for(int rep = 0; rep < 3; rep++) {
            t0 = now();
            hipLaunchKernelGGL(k_scatter, dim3(nb), dim3(256), 0, st, n, dP, d_v3, d_pot, OFF_GPM);
            CK(hipStreamSynchronize(st));
            t1 = now();
            printf("(c) zero-copy scatter of GravPM + Potential: %.2f ms (%.1f GB/s useful, 32 B per record)\n", t1 - t0, 32.0 * n / (t1 - t0) / 1e6);
        }
        t0 = now();
        CK(hipHostUnregister(P));
        printf("    hipHostUnregister %.1f ms\n", now() - t0);
    }
    // (d) staged
    double *h_pos, *h_v3, *h_pot;
    float *h_mass;
    uint8_t *h_type;
    CK(hipHostMalloc(&h_pos, 24 * n));
    CK(hipHostMalloc(&h_v3, 24 * n));
    CK(hipHostMalloc(&h_pot, 8 * n));
    CK(hipHostMalloc(&h_mass, 4 * n));
    CK(hipHostMalloc(&h_type, n));
    for(int rep = 0; rep < 3; rep++) {
        t0 = now();
#pragma omp parallel for schedule(static)
        for(int64_t i = 0; i < n; i++) {
            const char *r = P + i * STRIDE;
            const double *p = (const double *)r;
            h_pos[3 * i] = p[0];
            h_pos[3 * i + 1] = p[1];
            h_pos[3 * i + 2] = p[2];
            h_mass[i] = *(const float *)(r + OFF_MASS);
            h_type[i] = r[OFF_TYPE] & 7;
        }
        t1 = now();
        CK(hipMemcpyAsync(d_pos, h_pos, 24 * n, hipMemcpyHostToDevice, st));
        CK(hipMemcpyAsync(d_mass, h_mass, 4 * n, hipMemcpyHostToDevice, st));
        CK(hipMemcpyAsync(d_type, h_type, n, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double t2 = now();
        printf("(d) staged up: host pack %.2f ms + H2D %.2f ms\n", t1 - t0, t2 - t1);
    }
    for(int rep = 0; rep < 3; rep++) {
        t0 = now();
        CK(hipMemcpyAsync(h_v3, d_v3, 24 * n, hipMemcpyDeviceToHost, st));
        CK(hipMemcpyAsync(h_pot, d_pot, 8 * n, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        t1 = now();
#pragma omp parallel for schedule(static)
        for(int64_t i = 0; i < n; i++) {
            char *r = P + i * STRIDE;
            double *a = (double *)(r + OFF_GPM);
            a[0] = h_v3[3 * i];
            a[1] = h_v3[3 * i + 1];
            a[2] = h_v3[3 * i + 2];
            *(double *)(r + OFF_POT) = h_pot[i];
        }
        double t2 = now();
        printf("(d) staged down: D2H %.2f ms + host unpack %.2f ms\n", t1 - t0, t2 - t1);
    }
    return 0;
}
