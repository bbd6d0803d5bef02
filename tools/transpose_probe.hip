// transpose_probe.hip -- variants of the slab PM's complex-double transpose (csrc/pm.hip k_transpose) at its real shape:
// rows = Nmesh = 512 (x), cols = S = Py * (Nmesh / 2 + 1) = 512 * 257 (one rank) : in[r * S + c] -> out[c * rows + r]
// hipcc --offload-arch=gfx950 -O3 -o tools/_bin/transpose_probe tools/transpose_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

// the kernel of csrc/pm.hip: 32 x 32 tile, 256 threads
__global__ void __launch_bounds__(256) t32(int rows, int cols, const double2 *__restrict__ in, size_t in_ld, double2 *__restrict__ out, size_t out_ld)
{
    __shared__ double2 tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
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

// TR x TC tile (rows x cols of the input), 256 threads; reads TC-wide segments, writes TR-wide segments
template <int TR, int TC>
__global__ void __launch_bounds__(256) tt(int rows, int cols, const double2 *__restrict__ in, size_t in_ld, double2 *__restrict__ out, size_t out_ld)
{
    __shared__ double2 tile[TR][TC + 1];
    const int c0 = blockIdx.x * TC, r0 = blockIdx.y * TR;
    for(int e = threadIdx.x; e < TR * TC; e += 256) {
        const int lr = e / TC, lc = e % TC;
        const int r = r0 + lr, c = c0 + lc;
        if(r < rows && c < cols)
            tile[lr][lc] = in[(size_t)r * in_ld + c];
    }
    __syncthreads();
    for(int e = threadIdx.x; e < TR * TC; e += 256) {
        const int lc = e / TR, lr = e % TR;
        const int r = r0 + lr, c = c0 + lc;
        if(r < rows && c < cols)
            out[(size_t)c * out_ld + r] = tile[lr][lc];
    }
}

// no LDS: every thread moves one element, consecutive threads along the OUTPUT (coalesced writes, strided reads through L2)
__global__ void __launch_bounds__(256) tw(int rows, int cols, const double2 *__restrict__ in, size_t in_ld, double2 *__restrict__ out, size_t out_ld)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if(i >= (size_t)rows * cols)
        return;
    const int r = (int)(i % rows);
    const size_t c = i / rows;
    out[c * out_ld + r] = in[(size_t)r * in_ld + c];
}

template <class F> static float timeit(F f)
{
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    f();
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    for(int i = 0; i < 5; i++)
        f();
    CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms;
    CHK(hipEventElapsedTime(&ms, a, b));
    return ms / 5;
}

int main()
{
    const int rows = 512;
    const int cols = 512 * 257;
    const size_t n = (size_t)rows * cols;
    double2 *in, *out;
    CHK(hipMalloc(&in, n * sizeof(double2)));
    CHK(hipMalloc(&out, n * sizeof(double2)));
    CHK(hipMemset(in, 1, n * sizeof(double2)));
    const double gb = 2.0 * n * sizeof(double2) / 1e9;
#define RUN(name, call)                                                     \
    {                                                                       \
        const float ms = timeit([&] { call; });                            \
        printf("%-22s %7.3f ms  %6.2f TB/s\n", name, ms, gb / ms);          \
    }
    // forward: in[rows][cols] -> out[cols][rows]
    RUN("fwd 32x32 (pm.hip)", hipLaunchKernelGGL(t32, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, 0, rows, cols, in, (size_t)cols, out, (size_t)rows));
    RUN("fwd 64x32", hipLaunchKernelGGL((tt<64, 32>), dim3((cols + 31) / 32, (rows + 63) / 64), dim3(256), 0, 0, rows, cols, in, (size_t)cols, out, (size_t)rows));
    RUN("fwd 32x64", hipLaunchKernelGGL((tt<32, 64>), dim3((cols + 63) / 64, (rows + 31) / 32), dim3(256), 0, 0, rows, cols, in, (size_t)cols, out, (size_t)rows));
    RUN("fwd 64x64", hipLaunchKernelGGL((tt<64, 64>), dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, 0, rows, cols, in, (size_t)cols, out, (size_t)rows));
    RUN("fwd 16x64", hipLaunchKernelGGL((tt<16, 64>), dim3((cols + 63) / 64, (rows + 15) / 16), dim3(256), 0, 0, rows, cols, in, (size_t)cols, out, (size_t)rows));
    RUN("fwd 16x16", hipLaunchKernelGGL((tt<16, 16>), dim3((cols + 15) / 16, (rows + 15) / 16), dim3(256), 0, 0, rows, cols, in, (size_t)cols, out, (size_t)rows));
    RUN("fwd no-LDS by output", hipLaunchKernelGGL(tw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, rows, cols, in, (size_t)cols, out, (size_t)rows));
    // backward: in[cols][rows] -> out[rows][cols]  (rows' = cols, cols' = rows)
    RUN("bwd 32x32 (pm.hip)", hipLaunchKernelGGL(t32, dim3((rows + 31) / 32, (cols + 31) / 32), dim3(256), 0, 0, cols, rows, in, (size_t)rows, out, (size_t)cols));
    RUN("bwd 64x32", hipLaunchKernelGGL((tt<64, 32>), dim3((rows + 31) / 32, (cols + 63) / 64), dim3(256), 0, 0, cols, rows, in, (size_t)rows, out, (size_t)cols));
    RUN("bwd 32x64", hipLaunchKernelGGL((tt<32, 64>), dim3((rows + 63) / 64, (cols + 31) / 32), dim3(256), 0, 0, cols, rows, in, (size_t)rows, out, (size_t)cols));
    RUN("bwd 64x64", hipLaunchKernelGGL((tt<64, 64>), dim3((rows + 63) / 64, (cols + 63) / 64), dim3(256), 0, 0, cols, rows, in, (size_t)rows, out, (size_t)cols));
    RUN("copy (hipMemcpy D2D)", CHK(hipMemcpyAsync(out, in, n * sizeof(double2), hipMemcpyDeviceToDevice, 0)));
    return 0;
}
