// pmc_calib.hip -- what do FETCH_SIZE / WRITE_SIZE report on gfx950 for the access patterns of the walk kernels?
// MI355X_MICROARCH.md (HBM): FETCH_SIZE is half of the bytes of a 16 B/lane streaming read on this rocprofv3; "other access widths and
// WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern".  Every kernel here moves a KNOWN number of
// bytes (printed), far more than the 256 MiB Infinity Cache; tools/pmc_calib.sh runs the binary under rocprofv3 --pmc FETCH_SIZE and
// --pmc WRITE_SIZE (separate passes) and prints counter x 1024 / bytes per kernel.
//   k_read16   16 B per lane, consecutive (the guide's calibrated case: expect 0.5)
//   k_read4     4 B per lane, consecutive (256 B per wave instruction)
//   k_read4g    4 B per lane, 8 groups of 8 lanes on 8 different 2-KiB regions (k_walk_eval's list reads: one batch of 8 entries per group)
//   k_gather32  32 B per lane at random 32-byte records of a 1 GiB array (k_walk_eval's source gathers, two dwordx4 per lane)
//   k_write4    4 B per lane, consecutive
//   k_write4run runs of 10 consecutive 4-byte stores per wave instruction to 8 regions in turn (k_walk_lists8's appends)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

__global__ void __launch_bounds__(256) k_read16(const uint4 *__restrict__ p, size_t n, unsigned *sink)
{
    unsigned acc = 0;
    for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if(acc == 0x12345678u)
        *sink = acc;
}
__global__ void __launch_bounds__(256) k_read4(const unsigned *__restrict__ p, size_t n, unsigned *sink)
{
    unsigned acc = 0;
    for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        acc ^= p[i];
    if(acc == 0x12345678u)
        *sink = acc;
}
// wave w owns 8 consecutive regions of `cap` words; step e0: group g reads words [e0, e0 + 8) of region g
__global__ void __launch_bounds__(256) k_read4g(const unsigned *__restrict__ p, size_t nwaves, int cap, unsigned *sink)
{
    const int lane = threadIdx.x & 63, g = lane >> 3, s = lane & 7;
    unsigned acc = 0;
    for(size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < nwaves; w += (size_t)gridDim.x * 4) {
        const unsigned *r = p + (w * 8 + g) * (size_t)cap + s;
        for(int e0 = 0; e0 < cap; e0 += 8)
            acc ^= r[e0];
    }
    if(acc == 0x12345678u)
        *sink = acc;
}
__global__ void __launch_bounds__(256) k_gather32(const uint4 *__restrict__ p, size_t nrec, size_t nloads, unsigned *sink)
{
    unsigned acc = 0;
    for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nloads; i += (size_t)gridDim.x * 256) {
        // 8 lanes read 8 consecutive records (a leaf), the groups land on unrelated leaves
        const size_t leaf = ((i >> 3) * 0x9E3779B97F4A7C15ull) % (nrec / 8);
        const uint4 *r = p + 2 * (leaf * 8 + (i & 7));
        const uint4 a = r[0], b = r[1];
        acc ^= a.x ^ b.w;
    }
    if(acc == 0x12345678u)
        *sink = acc;
}
__global__ void __launch_bounds__(256) k_write4(unsigned *__restrict__ p, size_t n)
{
    for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        p[i] = (unsigned)i;
}
// wave w owns 8 regions of `cap` words; it appends runs of `run` words to region 0, 1, .. 7, 0, .. until they are full
__global__ void __launch_bounds__(256) k_write4run(unsigned *__restrict__ p, size_t nwaves, int cap, int run)
{
    const int lane = threadIdx.x & 63;
    for(size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < nwaves; w += (size_t)gridDim.x * 4)
        for(int at = 0; at < cap; at += run)
            for(int t = 0; t < 8; t++)
                if(lane < run && at + lane < cap)
                    p[(w * 8 + t) * (size_t)cap + at + lane] = (unsigned)(at + lane);
}

int main(int argc, char **argv)
{
    const size_t bytes = (size_t)(argc > 1 ? atoll(argv[1]) : 4) << 30; // GiB moved per kernel
    unsigned *buf, *sink;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, bytes));
    const int cap = 512;
    const size_t nwaves = bytes / 4 / cap / 8;
    const dim3 grid(256 * 8), block(256);
    hipLaunchKernelGGL(k_read16, grid, block, 0, 0, (const uint4 *)buf, bytes / 16, sink);
    hipLaunchKernelGGL(k_read4, grid, block, 0, 0, buf, bytes / 4, sink);
    hipLaunchKernelGGL(k_read4g, grid, block, 0, 0, buf, nwaves, cap, sink);
    const size_t nrec = ((size_t)1 << 30) / 32, nloads = bytes / 32;
    hipLaunchKernelGGL(k_gather32, grid, block, 0, 0, (const uint4 *)buf, nrec, nloads, sink);
    hipLaunchKernelGGL(k_write4, grid, block, 0, 0, buf, bytes / 4);
    hipLaunchKernelGGL(k_write4run, grid, block, 0, 0, buf, nwaves, cap, 10);
    CK(hipDeviceSynchronize());
    printf("bytes per kernel: k_read16 %zu k_read4 %zu k_read4g %zu k_gather32 %zu (requested; 1 GiB array: re-reads hit L2 / L3) k_write4 %zu k_write4run %zu\n", bytes, bytes,
           nwaves * 8 * (size_t)cap * 4, nloads * 32, bytes, nwaves * 8 * (size_t)cap * 4);
    return 0;
}
