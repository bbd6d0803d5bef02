// sort_bits_check.hip -- rocprim::radix_sort_pairs with begin_bit > 0: out-of-order elements and mismatched pairs at n = 4096 and 100 000 (ROCm 7.2.0 / rocprim 4.2), correct at 5 000 000.  Build: hipcc --offload-arch=gfx950 -O2; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstdint>
#define CHK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)
int main(int argc, char **argv)
{
    for(int n : {4096, 100000, 5000000}) for(int bb : {33, 32, 0}) {
        std::vector<uint64_t> k(n); std::vector<uint32_t> v(n);
        uint64_t s = 88172645463325252ull;
        for(int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; k[i] = s >> 1; v[i] = i; }
        uint64_t *ka, *kb; uint32_t *va, *vb;
        CHK(hipMalloc(&ka, n * 8)); CHK(hipMalloc(&kb, n * 8)); CHK(hipMalloc(&va, n * 4)); CHK(hipMalloc(&vb, n * 4));
        CHK(hipMemcpy(ka, k.data(), n * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(va, v.data(), n * 4, hipMemcpyHostToDevice));
        size_t tb = 0; void *tmp = nullptr;
        CHK(rocprim::radix_sort_pairs(nullptr, tb, ka, kb, va, vb, (size_t)n, bb, 64, 0));
        CHK(hipMalloc(&tmp, tb + 16));
        CHK(rocprim::radix_sort_pairs(tmp, tb, ka, kb, va, vb, (size_t)n, bb, 64, 0));
        CHK(hipDeviceSynchronize());
        std::vector<uint64_t> r(n); std::vector<uint32_t> rv(n);
        CHK(hipMemcpy(r.data(), kb, n * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(rv.data(), vb, n * 4, hipMemcpyDeviceToHost));
        long bad = 0, badpair = 0;
        for(int i = 1; i < n; i++) if((r[i] >> bb) < (r[i - 1] >> bb)) bad++;
        for(int i = 0; i < n; i++) if(r[i] != k[rv[i]]) badpair++;
        printf("n %d begin_bit %d: out-of-order %ld, key/value mismatches %ld, temp %zu\n", n, bb, bad, badpair, tb);
        hipFree(ka); hipFree(kb); hipFree(va); hipFree(vb); hipFree(tmp);
    }
    return 0;
}
