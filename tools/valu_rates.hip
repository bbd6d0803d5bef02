// valu_rates.hip -- issue cost of the fp64 / integer / LDS instructions the walk kernels are made of, on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/valu_rates.hip ; run on the GPU box.
// Every kernel runs ITERS x 16 independent instructions per lane in 8 waves per SIMD on all CUs; the figure printed is
// SIMD cycles per wave-instruction at the measured time, assuming the nominal 2.4 GHz clock (so ratios are what matters).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

constexpr int ITERS = 4096;
constexpr int NCH = 16;

#define KERNEL(name, decl, body)                                                      \
    __global__ void __launch_bounds__(256) name(double *out, double seed)             \
    {                                                                                 \
        double a[NCH];                                                                \
        int ia[NCH];                                                                  \
        for(int j = 0; j < NCH; j++) {                                                \
            a[j] = seed + threadIdx.x * 1e-3 + j;                                     \
            ia[j] = threadIdx.x + j;                                                  \
        }                                                                             \
        double b = seed * 0.999, c = seed * 1e-3;                                     \
        decl;                                                                         \
        for(int it = 0; it < ITERS; it++) {                                           \
            _Pragma("unroll") for(int j = 0; j < NCH; j++) { body; }                  \
        }                                                                             \
        double s = 0;                                                                 \
        for(int j = 0; j < NCH; j++)                                                  \
            s += a[j] + ia[j];                                                        \
        if(s == 1.2345)                                                               \
            out[0] = s + b + c;                                                       \
    }

KERNEL(k_fma, , asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)))
KERNEL(k_mul, , asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[j]) : "v"(b)))
KERNEL(k_add, , asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[j]) : "v"(c)))
KERNEL(k_min, , asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[j]) : "v"(c)))
KERNEL(k_rsq, , asm volatile("v_rsq_f64 %0, %0" : "+v"(a[j])))
KERNEL(k_rcp, , asm volatile("v_rcp_f64 %0, %0" : "+v"(a[j])))
KERNEL(k_rndne, , asm volatile("v_rndne_f64 %0, %0" : "+v"(a[j])))
KERNEL(k_floor, , asm volatile("v_floor_f64 %0, %0" : "+v"(a[j])))
KERNEL(k_fract, , asm volatile("v_fract_f64 %0, %0" : "+v"(a[j])))
KERNEL(k_cvt_i32_f64, , asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(ia[j]) : "v"(a[j])))
KERNEL(k_cvt_f64_i32, , asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[j]) : "v"(ia[j])))
KERNEL(k_cvt_f32_f64, float f[NCH], asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[j]) : "v"(a[j])); ia[j] = __float_as_int(f[j]))
KERNEL(k_cmp_f64, , asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[j]), "v"(b) : "vcc"))
KERNEL(k_cndmask, , asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia[j]) : "v"(ia[(j + 1) % NCH]) : "vcc"))
KERNEL(k_add_u32, , asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[j]) : "v"(ia[(j + 1) % NCH])))
KERNEL(k_lshl_add_u64, long long la[NCH]; for(int q = 0; q < NCH; q++) la[q] = threadIdx.x + q,
       asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(la[j]) : "v"(la[(j + 1) % NCH])); ia[j] = (int)la[j])
KERNEL(k_fma_f32, float f[NCH]; for(int q = 0; q < NCH; q++) f[q] = threadIdx.x + q,
       asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[j])); ia[j] = __float_as_int(f[j]))
KERNEL(k_rsq_f32, float f[NCH]; for(int q = 0; q < NCH; q++) f[q] = threadIdx.x + q + 1,
       asm volatile("v_rsq_f32 %0, %0" : "+v"(f[j])); ia[j] = __float_as_int(f[j]))
KERNEL(k_bpermute, , asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(ia[j]) : "v"(ia[(j + 1) % NCH] & 252)))
KERNEL(k_dpp_mov, , asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(ia[j]) : "v"(ia[(j + 1) % NCH])))

// round 6: the fp32 instructions a pre-classification of the node tests would be made of (plain and packed)
#define F32DECL float f[NCH]; for(int q = 0; q < NCH; q++) f[q] = threadIdx.x + q + 1; float fb = (float)b, fc = (float)c
#define F32USE ia[j] = __float_as_int(f[j])
KERNEL(k_add_f32, F32DECL, asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[j]) : "v"(fc)); F32USE)
KERNEL(k_mul_f32, F32DECL, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[j]) : "v"(fb)); F32USE)
KERNEL(k_max_f32, F32DECL, asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[j]) : "v"(fc)); F32USE)
KERNEL(k_max3_f32, F32DECL, asm volatile("v_max3_f32 %0, |%0|, |%1|, |%2|" : "+v"(f[j]) : "v"(fb), "v"(fc)); F32USE)
KERNEL(k_cmp_f32_vcc, F32DECL, asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f[j]), "v"(fb) : "vcc"); F32USE)
KERNEL(k_cmp_f32_sgpr, F32DECL; unsigned long long m = 0, asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m) : "v"(f[j]), "v"(fb)); ia[j] += (int)m)
KERNEL(k_cmp_f64_sgpr, unsigned long long m = 0, asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(m) : "v"(a[j]), "v"(b)); ia[j] += (int)m)
KERNEL(k_sqrt_f32, F32DECL, asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[j])); F32USE)
// packed fp32: a[j] (a VGPR pair) carries two floats
KERNEL(k_pk_add_f32, , asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[j]) : "v"(c)))
KERNEL(k_pk_mul_f32, , asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[j]) : "v"(b)))
KERNEL(k_pk_fma_f32, , asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)))
KERNEL(k_pk_add_f32_bc, , asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a[j]) : "v"(c)))
// a fp64 and a fp32 stream side by side: do they share the issue slots?
KERNEL(k_mix_f64_f32, F32DECL, asm volatile("v_add_f64 %0, %0, %2\n v_add_f32 %1, %1, %3" : "+v"(a[j]), "+v"(f[j]) : "v"(c), "v"(fc)); F32USE)
KERNEL(k_mix_f64_pk, double a2[NCH]; for(int q = 0; q < NCH; q++) a2[q] = a[q], asm volatile("v_add_f64 %0, %0, %2\n v_pk_add_f32 %1, %1, %2" : "+v"(a[j]), "+v"(a2[j]) : "v"(c)); ia[j] += (int)a2[j])

KERNEL(k_min_i32, , asm volatile("v_min_i32 %0, %0, %1" : "+v"(ia[j]) : "v"(ia[(j + 1) % NCH])))
KERNEL(k_min3_f32, F32DECL, asm volatile("v_min3_f32 %0, |%0|, |%1|, |%2|" : "+v"(f[j]) : "v"(fb), "v"(fc)); F32USE)
KERNEL(k_min_f32_abs, F32DECL, asm volatile("v_min_f32 %0, |%0|, |%1|" : "+v"(f[j]) : "v"(fb)); F32USE)
KERNEL(k_and_or_b32, , asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(ia[j]) : "v"(ia[(j + 1) % NCH]), "v"(ia[(j + 2) % NCH])))
KERNEL(k_cmp_i32_sgpr, unsigned long long m = 0, asm volatile("v_cmp_lt_i32 %0, %1, %2" : "=s"(m) : "v"(ia[j]), "v"(ia[(j + 1) % NCH])); ia[j] += (int)m)
KERNEL(k_cmp_f32_vcc_mov, F32DECL; unsigned long long m = 0, asm volatile("v_cmp_lt_f32 vcc, %1, %2\n s_mov_b64 %0, vcc" : "=s"(m) : "v"(f[j]), "v"(fb) : "vcc"); ia[j] += (int)m)
KERNEL(k_cmp_f32_lit, F32DECL; unsigned long long m = 0, asm volatile("v_cmp_lt_f32 %0, 0, %1" : "=s"(m) : "v"(f[j])); ia[j] += (int)m)
KERNEL(k_cmp_class_f32, F32DECL; unsigned long long m = 0, asm volatile("v_cmp_class_f32 %0, %1, %2" : "=s"(m) : "v"(f[j]), "v"(ia[j])); ia[j] += (int)m)
KERNEL(k_mbcnt, , asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0\n v_mbcnt_hi_u32_b32 %0, %1, %0" : "+v"(ia[j]) : "s"(j * 77)))
KERNEL(k_sub_f32_x6_cmp, F32DECL; unsigned long long m = 0, asm volatile("v_sub_f32 %1, %1, %2\n v_sub_f32 %1, %1, %2\n v_fma_f32 %1, %1, %2, %2\n v_cmp_lt_f32 %0, %1, %2" : "=s"(m), "+v"(f[j]) : "v"(fb)); ia[j] += (int)m)

KERNEL(k_cmp_f32_sgpr4, F32DECL; unsigned long long m[4]; m[0] = m[1] = m[2] = m[3] = 0, asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m[j & 3]) : "v"(f[j]), "v"(fb)); ia[j] += (int)m[j & 3])
KERNEL(k_cmp_f64_sgpr4, unsigned long long m[4]; m[0] = m[1] = m[2] = m[3] = 0, asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(m[j & 3]) : "v"(a[j]), "v"(b)); ia[j] += (int)m[j & 3])

// round 6, last experiment: the five comparisons of a list-kernel target pass as v_cmpx (exec narrowing: 2 for "discard", 3 negated for
// "not open") in one asm block, against five v_cmp into scalar pairs + scalar and / or
KERNEL(k_cmpx5_block, unsigned long long md = 0; unsigned long long mno = 0; unsigned long long sv = 0; unsigned long long act = 0xffffffffffff0fffull,
       asm volatile("s_mov_b64 %2, exec\n s_mov_b64 exec, %3\n v_cmpx_lt_f64 %5, %4\n v_cmpx_gt_f64 %4, %6\n s_mov_b64 %0, exec\n s_mov_b64 exec, %3\n"
                    "v_cmpx_ngt_f64 %4, %5\n v_cmpx_ngt_f64 %6, %4\n v_cmpx_nlt_f64 %4, %6\n s_mov_b64 %1, exec\n s_mov_b64 exec, %2"
                    : "=&s"(md), "=&s"(mno), "=&s"(sv) : "s"(act), "v"(a[j]), "v"(b), "v"(c) : "vcc");
       ia[j] += (int)(md ^ mno))
KERNEL(k_cmp5_sgpr, unsigned long long m0 = 0; unsigned long long m1 = 0; unsigned long long m2 = 0; unsigned long long m3 = 0; unsigned long long m4 = 0,
       asm volatile("v_cmp_lt_f64 %0, %6, %5\n v_cmp_gt_f64 %1, %5, %7\n v_cmp_gt_f64 %2, %5, %6\n v_cmp_gt_f64 %3, %7, %5\n v_cmp_lt_f64 %4, %5, %7\n"
                    "s_and_b64 %0, %0, %1\n s_or_b64 %2, %2, %3\n s_or_b64 %2, %2, %4"
                    : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(m4) : "v"(a[j]), "v"(b), "v"(c));
       ia[j] += (int)(m0 ^ m2))

__global__ void __launch_bounds__(256) k_lds128(double *out, double seed)
{
    __shared__ double tab[2048];
    for(int i = threadIdx.x; i < 2048; i += 256)
        tab[i] = seed + i;
    __syncthreads();
    double acc = 0;
    int idx = (threadIdx.x * 37) & 1023;
    for(int it = 0; it < ITERS; it++) {
#pragma unroll
        for(int j = 0; j < NCH; j++) {
            double2 v = *(const double2 *)&tab[(idx & 1023) * 2];
            acc += v.x;
            idx += (int)v.y & 3;
        }
    }
    if(acc == 1.2345)
        out[0] = acc;
}

typedef void (*kern_t)(double *, double);

static void run(const char *name, kern_t k, double *d_out, int ncu)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int blocks = ncu * 8; // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, 1.5);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, 1.5);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double winstr_per_simd = 8.0 * ITERS * NCH;
    printf("%-16s %8.3f ms  %6.2f cycles / wave-instruction (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / winstr_per_simd);
}

int main()
{
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs, clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    double *d_out;
    CHK(hipMalloc(&d_out, 64));
#define R(k) run(#k, k, d_out, p.multiProcessorCount)
    R(k_fma);
    R(k_mul);
    R(k_add);
    R(k_min);
    R(k_rsq);
    R(k_rcp);
    R(k_rndne);
    R(k_floor);
    R(k_fract);
    R(k_cvt_i32_f64);
    R(k_cvt_f64_i32);
    R(k_cvt_f32_f64);
    R(k_cmp_f64);
    R(k_cndmask);
    R(k_add_u32);
    R(k_lshl_add_u64);
    R(k_fma_f32);
    R(k_rsq_f32);
    R(k_bpermute);
    R(k_dpp_mov);
    R(k_lds128);
    R(k_add_f32);
    R(k_mul_f32);
    R(k_max_f32);
    R(k_max3_f32);
    R(k_cmp_f32_vcc);
    R(k_cmp_f32_sgpr);
    R(k_cmp_f64_sgpr);
    R(k_sqrt_f32);
    R(k_pk_add_f32);
    R(k_pk_mul_f32);
    R(k_pk_fma_f32);
    R(k_pk_add_f32_bc);
    R(k_mix_f64_f32);
    R(k_mix_f64_pk);
    R(k_min_i32);
    R(k_min3_f32);
    R(k_min_f32_abs);
    R(k_and_or_b32);
    R(k_cmp_i32_sgpr);
    R(k_cmp_f32_vcc_mov);
    R(k_cmp_f32_lit);
    R(k_cmp_class_f32);
    R(k_mbcnt);
    R(k_sub_f32_x6_cmp);
    R(k_cmp_f32_sgpr4);
    R(k_cmp_f64_sgpr4);
    R(k_cmpx5_block);
    R(k_cmp5_sgpr);
    return 0;
}
