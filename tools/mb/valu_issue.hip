// Micro-benchmark: issue cost per wave-instruction on a SIMD for a catalogue of vector instructions, 1 and 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define BODY8(T0, T1)                                                                                                  \
    asm volatile(T0 : "+v"(a0) : "v"(x), "v"(y)); asm volatile(T1 : "+v"(a1) : "v"(x), "v"(y));                      \
    asm volatile(T0 : "+v"(a2) : "v"(x), "v"(y)); asm volatile(T1 : "+v"(a3) : "v"(x), "v"(y));                      \
    asm volatile(T0 : "+v"(a4) : "v"(x), "v"(y)); asm volatile(T1 : "+v"(a5) : "v"(x), "v"(y));                      \
    asm volatile(T0 : "+v"(a6) : "v"(x), "v"(y)); asm volatile(T1 : "+v"(a7) : "v"(x), "v"(y));
#define KERNEL(NAME, T0, T1)                                                                                           \
    __global__ void __launch_bounds__(256) NAME(float* out, int iters, float x, float y)                               \
    {                                                                                                                  \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        for (int i = 0; i < iters; ++i) {                                                                              \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) { BODY8(T0, T1) }                                            \
        }                                                                                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                            \
    }
KERNEL(k_add, "v_add_f32_e32 %0, %0, %1", "v_add_f32_e32 %0, %0, %1")
KERNEL(k_add64, "v_add_f32_e64 %0, %0, %1", "v_add_f32_e64 %0, %0, %1")
KERNEL(k_addabs, "v_add_f32_e64 %0, |%0|, %1", "v_add_f32_e64 %0, |%0|, %1")
KERNEL(k_mul, "v_mul_f32_e32 %0, %0, %1", "v_mul_f32_e32 %0, %0, %1")
KERNEL(k_fmac, "v_fmac_f32_e32 %0, %1, %2", "v_fmac_f32_e32 %0, %1, %2")
KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2", "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_fma2, "v_fma_f32 %0, %0, %1, %1", "v_fma_f32 %0, %0, %1, %1")
KERNEL(k_max, "v_max_f32_e32 %0, %0, %1", "v_max_f32_e32 %0, %0, %1")
KERNEL(k_and, "v_and_b32_e32 %0, %0, %1", "v_xor_b32_e32 %0, %0, %1")
KERNEL(k_addu, "v_add_u32_e32 %0, %0, %1", "v_sub_u32_e32 %0, %0, %1")
KERNEL(k_lshl, "v_lshlrev_b32_e32 %0, 1, %0", "v_lshrrev_b32_e32 %0, 1, %0")
KERNEL(k_mov, "v_mov_b32_e32 %0, %1", "v_mov_b32_e32 %0, %2")
KERNEL(k_cndmask64, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]", "v_cndmask_b32_e64 %0, %0, %2, s[10:11]")
KERNEL(k_cmp, "v_cmp_lt_f32_e32 vcc, %0, %1", "v_cmp_gt_f32_e32 vcc, %0, %2")
KERNEL(k_cmp64, "v_cmp_lt_f32_e64 s[10:11], %0, %1", "v_cmp_gt_f32_e64 s[12:13], %0, %2")
KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 1, %1", "v_lshl_add_u32 %0, %0, 1, %2")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2", "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 1, 30", "v_bfe_u32 %0, %0, 2, 29")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2", "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1", "v_mul_lo_u32 %0, %0, %2")
KERNEL(k_rcp, "v_rcp_f32_e32 %0, %0", "v_rcp_f32_e32 %0, %0")
KERNEL(k_cvt, "v_cvt_f32_u32_e32 %0, %0", "v_cvt_u32_f32_e32 %0, %0")
KERNEL(k_med3, "v_med3_f32 %0, %0, %1, %2", "v_min3_f32 %0, %0, %1, %2")
KERNEL(k_mulk, "v_mul_f32_e32 %0, 0x3f8ccccd, %0", "v_mul_f32_e32 %0, 0x3f8ccccd, %0")
KERNEL(k_addsgpr, "v_add_f32_e32 %0, s20, %0", "v_add_f32_e32 %0, s21, %0")
typedef void (*kern_t)(float*, int, float, float);
int main()
{
    float* d;
    (void)hipMalloc(&d, 256 * 256 * 64 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    const int iters = 10000;
    struct { const char* name; kern_t k; } ks[] = {
        {"v_add_f32_e32", k_add}, {"v_add_f32_e64", k_add64}, {"v_add_f32_e64 |abs|", k_addabs}, {"v_mul_f32_e32", k_mul}, {"v_fmac_f32_e32", k_fmac},
        {"v_fma_f32 (3 regs)", k_fma}, {"v_fma_f32 (2 regs)", k_fma2}, {"v_max_f32_e32", k_max}, {"v_and/v_xor_b32", k_and}, {"v_add/v_sub_u32", k_addu},
        {"v_cmp_f32_e32", k_cmp}, {"v_cmp_f32_e64", k_cmp64}, {"v_lshl_add_u32", k_lshladd}, {"v_add3_u32", k_add3}, {"v_bfe_u32", k_bfe},
        {"v_mad_u32_u24", k_mad24}, {"v_mul_lo_u32", k_mullo}, {"v_rcp_f32", k_rcp}, {"v_cvt_f32_u32/u32_f32", k_cvt}, {"v_med3/v_min3_f32", k_med3},
        {"v_mul_f32 literal", k_mulk}, {"v_add_f32 sgpr", k_addsgpr}};
    for (auto& e : ks)
        for (int wps : {1, 2, 7}) {
            const int blocks = 256 * wps;
            float     ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                e.k<<<blocks, 256>>>(d, iters, 1.0f, 2.0f);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            const double instr_per_simd = (double)iters * 64 * wps;
            printf("%-24s %d waves/SIMD: %7.3f ms  %.2f ns per wave-instruction per SIMD\n", e.name, wps, ms, ms * 1e6 / instr_per_simd);
        }
    return 0;
}
