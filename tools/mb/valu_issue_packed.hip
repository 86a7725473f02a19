// Micro-benchmark: packed FP32 operations with distinct source registers, 1 / 2 / 7 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define BODY4(T0, T1)                                                                                  \
    asm volatile(T0 : "+v"(p0) : "v"(x), "v"(y)); asm volatile(T1 : "+v"(p1) : "v"(x), "v"(y));      \
    asm volatile(T0 : "+v"(p2) : "v"(x), "v"(y)); asm volatile(T1 : "+v"(p3) : "v"(x), "v"(y));
#define KERNEL(NAME, T0, T1)                                                                           \
    __global__ void __launch_bounds__(256) NAME(float* out, int iters, float xs, float ys)             \
    {                                                                                                  \
        float a = threadIdx.x;                                                                         \
        v2f   p0 = {a, a + 1}, p1 = {a + 2, a + 3}, p2 = {a + 4, a + 5}, p3 = {a + 6, a + 7};          \
        v2f   x = {xs + a, xs}, y = {ys, ys + a};                                                      \
        for (int i = 0; i < iters; ++i) {                                                              \
            _Pragma("unroll") for (int u = 0; u < 16; ++u) { BODY4(T0, T1) }                           \
        }                                                                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y; \
    }
KERNEL(k_pkfma, "v_pk_fma_f32 %0, %0, %1, %2", "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL(k_pkfma_sel, "v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]", "v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]")
KERNEL(k_pkmul, "v_pk_mul_f32 %0, %0, %1", "v_pk_mul_f32 %0, %0, %2")
KERNEL(k_pkadd, "v_pk_add_f32 %0, %0, %1", "v_pk_add_f32 %0, %0, %2")
KERNEL(k_pkmul_sel, "v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,1]", "v_pk_mul_f32 %0, %0, %2 op_sel_hi:[1,0]")
KERNEL(k_pkadd_neg, "v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]", "v_pk_add_f32 %0, %0, %2 neg_lo:[0,1] neg_hi:[0,1]")
typedef void (*kern_t)(float*, int, float, float);
int main()
{
    float* d;
    (void)hipMalloc(&d, 256 * 256 * 64 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    const int iters = 10000;
    struct { const char* name; kern_t k; } ks[] = {{"v_pk_fma_f32", k_pkfma}, {"v_pk_fma_f32 op_sel", k_pkfma_sel}, {"v_pk_mul_f32", k_pkmul},
                                                   {"v_pk_add_f32", k_pkadd}, {"v_pk_mul_f32 op_sel", k_pkmul_sel}, {"v_pk_add_f32 neg", k_pkadd_neg}};
    for (auto& e : ks)
        for (int wps : {1, 2, 7}) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                e.k<<<256 * wps, 256>>>(d, iters, 1.0f, 2.0f);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            printf("%-24s %d waves/SIMD: %7.3f ms  %.2f ns per wave-instruction per SIMD\n", e.name, wps, ms, ms * 1e6 / ((double)iters * 64 * wps));
        }
    return 0;
}
