// Micro-benchmark: issue cost of scalar instructions per SIMD (1, 2, 7 waves resident), alone and beside vector work.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(T) T T T T T T T T
__global__ void __launch_bounds__(256) k_salu(float* out, int iters, float x)
{
    float a = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile(REP8("s_and_b64 s[20:21], s[20:21], exec\n s_or_b64 s[22:23], s[22:23], s[20:21]\n") ::: "s20", "s21", "s22", "s23", "scc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void __launch_bounds__(256) k_mixed(float* out, int iters, float x)
{
    float a0 = threadIdx.x, a1 = a0 + 1;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile(REP8("s_and_b64 s[20:21], s[20:21], exec\n v_add_f32_e32 %0, %0, %2\n s_or_b64 s[22:23], s[22:23], s[20:21]\n v_mul_f32_e32 %1, %1, %2\n")
                         : "+v"(a0), "+v"(a1) : "v"(x) : "s20", "s21", "s22", "s23", "scc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1;
}
__global__ void __launch_bounds__(256) k_mixed_slow(float* out, int iters, float x)
{
    float a0 = threadIdx.x, a1 = a0 + 1;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile(REP8("s_and_b64 s[20:21], s[20:21], exec\n v_max_f32_e32 %0, %0, %2\n s_or_b64 s[22:23], s[22:23], s[20:21]\n v_min_f32_e32 %1, %1, %2\n")
                         : "+v"(a0), "+v"(a1) : "v"(x) : "s20", "s21", "s22", "s23", "scc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1;
}
typedef void (*kern_t)(float*, int, float);
int main()
{
    float* d;
    (void)hipMalloc(&d, 256 * 256 * 64 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    const int iters = 5000;
    struct { const char* name; kern_t k; int per_iter; } ks[] = {{"s_and_b64 / s_or_b64", k_salu, 128},
                                                                 {"1 scalar + 1 v_add/v_mul", k_mixed, 128},
                                                                 {"1 scalar + 1 v_max/v_min", k_mixed_slow, 128}};
    for (auto& e : ks)
        for (int wps : {1, 2, 7}) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                e.k<<<256 * wps, 256>>>(d, iters, 1.0f);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            printf("%-28s %d waves/SIMD: %7.3f ms  %.2f ns per scalar instruction (or scalar + vector pair) per SIMD\n", e.name, wps, ms,
                   ms * 1e6 / ((double)iters * e.per_iter * wps));
        }
    return 0;
}
