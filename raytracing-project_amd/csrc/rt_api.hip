// rt_api.hip - kernels and the extern "C" layer of librt_hip.so (see
// include/rt_hip.h).  gfx950 only.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_device.h"
#include "rt_kdrefine.h"
#include "rt_kdbuild.h"

#pragma clang fp contract(off)

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(RT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

extern "C" const char* rt_last_error(void) { return g_err; }

extern "C" int rt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" void rt_default_params(RtParams* p, int32_t width, int32_t height)
{
    memset(p, 0, sizeof *p);
    p->width = width, p->height = height;
    p->msaa = 1;
    p->backgroundlighting = 0.34f;
    p->diffusePower = 0.5f;
    p->ShadowFactor = 0.5f;
    p->depthPower = 2.5f;
    p->backgroundColor[0] = 0.2f, p->backgroundColor[1] = 0.5f, p->backgroundColor[2] = 1.0f;
    p->focusDepth = 0.0f;
    p->tile_w = p->tile_h = 0;
    p->rank = 0, p->world = 1;
    p->shadowSamples = 1, p->shadowRadius = 0.9f; // settings.h:12-13
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
#define RT_BLOCK 256

__device__ __forceinline__ void flush_counters(DCounters* ctr, unsigned nc, unsigned ns, unsigned nt,
                                               unsigned nn, unsigned ncam, unsigned nsh = 0, unsigned nch = 0)
{
    // one atomic per wave and counter
    for (int off = 32; off > 0; off >>= 1) {
        nc += __shfl_down(nc, off);
        ns += __shfl_down(ns, off);
        nt += __shfl_down(nt, off);
        nn += __shfl_down(nn, off);
        ncam += __shfl_down(ncam, off);
        nsh += __shfl_down(nsh, off);
        nch += __shfl_down(nch, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&ctr->rays_closest, (unsigned long long)nc);
        atomicAdd(&ctr->rays_shadow, (unsigned long long)ns);
        atomicAdd(&ctr->tri_tests, (unsigned long long)nt);
        atomicAdd(&ctr->node_visits, (unsigned long long)nn);
        atomicAdd(&ctr->camera_rays, (unsigned long long)ncam);
        if (nsh) atomicAdd(&ctr->shaded_hits, (unsigned long long)nsh);
        if (nch) atomicAdd(&ctr->child_rays, (unsigned long long)nch);
    }
}

__device__ __forceinline__ float clamp01(float v) // imageWriter.h:26-44
{
    if (v > 1.0f) v = 1.0f;
    if (v < 0.0f) v = 0.0f;
    return v;
}

// checkHit for a batch of rays
__global__ void __launch_bounds__(RT_BLOCK)
k_check_hits(DScene S, long long n, const float* __restrict__ org, const float* __restrict__ dst,
             int* tri, float* a, float* b, float* t, float* hp, DCounters* ctr)
{
    extern __shared__ uint2 s_stack[];
    const long long i = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
    unsigned        nt = 0, nn = 0, nc = 0;
    if (i < n) {
        const f3 o = mk3(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
        const f3 d = mk3(dst[3 * i], dst[3 * i + 1], dst[3 * i + 2]);
        HitRec   h;
        nc = 1;
        const bool got = kd_trace<false>(S, o, d, s_stack + threadIdx.x, RT_BLOCK, h, nt, nn);
        if (tri) tri[i] = got ? __float_as_int(S.tris[RT_SLOT_F4 * h.slot + 2].w) : -1;
        if (a) a[i] = h.a;
        if (b) b[i] = h.b;
        if (t) t[i] = h.t;
        if (hp) {
            const f3 p = got ? o + (d - o) * h.t : mk3(0.0f, 0.0f, 0.0f);
            hp[3 * i] = p.x, hp[3 * i + 1] = p.y, hp[3 * i + 2] = p.z;
        }
    }
    flush_counters(ctr, nc, 0, nt, nn, 0);
}

// performRayTracing for a batch of rays
__global__ void __launch_bounds__(RT_BLOCK)
k_trace_rays(DScene S, DParams P, int k0, float prev_ni, long long n, const float* __restrict__ org,
             const float* __restrict__ dst, float* rgb, float* minT, int* tri, DCounters* ctr)
{
    extern __shared__ uint2 s_stack[];
    const long long i = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
    unsigned        nt = 0, nn = 0, nc = 0, ns = 0, ncam = 0, nsh = 0, nch = 0;
    if (i < n) {
        const f3 o = mk3(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
        const f3 d = mk3(dst[3 * i], dst[3 * i + 1], dst[3 * i + 2]);
        f3       c;
        float    mt;
        int      ft;
        ncam = 1;
        trace_sample(S, P, o, d, k0, prev_ni, s_stack + threadIdx.x, RT_BLOCK, c, mt, ft, nc, ns, nt, nn, nsh, nch,
                     (unsigned long long)i * 32);
        rgb[3 * i] = c.x, rgb[3 * i + 1] = c.y, rgb[3 * i + 2] = c.z;
        if (minT) minT[i] = mt;
        if (tri) tri[i] = ft;
    }
    flush_counters(ctr, nc, ns, nt, nn, ncam, nsh, nch);
}

// Pixel-centre ray of every pixel slot.  MSAA == 1: the whole sample, pixel
// written.  MSAA > 1: visibility only (main.cpp:440-445 needs just hit/miss of
// the centre ray; its colour is discarded at main.cpp:452); missed pixels are
// final, hit pixels are flagged for k_samples.
template <bool FULL>
__global__ void __launch_bounds__(RT_BLOCK)
k_centre(DScene S, DParams P, long long n_slots, float* rgb, float* depth, int* tri,
         unsigned char* hit_flag, DCounters* ctr)
{
    extern __shared__ uint2 s_stack[];
    const long long slot = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
    unsigned        nt = 0, nn = 0, nc = 0, ns = 0, ncam = 0, nsh = 0, nch = 0;
    int             x, y;
    if (slot < n_slots && slot_to_pixel(P, slot, x, y)) {
        f3 o, d;
        pixel_ray(P, (double)x, (double)y, o, d);
        const size_t p = (size_t)y * P.W + x;
        ncam = 1;
        if (FULL) {
            f3    c;
            float mt;
            int   ft;
            trace_sample(S, P, o, d, 0, 1.0f, s_stack + threadIdx.x, RT_BLOCK, c, mt, ft, nc, ns, nt, nn, nsh, nch,
                         (unsigned long long)p * 32);
            rgb[3 * p] = clamp01(c.x), rgb[3 * p + 1] = clamp01(c.y), rgb[3 * p + 2] = clamp01(c.z);
            depth[p] = mt; // FLT_MAX on a miss (main.cpp:442-443)
            if (tri) tri[p] = ft;
        } else {
            HitRec h;
            nc = 1;
            const bool got = kd_trace<false>(S, o, d, s_stack + threadIdx.x, RT_BLOCK, h, nt, nn);
            if (tri) tri[p] = got ? __float_as_int(S.tris[RT_SLOT_F4 * h.slot + 2].w) : -1;
            hit_flag[slot] = got ? 1 : 0;
            if (!got) {
                rgb[3 * p] = clamp01(P.bg[0]), rgb[3 * p + 1] = clamp01(P.bg[1]),
                        rgb[3 * p + 2] = clamp01(P.bg[2]);
                depth[p] = RT_FLT_MAX;
            }
        }
    }
    flush_counters(ctr, nc, ns, nt, nn, ncam, nsh, nch);
}

__constant__ float c_msaa4[4][2] = {{-1.f / 8, -3.f / 8}, {3.f / 8, -1.f / 8}, {-3.f / 8, 1.f / 8},
                                    {1.f / 8, 3.f / 8}}; // main.cpp:452-458
__constant__ float c_msaa16[16][2] = {
    {-9.f / 32, -15.f / 32}, {-1.f / 32, -13.f / 32}, {7.f / 32, -11.f / 32}, {15.f / 32, -9.f / 32},
    {-11.f / 32, -7.f / 32}, {-3.f / 32, -5.f / 32},  {5.f / 32, -3.f / 32},  {13.f / 32, -1.f / 32},
    {-13.f / 32, 1.f / 32},  {-5.f / 32, 3.f / 32},   {3.f / 32, 5.f / 32},   {11.f / 32, 7.f / 32},
    {-15.f / 32, 9.f / 32},  {-7.f / 32, 11.f / 32},  {1.f / 32, 13.f / 32},  {9.f / 32, 15.f / 32}};
// main.cpp:469-500 (all offsets are exact binary fractions)

#include "rt_wavefront.h"

// One thread per (sample, pixel slot) of the pixels whose centre ray hit.
// samples[(slot * msaa + s)] = r g b minT.
__global__ void __launch_bounds__(RT_BLOCK)
k_samples(DScene S, DParams P, long long n_slots, const unsigned char* __restrict__ hit_flag,
          float4* samples, DCounters* ctr)
{
    extern __shared__ uint2 s_stack[];
    const long long idx = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
    const int       s = (int)(idx / n_slots);
    const long long slot = idx - (long long)s * n_slots;
    unsigned        nt = 0, nn = 0, nc = 0, ns = 0, ncam = 0, nsh = 0, nch = 0;
    int             x, y;
    if (s < P.msaa && hit_flag[slot] && slot_to_pixel(P, slot, x, y)) {
        const float ox = P.msaa == 4 ? c_msaa4[s][0] : c_msaa16[s][0];
        const float oy = P.msaa == 4 ? c_msaa4[s][1] : c_msaa16[s][1];
        f3          o, d, c;
        pixel_ray(P, (double)x + (double)ox, (double)y + (double)oy, o, d);
        float mt;
        int   ft;
        ncam = 1;
        trace_sample(S, P, o, d, 0, 1.0f, s_stack + threadIdx.x, RT_BLOCK, c, mt, ft, nc, ns, nt, nn, nsh, nch,
                     ((unsigned long long)y * P.W + x) * 32 + 1 + s);
        samples[slot * P.msaa + s] = make_float4(c.x, c.y, c.z, mt);
    }
    flush_counters(ctr, nc, ns, nt, nn, ncam, nsh, nch);
}

// main.cpp:449-507: ordered sum of the samples, divided by their number;
// depth = mean of minT over the samples that hit.
__global__ void k_resolve(DParams P, long long n_slots, const unsigned char* __restrict__ hit_flag,
                          const float4* __restrict__ samples, float* rgb, float* depth)
{
    const long long slot = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int             x, y;
    if (slot >= n_slots || !hit_flag[slot] || !slot_to_pixel(P, slot, x, y)) return;
    float r = 0, g = 0, b = 0, dsum = 0;
    int   hits = 0;
    for (int s = 0; s < P.msaa; ++s) {
        const float4 v = samples[slot * P.msaa + s];
        if (s == 0) r = v.x, g = v.y, b = v.z;
        else r += v.x, g += v.y, b += v.z;
        if (v.w != RT_FLT_MAX) dsum += v.w, ++hits;
    }
    const float  n = (float)P.msaa;
    const size_t p = (size_t)y * P.W + x;
    rgb[3 * p] = clamp01(__fdiv_rn(r, n));
    rgb[3 * p + 1] = clamp01(__fdiv_rn(g, n));
    rgb[3 * p + 2] = clamp01(__fdiv_rn(b, n));
    depth[p] = hits > 0 ? __fdiv_rn(dsum, (float)hits) : RT_FLT_MAX;
}

// ---- depth range, imageWriter.h:113-122 ----
__global__ void k_depth_range_partial(const float* __restrict__ depth, long long n, float* part_lo,
                                      float* part_hi)
{
    __shared__ float s_lo[256], s_hi[256];
    float            lo = RT_FLT_MAX, hi = 0.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const float v = depth[i];
        lo = fminf(lo, v);
        if (v != RT_FLT_MAX) hi = fmaxf(hi, v);
    }
    s_lo[threadIdx.x] = lo, s_hi[threadIdx.x] = hi;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            s_lo[threadIdx.x] = fminf(s_lo[threadIdx.x], s_lo[threadIdx.x + s]);
            s_hi[threadIdx.x] = fmaxf(s_hi[threadIdx.x], s_hi[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) part_lo[blockIdx.x] = s_lo[0], part_hi[blockIdx.x] = s_hi[0];
}

// imageWriter.h:125-127: the partial extremes -> denom = max(|lowest - f|, |highest - f|)
__global__ void __launch_bounds__(256)
k_depth_denominator(const float* __restrict__ part_lo, const float* __restrict__ part_hi, int n_part,
                    float focus, float* denom)
{
    __shared__ float s_lo[256], s_hi[256];
    float            lo = RT_FLT_MAX, hi = 0.0f;
    for (int i = threadIdx.x; i < n_part; i += 256) lo = fminf(lo, part_lo[i]), hi = fmaxf(hi, part_hi[i]);
    s_lo[threadIdx.x] = lo, s_hi[threadIdx.x] = hi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            s_lo[threadIdx.x] = fminf(s_lo[threadIdx.x], s_lo[threadIdx.x + s]);
            s_hi[threadIdx.x] = fmaxf(s_hi[threadIdx.x], s_hi[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *denom = fmaxf(fabsf(s_lo[0] - focus), fabsf(s_hi[0] - focus));
}

// imageWriter.h:129-134.  With part_lo != nullptr the block also reduces the extremes of what it
// writes (getDepthRange of the normalised plane, for the second normalisation of the 'r' path).
__global__ void __launch_bounds__(256)
k_depth_normalize(float* depth, long long n, const float* __restrict__ denom_p, float focus, float* part_lo,
                  float* part_hi)
{
    __shared__ float s_lo[256], s_hi[256];
    const float      denom = *denom_p;
    float            lo = RT_FLT_MAX, hi = 0.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const float v = depth[i];
        const float w = (v != RT_FLT_MAX) ? __fdiv_rn(fabsf(v - focus), denom) : 0.0f;
        depth[i] = w;
        lo = fminf(lo, w);
        if (w != RT_FLT_MAX) hi = fmaxf(hi, w);
    }
    if (!part_lo) return;
    s_lo[threadIdx.x] = lo, s_hi[threadIdx.x] = hi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            s_lo[threadIdx.x] = fminf(s_lo[threadIdx.x], s_lo[threadIdx.x + s]);
            s_hi[threadIdx.x] = fmaxf(s_hi[threadIdx.x], s_hi[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) part_lo[blockIdx.x] = s_lo[0], part_hi[blockIdx.x] = s_hi[0];
}

// ---- depth-of-field blur, imageWriter.h:137-231 ----
// The 13x13 kernel of a pixel depends on (i-6)^2 + (j-6)^2 only through
// -((..)/2) with integer division, i.e. 37 possible exponents.  Each thread
// evaluates those 37 powers once into registers (the tap loops are fully
// unrolled, so every table index is a compile-time constant), walks the 169
// taps in the reference's order to form the normalisation sum (the float sum
// depends on the order), divides the 37 entries by it (the same operands as
// the 169 divisions of the reference, hence the same weights) and gathers from
// an LDS copy of the block's 44x20 pixel neighbourhood.
// Tried and not kept: the taps as fused multiply-adds and the 37 weights times one reciprocal of the sum
// instead of 37 divisions (0.374 -> 0.340 ms per 4K frame, the error against the reference's blurred frame
// unchanged at 3.6e-7 - a tenth of the kernel for arithmetic that is no longer the reference's); two
// horizontally adjacent pixels per thread sharing their 14 reads per row (74 weights live at once: the
// compiler takes 255 registers, spills, and the kernel is five times slower); red and green of a tap in the two
// halves of packed multiplications and additions (the same separately rounded products and sums, four
// instructions per tap instead of six: 0.46 against 0.40 ms for range, normalisation and blur of a 4K frame -
// the extra moves into and out of the register pairs cost more than the two instructions saved; the same with
// the weights kept in register pairs and v_pk_mul_f32 naming the half it wants, no move left, 1 830 -> 1 500
// instructions per pixel: 0.399 against 0.400 ms - a packed FP32 multiplication or addition issues no faster
// here than the two plain ones it stands for); the LDS neighbourhood as 12-byte pixels read with ds_read2_b32 +
// ds_read_b32 instead of one read of the 16-byte pixel's first 12 bytes (0.402 against 0.398 ms: a quarter
// fewer LDS bytes change nothing; both together: 0.406; three colour planes read a dword at a time with
// immediate offsets, a quarter fewer LDS cycles and no address arithmetic: 0.392 against 0.399).  By the issue
// costs measured on this GPU (plain additions and multiplications 1 ns per wave and SIMD with several waves
// resident, compares, selects, packed and transcendental operations 1.75 ns and more) the vector instructions
// of a 4K frame's 504 waves per CU take 0.27 ms and their 170 LDS reads 0.30 ms of the CU's LDS pipe; the kernel
// takes 0.335 ms and follows neither figure down by more than 2 %.  Of its 1 830 vector instructions the 169
// taps' separately rounded products and sums are 1 014; only 21 of the 37 powers and divisions are ever computed
// (the others occur in no 13 x 13 kernel, the compiler drops them).
#define DOF_TW 32
#define DOF_TH 8
#define DOF_BLOCK (DOF_TW * DOF_TH)
#define DOF_LW (DOF_TW + 12)
#define DOF_LH (DOF_TH + 12)

// powf(e, x) for the float e = 2.7182817459106445 and x <= 0 (imageWriter.h:165-168):
// 2^(x * log2 e) with the product carried in two floats and the hardware exp2 (about one
// ulp); results below the normal range go through a 2^64 rescale so they come out as the
// denormals (or the zero) powf returns there.  Within 2 ulp of the correctly rounded value.
__device__ __forceinline__ float dof_pow_e(float x)
{
    const float Lh = 1.4426950216293335f, Ll = -2.4551930266625277e-08f; // log2(e), hi + lo
    const float th = x * Lh;
    const float tl = __builtin_fmaf(x, Lh, -th) + x * Ll;
    const bool  small = th < -120.0f;
    float       r = __builtin_amdgcn_exp2f(small ? th + 64.0f : th);
    r = __builtin_fmaf(r, tl * 0.6931471805599453f, r);
    return small ? r * 5.421010862427522e-20f : r; // 2^-64
}

__global__ void __launch_bounds__(DOF_BLOCK)
k_dof_blur(const float* __restrict__ rgb, const float* __restrict__ depthN, int W, int H,
           float depthPower, float* out, const float* __restrict__ denom2, float focus)
{
    __shared__ float4 s_px[DOF_LH * DOF_LW];
    const int bw = W - 12, bh = H - 12;
    const int tiles_x = (bw + DOF_TW - 1) / DOF_TW;
    const int x0 = (blockIdx.x % tiles_x) * DOF_TW, y0 = (blockIdx.x / tiles_x) * DOF_TH;
    for (int i = threadIdx.x; i < DOF_LH * DOF_LW; i += DOF_BLOCK) {
        const int ly = i / DOF_LW, lx = i - ly * DOF_LW;
        const int gx = min(x0 + lx, W - 1), gy = min(y0 + ly, H - 1); // clamped reads feed no written pixel
        const float* p = rgb + 3 * ((size_t)gy * W + gx);
        s_px[i] = make_float4(p[0], p[1], p[2], 0.0f);
    }
    __syncthreads();
    const int tx = threadIdx.x % DOF_TW, ty = threadIdx.x / DOF_TW;
    const int ox = x0 + tx, oy = y0 + ty;
    if (ox >= bw || oy >= bh) return;
    float dn = depthN[(size_t)(oy + 6) * W + ox + 6];
    if (denom2) // the plane is normalised once; imageWriter.h:416 normalises it again before the blur
        dn = (dn != RT_FLT_MAX) ? __fdiv_rn(fabsf(dn - focus), *denom2) : 0.0f;
    float I = 1.0f - dn;
    I *= I;
    float* o = out + 3 * ((size_t)oy * bw + ox);
    if (I >= 1.0f) { // identity kernel
        const float4 c = s_px[(ty + 6) * DOF_LW + tx + 6];
        o[0] = c.x, o[1] = c.y, o[2] = c.z;
        return;
    }
    const float  idp = I * depthPower;
    const float  p2 = idp * idp; // powf(x, 2) is the correctly rounded square
    const double c1 = 1.0 / ((double)(idp * 2.0f) * 3.14159265358979323846);
    float        tab[37];
#pragma unroll
    for (int k = 0; k < 37; ++k) tab[k] = (float)(c1 * (double)dof_pow_e((float)(-k) * p2));
    float total = 0.0f;
#pragma unroll
    for (int i = 0; i < 13; ++i)
#pragma unroll
        for (int j = 0; j < 13; ++j) total += tab[((i - 6) * (i - 6) + (j - 6) * (j - 6)) >> 1];
#pragma unroll
    for (int k = 0; k < 37; ++k) tab[k] = __fdiv_rn(tab[k], total);
    float r = 0.0f, g = 0.0f, b = 0.0f;
#pragma unroll
    for (int j = 0; j < 13; ++j) {
#pragma unroll
        for (int k = 0; k < 13; ++k) {
            const float  w = tab[((j - 6) * (j - 6) + (k - 6) * (k - 6)) >> 1];
            const float4 c = s_px[(ty + j) * DOF_LW + tx + k];
            r += c.x * w;
            g += c.y * w;
            b += c.z * w;
        }
    }
    o[0] = r, o[1] = g, o[2] = b;
}

// (unsigned char)(v * 255.0f) with NaN and out-of-int-range -> 0 (what x86
// cvttss2si + truncation to a byte gives)
__global__ void k_quantize(const float* __restrict__ v, long long n, unsigned char* out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = v[i] * 255.0f;
    int         q = 0;
    if (x == x && x < 2147483648.0f && x >= -2147483648.0f) q = (int)x;
    out[i] = (unsigned char)q;
}

// depth.bmp's grey value, imageWriter.h:387: (unsigned char)((1 - depth) * 255.0f)
__global__ void k_depth_grey(const float* __restrict__ depthN, long long n, unsigned char* out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = (1 - depthN[i]) * 255.0f;
    int         q = 0;
    if (x == x && x < 2147483648.0f && x >= -2147483648.0f) q = (int)x;
    out[i] = (unsigned char)q;
}

__global__ void k_pack_tiles(DParams P, long long n_slots, const float* __restrict__ rgb,
                             const float* __restrict__ depth, float4* payload)
{
    const long long slot = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_slots) return;
    int    x, y;
    float4 v = make_float4(0, 0, 0, 0);
    if (slot_to_pixel(P, slot, x, y)) {
        const size_t p = (size_t)y * P.W + x;
        v = make_float4(rgb[3 * p], rgb[3 * p + 1], rgb[3 * p + 2], depth[p]);
    }
    payload[slot] = v;
}

__global__ void k_unpack_tiles(DParams P, long long n_slots, const float4* __restrict__ payload,
                               float* rgb, float* depth)
{
    const long long slot = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_slots) return;
    int x, y;
    if (!slot_to_pixel(P, slot, x, y)) return;
    const float4 v = payload[slot];
    const size_t p = (size_t)y * P.W + x;
    rgb[3 * p] = v.x, rgb[3 * p + 1] = v.y, rgb[3 * p + 2] = v.z;
    depth[p] = v.w;
}

// ---------------------------------------------------------------------------
// scene
// ---------------------------------------------------------------------------
struct RtScene {
    int     device = 0;
    DScene  d{};
    void*   d_nodes = nullptr;
    void*   d_tris = nullptr;
    void*   d_edges = nullptr;
    void*   d_heads = nullptr;
    void*   d_shade = nullptr;
    void*   d_info = nullptr;
    void*   d_texcoords = nullptr;
    void*   d_mats = nullptr;
    void*   d_texinfo = nullptr;
    void*   d_texels = nullptr;
    void*   d_rank = nullptr;
    void*   d_knodes = nullptr;
    int     tree_depth = 0;
    size_t  bytes = 0;
    uint64_t tree_hash = 0;            // FNV-1a over the compact nodes and the triangle of every slot
    float    create_ms = 0, tree_ms = 0; // rt_scene_create in all / its two tree transformations
    int      tree_on_gpu = 0;
    // per-call workspace, grown on demand
    DCounters*     d_ctr = nullptr;
    unsigned char* d_flags = nullptr;
    size_t         flags_cap = 0;
    float4*        d_samples = nullptr;
    size_t         samples_cap = 0;
    hipEvent_t     ev0 = nullptr, ev1 = nullptr;
    // wavefront renderer
    bool       pushdown = false;   // inner nodes carry no triangles
    bool       refractive = false; // some material can spawn two child rays
    int        wf_grid_trace = 0, wf_grid_any = 0, wf_grid_pipe = 0;
    size_t     wf_lds = 0;
    size_t     n_knodes = 0;
    int        wf_cap_lo = 0;       // stack entries in LDS when one gives way to the copy of the tree's top
    size_t     wf_lds_per_cu = 0;
    size_t     wf_single_sums = (size_t)-1, wf_single_top = 0;
    int        wf_single_cap = 0;
    int        wf_cus = 0;                              // compute units of the device
    // the hand-out order of the single launch's regions (k_wf_order_*): the durations the last launch left, the
    // order made from them, the shape of the launch they belong to (0: none yet), and how many regions fit
    unsigned*  wf_cost = nullptr;
    int*       wf_order = nullptr;
    int*       wf_order_at = nullptr;                  // [WF_ORDER_RUNS][WF_ORDER_BINS]: scratch of the sort
    uint64_t   wf_order_key = 0;
    size_t     wf_order_cap = 0;
    bool       wf_feedback = true;                      // RT_WF_FEEDBACK=0: regions by index, always
    // a region that takes this share of what is left of the launch when its turn comes, or more, goes first (RT_WF_HOT)
    float      wf_hot = 0.3f;
    size_t     wf_single_lds = 0;                       // last LDS size the single launch was sized for, and
    int        wf_single_grid = 0;                      // the blocks that stay resident with it
    void*      d_stk_spill = nullptr; // deep traversal stack entries of the wavefront kernels
    void*      wf_mem = nullptr;   // one allocation carved into the queues
    void*      d_params = nullptr; // device copy of the frame's DParams (light positions)
    hipStream_t wf_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t  wf_join[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t  wf_fork = nullptr;
    size_t     wf_mem_bytes = 0;
    // debug / ablation switches, read from the environment once, when the scene is created
    struct {
        bool megakernel = false, multilaunch = false, kd_stats = false;
        int  pipes = 1, region = 0, stack_lds = 0, budget_mb = 0;
    } cfg;
    // frame buffers of rt_render_result_planes: device planes + pinned host bytes, grown on demand
    void*       d_frame = nullptr;
    void*       h_frame = nullptr;
    size_t      frame_dev_bytes = 0, frame_host_bytes = 0; // what d_frame / h_frame hold
    hipStream_t frame_stream = nullptr;
    void*       d_share = nullptr; // tile payload of this rank (+ the gathered payloads on rank 0)
    size_t      share_bytes = 0;
    int         first_rank_shares = RT_SHARES_PER_RANK; // the deal of shared frames (rt_scene_set_deal / rt_calibrate_deal)
    float       calib_ms[3] = {0, 0, 0};                // rank 0's last calibration: render even, render halved, post-process
    // optional per-kernel timing (rt_scene_set_timing)
    bool                                 timing = false;
    std::vector<hipEvent_t>              ev_pool; // pairs
    std::vector<int>                     ev_kind; // kernel class per pair
    size_t                               ev_used = 0;
};

enum { K_TRACE_CLOSEST = 0, K_TRACE_ANY = 1, K_FRAME = 2 };
#define RT_MAX_TIMED 4096

// Event pair around one launch when timing is on; returns the pair index or -1.
static int timing_begin(RtScene* s, int kind, hipStream_t st)
{
    if (!s->timing || s->ev_used >= RT_MAX_TIMED) return -1;
    if (s->ev_used * 2 + 1 >= s->ev_pool.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess) return -1;
        if (hipEventCreate(&b) != hipSuccess) {
            (void)hipEventDestroy(a);
            return -1;
        }
        s->ev_pool.push_back(a);
        s->ev_pool.push_back(b);
        s->ev_kind.push_back(kind);
    }
    const int i = (int)s->ev_used++;
    s->ev_kind[i] = kind;
    (void)hipEventRecord(s->ev_pool[2 * i], st);
    return i;
}
static void timing_end(RtScene* s, int i, hipStream_t st)
{
    if (i >= 0) (void)hipEventRecord(s->ev_pool[2 * i + 1], st);
}

static int scratch_for_device(int device, float** out); // the no-allocation entry point's array, below

template <class T>
static int upload(void** dptr, const std::vector<T>& v, size_t* total)
{
    const size_t bytes = std::max<size_t>(v.size() * sizeof(T), 64);
    HIP_TRY(hipMalloc(dptr, bytes));
    HIP_TRY(hipMemset(*dptr, 0, bytes));
    if (!v.empty()) HIP_TRY(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *total += bytes;
    return RT_OK;
}

extern "C" void rt_scene_destroy(RtScene* s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    void* ptrs[] = {s->d_nodes, s->d_tris,    s->d_shade, s->d_info,  s->d_texcoords, s->d_mats,
                    s->d_texinfo, s->d_texels, s->d_ctr,  s->d_flags, s->d_samples, s->d_rank, s->wf_mem, s->d_params, s->d_knodes, s->d_stk_spill, s->d_edges, s->d_heads, s->wf_cost, s->wf_order, s->wf_order_at};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (s->d_frame) (void)hipFree(s->d_frame);
    if (s->h_frame) (void)hipHostFree(s->h_frame);
    if (s->d_share) (void)hipFree(s->d_share);
    if (s->frame_stream) (void)hipStreamDestroy(s->frame_stream);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    for (hipEvent_t e : s->ev_pool) (void)hipEventDestroy(e);
    for (int k = 0; k < 4; ++k) {
        if (s->wf_stream[k]) (void)hipStreamDestroy(s->wf_stream[k]);
        if (s->wf_join[k]) (void)hipEventDestroy(s->wf_join[k]);
    }
    if (s->wf_fork) (void)hipEventDestroy(s->wf_fork);
    delete s;
}

// device memory that lives for one call
struct DevBuf {
    void* p = nullptr;
    ~DevBuf()
    {
        if (p) (void)hipFree(p);
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    int alloc(size_t bytes)
    {
        release();
        HIP_TRY(hipMalloc(&p, std::max<size_t>(bytes, 16)));
        return RT_OK;
    }
};

// ---------------------------------------------------------------------------
// the tree transformations of rt_scene_create on the device (rt_kdrefine.h)
// ---------------------------------------------------------------------------
// Level by level: the device filters and splits the triangle lists (one block per node of the
// level), the host keeps the node arrays and numbers new nodes in the order the host loops would.
// Returns 0 when the arrays now hold the transformed tree, 1 when the triangle pool overflowed
// (nothing changed, the caller runs the host loops), < 0 on a HIP error.
static double g_kd_gpu_ms = 0.0; // last rt_scene_create: time spent in here (RT_KD_STATS prints it)

static int kd_transform_gpu(const RtSceneDesc* desc, int device, std::vector<std::vector<int>>& lists,
                            std::vector<int>& new_left, std::vector<int>& axis_of, std::vector<float>& split_of,
                            std::vector<int>& depth_of, std::vector<KdBox>& cell, int& max_depth,
                            const KdSplitParams& prm, bool refine, size_t max_nodes)
{
    HIP_TRY(hipSetDevice(device));
    const int nv = desc->n_vertices, nt = desc->n_triangles;
    std::vector<int> loff(lists.size()), lcnt(lists.size());
    std::vector<int> flat;
    for (size_t k = 0; k < lists.size(); ++k) {
        loff[k] = (int)flat.size(), lcnt[k] = (int)lists[k].size();
        flat.insert(flat.end(), lists[k].begin(), lists[k].end());
    }
    const int pool_cap = (int)std::min<size_t>((size_t)1 << 28, std::max<size_t>((size_t)1 << 22, flat.size() * 256));
    DevBuf    d_vpos, d_tv, d_pool, d_ctr, d_items, d_results;
    int       rc;
    if ((rc = d_vpos.alloc(12 * (size_t)nv)) || (rc = d_tv.alloc(12 * (size_t)nt)) ||
        (rc = d_pool.alloc(4 * (size_t)pool_cap)) || (rc = d_ctr.alloc(8)))
        return rc;
    HIP_TRY(hipMemcpy(d_vpos.p, desc->vertex_pos, 12 * (size_t)nv, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_tv.p, desc->tri_vertex, 12 * (size_t)nt, hipMemcpyHostToDevice));
    if (!flat.empty()) HIP_TRY(hipMemcpy(d_pool.p, flat.data(), 4 * flat.size(), hipMemcpyHostToDevice));
    int ctr[2] = {(int)flat.size(), 0}; // used, overflow
    HIP_TRY(hipMemcpy(d_ctr.p, ctr, 8, hipMemcpyHostToDevice));
    int*   d_used = (int*)d_ctr.p, *d_over = d_used + 1;
    size_t items_cap = 0;
    auto   room = [&](size_t n_items, size_t item_bytes, size_t result_bytes) -> int {
        if (n_items <= items_cap) return RT_OK;
        d_items.release(), d_results.release();
        items_cap = n_items + n_items / 2 + 64;
        int r2;
        if ((r2 = d_items.alloc(items_cap * std::max(sizeof(KdHandItem), sizeof(KdSplitItem))))) return r2;
        (void)item_bytes, (void)result_bytes;
        return d_results.alloc(items_cap * std::max(sizeof(KdHandResult), sizeof(KdSplitResult)));
    };
    // ---- hand the inner nodes' triangles down, one level of the input tree after the other ----
    int in_depth = 0;
    for (size_t k = 0; k < lists.size(); ++k) in_depth = std::max(in_depth, depth_of[k]);
    std::vector<std::vector<int>> by_depth((size_t)in_depth + 1);
    for (size_t k = 0; k < lists.size(); ++k)
        if (new_left[k] >= 0) by_depth[depth_of[k]].push_back((int)k);
    std::vector<KdHandItem>   h_items;
    std::vector<KdHandResult> h_res;
    for (int d = 0; d <= in_depth; ++d) {
        h_items.clear();
        std::vector<int> who;
        for (int k : by_depth[d]) {
            if (lcnt[k] == 0) continue; // nothing to hand down: the children keep their own lists
            KdHandItem it;
            const int  l = new_left[k], r = l + 1;
            it.left = cell[l], it.right = cell[r];
            it.off = loff[k], it.cnt = lcnt[k];
            it.own_off[0] = loff[l], it.own_cnt[0] = lcnt[l], it.own_off[1] = loff[r], it.own_cnt[1] = lcnt[r];
            h_items.push_back(it);
            who.push_back(k);
        }
        if (h_items.empty()) continue;
        if ((rc = room(h_items.size(), sizeof(KdHandItem), sizeof(KdHandResult)))) return rc;
        HIP_TRY(hipMemcpy(d_items.p, h_items.data(), h_items.size() * sizeof(KdHandItem), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_kd_hand_down, dim3((unsigned)h_items.size()), dim3(KD_BLOCK), 0, 0, (const float*)d_vpos.p,
                           (const int32_t*)d_tv.p, (const KdHandItem*)d_items.p, (KdHandResult*)d_results.p,
                           (int32_t*)d_pool.p, d_used, pool_cap, d_over);
        HIP_TRY(hipGetLastError());
        h_res.resize(h_items.size());
        HIP_TRY(hipMemcpy(h_res.data(), d_results.p, h_res.size() * sizeof(KdHandResult), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(ctr, d_ctr.p, 8, hipMemcpyDeviceToHost));
        if (ctr[1]) return 1;
        for (size_t i = 0; i < who.size(); ++i) {
            const int k = who[i], l = new_left[k];
            for (int c = 0; c < 2; ++c) loff[l + c] = h_res[i].off[c], lcnt[l + c] = h_res[i].cnt[c];
            lcnt[k] = 0;
        }
    }
    // ---- split the leaves that hold many triangles, generation after generation ----
    std::vector<int> n_new_left = new_left, n_axis = axis_of, n_depth = depth_of;
    std::vector<float> n_split = split_of;
    std::vector<KdBox> n_cell = cell;
    int                n_max_depth = max_depth;
    if (refine) {
        std::vector<int> gen;
        for (size_t k = 0; k < n_new_left.size(); ++k)
            if (n_new_left[k] < 0) gen.push_back((int)k);
        std::vector<KdSplitItem>   s_items;
        std::vector<KdSplitResult> s_res;
        bool                       full = false;
        while (!gen.empty() && !full) {
            std::vector<int> who;
            s_items.clear();
            for (int k : gen)
                if (lcnt[k] > prm.leaf_target && n_depth[k] < 60) {
                    KdSplitItem it;
                    it.cell = n_cell[k], it.off = loff[k], it.cnt = lcnt[k];
                    s_items.push_back(it);
                    who.push_back(k);
                }
            gen.clear();
            if (s_items.empty()) break;
            if ((rc = room(s_items.size(), sizeof(KdSplitItem), sizeof(KdSplitResult)))) return rc;
            HIP_TRY(hipMemcpy(d_items.p, s_items.data(), s_items.size() * sizeof(KdSplitItem), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_kd_split, dim3((unsigned)s_items.size()), dim3(KD_BLOCK), 0, 0, (const float*)d_vpos.p,
                               (const int32_t*)d_tv.p, (const KdSplitItem*)d_items.p, (KdSplitResult*)d_results.p, prm,
                               (int32_t*)d_pool.p, d_used, pool_cap, d_over);
            HIP_TRY(hipGetLastError());
            s_res.resize(s_items.size());
            HIP_TRY(hipMemcpy(s_res.data(), d_results.p, s_res.size() * sizeof(KdSplitResult), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(ctr, d_ctr.p, 8, hipMemcpyDeviceToHost));
            if (ctr[1]) return 1;
            for (size_t i = 0; i < who.size(); ++i) { // in node order, as the host loop walks them
                if (n_new_left.size() + 2 > max_nodes) {
                    full = true;
                    break;
                }
                const KdSplitResult& r = s_res[i];
                if (r.axis < 0) continue;
                const int k = who[i], l = (int)n_new_left.size();
                n_new_left[k] = l, n_axis[k] = r.axis, n_split[k] = r.plane;
                KdBox b[2] = {n_cell[k], n_cell[k]};
                b[0].hi[r.axis] = (double)r.plane, b[1].lo[r.axis] = (double)r.plane;
                for (int c = 0; c < 2; ++c) {
                    n_new_left.push_back(-1), n_depth.push_back(n_depth[k] + 1), n_axis.push_back(0);
                    n_split.push_back(0.0f), n_cell.push_back(b[c]);
                    loff.push_back(r.off[c]), lcnt.push_back(r.cnt[c]);
                    gen.push_back(l + c);
                }
                n_max_depth = std::max(n_max_depth, n_depth[k] + 1);
                lcnt[k] = 0;
            }
        }
    }
    // ---- the lists back to the host ----
    HIP_TRY(hipMemcpy(ctr, d_ctr.p, 8, hipMemcpyDeviceToHost));
    std::vector<int> pool((size_t)ctr[0]);
    if (!pool.empty()) HIP_TRY(hipMemcpy(pool.data(), d_pool.p, 4 * pool.size(), hipMemcpyDeviceToHost));
    new_left.swap(n_new_left), axis_of.swap(n_axis), split_of.swap(n_split), depth_of.swap(n_depth), cell.swap(n_cell);
    max_depth = n_max_depth;
    lists.assign(new_left.size(), std::vector<int>());
    for (size_t k = 0; k < lists.size(); ++k)
        if (lcnt[k] > 0) lists[k].assign(pool.begin() + loff[k], pool.begin() + loff[k] + lcnt[k]);
    return 0;
}

extern "C" int rt_scene_create(const RtSceneDesc* desc, int device, RtScene** out)
{
    if (!desc || !out) return fail(RT_ERR_INVALID, "rt_scene_create: null argument");
    *out = nullptr;
    const auto t_create0 = std::chrono::steady_clock::now();
    double     tree_ms = 0.0;
    bool       tree_on_gpu = false;
    const int ndev = rt_device_count();
    const bool stats_only = getenv("RT_KD_STATS") != nullptr; // tree diagnostics run without a GPU
    if (ndev <= 0 && !stats_only) return fail(RT_ERR_NO_DEVICE, "rt_scene_create: no HIP device");
    if (!(stats_only && ndev <= 0) && (device < 0 || device >= ndev))
        return fail(RT_ERR_INVALID, "rt_scene_create: device %d out of range (0..%d)", device, ndev - 1);
    const int nv = desc->n_vertices, nt = desc->n_triangles, nn = desc->n_nodes,
              nm = desc->n_materials, ntc = desc->n_texcoords, ntex = desc->n_textures,
              nnt = desc->n_node_tris;
    if (nv < 0 || nt < 0 || nn < 0 || nm < 0 || ntc < 0 || ntex < 0 || nnt < 0)
        return fail(RT_ERR_INVALID, "rt_scene_create: negative count");
    if ((nv && (!desc->vertex_pos || !desc->vertex_normal)) ||
        (nt && (!desc->tri_vertex || !desc->tri_material)) || (nm && !desc->materials) ||
        (nn && !desc->nodes) || (nnt && !desc->node_tris) || (ntc && !desc->texcoords) ||
        (ntex && !desc->textures))
        return fail(RT_ERR_INVALID, "rt_scene_create: null array with non-zero count");
    if (nt && nm == 0) return fail(RT_ERR_INVALID, "rt_scene_create: triangles but no materials");
    if (nt >= (1 << 28)) return fail(RT_ERR_UNSUPPORTED, "rt_scene_create: too many triangles");
    for (int i = 0; i < nt; ++i) {
        for (int j = 0; j < 3; ++j)
            if (desc->tri_vertex[3 * i + j] < 0 || desc->tri_vertex[3 * i + j] >= nv)
                return fail(RT_ERR_INVALID, "rt_scene_create: triangle %d vertex index out of range", i);
        if (desc->tri_material[i] < 0 || desc->tri_material[i] >= nm)
            return fail(RT_ERR_INVALID, "rt_scene_create: triangle %d material out of range", i);
    }
    for (int i = 0; i < nm; ++i)
        if (desc->materials[i].texture >= ntex)
            return fail(RT_ERR_INVALID, "rt_scene_create: material %d texture out of range", i);
    for (int i = 0; i < ntex; ++i)
        if (!desc->textures[i].rgba || desc->textures[i].width <= 0 || desc->textures[i].height <= 0)
            return fail(RT_ERR_INVALID, "rt_scene_create: texture %d empty", i);

    // ---- breadth-first relabelling of the tree, children adjacent ----
    std::vector<int> order, depth_of;
    std::vector<int> new_left; // by new index
    std::vector<char> seen((size_t)nn, 0);
    int               max_depth = 0;
    if (nn > 0) {
        order.push_back(0);
        depth_of.push_back(0);
        seen[0] = 1;
        for (size_t head = 0; head < order.size(); ++head) {
            const RtKdNode& n = desc->nodes[order[head]];
            const bool      leaf = n.left < 0 && n.right < 0;
            if (!leaf) {
                if (n.left < 0 || n.right < 0 || n.left >= nn || n.right >= nn || seen[n.left] ||
                    seen[n.right] || n.left == n.right)
                    return fail(RT_ERR_INVALID, "rt_scene_create: node %d has bad children", order[head]);
                seen[n.left] = seen[n.right] = 1;
                new_left.push_back((int)order.size());
                order.push_back(n.left);
                order.push_back(n.right);
                depth_of.push_back(depth_of[head] + 1);
                depth_of.push_back(depth_of[head] + 1);
                max_depth = std::max(max_depth, depth_of[head] + 1);
            } else {
                new_left.push_back(-1);
            }
            if (n.tri_count < 0 || n.tri_offset < 0 || (long long)n.tri_offset + n.tri_count > nnt)
                return fail(RT_ERR_INVALID, "rt_scene_create: node %d triangle range out of bounds",
                            order[head]);
        }
    }
    if (max_depth + 1 > 64)
        return fail(RT_ERR_UNSUPPORTED, "rt_scene_create: KD tree depth %d exceeds 63", max_depth);

    // ---- triangle lists per node ----
    // own[k]: the node's AccelTreeNode::triangles.  rank[t]: position of triangle t in the
    // order "nodes breadth first, list order inside a node"; on one root-to-leaf path that is
    // the order the reference's walk tests triangles in (raytracing.cpp:311-338), so among
    // hits with equal t the larger rank is the one the reference ends up with.
    std::vector<std::vector<int>> lists(order.size());
    std::vector<int>              rank((size_t)std::max(nt, 1), 0);
    {
        int r = 0;
        for (size_t k = 0; k < order.size(); ++k) {
            const RtKdNode& n = desc->nodes[order[k]];
            for (int i = 0; i < n.tri_count; ++i) {
                const int t = desc->node_tris[n.tri_offset + i];
                if (t < 0 || t >= nt)
                    return fail(RT_ERR_INVALID, "rt_scene_create: node %d lists triangle %d", order[k], t);
                lists[k].push_back(t);
                rank[t] = r++;
            }
        }
    }
    // Triangles the reference keeps in inner nodes (those touching the node's split plane,
    // KDtree.cpp:136-152) are handed down to every leaf whose cell they overlap, so that a
    // ray only meets triangles of the cells it crosses.  The closest hit over all triangles
    // is unchanged; ties are settled by rank, not by test order.  RT_KD_PUSHDOWN=0 keeps the
    // reference's placement (for A/B timing).
    const char* pd_env = getenv("RT_KD_PUSHDOWN");
    const bool  pushdown = !(pd_env && pd_env[0] == '0');
    // per node: split axis and plane (leaves: unused).  The reference cycles x, y, z with
    // the depth (KDtree.cpp:75-133) and the plane is the left child's end on that axis.
    std::vector<int>   axis_of(order.size(), 0);
    std::vector<float> split_of(order.size(), 0.0f);
    for (size_t k = 0; k < order.size(); ++k) {
        axis_of[k] = depth_of[k] % 3;
        if (new_left[k] >= 0)
            split_of[k] = desc->nodes[desc->nodes[order[k]].left].bounds[2 * axis_of[k] + 1];
    }
    const auto t_tree0 = std::chrono::steady_clock::now();
    if (pushdown && nn > 0 && nt > 0) {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int v = 0; v < nv; ++v)
            for (int a = 0; a < 3; ++a) {
                lo[a] = std::min(lo[a], (double)desc->vertex_pos[3 * v + a]);
                hi[a] = std::max(hi[a], (double)desc->vertex_pos[3 * v + a]);
            }
        typedef KdBox Box;
        std::vector<Box> cell(order.size());
        for (int a = 0; a < 3; ++a) { // outer cells are open: stand-in far beyond the mesh
            const double ext = hi[a] - lo[a] + 10.0;
            cell[0].lo[a] = lo[a] - ext, cell[0].hi[a] = hi[a] + ext;
        }
        auto overlaps = [&](int t, const Box& b) -> bool {
            return kd_overlaps(desc->vertex_pos, desc->tri_vertex, t, b);
        };
        // RT_KD_REBUILD=1: drop the planes of the input tree and let the refinement below build
        // the whole tree from one root leaf holding every listed triangle.
        const char* rb_env = getenv("RT_KD_REBUILD");
        if (rb_env && rb_env[0] == '1') {
            std::vector<char> listed((size_t)nt, 0);
            std::vector<int>  all;
            for (auto& l : lists)
                for (int t : l)
                    if (!listed[t]) listed[t] = 1, all.push_back(t);
            lists.assign(1, all);
            new_left.assign(1, -1);
            depth_of.assign(1, 0);
            axis_of.assign(1, 0);
            split_of.assign(1, 0.0f);
            order.resize(1);
            cell.resize(1);
            max_depth = 0;
        }
        // cells of the input tree, top down
        for (size_t k = 0; k < order.size(); ++k) {
            if (new_left[k] < 0) continue;
            const int    axis = depth_of[k] % 3;
            const int    l = new_left[k], r = l + 1;
            const double plane = desc->nodes[desc->nodes[order[k]].left].bounds[2 * axis + 1];
            cell[l] = cell[k], cell[r] = cell[k];
            cell[l].hi[axis] = plane, cell[r].lo[axis] = plane;
        }
        const char* rf_env = getenv("RT_KD_REFINE");
        const bool  refine = !(rf_env && rf_env[0] == '0');
        const char*  lt_env = getenv("RT_KD_LEAF");
        const char*  ci_env = getenv("RT_KD_CI");
        KdSplitParams prm;
        // Leaves at or below this size are not split (4 until the plane heads, two at a time, made a test far cheaper
        // than a node step; 12 -> 16: final.obj's tree 6741 -> 2999 nodes, its 4K frame -0.5 %, ball 800x800 -6 %,
        // lego +1 %; 20 and more: slower).
        prm.leaf_target = lt_env ? atoi(lt_env) : 16;
        prm.c_trav = 1.0, prm.c_isect = ci_env ? atof(ci_env) : 3.0;
        const char* ce_env = getenv("RT_KD_EMPTY");
        prm.c_empty = ce_env ? atof(ce_env) : 1.0; // (0.8, the usual bonus for a plane with an empty side: 1.5 % slower - an empty leaf is not free here)
        for (int a = 0; a < 3; ++a) {
            const double pad = 0.05 * (hi[a] - lo[a]) + 1e-3;
            prm.mesh_box.lo[a] = lo[a] - pad, prm.mesh_box.hi[a] = hi[a] + pad;
        }
        const size_t max_nodes = (size_t)1 << 22; // (a node names its left child in 30 bits)
        // Both steps run on the device (rt_kdrefine.h); RT_KD_GPU=0, or a triangle pool that
        // overflows, leaves them to the host loops below, which produce the same tree.
        const char* kg_env = getenv("RT_KD_GPU");
        bool        on_gpu = ndev > 0 && !(kg_env && kg_env[0] == '0');
        if (on_gpu) {
            const int rc_gpu = kd_transform_gpu(desc, device, lists, new_left, axis_of, split_of, depth_of, cell,
                                                max_depth, prm, refine, max_nodes);
            if (rc_gpu < 0) return rc_gpu;
            on_gpu = rc_gpu == 0; // 1: pool overflow, nothing was changed
            tree_on_gpu = on_gpu;
        }
        if (!on_gpu)
        for (size_t k = 0; k < order.size(); ++k) {
            if (new_left[k] < 0) continue;
            const int l = new_left[k], r = l + 1;
            for (int t : lists[k]) {
                const bool inl = overlaps(t, cell[l]), inr = overlaps(t, cell[r]);
                if (inl || !inr) lists[l].push_back(t);
                if (inr || !inl) lists[r].push_back(t);
            }
            lists[k].clear();
        }
        // Leaves that ended up with many triangles are split further (surface-area
        // heuristic over the triangles' bounds inside the cell).  The planes of the
        // reference's tree stay; this only adds levels below its leaves, and the
        // closest hit over all triangles does not depend on how they are grouped.
        // RT_KD_REFINE=0 switches it off.  Tried and not kept: candidate planes and counts from
        // the triangles clipped to the cell ("perfect splits": more empty-space cuts, the node
        // budget runs out, 12.3 vs 12.0 ms per 4K frame), building the whole tree this way
        // from one root (RT_KD_REBUILD=1: 14.6 vs 13.3 ms; with the pipeline and the plane heads 6.56 vs 5.66 ms, and
        // 25-40 % slower on the 1080p frames), and dropping planes that only cut an empty
        // slab off where the other child's subtree alone is cheaper by this cost model plus a cost per leaf
        // visit (a fifth of the leaf visits are to empty leaves): 0 to 21 of 3300 planes qualify for a
        // leaf-visit cost of 0 to 16 node steps, frame times unchanged.
        if (refine && !on_gpu) {
            const int    leaf_target = prm.leaf_target;
            const double c_trav = prm.c_trav, c_isect = prm.c_isect, c_empty = prm.c_empty;
            const Box    mesh_box = prm.mesh_box;
            for (size_t k = 0; k < lists.size() && lists.size() + 2 <= max_nodes; ++k) {
                if (new_left[k] >= 0 || (int)lists[k].size() <= leaf_target || depth_of[k] >= 60) continue;
                Box cb = cell[k]; // the part of the cell rays can come from
                for (int a = 0; a < 3; ++a) {
                    cb.lo[a] = std::max(cb.lo[a], mesh_box.lo[a]);
                    cb.hi[a] = std::min(cb.hi[a], mesh_box.hi[a]);
                }
                const double sa = kd_area(cb);
                if (!(sa > 0.0)) continue;
                const size_t n = lists[k].size();
                double       best_cost = c_isect * (double)n;
                int          best_axis = -1;
                double       best_plane = 0.0;
                for (int a = 0; a < 3; ++a)
                    for (size_t i = 0; i < n; ++i) {
                        double mn, mx;
                        kd_tri_extent(desc->vertex_pos, desc->tri_vertex, lists[k][i], a, mn, mx);
                        for (int upper = 0; upper < 2; ++upper) { // candidate planes: the triangles' extents
                            const double pl = upper ? mx : mn;
                            if (!(pl > cb.lo[a] && pl < cb.hi[a])) continue;
                            double cost, plane;
                            if (!kd_split_cost(desc->vertex_pos, desc->tri_vertex, lists[k].data(), (int)n, cb, sa, a,
                                               pl, c_trav, c_isect, c_empty, cost, plane))
                                continue;
                            if (cost < best_cost) best_cost = cost, best_axis = a, best_plane = plane;
                        }
                    }
                if (best_axis < 0) continue;
                Box bl = cell[k], br = cell[k];
                bl.hi[best_axis] = best_plane, br.lo[best_axis] = best_plane;
                std::vector<int> ll, rr;
                for (int t : lists[k]) {
                    const bool inl = overlaps(t, bl), inr = overlaps(t, br);
                    if (inl || !inr) ll.push_back(t);
                    if (inr || !inl) rr.push_back(t);
                }
                if (ll.size() == n && rr.size() == n) continue; // nothing separated
                const int l = (int)lists.size();
                new_left[k] = l;
                axis_of[k] = best_axis;
                split_of[k] = (float)best_plane;
                for (int c = 0; c < 2; ++c) {
                    new_left.push_back(-1);
                    depth_of.push_back(depth_of[k] + 1);
                    axis_of.push_back(0);
                    split_of.push_back(0.0f);
                    cell.push_back(c == 0 ? bl : br);
                    lists.push_back(c == 0 ? ll : rr);
                }
                max_depth = std::max(max_depth, depth_of[k] + 1);
                lists[k].clear();
            }
        }
    }
    tree_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_tree0).count();
    if (max_depth + 1 > 64)
        return fail(RT_ERR_UNSUPPORTED, "rt_scene_create: KD tree depth %d exceeds 63", max_depth);

    if (getenv("RT_KD_STATS")) { // diagnostic: how often each triangle is listed
        std::vector<int> refs((size_t)std::max(nt, 1), 0);
        size_t           total = 0, leaves = 0;
        for (size_t k = 0; k < lists.size(); ++k) {
            if (new_left[k] < 0) ++leaves;
            for (int t : lists[k]) ++refs[t], ++total;
        }
        std::vector<int> idx((size_t)nt);
        for (int t = 0; t < nt; ++t) idx[t] = t;
        std::sort(idx.begin(), idx.end(), [&](int a, int b) { return refs[a] > refs[b]; });
        fprintf(stderr, "kd stats: %zu nodes, %zu leaves, %zu triangle references for %d triangles\n",
                lists.size(), leaves, total, nt);
        size_t cum = 0;
        for (int i = 0; i < std::min(nt, 24); ++i) {
            cum += refs[idx[i]];
            fprintf(stderr, "  tri %d: %d leaves (cumulative %.1f%%)\n", idx[i], refs[idx[i]], 100.0 * cum / total);
        }
    }
    if (ndev <= 0) return fail(RT_ERR_NO_DEVICE, "rt_scene_create: no HIP device");
    if (lists.empty()) { // a description without a tree: one empty leaf, every ray misses
        lists.emplace_back();
        new_left.push_back(-1);
        axis_of.push_back(0);
        split_of.push_back(0.0f);
        depth_of.push_back(0);
    }
    std::vector<DNode>  nodes(lists.size());
    std::vector<float4> slots, edges, heads;
    const float4        zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (size_t k = 0; k < lists.size(); ++k) {
        DNode     dn;
        const int axis = axis_of[k];
        dn.left = new_left[k];
        dn.split = dn.left >= 0 ? split_of[k] : 0.0f;
        if (!lists[k].empty() && (slots.size() / RT_SLOT_F4) % 2 == 1) { // lists start at even slots: heads go in pairs
            slots.insert(slots.end(), RT_SLOT_F4, zero4);
            edges.insert(edges.end(), 3, zero4);
            heads.push_back(zero4);
        }
        dn.tri_off = (int)(slots.size() / RT_SLOT_F4);
        if (lists[k].size() >= (1u << 28))
            return fail(RT_ERR_UNSUPPORTED, "rt_scene_create: node triangle list too long");
        dn.cnt_axis = (int)lists[k].size() | (axis << 28);
        for (int t : lists[k]) {
            const float* v0 = desc->vertex_pos + 3 * desc->tri_vertex[3 * t];
            const float* v1 = desc->vertex_pos + 3 * desc->tri_vertex[3 * t + 1];
            const float* v2 = desc->vertex_pos + 3 * desc->tri_vertex[3 * t + 2];
            // separately rounded, exactly what raytracing.cpp:366-368 computes per test
            volatile float e1x = v0[0] - v2[0], e1y = v0[1] - v2[1], e1z = v0[2] - v2[2];
            volatile float e2x = v1[0] - v2[0], e2y = v1[1] - v2[1], e2z = v1[2] - v2[2];
            volatile float p1 = e1x * e2y, p4 = e1z * e2y;
            float          idx;
            memcpy(&idx, &t, 4);
            // the plane head (tri_heads_keep): (e1 x e2, (e1 x e2) . v2) / B in double, rounded once
            const double a1[3] = {(double)e1x, (double)e1y, (double)e1z}, a2[3] = {(double)e2x, (double)e2y, (double)e2z};
            double       nrm[3], B = 0.0;
            for (int c = 0; c < 3; ++c) {
                const int j = (c + 1) % 3, l = (c + 2) % 3;
                nrm[c] = a1[j] * a2[l] - a1[l] * a2[j];
                B = std::max(B, std::fabs(a1[j] * a2[l]) + std::fabs(a1[l] * a2[j]));
            }
            float4 hd = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (B > 9.1e-13 && B < 1.0e12) { // 2^-40 .. 2^40: no quantity of the reference's test leaves the normal range
                const double kk = nrm[0] * (double)v2[0] + nrm[1] * (double)v2[1] + nrm[2] * (double)v2[2];
                hd = make_float4((float)(nrm[0] / B), (float)(nrm[1] / B), (float)(nrm[2] / B), (float)(kk / B));
                if (!std::isfinite(hd.w)) hd = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            // the edge record (tri_edges_keep): ga = (e2 x n) / n.n, gb = (n x e1) / n.n and their offsets, scaled so
            // that no component exceeds 1, in double, rounded once; zero (never decides anything) beside a zero head
            float4 g0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), g1 = g0, g2 = g0;
            const double nn = nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2];
            if ((hd.x != 0.0f || hd.y != 0.0f || hd.z != 0.0f) && nn > 0.0 && std::isfinite(nn)) {
                double ga[3], gb[3], G = 0.0;
                for (int c = 0; c < 3; ++c) {
                    const int j = (c + 1) % 3, l = (c + 2) % 3;
                    ga[c] = (a2[j] * nrm[l] - a2[l] * nrm[j]) / nn;
                    gb[c] = (nrm[j] * a1[l] - nrm[l] * a1[j]) / nn;
                    G = std::max(G, std::max(std::fabs(ga[c]), std::fabs(gb[c])));
                }
                const double sc = 1.0 / G;
                if (G > 0.0 && std::isfinite(sc) && sc > 1e-30 && sc < 1e30) {
                    double ka = 0.0, kb = 0.0;
                    for (int c = 0; c < 3; ++c) ga[c] *= sc, gb[c] *= sc, ka -= ga[c] * (double)v2[c], kb -= gb[c] * (double)v2[c];
                    g0 = make_float4((float)ga[0], (float)gb[0], (float)ga[1], (float)gb[1]); // the two coordinates side by side
                    g1 = make_float4((float)ga[2], (float)gb[2], (float)ka, (float)kb);
                    g2 = make_float4((float)sc, 0.0f, 0.0f, 0.0f);
                    if (!std::isfinite(g1.z) || !std::isfinite(g1.w)) g0 = g1 = g2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
            }
            edges.push_back(g0), edges.push_back(g1), edges.push_back(g2);
            heads.push_back(hd); // (interleaved in pairs below)
            slots.push_back(make_float4(e1x, e1y, e1z, e2x));
            slots.push_back(make_float4(e2y, e2z, v2[0], v2[1]));
            slots.push_back(make_float4(v2[2], p1, p4, idx));
        }
        nodes[k] = dn;
    }
    if (slots.size() / RT_SLOT_F4 >= (1u << 28))
        return fail(RT_ERR_UNSUPPORTED, "rt_scene_create: too many triangle references");
    if (heads.size() % 2 == 1) heads.push_back(zero4);
    for (size_t p2 = 0; p2 + 1 < heads.size(); p2 += 2) { // n0 k0, n1 k1  ->  n0.x n1.x n0.y n1.y, n0.z n1.z k0 k1
        const float4 a = heads[p2], b = heads[p2 + 1];
        heads[p2] = make_float4(a.x, b.x, a.y, b.y);
        heads[p2 + 1] = make_float4(a.z, b.z, a.w, b.w);
    }
    if (nodes.size() >= (1u << 27)) return fail(RT_ERR_UNSUPPORTED, "rt_scene_create: too many tree nodes");
    std::vector<uint2> knodes(nodes.size());
    for (size_t k = 0; k < nodes.size(); ++k) {
        const DNode& n = nodes[k];
        if (n.left >= 0) {
            unsigned bits;
            memcpy(&bits, &n.split, 4);
            knodes[k] = make_uint2(bits, ((unsigned)(n.cnt_axis >> 28) << 30) | ((unsigned)n.left * 8u)); // (the left child's record, in bytes)
        } else {
            knodes[k] = make_uint2((unsigned)n.tri_off, (3u << 30) | (unsigned)(n.cnt_axis & 0x0fffffff));
        }
    }

    std::vector<float4> shade((size_t)nt * 3);
    std::vector<int4>   info((size_t)nt);
    for (int t = 0; t < nt; ++t) {
        for (int j = 0; j < 3; ++j) {
            const float* nrm = desc->vertex_normal + 3 * desc->tri_vertex[3 * t + j];
            shade[3 * t + j] = make_float4(nrm[0], nrm[1], nrm[2], 0.0f);
        }
        int tc[3] = {0, 0, 0};
        for (int j = 0; j < 3; ++j) {
            const int v = desc->tri_texcoord ? desc->tri_texcoord[3 * t + j] : 0;
            tc[j] = (v >= 0 && v < ntc) ? v : 0;
        }
        info[t] = make_int4(desc->tri_material[t], tc[0], tc[1], tc[2]);
    }
    std::vector<float> texcoords((size_t)std::max(ntc, 1) * 3, 0.0f);
    if (ntc) memcpy(texcoords.data(), desc->texcoords, sizeof(float) * 3 * (size_t)ntc);
    std::vector<float4> mats((size_t)nm * 4);
    for (int i = 0; i < nm; ++i) {
        const RtMaterial& m = desc->materials[i];
        float             fi, fh, ft;
        const int32_t     tex = m.texture < 0 ? -1 : m.texture;
        memcpy(&fi, &m.illum, 4), memcpy(&fh, &m.has_Ni, 4), memcpy(&ft, &tex, 4);
        mats[4 * i] = make_float4(m.Kd[0], m.Kd[1], m.Kd[2], m.Ns);
        mats[4 * i + 1] = make_float4(m.Ks[0], m.Ks[1], m.Ks[2], m.Ni);
        mats[4 * i + 2] = make_float4(fi, fh, ft, 0.0f);
        mats[4 * i + 3] = make_float4(0, 0, 0, 0);
    }
    std::vector<int4>    texinfo((size_t)std::max(ntex, 1), make_int4(0, 1, 1, 0));
    std::vector<uint8_t> texels;
    for (int i = 0; i < ntex; ++i) {
        const RtTexture& t = desc->textures[i];
        const size_t     body = (size_t)t.width * t.height * 4, pad = (size_t)t.width * 2 * 4;
        if (texels.size() + body + pad > 0x7fffffffu)
            return fail(RT_ERR_UNSUPPORTED, "rt_scene_create: textures exceed 2 GiB");
        texinfo[i] = make_int4((int)texels.size(), t.width, t.height, 0);
        texels.insert(texels.end(), t.rgba, t.rgba + body);
        texels.insert(texels.end(), pad, (uint8_t)0);
    }
    if (texels.empty()) texels.assign(64, 0);

    HIP_TRY(hipSetDevice(device));
    {
        // The no-allocation entry point looks its array's address up once per device; doing it here means
        // that a process which has made a scene on a device never does it (and whatever the runtime loads
        // for it) inside that entry point.
        float* unused = nullptr;
        const int rc_s = scratch_for_device(device, &unused);
        if (rc_s && rc_s != RT_ERR_UNSUPPORTED) return rc_s; // (a device number beyond that entry point's table)
    }
    RtScene* s = new RtScene();
    s->device = device;
    s->tree_depth = max_depth;
    {
        auto on = [](const char* name) { const char* v = getenv(name); return v && v[0] == '1'; };
        auto num = [](const char* name) { const char* v = getenv(name); return v ? atoi(v) : 0; };
        s->cfg.megakernel = on("RT_MEGAKERNEL"), s->cfg.multilaunch = on("RT_WF_MULTILAUNCH");
        s->cfg.kd_stats = getenv("RT_KD_STATS") != nullptr;
        s->cfg.pipes = std::max(1, num("RT_WF_PIPES")), s->cfg.region = num("RT_WF_REGION");
        s->cfg.stack_lds = num("RT_WF_STACK_LDS"), s->cfg.budget_mb = num("RT_WF_BUDGET_MB");
        const char* fb = getenv("RT_WF_FEEDBACK");
        s->wf_feedback = !(fb && fb[0] == '0');
        if (const char* h = getenv("RT_WF_HOT")) s->wf_hot = (float)atof(h);
    }
    int rc;
    if ((rc = upload(&s->d_nodes, nodes, &s->bytes)) || (rc = upload(&s->d_tris, slots, &s->bytes)) || (rc = upload(&s->d_edges, edges, &s->bytes)) || (rc = upload(&s->d_heads, heads, &s->bytes)) ||
        (rc = upload(&s->d_shade, shade, &s->bytes)) || (rc = upload(&s->d_info, info, &s->bytes)) ||
        (rc = upload(&s->d_texcoords, texcoords, &s->bytes)) ||
        (rc = upload(&s->d_mats, mats, &s->bytes)) || (rc = upload(&s->d_texinfo, texinfo, &s->bytes)) ||
        (rc = upload(&s->d_texels, texels, &s->bytes)) || (rc = upload(&s->d_rank, rank, &s->bytes)) || (rc = upload(&s->d_knodes, knodes, &s->bytes))) {
        rt_scene_destroy(s);
        return rc;
    }
    hipError_t e = hipMalloc((void**)&s->d_ctr, sizeof(DCounters));
    if (e == hipSuccess) e = hipEventCreate(&s->ev0);
    if (e == hipSuccess) e = hipEventCreate(&s->ev1);
    if (e != hipSuccess) {
        rt_scene_destroy(s);
        return fail(RT_ERR_HIP, "rt_scene_create: %s", hipGetErrorString(e));
    }
    s->d.nodes = (const DNode*)s->d_nodes;
    s->d.tris = (const float4*)s->d_tris;
    s->d.edges = (const float4*)s->d_edges;
    s->d.heads = (const float4*)s->d_heads;
    s->d.tri_shade = (const float4*)s->d_shade;
    s->d.tri_info = (const int4*)s->d_info;
    s->d.texcoords = (const float*)s->d_texcoords;
    s->d.mats = (const float4*)s->d_mats;
    s->d.tex_info = (const int4*)s->d_texinfo;
    s->d.texels = (const uint8_t*)s->d_texels;
    s->d.tri_rank = (const int32_t*)s->d_rank;
    s->d.knodes = (const uint2*)s->d_knodes;
    s->n_knodes = knodes.size();
    s->d.n_nodes = (int)nodes.size();
    s->d.n_slots = (int)(slots.size() / RT_SLOT_F4);
    s->d.n_triangles = nt;
    s->d.stack_entries = max_depth + 1;
    {
        float v = 0.0f;
        for (int i = 0; i < nv; ++i)
            v = std::max(v, std::fabs(desc->vertex_pos[3 * i]) + std::fabs(desc->vertex_pos[3 * i + 1]) +
                                std::fabs(desc->vertex_pos[3 * i + 2]));
        s->d.vert_l1 = v * 1.000001f;
    }
    s->pushdown = pushdown;
    {
        uint64_t h = 1469598103934665603ull;
        auto     mix = [&](uint32_t v) {
            for (int b = 0; b < 4; ++b) h = (h ^ ((v >> (8 * b)) & 0xffu)) * 1099511628211ull;
        };
        for (const uint2& k : knodes) mix(k.x), mix(k.y);
        for (size_t i = 0; i < slots.size() / RT_SLOT_F4; ++i) mix((uint32_t)__builtin_bit_cast(int32_t, slots[RT_SLOT_F4 * i + 2].w));
        s->tree_hash = h;
    }
    s->tree_ms = (float)tree_ms, s->tree_on_gpu = tree_on_gpu ? 1 : 0;
    s->create_ms = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create0).count();
    for (int i = 0; i < nm; ++i) // raytracing.cpp:177,206: two children only if Ni > 1 and not a mirror
        if (desc->materials[i].Ni > 1.0f && desc->materials[i].illum != 3) s->refractive = true;
    if (!(s->pushdown && !s->cfg.megakernel)) // never silently: the frame path of this scene is the slow one
        fprintf(stderr, "rt_scene_create: RT_MEGAKERNEL / RT_KD_PUSHDOWN=0 set - frames of this scene render with one "
                        "thread per camera sample, not with the ray-queue pipeline\n");
    *out = s;
    return RT_OK;
}

extern "C" int rt_scene_info(const RtScene* s, RtSceneInfo* info)
{
    if (!s || !info) return fail(RT_ERR_INVALID, "rt_scene_info: null argument");
    info->n_nodes = s->d.n_nodes;
    info->n_triangles = s->d.n_triangles;
    info->tree_depth = s->tree_depth;
    info->n_slots = s->d.n_slots;
    info->device_bytes = s->bytes;
    info->tree_hash = s->tree_hash;
    info->create_ms = s->create_ms, info->tree_ms = s->tree_ms;
    info->tree_on_gpu = s->tree_on_gpu;
    info->render_path = (s->pushdown && !s->cfg.megakernel) ? (s->cfg.multilaunch ? 1 : 0) : 2;
    return RT_OK;
}

extern "C" int rt_scene_set_timing(RtScene* s, int enable)
{
    if (!s) return fail(RT_ERR_INVALID, "rt_scene_set_timing: null scene");
    s->timing = enable != 0;
    s->ev_used = 0;
    return RT_OK;
}

extern "C" int rt_scene_collect_timing(RtScene* s, RtKernelTimes* out)
{
    if (!s || !out) return fail(RT_ERR_INVALID, "rt_scene_collect_timing: null argument");
    memset(out, 0, sizeof *out);
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    for (size_t i = 0; i < s->ev_used; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, s->ev_pool[2 * i], s->ev_pool[2 * i + 1]) != hipSuccess) continue;
        switch (s->ev_kind[i]) {
        case K_TRACE_CLOSEST: out->trace_closest_ms += ms, out->trace_closest_launches++; break;
        case K_TRACE_ANY: out->trace_any_ms += ms, out->trace_any_launches++; break;
        default: out->frame_ms += ms, out->frames++; break;
        }
    }
    s->ev_used = 0;
    return RT_OK;
}

// ---------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------
static size_t stack_bytes(const RtScene* s) { return (size_t)s->d.stack_entries * RT_BLOCK * sizeof(uint2); }

static int make_dparams(const RtParams* p, const float* camera, const float* lights, int n_lights,
                        DParams* d)
{
    if (!p) return fail(RT_ERR_INVALID, "null params");
    if (n_lights < 0 || n_lights > RT_MAX_LIGHTS)
        return fail(RT_ERR_INVALID, "n_lights %d out of range (0..%d)", n_lights, RT_MAX_LIGHTS);
    if (n_lights && !lights) return fail(RT_ERR_INVALID, "null lights");
    if (p->msaa != 1 && p->msaa != 4 && p->msaa != 16)
        return fail(RT_ERR_INVALID, "msaa must be 1, 4 or 16 (got %d)", p->msaa);
    memset(d, 0, sizeof *d);
    d->W = p->width, d->H = p->height, d->msaa = p->msaa, d->n_lights = n_lights;
    d->backgroundlighting = p->backgroundlighting;
    d->diffusePower = p->diffusePower;
    d->ShadowFactor = p->ShadowFactor;
    {
        volatile float q = n_lights > 0 ? p->ShadowFactor / (float)n_lights : 0.0f; // (one rounded float division, as the kernels made it per hit)
        d->shadow_per_light = q;
    }
    memcpy(d->bg, p->backgroundColor, 12);
    if (camera) memcpy(d->cam, camera, sizeof d->cam);
    if (n_lights) memcpy(d->lights, lights, sizeof(float) * 3 * (size_t)n_lights);
    if (p->shadowSamples < 0 || p->shadowSamples > 6)
        return fail(RT_ERR_INVALID, "shadowSamples must be 0..6 (got %d)", p->shadowSamples);
    if (p->shadowSamples > 1) { // raytracing.cpp:147-150, the loop in float as written there
        const int      n = p->shadowSamples;
        volatile float delta = 2.f / (float)(n - 1 + 0.000000000000000000001f);
        int            cnt = 0;
        for (volatile float x = -1.f; x <= 1.f; x += delta)
            if (cnt < 8) d->soft_x[cnt++] = x;
        d->soft_n = cnt;
        d->soft_half = delta / 2.f, d->soft_quarter = delta / 4.f;
        d->soft_w = 1.0f / n / n / n;
        d->soft_radius = p->shadowRadius;
    }
    return RT_OK;
}

static int set_partition(const RtParams* p, int rank, DParams* d)
{
    if (p->width < 2 || p->height < 2 || p->width > 32768 || p->height > 32768)
        return fail(RT_ERR_INVALID, "frame size %dx%d out of range (2..32768)", p->width, p->height);
    int tw = p->tile_w, th = p->tile_h, world = p->world;
    if (tw == 0 && th == 0) {
        tw = th = 32;
        world = 1;
        rank = 0;
    }
    if (tw < 8 || th < 8 || (tw & 7) || (th & 7) || tw > 1024 || th > 1024)
        return fail(RT_ERR_INVALID, "tile size %dx%d must be multiples of 8 in 8..1024", tw, th);
    int span = p->rank_span <= 0 ? 1 : p->rank_span;
    if (p->tile_w == 0 && p->tile_h == 0) span = 1;
    if (world < 1 || rank < 0 || rank >= world || span > world - rank)
        return fail(RT_ERR_INVALID, "rank %d (span %d) / world %d invalid", rank, span, world);
    d->tile_w = tw, d->tile_h = th, d->rank = rank, d->world = world, d->span = span;
    d->tiles_x = (p->width + tw - 1) / tw;
    d->tiles_y = (p->height + th - 1) / th;
    // tiles t with t % world in [rank, rank + span): span per full round, the head of the last
    const long long n_tiles = (long long)d->tiles_x * d->tiles_y;
    const long long rem = n_tiles % world;
    d->n_my_tiles = (int)((n_tiles / world) * span + std::max<long long>(0, std::min<long long>(rem - rank, span)));
    return RT_OK;
}

extern "C" int64_t rt_tile_pixels(const RtParams* params, int32_t rank)
{
    DParams d;
    memset(&d, 0, sizeof d);
    if (!params || set_partition(params, rank, &d) != RT_OK) return -1;
    return (int64_t)d.n_my_tiles * d.tile_w * d.tile_h;
}

static void read_stats(RtScene* s, RtStats* st, hipStream_t stream)
{
    DCounters c;
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(&c, s->d_ctr, sizeof c, hipMemcpyDeviceToHost);
    st->rays_closest = c.rays_closest;
    st->rays_shadow = c.rays_shadow;
    st->tri_tests = c.tri_tests;
    st->node_visits = c.node_visits;
    st->camera_rays = c.camera_rays;
    st->shaded_hits = c.shaded_hits;
    st->child_rays = c.child_rays;
#ifdef WF_DIAG
    {
        unsigned long long d[16] = {0}, z[16] = {0};
        (void)hipMemcpyFromSymbol(d, HIP_SYMBOL(g_diag), sizeof d);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_diag), z, sizeof z);
        fprintf(stderr, "diag: node loop lanes %.4f (%llu of %llu), triangle loop lanes %.4f (%llu of %llu)\n",
                (double)c.node_visits / (double)(d[0] ? d[0] : 1), (unsigned long long)c.node_visits, d[0],
                (double)c.tri_tests / (double)(d[1] ? d[1] : 1), (unsigned long long)c.tri_tests, d[1]);
        fprintf(stderr, "diag: leaf visits: %llu empty, %llu with triangles\n", d[2], d[3]);
        fprintf(stderr, "diag: triangle loop: %llu wave trips, %llu with a lane past the plane head (%.3f), %llu such lanes (%.3f of the tests)\n",
                d[1] / 128, d[8], (double)d[8] / (double)(d[1] ? d[1] / 64 : 1), d[9], (double)d[9] / (double)(c.tri_tests ? c.tri_tests : 1));
        fprintf(stderr, "diag: exact tests: %llu wave trips with a lane that hits (%.3f of the trips past the head), %llu such lanes (%.3f of the lanes past the head)\n",
                d[12], (double)d[12] / (double)(d[8] ? d[8] : 1), d[13], (double)d[13] / (double)(d[9] ? d[9] : 1));
        fprintf(stderr, "diag: edge records: %llu wave trips with a lane past them (%.3f of the trips past the head), %llu such lanes\n",
                d[14], (double)d[14] / (double)(d[8] ? d[8] : 1), d[15]);
        fprintf(stderr, "diag: leaves with triangles visited by waves: %llu; trips of the triangle loop (two triangles each): %llu\n", d[10], d[11]);
        fprintf(stderr, "diag: shadow rays %llu, occluded %llu, their tests %llu and node steps %llu\n", d[4], d[5], d[6], d[7]);
    }
#endif
#ifdef WF_SECT
    {
        unsigned long long d[16] = {0}, z[16] = {0};
        (void)hipMemcpyFromSymbol(d, HIP_SYMBOL(g_sect), sizeof d);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sect), z, sizeof z);
        static const char* names[10] = {"take a batch", "form or fetch the ray", "closest-hit walk", "pixel-centre walk", "shading",
                                        "fetch a shadow ray", "shadow walk", "report the answer", "wait at the barrier", "-"};
        double sum = 0;
        for (int i = 0; i < 10; ++i) sum += (double)d[i];
        for (int i = 0; i < 9; ++i) fprintf(stderr, "sect: %-24s %6.2f %%  (%llu cycles over all waves)\n", names[i], 100.0 * (double)d[i] / (sum > 0 ? sum : 1), d[i]);
    }
#endif
    float ms = 0;
    (void)hipEventElapsedTime(&ms, s->ev0, s->ev1);
    st->kernel_ms = ms;
    st->reserved = 0;
}

static inline unsigned blocks_for(long long n, int block) { return (unsigned)((n + block - 1) / block); }

// ---------------------------------------------------------------------------
// batched rays
// ---------------------------------------------------------------------------

extern "C" int rt_check_hits(RtScene* s, int64_t n, const float* origins, const float* dests,
                             int32_t* tri, float* a, float* b, float* t, float* hitpoint)
{
    if (!s) return fail(RT_ERR_INVALID, "rt_check_hits: null scene");
    if (n < 0 || n > (int64_t)1 << 31) return fail(RT_ERR_INVALID, "rt_check_hits: n out of range");
    if (n == 0) return RT_OK;
    if (!origins || !dests) return fail(RT_ERR_INVALID, "rt_check_hits: null rays");
    HIP_TRY(hipSetDevice(s->device));
    DevBuf dor, dds, dtri, da, db, dt, dhp;
    int    rc;
    if ((rc = dor.alloc(12 * n)) || (rc = dds.alloc(12 * n)) || (rc = dtri.alloc(4 * n)) ||
        (rc = da.alloc(4 * n)) || (rc = db.alloc(4 * n)) || (rc = dt.alloc(4 * n)) ||
        (rc = dhp.alloc(12 * n)))
        return rc;
    HIP_TRY(hipMemcpy(dor.p, origins, 12 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dds.p, dests, 12 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(s->d_ctr, 0, sizeof(DCounters), 0));
    hipLaunchKernelGGL(k_check_hits, dim3(blocks_for(n, RT_BLOCK)), dim3(RT_BLOCK), stack_bytes(s), 0,
                       s->d, (long long)n, (const float*)dor.p, (const float*)dds.p, (int*)dtri.p,
                       (float*)da.p, (float*)db.p, (float*)dt.p, (float*)dhp.p, s->d_ctr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (tri) HIP_TRY(hipMemcpy(tri, dtri.p, 4 * n, hipMemcpyDeviceToHost));
    if (a) HIP_TRY(hipMemcpy(a, da.p, 4 * n, hipMemcpyDeviceToHost));
    if (b) HIP_TRY(hipMemcpy(b, db.p, 4 * n, hipMemcpyDeviceToHost));
    if (t) HIP_TRY(hipMemcpy(t, dt.p, 4 * n, hipMemcpyDeviceToHost));
    if (hitpoint) HIP_TRY(hipMemcpy(hitpoint, dhp.p, 12 * n, hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_trace_rays(RtScene* s, const float* lights, int32_t n_lights, const RtParams* params,
                             int32_t k, float prev_Ni, int64_t n, const float* origins,
                             const float* dests, float* rgb, float* min_t, int32_t* tri, RtStats* stats)
{
    if (!s) return fail(RT_ERR_INVALID, "rt_trace_rays: null scene");
    if (k < 0 || k > 5) return fail(RT_ERR_INVALID, "rt_trace_rays: k %d out of range (0..5)", k);
    if (n < 0 || n > (int64_t)1 << 31) return fail(RT_ERR_INVALID, "rt_trace_rays: n out of range");
    DParams P;
    int     rc = make_dparams(params, nullptr, lights, n_lights, &P);
    if (rc) return rc;
    if (stats) memset(stats, 0, sizeof *stats);
    if (n == 0) return RT_OK;
    if (!origins || !dests || !rgb) return fail(RT_ERR_INVALID, "rt_trace_rays: null array");
    HIP_TRY(hipSetDevice(s->device));
    DevBuf dor, dds, drgb, dmt, dtri;
    if ((rc = dor.alloc(12 * n)) || (rc = dds.alloc(12 * n)) || (rc = drgb.alloc(12 * n)) ||
        (rc = dmt.alloc(4 * n)) || (rc = dtri.alloc(4 * n)))
        return rc;
    HIP_TRY(hipMemcpy(dor.p, origins, 12 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dds.p, dests, 12 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(s->d_ctr, 0, sizeof(DCounters), 0));
    HIP_TRY(hipEventRecord(s->ev0, 0));
    hipLaunchKernelGGL(k_trace_rays, dim3(blocks_for(n, RT_BLOCK)), dim3(RT_BLOCK), stack_bytes(s), 0,
                       s->d, P, k, prev_Ni, (long long)n, (const float*)dor.p, (const float*)dds.p,
                       (float*)drgb.p, (float*)dmt.p, (int*)dtri.p, s->d_ctr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->ev1, 0));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(rgb, drgb.p, 12 * n, hipMemcpyDeviceToHost));
    if (min_t) HIP_TRY(hipMemcpy(min_t, dmt.p, 4 * n, hipMemcpyDeviceToHost));
    if (tri) HIP_TRY(hipMemcpy(tri, dtri.p, 4 * n, hipMemcpyDeviceToHost));
    if (stats) read_stats(s, stats, 0);
    return RT_OK;
}

// ---------------------------------------------------------------------------
// frames
// ---------------------------------------------------------------------------
static int ensure_workspace(RtScene* s, long long n_slots, int msaa)
{
    if (msaa == 1) return RT_OK;
    if ((size_t)n_slots > s->flags_cap) {
        if (s->d_flags) (void)hipFree(s->d_flags);
        s->d_flags = nullptr, s->flags_cap = 0;
        HIP_TRY(hipMalloc((void**)&s->d_flags, (size_t)n_slots));
        s->flags_cap = (size_t)n_slots;
    }
    const size_t need = (size_t)n_slots * msaa;
    if (need > s->samples_cap) {
        if (s->d_samples) (void)hipFree(s->d_samples);
        s->d_samples = nullptr, s->samples_cap = 0;
        HIP_TRY(hipMalloc((void**)&s->d_samples, need * sizeof(float4)));
        s->samples_cap = need;
    }
    return RT_OK;
}

// ---------------------------------------------------------------------------
// wavefront renderer (rt_wavefront.h)
// ---------------------------------------------------------------------------
// Traversal stack entries per thread kept in LDS.  Measured per 4K frame: 10.59 / 10.53 / 10.55 / 10.69 ms
// with 12 / 10 / 8 / 6; six leave room for the colour sums of a region of 512 samples beside
// them at seven blocks per CU (seven entries: six blocks, 10.8 vs 10.3 ms).
#define WF_STACK_LDS 5
#define WF_ACC_SAMPLES 512 // samples per region the single launch holds colour sums for (24 bytes each, LDS)
#define WF_WORKSPACE_BUDGET ((size_t)12 << 30) // bytes of queues per scene at most

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static int wf_prepare(RtScene* s)
{
    if (s->wf_grid_trace) return RT_OK;
    // LDS holds the first WF_STACK_LDS entries of every thread's traversal stack (fewer if the
    // tree is shallower); the rest of the tree's depth is backed by a buffer in HBM that is
    // touched only by the rare ray with that many far children pending.
    int cap = s->cfg.stack_lds > 0 ? s->cfg.stack_lds : WF_STACK_LDS;
    cap = std::max(1, std::min(cap, (int)s->d.stack_entries));
    if (cap < 2 && s->d.stack_entries >= 2) cap = 2; // a pop reads the two topmost entries with one instruction
    s->d.stk_cap = cap;
    s->wf_lds = (size_t)cap * WF_BLOCK * WF_STACK_ENTRY_BYTES; // (a tree of one node never pushes: nothing is read)
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, s->device));
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_wf_trace<false>, WF_BLOCK, s->wf_lds));
    const int g_trace = std::max(1, per_cu) * prop.multiProcessorCount;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_wf_trace<true>, WF_BLOCK, s->wf_lds));
    const int g_any = std::max(1, per_cu) * prop.multiProcessorCount;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_wf_pipeline<false, false>, WF_BLOCK, s->wf_lds));
    const int g_pipe = std::max(1, per_cu) * prop.multiProcessorCount;
    if (s->cfg.kd_stats)
        fprintf(stderr, "wavefront: stack %d entries (%d in LDS, %zu B per block), pipeline %d blocks per CU\n",
                (int)s->d.stack_entries, cap, s->wf_lds, per_cu);
    s->wf_cap_lo = s->cfg.stack_lds > 0 ? cap : std::min(cap, 2); // the fewest the single launch may settle for (an explicit RT_WF_STACK_LDS is kept)
    const size_t deep = (size_t)(s->d.stack_entries - s->wf_cap_lo);
    const size_t blocks = (size_t)std::max(g_trace, std::max(g_any, g_pipe));
    const size_t n_deep = std::max<size_t>(deep * blocks * WF_BLOCK, 64);
    HIP_TRY(hipMalloc(&s->d_stk_spill, n_deep * WF_STACK_ENTRY_BYTES));
    s->d.stk_spill = (uint2*)s->d_stk_spill;
    s->wf_grid_trace = g_trace, s->wf_grid_any = g_any, s->wf_grid_pipe = g_pipe;
    s->wf_cus = prop.multiProcessorCount;
    s->wf_lds_per_cu = (size_t)prop.maxSharedMemoryPerMultiProcessor;
    return RT_OK;
}

struct WfPlan {
    long long chunk_slots; // pixel slots per chunk (a multiple of S)
    int       S;           // pixel slots per region
    int       G, R;        // regions per chunk, ray slots per region
    int       n_samples;   // samples per chunk
    int       ns;          // shadow rays per hit and light
};

static size_t wf_rec_stride(int L, int ns)
{
    return (sizeof(WfRecord) + 4 * (size_t)L + (ns > 1 ? 4 * (size_t)((L + 3) / 4) : 0) + 15) & ~(size_t)15;
}

// Bytes per region: `multi` adds what only the one-launch-per-stage path uses.
static size_t wf_region_bytes(long long R, int S, int msaa, int L, int ns, bool multi)
{
    const size_t rays = 2 * (12 + 12 + 12);
    const size_t shadow = 2 * (size_t)L * ns * 16 * (ns > 1 ? 2 : 1);
    const size_t rec = 2 * wf_rec_stride(L, ns);
    return (size_t)R * (rays + shadow + rec + (multi ? 16 : 0)) + (size_t)S * msaa * (3 * 8 + 4 + 4) +
           (size_t)S * (multi ? 1 : 8) + 64;
}

static WfPlan wf_plan(const RtScene* s, const DParams& P, long long n_slots, int n_pipes, int S, bool multi)
{
    // rays of one level per camera sample: 1 along mirror chains, up to 2^4 with refraction
    // (levels k = 0..4 spawn, raytracing.cpp:166-168)
    const int       growth = s->refractive ? 16 : 1;
    const int       L = std::max(P.n_lights, 1);
    const long long R = (long long)S * P.msaa * growth;
    const int       ns = P.soft_n > 0 ? P.soft_n * P.soft_n * P.soft_n : 1;
    const size_t    per_region = wf_region_bytes(R, S, P.msaa, L, ns, multi);
    size_t budget = WF_WORKSPACE_BUDGET;
    if (s->cfg.budget_mb > 0) budget = (size_t)s->cfg.budget_mb << 20; // tests: force a frame through several chunks
    long long g_max = (long long)(budget / n_pipes / per_region);
    g_max = std::min<long long>(g_max, (long long)0x7fffffff / (R * L * ns) / 2 - 1);
    g_max = std::max<long long>(g_max, 1);
    const long long g_need = (n_slots + S - 1) / S;
    WfPlan          pl;
    pl.G = (int)std::min(g_need, g_max);
    pl.R = (int)R;
    pl.S = S;
    pl.ns = ns;
    pl.chunk_slots = (long long)pl.G * S;
    pl.n_samples = (int)(pl.chunk_slots * P.msaa);
    return pl;
}

#define WF_MAX_PIPES 4
#define WF_N_BUFFERS 24

static size_t wf_chunk_bytes(const WfPlan& pl, int L, bool multi, size_t* off)
{
    const size_t cap = (size_t)pl.G * pl.R, ns = (size_t)pl.n_samples, G = (size_t)pl.G;
    size_t       need = 0;
    int          k = 0;
    auto         take = [&](size_t bytes) {
        if (off) off[k] = need;
        ++k;
        need += align256(bytes);
    };
    for (int q = 0; q < 2; ++q) take(cap * 12), take(cap * 12), take(cap * 12), take(G * 4); // 0..7 ray queues: origin, destination, aux
    take(multi ? cap * 16 : 0);                                   // 8 hits
    take(2 * cap * 16 * (size_t)L * pl.ns);                       // 9 sh_org
    take(pl.ns > 1 ? 2 * cap * 16 * (size_t)L * pl.ns : 0);       // 10 sh_dst
    take(2 * G * 4);                                              // 11 sh_cnt
    take(2 * cap * wf_rec_stride(L, pl.ns));                      // 12 rec
    take(multi ? (G + 1) * 2 * 4 : 0);                            // 13 prefix
    take(multi ? ns * 8 * 3 : 0);                                 // 14 acc
    take(multi ? 0 : (size_t)pl.chunk_slots * 8);                 // 15 hit_list
    take(sizeof(WfCounters));                                     // 16
    take(multi ? (size_t)pl.chunk_slots : 0);                     // 17 centre flags
    take(ns * 4);                                                 // 18 s_minT
    take(ns * 4);                                                 // 19 s_tri
    return need;
}

// Buffers of pipeline `pipe` inside the scene's workspace.
static void wf_carve(unsigned char* m, const WfPlan& pl, int L, WfBuffers* B, unsigned char** centre_flags,
                     const size_t* off)
{
    for (int q = 0; q < 2; ++q) {
        B->rays[q].o = (WfVec3*)(m + off[4 * q]);
        B->rays[q].d = (WfVec3*)(m + off[4 * q + 1]);
        B->rays[q].aux = (WfRayAux*)(m + off[4 * q + 2]);
        B->rays[q].cnt = (int*)(m + off[4 * q + 3]);
    }
    B->hits = (float4*)(m + off[8]);
    B->sh_org = (float4*)(m + off[9]);
    B->sh_dst = (float4*)(m + off[10]);
    B->sh_cnt = (int*)(m + off[11]);
    B->rec = m + off[12];
    B->rec_stride = (int)wf_rec_stride(L, pl.ns);
    B->prefix = (int*)(m + off[13]);
    B->acc = (long long*)(m + off[14]);
    B->s_minT = (float*)(m + off[18]);
    B->s_tri = (int*)(m + off[19]);
    B->hit_list = (int2*)(m + off[15]);
    B->ctr = (WfCounters*)(m + off[16]);
    *centre_flags = m + off[17];
    B->ns = pl.ns;
    B->cap = pl.G * pl.R;
    B->G = pl.G, B->R = pl.R, B->S = pl.S;
    B->n_samples = pl.n_samples;
}

static void wf_launch_trace(RtScene* s, const WfBuffers& C, int q, int msaa, int* head, hipStream_t st)
{
    const int te = timing_begin(s, K_TRACE_CLOSEST, st);
    hipLaunchKernelGGL(k_wf_trace<false>, dim3(s->wf_grid_trace), dim3(WF_BLOCK), s->wf_lds, st, s->d, C,
                       (const DParams*)s->d_params, q, msaa, head, s->d_ctr);
    timing_end(s, te, st);
}

// One frame part (the pixel slots [0, n_slots) of this rank).  The product path is one persistent
// launch over one set of buffers per resident block, whatever the size of the frame.  With
// RT_WF_MULTILAUNCH=1 the slots are cut into chunks that fit the workspace (a set of buffers per
// region) and every stage is a launch of its own; RT_WF_PIPES > 1 then puts that many chunks in
// flight on streams of their own (measured slower than one chunk after the other, so the
// default is 1).
static int render_wavefront(RtScene* s, const DParams& P, long long n_slots, float* d_rgb, float* d_depth,
                            int32_t* d_tri, hipStream_t user, bool count)
{
    int rc = wf_prepare(s);
    if (rc) return rc;
    const int L = std::max(P.n_lights, 1);
    // pipelines: enough chunks to overlap tails, but no chunk below ~64k pixel slots
    int n_pipes = s->cfg.pipes;
    n_pipes = std::max(1, std::min(n_pipes, WF_MAX_PIPES));
    n_pipes = (int)std::max<long long>(1, std::min<long long>(n_pipes, n_slots / 65536));
    // pixel slots per region: small regions balance the blocks of the single-launch pipeline
    // (a region is a block's unit of work), the multi-launch path is indifferent
    const bool single = !s->cfg.multilaunch;
    if (!single && P.soft_n > 0)
        return fail(RT_ERR_UNSUPPORTED, "soft shadows are not available with RT_WF_MULTILAUNCH=1");
    int        S = 512;
    if (single) { // measured on a 4K MSAA4 frame: 128 slots for the whole frame (9.7 ms; 192: 10.9, 64: 11.7),
                  // 64 for an eighth of it (2.0 ms; 128: 2.8)
        const long long per_block = n_slots / (16LL * std::max(s->wf_grid_pipe, 1));
        S = per_block >= 512 ? 256 : (per_block >= 100 ? 128 : 64);
        // without MSAA a region of 64 slots is one wave of camera rays for a block of four
        // (1080p: redballs 3.03 -> 2.59 ms, refractionballs 3.59 -> 2.77 ms with 128)
        // - and 192 slots, three waves of them, serve frames of half a million pixels and more better still
        // (1080p: redballs 1.34 / 1.24 / 1.21 ms with 128 / 192 / 256, refractionballs 1.57 / 1.43 / 1.35 ms, but
        // fin.obj, whose longest regions end the launch, 1.33 / 1.36 / 1.52 ms)
        if (P.msaa == 1) S = std::max(S, n_slots >= 500000 ? 192 : 128);
    }
    if (s->cfg.region > 0) S = s->cfg.region;
    S = std::max(64, std::min(4096, (S / 64) * 64));
    if (single) {
        n_pipes = 1;
        S = std::max(32, std::min(S, WF_ACC_SAMPLES / P.msaa / 32 * 32)); // the region's colour sums live in LDS
    }
    // blocks the single launch can keep resident with its stacks and colour sums in LDS
    // Beside them, where LDS is left: a copy of the tree's first levels (wf_node_pair).  A walk is a chain of
    // dependent loads, and a record from LDS arrives in half the time of one from the vector L1; the nodes are
    // numbered breadth first, so the first 2 KB are the top eight levels every walk starts with.  What fits
    // without costing a resident block is taken, up to 8 KB; if that is less than 1 KB, one stack entry per
    // thread gives way (final.obj 4K MSAA4: 5 entries 6.19 ms, 4 entries 6.27 ms, 4 entries + 2 KB of nodes 6.00 ms).
    // The records' pending counts, one byte per ray slot and level parity, in LDS where a region's queue is short
    // enough (no refraction: a ray slot per sample) and the lights are at most four: an LDS atomic per answer
    // instead of a transaction with the memory fabric.  They are part of what the block's other LDS contents leave.
    const long long R_plan = (long long)S * P.msaa * (s->refractive ? 16 : 1);
    const long long pend_max = getenv("RT_WF_PEND_MAX") ? atoi(getenv("RT_WF_PEND_MAX")) : 1024;
    const bool      pend_lds = single && P.soft_n == 0 && L <= 4 && R_plan <= pend_max && !getenv("RT_WF_NO_LDS_PENDING");
    const size_t    pend_bytes = pend_lds ? (size_t)((R_plan + 3) & ~3LL) : 0;
    const size_t sums_bytes = (size_t)S * P.msaa * 24 + 2 * pend_bytes; // (kept as one figure: what has to fit beside the stacks)
    int          grid_single = s->wf_grid_pipe;
    if (single) {
        if (s->wf_single_sums != sums_bytes) {
            // (the runtime's answer, capped by LDS handed out in units of 512 bytes, the kernel's own static LDS
            // included: at 23296 dynamic bytes the runtime still says seven blocks and the launch runs like six)
            auto blocks_with = [&](size_t lds, int* per_cu) -> int {
                HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, k_wf_pipeline<false, false>, WF_BLOCK, lds));
                const size_t unit = (lds + sizeof(WfShared) + 511) & ~(size_t)511;
                if (s->wf_lds_per_cu > 0) *per_cu = std::min<int>(*per_cu, (int)(s->wf_lds_per_cu / unit));
                return RT_OK;
            };
            const char*  top_env = getenv("RT_WF_TOP"); // bytes; 0 switches the copy off
            const size_t top_max = std::min<size_t>((size_t)s->n_knodes * 8, top_env ? (size_t)atoi(top_env) & ~(size_t)7 : 8192);
            const int    cap_hi = s->d.stk_cap, cap_lo = s->wf_cap_lo;
            // as many stack entries as keep every block resident that the fewest entries would ...
            int per_cu_max = 0, rc2;
            if ((rc2 = blocks_with((size_t)cap_lo * WF_BLOCK * WF_STACK_ENTRY_BYTES + sums_bytes, &per_cu_max))) return rc2;
            int cap_fit = cap_lo;
            for (int c = cap_hi; c > cap_lo; --c) {
                int per_cu = 0;
                if ((rc2 = blocks_with((size_t)c * WF_BLOCK * WF_STACK_ENTRY_BYTES + sums_bytes, &per_cu))) return rc2;
                if (per_cu >= per_cu_max) {
                    cap_fit = c;
                    break;
                }
            }
            // ... and what is left for the nodes, with those entries and with one less
            size_t top_of[2] = {0, 0};
            for (int v = 0; v < 2; ++v) {
                const int c = cap_fit - v;
                if (c < cap_lo) break;
                const size_t stacks = (size_t)c * WF_BLOCK * WF_STACK_ENTRY_BYTES;
                for (size_t top = top_max; top > 0; top = top > 256 ? ((top - 1) & ~(size_t)255) : 0) {
                    int per_cu = 0;
                    if ((rc2 = blocks_with(stacks + sums_bytes + top, &per_cu))) return rc2;
                    if (per_cu >= per_cu_max) {
                        top_of[v] = top;
                        break;
                    }
                }
            }
            // (final.obj 4K MSAA4 with the pending counts in LDS: 4 entries + 1 KB 5.67 ms, 3 entries + 2 / 3 KB 5.69 / 5.71 ms)
            const size_t want = std::min<size_t>(top_max, 1024);
            const int    v = (top_of[0] >= want || cap_fit - 1 < cap_lo || top_of[1] < want) ? 0 : 1;
            const int    per_cu_of[2] = {per_cu_max, per_cu_max};
            s->wf_single_cap = cap_fit - v;
            s->wf_single_top = top_of[v];
            s->wf_single_lds = (size_t)s->wf_single_cap * WF_BLOCK * WF_STACK_ENTRY_BYTES + sums_bytes + s->wf_single_top;
            s->wf_single_grid = std::min(s->wf_grid_pipe, std::max(1, per_cu_of[v]) * s->wf_cus);
            s->wf_single_sums = sums_bytes;
            if (s->cfg.kd_stats)
                fprintf(stderr, "wavefront: single launch with %d stack entries, %zu B of nodes and %zu B of pending counts in LDS, %zu B per block, %d blocks per CU\n",
                        s->wf_single_cap, s->wf_single_top, 2 * pend_bytes, s->wf_single_lds, per_cu_of[v]);
        }
        grid_single = s->wf_single_grid;
    }
    const size_t lds_single = s->wf_single_lds;
    // the single launch keeps one set of buffers per resident block, whatever the size of the frame
    const long long plan_slots = single ? std::min(n_slots, (long long)grid_single * S) : (n_slots + n_pipes - 1) / n_pipes;
    WfPlan pl = wf_plan(s, P, plan_slots, n_pipes, S, !single);
    size_t off[WF_N_BUFFERS];
    const size_t per_pipe = wf_chunk_bytes(pl, L, !single, off);
    const size_t need = per_pipe * n_pipes;
    if (need > s->wf_mem_bytes) {
        if (s->wf_mem) (void)hipFree(s->wf_mem);
        s->wf_mem = nullptr, s->wf_mem_bytes = 0;
        HIP_TRY(hipMalloc(&s->wf_mem, need));
        s->wf_mem_bytes = need;
    }
    if (!s->d_params) HIP_TRY(hipMalloc(&s->d_params, sizeof(DParams)));
    for (int k = 0; k < n_pipes; ++k)
        if (!s->wf_stream[k]) {
            HIP_TRY(hipStreamCreateWithFlags(&s->wf_stream[k], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&s->wf_join[k], hipEventDisableTiming));
        }
    if (!s->wf_fork) HIP_TRY(hipEventCreateWithFlags(&s->wf_fork, hipEventDisableTiming));
    HIP_TRY(hipMemcpyAsync(s->d_params, &P, sizeof(DParams), hipMemcpyHostToDevice, user));
    const int te = timing_begin(s, K_FRAME, user);
    HIP_TRY(hipEventRecord(s->wf_fork, user));
    WfBuffers      C[WF_MAX_PIPES];
    unsigned char* flags[WF_MAX_PIPES];
    for (int k = 0; k < n_pipes; ++k) {
        HIP_TRY(hipStreamWaitEvent(s->wf_stream[k], s->wf_fork, 0));
        wf_carve((unsigned char*)s->wf_mem + per_pipe * k, pl, L, &C[k], &flags[k], off);
    }
    if (single) {
        hipStream_t st = s->wf_stream[0];
        HIP_TRY(hipMemsetAsync(C[0].ctr, 0, sizeof(WfCounters), st));
        auto kern = P.soft_n > 0 ? (count ? k_wf_pipeline<true, true> : k_wf_pipeline<false, true>)
                                 : (count ? k_wf_pipeline<true, false> : k_wf_pipeline<false, false>);
        WfLaunch A;
        if (P.msaa > 1 || !d_tri) C[0].s_tri = nullptr; // nobody reads the samples' triangles then
        A.S = s->d, A.P = P, A.B = C[0], A.slot0 = 0, A.n = (int)n_slots, A.n_regions = (int)((n_slots + S - 1) / S);
        A.S.stk_cap = s->wf_single_cap;
        A.rgb = d_rgb, A.depth = d_depth, A.tri = d_tri, A.ctr = s->d_ctr;
        // Regions in the order of the durations the last launch of this shape measured, longest first
        // (k_wf_order_*); by index the first time, after a change of shape, and for launches of a few regions.
        A.order = nullptr, A.cost = nullptr;
        A.top_bytes = (unsigned)s->wf_single_top;
        A.pend_bytes = (unsigned)pend_bytes;
        const bool feedback = s->wf_feedback && A.n_regions > pl.G;
        if (feedback) {
            if ((size_t)A.n_regions > s->wf_order_cap) {
                if (s->wf_cost) (void)hipFree(s->wf_cost);
                if (s->wf_order) (void)hipFree(s->wf_order);
                s->wf_cost = nullptr, s->wf_order = nullptr, s->wf_order_cap = 0, s->wf_order_key = 0;
                HIP_TRY(hipMalloc((void**)&s->wf_cost, sizeof(unsigned) * (size_t)A.n_regions));
                HIP_TRY(hipMalloc((void**)&s->wf_order, sizeof(int) * (size_t)A.n_regions));
                if (!s->wf_order_at) HIP_TRY(hipMalloc((void**)&s->wf_order_at, sizeof(int) * WF_ORDER_RUNS * WF_ORDER_BINS));
                s->wf_order_cap = (size_t)A.n_regions;
            }
            // what the durations belong to: the frame, its sampling, this rank's part of it, the region size
            uint64_t key = 1469598103934665603ull;
            for (uint64_t v : {(uint64_t)P.W, (uint64_t)P.H, (uint64_t)P.msaa, (uint64_t)P.tile_w, (uint64_t)P.tile_h,
                               (uint64_t)P.rank, (uint64_t)P.world, (uint64_t)P.span, (uint64_t)n_slots, (uint64_t)S,
                               (uint64_t)P.soft_n})
                key = (key ^ v) * 1099511628211ull;
            if (key == 0) key = 1;
            A.cost = s->wf_cost;
            if (s->wf_order_key == key) A.order = s->wf_order;
            s->wf_order_key = key;
        }
        hipLaunchKernelGGL(kern, dim3(pl.G), dim3(WF_BLOCK), lds_single, st, A); // pl.G <= resident blocks
        HIP_TRY(hipGetLastError());
        if (feedback) {
            const WfCounters* ctr = C[0].ctr; // (zeroed before the launch; the order kernels run before the next launch's memset)
            hipLaunchKernelGGL(k_wf_order_count, dim3(WF_ORDER_BLOCKS), dim3(1024), 0, st, (const unsigned*)s->wf_cost,
                               A.n_regions, s->wf_hot, (int)pl.G, ctr, s->wf_order_at);
            hipLaunchKernelGGL(k_wf_order_offsets, dim3(1), dim3(1024), 0, st, s->wf_order_at);
            hipLaunchKernelGGL(k_wf_order_place, dim3(WF_ORDER_BLOCKS), dim3(1024), 0, st, (const unsigned*)s->wf_cost,
                               A.n_regions, s->wf_hot, (int)pl.G, ctr, (const int*)s->wf_order_at, s->wf_order);
            HIP_TRY(hipGetLastError());
        }
    }
    for (long long round0 = 0; !single && round0 < n_slots; round0 += pl.chunk_slots * n_pipes) {
        long long slot0[WF_MAX_PIPES];
        int       n[WF_MAX_PIPES];
        int       live = 0;
        for (int k = 0; k < n_pipes; ++k) {
            slot0[k] = round0 + pl.chunk_slots * k;
            n[k] = (int)std::max<long long>(0, std::min<long long>(pl.chunk_slots, n_slots - slot0[k]));
            if (n[k] > 0) live = k + 1;
        }
        // level l lives in ray queue l & 1; with MSAA the centre rays pass through queue 1 first
        const int q0 = P.msaa > 1 ? 1 : 0;
        for (int k = 0; k < live; ++k) {
            hipStream_t st = s->wf_stream[k];
            HIP_TRY(hipMemsetAsync(C[k].ctr, 0, sizeof(WfCounters), st));
            hipLaunchKernelGGL(k_wf_gen_centre, dim3(C[k].G), dim3(WF_BLOCK), 0, st, P, C[k], q0, slot0[k], n[k],
                               s->d_ctr);
            hipLaunchKernelGGL(k_wf_worklist, dim3(1), dim3(1024), 0, st, C[k], q0, 0);
        }
        for (int k = 0; k < live; ++k)
            wf_launch_trace(s, C[k], q0, P.msaa, &C[k].ctr->head_rays[WF_MAX_DEPTH], s->wf_stream[k]);
        HIP_TRY(hipGetLastError());
        bool traced0 = true;
        if (P.msaa > 1) {
            for (int k = 0; k < live; ++k) {
                hipStream_t st = s->wf_stream[k];
                hipLaunchKernelGGL(k_wf_msaa_emit, dim3(C[k].G), dim3(WF_BLOCK), 0, st, P, C[k], slot0[k], d_rgb,
                                   d_depth, d_tri, flags[k], s->d_ctr);
                hipLaunchKernelGGL(k_wf_worklist, dim3(1), dim3(1024), 0, st, C[k], 0, 0);
            }
            traced0 = false;
        }
        for (int level = 0; level < 5; ++level) {
            const int q = level & 1;
            for (int k = 0; k < live; ++k) {
                hipStream_t st = s->wf_stream[k];
                if (level > 0 || !traced0) wf_launch_trace(s, C[k], q, P.msaa, &C[k].ctr->head_rays[level], st);
                hipLaunchKernelGGL(k_wf_shade, dim3(C[k].G), dim3(WF_BLOCK), 0, st, s->d, P, C[k], level, slot0[k],
                                   s->d_ctr);
                // work lists of the next level's rays (block 0) and of this level's shadow queue
                hipLaunchKernelGGL(k_wf_worklist, dim3(P.n_lights > 0 ? 2 : 1), dim3(1024), 0, st, C[k], q ^ 1, q);
            }
            for (int k = 0; k < live; ++k) {
                hipStream_t st = s->wf_stream[k];
                if (P.n_lights > 0) {
                    const int ta = timing_begin(s, K_TRACE_ANY, st);
                    hipLaunchKernelGGL(k_wf_trace<true>, dim3(s->wf_grid_any), dim3(WF_BLOCK), s->wf_lds, st, s->d, C[k],
                                       (const DParams*)s->d_params, q, P.msaa, &C[k].ctr->head_shadow[level][0],
                                       s->d_ctr);
                    timing_end(s, ta, st);
                }
            }
            HIP_TRY(hipGetLastError());
        }
        for (int k = 0; k < live; ++k)
            hipLaunchKernelGGL(k_wf_finalize, dim3(blocks_for(n[k], WF_BLOCK)), dim3(WF_BLOCK), 0,
                               s->wf_stream[k], P, C[k], slot0[k], n[k], (const unsigned char*)flags[k], d_rgb,
                               d_depth, d_tri);
        HIP_TRY(hipGetLastError());
    }
    for (int k = 0; k < n_pipes; ++k) {
        HIP_TRY(hipEventRecord(s->wf_join[k], s->wf_stream[k]));
        HIP_TRY(hipStreamWaitEvent(user, s->wf_join[k], 0));
    }
    timing_end(s, te, user);
    return RT_OK;
}

extern "C" int rt_render_frame_device(RtScene* s, const float camera[24], const float* lights,
                                      int32_t n_lights, const RtParams* params, float* d_rgb,
                                      float* d_depth, int32_t* d_tri, void* stream_, RtStats* stats)
{
    if (!s || !camera || !d_rgb || !d_depth)
        return fail(RT_ERR_INVALID, "rt_render_frame: null argument");
    DParams P;
    int     rc = make_dparams(params, camera, lights, n_lights, &P);
    if (rc) return rc;
    if ((rc = set_partition(params, params->rank, &P))) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(hipSetDevice(s->device));
    const long long n_slots = (long long)P.n_my_tiles * P.tile_w * P.tile_h;
    if (stats) memset(stats, 0, sizeof *stats);
    HIP_TRY(hipMemsetAsync(s->d_ctr, 0, sizeof(DCounters), stream));
    if (stats) HIP_TRY(hipEventRecord(s->ev0, stream));
    const bool wavefront = s->pushdown && !s->cfg.megakernel;
    if (n_slots > 0 && wavefront) {
        if ((rc = render_wavefront(s, P, n_slots, d_rgb, d_depth, d_tri, stream, stats != nullptr))) return rc;
    } else if (n_slots > 0) {
        if ((rc = ensure_workspace(s, n_slots, P.msaa))) return rc;
        const size_t lds = stack_bytes(s);
        if (P.msaa == 1) {
            const int tc = timing_begin(s, K_FRAME, stream);
            hipLaunchKernelGGL(k_centre<true>, dim3(blocks_for(n_slots, RT_BLOCK)), dim3(RT_BLOCK), lds,
                               stream, s->d, P, n_slots, d_rgb, d_depth, d_tri, (unsigned char*)nullptr,
                               s->d_ctr);
            timing_end(s, tc, stream);
            HIP_TRY(hipGetLastError());
        } else {
            int te = timing_begin(s, K_FRAME, stream);
            hipLaunchKernelGGL(k_centre<false>, dim3(blocks_for(n_slots, RT_BLOCK)), dim3(RT_BLOCK), lds,
                               stream, s->d, P, n_slots, d_rgb, d_depth, d_tri, s->d_flags, s->d_ctr);
            timing_end(s, te, stream);
            HIP_TRY(hipGetLastError());
            te = timing_begin(s, K_FRAME, stream);
            hipLaunchKernelGGL(k_samples, dim3(blocks_for(n_slots * P.msaa, RT_BLOCK)), dim3(RT_BLOCK),
                               lds, stream, s->d, P, n_slots, (const unsigned char*)s->d_flags,
                               s->d_samples, s->d_ctr);
            timing_end(s, te, stream);
            HIP_TRY(hipGetLastError());
            te = timing_begin(s, K_FRAME, stream);
            hipLaunchKernelGGL(k_resolve, dim3(blocks_for(n_slots, 256)), dim3(256), 0, stream, P,
                               n_slots, (const unsigned char*)s->d_flags,
                               (const float4*)s->d_samples, d_rgb, d_depth);
            timing_end(s, te, stream);
            HIP_TRY(hipGetLastError());
        }
    }
    if (stats) {
        HIP_TRY(hipEventRecord(s->ev1, stream));
        read_stats(s, stats, stream);
        HIP_TRY(hipGetLastError());
    }
    return RT_OK;
}

extern "C" int rt_render_frame(RtScene* s, const float camera[24], const float* lights,
                               int32_t n_lights, const RtParams* params, float* rgb, float* depth,
                               int32_t* tri, RtStats* stats)
{
    if (!s || !params || !rgb || !depth) return fail(RT_ERR_INVALID, "rt_render_frame: null argument");
    if (params->width < 2 || params->height < 2 || params->width > 32768 || params->height > 32768)
        return fail(RT_ERR_INVALID, "frame size %dx%d out of range (2..32768)", params->width,
                    params->height);
    HIP_TRY(hipSetDevice(s->device));
    const size_t npx = (size_t)params->width * params->height;
    DevBuf       drgb, ddepth, dtri;
    int          rc;
    if ((rc = drgb.alloc(12 * npx)) || (rc = ddepth.alloc(4 * npx)) || (rc = dtri.alloc(4 * npx)))
        return rc;
    HIP_TRY(hipMemset(drgb.p, 0, 12 * npx));
    HIP_TRY(hipMemset(ddepth.p, 0, 4 * npx));
    HIP_TRY(hipMemset(dtri.p, 0xff, 4 * npx));
    RtStats local;
    rc = rt_render_frame_device(s, camera, lights, n_lights, params, (float*)drgb.p, (float*)ddepth.p,
                                (int32_t*)dtri.p, nullptr, stats ? stats : &local);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(rgb, drgb.p, 12 * npx, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(depth, ddepth.p, 4 * npx, hipMemcpyDeviceToHost));
    if (tri) HIP_TRY(hipMemcpy(tri, dtri.p, 4 * npx, hipMemcpyDeviceToHost));
    return RT_OK;
}

#ifdef WF_TIMELINE
// diagnostic builds only: copy out (and clear) the region time stamps of k_wf_pipeline
extern "C" int rt_debug_timeline(unsigned long long* out, int64_t n_regions, int clear)
{
    if (n_regions > WF_TL_REGIONS) n_regions = WF_TL_REGIONS;
    if (out) HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), (size_t)n_regions * 16 * 8));
    if (clear) {
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(g_timeline)));
        HIP_TRY(hipMemset(p, 0, sizeof(unsigned long long) * WF_TL_REGIONS * 16));
    }
    return RT_OK;
}
#endif

// ---------------------------------------------------------------------------
// post-process
// ---------------------------------------------------------------------------
#define RANGE_BLOCKS 1024

static int check_device(int device)
{
    const int ndev = rt_device_count();
    if (ndev <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device");
    if (device < 0 || device >= ndev) return fail(RT_ERR_INVALID, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    return RT_OK;
}

// ---------------------------------------------------------------------------
// the reference's KD tree, built on the device (rt_kdbuild.h)
// ---------------------------------------------------------------------------
extern "C" int rt_build_kd_tree(int device, int32_t n_vertices, const float* vertex_pos, int32_t n_triangles,
                                const int32_t* tri_vertex, int32_t max_nodes, RtKdNode* nodes_out, int32_t* n_nodes_out,
                                int32_t* node_tris_out, int32_t* depth_out)
{
    int rc = check_device(device);
    if (rc) return rc;
    if (!vertex_pos || !tri_vertex || !nodes_out || !n_nodes_out || !node_tris_out || n_vertices < 1 || n_triangles < 1 ||
        max_nodes < 1)
        return fail(RT_ERR_INVALID, "rt_build_kd_tree: null or empty argument");
    const int nt = n_triangles;
    for (int i = 0; i < 3 * nt; ++i)
        if (tri_vertex[i] < 0 || tri_vertex[i] >= n_vertices)
            return fail(RT_ERR_INVALID, "rt_build_kd_tree: triangle %d names vertex %d of %d", i / 3, tri_vertex[i], n_vertices);
    // root box, KDtree.cpp:15-60: the bounds of the vertices the triangles use, grown by 5.1
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) lo[a] = hi[a] = vertex_pos[3 * tri_vertex[0] + a];
    for (int i = 0; i < 3 * nt; ++i)
        for (int a = 0; a < 3; ++a) {
            const float c = vertex_pos[3 * tri_vertex[i] + a];
            if (c < lo[a]) lo[a] = c;
            else if (c > hi[a]) hi[a] = c;
        }
    const int cap_nodes = 4 * nt + 64, cap_items = 2 * (nt / 11) + 16;
    DevBuf    d_pos, d_tv, d_list[2], d_kept, d_nodes, d_items[2], d_ctl;
    if ((rc = d_pos.alloc(sizeof(float) * 3 * (size_t)n_vertices)) || (rc = d_tv.alloc(sizeof(int32_t) * 3 * (size_t)nt)) ||
        (rc = d_list[0].alloc(sizeof(int32_t) * (size_t)nt)) || (rc = d_list[1].alloc(sizeof(int32_t) * (size_t)nt)) ||
        (rc = d_kept.alloc(sizeof(int32_t) * (size_t)nt)) || (rc = d_nodes.alloc(sizeof(KdbNode) * (size_t)cap_nodes)) ||
        (rc = d_items[0].alloc(sizeof(KdbItem) * (size_t)cap_items)) || (rc = d_items[1].alloc(sizeof(KdbItem) * (size_t)cap_items)) ||
        (rc = d_ctl.alloc(sizeof(KdbCtl))))
        return rc;
    HIP_TRY(hipMemcpy(d_pos.p, vertex_pos, sizeof(float) * 3 * (size_t)n_vertices, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_tv.p, tri_vertex, sizeof(int32_t) * 3 * (size_t)nt, hipMemcpyHostToDevice));
    {
        std::vector<int32_t> all((size_t)nt);
        for (int t = 0; t < nt; ++t) all[(size_t)t] = t;
        HIP_TRY(hipMemcpy(d_list[0].p, all.data(), sizeof(int32_t) * (size_t)nt, hipMemcpyHostToDevice));
        KdbNode root{};
        for (int a = 0; a < 3; ++a) root.b[2 * a] = lo[a] - 5.1f, root.b[2 * a + 1] = hi[a] + 5.1f;
        root.left = root.right = -1;
        HIP_TRY(hipMemcpy(d_nodes.p, &root, sizeof root, hipMemcpyHostToDevice));
        const KdbItem first{0, 0, 0, nt};
        HIP_TRY(hipMemcpy(d_items[0].p, &first, sizeof first, hipMemcpyHostToDevice));
        KdbCtl ctl{};
        ctl.n_nodes = 1;
        HIP_TRY(hipMemcpy(d_ctl.p, &ctl, sizeof ctl, hipMemcpyHostToDevice));
    }
    int n_items = 1, kept_used = 0, n_nodes = 1;
    for (int level = 0; n_items > 0; ++level) {
        if (level > 63) return fail(RT_ERR_UNSUPPORTED, "rt_build_kd_tree: the tree is deeper than 63 levels");
        const int cur = level & 1;
        hipLaunchKernelGGL(k_kd_build_level, dim3((unsigned)n_items), dim3(KDB_BLOCK), 0, 0, (const float*)d_pos.p,
                           (const int32_t*)d_tv.p, (const KdbItem*)d_items[cur].p, (const int32_t*)d_list[cur].p,
                           (int32_t*)d_list[cur ^ 1].p, (int32_t*)d_kept.p, (KdbNode*)d_nodes.p, (KdbItem*)d_items[cur ^ 1].p,
                           (KdbCtl*)d_ctl.p, cap_nodes, cap_items, nt);
        HIP_TRY(hipGetLastError());
        KdbCtl ctl;
        HIP_TRY(hipMemcpy(&ctl, d_ctl.p, sizeof ctl, hipMemcpyDeviceToHost)); // (waits for the level)
        if (ctl.overflow) return fail(RT_ERR_UNSUPPORTED, "rt_build_kd_tree: more than %d nodes", cap_nodes);
        n_items = ctl.n_next, n_nodes = ctl.n_nodes, kept_used = ctl.kept_used;
        ctl.n_next = 0, ctl.list_used = 0;
        HIP_TRY(hipMemcpy(d_ctl.p, &ctl, sizeof ctl, hipMemcpyHostToDevice));
    }
    if (kept_used != nt) return fail(RT_ERR_HIP, "rt_build_kd_tree: %d of %d triangles placed", kept_used, nt);
    if (n_nodes > max_nodes) return fail(RT_ERR_INVALID, "rt_build_kd_tree: %d nodes, room for %d", n_nodes, max_nodes);
    std::vector<KdbNode> bn((size_t)n_nodes);
    std::vector<int32_t> kept((size_t)nt);
    HIP_TRY(hipMemcpy(bn.data(), d_nodes.p, sizeof(KdbNode) * (size_t)n_nodes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(kept.data(), d_kept.p, sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost));
    // The reference's numbering: a node's children get the next two numbers when the node is expanded, and the
    // left subtree is expanded before the right one (KDtree.cpp:75-130).
    std::vector<int32_t> final_id((size_t)n_nodes, -1), order;
    std::vector<int32_t> stack{0};
    int32_t              next_id = 1, depth = 0;
    final_id[0] = 0;
    while (!stack.empty()) {
        const int32_t b = stack.back();
        stack.pop_back();
        depth = std::max(depth, bn[(size_t)b].depth);
        if (bn[(size_t)b].left >= 0) {
            final_id[(size_t)bn[(size_t)b].left] = next_id, final_id[(size_t)bn[(size_t)b].right] = next_id + 1;
            next_id += 2;
            stack.push_back(bn[(size_t)b].right);
            stack.push_back(bn[(size_t)b].left);
        }
    }
    if (next_id != n_nodes) return fail(RT_ERR_HIP, "rt_build_kd_tree: %d of %d nodes reachable", next_id, n_nodes);
    std::vector<int32_t> by_final((size_t)n_nodes);
    for (int32_t b = 0; b < n_nodes; ++b) by_final[(size_t)final_id[(size_t)b]] = b;
    int32_t off = 0;
    for (int32_t f = 0; f < n_nodes; ++f) {
        const KdbNode& k = bn[(size_t)by_final[(size_t)f]];
        RtKdNode&      o = nodes_out[f];
        for (int a = 0; a < 6; ++a) o.bounds[a] = k.b[a];
        o.left = k.left >= 0 ? final_id[(size_t)k.left] : -1;
        o.right = k.right >= 0 ? final_id[(size_t)k.right] : -1;
        o.tri_offset = off, o.tri_count = k.kept_cnt;
        for (int32_t i = 0; i < k.kept_cnt; ++i) node_tris_out[off + i] = kept[(size_t)(k.kept_off + i)];
        off += k.kept_cnt;
    }
    *n_nodes_out = n_nodes;
    if (depth_out) *depth_out = depth;
    return RT_OK;
}

// Scratch layout of the depth passes: [0, RANGE_BLOCKS) partial minima, [RANGE_BLOCKS, 2 RANGE_BLOCKS)
// partial maxima, then the two denominators.
#define DEPTH_SCRATCH_FLOATS (2 * RANGE_BLOCKS + 64)

static int depth_blocks(int64_t n) { return (int)std::min<long long>(RANGE_BLOCKS, (n + 255) / 256); }

// One normalisation.  again: also leave the denominator of a second normalisation of the
// result in d_part[2 * RANGE_BLOCKS + 1].
static int normalize_depth_impl(int64_t n, float focus, float* d_depth, float* d_part, hipStream_t st,
                                bool again = false)
{
    const int nb = depth_blocks(n);
    float*    lo = d_part, *hi = d_part + RANGE_BLOCKS, *denom = d_part + 2 * RANGE_BLOCKS;
    hipLaunchKernelGGL(k_depth_range_partial, dim3(nb), dim3(256), 0, st, (const float*)d_depth, (long long)n, lo, hi);
    hipLaunchKernelGGL(k_depth_denominator, dim3(1), dim3(256), 0, st, (const float*)lo, (const float*)hi, nb, focus,
                       denom);
    hipLaunchKernelGGL(k_depth_normalize, dim3(nb), dim3(256), 0, st, d_depth, (long long)n, (const float*)denom,
                       focus, again ? lo : (float*)nullptr, again ? hi : (float*)nullptr);
    if (again)
        hipLaunchKernelGGL(k_depth_denominator, dim3(1), dim3(256), 0, st, (const float*)lo, (const float*)hi, nb,
                           focus, denom + 1);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

// Scratch of the depth passes for the allocating entry points: one block of DEPTH_SCRATCH_FLOATS per
// (device, stream), allocated at the first call on that stream and kept until the process ends.
// Calls on one stream are ordered, so they may share a block; calls on different streams get
// different blocks and may overlap.  (A hipMallocAsync / hipFreeAsync pair per call used to sit
// inside every frame's post-process.)  Beyond DEPTH_SCRATCH_STREAMS streams per device a call takes
// its block from the stream's memory pool as before.
#define DEPTH_SCRATCH_STREAMS 64
namespace {
struct StreamScratch {
    int         device;
    hipStream_t stream;
    float*      p;
};
std::vector<StreamScratch> g_stream_scratch;
std::mutex                 g_stream_scratch_mutex;
} // namespace

static float* stream_scratch(int device, hipStream_t st) // `device` is current
{
    std::lock_guard<std::mutex> lock(g_stream_scratch_mutex);
    size_t                      on_device = 0;
    for (const StreamScratch& e : g_stream_scratch) {
        if (e.device != device) continue;
        if (e.stream == st) return e.p;
        ++on_device;
    }
    if (on_device >= DEPTH_SCRATCH_STREAMS) return nullptr;
    float* p = nullptr;
    if (hipMalloc((void**)&p, DEPTH_SCRATCH_FLOATS * sizeof(float)) != hipSuccess) return nullptr;
    g_stream_scratch.push_back({device, st, p});
    return p;
}

// Runs fn(scratch) with the stream's own block, or with one from the stream's memory pool.
template <class F>
static int with_depth_scratch(int device, hipStream_t st, F fn)
{
    if (float* p = stream_scratch(device, st)) return fn(p);
    float* d_part = nullptr;
    HIP_TRY(hipMallocAsync((void**)&d_part, DEPTH_SCRATCH_FLOATS * sizeof(float), st));
    const int rc = fn(d_part);
    (void)hipFreeAsync(d_part, st);
    return rc;
}

extern "C" int rt_normalize_depth_device(int device, int64_t n, float focus, float* d_depth,
                                         void* stream)
{
    int rc = check_device(device);
    if (rc) return rc;
    if (n < 0) return fail(RT_ERR_INVALID, "rt_normalize_depth: negative n");
    if (n == 0) return RT_OK;
    if (!d_depth) return fail(RT_ERR_INVALID, "rt_normalize_depth: null depth");
    return with_depth_scratch(device, (hipStream_t)stream, [&](float* d_part) {
        return normalize_depth_impl(n, focus, d_depth, d_part, (hipStream_t)stream);
    });
}

// ---- the entry point that allocates nothing ----
// Partial minima / maxima and the denominator of rt_normalize_depth_device_noalloc: the same
// DEPTH_SCRATCH_FLOATS floats rt_normalize_depth_device takes from the stream's memory pool, here
// in a file-scope device array.  Every GPU the library runs on holds its own copy (8.25 KB) until
// the process ends.  The
// array's name is a host-side stand-in, not a device address: host code uses the per-device
// pointer hipGetSymbolAddress returns, looked up once per device with that device current.
static __device__ __align__(256) float scratch[DEPTH_SCRATCH_FLOATS];
#define RT_NOALLOC_MAX_DEVICES 64
static float*     g_scratch_ptr[RT_NOALLOC_MAX_DEVICES];
static std::mutex g_scratch_mutex;

static int scratch_for_device(int device, float** out)
{
    if (device >= RT_NOALLOC_MAX_DEVICES)
        return fail(RT_ERR_UNSUPPORTED, "rt_normalize_depth_device_noalloc: device %d beyond the %d supported", device,
                    RT_NOALLOC_MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) { // `device` is current: check_device() set it
        float* p = nullptr;
        HIP_TRY(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(RT_ERR_HIP, "rt_normalize_depth_device_noalloc: no address for the scratch array");
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return RT_OK;
}

extern "C" int rt_normalize_depth_device_noalloc(int device, int64_t n, float focus, float* d_depth,
                                                 void* stream)
{
    int rc = check_device(device);
    if (rc) return rc;
    if (n < 0) return fail(RT_ERR_INVALID, "rt_normalize_depth_device_noalloc: negative n");
    if (n > RT_NOALLOC_MAX_DEPTH_VALUES)
        return fail(RT_ERR_INVALID, "rt_normalize_depth_device_noalloc: n = %lld exceeds the maximum of %lld values",
                    (long long)n, (long long)RT_NOALLOC_MAX_DEPTH_VALUES);
    if (n == 0) return RT_OK;
    if (!d_depth) return fail(RT_ERR_INVALID, "rt_normalize_depth_device_noalloc: null depth");
    float* part = nullptr;
    if ((rc = scratch_for_device(device, &part))) return rc;
    // both kernels write every partial they later read (nb blocks each), nothing has to start at zero
    return normalize_depth_impl(n, focus, d_depth, part, (hipStream_t)stream);
}

extern "C" int rt_normalize_depth(int device, int64_t n, float focus, float* depth)
{
    int rc = check_device(device);
    if (rc) return rc;
    if (n < 0) return fail(RT_ERR_INVALID, "rt_normalize_depth: negative n");
    if (n == 0) return RT_OK;
    if (!depth) return fail(RT_ERR_INVALID, "rt_normalize_depth: null depth");
    DevBuf d, part;
    if ((rc = d.alloc(4 * n)) || (rc = part.alloc(DEPTH_SCRATCH_FLOATS * sizeof(float)))) return rc;
    HIP_TRY(hipMemcpy(d.p, depth, 4 * n, hipMemcpyHostToDevice));
    if ((rc = normalize_depth_impl(n, focus, (float*)d.p, (float*)part.p, 0))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(depth, d.p, 4 * n, hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_dof_blur_device(int device, int32_t W, int32_t H, float depth_power,
                                  const float* d_rgb, const float* d_depth_n, float* d_out,
                                  void* stream)
{
    int rc = check_device(device);
    if (rc) return rc;
    if (W < 13 || H < 13 || W > 32768 || H > 32768)
        return fail(RT_ERR_INVALID, "rt_dof_blur: frame %dx%d must be at least 13x13", W, H);
    if (!d_rgb || !d_depth_n || !d_out) return fail(RT_ERR_INVALID, "rt_dof_blur: null argument");
    const unsigned n_blocks = (unsigned)(((W - 12 + DOF_TW - 1) / DOF_TW) * ((H - 12 + DOF_TH - 1) / DOF_TH));
    hipLaunchKernelGGL(k_dof_blur, dim3(n_blocks), dim3(DOF_BLOCK), 0, (hipStream_t)stream, d_rgb,
                       d_depth_n, (int)W, (int)H, depth_power, d_out, (const float*)nullptr, 0.0f);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

extern "C" int rt_postprocess_device(int device, int32_t W, int32_t H, float focus, float depth_power,
                                     const float* d_rgb, float* d_depth, float* d_blurred, void* stream_)
{
    int rc = check_device(device);
    if (rc) return rc;
    if (W < 2 || H < 2 || W > 32768 || H > 32768)
        return fail(RT_ERR_INVALID, "rt_postprocess: frame size %dx%d out of range (2..32768)", W, H);
    if (!d_depth) return fail(RT_ERR_INVALID, "rt_postprocess: null depth");
    const bool blur = d_blurred != nullptr;
    if (blur && (W < 13 || H < 13)) return fail(RT_ERR_INVALID, "rt_postprocess: frame %dx%d must be at least 13x13 for the blur", W, H);
    if (blur && !d_rgb) return fail(RT_ERR_INVALID, "rt_postprocess: null rgb");
    hipStream_t st = (hipStream_t)stream_;
    return with_depth_scratch(device, st, [&](float* d_part) {
        int rc2 = normalize_depth_impl((int64_t)W * H, focus, d_depth, d_part, st, blur);
        if (rc2 == RT_OK && blur) {
            const unsigned n_blocks = (unsigned)(((W - 12 + DOF_TW - 1) / DOF_TW) * ((H - 12 + DOF_TH - 1) / DOF_TH));
            hipLaunchKernelGGL(k_dof_blur, dim3(n_blocks), dim3(DOF_BLOCK), 0, st, d_rgb, (const float*)d_depth, (int)W,
                               (int)H, depth_power, d_blurred, (const float*)(d_part + 2 * RANGE_BLOCKS + 1), focus);
            if (hipGetLastError() != hipSuccess) rc2 = fail(RT_ERR_HIP, "rt_postprocess: blur launch failed");
        }
        return rc2;
    });
}

extern "C" int rt_dof_blur(int device, int32_t W, int32_t H, float depth_power, const float* rgb,
                           const float* depth_n, float* out)
{
    int rc = check_device(device);
    if (rc) return rc;
    if (W < 13 || H < 13 || W > 32768 || H > 32768)
        return fail(RT_ERR_INVALID, "rt_dof_blur: frame %dx%d must be at least 13x13", W, H);
    if (!rgb || !depth_n || !out) return fail(RT_ERR_INVALID, "rt_dof_blur: null argument");
    const size_t npx = (size_t)W * H, nout = (size_t)(W - 12) * (H - 12);
    DevBuf       drgb, dd, dout;
    if ((rc = drgb.alloc(12 * npx)) || (rc = dd.alloc(4 * npx)) || (rc = dout.alloc(12 * nout)))
        return rc;
    HIP_TRY(hipMemcpy(drgb.p, rgb, 12 * npx, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dd.p, depth_n, 4 * npx, hipMemcpyHostToDevice));
    if ((rc = rt_dof_blur_device(device, W, H, depth_power, (const float*)drgb.p, (const float*)dd.p,
                                 (float*)dout.p, nullptr)))
        return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout.p, 12 * nout, hipMemcpyDeviceToHost));
    return RT_OK;
}

// The 'r' path with the frame kept on the device: render, both normalisations and the blur,
// then the three 8-bit planes the writers put into result.ppm/.bmp, depth.bmp and blurred.bmp
// (58 MB for a 4K frame instead of 166 MB of float planes), copied into pinned host memory the
// scene owns.  Frame buffers and the pinned block are kept between calls and grown on demand.
// With a communicator the frame is shared: this rank renders its tiles (32 x 32, dealt over
// shares, rank 0 possibly fewer than the others), the payloads are gathered on rank 0, which
// rebuilds the frame and goes on alone.
static int gather_rccl(void* comm, int rank, int n_ranks, const float* d_payload, const int64_t* floats_per_rank,
                       float* d_gathered, hipStream_t st);

extern "C" int rt_deal_params(const RtParams* frame, int32_t n_ranks, int32_t first_rank_shares, int32_t rank,
                              RtParams* out)
{
    if (!frame || !out) return fail(RT_ERR_INVALID, "rt_deal_params: null argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(RT_ERR_INVALID, "rt_deal_params: rank %d of %d", rank, n_ranks);
    if (first_rank_shares < 1 || first_rank_shares > RT_SHARES_PER_RANK)
        return fail(RT_ERR_INVALID, "rt_deal_params: first_rank_shares %d not in 1..%d", first_rank_shares, RT_SHARES_PER_RANK);
    *out = *frame;
    out->tile_w = out->tile_h = 32;
    if (n_ranks == 1) {
        out->world = 1, out->rank = 0, out->rank_span = 1;
        return RT_OK;
    }
    out->world = first_rank_shares + RT_SHARES_PER_RANK * (n_ranks - 1);
    out->rank = rank == 0 ? 0 : first_rank_shares + RT_SHARES_PER_RANK * (rank - 1);
    out->rank_span = rank == 0 ? first_rank_shares : RT_SHARES_PER_RANK;
    return RT_OK;
}

extern "C" int rt_scene_set_region_order(RtScene* s, int32_t from_last_frame)
{
    if (!s) return fail(RT_ERR_INVALID, "rt_scene_set_region_order: null scene");
    s->wf_feedback = from_last_frame != 0;
    s->wf_order_key = 0; // whatever was measured before does not count from here on
    return RT_OK;
}

extern "C" int rt_scene_set_deal(RtScene* s, int32_t first_rank_shares)
{
    if (!s) return fail(RT_ERR_INVALID, "rt_scene_set_deal: null scene");
    if (first_rank_shares < 1 || first_rank_shares > RT_SHARES_PER_RANK)
        return fail(RT_ERR_INVALID, "rt_scene_set_deal: first_rank_shares %d not in 1..%d", first_rank_shares, RT_SHARES_PER_RANK);
    s->first_rank_shares = first_rank_shares;
    return RT_OK;
}

// Where the parts of the 'r' path's frame lie: device planes and the pinned host bytes.
struct FramePlan {
    size_t npx, nbl;
    bool   blur;
    size_t o_depth, o_blur, o_rgb8, o_depth8, o_blur8, total; // device
    size_t h_depth8, h_blur8, h_total;                        // host
};

static int frame_buffers(RtScene* s, int W, int H, FramePlan* f)
{
    if (W < 2 || H < 2 || W > 32768 || H > 32768)
        return fail(RT_ERR_INVALID, "frame size %dx%d out of range (2..32768)", W, H);
    HIP_TRY(hipSetDevice(s->device));
    f->npx = (size_t)W * H;
    f->blur = W > 12 && H > 12;
    f->nbl = f->blur ? (size_t)(W - 12) * (H - 12) : 0;
    // device: rgb f32 | depth f32 | blurred f32 | rgb8 | depth8 | blurred8 (each part 256-byte aligned)
    f->o_depth = align256(12 * f->npx), f->o_blur = f->o_depth + align256(4 * f->npx);
    f->o_rgb8 = f->o_blur + align256(12 * f->nbl + 16), f->o_depth8 = f->o_rgb8 + align256(3 * f->npx);
    f->o_blur8 = f->o_depth8 + align256(f->npx), f->total = f->o_blur8 + align256(3 * f->nbl + 16);
    f->h_depth8 = align256(3 * f->npx), f->h_blur8 = f->h_depth8 + align256(f->npx);
    f->h_total = f->h_blur8 + align256(3 * f->nbl + 16);
    // (sizes and offsets depend on the frame's shape, not only on its pixel count: 2 x 5000 has no
    // blurred plane, 100 x 100 has one of 7744 pixels)
    if (f->total > s->frame_dev_bytes) {
        if (s->d_frame) (void)hipFree(s->d_frame);
        s->d_frame = nullptr, s->frame_dev_bytes = 0;
        HIP_TRY(hipMalloc(&s->d_frame, f->total));
        s->frame_dev_bytes = f->total;
    }
    if (f->h_total > s->frame_host_bytes) {
        if (s->h_frame) (void)hipHostFree(s->h_frame);
        s->h_frame = nullptr, s->frame_host_bytes = 0;
        HIP_TRY(hipHostMalloc(&s->h_frame, f->h_total, hipHostMallocDefault));
        s->frame_host_bytes = f->h_total;
    }
    if (!s->frame_stream) HIP_TRY(hipStreamCreateWithFlags(&s->frame_stream, hipStreamNonBlocking));
    return RT_OK;
}

static int share_buffer(RtScene* s, size_t need)
{
    if (need > s->share_bytes) {
        if (s->d_share) (void)hipFree(s->d_share);
        s->d_share = nullptr, s->share_bytes = 0;
        HIP_TRY(hipMalloc(&s->d_share, need));
        s->share_bytes = need;
    }
    return RT_OK;
}

// Post-process of the frame in the scene's buffers and the three planes to the host.
static int finish_planes(RtScene* s, const FramePlan& f, const RtParams* params, const uint8_t** rgb8,
                         const uint8_t** depth8, const uint8_t** blurred8)
{
    const int      W = params->width, H = params->height;
    hipStream_t    st = s->frame_stream;
    unsigned char* d = (unsigned char*)s->d_frame;
    unsigned char* h = (unsigned char*)s->h_frame;
    float*         d_rgb = (float*)d, *d_depth = (float*)(d + f.o_depth), *d_blur = (float*)(d + f.o_blur);
    int            rc;
    hipLaunchKernelGGL(k_quantize, dim3(blocks_for(3 * f.npx, 256)), dim3(256), 0, st, (const float*)d_rgb,
                       (long long)(3 * f.npx), d + f.o_rgb8);
    HIP_TRY(hipMemcpyAsync(h, d + f.o_rgb8, 3 * f.npx, hipMemcpyDeviceToHost, st));
    if ((rc = rt_postprocess_device(s->device, W, H, params->focusDepth, params->depthPower, d_rgb, d_depth,
                                    f.blur ? d_blur : nullptr, st)))
        return rc;
    hipLaunchKernelGGL(k_depth_grey, dim3(blocks_for(f.npx, 256)), dim3(256), 0, st, (const float*)d_depth,
                       (long long)f.npx, d + f.o_depth8);
    HIP_TRY(hipMemcpyAsync(h + f.h_depth8, d + f.o_depth8, f.npx, hipMemcpyDeviceToHost, st));
    if (f.blur) {
        hipLaunchKernelGGL(k_quantize, dim3(blocks_for(3 * f.nbl, 256)), dim3(256), 0, st, (const float*)d_blur,
                           (long long)(3 * f.nbl), d + f.o_blur8);
        HIP_TRY(hipMemcpyAsync(h + f.h_blur8, d + f.o_blur8, 3 * f.nbl, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    *rgb8 = h, *depth8 = h + f.h_depth8, *blurred8 = f.blur ? h + f.h_blur8 : nullptr;
    return RT_OK;
}

static int render_planes_impl(RtScene* s, void* comm, int rank, int n_ranks, const float camera[24],
                              const float* lights, int32_t n_lights, const RtParams* params, const uint8_t** rgb8,
                              const uint8_t** depth8, const uint8_t** blurred8, RtStats* stats)
{
    if (params->tile_w != 0 || params->tile_h != 0)
        return fail(RT_ERR_INVALID, "the result planes are whole frames (tile_w = tile_h = 0); the library deals the tiles");
    FramePlan f;
    int       rc = frame_buffers(s, params->width, params->height, &f);
    if (rc) return rc;
    hipStream_t    st = s->frame_stream;
    unsigned char* d = (unsigned char*)s->d_frame;
    float*         d_rgb = (float*)d, *d_depth = (float*)(d + f.o_depth);
    if (n_ranks > 1) {
        std::vector<RtParams> deal((size_t)n_ranks);
        std::vector<int64_t>  floats((size_t)n_ranks);
        int64_t               all = 0;
        for (int r = 0; r < n_ranks; ++r) {
            if ((rc = rt_deal_params(params, n_ranks, s->first_rank_shares, r, &deal[r]))) return rc;
            floats[r] = rt_tile_pixels(&deal[r], deal[r].rank) * 4, all += floats[r];
        }
        const RtParams& p = deal[rank];
        if ((rc = share_buffer(s, align256((size_t)floats[rank] * 4) + (rank == 0 ? (size_t)all * 4 : 0) + 512))) return rc;
        float* d_payload = (float*)s->d_share;
        float* d_gathered = (float*)((unsigned char*)s->d_share + align256((size_t)floats[rank] * 4));
        if ((rc = rt_render_frame_device(s, camera, lights, n_lights, &p, d_rgb, d_depth, nullptr, st, stats))) return rc;
        if ((rc = rt_pack_tiles_device(s->device, &p, d_rgb, d_depth, d_payload, st))) return rc;
        if ((rc = gather_rccl(comm, rank, n_ranks, d_payload, floats.data(), d_gathered, st))) return rc;
        if (rank != 0) {
            HIP_TRY(hipStreamSynchronize(st));
            *rgb8 = *depth8 = *blurred8 = nullptr;
            return RT_OK;
        }
        int64_t off = floats[0];
        for (int r = 1; r < n_ranks; ++r) { // rank 0's own tiles are in the frame already
            if (floats[r] > 0 &&
                (rc = rt_unpack_tiles_device(s->device, &deal[r], deal[r].rank, d_gathered + off, d_rgb, d_depth, st)))
                return rc;
            off += floats[r];
        }
    } else if ((rc = rt_render_frame_device(s, camera, lights, n_lights, params, d_rgb, d_depth, nullptr, st, stats))) {
        return rc;
    }
    return finish_planes(s, f, params, rgb8, depth8, blurred8);
}

extern "C" int rt_render_result_planes(RtScene* s, const float camera[24], const float* lights, int32_t n_lights,
                                       const RtParams* params, const uint8_t** rgb8, const uint8_t** depth8,
                                       const uint8_t** blurred8, RtStats* stats)
{
    if (!s || !params || !rgb8 || !depth8 || !blurred8)
        return fail(RT_ERR_INVALID, "rt_render_result_planes: null argument");
    return render_planes_impl(s, nullptr, 0, 1, camera, lights, n_lights, params, rgb8, depth8, blurred8, stats);
}

extern "C" int rt_render_result_planes_shared(RtScene* s, void* comm, int32_t rank, int32_t n_ranks,
                                              const float camera[24], const float* lights, int32_t n_lights,
                                              const RtParams* params, const uint8_t** rgb8, const uint8_t** depth8,
                                              const uint8_t** blurred8, RtStats* stats)
{
    if (!s || !params || !rgb8 || !depth8 || !blurred8)
        return fail(RT_ERR_INVALID, "rt_render_result_planes_shared: null argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return fail(RT_ERR_INVALID, "rt_render_result_planes_shared: rank %d of %d", rank, n_ranks);
    if (n_ranks > 1 && !comm) return fail(RT_ERR_INVALID, "rt_render_result_planes_shared: null communicator");
    return render_planes_impl(s, comm, rank, n_ranks, camera, lights, n_lights, params, rgb8, depth8, blurred8, stats);
}

extern "C" int rt_render_result_planes_rehearsal(RtScene* s, int32_t n_ranks, const float camera[24],
                                                 const float* lights, int32_t n_lights, const RtParams* params,
                                                 const uint8_t** rgb8, const uint8_t** depth8,
                                                 const uint8_t** blurred8, RtStats* stats)
{
    if (!s || !params || !rgb8 || !depth8 || !blurred8)
        return fail(RT_ERR_INVALID, "rt_render_result_planes_rehearsal: null argument");
    if (n_ranks < 1 || n_ranks > 4096) return fail(RT_ERR_INVALID, "rt_render_result_planes_rehearsal: %d ranks", n_ranks);
    if (params->tile_w != 0 || params->tile_h != 0)
        return fail(RT_ERR_INVALID, "the result planes are whole frames (tile_w = tile_h = 0); the library deals the tiles");
    FramePlan f;
    int       rc = frame_buffers(s, params->width, params->height, &f);
    if (rc) return rc;
    hipStream_t    st = s->frame_stream;
    unsigned char* d = (unsigned char*)s->d_frame;
    float*         d_rgb = (float*)d, *d_depth = (float*)(d + f.o_depth);
    std::vector<RtParams> deal((size_t)n_ranks);
    std::vector<int64_t>  floats((size_t)n_ranks), off((size_t)n_ranks);
    int64_t               all = 0;
    for (int r = 0; r < n_ranks; ++r) {
        if ((rc = rt_deal_params(params, n_ranks, s->first_rank_shares, r, &deal[r]))) return rc;
        floats[r] = rt_tile_pixels(&deal[r], deal[r].rank) * 4, off[r] = all, all += floats[r];
    }
    // [gathered payloads of all ranks][the frame a rank other than 0 renders into: rgb, depth]
    const size_t o_other = align256((size_t)all * 4 + 16);
    if ((rc = share_buffer(s, o_other + align256(12 * f.npx) + align256(4 * f.npx)))) return rc;
    float* d_gathered = (float*)s->d_share;
    float* o_rgb = (float*)((unsigned char*)s->d_share + o_other);
    float* o_depth = (float*)((unsigned char*)o_rgb + align256(12 * f.npx));
    RtStats sum, one;
    memset(&sum, 0, sizeof sum);
    for (int r = 0; r < n_ranks; ++r) {
        float* t_rgb = r == 0 ? d_rgb : o_rgb, *t_depth = r == 0 ? d_depth : o_depth;
        if ((rc = rt_render_frame_device(s, camera, lights, n_lights, &deal[r], t_rgb, t_depth, nullptr, st,
                                         stats ? &one : nullptr)))
            return rc;
        if (stats) {
            sum.rays_closest += one.rays_closest, sum.rays_shadow += one.rays_shadow, sum.tri_tests += one.tri_tests;
            sum.node_visits += one.node_visits, sum.camera_rays += one.camera_rays, sum.kernel_ms += one.kernel_ms;
            sum.shaded_hits += one.shaded_hits, sum.child_rays += one.child_rays;
        }
        if (floats[r] > 0 && (rc = rt_pack_tiles_device(s->device, &deal[r], t_rgb, t_depth, d_gathered + off[r], st))) return rc;
    }
    for (int r = 1; r < n_ranks; ++r)
        if (floats[r] > 0 &&
            (rc = rt_unpack_tiles_device(s->device, &deal[r], deal[r].rank, d_gathered + off[r], d_rgb, d_depth, st)))
            return rc;
    if (stats) *stats = sum;
    return finish_planes(s, f, params, rgb8, depth8, blurred8);
}

// One frame part on the frame stream, timed with the scene's event pair: ms, or < 0 on an error.
static int timed_part(RtScene* s, const float camera[24], const float* lights, int32_t n_lights, const RtParams* p,
                      float* d_rgb, float* d_depth, int runs, float* ms)
{
    hipStream_t st = s->frame_stream;
    int         rc;
    if ((rc = rt_render_frame_device(s, camera, lights, n_lights, p, d_rgb, d_depth, nullptr, st, nullptr))) return rc; // warm-up
    HIP_TRY(hipEventRecord(s->ev0, st));
    for (int i = 0; i < runs; ++i)
        if ((rc = rt_render_frame_device(s, camera, lights, n_lights, p, d_rgb, d_depth, nullptr, st, nullptr))) return rc;
    HIP_TRY(hipEventRecord(s->ev1, st));
    HIP_TRY(hipEventSynchronize(s->ev1));
    HIP_TRY(hipEventElapsedTime(ms, s->ev0, s->ev1));
    *ms /= (float)runs;
    return RT_OK;
}

static int bcast_int_rccl(void* comm, int rank, int n_ranks, int32_t* value, int32_t* d_word, hipStream_t st);

extern "C" int rt_calibrate_deal(RtScene* s, void* comm, int32_t rank, int32_t n_ranks, const float camera[24],
                                 const float* lights, int32_t n_lights, const RtParams* params,
                                 int32_t* first_rank_shares)
{
    if (!s || !params || !camera) return fail(RT_ERR_INVALID, "rt_calibrate_deal: null argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(RT_ERR_INVALID, "rt_calibrate_deal: rank %d of %d", rank, n_ranks);
    if (n_ranks > 1 && !comm) return fail(RT_ERR_INVALID, "rt_calibrate_deal: null communicator");
    if (params->tile_w != 0 || params->tile_h != 0) return fail(RT_ERR_INVALID, "rt_calibrate_deal: whole-frame params expected");
    int32_t shares0 = RT_SHARES_PER_RANK;
    if (n_ranks > 1) {
        FramePlan f;
        int       rc = frame_buffers(s, params->width, params->height, &f);
        if (rc) return rc;
        hipStream_t    st = s->frame_stream;
        unsigned char* d = (unsigned char*)s->d_frame;
        float*         d_rgb = (float*)d, *d_depth = (float*)(d + f.o_depth), *d_blur = (float*)(d + f.o_blur);
        if (rank == 0) {
            RtParams even, half, whole = *params;
            if ((rc = rt_deal_params(params, n_ranks, RT_SHARES_PER_RANK, 0, &even))) return rc;
            if ((rc = rt_deal_params(params, n_ranks, RT_SHARES_PER_RANK / 2, 0, &half))) return rc;
            float t_even = 0, t_half = 0, t_post = 0;
            if ((rc = timed_part(s, camera, lights, n_lights, &even, d_rgb, d_depth, 2, &t_even))) return rc;
            if ((rc = timed_part(s, camera, lights, n_lights, &half, d_rgb, d_depth, 2, &t_half))) return rc;
            // the post-process on a real frame: the whole frame once, then the passes timed on copies of its depth
            if ((rc = rt_render_frame_device(s, camera, lights, n_lights, &whole, d_rgb, d_depth, nullptr, st, nullptr))) return rc;
            if ((rc = share_buffer(s, align256(4 * f.npx) + 512))) return rc;
            HIP_TRY(hipMemcpyAsync(s->d_share, d_depth, 4 * f.npx, hipMemcpyDeviceToDevice, st));
            for (int run = 0; run < 3; ++run) { // the first run warms up, two are timed
                if (run == 1) HIP_TRY(hipEventRecord(s->ev0, st));
                HIP_TRY(hipMemcpyAsync(d_depth, s->d_share, 4 * f.npx, hipMemcpyDeviceToDevice, st));
                if ((rc = rt_postprocess_device(s->device, params->width, params->height, params->focusDepth,
                                                params->depthPower, d_rgb, d_depth, f.blur ? d_blur : nullptr, st)))
                    return rc;
            }
            HIP_TRY(hipEventRecord(s->ev1, st));
            HIP_TRY(hipEventSynchronize(s->ev1));
            HIP_TRY(hipEventElapsedTime(&t_post, s->ev0, s->ev1));
            t_post /= 2.0f; // (each run with the copy of one depth plane in front: microseconds)
            const double f_even = 1.0 / n_ranks;
            const double f_half = (RT_SHARES_PER_RANK / 2) / (double)(RT_SHARES_PER_RANK / 2 + RT_SHARES_PER_RANK * (n_ranks - 1));
            const double slope = ((double)t_even - (double)t_half) / (f_even - f_half); // ms per unit share of the frame
            if (slope > 0.0) {
                double f0 = (1.0 - (n_ranks - 1) * ((double)t_post + 0.1) / slope) / n_ranks; // + 0.1 ms for the unpack kernels
                f0 = std::min(std::max(f0, 0.2 / n_ranks), 1.0 / n_ranks);
                shares0 = (int32_t)std::lround(RT_SHARES_PER_RANK * (n_ranks - 1) * f0 / (1.0 - f0));
            }
            shares0 = std::min<int32_t>(std::max<int32_t>(shares0, 1), RT_SHARES_PER_RANK);
            s->calib_ms[0] = t_even, s->calib_ms[1] = t_half, s->calib_ms[2] = t_post;
        }
        if ((rc = bcast_int_rccl(comm, rank, n_ranks, &shares0, (int32_t*)s->d_ctr, st))) return rc;
        if (shares0 < 1 || shares0 > RT_SHARES_PER_RANK) return fail(RT_ERR_HIP, "rt_calibrate_deal: received %d shares", shares0);
    }
    s->first_rank_shares = shares0;
    if (first_rank_shares) *first_rank_shares = shares0;
    return RT_OK;
}

extern "C" int rt_quantize(int device, int64_t n, const float* v, uint8_t* out)
{
    int rc = check_device(device);
    if (rc) return rc;
    if (n < 0) return fail(RT_ERR_INVALID, "rt_quantize: negative n");
    if (n == 0) return RT_OK;
    if (!v || !out) return fail(RT_ERR_INVALID, "rt_quantize: null argument");
    DevBuf dv, dout;
    if ((rc = dv.alloc(4 * n)) || (rc = dout.alloc(n))) return rc;
    HIP_TRY(hipMemcpy(dv.p, v, 4 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_quantize, dim3(blocks_for(n, 256)), dim3(256), 0, 0, (const float*)dv.p,
                       (long long)n, (unsigned char*)dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout.p, n, hipMemcpyDeviceToHost));
    return RT_OK;
}

// ---------------------------------------------------------------------------
// tiles
// ---------------------------------------------------------------------------
extern "C" int rt_pack_tiles_device(int device, const RtParams* params, const float* d_rgb,
                                    const float* d_depth, float* d_payload, void* stream)
{
    int rc = check_device(device);
    if (rc) return rc;
    if (!params || !d_rgb || !d_depth || !d_payload)
        return fail(RT_ERR_INVALID, "rt_pack_tiles: null argument");
    DParams P;
    memset(&P, 0, sizeof P);
    P.W = params->width, P.H = params->height;
    if ((rc = set_partition(params, params->rank, &P))) return rc;
    const long long n = (long long)P.n_my_tiles * P.tile_w * P.tile_h;
    if (n == 0) return RT_OK;
    hipLaunchKernelGGL(k_pack_tiles, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, P, n,
                       d_rgb, d_depth, (float4*)d_payload);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

extern "C" int rt_unpack_tiles_device(int device, const RtParams* params, int32_t src_rank,
                                      const float* d_payload, float* d_rgb, float* d_depth,
                                      void* stream)
{
    int rc = check_device(device);
    if (rc) return rc;
    if (!params || !d_rgb || !d_depth || !d_payload)
        return fail(RT_ERR_INVALID, "rt_unpack_tiles: null argument");
    DParams P;
    memset(&P, 0, sizeof P);
    P.W = params->width, P.H = params->height;
    if ((rc = set_partition(params, src_rank, &P))) return rc;
    const long long n = (long long)P.n_my_tiles * P.tile_w * P.tile_h;
    if (n == 0) return RT_OK;
    hipLaunchKernelGGL(k_unpack_tiles, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, P,
                       n, (const float4*)d_payload, d_rgb, d_depth);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

// ---------------------------------------------------------------------------
// several GPUs: the gather of the tile payloads over RCCL
// ---------------------------------------------------------------------------
// RCCL is bound at run time (dlopen): a process that never shares a frame does not load it, and
// a process that already holds a copy (PyTorch ships one under the same soname) uses that one.
namespace {
struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
RcclApi    g_rccl;
std::mutex g_rccl_mutex;

int rccl_load()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.ok) return RT_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(RT_ERR_UNSUPPORTED, "RCCL (librccl.so.1) cannot be loaded: %s", dlerror());
    bool all = true;
    auto sym = [&](const char* name) {
        void* p = dlsym(h, name);
        if (!p) all = false;
        return p;
    };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
    g_rccl.Send = (decltype(g_rccl.Send))sym("ncclSend");
    g_rccl.Recv = (decltype(g_rccl.Recv))sym("ncclRecv");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    if (!all) return fail(RT_ERR_UNSUPPORTED, "librccl.so.1 lacks a symbol the gather needs");
    g_rccl.ok = true;
    return RT_OK;
}
} // namespace

#define RCCL_TRY(expr)                                                                             \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess)                                                                     \
            return fail(RT_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));            \
    } while (0)

extern "C" int rt_rccl_unique_id(void* id128)
{
    if (!id128) return fail(RT_ERR_INVALID, "rt_rccl_unique_id: null argument");
    int rc = rccl_load();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == RT_RCCL_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return RT_OK;
}

extern "C" int rt_rccl_comm_create(const void* id128, int32_t n_ranks, int32_t rank, int device, void** comm)
{
    if (!id128 || !comm) return fail(RT_ERR_INVALID, "rt_rccl_comm_create: null argument");
    *comm = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return fail(RT_ERR_INVALID, "rt_rccl_comm_create: rank %d of %d", rank, n_ranks);
    int rc = check_device(device);
    if (rc) return rc;
    if ((rc = rccl_load())) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    RCCL_TRY(g_rccl.CommInitRank(&c, n_ranks, id, rank));
    *comm = (void*)c;
    return RT_OK;
}

extern "C" void rt_rccl_comm_destroy(void* comm)
{
    if (comm && g_rccl.ok) (void)g_rccl.CommDestroy((ncclComm_t)comm);
}

extern "C" int rt_gather_tiles_rccl(void* comm, int32_t rank, int32_t n_ranks, const float* d_payload,
                                    const int64_t* floats_per_rank, float* d_gathered, void* stream_)
{
    return gather_rccl(comm, rank, n_ranks, d_payload, floats_per_rank, d_gathered, (hipStream_t)stream_);
}

// One 32-bit value from rank 0 to every rank: through a device word, grouped sends on rank 0.
static int bcast_int_rccl(void* comm, int rank, int n_ranks, int32_t* value, int32_t* d_word, hipStream_t st)
{
    if (n_ranks == 1) return RT_OK;
    int rc = rccl_load();
    if (rc) return rc;
    if (rank == 0) {
        HIP_TRY(hipMemcpyAsync(d_word, value, 4, hipMemcpyHostToDevice, st));
        RCCL_TRY(g_rccl.GroupStart());
        for (int r = 1; r < n_ranks; ++r) RCCL_TRY(g_rccl.Send(d_word, 1, ncclFloat, r, (ncclComm_t)comm, st));
        RCCL_TRY(g_rccl.GroupEnd());
        HIP_TRY(hipStreamSynchronize(st));
    } else {
        RCCL_TRY(g_rccl.GroupStart());
        RCCL_TRY(g_rccl.Recv(d_word, 1, ncclFloat, 0, (ncclComm_t)comm, st));
        RCCL_TRY(g_rccl.GroupEnd());
        HIP_TRY(hipMemcpyAsync(value, d_word, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return RT_OK;
}

static int gather_rccl(void* comm, int rank, int n_ranks, const float* d_payload, const int64_t* floats_per_rank,
                       float* d_gathered, hipStream_t st)
{
    if (!comm || !floats_per_rank) return fail(RT_ERR_INVALID, "rt_gather_tiles_rccl: null argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return fail(RT_ERR_INVALID, "rt_gather_tiles_rccl: rank %d of %d", rank, n_ranks);
    int64_t total = 0;
    for (int r = 0; r < n_ranks; ++r) {
        if (floats_per_rank[r] < 0 || floats_per_rank[r] > ((int64_t)1 << 34))
            return fail(RT_ERR_INVALID, "rt_gather_tiles_rccl: payload size of rank %d out of range", r);
        total += floats_per_rank[r];
    }
    if (floats_per_rank[rank] > 0 && !d_payload) return fail(RT_ERR_INVALID, "rt_gather_tiles_rccl: null payload");
    if (rank == 0 && total > 0 && !d_gathered) return fail(RT_ERR_INVALID, "rt_gather_tiles_rccl: null gather buffer on rank 0");
    int rc = rccl_load();
    if (rc) return rc;
    if (rank == 0) { // rank 0's own payload moves inside the device; every peer sends over its own xGMI link
        if (floats_per_rank[0] > 0)
            HIP_TRY(hipMemcpyAsync(d_gathered, d_payload, (size_t)floats_per_rank[0] * 4, hipMemcpyDeviceToDevice, st));
        if (n_ranks == 1) return RT_OK;
        RCCL_TRY(g_rccl.GroupStart());
        int64_t off = floats_per_rank[0];
        for (int r = 1; r < n_ranks; ++r) {
            if (floats_per_rank[r] > 0)
                RCCL_TRY(g_rccl.Recv(d_gathered + off, (size_t)floats_per_rank[r], ncclFloat, r, (ncclComm_t)comm, st));
            off += floats_per_rank[r];
        }
        RCCL_TRY(g_rccl.GroupEnd());
    } else if (floats_per_rank[rank] > 0) {
        RCCL_TRY(g_rccl.GroupStart());
        RCCL_TRY(g_rccl.Send(d_payload, (size_t)floats_per_rank[rank], ncclFloat, 0, (ncclComm_t)comm, st));
        RCCL_TRY(g_rccl.GroupEnd());
    }
    return RT_OK;
}
