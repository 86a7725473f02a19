// rt_wavefront.h - the frame renderer as a wavefront of ray queues.
//
// The reference shades a camera sample by recursion (raytracing.cpp:61-274):
// closest hit, one shadow query per light, then up to two child rays, five
// levels deep.  Here every level of that recursion is one pass over a queue of
// rays held in HBM:
//
//   k_wf_trace<false>   closest hit of every ray of the level   -> hit records
//   k_wf_shade          hit geometry; emits the shadow rays (one queue per
//                       light) and the child rays of the next level (compacted
//                       with wave ballots + one atomic per wave); misses add
//                       the background colour
//   k_wf_trace<true>    any-hit query per shadow ray             -> occlusion flags
//   k_wf_resolve        the light loop with the occlusion flags known; adds the
//                       hit's weighted colour to its camera sample
//
// Queues are cut into regions.  Region r of every queue of a chunk belongs to
// the same S (128 to 512) pixel slots: their camera rays, then whatever those
// rays spawn, level after level.  A producer block owns one region, appends
// into it densely with an LDS counter and publishes the fill count; no global
// atomic is involved and a region cannot overflow (its size is the largest
// number of rays its pixels can have in flight).  k_wf_worklist turns the fill
// counts into a list of 64-ray work items; the two trace kernels are
// persistent: a fixed grid of waves pulls items off that list with one atomic
// per 256 rays until it is empty, so no launch depends on a count the host
// would have to read back.  They keep the KD nodes and the traversal stack in
// LDS and none of the shading state in registers, which is what gives them the
// occupancy to hide the latency of the dependent node -> triangle loads.
//
// A camera sample's colour is the sum over its ray tree of (path weight x
// local colour); the recursion is linear in the child colours
// (raytracing.cpp:191, 240, 259, 268).  The sum is accumulated in 64-bit fixed
// point (2^-32 units), so it does not depend on the order in which rays
// finish: frames are bit-reproducible and independent of the tile partition.
#pragma once
#include "rt_device.h"

#pragma clang fp contract(off)

#ifndef WF_BLOCK
#define WF_BLOCK 256
#endif
#define WF_MAX_DEPTH 6 // ray levels k = 0..5 (raytracing.cpp:63,166-168)
#define WF_FIX_SCALE 4294967296.0

__device__ __forceinline__ float wf_clamp01(float v) // RGBValue, imageWriter.h:26-44
{
    if (v > 1.0f) v = 1.0f;
    if (v < 0.0f) v = 0.0f;
    return v;
}

#define WF_ITEMS_PER_FETCH 4 // 64-ray work items taken per atomic

struct WfCounters {
    int head_rays[WF_MAX_DEPTH + 1]; // persistent-kernel fetch counters, one per launch
    int head_shadow[WF_MAX_DEPTH][RT_MAX_LIGHTS_DEV];
    int head_region;                 // k_wf_pipeline: next region to hand out
    int pad;
};

// Block-local state of the producer stages (LDS).
struct WfShared {
    int child;                  // rays appended to the next level's region
    int n;                      // rays appended by the camera stages
    int region;                 // k_wf_pipeline: the region the block works on
    int sh;                     // shadow rays appended
};

// One queue of rays (structure of arrays); region r occupies [r * R, (r + 1) * R).
struct WfRays {
    float4* o_w;  // origin xyz, path weight
    float4* d_ni; // destination xyz, prev_Ni
    uint2*  meta; // x: sample index, y: k | shared << 8
    int*    cnt;  // [G] rays in each region
    float4* hits; // per ray slot: bits(orig triangle or -1), a, b, t - one set per queue, so the
                  // closest hits of a level can be traced while the level before is still resolved
};

struct WfBuffers {
    WfRays         rays[2];   // ping-pong between levels
    float*         wlocal;    // per ray slot: weight of the hit's own colour (0 for a miss)
    float4*        sh_org;    // [cap * lights * ns]: shadow ray origin xyz, bits(light << 24 | ray slot in its region);
                              // one queue per region for all lights, R * lights * ns entries each
    float4*        sh_dst;    // [cap * lights * ns]: soft shadows only, the sample's destination
    int*           sh_cnt;    // [G]
    unsigned char* occl;      // [light][cap]: 1 = occluded (soft shadows: occluded samples), indexed by ray slot
    int*           prefix;    // [G + 1] for the ray queue, then [G + 1] for the shadow queue
    long long*     acc;       // [3][n_samples] fixed-point colour sums
    float*         s_minT;    // per sample: t of the camera ray's hit (FLT_MAX = miss)
    int*           s_tri;     // per sample: triangle of the camera ray's hit
    WfCounters*    ctr;
    int            cap;       // slots of every queue = G * R
    int            G, R;      // regions, ray slots per region
    int            S;         // pixel slots per region
    int            n_samples; // samples of a chunk (upper bound)
    int            ns;        // shadow rays per hit and light: 1, or soft_n^3 with soft shadows
};

// ---------------------------------------------------------------------------
// work lists and the persistent trace kernels
// ---------------------------------------------------------------------------
// Work-item prefix sums: item counts ceil(cnt / 64) per region, exclusive scan.
// Block 0: the G regions of ray queue `q`.  Block 1: the G regions of the shadow queue.
__global__ void __launch_bounds__(1024) k_wf_worklist(WfBuffers B, int q)
{
    __shared__ int s_part[1024];
    const int      n = B.G;
    const int*     cnt = blockIdx.x == 0 ? B.rays[q].cnt : B.sh_cnt;
    int*           out = blockIdx.x == 0 ? B.prefix : B.prefix + (B.G + 1);
    const int      per = (n + 1023) / 1024;
    const int      lo = min((int)threadIdx.x * per, n), hi = min(lo + per, n);
    int            sum = 0;
    for (int i = lo; i < hi; ++i) sum += (cnt[i] + 63) >> 6;
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) { // inclusive scan of the per-thread sums
        int v = 0;
        if ((int)threadIdx.x >= off) v = s_part[threadIdx.x - off];
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = s_part[threadIdx.x] - sum;
    for (int i = lo; i < hi; ++i) {
        out[i] = run;
        run += (cnt[i] + 63) >> 6;
    }
    if (threadIdx.x == 1023) out[n] = s_part[1023];
}

// Per-thread traversal stack.  An entry is the far child still to be visited (16-bit node
// index) and the ray parameter at which the ray enters it.  Entry e of a thread sits at
// index e * WF_BLOCK of the LDS arrays while e < cap, else at (e - cap) * WF_BLOCK of the
// spill arrays in HBM.  Rays rarely have more than a dozen far children pending, so a short
// LDS part serves nearly all pushes and pops while the tree's full depth stays available;
// LDS per block, and with it the number of resident waves, does not grow with the depth of
// the tree.
struct WfStack {
    unsigned short* lds_node;
    float*          lds_t;
    unsigned short* spill_node;
    float*          spill_t;
    int             cap;
};

// LDS layout per block: float t[cap][WF_BLOCK], then ushort node[cap][WF_BLOCK]
#define WF_STACK_ENTRY_BYTES 6

__device__ __forceinline__ WfStack wf_stack(const DScene& S, unsigned char* s_mem)
{
    WfStack k;
    k.cap = S.stk_cap;
    k.lds_t = (float*)s_mem + threadIdx.x;
    k.lds_node = (unsigned short*)(s_mem + (size_t)k.cap * WF_BLOCK * sizeof(float)) + threadIdx.x;
    const size_t deep = (size_t)(S.stack_entries - S.stk_cap) * WF_BLOCK;
    k.spill_t = S.stk_spill_t + (size_t)blockIdx.x * deep + threadIdx.x;
    k.spill_node = S.stk_spill + (size_t)blockIdx.x * deep + threadIdx.x;
    return k;
}

__device__ __forceinline__ void wf_stk_put(const WfStack& k, int e, int node, float t)
{
    if (e < k.cap) k.lds_node[e * WF_BLOCK] = (unsigned short)node, k.lds_t[e * WF_BLOCK] = t;
    else k.spill_node[(size_t)(e - k.cap) * WF_BLOCK] = (unsigned short)node, k.spill_t[(size_t)(e - k.cap) * WF_BLOCK] = t;
}

__device__ __forceinline__ int wf_stk_node(const WfStack& k, int e)
{
    return e < k.cap ? k.lds_node[e * WF_BLOCK] : k.spill_node[(size_t)(e - k.cap) * WF_BLOCK];
}

__device__ __forceinline__ float wf_stk_t(const WfStack& k, int e)
{
    return e < k.cap ? k.lds_t[e * WF_BLOCK] : k.spill_t[(size_t)(e - k.cap) * WF_BLOCK];
}

// Closest-hit (ANY = false) or any-hit (ANY = true) walk of one ray over the compact node
// array.  Inner nodes carry no triangles (they were handed down to the leaves).  The
// traversal arithmetic (plane distances) is free to use fused multiply-add: it only
// decides which cells are visited, with a safety margin; which triangle is hit, and where,
// is decided by tri_test alone.  The result equals the reference's leaf walk
// (raytracing.cpp:281-351): the smallest t over all triangles, ties to the triangle that
// walk would have tested last.  COUNT = false leaves the per-node and per-test statistics
// out of the loops (two VALU instructions per step that only RtStats reads).
// Tried and not kept: fetching the node through the scalar cache while all lanes of a wave
// sit at the same node (13.0 vs 12.0 ms per 4K frame: the uniformity test and the wait for
// the scalar load cost more than the per-lane axis selects they save); skipping the re-test
// of the triangle that already is the closest hit (3.5 % of the tests, no gain).
// A plane filter in front of the exact test - three tests in four concern a triangle whose
// plane the ray crosses outside the current cell, and such a crossing can be skipped with a
// 16-byte plane record and ~20 instructions - removed 78 % of the exact tests (11.2 -> 2.5 per
// ray) with all results unchanged, and made the frame slower (12.6 vs 12.05 ms): the extra
// load per test costs more than the arithmetic it saves, and a large triangle no longer sets
// the closest hit early from a distant cell (6 % more node steps).
template <bool ANY, bool COUNT = true>
__device__ __forceinline__ bool wf_traverse(const DScene& S, const WfStack& stk, f3 o, f3 d, HitRec& h,
                                            unsigned& n_tests, unsigned& n_nodes)
{
    float minT = RT_FLT_MAX;
    h.slot = -1, h.a = 0.0f, h.b = 0.0f;
    int   best_orig = -1;
    int   node = 0, sp = 0;
    float tmin_e = 0.0f, tmax_e = RT_FLT_MAX; // entry / exit t of the current cell, margins applied
    // per axis: 1/d, o/d and whether d > 0.  A zero component gets a huge positive stand-in:
    // the plane is then "reached" only at +-infinity-like t with the correct side.
    const float big = 1e18f;
    const float ix = d.x != 0.0f ? __builtin_amdgcn_rcpf(d.x) : big,
                iy = d.y != 0.0f ? __builtin_amdgcn_rcpf(d.y) : big,
                iz = d.z != 0.0f ? __builtin_amdgcn_rcpf(d.z) : big;
    const float oix = o.x * ix, oiy = o.y * iy, oiz = o.z * iz;
    for (;;) {
        uint2 nd = S.knodes[node];
        if (COUNT) ++n_nodes;
        while ((nd.y >> 30) != 3u) {
            const unsigned axis = nd.y >> 30;
            const int      left = (int)(nd.y & 0x3fffffffu);
            const float    ia = axis == 0 ? ix : (axis == 1 ? iy : iz);
            const float    oi = axis == 0 ? oix : (axis == 1 ? oiy : oiz);
            const float    tplane = __builtin_fmaf(__uint_as_float(nd.x), ia, -oi);
            // origin side of the plane: below it iff (split - o) > 0 iff tplane and d (hence 1/d) agree in sign
            const bool below = (tplane > 0.0f) == (ia > 0.0f);
            const int  near = left + (below ? 0 : 1);
            if (tplane <= 0.0f || !(tplane <= tmax_e)) {
                node = near;
            } else if (tplane < tmin_e) {
                node = left + (below ? 1 : 0);
            } else {
                wf_stk_put(stk, sp, left + (below ? 1 : 0), tplane);
                ++sp;
                node = near;
                tmax_e = tplane * (1.0f + RT_TRAV_EPS);
            }
            nd = S.knodes[node];
            if (COUNT) ++n_nodes;
        }
        const int cnt = (int)(nd.y & 0x3fffffffu);
        const int first = (int)nd.x;
        for (int i = 0; i < cnt; ++i) {
            const int    slot = first + i;
            const float4 q0 = S.tris[3 * slot], q1 = S.tris[3 * slot + 1], q2 = S.tris[3 * slot + 2];
            float        a, b, t;
            if (COUNT) ++n_tests;
            if (tri_test(q0, q1, q2, o, d, minT, a, b, t)) {
                if (ANY) return true;
                const int orig = __float_as_int(q2.w);
                bool      take = t != minT || best_orig < 0;
                if (!take) take = S.tri_rank[orig] > S.tri_rank[best_orig];
                if (take) h.a = a, h.b = b, minT = t, best_orig = orig;
            }
        }
        // A pending subtree starts where the cell above it on the stack ends, give or take the
        // margins: entry t of the top entry = tmax_e / (1 + eps), and every deeper entry starts
        // at most a factor (1 + eps) per level earlier.  A hit clearly inside the current cell
        // therefore ends the walk without popping (and loading) the entries one by one.
        if (!ANY && minT < tmax_e * 0.998f) break;
        bool found = false;
        while (sp > 0) {
            --sp;
            const float et = wf_stk_t(stk, sp);
            if (!ANY && et * (1.0f - RT_TRAV_EPS) > minT) continue;
            node = wf_stk_node(stk, sp);
            tmin_e = et * (1.0f - RT_TRAV_EPS);
            tmax_e = sp > 0 ? wf_stk_t(stk, sp - 1) * (1.0f + RT_TRAV_EPS) : RT_FLT_MAX;
            found = true;
            break;
        }
        if (!found) break;
    }
    h.t = minT;
    h.slot = best_orig; // the triangle, not the slot
    return best_orig >= 0;
}

// Persistent trace kernel: a fixed grid of waves pulls 64-ray work items off the list
// (WF_ITEMS_PER_FETCH per atomic) until it is empty.  The 64 lanes of a wave start
// their rays together: neighbouring rays visit the same nodes and leaves at the same
// time, so most of a wave's loads fall into a few cache lines.  (Refilling single lanes
// as their rays finish was measured: it keeps more lanes busy but scatters the lanes
// over the tree, and the extra cache lines per load cost more than the idle lanes.)
//   ANY = false: queue of a level -> hit records.
//   ANY = true : shadow queue of a level (regions of R * n_lights slots); every ray ends at
//                the light its record names -> occlusion bytes.
// LDS: per-thread stacks of ushort node indices.  (Keeping the top of the node array in
// LDS too was measured: 27.7 ms per 4K frame without it, 29.9 / 32.5 / 39.5 ms with 32 / 48 /
// 64 KB per block - the nodes stay in the vector L1 and the LDS costs occupancy.)
template <bool ANY>
__global__ void __launch_bounds__(WF_BLOCK)
k_wf_trace(DScene S, const float4* __restrict__ o_w, const float4* __restrict__ d_ni, DParams const* lights,
           const int* __restrict__ cnt, const int* __restrict__ prefix, int n_lights, int n_regions, int R, int cap,
           int* head, float4* hits, unsigned char* occl, DCounters* ctr)
{
    extern __shared__ __align__(16) unsigned char s_mem[];
    const WfStack   stk = wf_stack(S, s_mem);
    const int       total = prefix[n_regions];
    const int       lane = threadIdx.x & 63;
    unsigned        n_tests = 0, n_nodes = 0, n_rays = 0;
    int             r = 0; // region of the previous item: consecutive items mostly share it
    // several items per atomic while there is plenty of work, single items when the
    // list is short compared with the waves in flight (keeps the tail of the launch short)
    const int per_fetch = total >= 8 * (int)gridDim.x * (WF_BLOCK / 64) ? WF_ITEMS_PER_FETCH : 1;
    for (;;) {
        int t0 = 0;
        if (lane == 0) t0 = atomicAdd(head, per_fetch);
        t0 = __builtin_amdgcn_readfirstlane(t0);
        if (t0 >= total) break;
        const int t1 = min(t0 + per_fetch, total);
        for (int t = t0; t < t1; ++t) {
            if (!(prefix[r] <= t && t < prefix[r + 1])) { // wave-uniform binary search
                int lo = 0, hi = n_regions; // invariant: prefix[lo] <= t < prefix[hi]
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (prefix[mid] <= t) lo = mid;
                    else hi = mid;
                }
                r = lo;
            }
            const int idx = (t - prefix[r]) * 64 + lane;
            if (idx >= cnt[r]) continue;
            const size_t i = (size_t)r * R * (ANY ? n_lights : 1) + idx;
            const float4 ow = o_w[i];
            const f3     o = mk3(ow.x, ow.y, ow.z);
            const int    li = ANY ? (int)((unsigned)__float_as_int(ow.w) >> 24) : 0;
            f3           dst;
            if (ANY) dst = mk3(lights->lights[3 * li], lights->lights[3 * li + 1], lights->lights[3 * li + 2]);
            else {
                const float4 dn = d_ni[i];
                dst = mk3(dn.x, dn.y, dn.z);
            }
            HitRec h;
            ++n_rays;
            const bool got = wf_traverse<ANY>(S, stk, o, dst - o, h, n_tests, n_nodes);
            if (ANY) occl[(size_t)li * cap + (size_t)r * R + (__float_as_int(ow.w) & 0xffffff)] = got ? 1 : 0;
            else hits[i] = make_float4(__int_as_float(got ? h.slot : -1), h.a, h.b, h.t);
        }
    }
    for (int s = 32; s > 0; s >>= 1) {
        n_tests += __shfl_down(n_tests, s);
        n_nodes += __shfl_down(n_nodes, s);
        n_rays += __shfl_down(n_rays, s);
    }
    if (lane == 0) {
        atomicAdd(&ctr->tri_tests, (unsigned long long)n_tests);
        atomicAdd(&ctr->node_visits, (unsigned long long)n_nodes);
        atomicAdd(ANY ? &ctr->rays_shadow : &ctr->rays_closest, (unsigned long long)n_rays);
    }
}

// ---------------------------------------------------------------------------
// queue helpers
// ---------------------------------------------------------------------------
// Positions inside the block's own region: `per` consecutive slots for every
// lane with mine == true; one LDS atomic per wave.
__device__ __forceinline__ int wf_append(int* lds_counter, bool mine, int per = 1)
{
    const unsigned long long m = __ballot(mine);
    if (m == 0) return -1;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    int       base = 0;
    if (lane == leader) base = atomicAdd(lds_counter, __popcll(m) * per);
    base = __shfl(base, leader);
    return mine ? base + __popcll(m & ((1ull << lane) - 1ull)) * per : -1;
}

__device__ __forceinline__ void wf_acc_add(const WfBuffers& B, unsigned sample, f3 c, bool shared)
{
    const long long r = (long long)llrint((double)c.x * WF_FIX_SCALE);
    const long long g = (long long)llrint((double)c.y * WF_FIX_SCALE);
    const long long b = (long long)llrint((double)c.z * WF_FIX_SCALE);
    long long*      pr = B.acc + sample;
    long long*      pg = B.acc + (size_t)B.n_samples + sample;
    long long*      pb = B.acc + 2 * (size_t)B.n_samples + sample;
    if (shared) { // several rays of this sample may be in the level: order-free integer adds
        atomicAdd((unsigned long long*)pr, (unsigned long long)r);
        atomicAdd((unsigned long long*)pg, (unsigned long long)g);
        atomicAdd((unsigned long long*)pb, (unsigned long long)b);
    } else { // the only ray of its sample in this level
        *pr += r;
        *pg += g;
        *pb += b;
    }
}

// Hit geometry shared by k_wf_shade and k_wf_resolve: raytracing.cpp:76-108.
struct WfSurf {
    f3    hp, N, viewDir, dir;
    f3    Kd, Ks;
    float Ns, Ni;
    int   illum, has_Ni, tex;
    int4  info;
};

__device__ __forceinline__ void wf_surface(const DScene& S, f3 o, f3 dst, float4 hit, WfSurf& s)
{
    const int orig = __float_as_int(hit.x);
    s.dir = dst - o;
    s.hp = o + s.dir * hit.w; // raytracing.cpp:384
    const float4 s0 = S.tri_shade[3 * orig], s1 = S.tri_shade[3 * orig + 1], s2 = S.tri_shade[3 * orig + 2];
    s.info = S.tri_info[orig];
    const float4 m0 = S.mats[4 * s.info.x], m1 = S.mats[4 * s.info.x + 1], m2 = S.mats[4 * s.info.x + 2];
    s.Kd = mk3(m0.x, m0.y, m0.z), s.Ks = mk3(m1.x, m1.y, m1.z);
    s.Ns = m0.w, s.Ni = m1.w;
    s.illum = __float_as_int(m2.x), s.has_Ni = __float_as_int(m2.y), s.tex = __float_as_int(m2.z);
    const float u = hit.y, v = hit.z, w = 1.0f - u - v;
    s.N = (mk3(s0.x, s0.y, s0.z) * u + mk3(s1.x, s1.y, s1.z) * v) + mk3(s2.x, s2.y, s2.z) * w;
    normalize3(s.N);
    s.viewDir = o - s.hp;
    normalize3(s.viewDir);
}

// Per-light terms of raytracing.cpp:110-143 up to the shadow query: returns
// whether a shadow ray is cast and the self-shadow value set before it.
__device__ __forceinline__ bool wf_light_setup(const WfSurf& s, f3 lpos, f3& lightDir, float& angle,
                                               float& shadowMax)
{
    lightDir = lpos - s.hp;
    normalize3(lightDir);
    angle = dot3(lightDir, s.N);
    if (angle <= 0.2f) shadowMax = 1.0f;
    else if (angle < 0.5f) {
        angle -= 0.2f;
        shadowMax = 3.33f * (0.3f - angle);
    }
    return angle > 0.0f;
}

// ---------------------------------------------------------------------------
// shade: one block per region, one thread per ray of level `level`
// ---------------------------------------------------------------------------
template <bool SOFT = false, class ST = DScene, class PT = DParams>
__device__ __forceinline__ void wf_shade_region(const ST& S_, const PT& P, const WfBuffers& B,
                                                const WfRays in, const WfRays out, int r, WfShared* sh,
                                                unsigned& n_shaded, unsigned& n_child, long long slot0 = 0)
{
    const DScene S = wf_scene(S_);
    int&         s_child = sh->child;
    int&         s_sh = sh->sh;
    const int    n = in.cnt[r];
    const size_t base = (size_t)r * B.R;
    const size_t sbase = base * P.n_lights * (SOFT ? B.ns : 1); // the region's shadow queue
    if (threadIdx.x == 0) s_child = 0, s_sh = 0;
    __syncthreads();
    const int n_pad = (n + 63) & ~63; // whole waves iterate together: the ballots need every lane
    for (int j = threadIdx.x; j < n_pad; j += WF_BLOCK) {
        const bool   valid = j < n;
        const size_t i = base + j;
        uint2        meta = make_uint2(0, 0);
        float4       ow = make_float4(0, 0, 0, 0), dn = make_float4(0, 0, 0, 0), hit = make_float4(0, 0, 0, 0);
        if (valid) meta = in.meta[i], ow = in.o_w[i], dn = in.d_ni[i], hit = in.hits[i];
        const int   orig = valid ? __float_as_int(hit.x) : -1;
        const int   k = (int)(meta.y & 0xff);
        bool        shared = (meta.y >> 8) & 1;
        const f3    o = mk3(ow.x, ow.y, ow.z), dst = mk3(dn.x, dn.y, dn.z);
        const float w = ow.w;
        if (valid && k == 0) { // the camera ray fixes the sample's depth and triangle
            B.s_minT[meta.x] = hit.w;
            B.s_tri[meta.x] = orig;
        }
        float      wl = 0.0f;
        WfSurf     s;
        const bool is_hit = valid && orig >= 0;
        if (valid && !is_hit) // raytracing.cpp:74
            wf_acc_add(B, meta.x, mk3(P.bg[0], P.bg[1], P.bg[2]) * w, shared);
        if (is_hit) wf_surface(S, o, dst, hit, s), ++n_shaded;
        // ---- shadow rays: one queue for all lights (a wave of shadow rays then is full whatever
        // the share of each light; rays of one light and one wave of hits stay neighbours) ----
        float shadowMax = 0.0f;
        for (int li = 0; li < P.n_lights; ++li) {
            bool cast = false;
            f3   lightDir = mk3(0, 0, 0);
            if (is_hit) {
                float angle;
                cast = wf_light_setup(s, mk3(P.lights[3 * li], P.lights[3 * li + 1], P.lights[3 * li + 2]),
                                      lightDir, angle, shadowMax);
                // An occluded light can only raise shadowMax to 1 (raytracing.cpp:142-143).  Here
                // shadowMax is its value with every query so far answered "free"; if that is 1
                // already - set by a grazing angle at this light, or carried from an earlier one -
                // the answer cannot change the colour and the query is not traced.  (Soft shadows
                // take the maximum with a sum that rounding may carry past 1: all of them are traced.)
                if (!SOFT && shadowMax == 1.0f) cast = false;
                shadowMax = (shadowMax < 0.0f) ? 0.0f : shadowMax;
            }
            const int pos = wf_append(&s_sh, cast, SOFT ? B.ns : 1);
            if (cast) {
                const f3  so = s.hp + lightDir * 0.001f; // raytracing.cpp:142
                const int tag = (li << 24) | j;
                if (!SOFT) {
                    B.sh_org[sbase + pos] = make_float4(so.x, so.y, so.z, __int_as_float(tag));
                } else { // raytracing.cpp:145-160: ns rays towards jittered points around the light
                    const int slot = P.msaa == 1 ? (int)meta.x : (int)(meta.x / (unsigned)P.msaa);
                    const int sn = P.msaa == 1 ? 0 : (int)(meta.x % (unsigned)P.msaa) + 1;
                    int       px = 0, py = 0;
                    slot_to_pixel(P, slot0 + slot, px, py);
                    const unsigned long long id = ((unsigned long long)py * P.W + px) * 32 + sn;
                    const size_t             q0 = sbase + pos;
                    B.occl[(size_t)li * B.cap + i] = 0;
                    for (int si = 0; si < B.ns; ++si) {
                        float4 dd = make_float4(0, 0, 0, 0);
                        soft_dest(&P.lights[0], &P.soft_x[0], P.soft_n, P.soft_half, P.soft_quarter, P.soft_radius, li, si, id,
                                  (meta.y >> 16) & 0xffu, dd.x, dd.y, dd.z);
                        B.sh_org[q0 + si] = make_float4(so.x, so.y, so.z, __int_as_float(tag));
                        B.sh_dst[q0 + si] = dd;
                    }
                }
            }
        }
        // ---- children, raytracing.cpp:166-271 ----
        int      nchild = 0;
        f3       c_o[2], c_d[2];
        float    c_w[2];
        unsigned c_kind[2] = {0, 0}; // 1 mirror, 2 transmitted, 3 reflected (the soft-shadow generator's key)
        float n2 = 1.0f;
        if (is_hit) {
            wl = w;
            const int newK = k + 1;
            if (newK < 5) {
                const float n1 = (dn.w != 0.0f) ? dn.w : 1.0f;
                n2 = s.has_Ni ? s.Ni : 1.0f;
                if (s.illum == 3) {
                    const f3 refl = s.dir - s.N * (2.0f * dot3(s.N, s.dir));
                    wl = 0.5f * w;
                    c_o[0] = s.hp + refl * 0.001f, c_d[0] = s.hp + refl, c_w[0] = 0.5f * w, c_kind[0] = 1;
                    nchild = 1;
                } else if (s.Ni > 1.0f) {
                    f3 I = s.dir;
                    normalize3(I);
                    const float eta = (n1 != n2) ? n2 : __fdiv_rn(1.0f, n2);
                    const float cosI = dot3(I, s.N);
                    const float sin2T =
                        (float)(((double)eta * (double)eta) * (1.0 - ((double)cosI * (double)cosI)));
                    float rf = sin2T;
                    rf = rf < 0.0f ? 0.0f : rf;
                    rf = rf > 1.0f ? 1.0f : rf;
                    float wLocal = 0.5f * rf;
                    if (sin2T <= 1.0f) {
                        const float f = (float)((double)(eta * cosI) - sqrt((double)(1.0f - sin2T)));
                        const f3    tRay = I * eta + s.N * f;
                        wLocal += 0.1f * (1.0f - rf);
                        const float wT = 0.9f * (1.0f - rf) * w;
                        if (wT != 0.0f) {
                            c_o[nchild] = s.hp + tRay * 0.001f, c_d[nchild] = s.hp + tRay, c_w[nchild] = wT;
                            c_kind[nchild] = 2;
                            ++nchild;
                        }
                    }
                    const float wR = 0.5f * rf * w;
                    if (wR != 0.0f) {
                        const f3 refRay = I - s.N * (2.0f * dot3(I, s.N));
                        c_o[nchild] = s.hp + refRay * 0.001f, c_d[nchild] = s.hp + refRay, c_w[nchild] = wR;
                        c_kind[nchild] = 3;
                        ++nchild;
                    }
                    wl = wLocal * w;
                    shared = shared || nchild > 1;
                }
            }
        }
        if (valid) B.wlocal[i] = wl;
        n_child += nchild;
        for (int c = 0; c < 2; ++c) {
            const bool mine = c < nchild;
            const int  pos = wf_append(&s_child, mine);
            if (mine && pos < B.R) { // cannot exceed R by construction; the guard keeps a bug from writing outside
                const size_t p = base + pos;
                out.o_w[p] = make_float4(c_o[c].x, c_o[c].y, c_o[c].z, c_w[c]);
                out.d_ni[p] = make_float4(c_d[c].x, c_d[c].y, c_d[c].z, n2);
                unsigned my = (unsigned)(k + 1) | ((shared ? 1u : 0u) << 8);
                if (SOFT) my |= ((((meta.y >> 16) & 0x3fu) << 2) | c_kind[c]) << 16; // the generator's key
                out.meta[p] = make_uint2(meta.x, my);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out.cnt[r] = min(s_child, B.R);
    if (threadIdx.x == 0) B.sh_cnt[r] = s_sh;
}

__global__ void __launch_bounds__(WF_BLOCK)
k_wf_shade(DScene S, DParams P, WfBuffers B, int level, DCounters* ctr)
{
    __shared__ WfShared sh;
    unsigned            n_shaded = 0, n_child = 0;
    wf_shade_region(S, P, B, B.rays[level & 1], B.rays[(level + 1) & 1], blockIdx.x, &sh, n_shaded, n_child);
    for (int s = 32; s > 0; s >>= 1) n_shaded += __shfl_down(n_shaded, s), n_child += __shfl_down(n_child, s);
    if ((threadIdx.x & 63) == 0 && (n_shaded | n_child)) {
        atomicAdd(&ctr->shaded_hits, (unsigned long long)n_shaded);
        atomicAdd(&ctr->child_rays, (unsigned long long)n_child);
    }
}

// ---------------------------------------------------------------------------
// resolve: the light loop of raytracing.cpp:110-164 with the shadow queries answered
// ---------------------------------------------------------------------------
template <bool SOFT = false, class ST = DScene, class PT = DParams>
__device__ __forceinline__ void wf_resolve_region(const ST& S_, const PT& P, const WfBuffers& B,
                                                  const WfRays in, int r)
{
    const DScene S = wf_scene(S_);
    const int    n = in.cnt[r];
    const size_t base = (size_t)r * B.R;
    const float  shadowPart = __fdiv_rn(P.ShadowFactor, (float)P.n_lights); // raytracing.cpp:102
    for (int j = threadIdx.x; j < n; j += WF_BLOCK) {
        const size_t i = base + j;
        const float  wl = B.wlocal[i];
        const float4 hit = in.hits[i];
        if (__float_as_int(hit.x) < 0) continue;
        const uint2  meta = in.meta[i];
        const float4 ow = in.o_w[i], dn = in.d_ni[i];
        WfSurf       s;
        wf_surface(S, mk3(ow.x, ow.y, ow.z), mk3(dn.x, dn.y, dn.z), hit, s);
        f3 res;
        if (s.tex < 0) res = s.Kd * P.backgroundlighting; // raytracing.cpp:92-95
        else res = sample_texture(S, s.tex, s.info, s.hp);
        float shadowMax = 0.0f; // carried from light to light (raytracing.cpp:104)
        for (int li = 0; li < P.n_lights; ++li) {
            const f3 lpos = mk3(P.lights[3 * li], P.lights[3 * li + 1], P.lights[3 * li + 2]);
            f3       lightDir = lpos - s.hp;
            normalize3(lightDir);
            float angle = dot3(s.N, lightDir);
            if (angle > 0.0f) res = res + (s.Kd * angle) * P.diffusePower;
            angle = dot3(lightDir, s.N);
            if (angle <= 0.2f) shadowMax = 1.0f;
            else if (angle < 0.5f) {
                angle -= 0.2f;
                shadowMax = 3.33f * (0.3f - angle);
            }
            if (angle > 0.0f) {
                f3 halfway = lightDir + s.viewDir;
                normalize3(halfway);
                angle = dot3(s.N, halfway);
                if (angle > 0.0f) res = res + s.Ks * powf(angle, s.Ns);
                const int occluded = B.occl[(size_t)li * B.cap + i];
                if (!SOFT) {
                    if (occluded) shadowMax = 1.0f;
                } else { // raytracing.cpp:155-157: one float addition per occluded sample
                    float softShadow = 0.0f;
                    for (int c = 0; c < occluded; ++c) softShadow += P.soft_w;
                    shadowMax = (shadowMax < softShadow) ? softShadow : shadowMax;
                }
            }
            shadowMax = (shadowMax < 0.0f) ? 0.0f : shadowMax; // std::max(shadowMax, softShadow) with softShadow >= 0
            const float sm = shadowMax * shadowPart;
            res = res - mk3(sm, sm, sm);
        }
        wf_acc_add(B, meta.x, res * wl, (meta.y >> 8) & 1);
    }
}

__global__ void __launch_bounds__(WF_BLOCK)
k_wf_resolve(DScene S, DParams P, WfBuffers B, int level)
{
    wf_resolve_region(S, P, B, B.rays[level & 1], blockIdx.x);
}

// ---------------------------------------------------------------------------
// camera rays
// ---------------------------------------------------------------------------
// Block r: centre rays of pixel slots slot0 + [r * S, (r + 1) * S) that lie inside
// the frame, appended to region r of queue 0.  meta.x = slot - slot0.
template <class PT>
__device__ __forceinline__ void wf_gen_region(const PT& P, const WfBuffers& B, const WfRays q,
                                              long long slot0, int n, DCounters* ctr, int r, WfShared* sh)
{
    int&         s_n = sh->n;
    const size_t base = (size_t)r * B.R;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int s_pad = (B.S + 63) & ~63;
    for (int j = threadIdx.x; j < s_pad; j += WF_BLOCK) {
        const int i = j < B.S ? r * B.S + j : n;
        int       x = 0, y = 0;
        const bool valid = i < n && slot_to_pixel(P, slot0 + i, x, y);
        const int  pos = wf_append(&s_n, valid);
        if (valid) {
            f3 o, d;
            pixel_ray(P, (double)x, (double)y, o, d);
            q.o_w[base + pos] = make_float4(o.x, o.y, o.z, 1.0f);
            q.d_ni[base + pos] = make_float4(d.x, d.y, d.z, 1.0f); // prev_Ni = 1.0f (main.cpp:528)
            q.meta[base + pos] = make_uint2((unsigned)i, 0u);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        q.cnt[r] = s_n;
        atomicAdd(&ctr->camera_rays, (unsigned long long)s_n);
    }
}

__global__ void __launch_bounds__(WF_BLOCK)
k_wf_gen_centre(DParams P, WfBuffers B, long long slot0, int n, DCounters* ctr)
{
    __shared__ WfShared sh;
    wf_gen_region(P, B, B.rays[0], slot0, n, ctr, blockIdx.x, &sh);
}

// MSAA > 1, after the centre rays are traced: misses are final pixels; every hit
// pixel queues its `msaa` sample rays (main.cpp:445-507) into region r of queue 1.
// Sample index = (slot - slot0) * msaa + s.
template <class PT>
__device__ __forceinline__ void wf_msaa_region(const PT& P, const WfBuffers& B, const WfRays in,
                                               const WfRays q, long long slot0, float* rgb, float* depth,
                                               int* tri, unsigned char* centre_hit, DCounters* ctr, int r,
                                               WfShared* sh)
{
    int&         s_n = sh->n;
    const int    n = in.cnt[r];
    const size_t base = (size_t)r * B.R;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int n_pad = (n + 63) & ~63;
    for (int j = threadIdx.x; j < n_pad; j += WF_BLOCK) {
        const bool valid = j < n;
        bool       got = false;
        int        x = 0, y = 0, slot = 0;
        if (valid) {
            slot = (int)in.meta[base + j].x;
            const float4 hit = in.hits[base + j];
            got = __float_as_int(hit.x) >= 0;
            slot_to_pixel(P, slot0 + slot, x, y);
            const size_t p = (size_t)y * P.W + x;
            if (tri) tri[p] = __float_as_int(hit.x);
            centre_hit[slot] = got ? 1 : 0;
            if (!got) {
                rgb[3 * p] = wf_clamp01(P.bg[0]);
                rgb[3 * p + 1] = wf_clamp01(P.bg[1]);
                rgb[3 * p + 2] = wf_clamp01(P.bg[2]);
                depth[p] = RT_FLT_MAX;
            }
        }
        const int pos = wf_append(&s_n, got, P.msaa);
        if (got && pos + P.msaa <= B.R) {
            for (int s = 0; s < P.msaa; ++s) {
                const float ox = P.msaa == 4 ? c_msaa4[s][0] : c_msaa16[s][0];
                const float oy = P.msaa == 4 ? c_msaa4[s][1] : c_msaa16[s][1];
                f3          o, d;
                pixel_ray(P, (double)x + (double)ox, (double)y + (double)oy, o, d);
                const size_t p = base + pos + s;
                q.o_w[p] = make_float4(o.x, o.y, o.z, 1.0f);
                q.d_ni[p] = make_float4(d.x, d.y, d.z, 1.0f);
                q.meta[p] = make_uint2((unsigned)(slot * P.msaa + s), 0u);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        q.cnt[r] = min(s_n, B.R);
        atomicAdd(&ctr->camera_rays, (unsigned long long)s_n);
    }
}

__global__ void __launch_bounds__(WF_BLOCK)
k_wf_msaa_emit(DParams P, WfBuffers B, long long slot0, float* rgb, float* depth, int* tri,
               unsigned char* centre_hit, DCounters* ctr)
{
    __shared__ WfShared sh;
    wf_msaa_region(P, B, B.rays[0], B.rays[1], slot0, rgb, depth, tri, centre_hit, ctr, blockIdx.x, &sh);
}

// Pixels from sample sums.  MSAA == 1: sample index = slot - slot0.  MSAA > 1:
// ordered float sum of the samples / msaa, depth = mean of minT over the samples
// that hit (main.cpp:449-507); only pixels whose centre ray hit.
template <class PT>
__device__ __forceinline__ void wf_finalize_slot(const PT& P, const WfBuffers& B, long long slot0, int i,
                                                 const unsigned char* __restrict__ centre_hit, float* rgb,
                                                 float* depth, int* tri)
{
    int x, y;
    if (!slot_to_pixel(P, slot0 + i, x, y)) return;
    const size_t p = (size_t)y * P.W + x;
    const size_t ns = (size_t)B.n_samples;
    if (P.msaa == 1) {
        const float r = (float)((double)B.acc[i] / WF_FIX_SCALE);
        const float g = (float)((double)B.acc[ns + i] / WF_FIX_SCALE);
        const float b = (float)((double)B.acc[2 * ns + i] / WF_FIX_SCALE);
        rgb[3 * p] = wf_clamp01(r);
        rgb[3 * p + 1] = wf_clamp01(g);
        rgb[3 * p + 2] = wf_clamp01(b);
        depth[p] = B.s_minT[i];
        if (tri) tri[p] = B.s_tri[i];
        return;
    }
    if (!centre_hit[i]) return;
    float r = 0, g = 0, b = 0, dsum = 0;
    int   hits = 0;
    for (int s = 0; s < P.msaa; ++s) {
        const size_t k = (size_t)i * P.msaa + s;
        const float  sr = (float)((double)B.acc[k] / WF_FIX_SCALE);
        const float  sg = (float)((double)B.acc[ns + k] / WF_FIX_SCALE);
        const float  sb = (float)((double)B.acc[2 * ns + k] / WF_FIX_SCALE);
        if (s == 0) r = sr, g = sg, b = sb;
        else r += sr, g += sg, b += sb;
        const float mt = B.s_minT[k];
        if (mt != RT_FLT_MAX) dsum += mt, ++hits;
    }
    const float nn = (float)P.msaa;
    rgb[3 * p] = wf_clamp01(__fdiv_rn(r, nn));
    rgb[3 * p + 1] = wf_clamp01(__fdiv_rn(g, nn));
    rgb[3 * p + 2] = wf_clamp01(__fdiv_rn(b, nn));
    depth[p] = hits > 0 ? __fdiv_rn(dsum, (float)hits) : RT_FLT_MAX;
}

__global__ void __launch_bounds__(WF_BLOCK)
k_wf_finalize(DParams P, WfBuffers B, long long slot0, int n, const unsigned char* __restrict__ centre_hit,
              float* rgb, float* depth, int* tri)
{
    const int i = blockIdx.x * WF_BLOCK + threadIdx.x;
    if (i < n) wf_finalize_slot(P, B, slot0, i, centre_hit, rgb, depth, tri);
}

// ---------------------------------------------------------------------------
// the whole frame part in one persistent launch
// ---------------------------------------------------------------------------
// Measured with a -DWF_TIMELINE build (tools/timeline.py): the median region of a 4K MSAA4
// frame of final.obj takes 0.25 ms under load, the slowest 3 ms; the blocks stay busy until 0.9 ms
// before the end of a 11.9 ms launch (1.0 of 2.6 ms for an eighth of the frame).  Tried against
// that tail and not kept: raising the wave priority (s_setprio) with the level of the recursion
// (no change at all); postponing regions whose first trace hits no material that spawns rays, so
// that the others start earlier (13.2 vs 12.0 ms, 2.73 vs 2.56 ms: in this scene nearly every
// region touches a mirror, the bookkeeping only costs).
// What ends the launch is the longest region, not the order: 64 pixels of mirror facing
// mirror run five levels of long rays, 1.5-2 ms however the rest is scheduled.  Tried and not
// kept: regions handed out by descending cost estimate from a first pass (the second launch
// costs more than the shorter tail gains); units of 128/64/32/16 pixel slots handed out
// coarse first (the chain of dependent stages does not get shorter with fewer rays, and
// regions below 64 slots leave most of a block's waves empty: 16 slots 5.9 ms vs 2.6 ms for
// an eighth of a 4K frame).
// A block takes a region (one atomic), runs every stage of its pixels - camera rays,
// MSAA gating, five levels of trace / shade / shadow trace / resolve, pixel resolve - and
// takes the next region.  Stages of a region are separated by block barriers only; there
// is no device-wide synchronisation and no launch boundary inside the frame, so a CU
// always has other blocks' stages to run while one block waits for its slowest ray, and
// the bandwidth-bound stages of some blocks overlap the arithmetic-bound traces of others.
template <bool ANY, bool COUNT>
__device__ __forceinline__ void wf_trace_region(const DScene& S, const float4* __restrict__ o_w,
                                                const float4* __restrict__ d_ni, f3 light, int n,
                                                size_t base, float4* hits, unsigned char* occl,
                                                const WfStack& stk, unsigned& n_tests, unsigned& n_nodes,
                                                unsigned& n_rays)
{
    for (int j = threadIdx.x; j < n; j += WF_BLOCK) {
        const size_t i = base + j;
        const float4 ow = o_w[i];
        const f3     o = mk3(ow.x, ow.y, ow.z);
        f3           dst = light;
        if (!ANY) {
            const float4 dn = d_ni[i];
            dst = mk3(dn.x, dn.y, dn.z);
        }
        HitRec h;
        ++n_rays;
        const bool got = wf_traverse<ANY, COUNT>(S, stk, o, dst - o, h, n_tests, n_nodes);
        if (ANY) occl[__float_as_int(ow.w)] = got ? 1 : 0;
        else hits[i] = make_float4(__int_as_float(got ? h.slot : -1), h.a, h.b, h.t);
    }
}

// Shadow queries of one level and closest-hit queries of the next in one pass: both sets
// of rays come out of the same shade stage and neither needs the other's answers, so a
// region's chain of dependent stages is one trace per level instead of two.  The index
// space is [closest rays][shadow rays], each segment padded to whole waves so a wave runs one
// kind of walk; the closest-hit rays (the longer walks) start first.
template <bool COUNT, bool SOFT = false, class PT = DParams>
__device__ __forceinline__ void wf_trace_merged(const DScene& S, const PT& P, const WfBuffers& B,
                                                const WfRays next, int n_next, int r, const WfStack& stk,
                                                unsigned& n_tests, unsigned& n_nodes, unsigned& n_closest,
                                                unsigned& n_shadow)
{
    const size_t base = (size_t)r * B.R;
    const int    seg_c = (n_next + 63) & ~63;
    const int    n_sh = B.sh_cnt[r];
    const int    total = seg_c + n_sh;
    const size_t sbase = base * P.n_lights * (SOFT ? B.ns : 1);
    for (int i = threadIdx.x; i < total; i += WF_BLOCK) {
        if (i < seg_c) {
            if (i >= n_next) continue;
            const size_t k = base + i;
            const float4 ow = next.o_w[k], dn = next.d_ni[k];
            const f3     o = mk3(ow.x, ow.y, ow.z);
            HitRec       h;
            ++n_closest;
            const bool got = wf_traverse<false, COUNT>(S, stk, o, mk3(dn.x, dn.y, dn.z) - o, h, n_tests, n_nodes);
            next.hits[k] = make_float4(__int_as_float(got ? h.slot : -1), h.a, h.b, h.t);
        } else {
            const size_t qi = sbase + (i - seg_c);
            const float4 ow = B.sh_org[qi];
            const f3     o = mk3(ow.x, ow.y, ow.z);
            const int    li = (int)((unsigned)__float_as_int(ow.w) >> 24);
            const size_t hit_slot = base + (size_t)(__float_as_int(ow.w) & 0xffffff);
            f3           l = mk3(P.lights[3 * li], P.lights[3 * li + 1], P.lights[3 * li + 2]);
            if (SOFT) {
                const float4 dd = B.sh_dst[qi];
                l = mk3(dd.x, dd.y, dd.z);
            }
            HitRec h;
            ++n_shadow;
            const bool got = wf_traverse<true, COUNT>(S, stk, o, l - o, h, n_tests, n_nodes);
            if (!SOFT) {
                B.occl[(size_t)li * B.cap + hit_slot] = got ? 1 : 0;
            } else if (got) { // count in the hit's byte (at most 216 samples): add into its 32-bit word
                const size_t b = (size_t)li * B.cap + hit_slot;
                atomicAdd((unsigned*)(B.occl + (b & ~(size_t)3)), 1u << (8 * (unsigned)(b & 3)));
            }
        }
    }
}

// Compiled for 7 waves per SIMD (72 VGPRs; what does not fit spills, outside the traversal
// loops).  Measured per 4K frame: 25.6 / 23.6 / 23.3 / 22.6 / 25.4 ms for 4 / 5 / 6 / 7 / 8.
#ifndef WF_PIPE_WAVES
#define WF_PIPE_WAVES 7
#endif
#ifdef WF_TIMELINE
// Diagnostic builds only (-DWF_TIMELINE, read by tools/timeline.py): 16 time stamps of the 100 MHz
// wall clock per region - start, end of each stage - and the block that ran it.
#define WF_TL_REGIONS 140000
__device__ unsigned long long g_timeline[WF_TL_REGIONS * 16];
#define WF_TL(slot) do { if (threadIdx.x == 0 && r < WF_TL_REGIONS) g_timeline[(size_t)r * 16 + (slot)] = wall_clock64(); } while (0)
#else
#define WF_TL(slot) do { } while (0)
#endif
// Everything the launch needs, as ONE kernel argument.  The kernel never names the argument:
// each stage reads the fields it needs from the kernarg segment (constant address space) through
// a pointer that is laundered in place (wf_args), so the loads cannot be hoisted to the top of
// the launch and the values live in scalar registers only while their stage runs.  With the
// three structs as ordinary by-value arguments every pointer in them stayed live across the
// whole persistent loop (> 100 scalar registers); what did not fit was kept in vector lanes and
// fetched back with v_readlane - from the moment two more fields were added to WfBuffers also
// inside the node and triangle loops (16 reloads per iteration, 13.0 vs 12.0 ms per 4K frame).
struct WfLaunch {
    DScene         S;
    DParams        P;
    WfBuffers      B;
    long long      slot0;
    int            n, n_regions; // pixel slots of the frame part, regions of B.S slots they make
    float*         rgb;
    float*         depth;
    int*           tri;
    unsigned char* centre_hit;
    DCounters*     ctr;
};
typedef const __attribute__((address_space(4))) WfLaunch* WfLaunchC;

// A struct out of the kernarg segment; unused fields vanish, pointer fields stay typed loads from
// constant memory, which is what lets the compiler treat them as global pointers.  (The copy
// across address spaces only exists in the device pass; the host pass never runs this.)
template <class T>
__device__ __forceinline__ T wf_ld(const __attribute__((address_space(4))) T& src)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return src;
#else
    (void)src;
    return T();
#endif
}

__device__ __forceinline__ DScene wf_scene(const DScene& s) { return s; }
__device__ __forceinline__ DScene wf_scene(const __attribute__((address_space(4))) DScene& s) { return wf_ld(s); }

__device__ __forceinline__ WfLaunchC wf_args()
{
    WfLaunchC p = (WfLaunchC)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

template <bool COUNT, bool SOFT>
__global__ void __launch_bounds__(WF_BLOCK, WF_PIPE_WAVES)
k_wf_pipeline(WfLaunch)
{
    extern __shared__ __align__(16) unsigned char s_mem[];
    __shared__ WfShared sh;
    unsigned            n_tests = 0, n_nodes = 0, n_closest = 0, n_shadow = 0, n_shaded = 0, n_child = 0;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) sh.region = atomicAdd(&wf_args()->B.ctr->head_region, 1);
        __syncthreads();
        const int r = sh.region;
        if (r >= wf_args()->n_regions) break;
        // The workspace has one set of queues, colour sums and flags per block, not per region: the
        // block runs region r as "region qr" of the buffers, with the first pixel slot and the slot
        // count shifted to match.  Consecutive regions of a block reuse the same few hundred KB,
        // which stay in the L2 of its XCD instead of streaming through HBM (11.5 -> 11.1 ms per 4K
        // frame), and the workspace does not grow with the frame.
        const int qr = blockIdx.x;
#define WF_SLOT0(A) ((A)->slot0 + (long long)(r - qr) * (A)->B.S)
#define WF_N(A) ((A)->n + (qr - r) * (A)->B.S)
        WF_TL(0);
#ifdef WF_TIMELINE
        if (threadIdx.x == 0 && r < WF_TL_REGIONS) g_timeline[(size_t)r * 16 + 15] = blockIdx.x;
#endif
        {
            const WfLaunchC A = wf_args();
            const WfBuffers B = wf_ld(A->B);
            const size_t    ns = (size_t)B.n_samples;
            const int       msaa = A->P.msaa;
            for (int j = threadIdx.x; j < B.S * msaa; j += WF_BLOCK) { // this region's colour sums
                const size_t k = (size_t)qr * B.S * msaa + j;
                B.acc[k] = 0, B.acc[ns + k] = 0, B.acc[2 * ns + k] = 0;
            }
            wf_gen_region(A->P, B, wf_ld(A->B.rays[0]), WF_SLOT0(A), WF_N(A), A->ctr, qr, &sh);
        }
        __syncthreads();
        {
            const WfLaunchC A = wf_args();
            const DScene    S = wf_ld(A->S);
            const WfRays    q0 = wf_ld(A->B.rays[0]);
            wf_trace_region<false, COUNT>(S, q0.o_w, q0.d_ni, mk3(0, 0, 0), q0.cnt[qr], (size_t)qr * A->B.R, q0.hits,
                                          nullptr, wf_stack(S, s_mem), n_tests, n_nodes, n_closest);
        }
        __syncthreads();
        WF_TL(1);
        int q = 0;
        if (wf_args()->P.msaa > 1) {
            {
                const WfLaunchC A = wf_args();
                const WfBuffers B = wf_ld(A->B);
                wf_msaa_region(A->P, B, wf_ld(A->B.rays[0]), wf_ld(A->B.rays[1]), WF_SLOT0(A), A->rgb, A->depth, A->tri, A->centre_hit,
                               A->ctr, qr, &sh);
            }
            __syncthreads();
            q = 1;
            {
                const WfLaunchC A = wf_args();
                const DScene    S = wf_ld(A->S);
                const WfRays    q1 = wf_ld(A->B.rays[1]);
                wf_trace_region<false, COUNT>(S, q1.o_w, q1.d_ni, mk3(0, 0, 0), q1.cnt[qr], (size_t)qr * A->B.R,
                                              q1.hits, nullptr, wf_stack(S, s_mem), n_tests, n_nodes, n_closest);
            }
            __syncthreads();
        }
        WF_TL(2);
        // level by level: shade(l) -> [shadow(l) + closest(l + 1)] -> resolve(l)
        if (wf_args()->B.rays[q].cnt[qr] > 0) {
            {
                const WfLaunchC A = wf_args();
                const WfBuffers B = wf_ld(A->B);
                wf_shade_region<SOFT>(A->S, A->P, B, wf_ld(A->B.rays[q]), wf_ld(A->B.rays[q ^ 1]), qr, &sh, n_shaded, n_child,
                                      WF_SLOT0(A));
            }
            __syncthreads();
            for (int level = 0; level < 5; ++level, q ^= 1) {
                int n_out;
                {
                    const WfLaunchC A = wf_args();
                    const DScene    S = wf_ld(A->S);
                    const WfBuffers B = wf_ld(A->B);
                    const WfRays    out = wf_ld(A->B.rays[q ^ 1]);
                    n_out = out.cnt[qr];
                    wf_trace_merged<COUNT, SOFT>(S, A->P, B, out, n_out, qr, wf_stack(S, s_mem), n_tests, n_nodes,
                                                 n_closest, n_shadow);
                }
                __syncthreads();
                WF_TL(3 + 2 * level);
                {
                    const WfLaunchC A = wf_args();
                    const WfBuffers B = wf_ld(A->B);
                    wf_resolve_region<SOFT>(A->S, A->P, B, wf_ld(A->B.rays[q]), qr);
                }
                if (n_out == 0) break;
                __syncthreads(); // resolve still reads the queue the next shade writes its children to
                {
                    const WfLaunchC A = wf_args();
                    const WfBuffers B = wf_ld(A->B);
                    wf_shade_region<SOFT>(A->S, A->P, B, wf_ld(A->B.rays[q ^ 1]), wf_ld(A->B.rays[q]), qr, &sh, n_shaded,
                                          n_child, WF_SLOT0(A));
                }
                __syncthreads();
                WF_TL(4 + 2 * level);
            }
            __syncthreads();
        }
        WF_TL(13);
        {
            const WfLaunchC A = wf_args();
            const WfBuffers B = wf_ld(A->B);
            for (int j = threadIdx.x; j < B.S; j += WF_BLOCK) {
                const int i = qr * B.S + j;
                if (i < WF_N(A)) wf_finalize_slot(A->P, B, WF_SLOT0(A), i, A->centre_hit, A->rgb, A->depth, A->tri);
            }
        }
        WF_TL(14);
    }
#undef WF_SLOT0
#undef WF_N
    DCounters* ctr = wf_args()->ctr;
    for (int s = 32; s > 0; s >>= 1) {
        n_tests += __shfl_down(n_tests, s);
        n_nodes += __shfl_down(n_nodes, s);
        n_closest += __shfl_down(n_closest, s);
        n_shadow += __shfl_down(n_shadow, s);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&ctr->tri_tests, (unsigned long long)n_tests);
        atomicAdd(&ctr->node_visits, (unsigned long long)n_nodes);
        atomicAdd(&ctr->rays_closest, (unsigned long long)n_closest);
        atomicAdd(&ctr->rays_shadow, (unsigned long long)n_shadow);
    }
    if (COUNT) { // the timed variant leaves these two out
        for (int s = 32; s > 0; s >>= 1) n_shaded += __shfl_down(n_shaded, s), n_child += __shfl_down(n_child, s);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&ctr->shaded_hits, (unsigned long long)n_shaded);
            atomicAdd(&ctr->child_rays, (unsigned long long)n_child);
        }
    }
}
