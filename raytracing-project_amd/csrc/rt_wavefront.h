// rt_wavefront.h - the frame renderer as a wavefront of ray queues.
//
// The reference shades a camera sample by recursion (raytracing.cpp:61-274):
// closest hit, one shadow query per light, then up to two child rays, five
// levels deep.  Here every level of that recursion is one pass over a queue of
// rays.  A ray is traced and shaded by the same lane in one go:
//
//   closest hit  ->  hit geometry, the local colour with every shadow query assumed
//                    "free", the shadow rays (appended to the level's shadow queue) and
//                    the child rays (appended to the next level's queue, compacted with
//                    wave ballots)
//
// and the shadow rays of level l are traced in the same pass as the closest hits of
// level l + 1.  A hit whose shadow queries cannot matter is final at once; any other
// leaves a 36-byte record (colour so far, weight, what each light does to the shadow
// term) and a count of pending queries.  Every shadow ray reports its answer with one
// atomic on that count; the ray that brings it to zero has all answers in the returned
// word and adds the hit's colour to its camera sample.  There is no hit buffer, no
// second evaluation of the hit geometry and no pass of its own for the light loop.
//
// Queues are cut into regions: the S (64 to 512) pixel slots of a region, their camera
// rays, then whatever those rays spawn, level after level.  A region cannot overflow (its
// size is the largest number of rays its pixels can have in flight) and appending needs
// an LDS counter only.  Two drivers exist:
//   k_wf_pipeline   one persistent launch; a block takes a region and runs all its
//                   stages, separated by block barriers only (the product path)
//   k_wf_gen_centre / k_wf_trace / k_wf_msaa_emit / k_wf_shade / k_wf_finalize
//                   the same per-ray code as one launch per stage over all regions
//                   (RT_WF_MULTILAUNCH=1: cross-check and single-stage profiling)
//
// A camera sample's colour is the sum over its ray tree of (path weight x
// local colour); the recursion is linear in the child colours
// (raytracing.cpp:191, 240, 259, 268).  The sum is accumulated in 64-bit fixed
// point (2^-32 units), so it does not depend on the order in which rays
// finish: frames are bit-reproducible and independent of the tile partition.
#pragma once
#include "rt_device.h"

#pragma clang fp contract(off)

// (Blocks of 128 threads - half as many waves held at every stage barrier, twice as many blocks - take 6.9 ms per
// 4K MSAA4 frame of final.obj against 6.2 ms, blocks of 512 threads 7.3 ms: -DWF_BLOCK=...)
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
    unsigned long long cost_sum;     // k_wf_pipeline: sum of the durations its regions took (k_wf_order_*)
};

// Block-local state of the producer stages (LDS).  Levels rotate through three counters
// each: a stage reads its inputs' counts, appends with the counters of its outputs and
// clears those of the stage after it.
struct WfShared {
    int child[3]; // rays of level l: child[l % 3]
    int shd[3];   // shadow rays of level l: shd[l % 3]
    int n_hit;    // k_wf_pipeline, MSAA > 1: pixels whose centre ray hit
    int n_cam;    // camera rays generated (statistics)
    int n_closest, n_shadow; // rays traced (statistics; counted per wave here, not in a register per lane)
    int region;   // k_wf_pipeline: the region the block works on
    int batch[2]; // k_wf_pipeline: next 64 rays of the stage to hand to a wave (stages alternate between the two)
#ifdef WF_SECT
    unsigned long long sect[10]; // cycles per part of a stage, summed over the block's waves (flushed when the block ends)
#endif
};

// One level's rays (structure of arrays); region r occupies [r * R, (r + 1) * R).
// meta: bits 0-15 the sample (numbered inside the region), 16-18 the level k,
// 19-26 the route from the camera ray (the soft-shadow generator's key).
// Three arrays of twelve bytes per ray: what the walk needs (origin, destination) apart from what only the shading
// needs afterwards (weight, prev_Ni, meta) - the shading fetches its part after the walk rather than carry it
// through, and as the fourth words of the first two arrays plus a third array that fetch touched eighteen cache
// lines per wave, most of them gone from the L2 by then; as one array it touches six.
struct WfVec3 {
    float x, y, z;
};
struct WfRayAux {
    float    w;  // path weight
    float    ni; // prev_Ni
    unsigned meta;
};
struct WfRays {
    WfVec3*   o;   // origin
    WfVec3*   d;   // destination
    WfRayAux* aux;
    int*      cnt; // [G] rays in each region
};

__device__ __forceinline__ unsigned wf_meta(unsigned sample, unsigned k, unsigned route)
{
    return sample | (k << 16) | ((route & 0xffu) << 19);
}

// A shadow ray's tag (w of its queue entry): bits 0-22 the ray slot of its hit, 23 WF_TAG_SOLE, 24-31 the light.
#define WF_TAG_SOLE 0x800000u
#define WF_TAG_SLOT 0x7fffffu

// Record of a hit that waits for shadow answers; `rec_stride` bytes apart.  After it: per light
// the value the light's angle sets shadowMax to before its query (-1 = leaves it), then, with
// soft shadows, per light the number of occluded samples (one byte each).
struct WfRecord {
    float    cr, cg, cb; // colour with all lights unoccluded, before the shadow term
    float    wl;         // weight of that colour in its camera sample
    unsigned sample;     // of the region
    unsigned pending;    // bits 0-7 queries still out, 8-31 lights 0-23 found occluded; soft shadows: queries only
    unsigned occ_hi[2];  // lights 24-87 found occluded
    // (a hit with a single query, hard shadows: `pending` and occ_hi[0] hold the sum of shadowMax over the lights
    // with that query answered "free" and "occluded", as float bits - see WF_TAG_SOLE)
};

// Buffers that exist once per level parity lie back to back: parity 1 starts one stride
// (the size named in the comment) after parity 0.
struct WfBuffers {
    WfRays         rays[2];   // level l in rays[l & 1]
    float4*        hits;      // [cap] multi-launch path only: bits(triangle or -1), a, b, t
    float4*        sh_org;    // x2 [cap * lights * ns]: shadow ray origin xyz, bits(light << 24 | ray slot in
                              // its region); one queue per region for all lights
    float4*        sh_dst;    // x2 [cap * lights * ns]: soft shadows only, the sample's destination
    int*           sh_cnt;    // x2 [G]
    unsigned char* rec;       // x2 [cap * rec_stride]: per ray slot, the record of a hit that waits for shadow
                              // answers (WfRecord, then one float per light, then the soft-shadow counts)
    int*           prefix;    // multi-launch: [G + 1] for the ray queue, then [G + 1] for the shadow queue
    long long*     acc;       // [3][n_samples] fixed-point colour sums; the single launch keeps them in LDS instead
    float*         s_minT;    // per sample: t of the camera ray's hit (FLT_MAX = miss)
    int*           s_tri;     // per sample: triangle of the camera ray's hit (null when no caller reads it: MSAA > 1
                              // takes the triangle of the pixel-centre ray, or no triangle plane is asked for)
    int2*          hit_list;  // k_wf_pipeline, MSAA > 1: [G * S] pixel slots whose centre ray hit: slot, x | y << 16
    WfCounters*    ctr;
    int            cap;       // slots of every queue = G * R
    int            G, R;      // regions, ray slots per region
    int            S;         // pixel slots per region
    int            n_samples; // samples the buffers hold = G * S * msaa
    int            ns;        // shadow rays per hit and light: 1, or soft_n^3 with soft shadows
    int            rec_stride; // bytes per record: sizeof(WfRecord) + 4 * lights (+ 4 * ceil(lights / 4)), in 16s
};

// ---------------------------------------------------------------------------
// work lists and the persistent trace kernels
// ---------------------------------------------------------------------------
// Work-item prefix sums: item counts ceil(cnt / 64) per region, exclusive scan.
// Block 0: the G regions of ray queue `q`.  Block 1: the G regions of the shadow queue of parity `par`.
__global__ void __launch_bounds__(1024) k_wf_worklist(WfBuffers B, int q, int par)
{
    __shared__ int s_part[1024];
    const int      n = B.G;
    const int*     cnt = blockIdx.x == 0 ? B.rays[q].cnt : B.sh_cnt + (size_t)par * B.G;
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

// Per-thread traversal stack.  An entry is the far child still to be visited (node index) and the
// ray parameter at which the ray enters it, eight bytes written and read as one.  Entry e of a
// thread sits at index e * WF_BLOCK of the LDS array while e < cap, else at (e - cap) * WF_BLOCK of
// the spill array in HBM.  Rays rarely have more than a handful of far children pending, so a short
// LDS part serves nearly all pushes and pops while the tree's full depth stays available; LDS per
// block, and with it the number of resident waves, does not grow with the depth of the tree.
// (Until the node index took 32 bits the entry was a 16-bit node beside a float in two arrays: two
// LDS instructions and two addresses per push, and no scene beyond 65535 nodes on this path.
// The far child's record itself on the stack, twelve bytes per entry, so that a pop need not fetch it - one load
// less in the chain of every pop: 6.32 ms per 4K MSAA4 frame of final.obj against 6.00 ms with the LDS that is
// there, and 1.46 against 1.43 ms for a 1080p frame whose five entries all fit either way: the push has to wait
// for the children's records, which the push of an index does not.)
typedef unsigned __attribute__((ext_vector_type(2))) wf_u2; // (a plain vector type: assignable across address spaces)
typedef __attribute__((address_space(3))) wf_u2* WfLdsEntries; // typed as LDS: ds_read_b64 / ds_write_b64, not flat
struct WfStack {
    WfLdsEntries lds;
    uint2*       spill;
    int          cap;
    WfLdsEntries top;       // k_wf_pipeline: a copy of the first top_bytes of the node array in LDS
    unsigned     top_bytes; // (0: none)
    unsigned     top_pairs; // a pair of records that starts before this offset lies in the copy (0: none does)
};

// LDS layout per block: uint2 entry[cap][WF_BLOCK]
#define WF_STACK_ENTRY_BYTES 8

__device__ __forceinline__ WfStack wf_stack(const DScene& S, unsigned char* s_mem)
{
    WfStack k;
    k.cap = S.stk_cap;
    k.lds = (WfLdsEntries)s_mem + threadIdx.x;
    const size_t deep = (size_t)(S.stack_entries - S.stk_cap) * WF_BLOCK;
    k.spill = S.stk_spill + (size_t)blockIdx.x * deep + threadIdx.x;
    k.top = (WfLdsEntries)s_mem, k.top_bytes = 0, k.top_pairs = 0;
    return k;
}

// (The two branches address different address spaces on purpose: written against one generic pointer the
// compiler joins them into flat stores and loads, which go through the LDS and the vector-memory path both.)
__device__ __forceinline__ void wf_stk_put(const WfStack& k, int e, int node, float t)
{
    const uint2 v = make_uint2((unsigned)node, __float_as_uint(t));
    if (e < k.cap) k.lds[e * WF_BLOCK] = wf_u2{v.x, v.y};
    else k.spill[(size_t)(e - k.cap) * WF_BLOCK] = v;
}

// The entry on top of a stack of sp >= 1 entries and the entry t of the one below it (3.4e38, just short of FLT_MAX, if there is none):
// what a pop needs, fetched together - the popped subtree ends where the entry below begins.
__device__ __forceinline__ void wf_stk_top2(const WfStack& k, int sp, uint2& top, float& t_below)
{
    if (sp <= k.cap) { // both in LDS: one instruction (ds_read2st64_b64)
        const int          lo = sp >= 2 ? sp - 2 : 0; // with a single entry the pair read is entries 0 and 1
        const WfLdsEntries p = k.lds + lo * WF_BLOCK;
        const wf_u2        a = p[0], b = p[WF_BLOCK];
        top = sp >= 2 ? make_uint2(b.x, b.y) : make_uint2(a.x, a.y);
        t_below = sp >= 2 ? __uint_as_float(a.y) : 3.4e38f;
    } else {
        const int e = sp - 1;
        top = k.spill[(size_t)(e - k.cap) * WF_BLOCK];
        t_below = __uint_as_float(e - 1 < k.cap ? ((wf_u2)k.lds[(e - 1) * WF_BLOCK]).y : k.spill[(size_t)(e - 1 - k.cap) * WF_BLOCK].y); // (cap >= 1: e >= 1)
    }
}

// Closest-hit (ANY = false) or any-hit (ANY = true) walk of one ray over the compact node
// array.  Inner nodes carry no triangles (they were handed down to the leaves).  The
// traversal arithmetic (plane distances) is free to use fused multiply-add: it only
// decides which cells are visited, with a safety margin; which triangle is hit, and where,
// is decided by the exact test (tri_test_line) alone.  The result equals the reference's leaf walk
// (raytracing.cpp:281-351): the smallest t over all triangles, ties to the triangle that
// walk would have tested last.  COUNT = false leaves the per-node and per-test statistics
// out of the loops (two VALU instructions per step that only RtStats reads).
// A leaf's triangles pass three stages, each fetched only for what the one before let through (final.obj
// 4K MSAA4, per frame): the plane heads, two at a time (1.63 G tests, 18 M wave trips); the edge record of
// the quarter whose plane the ray crosses inside the cell (12 M wave trips); the exact test for what is left
// (2.7 M wave trips, nearly all with a hit).  A 16-byte plane record in front of an exact test that already
// returned early was tried before the heads existed and lost to its own load (12.6 vs 12.05 ms); it pays
// once it replaces the first part of the slot instead of preceding it.
// Tried and not kept: fetching the node through the scalar cache while all lanes of a wave
// sit at the same node (13.0 vs 12.0 ms per 4K frame: the uniformity test and the wait for
// the scalar load cost more than the per-lane axis selects they save); skipping the re-test
// of the triangle that already is the closest hit (3.5 % of the tests, no gain).
// Lanes at work (a -DWF_DIAG build counts them): 94-99 % in the node loop, 69-75 % in the triangle loop - a
// wave tests as long as its fullest leaf.  Stepping over empty leaves inside the node loop (pop
// and descend again without leaving it) raised the triangle loop to 77 % and cut its trips by
// a tenth, but the heavier loop body ran 30 % more trips at 72 %: 11.1 vs 9.4 ms.
// A box around each leaf's triangles, tested against the part of the ray inside the leaf's cell
// before the triangles are: skips 47 % of the visits to leaves with triangles and 30 % of the
// tests (11.2 -> 7.8 per ray), all results unchanged - and the frame takes 9.2-9.4 ms as before: the
// tests it saves are the cheapest ones, and the box is one more dependent load per leaf.
// The exact test without its early returns directly behind the plane heads (no edge records): 7.70 vs 7.46 ms.
// What the launch is sensitive to, measured by padding (final.obj 4K MSAA4, 6.28 ms): four / eight more vector
// instructions per node step (+4 / +9 % of all) cost 2.2 / 3.2 %, twelve more per edge record nothing, one more
// 8-byte load per node step (always a hit in the vector L1, +21 % vector memory instructions) 10 %.  The node
// loop - every lane at work, one load per step and nothing to overlap it with but the step's own arithmetic - is
// where time goes; the stages behind the plane heads run in its shadow (eight more vector instructions per trip
// of the triangle loop: nothing; one more load per trip, from another array: 1.6 %).  The chosen child's record
// alone, fetched after the plane has decided (8 bytes per step instead of 16, the decision no longer hidden
// behind the load): 6.10 against 6.14 ms, no difference.  The exact record asked for together with the edge record,
// so that the one candidate in four that goes on does not wait for a second load: 6.58 against 5.98 ms - three
// more loads per candidate cost more than the wait they save.  With 4 GB less traffic per frame behind it (5.71 ms)
// the launch answers to vector instructions again: eight / sixteen independent ones more per node step cost 4.4 / 6.5 %.  Accordingly: the edge records in packed
// pairs (-2.4 % vector instructions) and edge records of two loads instead of three (unscaled gradients with one
// error bound per scene) changed no frame time, the latter lost 7 % on fin.obj, whose triangles differ most in size.
#ifdef WF_DIAG
// Diagnostic builds only (-DWF_DIAG): lane slots (wave trips x 64) of the node loop and of the triangle
// loop of the counted launch; rt_api.hip prints them against the lanes that did work (final.obj 4K MSAA4:
// 94 % in the node loop, 69 % in the triangle loop).
__device__ unsigned long long g_diag[16]; // [4..7]: shadow rays traced / occluded, their triangle tests / node steps;
                                         // [8] wave trips (per head) in which a lane passes the plane head, [9] such lanes,
                                         // [10] visits of leaves with triangles by waves, [11] trips of the triangle loop they make,
                                         // [12] wave trips of the exact test with a hit, [13] such lanes, [14] wave trips past the edge records, [15] such lanes
#endif
#ifdef WF_SECT
// Diagnostic builds only (-DWF_SECT): core-clock cycles the waves of the counted launch spend in each part of a
// stage, summed over waves: [0] taking a batch, [1] forming or fetching the ray, [2] closest-hit walk, [3] any-hit
// walk of a pixel-centre ray, [4] shading, [5] fetching a shadow ray, [6] its walk, [7] reporting its answer,
// [8] waiting at the stage's barrier, [9] pixel resolve and taking the next region.
__device__ unsigned long long g_sect[16];
#define WF_SC_BEGIN unsigned long long sc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tp_ = __builtin_readcyclecounter()
#define WF_SC(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); sc_[i] += n_ - tp_; tp_ = n_; } while (0)
#define WF_SC_END do { if (COUNT && (threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 10; ++i_) if (sc_[i_]) atomicAdd(&sh->sect[i_], sc_[i_]); } while (0)
#else
#define WF_SC_BEGIN do { } while (0)
#define WF_SC(i) do { } while (0)
#define WF_SC_END do { } while (0)
#endif
struct wf_false { static constexpr bool value = false; };
struct wf_true { static constexpr bool value = true; };

// A node's record.  The walk names a node by where its record lies, in bytes from the first (what an inner
// node stores for its left child, and what goes on the stack): the address is the array plus that word, with no
// index arithmetic per step.
__device__ __forceinline__ uint2 wf_node(const DScene& S, unsigned at)
{
    return *(const uint2*)((const char*)S.knodes + at);
}

// The same through the copy of the tree's first levels in LDS where the record lies in it (the two address
// spaces in two branches, see wf_stk_put).  The records of the first levels are what every walk starts with
// and what the far children popped last are; from LDS they arrive in about half the time of a hit in the vector L1.
__device__ __forceinline__ uint2 wf_node(const DScene& S, const WfStack& k, unsigned at)
{
    if (at < k.top_bytes) {
        const wf_u2 v = k.top[at >> 3];
        return make_uint2(v.x, v.y);
    }
    return wf_node(S, at);
}
__device__ __forceinline__ void wf_node_pair(const DScene& S, const WfStack& k, unsigned at, uint2& c0, uint2& c1)
{
    // (from the copy only if every lane of the wave is in it - a wave with lanes on both sides would wait for
    // both loads, and a branch per lane costs more than it saves: 6.08 ms per lane, 6.00 ms per wave)
    if (__builtin_amdgcn_ballot_w64(at >= k.top_pairs) == 0ull) {
        const WfLdsEntries p = (WfLdsEntries)((__attribute__((address_space(3))) unsigned char*)k.top + at); // (a multiple of 8)
        const wf_u2        a = p[0], b = p[1];
        c0 = make_uint2(a.x, a.y), c1 = make_uint2(b.x, b.y);
    } else c0 = wf_node(S, at), c1 = wf_node(S, at + 8u);
}

template <bool ANY, bool COUNT = true>
__device__ __forceinline__ bool wf_traverse(const DScene& S, const WfStack& stk, f3 o, f3 d, HitRec& h,
                                            unsigned& n_tests, unsigned& n_nodes)
{
    float minT = RT_FLT_MAX;
    h.slot = -1, h.a = 0.0f, h.b = 0.0f;
    int   best_orig = -1;
    int   node = 0, sp = 0;
    float tmin_e = 0.0f, tmax_e = RT_FLT_MAX; // entry / exit t of the current cell, margins applied
    // per axis 1/d.  A zero component gets a huge positive stand-in: the plane is then "reached" only at
    // +-infinity-like t with the correct side.
    const float big = 1e18f;
    const float ix = d.x != 0.0f ? __builtin_amdgcn_rcpf(d.x) : big,
                iy = d.y != 0.0f ? __builtin_amdgcn_rcpf(d.y) : big,
                iz = d.z != 0.0f ? __builtin_amdgcn_rcpf(d.z) : big;
    v2f eDN; // the plane heads' error bounds for this ray
    {
        float eD, eN;
        tri_head_bounds(o, d, S.vert_l1, eD, eN);
        eDN = v2f{eD, eN};
    }
    const RayPairs rp = {v2f{o.x, o.y}, v2f{o.z, d.x}, v2f{d.y, d.z}};
    for (;;) {
        uint2 nd = wf_node(S, stk, (unsigned)node);
        if (COUNT) ++n_nodes;
#ifdef WF_DIAG
        if (COUNT) { const unsigned long long m_ = __ballot(1); if ((int)(threadIdx.x & 63) == __ffsll((long long)m_) - 1) atomicAdd(&g_diag[0], 64ull); }
#endif
        while (nd.y < 0xc0000000u) { // axis 0..2 in bits 30-31, 3 = leaf
#ifdef WF_DIAG
            if (COUNT) { const unsigned long long m_ = __ballot(1); if ((int)(threadIdx.x & 63) == __ffsll((long long)m_) - 1) atomicAdd(&g_diag[0], 64ull); }
#endif
            const unsigned left = nd.y & 0x3fffffffu; // where the left child's record is, in bytes from the first
            // Both children lie side by side: their records are asked for before the plane decides which
            // one the walk takes, so the decision is made while the load is under way and the chain
            // from node to node is the load alone.
            uint2 c0, c1;
            wf_node_pair(S, stk, left, c0, c1);
            const bool  ax0 = nd.y < 0x40000000u, ax2 = (int)nd.y < 0; // (two compares on the word itself, no shift)
            const float ia = ax2 ? iz : (ax0 ? ix : iy);
            const float oa = ax2 ? o.z : (ax0 ? o.x : o.y);
            // (o / d per axis kept beside 1 / d for one fused operation here: the compiler did not keep the three
            // products in registers but formed them again at every step)
            const float tplane = (__uint_as_float(nd.x) - oa) * ia;
            // origin side of the plane: below it iff (split - o) > 0 iff tplane and d (hence 1/d) agree in sign
            const bool pos = tplane > 0.0f, below = pos == (ia > 0.0f);
            const bool in = pos && tplane <= tmax_e, near = tplane < tmin_e; // the plane is crossed before the cell ends / before it begins
            // the child the walk goes on with, as its place in the pair: the near one, or the far one alone if the
            // plane lies before the cell's stretch of the ray ...
            const bool     first = below != (in && near);
            const unsigned sel = first ? 0u : 8u;
            if (in && !near) { // ... or both: the far one waits on the stack
                wf_stk_put(stk, sp, (int)(left + (sel ^ 8u)), tplane);
                ++sp;
                tmax_e = tplane * (1.0f + RT_TRAV_EPS);
            }
            nd = first ? c0 : c1;
            if (COUNT) ++n_nodes;
        }
        const int cnt = (int)(nd.y & 0x3fffffffu);
        const int first = (int)nd.x;
#ifdef WF_DIAG
        if (COUNT) {
            atomicAdd(&g_diag[cnt == 0 ? 2 : 3], 1ull);
            int mx_ = cnt; // the longest list among the wave's lanes: two triangles per trip of the loop
            for (int d_ = 32; d_ > 0; d_ >>= 1) mx_ = max(mx_, __shfl_xor(mx_, d_));
            const unsigned long long a_ = __ballot(1);
            if ((int)(threadIdx.x & 63) == __ffsll((long long)a_) - 1 && mx_ > 0)
                atomicAdd(&g_diag[10], 1ull), atomicAdd(&g_diag[11], (unsigned long long)((mx_ + 1) / 2));
        }
#endif
        const float lo = tmin_e * 0.9999f, hi = ANY ? RT_FLT_MAX : fminf(minT, tmax_e * 1.0001f);
        v2f         loh = {lo * 0.999999f, hi * 1.000001f}; // the stretch of the ray inside the cell, margins applied
        for (int i = 0; i < cnt; i += 2) { // slots first + i and first + i + 1 (a list starts at an even slot)
            const float4* hp = S.heads + (first + i);
            float4        h0 = hp[0], h1 = hp[1];
            // (the next pair asked for before this one is tested, so that a leaf's trips do not each wait a full
            // load: 6.39 vs 6.20 ms - eight more live registers, five of them spilled, and the copies)
            // (the rest of a slot asked for with its head and waited for only by survivors: 8.08 vs 7.67 ms)
            const bool two = i + 1 < cnt;
            if (COUNT) n_tests += two ? 2u : 1u;
#ifdef WF_DIAG
            if (COUNT) { const unsigned long long m_ = __ballot(1); if ((int)(threadIdx.x & 63) == __ffsll((long long)m_) - 1) atomicAdd(&g_diag[1], 128ull); }
#endif
            // (the heads as two 16-byte loads, complete here: the compiler would otherwise fetch words of them
            // later, behind a wait for a whole slot)
            asm volatile("" ::"v"(h0.x), "v"(h0.y), "v"(h0.z), "v"(h0.w), "v"(h1.x), "v"(h1.y), "v"(h1.z), "v"(h1.w));
            v2f  D2, N2;
            bool keep0, keep1;
            tri_heads_keep(h0, h1, rp, eDN, loh, D2, N2, keep0, keep1);
            keep1 = keep1 && two;
#ifdef WF_DIAG
            if (COUNT) {
                const unsigned long long a_ = __ballot(1), m0_ = __ballot(keep0), m1_ = __ballot(keep1);
                if ((int)(threadIdx.x & 63) == __ffsll((long long)a_) - 1)
                    atomicAdd(&g_diag[8], (m0_ ? 1ull : 0ull) + (m1_ ? 1ull : 0ull)), atomicAdd(&g_diag[9], (unsigned long long)(__popcll(m0_) + __popcll(m1_)));
            }
#endif
            if (!(keep0 || keep1)) continue;
            // what follows a head that lets the ray through: the edge record, then the exact test
            auto candidate = [&](const int slot, auto second) -> bool {
                {
                    const float4* er = S.edges + (size_t)3 * slot;
                    const float4  E0 = er[0], E1 = er[1];
                    const float   es = er[2].x;
                    const bool    keep2 = tri_edges_keep<decltype(second)::value>(E0, E1, es, rp, D2, N2, eDN.x, eDN.y);
#ifdef WF_DIAG
                    if (COUNT) {
                        const unsigned long long a_ = __ballot(1), m_ = __ballot(keep2);
                        if ((int)(threadIdx.x & 63) == __ffsll((long long)a_) - 1 && m_) atomicAdd(&g_diag[14], 1ull), atomicAdd(&g_diag[15], (unsigned long long)__popcll(m_));
                    }
#endif
                    if (!keep2) return false;
                }
                const float4* rec = S.tris + (size_t)RT_SLOT_F4 * slot;
                const float4  q0 = rec[0], q1 = rec[1], q2 = rec[2];
                float         a, b, t;
#ifdef WF_DIAG
                if (COUNT) {
                    float a_, b_, t_;
                    const bool ok_ = tri_test_line(q0, q1, q2, o, d, minT, a_, b_, t_);
                    const unsigned long long m_ = __ballot(1), h_ = __ballot(ok_);
                    if ((int)(threadIdx.x & 63) == __ffsll((long long)m_) - 1 && h_) atomicAdd(&g_diag[12], 1ull), atomicAdd(&g_diag[13], (unsigned long long)__popcll(h_));
                }
#endif
                if (tri_test_line(q0, q1, q2, o, d, minT, a, b, t)) {
                    if (ANY) return true;
                    const int orig = __float_as_int(q2.w);
                    bool      take = t != minT || best_orig < 0;
                    if (!take) take = S.tri_rank[orig] > S.tri_rank[best_orig];
                    if (take) h.a = a, h.b = b, minT = t, best_orig = orig, loh.y = fminf(loh.y, t * 1.000001f);
                }
                return false;
            };
            if (keep0 && candidate(first + i, wf_false())) return true;
            if (keep1 && candidate(first + i + 1, wf_true())) return true;
        }
        // A pending subtree starts where the cell above it on the stack ends, give or take the
        // margins: entry t of the top entry = tmax_e / (1 + eps), and every deeper entry starts
        // at most a factor (1 + eps) per level earlier.  A hit clearly inside the current cell
        // therefore ends the walk without popping (and loading) the entries one by one.
        if (!ANY && minT < tmax_e * 0.998f) break;
        bool found = false;
        while (sp > 0) {
            uint2 top;
            float t_below;
            wf_stk_top2(stk, sp, top, t_below);
            --sp;
            const float et = __uint_as_float(top.y);
            if (!ANY && et * (1.0f - RT_TRAV_EPS) > minT) continue;
            node = (int)top.x;
            tmin_e = et * (1.0f - RT_TRAV_EPS);
            tmax_e = t_below * (1.0f + RT_TRAV_EPS);
            found = true;
            break;
        }
        if (!found) break;
    }
    h.t = minT;
    h.slot = best_orig; // the triangle, not the slot
    return best_orig >= 0;
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
    // (the lanes of m below this one, by the instruction pair made for it: written as a popcount under the mask
    // (1 << lane) - 1 the mask was formed once per launch, spilled, and fetched back from scratch memory at every append)
    const int before = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    return mine ? base + before * per : -1;
}

// v in units of 2^-32, rounded to nearest even: llrint((double)v * 2^32) without the double arithmetic
// and the emulated 64-bit conversion.  For a = |v|, floor(a) and a - floor(a) are exact in float, and so
// is the fraction times 2^32; only that product can have a part below one unit, which rintf rounds
// as llrint would, and rounding to even is symmetric in the sign.  (|v| < 2^31; colour terms are of
// the order of 1.)
__device__ __forceinline__ long long wf_fix(float v)
{
    const float     a = fabsf(v), fl = floorf(a);
    const unsigned  lo = (unsigned)rintf((a - fl) * 4294967296.0f);
    const long long m = (long long)(((unsigned long long)(unsigned)fl << 32) | lo);
    return v < 0.0f ? -m : m;
}

// Where the colour sums of the region being worked on are: three planes of `n` 64-bit sums,
// `at[sample]` the first.  The single launch keeps them in LDS (a region belongs to one block, and
// an atomic on HBM is a transaction with the memory fabric: with the sums in HBM they made up a
// fifth of the launch's traffic); the multi-launch kernels in the workspace.
typedef __attribute__((address_space(3))) long long* WfLdsSums; // typed as LDS: ds_add_u64, not a flat atomic
typedef __attribute__((address_space(3))) unsigned char* WfLdsBytes;
typedef __attribute__((address_space(3))) unsigned* WfLdsWords;
template <class PTR>
struct WfAccT {
    PTR      at;
    unsigned n;
};
typedef WfAccT<long long*> WfAcc;   // in the workspace
typedef WfAccT<WfLdsSums>  WfAccL;  // in LDS

// Several rays of a sample may finish at the same time: order-free integer adds.
template <class ACC>
__device__ __forceinline__ void wf_acc_add(const ACC& a, unsigned sample, f3 c)
{
    __hip_atomic_fetch_add(a.at + sample, wf_fix(c.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(a.at + a.n + sample, wf_fix(c.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(a.at + 2 * a.n + sample, wf_fix(c.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The camera ray is the first to touch its sample's sums: it sets them.
template <class ACC>
__device__ __forceinline__ void wf_acc_set(const ACC& a, unsigned sample, f3 c)
{
    a.at[sample] = wf_fix(c.x);
    a.at[a.n + sample] = wf_fix(c.y);
    a.at[2 * a.n + sample] = wf_fix(c.z);
}

// Hit geometry: raytracing.cpp:76-108.
struct WfSurf {
    f3    hp, N, viewDir;
    f3    Kd, Ks;
    float Ns, Ni;
    int   illum, has_Ni, tex;
    int4  info;
    bool  spec; // Ks != 0: with Ks = 0 the specular term adds +0 to a colour that is never -0, i.e. nothing
};

// Unit vector with the hardware's reciprocal square root (1 ulp) for the two directions that only enter
// the specular term - towards the viewer and the halfway vector (raytracing.cpp:99, 131-132); the normal
// and the light direction, which place rays, keep the reference's sqrt and division (normalize3).
__device__ __forceinline__ void wf_unit(f3& a)
{
    const float l2 = dot3(a, a);
    if (l2 == 0.0f) return;
    const float r = __builtin_amdgcn_rsqf(l2);
    a.x *= r, a.y *= r, a.z *= r;
}

__device__ __forceinline__ void wf_surface(const DScene& S, f3 o, f3 dir, int orig, float a, float b, float t,
                                           WfSurf& s)
{
    s.hp = o + dir * t; // raytracing.cpp:384
    const float4 s0 = S.tri_shade[3 * orig], s1 = S.tri_shade[3 * orig + 1], s2 = S.tri_shade[3 * orig + 2];
    s.info = S.tri_info[orig];
    const float4 m0 = S.mats[4 * s.info.x], m1 = S.mats[4 * s.info.x + 1], m2 = S.mats[4 * s.info.x + 2];
    s.Kd = mk3(m0.x, m0.y, m0.z), s.Ks = mk3(m1.x, m1.y, m1.z);
    s.Ns = m0.w, s.Ni = m1.w;
    s.illum = __float_as_int(m2.x), s.has_Ni = __float_as_int(m2.y), s.tex = __float_as_int(m2.z);
    const float u = a, v = b, w = 1.0f - u - v;
    s.N = (mk3(s0.x, s0.y, s0.z) * u + mk3(s1.x, s1.y, s1.z) * v) + mk3(s2.x, s2.y, s2.z) * w;
    normalize3(s.N);
    s.spec = s.Ks.x != 0.0f || s.Ks.y != 0.0f || s.Ks.z != 0.0f;
    s.viewDir = o - s.hp;
    if (s.spec) wf_unit(s.viewDir);
}

// The specular power (raytracing.cpp:134): 0 < x <= 1 to the material's hardness, as 2^(n log2 x) with the
// hardware's log2 and exp2 (1 ulp each).  The error of the product n log2 x carries over: relative 1e-5 for a
// hardness of 150, i.e. at most 1e-5 of a colour unit where the highlight is brightest, 1/400 of a byte step -
// against some hundred instructions for the library's powf, two of them per hit and light.
__device__ __forceinline__ float wf_pow(float x, float n)
{
    // (the hardware's log2 of a denormal is -inf; times a hardness of 0 that would be NaN where powf gives 1)
    return n == 0.0f ? 1.0f : __builtin_amdgcn_exp2f(n * __builtin_amdgcn_logf(x));
}


// Where the data of the region being worked on lie, and the counters its stage appends with.
template <class ACC>
struct WfRegion {
    size_t    base;      // first ray slot of the region in every queue: region * R
    size_t    sample0;   // first sample of the region in s_minT / s_tri: region * S * msaa
    ACC       acc;       // the colour sums, and
    unsigned  acc0;      // the region's first sample in them
    long long slot_base; // pixel slot of the region's first pixel slot
    size_t    lvl0;      // first slot of the region in the buffers of the level being shaded (shadow queue and
                         // records, kept per level parity): parity * cap + base
    int*      n_child;   // rays appended to the next level's queue
    int*      n_sh;      // shadow rays appended
    WfLdsBytes pend_out; // k_wf_pipeline, where LDS has room: one byte per ray slot of the level being shaded - the count of
                         // pending queries (bits 0-2) and the lights found occluded (bits 3-6) of the slot's record; null:
                         // the word in the record in HBM
};

// ---------------------------------------------------------------------------
// one ray after its closest-hit walk: raytracing.cpp:70-271 without the recursion
// ---------------------------------------------------------------------------
// Called by all lanes of a wave together (the appends ballot); `valid` says whether the lane
// holds a ray.  j: the ray's slot in its region's queue.
//
// The light loop of raytracing.cpp:110-164 is evaluated with every shadow query taken as
// "free"; what an occluded light changes is the shadow term alone.  That term is kept apart:
// the colour is C - shadowPart * (sum over the lights of shadowMax), C = ambient or texture +
// diffuse + specular of all lights (the reference subtracts light by light between the
// additions: same value up to the last bits of a float).  shadowMax is carried from light to
// light (raytracing.cpp:104); per light the hit records v = the value the light's angle sets
// it to before its query, or -1 if the angle leaves it alone.
template <bool SOFT, class PT, class ACC>
__device__ __forceinline__ void wf_shade_ray(const DScene& S, const PT& P, const WfBuffers& B, const WfRays out,
                                             const WfRegion<ACC>& g, bool valid, int j, f3 o, f3 dir, float w,
                                             float prev_ni, unsigned meta, bool got, int orig, float ha, float hb,
                                             float ht, unsigned& n_shaded, unsigned& n_child)
{
    const int      L = P.n_lights;
    const unsigned sl = meta & 0xffffu;
    const int      k = (int)((meta >> 16) & 7u);
    const unsigned route = (meta >> 19) & 0xffu;
    const size_t   sample = g.sample0 + sl;
    unsigned char* rec = B.rec + (g.lvl0 + j) * B.rec_stride;
    float*         rec_v = (float*)(rec + sizeof(WfRecord));
    const int      per = SOFT ? B.ns : 1;
    const size_t   sq0 = g.lvl0 * L * per; // the region's shadow queue
    const bool is_hit = valid && got;
    bool       final_now = valid && !got;
    f3         colour = mk3(0, 0, 0);
    if (final_now) colour = mk3(P.bg[0], P.bg[1], P.bg[2]) * w; // raytracing.cpp:74
    WfSurf s;
    f3     C = mk3(0, 0, 0);
    if (is_hit) {
        wf_surface(S, o, dir, orig, ha, hb, ht, s), ++n_shaded;
        if (s.tex < 0) C = s.Kd * P.backgroundlighting; // raytracing.cpp:92-95
        else C = sample_texture(S, s.tex, s.info, s.hp);
    }
    // ---- children first, raytracing.cpp:166-271 (their registers are free again before the light
    // loop starts): A = mirror or transmitted ray, B = reflected ray of a refracting surface ----
    float wl = 0.0f;
    {
        bool     hasA = false, hasB = false;
        f3       a_o = mk3(0, 0, 0), a_d = mk3(0, 0, 0), b_o = mk3(0, 0, 0), b_d = mk3(0, 0, 0);
        float    a_w = 0.0f, b_w = 0.0f, n2 = 1.0f;
        unsigned a_kind = 0; // 1 mirror, 2 transmitted, 3 reflected (the soft-shadow generator's key)
        if (is_hit) {
            wl = w;
            if (k + 1 < 5) {
                const float n1 = (prev_ni != 0.0f) ? prev_ni : 1.0f;
                n2 = s.has_Ni ? s.Ni : 1.0f;
                if (s.illum == 3) {
                    const f3 refl = dir - s.N * (2.0f * dot3(s.N, dir));
                    wl = 0.5f * w;
                    a_o = s.hp + refl * 0.001f, a_d = s.hp + refl, a_w = 0.5f * w, a_kind = 1, hasA = true;
                } else if (s.Ni > 1.0f) {
                    f3 I = dir;
                    normalize3(I);
                    const float eta = (n1 != n2) ? n2 : __fdiv_rn(1.0f, n2);
                    const float cosI = dot3(I, s.N);
                    const float sin2T = (float)(((double)eta * (double)eta) * (1.0 - ((double)cosI * (double)cosI)));
                    float       rf = sin2T;
                    rf = rf < 0.0f ? 0.0f : rf;
                    rf = rf > 1.0f ? 1.0f : rf;
                    float wLocal = 0.5f * rf;
                    if (sin2T <= 1.0f) {
                        const float f = (float)((double)(eta * cosI) - sqrt((double)(1.0f - sin2T)));
                        const f3    tRay = I * eta + s.N * f;
                        wLocal += 0.1f * (1.0f - rf);
                        const float wT = 0.9f * (1.0f - rf) * w;
                        if (wT != 0.0f) a_o = s.hp + tRay * 0.001f, a_d = s.hp + tRay, a_w = wT, a_kind = 2, hasA = true;
                    }
                    const float wR = 0.5f * rf * w;
                    if (wR != 0.0f) {
                        const f3 refRay = I - s.N * (2.0f * dot3(I, s.N));
                        b_o = s.hp + refRay * 0.001f, b_d = s.hp + refRay, b_w = wR, hasB = true;
                    }
                    wl = wLocal * w;
                }
            }
        }
        n_child += (hasA ? 1u : 0u) + (hasB ? 1u : 0u);
        int pos = wf_append(g.n_child, hasA);
        if (hasA && pos < B.R) { // cannot exceed R by construction; the guard keeps a bug from writing outside
            const size_t p = g.base + pos;
            out.o[p] = WfVec3{a_o.x, a_o.y, a_o.z};
            out.d[p] = WfVec3{a_d.x, a_d.y, a_d.z};
            out.aux[p] = WfRayAux{a_w, n2, wf_meta(sl, (unsigned)(k + 1), SOFT ? ((route & 0x3fu) << 2) | a_kind : 0u)};
        }
        pos = wf_append(g.n_child, hasB);
        if (hasB && pos < B.R) {
            const size_t p = g.base + pos;
            out.o[p] = WfVec3{b_o.x, b_o.y, b_o.z};
            out.d[p] = WfVec3{b_d.x, b_d.y, b_d.z};
            out.aux[p] = WfRayAux{b_w, n2, wf_meta(sl, (unsigned)(k + 1), SOFT ? ((route & 0x3fu) << 2) | 3u : 0u)};
        }
    }
    // ---- the light loop ----
    float              shadowMax = 0.0f, T = 0.0f;
    float              shadowMax1 = 0.0f, T1 = 0.0f; // the same with the first query answered "occluded"
    float4             v4 = make_float4(-1.0f, -1.0f, -1.0f, -1.0f); // what the first four lights' angles set shadowMax to
    int                ncast = 0;
    unsigned           first_cast = 0; // the first query: its light << 24 | its place in the region's shadow queue
    unsigned long long soft_id = 0;
    if (SOFT && is_hit) {
        const int msaa = P.msaa;
        const int slot = msaa == 1 ? (int)sl : (int)(sl / (unsigned)msaa);
        const int sn = msaa == 1 ? 0 : (int)(sl % (unsigned)msaa) + 1;
        int       px = 0, py = 0;
        slot_to_pixel(P, g.slot_base + slot, px, py);
        soft_id = ((unsigned long long)py * P.W + px) * 32 + sn;
    }
    for (int li = 0; li < L; ++li) {
        bool cast = false;
        f3   lightDir = mk3(0, 0, 0);
        if (is_hit) {
            lightDir = mk3(P.lights[3 * li], P.lights[3 * li + 1], P.lights[3 * li + 2]) - s.hp;
            normalize3(lightDir);
            float angle = dot3(s.N, lightDir);
            if (angle > 0.0f) C = C + (s.Kd * angle) * P.diffusePower;
            angle = dot3(lightDir, s.N);
            float v = -1.0f;
            if (angle <= 0.2f) v = 1.0f;
            else if (angle < 0.5f) {
                angle -= 0.2f;
                v = 3.33f * (0.3f - angle);
                v = v < 0.0f ? 0.0f : v; // the std::max(shadowMax, softShadow) with softShadow >= 0 that follows
            }
            if (v >= 0.0f) shadowMax = v, shadowMax1 = v;
            if (angle > 0.0f) {
                if (s.spec) { // (a wave on a wall without highlights skips the block: walls are large)
                    f3 halfway = lightDir + s.viewDir;
                    wf_unit(halfway);
                    const float ah = dot3(s.N, halfway);
                    if (ah > 0.0f) C = C + s.Ks * wf_pow(ah, s.Ns);
                }
                // An occluded light can only raise shadowMax to 1 (raytracing.cpp:142-143).  Here
                // shadowMax is its value with every query so far answered "free"; if that is 1
                // already - set by a grazing angle at this light, or carried from an earlier one -
                // the answer cannot change the colour and the query is not traced.  (Soft shadows
                // take the maximum with a sum that rounding may carry past 1: all of them are traced.)
                cast = SOFT || shadowMax != 1.0f;
            }
            T += shadowMax;
            if (cast && ncast == 0) shadowMax1 = 1.0f; // raytracing.cpp:142-143 with that query occluded
            T1 += shadowMax1;
            // (what the angles set shadowMax to is written behind the loop, and only where a record is left: a store
            // per light into the record of every hit dirtied a line per hit, 40 % of them for hits that leave no
            // record at all.  li is the same for every lane: the compares are scalar.)
            v4.x = li == 0 ? v : v4.x, v4.y = li == 1 ? v : v4.y, v4.z = li == 2 ? v : v4.z, v4.w = li == 3 ? v : v4.w;
            if (li >= 4) rec_v[li] = v;
        }
        // one queue for all lights: a wave of shadow rays is full whatever the share of each light
        const int pos = wf_append(g.n_sh, cast, per);
        if (cast) {
            if (ncast == 0) first_cast = ((unsigned)li << 24) | (unsigned)pos; // (a region's queue has R * lights < 2^23 places)
            ++ncast;
            const f3  so = s.hp + lightDir * 0.001f; // raytracing.cpp:142
            const int tag = (li << 24) | j;
            if (!SOFT) {
                B.sh_org[sq0 + pos] = make_float4(so.x, so.y, so.z, __int_as_float(tag));
            } else { // raytracing.cpp:145-160: ns rays towards jittered points around the light
                for (int si = 0; si < per; ++si) {
                    float4 dd = make_float4(0, 0, 0, 0);
                    soft_dest(&P.lights[0], &P.soft_x[0], P.soft_n, P.soft_half, P.soft_quarter, P.soft_radius, li, si,
                              soft_id, route, dd.x, dd.y, dd.z);
                    B.sh_org[sq0 + pos + si] = make_float4(so.x, so.y, so.z, __int_as_float(tag));
                    B.sh_dst[sq0 + pos + si] = dd;
                }
            }
        }
    }
    if (is_hit) {
        if (ncast == 0) { // no answer can change it: final
            const float sm = L > 0 ? T * P.shadow_per_light : 0.0f; // raytracing.cpp:102; no light, no term
            colour = (C - mk3(sm, sm, sm)) * wl;
            final_now = true;
        } else if (!SOFT && ncast == 1) {
            // A hit with one query needs no count of pending answers: both outcomes of the light loop are known
            // here, the query's ray picks one (WF_TAG_SOLE in its tag) - no atomic, no per-light values read back.
            *(float4*)rec = make_float4(C.x, C.y, C.z, wl);
            *(uint4*)(rec + 16) = make_uint4(sl, __float_as_uint(T), __float_as_uint(T1), 0u);
            *(unsigned*)&B.sh_org[sq0 + (first_cast & WF_TAG_SLOT)].w = (first_cast & 0xff000000u) | (unsigned)j | WF_TAG_SOLE;
        } else {
            *(float4*)rec = make_float4(C.x, C.y, C.z, wl);
            *(uint4*)(rec + 16) = make_uint4(sl, (unsigned)(ncast * per), 0u, 0u);
            *(float4*)rec_v = v4; // (a record with queries has lights: its stride holds at least these 16 bytes)
            if (!SOFT && g.pend_out) g.pend_out[j] = (unsigned char)ncast; // (at most four lights there: three bits)
            if (SOFT) {
                unsigned* rec_n = (unsigned*)(rec_v + L);
                for (int x = 0; x < ((L + 3) >> 2); ++x) rec_n[x] = 0u;
            }
        }
    }
    if (valid) {
        if (k == 0) { // the camera ray fixes the sample's depth and triangle
            B.s_minT[sample] = ht;
            if (B.s_tri) B.s_tri[sample] = got ? orig : -1;
            wf_acc_set(g.acc, g.acc0 + sl, final_now ? colour : mk3(0, 0, 0));
        } else if (final_now) wf_acc_add(g.acc, g.acc0 + sl, colour);
    }
}

// A word that other lanes change with atomics, read past the vector L1.
__device__ __forceinline__ unsigned wf_peek(const unsigned* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A shadow ray reports its answer.  The ray that brings the hit's count of pending queries to
// zero knows every answer and finishes the light loop: raytracing.cpp:142-164 with the
// queries answered.  rec: the hit's record; sample0: first sample of the hit's region.
// An atomic on HBM is a transaction with the memory fabric on this hardware whatever scope it
// asks for (36.7 M of them per 4K frame of final.obj, 1.2 GB written) - which is why the count lives in a byte of
// LDS instead wherever a region's records are few enough for it (`pend`: no refraction, at most four lights, hard
// shadows; 5.1 -> 3.6 GB written per frame, 5.79 -> 5.70 ms), and why a hit with a single query has no count at all.  Tried instead and not
// kept: one byte per light set by plain stores and a lane per record finishing the light loop in
// the stage after (no atomics, but the records have left the L2 by then: 10.1 vs 9.7 ms, 14.4 vs
// 13.8 GB); non-temporal loads and stores for the queue entries (10.4 ms, traffic unchanged);
// colour sums and records as arrays of structures (atomics to three words of one line: 11.2 ms).
template <bool SOFT, class PT, class ACC>
__device__ __forceinline__ void wf_shadow_done(const PT& P, const WfBuffers& B, size_t slot, const ACC& acc,
                                               unsigned acc0, int li, bool occluded, bool sole,
                                               WfLdsBytes pend = nullptr, unsigned j = 0)
{
    const int      L = P.n_lights;
    unsigned char* rec = B.rec + slot * B.rec_stride;
    if (!SOFT && sole) { // the hit's only query: the light loop was finished for both answers when the hit was shaded
        const float4 c = *(const float4*)rec;
        const uint4  u = *(const uint4*)(rec + 16);
        const float  sm = __uint_as_float(occluded ? u.z : u.y) * P.shadow_per_light;
        wf_acc_add(acc, acc0 + u.x, (mk3(c.x, c.y, c.z) - mk3(sm, sm, sm)) * c.w);
        return;
    }
    WfRecord*      R = (WfRecord*)rec;
    const float*   rec_v = (const float*)(rec + sizeof(WfRecord));
    unsigned*      rec_n = (unsigned*)(rec_v + L);
    unsigned       bits = 0;
    if (!SOFT && pend) {
        // The count and the answers in a byte of LDS (four records to a word; a region belongs to one block): the
        // query count drops by one and the light's bit is set in one add - the byte never reaches zero before its
        // last answer and never 128, so nothing is borrowed from or carried into its neighbours.
        const unsigned sh8 = 8u * (j & 3u);
        const unsigned delta = ((occluded ? (8u << li) : 0u) - 1u) << sh8;
        const unsigned old = (__hip_atomic_fetch_add((WfLdsWords)(pend + (j & ~3u)), delta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> sh8) & 0xffu;
        if ((old & 7u) != 1u) return;
        bits = ((old + (occluded ? (8u << li) : 0u)) >> 3) & 0xfu;
    } else if (!SOFT) {
        unsigned delta = 0xffffffffu; // one query less pending
        if (occluded) {
            if (li < 24) delta += 1u << (8 + li);
            else {
                atomicOr(&R->occ_hi[(li - 24) >> 5], 1u << ((li - 24) & 31));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // performed before the count drops
            }
        }
        const unsigned old = atomicAdd(&R->pending, delta);
        if ((old & 0xffu) != 1u) return;
        bits = (old + delta) >> 8;
    } else {
        if (occluded) { // count in the light's byte (at most 216 samples): add into its 32-bit word
            atomicAdd(&rec_n[li >> 2], 1u << (8 * (unsigned)(li & 3)));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (atomicSub(&R->pending, 1u) != 1u) return;
    }
    const float4 c = *(const float4*)rec;
    const unsigned sample = acc0 + R->sample;
    float        shadowMax = 0.0f, T = 0.0f;
    for (int l2 = 0; l2 < L; ++l2) {
        const float v = rec_v[l2];
        if (v >= 0.0f) shadowMax = v;
        if (!SOFT) {
            const unsigned occ = l2 < 24 ? (bits >> l2) & 1u : (wf_peek(&R->occ_hi[(l2 - 24) >> 5]) >> ((l2 - 24) & 31)) & 1u;
            if (occ) shadowMax = 1.0f;
        } else { // raytracing.cpp:155-157: one float addition per occluded sample
            const unsigned n_occ = (wf_peek(&rec_n[l2 >> 2]) >> (8 * (unsigned)(l2 & 3))) & 0xffu;
            float          softShadow = 0.0f;
            for (unsigned x = 0; x < n_occ; ++x) softShadow += P.soft_w;
            shadowMax = (shadowMax < softShadow) ? softShadow : shadowMax;
        }
        T += shadowMax;
    }
    const float sm = T * P.shadow_per_light;
    wf_acc_add(acc, sample, (mk3(c.x, c.y, c.z) - mk3(sm, sm, sm)) * c.w);
}

// Pixels from sample sums.  MSAA == 1: one sample per slot.  MSAA > 1: ordered float sum of
// the samples / msaa, depth = mean of minT over the samples that hit (main.cpp:449-507);
// the caller passes only pixels whose centre ray hit.
template <class PT, class ACC>
__device__ __forceinline__ void wf_finalize_pixel(const PT& P, const WfBuffers& B, const ACC& acc, unsigned acc_i,
                                                  size_t p, size_t sample, float* rgb, float* depth, int* tri)
{
    if (P.msaa == 1) {
        rgb[3 * p] = wf_clamp01((float)((double)acc.at[acc_i] / WF_FIX_SCALE));
        rgb[3 * p + 1] = wf_clamp01((float)((double)acc.at[acc.n + acc_i] / WF_FIX_SCALE));
        rgb[3 * p + 2] = wf_clamp01((float)((double)acc.at[2 * acc.n + acc_i] / WF_FIX_SCALE));
        depth[p] = B.s_minT[sample];
        if (tri && B.s_tri) tri[p] = B.s_tri[sample];
        return;
    }
    float r = 0, g = 0, b = 0, dsum = 0;
    int   hits = 0;
    for (int s = 0; s < P.msaa; ++s) {
        const float sr = (float)((double)acc.at[acc_i + s] / WF_FIX_SCALE);
        const float sg = (float)((double)acc.at[acc.n + acc_i + s] / WF_FIX_SCALE);
        const float sb = (float)((double)acc.at[2 * acc.n + acc_i + s] / WF_FIX_SCALE);
        if (s == 0) r = sr, g = sg, b = sb;
        else r += sr, g += sg, b += sb;
        const float mt = B.s_minT[sample + s];
        if (mt != RT_FLT_MAX) dsum += mt, ++hits;
    }
    const float nn = (float)P.msaa;
    rgb[3 * p] = wf_clamp01(__fdiv_rn(r, nn));
    rgb[3 * p + 1] = wf_clamp01(__fdiv_rn(g, nn));
    rgb[3 * p + 2] = wf_clamp01(__fdiv_rn(b, nn));
    depth[p] = hits > 0 ? __fdiv_rn(dsum, (float)hits) : RT_FLT_MAX;
}

// The pixel of a centre ray that missed (main.cpp:439-445).
template <class PT>
__device__ __forceinline__ void wf_background_pixel(const PT& P, size_t p, float* rgb, float* depth)
{
    rgb[3 * p] = wf_clamp01(P.bg[0]);
    rgb[3 * p + 1] = wf_clamp01(P.bg[1]);
    rgb[3 * p + 2] = wf_clamp01(P.bg[2]);
    depth[p] = RT_FLT_MAX;
}

// ---------------------------------------------------------------------------
// one launch per stage (RT_WF_MULTILAUNCH=1)
// ---------------------------------------------------------------------------
// Block r: centre rays of pixel slots slot0 + [r * S, (r + 1) * S) that lie inside the
// frame, appended to region r of queue `q`.  meta: the slot inside the region.
__global__ void __launch_bounds__(WF_BLOCK)
k_wf_gen_centre(DParams P, WfBuffers B, int q, long long slot0, int n, DCounters* ctr)
{
    __shared__ int s_n;
    const int      r = blockIdx.x;
    const WfRays   out = B.rays[q];
    const size_t   base = (size_t)r * B.R;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int s_pad = (B.S + 63) & ~63;
    for (int j = threadIdx.x; j < s_pad; j += WF_BLOCK) {
        const int  i = j < B.S ? r * B.S + j : n;
        int        x = 0, y = 0;
        const bool valid = i < n && slot_to_pixel(P, slot0 + i, x, y);
        const int  pos = wf_append(&s_n, valid);
        if (valid) {
            f3 o, d;
            pixel_ray(P, (double)x, (double)y, o, d);
            out.o[base + pos] = WfVec3{o.x, o.y, o.z};
            out.d[base + pos] = WfVec3{d.x, d.y, d.z};
            out.aux[base + pos] = WfRayAux{1.0f, 1.0f, wf_meta((unsigned)j, 0u, 0u)}; // prev_Ni = 1.0f (main.cpp:528)
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out.cnt[r] = s_n;
        atomicAdd(&ctr->camera_rays, (unsigned long long)s_n);
    }
}

// Persistent trace kernel: a fixed grid of waves pulls 64-ray work items off the list
// (WF_ITEMS_PER_FETCH per atomic) until it is empty.  The 64 lanes of a wave start
// their rays together: neighbouring rays visit the same nodes and leaves at the same
// time, so most of a wave's loads fall into a few cache lines.  (Refilling single lanes
// as their rays finish was measured: it keeps more lanes busy but scatters the lanes
// over the tree, and the extra cache lines per load cost more than the idle lanes.)
//   ANY = false: queue `q` -> hit records.
//   ANY = true : shadow queue of parity `q` (regions of R * n_lights slots); every ray ends at
//                the light its record names and reports to its hit's record.
// LDS: per-thread stacks of ushort node indices.  (Keeping the top of the node array in
// LDS too was measured: 27.7 ms per 4K frame without it, 29.9 / 32.5 / 39.5 ms with 32 / 48 /
// 64 KB per block - the nodes stay in the vector L1 and the LDS costs occupancy.)
template <bool ANY>
__global__ void __launch_bounds__(WF_BLOCK)
k_wf_trace(DScene S, WfBuffers B, const DParams* __restrict__ Pp, int q, int msaa, int* head, DCounters* ctr)
{
    extern __shared__ __align__(16) unsigned char s_mem[];
    const WfStack stk = wf_stack(S, s_mem);
    const int*    prefix = ANY ? B.prefix + (B.G + 1) : B.prefix;
    const int*    cnt = ANY ? B.sh_cnt + (size_t)q * B.G : B.rays[q].cnt;
    const int     n_regions = B.G;
    const int     total = prefix[n_regions];
    const int     lane = threadIdx.x & 63;
    const int     L = ANY ? Pp->n_lights : 1;
    unsigned      n_tests = 0, n_nodes = 0, n_rays = 0;
    int           r = 0; // region of the previous item: consecutive items mostly share it
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
            HitRec h;
            ++n_rays;
            if (!ANY) {
                const size_t i = (size_t)r * B.R + idx;
                const WfVec3 ow = B.rays[q].o[i], dn = B.rays[q].d[i];
                const f3     o = mk3(ow.x, ow.y, ow.z);
                const bool   got = wf_traverse<false>(S, stk, o, mk3(dn.x, dn.y, dn.z) - o, h, n_tests, n_nodes);
                B.hits[i] = make_float4(__int_as_float(got ? h.slot : -1), h.a, h.b, h.t);
            } else {
                const size_t i = ((size_t)q * B.cap + (size_t)r * B.R) * L + idx;
                const float4 ow = B.sh_org[i];
                const f3     o = mk3(ow.x, ow.y, ow.z);
                const int    li = (int)((unsigned)__float_as_int(ow.w) >> 24);
                const f3     dst = mk3(Pp->lights[3 * li], Pp->lights[3 * li + 1], Pp->lights[3 * li + 2]);
                const bool   got = wf_traverse<true>(S, stk, o, dst - o, h, n_tests, n_nodes);
                const WfAcc acc = {B.acc, (unsigned)B.n_samples};
                wf_shadow_done<false>(*Pp, B, (size_t)q * B.cap + (size_t)r * B.R + (__float_as_uint(ow.w) & WF_TAG_SLOT), acc,
                                      (unsigned)(r * B.S * msaa), li, got, (__float_as_uint(ow.w) & WF_TAG_SOLE) != 0u);
            }
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

// MSAA > 1, after the centre rays (queue 1) are traced: misses are final pixels; every hit
// pixel queues its `msaa` sample rays (main.cpp:445-507) into region r of queue 0.
__global__ void __launch_bounds__(WF_BLOCK)
k_wf_msaa_emit(DParams P, WfBuffers B, long long slot0, float* rgb, float* depth, int* tri,
               unsigned char* centre_hit, DCounters* ctr)
{
    __shared__ int s_n;
    const int      r = blockIdx.x;
    const WfRays   in = B.rays[1], out = B.rays[0];
    const int      n = in.cnt[r];
    const size_t   base = (size_t)r * B.R;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int n_pad = (n + 63) & ~63;
    for (int j = threadIdx.x; j < n_pad; j += WF_BLOCK) {
        const bool valid = j < n;
        bool       got = false;
        int        x = 0, y = 0, slot = 0;
        if (valid) {
            slot = (int)(in.aux[base + j].meta & 0xffffu);
            const float4 hit = B.hits[base + j];
            got = __float_as_int(hit.x) >= 0;
            slot_to_pixel(P, slot0 + (long long)r * B.S + slot, x, y);
            const size_t p = (size_t)y * P.W + x;
            if (tri) tri[p] = __float_as_int(hit.x);
            centre_hit[(size_t)r * B.S + slot] = got ? 1 : 0;
            if (!got) wf_background_pixel(P, p, rgb, depth);
        }
        const int pos = wf_append(&s_n, got, P.msaa);
        if (got && pos + P.msaa <= B.R) {
            for (int s = 0; s < P.msaa; ++s) {
                const float ox = P.msaa == 4 ? c_msaa4[s][0] : c_msaa16[s][0];
                const float oy = P.msaa == 4 ? c_msaa4[s][1] : c_msaa16[s][1];
                f3          o, d;
                pixel_ray(P, (double)x + (double)ox, (double)y + (double)oy, o, d);
                const size_t p = base + pos + s;
                out.o[p] = WfVec3{o.x, o.y, o.z};
                out.d[p] = WfVec3{d.x, d.y, d.z};
                out.aux[p] = WfRayAux{1.0f, 1.0f, wf_meta((unsigned)(slot * P.msaa + s), 0u, 0u)};
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out.cnt[r] = min(s_n, B.R);
        atomicAdd(&ctr->camera_rays, (unsigned long long)s_n);
    }
}

// Block r: the rays of level `level` of region r, traced before (hit records).
__global__ void __launch_bounds__(WF_BLOCK)
k_wf_shade(DScene S, DParams P, WfBuffers B, int level, long long slot0, DCounters* ctr)
{
    __shared__ int s_child, s_sh;
    const int      r = blockIdx.x;
    const WfRays   in = B.rays[level & 1], out = B.rays[(level + 1) & 1];
    const int      n = in.cnt[r];
    unsigned       n_shaded = 0, n_child = 0;
    WfRegion<WfAcc> g;
    g.base = (size_t)r * B.R, g.sample0 = (size_t)r * B.S * P.msaa, g.slot_base = slot0 + (long long)r * B.S;
    g.lvl0 = (size_t)(level & 1) * B.cap + g.base, g.n_child = &s_child, g.n_sh = &s_sh;
    g.pend_out = nullptr;
    g.acc.at = B.acc, g.acc.n = (unsigned)B.n_samples, g.acc0 = (unsigned)g.sample0;
    if (threadIdx.x == 0) s_child = 0, s_sh = 0;
    __syncthreads();
    const int n_pad = (n + 63) & ~63; // whole waves iterate together: the ballots need every lane
    for (int j = threadIdx.x; j < n_pad; j += WF_BLOCK) {
        const bool   valid = j < n;
        const size_t i = g.base + j;
        unsigned     meta = 0;
        float4       ow = make_float4(0, 0, 0, 0), dn = make_float4(0, 0, 0, 0), hit = make_float4(0, 0, 0, 0);
        if (valid) {
            const WfVec3   ov = in.o[i], dv = in.d[i];
            const WfRayAux ax = in.aux[i];
            meta = ax.meta, ow = make_float4(ov.x, ov.y, ov.z, ax.w), dn = make_float4(dv.x, dv.y, dv.z, ax.ni), hit = B.hits[i];
        }
        const f3 o = mk3(ow.x, ow.y, ow.z);
        wf_shade_ray<false>(S, P, B, out, g, valid, j, o, mk3(dn.x, dn.y, dn.z) - o, ow.w, dn.w, meta,
                            __float_as_int(hit.x) >= 0, __float_as_int(hit.x), hit.y, hit.z, hit.w, n_shaded, n_child);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out.cnt[r] = min(s_child, B.R);
        B.sh_cnt[(size_t)(level & 1) * B.G + r] = s_sh;
    }
    for (int s = 32; s > 0; s >>= 1) n_shaded += __shfl_down(n_shaded, s), n_child += __shfl_down(n_child, s);
    if ((threadIdx.x & 63) == 0 && (n_shaded | n_child)) {
        atomicAdd(&ctr->shaded_hits, (unsigned long long)n_shaded);
        atomicAdd(&ctr->child_rays, (unsigned long long)n_child);
    }
}

__global__ void __launch_bounds__(WF_BLOCK)
k_wf_finalize(DParams P, WfBuffers B, long long slot0, int n, const unsigned char* __restrict__ centre_hit,
              float* rgb, float* depth, int* tri)
{
    const int i = blockIdx.x * WF_BLOCK + threadIdx.x;
    if (i >= n || (P.msaa > 1 && !centre_hit[i])) return;
    int x, y;
    if (!slot_to_pixel(P, slot0 + i, x, y)) return;
    const WfAcc acc = {B.acc, (unsigned)B.n_samples};
    wf_finalize_pixel(P, B, acc, (unsigned)(i * P.msaa), (size_t)y * P.W + x, (size_t)i * P.msaa, rgb, depth, tri);
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
// an eighth of a 4K frame; the same with the fused stages, only the last 10 / 25 / 50 % of the
// slots in regions of half the size: 9.80 / 9.90 / 10.16 ms against 9.79, 2.02 / 2.01 / 2.06
// against 2.04 ms for an eighth).
// A block takes a region (one atomic), runs every stage of its pixels - centre rays and MSAA
// gating, camera samples, five levels of child and shadow rays, pixel resolve - and takes the
// next region.  Stages of a region are separated by block barriers only; there is no
// device-wide synchronisation and no launch boundary inside the frame, so a CU always has
// other blocks' stages to run while one block waits for its slowest ray.
//
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
    DScene     S;
    DParams    P;
    WfBuffers  B;
    long long  slot0;
    int        n, n_regions; // pixel slots of the frame part, regions of B.S slots they make
    float*     rgb;
    float*     depth;
    int*       tri;
    DCounters* ctr;
    const int* order; // regions in the order they are handed out (null: by index), see k_wf_order_place
    unsigned*  cost;  // per region: how long it took this time (null: not recorded)
    unsigned   top_bytes; // of the node array, from its start, that every block copies into LDS (0: none)
    unsigned   pend_bytes; // LDS bytes per level parity for the records' pending counts, behind that copy (0: the counts stay in the records)
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

__device__ __forceinline__ WfLaunchC wf_args()
{
    WfLaunchC p = (WfLaunchC)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

struct WfStat {
    unsigned n_tests, n_nodes, n_shaded, n_child;
};

#define WF_ST_CENTRE 0 // MSAA > 1: the pixel-centre rays that gate the samples (main.cpp:435-445)
#define WF_ST_CAMERA 1 // level 0: one ray per pixel, or msaa per pixel whose centre hit
#define WF_ST_LEVEL 2  // closest hits of level `level` and the shadow rays of the level before

// One stage of region r, run by block qr on its own set of buffers.  The index space is
// [closest-hit rays, padded to whole waves][shadow rays]: a wave runs one kind of walk, the
// closest-hit rays (the longer walks) start first.  Shadow queries of one level and closest-hit
// queries of the next come out of the same shade step and neither needs the other's answers, so
// a region's chain of dependent stages is one pass per level.
template <bool COUNT, bool SOFT>
__device__ __forceinline__ void wf_stage(const int kind, const int level, const int r, const int qr, WfShared* sh,
                                         unsigned char* s_mem, WfStat& st)
{
    const WfLaunchC A = wf_args();
    const DScene    S = wf_ld(A->S);
    const WfBuffers B = wf_ld(A->B);
    const auto&     P = A->P;
    WfStack         stk = wf_stack(S, s_mem);
    const int       msaa = P.msaa;
    const int       n_slots = A->n - r * B.S; // pixel slots from the region's first to the end of the frame part
    WfRegion<WfAccL> g;
    g.base = (size_t)qr * B.R, g.sample0 = (size_t)qr * B.S * msaa;
    g.slot_base = A->slot0 + (long long)r * B.S;
    g.lvl0 = (size_t)__builtin_amdgcn_readfirstlane((level & 1) * B.cap) + g.base;
    g.n_child = nullptr, g.n_sh = nullptr; // (set where they are used, below)
    const size_t in0 = (size_t)__builtin_amdgcn_readfirstlane(((level + 1) & 1) * B.cap) + g.base; // the level before
    const int    ms = msaa == 16 ? 4 : (msaa == 4 ? 2 : 0); // msaa is 1, 4 or 16
    g.acc.at = (WfLdsSums)(s_mem + (size_t)S.stk_cap * WF_BLOCK * WF_STACK_ENTRY_BYTES); // after the stacks
    g.acc.n = (unsigned)(B.S * msaa), g.acc0 = 0;
    stk.top = (WfLdsEntries)(s_mem + (size_t)S.stk_cap * WF_BLOCK * WF_STACK_ENTRY_BYTES + (size_t)B.S * msaa * 24); // after the sums
    stk.top_bytes = A->top_bytes;
    stk.top_pairs = __builtin_amdgcn_readfirstlane(A->top_bytes >= 16u ? A->top_bytes - 8u : 0u);
    const unsigned   pend_bytes = A->pend_bytes;
    const WfLdsBytes pend0 = (WfLdsBytes)stk.top + A->top_bytes; // two planes of pend_bytes: level parity 0, 1
    g.pend_out = pend_bytes ? pend0 + (level & 1) * pend_bytes : (WfLdsBytes) nullptr;
    const WfLdsBytes pend_in = pend_bytes ? pend0 + ((level + 1) & 1) * pend_bytes : (WfLdsBytes) nullptr;
    // (what is read from LDS is the same for every lane; said so, it lives in scalar registers and
    // not, as a vector register per value, in scratch during the walks)
    int n_first = B.S, n_sh = 0;
    if (kind == WF_ST_CAMERA && msaa > 1) n_first = __builtin_amdgcn_readfirstlane(sh->n_hit) * msaa;
    if (kind == WF_ST_LEVEL)
        n_first = __builtin_amdgcn_readfirstlane(sh->child[level % 3]),
        n_sh = __builtin_amdgcn_readfirstlane(sh->shd[(level + 2) % 3]);
    if (kind != WF_ST_CENTRE && threadIdx.x == 0) // the outputs of the stage after this one
        sh->child[(level + 2) % 3] = 0, sh->shd[(level + 1) % 3] = 0;
    const int par = kind == WF_ST_CENTRE ? 1 : (level & 1); // centre 1, camera 0, level l: l & 1
    if (threadIdx.x == 0) sh->batch[par ^ 1] = 0;             // the stage after this one counts from 0
    const WfRays in = wf_ld(A->B.rays[level & 1]), out = wf_ld(A->B.rays[(level + 1) & 1]);
    // (batches of 32 or 16 rays, to shorten the stages of the last regions of a launch: the stages of those regions
    // take as long as before - it is the longest walk that counts, not how many lanes walk - and 1080p frames
    // take 2.6 and 3.9 ms instead of 2.0)
    const int bw = 64;
    const int    seg_c = (n_first + bw - 1) & ~(bw - 1);
    const int    total = seg_c + n_sh;
    const int    lane = threadIdx.x & 63;
    const size_t sq_in = in0 * P.n_lights * (SOFT ? B.ns : 1);
    // A wave takes the next 64 rays when it is through with the last: the walks of a mirror corridor are
    // ten times as long as those next to them, and with a fixed share per wave three waves wait for the fourth.
    WF_SC_BEGIN;
    for (;;) {
        int i0 = 0;
        if (lane == 0) i0 = atomicAdd(&sh->batch[par], 1) * bw;
        i0 = __builtin_amdgcn_readfirstlane(i0);
        WF_SC(0);
        if (i0 >= total) break;
        int idx = i0 + lane;
        asm volatile("" : "+v"(idx)); // keeps per-lane addresses from being formed (and spilled) ahead of the loop
        if (i0 < seg_c) {
            // Before the walk only the ray itself is formed; whatever else the shading needs (weight,
            // prev_Ni, sample, queue addresses) is fetched or derived again afterwards from the
            // lane's index, so that it does not sit in registers - or in scratch - during the walk.
            bool valid = lane < bw && idx < n_first;
            f3   o = mk3(0, 0, 0), dst = mk3(0, 0, 0);
            int  xy = 0;
            if (kind == WF_ST_LEVEL) {
                if (valid) {
                    const WfVec3 ow = in.o[g.base + idx], dn = in.d[g.base + idx];
                    o = mk3(ow.x, ow.y, ow.z), dst = mk3(dn.x, dn.y, dn.z);
                }
            } else {
                int x = 0, y = 0, sn = -1;
                if (kind == WF_ST_CAMERA && msaa > 1) { // pixel of a centre hit, as the centre stage left it
                    xy = valid ? B.hit_list[(size_t)qr * B.S + (idx >> ms)].y : 0;
                    x = xy & 0xffff, y = xy >> 16;
                    sn = idx & (msaa - 1);
                } else {
                    valid = valid && idx < n_slots && slot_to_pixel(P, g.slot_base + idx, x, y);
                    xy = x | (y << 16);
                }
                if (valid) { // (the four blend weights from a table per column and row instead of two double divisions per
                             // ray: the same rays bit for bit and the same frame times - the divisions hide behind the walks)
                    double fx = (double)x, fy = (double)y;
                    if (sn >= 0) {
                        fx += (double)(msaa == 4 ? c_msaa4[sn][0] : c_msaa16[sn][0]);
                        fy += (double)(msaa == 4 ? c_msaa4[sn][1] : c_msaa16[sn][1]);
                    }
                    pixel_ray(P, fx, fy, o, dst);
                }
                const unsigned long long cam = __ballot(valid);
                if (lane == 0 && cam) atomicAdd(&sh->n_cam, (int)__popcll(cam));
            }
            {
                const unsigned long long m = __ballot(valid);
                if (lane == 0 && m) atomicAdd(&sh->n_closest, (int)__popcll(m));
            }
            const f3 dir = dst - o;
            HitRec   h;
            h.slot = -1, h.a = 0.0f, h.b = 0.0f, h.t = RT_FLT_MAX;
            bool got = false;
            WF_SC(1);
            // The pixel-centre ray only decides whether the pixel's samples are traced (main.cpp:439-445); unless
            // the caller wants its triangle, any hit answers that, and the walk ends at the first one.
            if (kind == WF_ST_CENTRE && !A->tri) {
                if (valid) got = wf_traverse<true, COUNT>(S, stk, o, dir, h, st.n_tests, st.n_nodes);
            } else if (valid) got = wf_traverse<false, COUNT>(S, stk, o, dir, h, st.n_tests, st.n_nodes);
            WF_SC(kind == WF_ST_CENTRE ? 3 : 2);
            int j = idx;
            asm volatile("" : "+v"(j)); // what follows is computed after the walk, not carried through it
            if (kind == WF_ST_CENTRE) { // misses are final pixels, hits queue their samples
                if (valid) {
                    const size_t p = (size_t)(xy >> 16) * P.W + (xy & 0xffff);
                    if (A->tri) A->tri[p] = h.slot;
                    if (!got) wf_background_pixel(P, p, A->rgb, A->depth);
                }
                const int pos = wf_append(&sh->n_hit, valid && got);
                if (valid && got) B.hit_list[(size_t)qr * B.S + pos] = make_int2(j, xy);
            } else {
                float    w = 1.0f, ni = 1.0f; // prev_Ni = 1.0f for camera rays (main.cpp:528)
                unsigned meta = 0;
                if (valid) {
                    if (kind == WF_ST_LEVEL) {
                        const WfRayAux ax = in.aux[g.base + j];
                        w = ax.w, ni = ax.ni, meta = ax.meta;
                    }
                    else if (msaa > 1) meta = wf_meta((unsigned)((B.hit_list[(size_t)qr * B.S + (j >> ms)].x << ms) + (j & (msaa - 1))), 0u, 0u);
                    else meta = wf_meta((unsigned)j, 0u, 0u);
                }
                // (the two counters' addresses are formed here, from a level the compiler cannot look through: formed
                // at the top of the stage they sat in a vector register each - the same for every lane - that was
                // spilled before the walks and fetched back from scratch memory at every light of every batch)
                int lvl = level;
                asm volatile("" : "+s"(lvl));
                g.n_child = &sh->child[(lvl + 1) % 3], g.n_sh = &sh->shd[lvl % 3];
                wf_shade_ray<SOFT>(S, P, B, out, g, valid, j, o, dir, w, ni, meta, got, h.slot, h.a, h.b, h.t,
                                   st.n_shaded, st.n_child);
            }
            WF_SC(4);
        } else if (lane < bw && idx < total) {
            const float4 ow = B.sh_org[sq_in + (idx - seg_c)];
            const f3     o = mk3(ow.x, ow.y, ow.z);
            const int    li = (int)((unsigned)__float_as_int(ow.w) >> 24);
            f3           l = mk3(P.lights[3 * li], P.lights[3 * li + 1], P.lights[3 * li + 2]);
            if (SOFT) {
                const float4 dd = B.sh_dst[sq_in + (idx - seg_c)];
                l = mk3(dd.x, dd.y, dd.z);
            }
            HitRec h;
            if (idx == i0) atomicAdd(&sh->n_shadow, min(bw, total - i0));
#ifdef WF_DIAG
            const unsigned t0_ = st.n_tests, n0_ = st.n_nodes;
#endif
            WF_SC(5);
            const bool got = wf_traverse<true, COUNT>(S, stk, o, l - o, h, st.n_tests, st.n_nodes);
            WF_SC(6);
#ifdef WF_DIAG
            if (COUNT) {
                atomicAdd(&g_diag[4], 1ull), atomicAdd(&g_diag[5], got ? 1ull : 0ull);
                atomicAdd(&g_diag[6], (unsigned long long)(st.n_tests - t0_)), atomicAdd(&g_diag[7], (unsigned long long)(st.n_nodes - n0_));
            }
#endif
            wf_shadow_done<SOFT>(P, B, in0 + (__float_as_uint(ow.w) & WF_TAG_SLOT),
                                 g.acc, g.acc0, li, got, (__float_as_uint(ow.w) & WF_TAG_SOLE) != 0u, pend_in,
                                 __float_as_uint(ow.w) & WF_TAG_SLOT);
            WF_SC(7);
        }
    }
    __syncthreads();
    WF_SC(8);
    WF_SC_END;
}

// The order in which the next frame's regions are handed out: the longest first.  The launch ends when its
// last region does, and a region that mirrors keep busy for five levels takes ten to twenty times the median
// (1 ms against 0.1 ms at 1080p) however empty the GPU around it is; handed out by index such a region starts
// whenever its turn comes, and the blocks that are through wait for it (a quarter of the launch for a 1080p
// frame, a third for an eighth of a 4K frame).  Nothing cheap predicts a region's cost before it runs - but
// the frame before this one measured it: a renderer is asked for the same or a neighbouring view again and
// again, and the cost of a patch of screen changes slowly with the view.  So every region leaves its duration
// behind (WfLaunch::cost), and the kernels below sort the regions by it, longest first, for the next launch of the
// same shape: bins of a quarter octave, stable counting sort.  It is a hint and nothing else: every
// region is rendered whatever the order, the sums are order-free, the frame is the same bit for bit.
#define WF_ORDER_BINS 128
__device__ __forceinline__ int wf_order_bin(unsigned c) // bin 0 holds the longest: quarter octaves of the cost, reversed
{
    if (c == 0) return WF_ORDER_BINS - 1;
    const int e = 31 - __clz((int)c);
    const int q = e >= 2 ? (int)((c >> (e - 2)) & 3u) : 0;
    return WF_ORDER_BINS - 1 - (4 * e + q);
}

// Three small launches: the n regions are cut into 16 * WF_ORDER_BLOCKS contiguous runs, one per wave;
//   k_wf_order_count    regions of each run per bin
//   k_wf_order_offsets  where each run's regions of each bin go (one block)
//   k_wf_order_place    every region into its place
// (as one block of 16 waves the sort took 0.11 ms per 4K frame, a sixtieth of the launch it serves: three
// passes of some hundred instructions per 64 regions are work for more than one CU.)
#define WF_ORDER_BLOCKS 16
#define WF_ORDER_RUNS (16 * WF_ORDER_BLOCKS)

// The bin of region i.  Only the regions a launch by index would end up waiting for are moved to the front:
// region i of n starts when about i / n of the launch is over (the launch takes the sum of the durations over
// the blocks), and matters if it would still be running near the end - if it takes `hot` times what is left by
// then, or longer.  Everything else stays in index order behind them, in the last bin.  (Sorting all regions by
// duration puts the expensive ones on the GPU at the same time, where they take longer than mixed with cheap
// ones: lego 4K 3.75 ms by index, 4.0 ms sorted; final 4K: the tail gone and nearly as much lost before it.
// Handing the regions that go first out in halves or quarters, to as many blocks, so that their chains of levels get
// shorter: 1080p frames 10 to 70 % slower when every such region is cut (smaller regions fill a block's waves
// worse, and most of these regions are not what the launch ends with); cut only where a handful of regions outlast
// all others - at least 0.7 of the longest, at most a sixteenth of the blocks - fin.obj 1080p, whose eight regions
// of 1.1 to 1.4 ms are its launch, 1.33 -> 1.22 ms, redballs 1.24 -> 1.28 ms, every 4K frame unchanged: not kept.
// A fixed order for the first frame of a view, when no durations exist yet - a stride of 0.618 n through the regions,
// or back to front - helps where the expensive regions happen to lie late in index order and hurts elsewhere:
// against by index, the stride lego 3.54 -> 3.34, fin 1.40 -> 1.30, redballs 1.50 -> 1.63, final 5.87 -> 6.08 ms;
// back to front final 5.61, refractionballs 1.65 -> 1.72 ms: no rule without knowing the scene, by index stays.)
struct WfOrderRule {
    float left_per_region; // hot * (sum of the durations / blocks) / n
    int   n;
};
__device__ __forceinline__ WfOrderRule wf_order_rule(const WfCounters* ctr, int n, float hot, int blocks)
{
    WfOrderRule r;
    r.left_per_region = (float)((double)ctr->cost_sum / (double)max(blocks, 1)) * hot / (float)max(n, 1);
    r.n = n;
    return r;
}
__device__ __forceinline__ int wf_order_bin_of(const WfOrderRule& r, unsigned c, int i)
{
    return (float)c >= r.left_per_region * (float)(r.n - i) ? min(wf_order_bin(c), WF_ORDER_BINS - 2) : WF_ORDER_BINS - 1;
}
__device__ __forceinline__ void wf_order_run(int n, int& lo, int& hi) // the run of this wave
{
    const int per = ((n + WF_ORDER_RUNS - 1) / WF_ORDER_RUNS + 63) & ~63; // whole rounds of 64
    const int run = blockIdx.x * 16 + (threadIdx.x >> 6);
    lo = min(run * per, n), hi = min(lo + per, n);
}

__global__ void __launch_bounds__(1024)
k_wf_order_count(const unsigned* __restrict__ cost, int n, float hot, int blocks, const WfCounters* __restrict__ ctr,
                 int* __restrict__ at) // [WF_ORDER_RUNS][WF_ORDER_BINS]
{
    __shared__ int s_at[16][WF_ORDER_BINS];
    const int      t = threadIdx.x, w = t >> 6, lane = t & 63;
    for (int i = t; i < 16 * WF_ORDER_BINS; i += 1024) (&s_at[0][0])[i] = 0;
    __syncthreads();
    const WfOrderRule rule = wf_order_rule(ctr, n, hot, blocks);
    int               lo, hi;
    wf_order_run(n, lo, hi);
    for (int i0 = lo; i0 < hi; i0 += 64) { // (nearly every region stays where it is: counted by a ballot, not by 64 atomics on one word)
        const int                i = i0 + lane;
        const int                b = i < hi ? wf_order_bin_of(rule, cost[i], i) : -1;
        const unsigned long long stay = __ballot(b == WF_ORDER_BINS - 1);
        if (lane == 0) s_at[w][WF_ORDER_BINS - 1] += __popcll(stay);
        if (b >= 0 && b != WF_ORDER_BINS - 1) atomicAdd(&s_at[w][b], 1);
    }
    __syncthreads();
    for (int i = t; i < 16 * WF_ORDER_BINS; i += 1024) at[(size_t)blockIdx.x * 16 * WF_ORDER_BINS + i] = (&s_at[0][0])[i];
}

__global__ void __launch_bounds__(1024) k_wf_order_offsets(int* __restrict__ at)
{
    // thread = (segment of WF_ORDER_RUNS / 8 runs, bin): its runs' counts are fetched together, not one after the other
    constexpr int  SEG = WF_ORDER_RUNS / 8;
    __shared__ int s_seg[8][WF_ORDER_BINS];
    __shared__ int s_total[WF_ORDER_BINS];
    const int      b = threadIdx.x & (WF_ORDER_BINS - 1), g = threadIdx.x / WF_ORDER_BINS;
    int            c[SEG];
#pragma unroll
    for (int k = 0; k < SEG; ++k) c[k] = at[(g * SEG + k) * WF_ORDER_BINS + b];
    int run = 0;
#pragma unroll
    for (int k = 0; k < SEG; ++k) { // within the segment: where each run's regions of bin b start
        const int v = c[k];
        c[k] = run;
        run += v;
    }
    s_seg[g][b] = run;
    __syncthreads();
    if (g == 0) {
        int tot = 0;
        for (int k = 0; k < 8; ++k) {
            const int v = s_seg[k][b];
            s_seg[k][b] = tot;
            tot += v;
        }
        s_total[b] = tot;
    }
    __syncthreads();
    int before = s_seg[g][b]; // the segments before this one, plus all regions of the bins before b
    for (int k = 0; k < b; ++k) before += s_total[k];
#pragma unroll
    for (int k = 0; k < SEG; ++k) at[(g * SEG + k) * WF_ORDER_BINS + b] = c[k] + before;
}

// Stable: regions of one bin keep their index order, so neighbouring regions - which walk the same part of the
// tree - are still handed out one after the other.
__global__ void __launch_bounds__(1024)
k_wf_order_place(const unsigned* __restrict__ cost, int n, float hot, int blocks, const WfCounters* __restrict__ ctr,
                 const int* __restrict__ at, int* __restrict__ order)
{
    __shared__ int s_at[16][WF_ORDER_BINS];
    const int      t = threadIdx.x, w = t >> 6, lane = t & 63;
    for (int i = t; i < 16 * WF_ORDER_BINS; i += 1024) (&s_at[0][0])[i] = at[(size_t)blockIdx.x * 16 * WF_ORDER_BINS + i];
    __syncthreads();
    const WfOrderRule rule = wf_order_rule(ctr, n, hot, blocks);
    int               lo, hi;
    wf_order_run(n, lo, hi);
    for (int i0 = lo; i0 < hi; i0 += 64) {
        const int          i = i0 + lane;
        const bool         valid = i < hi;
        const int          b = valid ? wf_order_bin_of(rule, cost[i], i) : -1;
        unsigned long long todo = __ballot(valid);
        while (todo) { // the lanes of one bin take consecutive places, in lane order
            const int                leader = __ffsll((long long)todo) - 1;
            const int                lb = __shfl(b, leader);
            const unsigned long long m = __ballot(valid && b == lb);
            const int                base = s_at[w][lb];
            if (valid && b == lb) order[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
            __builtin_amdgcn_wave_barrier();
            if (lane == leader) s_at[w][lb] = base + __popcll(m);
            __builtin_amdgcn_wave_barrier();
            todo &= ~m;
        }
    }
}

template <bool COUNT, bool SOFT>
__global__ void __launch_bounds__(WF_BLOCK, WF_PIPE_WAVES)
k_wf_pipeline(WfLaunch)
{
    extern __shared__ __align__(16) unsigned char s_mem[];
    __shared__ WfShared sh;
    WfStat              st = {0, 0, 0, 0};
    // The workspace has one set of queues, records, colour sums and lists per block, not per
    // region: the block runs region r on set qr.  Consecutive regions of a block reuse the same few
    // hundred KB (11.5 -> 10.6 ms per 4K frame against a set per region), and the workspace does
    // not grow with the frame.
    const int qr = blockIdx.x;
    { // the tree's first levels into LDS (the loop's first barrier comes before any walk)
        const WfLaunchC A = wf_args();
        const unsigned  n_top = A->top_bytes >> 3;
        const WfLdsEntries top = (WfLdsEntries)(s_mem + (size_t)A->S.stk_cap * WF_BLOCK * WF_STACK_ENTRY_BYTES + (size_t)A->B.S * A->P.msaa * 24);
        for (unsigned i = threadIdx.x; i < n_top; i += WF_BLOCK) {
            const uint2 v = A->S.knodes[i];
            top[i] = wf_u2{v.x, v.y};
        }
    }
#ifdef WF_SECT
    if (threadIdx.x < 10) sh.sect[threadIdx.x] = 0;
#endif
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const int        next = atomicAdd(&wf_args()->B.ctr->head_region, 1);
            const WfLaunchC  A0 = wf_args();
            sh.region = (A0->order && next < A0->n_regions) ? A0->order[next] : next;
            sh.child[1] = 0, sh.shd[0] = 0, sh.n_hit = 0, sh.n_cam = 0; // what the camera stage appends to
            sh.batch[0] = 0, sh.batch[1] = 0;
            sh.n_closest = 0, sh.n_shadow = 0;
        }
        __syncthreads();
        const int r = __builtin_amdgcn_readfirstlane(sh.region);
        if (r >= wf_args()->n_regions) break;
        const unsigned long long t_region = wf_args()->cost ? wall_clock64() : 0ull;
        WF_TL(0);
#ifdef WF_TIMELINE
        if (threadIdx.x == 0 && r < WF_TL_REGIONS) g_timeline[(size_t)r * 16 + 15] = blockIdx.x;
#endif
        // One call site for all stages: a single copy of the walks and of the shading code in the
        // kernel (three specialised copies: 10.6 ms per 4K frame).
        for (int stage = wf_args()->P.msaa > 1 ? 0 : 1; stage <= 6; ++stage) {
            const int level = stage > 1 ? stage - 1 : 0; // stage 6 = level 5: only the shadow rays of level 4
            if (stage > 1 && __builtin_amdgcn_readfirstlane(sh.child[level % 3] | sh.shd[(level + 2) % 3]) == 0) break;
            wf_stage<COUNT, SOFT>(stage == 0 ? WF_ST_CENTRE : (stage == 1 ? WF_ST_CAMERA : WF_ST_LEVEL), level, r, qr,
                                  &sh, s_mem, st); // ends with the stage's barrier
            WF_TL(1 + stage);
        }
        WF_TL(13);
        {
            const WfLaunchC A = wf_args();
            const WfBuffers B = wf_ld(A->B);
            const int       msaa = A->P.msaa;
            const long long slot_base = A->slot0 + (long long)r * B.S;
            const WfAccL    acc = {(WfLdsSums)(s_mem + (size_t)A->S.stk_cap * WF_BLOCK * WF_STACK_ENTRY_BYTES),
                                   (unsigned)(B.S * msaa)};
            const int       n_px = msaa > 1 ? __builtin_amdgcn_readfirstlane(sh.n_hit) : min(B.S, A->n - r * B.S);
            for (int j = threadIdx.x; j < n_px; j += WF_BLOCK) {
                int slot = j, x = 0, y = 0;
                if (msaa > 1) {
                    const int2 e = B.hit_list[(size_t)qr * B.S + j];
                    slot = e.x, x = e.y & 0xffff, y = e.y >> 16;
                } else if (!slot_to_pixel(A->P, slot_base + slot, x, y)) continue;
                wf_finalize_pixel(A->P, B, acc, (unsigned)(slot * msaa), (size_t)y * A->P.W + x,
                                  ((size_t)qr * B.S + slot) * msaa, A->rgb, A->depth, A->tri);
            }
            if (threadIdx.x == 0) {
                atomicAdd(&A->ctr->camera_rays, (unsigned long long)sh.n_cam);
                atomicAdd(&A->ctr->rays_closest, (unsigned long long)sh.n_closest);
                atomicAdd(&A->ctr->rays_shadow, (unsigned long long)sh.n_shadow);
            }
        }
        WF_TL(14);
        if (threadIdx.x == 0 && wf_args()->cost) { // (100 MHz ticks; a region takes 20 us to a few ms)
            const unsigned took = (unsigned)min(wall_clock64() - t_region, 0xfffffffeull) + 1u;
            wf_args()->cost[r] = took;
            atomicAdd(&wf_args()->B.ctr->cost_sum, (unsigned long long)took);
        }
    }
#ifdef WF_SECT
    __syncthreads();
    if (COUNT && threadIdx.x < 10) atomicAdd(&g_sect[threadIdx.x], sh.sect[threadIdx.x]);
#endif
    DCounters* ctr = wf_args()->ctr;
    for (int s = 32; s > 0; s >>= 1) {
        st.n_tests += __shfl_down(st.n_tests, s);
        st.n_nodes += __shfl_down(st.n_nodes, s);
    }
    if (COUNT) { // the timed variant leaves these out
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&ctr->tri_tests, (unsigned long long)st.n_tests);
            atomicAdd(&ctr->node_visits, (unsigned long long)st.n_nodes);
        }
        for (int s = 32; s > 0; s >>= 1)
            st.n_shaded += __shfl_down(st.n_shaded, s), st.n_child += __shfl_down(st.n_child, s);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&ctr->shaded_hits, (unsigned long long)st.n_shaded);
            atomicAdd(&ctr->child_rays, (unsigned long long)st.n_child);
        }
    }
}
