// rt_device.h - device-side data layout and the per-ray device functions of
// the gfx950 ray tracer (KD traversal, ray/triangle test, shading).
//
// Numerics contract: every operation that decides geometry (which triangle is
// hit, where, and the rays spawned from there) is evaluated in the reference's
// own order with separately rounded multiplies and adds, so the results are
// bit-identical to the reference compiled for x86-64 (no fused multiply-add
// there).  This file is compiled with -ffp-contract=off and additionally
// carries the pragma below.  Colour accumulation uses path weights instead of
// the reference's recursion (it is linear in the child colours), which moves
// rounding by a few ulp only; the frame pipeline forms the specular power with
// the hardware's log2 and exp2 (rt_wavefront.h: wf_pow, relative error 1e-5,
// far below a byte step), the per-ray entry points with the library's powf.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

#define RT_FLT_MAX 3.402823466e+38f
#define RT_MAX_LIGHTS_DEV 64
#define RT_MAX_PENDING 6

// One KD node, 16 bytes, nodes in breadth-first order so that the two
// children of a node are adjacent (right = left + 1) and depth grows with the
// index.  KDtree.h:9-27 holds six bounds per node; only the split plane
// (leftChild's end on the node's axis, KDtree.cpp:245-264) is needed here.
struct DNode {
    float   split;
    int32_t left;     // index of the left child, -1 for a leaf
    int32_t tri_off;  // first triangle slot of this node
    int32_t cnt_axis; // triangle count | axis << 28
};

// Triangle slots: every node's list in node order; a triangle may be listed by several leaves.  A list starts at
// an even slot (an unused slot fills the gap).  Ties in t go to the triangle the reference's walk would have
// tested last on its root-to-leaf path (raytracing.cpp:311-338), recorded per triangle in DScene::tri_rank.
// Per slot, in three arrays that a walk reads in this order, each only for what the one before let through:
//   heads  one float4 per slot, the plane heads of slots 2p and 2p + 1 interleaved (tri_heads_keep):
//            heads[2p]     = n0.x n1.x n0.y n1.y
//            heads[2p + 1] = n0.z n1.z k0   k1        (n, k: the triangle's plane, scaled; zero for no triangle)
//   edges  three float4: (ga.x gb.x ga.y gb.y) s, (ga.z gb.z ka kb) s, (s, -, -, -)      (tri_edges_keep)
//   tris   three float4, what the exact test takes:
//            q0 = e1.x e1.y e1.z e2.x      e1 = v0 - v2, e2 = v1 - v2 (raytracing.cpp:366-368)
//            q1 = e2.y e2.z v2.x v2.y
//            q2 = v2.z e1.x*e2.y e1.z*e2.y bits(original triangle index)
// Compact form of the same tree for the wavefront trace kernels, 8 bytes per node:
//   inner: x = bits(split plane), y = axis << 30 | index of the left child
//   leaf : x = first triangle slot, y = 3 << 30 | triangle count
// (only built when inner nodes carry no triangles of their own).
#define RT_SLOT_F4 3 // float4 per triangle slot in DScene::tris
struct DScene {
    const DNode*   nodes;
    const uint2*   knodes;
    const float4*  tris;
    const float4*  heads;
    const float4*  edges;
    const float4*  tri_shade; // per original triangle: 3 float4 = nA nB nC (xyz) + w: material, t0|t1.., see pack
    const int4*    tri_info;  // per original triangle: material, t0, t1, t2
    const float*   texcoords; // 3 per entry
    const float4*  mats;      // 4 float4 per material: Kd.xyz Ns | Ks.xyz Ni | illum has_Ni tex _ | pad
    const int4*    tex_info;  // per texture: byte offset, width, height, 0
    const uint8_t* texels;
    const int32_t* tri_rank;  // per original triangle: order in which the reference's walk tests it
    int32_t        n_nodes, n_slots, n_triangles, stack_entries;
    // wavefront traversal stack: the first stk_cap entries of a thread live in LDS, deeper
    // ones in stk_spill (per block: [entry - stk_cap][thread]); an entry is (node, bits of the entry t)
    uint2*          stk_spill;
    int32_t         stk_cap;
    float           vert_l1; // largest |x| + |y| + |z| over the vertices (tri_heads_keep's bound on n . v2)
};

struct DParams {
    int32_t W, H, msaa, n_lights;
    float   backgroundlighting, diffusePower, ShadowFactor;
    float   shadow_per_light; // ShadowFactor / n_lights as the host's float division gives it (raytracing.cpp:102); 0 without lights
    float   bg[3];
    float   pad1;
    int32_t tile_w, tile_h, rank, world;
    int32_t tiles_x, tiles_y, n_my_tiles, span; // span: consecutive shares this call covers, from `rank`
    float   cam[24];
    float   lights[3 * RT_MAX_LIGHTS_DEV];
    // soft shadows (raytracing.cpp:145-160): soft_n = values the reference's float loop
    // `for (x = -1; x <= 1; x += delta)` takes (0 = hard shadows), soft_x those values,
    // soft_w = 1.0f / n / n / n, the jitter terms delta / 2 and delta / 4
    int32_t soft_n, pad2;
    float   soft_half, soft_quarter, soft_w, soft_radius;
    float   soft_x[8];
};

// The counter-based stand-in for the rand() calls of raytracing.cpp:151-153: 0 .. 2^31 - 1 from
// (camera ray id, route through the ray tree, light, sample ordinal, component).
__device__ __forceinline__ unsigned soft_rand(unsigned long long id, unsigned path, unsigned light, unsigned sample,
                                              unsigned comp)
{
    unsigned long long z = id * 0x9E3779B97F4A7C15ull +
                           (((unsigned long long)path << 24) | ((unsigned long long)light << 16) |
                            ((unsigned long long)sample << 2) | comp);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (unsigned)(z >> 33);
}

// Destination of soft-shadow sample `si` of light `li` (raytracing.cpp:148-154)
template <class FP> // FP: pointer to the light / loop-value arrays, in whatever address space they live
__device__ __forceinline__ void soft_dest(FP lights, FP soft_x, int soft_n, float half,
                                          float quarter, float radius, int li, int si, unsigned long long id,
                                          unsigned path, float& dx, float& dy, float& dz)
{
    const int    iz = si % soft_n, iy = (si / soft_n) % soft_n, ix = si / (soft_n * soft_n);
    const double RM = 2147483647.0;
    const float  rx = (float)((double)half * (double)soft_rand(id, path, li, si, 0) / RM - (double)quarter);
    const float  ry = (float)((double)half * (double)soft_rand(id, path, li, si, 1) / RM - (double)quarter);
    const float  rz = (float)((double)half * (double)soft_rand(id, path, li, si, 2) / RM - (double)quarter);
    dx = lights[3 * li] + (soft_x[ix] * radius + rx);
    dy = lights[3 * li + 1] + (soft_x[iy] * radius + ry);
    dz = lights[3 * li + 2] + (soft_x[iz] * radius + rz);
}

struct DCounters {
    unsigned long long rays_closest, rays_shadow, tri_tests, node_visits, camera_rays;
    unsigned long long shaded_hits, child_rays;
};

struct f3 {
    float x, y, z;
};
__host__ __device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
// Vec3D.h:12-18 (vector times float)
__device__ __forceinline__ f3 operator*(f3 a, float f) { return f3{a.x * f, a.y * f, a.z * f}; }
// Vec3D.h:192-194
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// Vec3D.h:142-151: length by sqrt, then multiply by the reciprocal
__device__ __forceinline__ float normalize3(f3& a)
{
    float len = __fsqrt_rn(dot3(a, a));
    if (len == 0.0f) return 0.0f;
    float rez = __fdiv_rn(1.0f, len);
    a.x *= rez;
    a.y *= rez;
    a.z *= rez;
    return len;
}

// a = na / D, b = nb / D, c = nc / D, each correctly rounded (round to nearest even), i.e.
// bit-identical to IEEE division.  The instruction sequence is the one the compiler emits
// for an f32 division (reciprocal, one Newton step, two residual corrections) with the
// scaling and fix-up stages left out; those only act when an exponent is extreme, so the
// short form is used when |D| and the numerators lie in [2^-40, 2^40] (or are zero) and the
// full division otherwise.  The reciprocal refinement is shared by the three quotients.
__device__ __forceinline__ void div3_rn(float na, float nb, float nc, float D, float& a, float& b, float& c)
{
    const float lo = 9.094947e-13f, hi = 1.0995116e12f; // 2^-40, 2^40
    const float aD = fabsf(D), m1 = fabsf(na), m2 = fabsf(nb), m3 = fabsf(nc);
    const bool  ok = aD >= lo && aD <= hi && (m1 == 0.0f || (m1 >= lo && m1 <= hi)) &&
                    (m2 == 0.0f || (m2 >= lo && m2 <= hi)) && (m3 == 0.0f || (m3 >= lo && m3 <= hi));
    if (ok) {
        const float r0 = __builtin_amdgcn_rcpf(D);
        const float e0 = __builtin_fmaf(-D, r0, 1.0f);
        const float r1 = __builtin_fmaf(e0, r0, r0);
        float q, e;
        q = na * r1, e = __builtin_fmaf(-D, q, na), q = __builtin_fmaf(e, r1, q);
        e = __builtin_fmaf(-D, q, na), a = __builtin_fmaf(e, r1, q);
        q = nb * r1, e = __builtin_fmaf(-D, q, nb), q = __builtin_fmaf(e, r1, q);
        e = __builtin_fmaf(-D, q, nb), b = __builtin_fmaf(e, r1, q);
        q = nc * r1, e = __builtin_fmaf(-D, q, nc), q = __builtin_fmaf(e, r1, q);
        e = __builtin_fmaf(-D, q, nc), c = __builtin_fmaf(e, r1, q);
    } else {
        a = __fdiv_rn(na, D);
        b = __fdiv_rn(nb, D);
        c = __fdiv_rn(nc, D);
    }
}

struct HitRec {
    int   slot; // -1: miss
    float a, b, t;
};

// Ray/triangle test, raytracing.cpp:354-386 with Matrix33.h:124-139 spelled
// out.  det(P0,P1,P2) = P0x*P1y*P2z + P1x*P2y*P0z + P2x*P0y*P1z
//                     - P0z*P1y*P2x - P1z*P2y*P0x - P2z*P0y*P1x,
// every triple product as (x*y)*z and the sum left to right.
//
// The reference evaluates a, b, t in full and then rejects on
// a < 0 || b < 0 || a + b > 1 || t < 0 || t > minT.  Most tests end in a rejection, and
// a rejection can usually be proven from the numerators alone: a correctly rounded
// quotient has the sign of the exact one unless it underflows, and it exceeds a bound
// whenever the numerator exceeds bound * |D| by more than a few ulps.  So the numerators
// are evaluated one at a time (each in the reference's operation order) and the test
// returns as soon as one of these certain rejections applies; the divisions are reached
// only by tests that are hits or lie within 1e-5 of an edge.  Whatever is not certain
// falls through to the full evaluation, so the outcome is the reference's in every case
// (finite inputs assumed: a NaN numerator would slip through the reference's comparisons).
// Measured on final.obj at 4K MSAA4: 1.44 G tests reach the first numerator pair, 55 % go on
// to the second, 21 % to the third, 10 % to the divisions; per wave (a stage runs when any
// lane needs it) the figures are 60 % / 26 % / 13 %.  Testing two triangles per call with
// packed FP32 arithmetic was tried and is slower (13.6-15.6 vs 12.7 ms per frame): the pair
// needs twice the registers and survives a stage whenever either triangle does.  Packing within one
// test - D with the numerator of a, the numerator of b with that of t, term by term, v_pk_mul_f32 /
// v_pk_add_f32 being separately rounded like the scalar forms - halves the arithmetic instructions and
// keeps every bit, but the operands have to sit in aligned register pairs: 92 instead of 48 bytes of
// scratch per lane under the 72-register cap, 9.7-9.8 vs 9.2 ms per 4K frame.
__device__ __forceinline__ bool tri_test(const float4 q0, const float4 q1, const float4 q2, f3 o,
                                         f3 d, float minT, float& a, float& b, float& t)
{
    const float e1x = q0.x, e1y = q0.y, e1z = q0.z, e2x = q0.w, e2y = q1.x, e2z = q1.y;
    const float cx = o.x - q1.z, cy = o.y - q1.w, cz = o.z - q2.x;
    const float p1 = q2.y, p4 = q2.z; // e1x*e2y, e1z*e2y
    const float e2x_dy = e2x * d.y, dx_e1y = d.x * e1y, e2z_dy = e2z * d.y, dz_e1y = d.z * e1y;
    const float D = p1 * d.z + e2x_dy * e1z + dx_e1y * e2z - p4 * d.x - e2z_dy * e1x - dz_e1y * e2x;
    const float da = (cx * e2y) * d.z + e2x_dy * cz + (d.x * cy) * e2z - (cz * e2y) * d.x -
                     e2z_dy * cx - (d.z * cy) * e2x;
#ifndef RT_TRI_NO_STAGES
    // |D| in [2^-40, 2^40] and |numerator| >= 2^-80 keep every quotient below clear of underflow
    const float aD = fabsf(D);
    const bool  Dok = aD >= 9.094947e-13f && aD <= 1.0995116e12f;
    const float tiny = 8.271806e-25f;   // 2^-80
    const float aDm = aD * 1.00001f;    // |n| > aDm  =>  |n / D| > 1 after rounding
    const int   sD = __float_as_int(D);
    if (Dok) {
        const bool negA = (__float_as_int(da) ^ sD) < 0; // da / D < 0
        if (negA ? fabsf(da) >= tiny : fabsf(da) > aDm) return false; // a < 0, or a > 1 (then b < 0 or a + b > 1)
    }
#endif
    const float db = (e1x * cy) * d.z + (cx * d.y) * e1z + dx_e1y * cz - (e1z * cy) * d.x -
                     (cz * d.y) * e1x - dz_e1y * cx;
#ifndef RT_TRI_NO_STAGES
    if (Dok) {
        const bool negB = (__float_as_int(db) ^ sD) < 0;
        if (negB && fabsf(db) >= tiny) return false; // b < 0
        // here a >= 0 or a is tiny, b likewise: a + b > 1 for certain when (da + db) / D clears 1
        if (!negB && (__float_as_int(da) ^ sD) >= 0 && fabsf(da + db) > aDm) return false;
    }
#endif
    const float dt = p1 * cz + (e2x * cy) * e1z + (cx * e1y) * e2z - p4 * cx - (e2z * cy) * e1x -
                     (cz * e1y) * e2x;
#ifndef RT_TRI_NO_STAGES
    if (Dok) {
        const bool posQ = (__float_as_int(dt) ^ sD) >= 0; // dt / D > 0, i.e. t = -(dt / D) < 0
        if (fabsf(dt) >= tiny && (posQ || fabsf(dt) > minT * aDm)) return false; // t < 0, or t > minT
    }
#endif
    if (D == 0.0f) { // Matrix33.h:132-134: solve() returns (0,0,0); t = -0.0f
        a = 0.0f;
        b = 0.0f;
        t = -0.0f;
    } else {
        float q;
        div3_rn(da, db, dt, D, a, b, q);
        t = -q;
    }
    return !(a < 0.0f || b < 0.0f || a + b > 1.0f || t < 0.0f || t > minT);
}

// The reference's test with nothing in front of it: the four determinants in its operation order, the three
// quotients, its five comparisons.  This is the exact test of the frame walks, where two filters stand before
// it (the plane heads and the edge records below): what reaches it is a wave of rays most of which hit, and a
// stage that any lane needs is run by the whole wave, so tri_test's early returns would only add their
// comparisons and the bookkeeping of who is still in (66 of 162 vector instructions).
// How the walks' test got here (final.obj, 4K MSAA4, 1.44 G tests per frame): the walk knows which stretch
// [lo, hi] of the ray lies in the cell it is in, and every cell lists every triangle that reaches into it, so a
// triangle whose plane the ray crosses before lo has been tested in an earlier cell, one crossed after hi will
// be met in a later one unless a nearer hit ends the walk first, one crossed behind the origin is no hit: 33 %
// of the tests cross behind the origin, 15 % before the cell, 27 % beyond it, 6 % beyond the closest hit so
// far, 19-25 % inside the cell.  First D and the numerator of t went to the front of the exact evaluation
// (8.21 -> 8.07 ms per frame), then the plane head took their place at a third of the instructions (7.67 ms),
// then the edge records dropped the three in four of the remaining tests whose wave holds no hit (7.30 ms).
__device__ __forceinline__ bool tri_test_line(const float4 q0, const float4 q1, const float4 q2, f3 o, f3 d,
                                              float minT, float& a, float& b, float& t)
{
    const float e1x = q0.x, e1y = q0.y, e1z = q0.z, e2x = q0.w, e2y = q1.x, e2z = q1.y;
    const float cx = o.x - q1.z, cy = o.y - q1.w, cz = o.z - q2.x;
    const float p1 = q2.y, p4 = q2.z; // e1x*e2y, e1z*e2y
    const float e2x_dy = e2x * d.y, dx_e1y = d.x * e1y, e2z_dy = e2z * d.y, dz_e1y = d.z * e1y;
    const float D = p1 * d.z + e2x_dy * e1z + dx_e1y * e2z - p4 * d.x - e2z_dy * e1x - dz_e1y * e2x;
    const float da = (cx * e2y) * d.z + e2x_dy * cz + (d.x * cy) * e2z - (cz * e2y) * d.x -
                     e2z_dy * cx - (d.z * cy) * e2x;
    const float db = (e1x * cy) * d.z + (cx * d.y) * e1z + dx_e1y * cz - (e1z * cy) * d.x -
                     (cz * d.y) * e1x - dz_e1y * cx;
    const float dt = p1 * cz + (e2x * cy) * e1z + (cx * e1y) * e2z - p4 * cx - (e2z * cy) * e1x -
                     (cz * e1y) * e2x;
    if (D == 0.0f) { // Matrix33.h:132-134: solve() returns (0,0,0); t = -0.0f
        a = 0.0f;
        b = 0.0f;
        t = -0.0f;
    } else {
        float q;
        div3_rn(da, db, dt, D, a, b, q);
        t = -q;
    }
    return !(a < 0.0f || b < 0.0f || a + b > 1.0f || t < 0.0f || t > minT);
}

// The plane head of a slot, and what it decides.  With n = e1 x e2, the determinant of the reference's test is
// D = n . dir and the numerator of t is N = n . (o - v2), t = -(N / D): where the ray crosses the triangle's
// plane.  A walk that knows the stretch [lo, hi] of the ray inside its cell can drop three tests in four on D
// and N alone - in the reference's operation order that takes 45 instructions, but the decision only has to
// be safe, not bit-exact: the head holds (n, n . v2) divided by
// B = max_i (|e1_j e2_k| + |e1_k e2_j|), so that every component of the scaled normal, and of the sum of
// absolute products behind it, is at most 1.  Then
//   |D_ref / B - Dh| <= 11 * 2^-24 * (|dx| + |dy| + |dz|)            (7 roundings of the reference's sum of
//   |N_ref / B - Nh| <= 13 * 2^-24 * (|ox| + |oy| + |oz| + V)         six triple products, the rest ours)
// with V >= |v2x| + |v2y| + |v2z|: two numbers per ray (tri_head_bounds: eD = 2^-20 * ..., eN = 2^-19 * ...,
// neither below 2^-60, so that what is called certain stays clear of underflow in the reference's quotient
// with B in [2^-40, 2^40]), no per-test error arithmetic.  The test is dropped only if t < 0, t > hi or t < lo holds for every (D, N) within those bounds
// and the sign of D is certain; anything else goes on (to the edge record, then to tri_test_line), so hits and their a, b, t
// stay the reference's bits.  A degenerate triangle (B = 0 or not finite) has a zero head: never dropped.
__device__ __forceinline__ void tri_head_bounds(f3 o, f3 d, float vert_l1, float& eD, float& eN)
{
    const float sd = fabsf(d.x) + fabsf(d.y) + fabsf(d.z), so = fabsf(o.x) + fabsf(o.y) + fabsf(o.z) + vert_l1;
    const float huge = 1.1529215e18f, floor_ = 8.6736174e-19f, inf = __builtin_inff(); // 2^60, 2^-60
    eD = sd < huge ? fmaxf(sd * 9.5367432e-7f, floor_) : inf; // (not less-than also catches NaN: no test is dropped)
    eN = so < huge ? fmaxf(so * 1.9073486e-6f, floor_) : inf;
}

// Two plane heads at a time, in the halves of packed FP32 operations (v_pk_fma_f32 and friends: two separately
// rounded results per instruction, a scalar operand taken from either half of a register pair for both).  A wave
// runs the triangle loop as long as its fullest leaf, four trips on average; in pairs it is two, of 24 vector
// instructions for two heads against 18 for one.  The ray travels as three register pairs.
typedef float v2f __attribute__((ext_vector_type(2)));
#define RT_PK(name, text)                                                                                     \
    __device__ __forceinline__ v2f name(v2f a, v2f b) { v2f r; asm(text : "=v"(r) : "v"(a), "v"(b)); return r; }
#define RT_PK3(name, text)                                                                                    \
    __device__ __forceinline__ v2f name(v2f a, v2f b, v2f c) { v2f r; asm(text : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
RT_PK(pk_mul_lo, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]")                        // a * b.lo
RT_PK(pk_mul_hi, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]")           // a * b.hi
RT_PK(pk_add_lo, "v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]")                        // a + b.lo
RT_PK(pk_sub_lo, "v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]") // a - b.lo
RT_PK(pk_add_hi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]")           // a + b.hi
RT_PK(pk_rsub_hi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[1,0]") // b.hi - a
RT_PK3(pk_fma_lo, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]")                 // a * b.lo + c
RT_PK3(pk_fma_hi, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]")  // a * b.hi + c
RT_PK3(pk_fms_lo, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]") // a * b.lo - c
RT_PK3(pk_fms_hi, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]") // a * b.hi - c
#undef RT_PK
#undef RT_PK3

struct RayPairs {
    v2f oxy, ozdx, dyz; // (o.x, o.y) (o.z, d.x) (d.y, d.z)
};

// h0 = n0.x n1.x n0.y n1.y, h1 = n0.z n1.z k0 k1; eDN = (eD, eN); loh = (lo_s, hi_s), lo_s = lo / 1.000001,
// hi_s = hi * 1.000001 (the rounding of the reference's quotient).  D, N: the heads' D / B and N / B.
__device__ __forceinline__ void tri_heads_keep(const float4 h0, const float4 h1, const RayPairs& r, v2f eDN, v2f loh,
                                               v2f& D, v2f& N, bool& keep0, bool& keep1)
{
    const v2f nx = {h0.x, h0.y}, ny = {h0.z, h0.w}, nz = {h1.x, h1.y}, kk = {h1.z, h1.w};
    D = pk_fma_hi(nx, r.ozdx, pk_fma_lo(ny, r.dyz, pk_mul_hi(nz, r.dyz)));
    N = pk_fma_lo(nx, r.oxy, pk_fma_hi(ny, r.oxy, pk_fms_lo(nz, r.ozdx, kk)));
    const v2f aD = {fabsf(D.x), fabsf(D.y)};
    const v2f lowD = pk_sub_lo(aD, eDN), upD = pk_add_lo(aD, eDN); // |D| / B lies in [lowD, upD]
    // nq = N sign(D) = -t |D|
    const v2f nq = {__uint_as_float(__float_as_uint(N.x) ^ (__float_as_uint(D.x) & 0x80000000u)),
                    __uint_as_float(__float_as_uint(N.y) ^ (__float_as_uint(D.y) & 0x80000000u))};
    const v2f far_ = pk_add_hi(nq, eDN), near_ = pk_rsub_hi(nq, eDN); // nq + eN, eN - nq
    const v2f hiD = pk_mul_hi(upD, loh), loD = pk_mul_lo(lowD, loh);
    // (no short circuits: one straight run of instructions, no branch around half of it)
    const float eN = eDN.y;
    {
        const int sure = lowD.x > 0.0f;
        const int behind = nq.x > eN;        // t < 0
        const int beyond = far_.x < -hiD.x;  // t > hi
        const int before = near_.x < loD.x;  // t < lo
        keep0 = !(sure & (behind | beyond | before));
    }
    {
        const int sure = lowD.y > 0.0f;
        const int behind = nq.y > eN;
        const int beyond = far_.y < -hiD.y;
        const int before = near_.y < loD.y;
        keep1 = !(sure & (behind | beyond | before));
    }
}

// The edge record of a slot, and what it decides.  A ray that crosses the triangle's plane inside the cell (the
// plane head lets it through) still misses the triangle three times in four, and it does so together with its
// neighbours: of the wave trips through the exact test that any lane asks for, three in four find no hit in any
// lane.  Where the plane is crossed decides a and b: with n = e1 x e2, ga = (e2 x n) / n.n and gb = (n x e1) / n.n,
// a = ga . (P - v2) and b = gb . (P - v2) for a point P of the plane, and with P = o + t dir, t = -N / D:
//   a D = (ga . o + ka) D - N (ga . dir),  ka = -ga . v2      (likewise b; all of it times the head's 1 / B)
// - eight fused operations per coordinate from the head's D and N, no division; the two coordinates go through the
// halves of packed operations like the two heads of a pair (nine instructions for both; 3.11 -> 3.04 G vector
// instructions per 4K frame of final.obj, its duration unchanged).  The record holds (ga, ka) and
// (gb, kb) times s = 1 / max |component of ga, gb|, and s.  Again the decision only has to be safe: the reference
// rejects on the SIGN of its a and b, and the sign of a rounded determinant is certain once the exact one
// exceeds 7 * 2^-24 times the sum of the absolute triple products, which is at most |c|_1 |dir|_1 max|e2_j|
// (c = o - v2); times s / B that is at most 21 * 2^-24 * so * sd (s <= sqrt 3 |n| / |e|, B >= |n| / sqrt 3; so, sd
// the two sums of tri_head_bounds), our own roundings add 26 * 2^-24 * so * sd: T = 2^-18 so sd covers both.
// a + b > 1 is certain when s |D| (a + b - 1) clears 4 T plus 2^-12 (|a| + |b|) s |D| - the relative error of
// the reference's quotients, 7 * 2^-24 |dir|_1 B / |D|, is below that for |D| / B >= 2^-8 |dir|_1, and steeper
// crossings only are decided on this rule.  Nothing is decided unless the sign of D is certain; a degenerate
// triangle has a zero record: never dropped.  What is kept goes on to the exact test, so hits and their a, b, t
// stay the reference's bits.
// The two coordinates travel in the halves of packed operations: E0 = ga.x gb.x ga.y gb.y, E1 = ga.z gb.z ka kb
// (times s).  D2, N2: the D / B and N / B of the pair of heads the slot's head was tested in; SECOND: the slot
// is the second of the pair.
template <bool SECOND>
__device__ __forceinline__ bool tri_edges_keep(const float4 E0, const float4 E1, const float s, const RayPairs& r, v2f D2,
                                               v2f N2, float eD, float eN)
{
    const v2f gx = {E0.x, E0.y}, gy = {E0.z, E0.w}, gz = {E1.x, E1.y}, kk = {E1.z, E1.w};
    const v2f Go = pk_fma_lo(gx, r.oxy, pk_fma_hi(gy, r.oxy, pk_fma_lo(gz, r.ozdx, kk))); // (ga . o + ka, gb . o + kb)
    const v2f Gd = pk_fma_hi(gx, r.ozdx, pk_fma_lo(gy, r.dyz, pk_mul_hi(gz, r.dyz)));     // (ga . dir, gb . dir)
    const v2f X = SECOND ? pk_fms_hi(Go, D2, pk_mul_hi(Gd, N2)) : pk_fms_lo(Go, D2, pk_mul_lo(Gd, N2)); // Go D - N Gd
    const float    Dh = SECOND ? D2.y : D2.x;
    const unsigned sg = __float_as_uint(Dh) & 0x80000000u;
    const float    xa = __uint_as_float(__float_as_uint(X.x) ^ sg), xb = __uint_as_float(__float_as_uint(X.y) ^ sg);
    const float    T = (eD * 2097152.0f) * eN; // 2^-18 so sd (eD = 2^-20 sd, eN = 2^-19 so)
    const float    aD = fabsf(Dh), lowD = aD - eD;
    const int      sure = lowD > 0.0f;
    const int      ra = xa < -T, rb = xb < -T;
    const float    xs = __builtin_fmaf(-s, aD, xa + xb); // s |D| (a + b - 1)
    const float    thr = __builtin_fmaf(fabsf(xa) + fabsf(xb), 2.44140625e-4f, 4.0f * T);
    const int      steep = lowD >= eD * 4096.0f;
    const int      rc = steep & (xs > thr);
    return !(sure & (ra | rb | rc));
}

#define RT_TRAV_EPS 1e-5f

// Closest-hit (ANY = false) or any-hit (ANY = true) query over the KD tree.
// Iterative, front to back, with a per-thread stack of (far child, entry t)
// kept in LDS at stk[entry * stride].  The exit t of a popped subtree is the
// entry t of the stack entry below it, so it is not stored.  The result equals
// the reference's leaf walk (raytracing.cpp:281-351): the smallest t over all
// triangles, ties to the triangle the walk would have tested last.
template <bool ANY>
__device__ __forceinline__ bool kd_trace(const DScene& S, f3 o, f3 dst, uint2* stk, int stride,
                                         HitRec& h, unsigned& n_tests, unsigned& n_nodes)
{
    const f3 d = dst - o; // raytracing.cpp:284
    float    minT = RT_FLT_MAX;
    h.slot = -1, h.a = 0.0f, h.b = 0.0f;
    int   node = 0, sp = 0;
    float tmin = 0.0f, tmax = RT_FLT_MAX;
    const float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
    if (S.n_nodes == 0) {
        h.t = minT;
        return false;
    }
    for (;;) {
        const DNode nd = S.nodes[node];
        ++n_nodes;
        const int cnt = nd.cnt_axis & 0x0fffffff;
        for (int i = 0; i < cnt; ++i) {
            const int    slot = nd.tri_off + i;
            const float4 q0 = S.tris[RT_SLOT_F4 * slot], q1 = S.tris[RT_SLOT_F4 * slot + 1], q2 = S.tris[RT_SLOT_F4 * slot + 2];
            float        a, b, t;
            ++n_tests;
            if (tri_test(q0, q1, q2, o, d, minT, a, b, t)) {
                if (ANY) return true;
                bool take = t != minT || h.slot < 0;
                if (!take) { // equal t: keep the one the reference's test order ends with
                    const int cur = __float_as_int(S.tris[RT_SLOT_F4 * h.slot + 2].w);
                    take = S.tri_rank[__float_as_int(q2.w)] > S.tri_rank[cur];
                }
                if (take) h.slot = slot, h.a = a, h.b = b, minT = t;
            }
        }
        if (nd.left >= 0) {
            const int   axis = nd.cnt_axis >> 28;
            const float oa = axis == 0 ? o.x : (axis == 1 ? o.y : o.z);
            const float da = axis == 0 ? d.x : (axis == 1 ? d.y : d.z);
            const float ia = axis == 0 ? ix : (axis == 1 ? iy : iz);
            const bool  below = (oa < nd.split) || (oa == nd.split && da <= 0.0f);
            const int   first = nd.left + (below ? 0 : 1), second = nd.left + (below ? 1 : 0);
            const float tplane = (nd.split - oa) * ia;
            if (da == 0.0f || tplane <= 0.0f || !(tplane <= tmax * (1.0f + RT_TRAV_EPS))) {
                node = first;
            } else if (tplane < tmin * (1.0f - RT_TRAV_EPS)) {
                node = second;
            } else {
                stk[sp * stride] = make_uint2((unsigned)second, __float_as_uint(tplane));
                ++sp;
                node = first;
                tmax = tplane;
            }
            continue;
        }
        // leaf: next subtree that can still hold a closer hit
        bool found = false;
        while (sp > 0) {
            --sp;
            const uint2 e = stk[sp * stride];
            const float et = __uint_as_float(e.y);
            if (!ANY && et > minT * (1.0f + RT_TRAV_EPS)) continue;
            node = (int)e.x;
            tmin = et;
            tmax = sp > 0 ? __uint_as_float(stk[(sp - 1) * stride].y) : RT_FLT_MAX;
            found = true;
            break;
        }
        if (!found) break;
    }
    h.t = minT;
    return h.slot >= 0;
}

// Material::getTexture / getTexture2, mesh.cpp:485-553: nearest texel of a
// planar world-x/y mapping.  The texel index can reach one row past the image;
// textures carry two zero rows.
__device__ __forceinline__ f3 sample_texture(const DScene& S, int tex, int4 info, f3 hp)
{
    const float* t0 = S.texcoords + 3 * info.y;
    const float* t1 = S.texcoords + 3 * info.z;
    const float* t2 = S.texcoords + 3 * info.w;
    float        width, height;
    if ((double)fabsf(t0[0] - t2[0]) > 0.0001) width = __fdiv_rn(hp.x, t0[0] - t2[0]);
    else if ((double)fabsf(t0[0] - t1[0]) > 0.0001) width = __fdiv_rn(hp.x, t0[0] - t1[0]);
    else width = 0.0f;
    if ((double)fabsf(t0[1] - t2[1]) > 0.0001) height = __fdiv_rn(hp.y, t0[1] - t2[1]);
    else if ((double)fabsf(t0[1] - t1[1]) > 0.0001) height = __fdiv_rn(hp.y, t0[1] - t1[1]);
    else height = 0.0f;
    float u = width, v = height;
    if (u > 0.0f) u = u - (float)(int)u;
    else u = u - (float)(int)u + 1.0f;
    if (v > 0.0f) v = v - (float)(int)v;
    else v = v - (float)(int)v + 1.0f;
    const int4 ti = S.tex_info[tex];
    int        pos = (int)(u * (float)ti.y + (float)((int)(v * (float)ti.z) * ti.y));
    const int  last = (ti.z + 2) * ti.y - 1;
    pos = pos < 0 ? 0 : (pos > last ? last : pos);
    const uint8_t* px = S.texels + (size_t)ti.x + (size_t)pos * 4;
    return mk3(__fdiv_rn((float)px[0], 255.0f), __fdiv_rn((float)px[1], 255.0f),
               __fdiv_rn((float)px[2], 255.0f));
}

struct PendingRay {
    f3    o, d;
    float w;  // path weight of this ray's colour in the camera sample
    float ni; // prev_Ni handed to performRayTracing
    int   k;
    unsigned path; // route from the camera ray, two bits per level (1 mirror, 2 transmitted, 3 reflected)
};

// One camera sample: performRayTracing(origin, dest, k0, prev_ni, minT) of
// raytracing.cpp:61-274, with the recursion unrolled into a list of pending
// rays carrying path weights.
__device__ __forceinline__ void trace_sample(const DScene& S, const DParams& P, f3 origin, f3 dest,
                                             int k0, float prev_ni, uint2* stk, int stride, f3& rgb, float& minT_out,
                                             int& tri_out, unsigned& n_closest, unsigned& n_shadow,
                                             unsigned& n_tests, unsigned& n_nodes, unsigned& n_shaded,
                                             unsigned& n_child, unsigned long long id = 0)
{
    PendingRay pend[RT_MAX_PENDING];
    int        np = 1;
    pend[0].o = origin, pend[0].d = dest, pend[0].w = 1.0f, pend[0].ni = prev_ni, pend[0].k = k0, pend[0].path = 0;
    f3   acc = mk3(0.0f, 0.0f, 0.0f);
    bool first = true;
    const float shadowPart = __fdiv_rn(P.ShadowFactor, (float)P.n_lights); // raytracing.cpp:102
    while (np > 0) {
        const PendingRay r = pend[--np];
        HitRec           hit;
        ++n_closest;
        const bool got = kd_trace<false>(S, r.o, r.d, stk, stride, hit, n_tests, n_nodes);
        int orig = -1;
        if (got) orig = __float_as_int(S.tris[RT_SLOT_F4 * hit.slot + 2].w);
        if (first) {
            minT_out = hit.t;
            tri_out = orig;
            first = false;
        }
        if (!got) { // raytracing.cpp:74
            acc = acc + mk3(P.bg[0], P.bg[1], P.bg[2]) * r.w;
            continue;
        }
        ++n_shaded;
        const f3     dir = r.d - r.o;
        const f3     hp = r.o + dir * hit.t; // raytracing.cpp:384
        const float4 s0 = S.tri_shade[3 * orig], s1 = S.tri_shade[3 * orig + 1],
                     s2 = S.tri_shade[3 * orig + 2];
        const int4   info = S.tri_info[orig];
        const float4 m0 = S.mats[4 * info.x], m1 = S.mats[4 * info.x + 1], m2 = S.mats[4 * info.x + 2];
        const f3     Kd = mk3(m0.x, m0.y, m0.z), Ks = mk3(m1.x, m1.y, m1.z);
        const float  Ns = m0.w, Ni = m1.w;
        const int    illum = __float_as_int(m2.x), has_Ni = __float_as_int(m2.y),
                  tex = __float_as_int(m2.z);
        // raytracing.cpp:83-85
        const float u = hit.a, v = hit.b, w = 1.0f - u - v;
        f3 N = (mk3(s0.x, s0.y, s0.z) * u + mk3(s1.x, s1.y, s1.z) * v) + mk3(s2.x, s2.y, s2.z) * w;
        normalize3(N);
        f3 res;
        if (tex < 0) res = Kd * P.backgroundlighting; // raytracing.cpp:92-95
        else res = sample_texture(S, tex, info, hp);
        float shadowMax = 0.0f; // never reset between lights (raytracing.cpp:104)
        f3    viewDir = r.o - hp;
        normalize3(viewDir);
        for (int li = 0; li < P.n_lights; ++li) { // raytracing.cpp:110-164
            const f3 lpos = mk3(P.lights[3 * li], P.lights[3 * li + 1], P.lights[3 * li + 2]);
            f3       lightDir = lpos - hp;
            normalize3(lightDir);
            float angle = dot3(N, lightDir);
            if (angle > 0.0f) res = res + (Kd * angle) * P.diffusePower;
            angle = dot3(lightDir, N);
            if (angle <= 0.2f) shadowMax = 1.0f;
            else if (angle < 0.5f) {
                angle -= 0.2f;
                shadowMax = 3.33f * (0.3f - angle);
            }
            float softShadow = 0.0f;
            if (angle > 0.0f) {
                f3 halfway = lightDir + viewDir;
                normalize3(halfway);
                angle = dot3(N, halfway);
                if (angle > 0.0f) res = res + Ks * powf(angle, Ns);
                HitRec   sh;
                const f3 so = hp + lightDir * 0.001f;
                if (P.soft_n == 0) {
                    ++n_shadow;
                    if (kd_trace<true>(S, so, lpos, stk, stride, sh, n_tests, n_nodes)) shadowMax = 1.0f;
                } else { // raytracing.cpp:145-160
                    const int ns = P.soft_n * P.soft_n * P.soft_n;
                    for (int si = 0; si < ns; ++si) {
                        f3 sd;
                        soft_dest(&P.lights[0], &P.soft_x[0], P.soft_n, P.soft_half, P.soft_quarter, P.soft_radius, li,
                                  si, id, r.path, sd.x, sd.y, sd.z);
                        ++n_shadow;
                        if (kd_trace<true>(S, so, sd, stk, stride, sh, n_tests, n_nodes)) softShadow += P.soft_w;
                    }
                }
            }
            shadowMax = (shadowMax < softShadow) ? softShadow : shadowMax; // std::max(shadowMax, softShadow)
            const float sm = shadowMax * shadowPart;
            res = res - mk3(sm, sm, sm);
        }
        const int newK = r.k + 1;
        if (newK >= 5) { // raytracing.cpp:166-168
            acc = acc + res * r.w;
            continue;
        }
        const float n1 = (r.ni != 0.0f) ? r.ni : 1.0f;
        const float n2 = has_Ni ? Ni : 1.0f;
        if (illum == 3) { // raytracing.cpp:177-197
            const f3 refl = dir - N * (2.0f * dot3(N, dir));
            acc = acc + res * (0.5f * r.w);
            ++n_child;
            PendingRay& c = pend[np++];
            c.o = hp + refl * 0.001f, c.d = hp + refl, c.w = 0.5f * r.w, c.ni = n2, c.k = newK, c.path = r.path * 4 + 1;
            continue;
        }
        if (Ni > 1.0f) { // raytracing.cpp:206-271
            f3 I = dir;
            normalize3(I);
            const float eta = (n1 != n2) ? n2 : __fdiv_rn(1.0f, n2);
            const float cosI = dot3(I, N);
            // pow(float, int) is evaluated in double there (raytracing.cpp:221)
            const float sin2T =
                (float)(((double)eta * (double)eta) * (1.0 - ((double)cosI * (double)cosI)));
            float rf = sin2T;
            rf = rf < 0.0f ? 0.0f : rf;
            rf = rf > 1.0f ? 1.0f : rf;
            float wLocal = 0.5f * rf;
            if (sin2T <= 1.0f) {
                // sqrt is the double overload in that translation unit (raytracing.cpp:229)
                const float f = (float)((double)(eta * cosI) - sqrt((double)(1.0f - sin2T)));
                const f3    tRay = I * eta + N * f;
                wLocal += 0.1f * (1.0f - rf);
                const float wT = 0.9f * (1.0f - rf) * r.w;
                if (wT != 0.0f) {
                    ++n_child;
                    PendingRay& c = pend[np++];
                    c.o = hp + tRay * 0.001f, c.d = hp + tRay, c.w = wT, c.ni = n2, c.k = newK, c.path = r.path * 4 + 2;
                }
            }
            const float wR = 0.5f * rf * r.w;
            if (wR != 0.0f) {
                const f3    refRay = I - N * (2.0f * dot3(I, N));
                ++n_child;
                PendingRay& c = pend[np++];
                c.o = hp + refRay * 0.001f, c.d = hp + refRay, c.w = wR, c.ni = n2, c.k = newK, c.path = r.path * 4 + 3;
            }
            acc = acc + res * (wLocal * r.w);
            continue;
        }
        acc = acc + res * r.w;
    }
    rgb = acc;
}

// RayTracer::raytrace, main.cpp:514-526: bilinear blend of the corner rays;
// the weights are formed in double and converted to float before they touch
// the vectors (Vec3D.h:12-18).
template <class PT> // PT: DParams, by value or in the constant address space
__device__ __forceinline__ void pixel_ray(const PT& P, double x, double y, f3& o, f3& d)
{
    const double xscale = 1.0 - x / (double)(P.W - 1);
    const double yscale = 1.0 - y / (double)(P.H - 1);
    const float  xs = (float)xscale, xs1 = (float)(1 - xscale), ys = (float)yscale,
                ys1 = (float)(1 - yscale);
    const auto* c = P.cam;
    const f3 o00 = mk3(c[0], c[1], c[2]), d00 = mk3(c[3], c[4], c[5]), o01 = mk3(c[6], c[7], c[8]),
             d01 = mk3(c[9], c[10], c[11]), o10 = mk3(c[12], c[13], c[14]),
             d10 = mk3(c[15], c[16], c[17]), o11 = mk3(c[18], c[19], c[20]),
             d11 = mk3(c[21], c[22], c[23]);
    o = (o00 * xs + o10 * xs1) * ys + (o01 * xs + o11 * xs1) * ys1;
    d = (d00 * xs + d10 * xs1) * ys + (d01 * xs + d11 * xs1) * ys1;
}

// Pixel slot -> pixel.  Slots enumerate this rank's tiles in order (the tiles of its
// shares rank .. rank + span - 1, round after round); inside a
// tile pixels go by 8x8 blocks so that the 64 lanes of a wave cover a compact
// screen area.
template <class PT>
__device__ __forceinline__ bool slot_to_pixel(const PT& P, long long slot64, int& x, int& y)
{
    // a rank's slots number at most (32768 + 1023)^2 < 2^31: one range check, then 32-bit arithmetic
    // (64-bit integer division costs some hundred instructions on this hardware)
    const unsigned tile_px = (unsigned)(P.tile_w * P.tile_h);
    if (slot64 < 0 || slot64 >= (long long)P.n_my_tiles * tile_px) return false;
    const unsigned slot = (unsigned)slot64;
    const unsigned lt = slot / tile_px;
    const unsigned within = slot - lt * tile_px;
    const unsigned span = (unsigned)P.span, tiles_x = (unsigned)P.tiles_x;
    const unsigned round = lt / span;
    const unsigned gt = round * (unsigned)P.world + (unsigned)P.rank + (lt - round * span);
    const unsigned ty = gt / tiles_x, tx = gt - ty * tiles_x;
    const unsigned blk = within >> 6, in = within & 63;
    const unsigned bpr = (unsigned)P.tile_w >> 3; // 8x8 blocks per tile row
    const unsigned by = blk / bpr;
    x = (int)(tx * (unsigned)P.tile_w + (blk - by * bpr) * 8 + (in & 7));
    y = (int)(ty * (unsigned)P.tile_h + by * 8 + (in >> 3));
    return x < P.W && y < P.H;
}
