// rt_kdbuild.h - the reference's KD tree built on the GPU.
//
// buildKDtree / splitSpaces / calcBestSplit (KDtree.cpp:15-230) as the reference runs them: the root box is
// the mesh's bounds grown by 5.1, the split axis cycles x, y, z with the depth, a node with fewer than 11
// triangles or an extent under 1e-4 on its axis is a leaf, and an inner node tries 999 planes at
// start + extent * i / 1000 with the cost
//     both * extent * 20 / depth + left * leftExtent + right * rightExtent      (first minimum wins)
// where left / right count the triangles strictly below / above the plane and `both` the rest, which stay
// in the node.  At depth 0 the division by the depth makes every cost inf or NaN, nothing beats FLT_MAX and
// the plane stays at 0.0 (KDtree.cpp:174, 219).
//
// The reference recurses; here the tree grows level by level, one block per node of the level:
//   * the largest and smallest coordinate of each of the node's triangles on its axis go through LDS in
//     chunks, and every thread counts left and right for its four candidate planes as the chunks pass
//     (999 x n comparisons per node - the host path sorts the two arrays and searches them instead; here
//     the comparisons are spread over 256 lanes and the node's triangles are read once);
//   * cost and plane are formed in the reference's operation order in float, the block keeps the smallest
//     cost below FLT_MAX, the lowest candidate among equal ones: what the sequential scan ends with;
//   * the triangles are dealt to the left child, the right child or the node itself with an
//     order-preserving block scan, so every list keeps the order the reference's loop gives it.
// Children get provisional numbers in the order blocks happen to ask for them; rt_build_kd_tree renumbers
// the finished tree on the host in the order the reference's recursion creates nodes (both children when
// their parent is expanded, left subtree first).  The result is the tree csrc/host/kd_builder.cpp builds, node
// for node and list for list - and that one is pinned to the reference's own (tests/golden/*.npz).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

#define KDB_BLOCK 256
#define KDB_CHUNK 2048 // triangles whose extremes sit in LDS at a time
#define KDB_CAND 4     // candidate planes per thread: 4 x 256 >= 999

struct KdbItem {
    int32_t node, depth, off, cnt; // provisional node id, its depth, its triangles in the level's list
};

struct KdbNode {
    float   b[6];
    int32_t left, right; // provisional ids, -1 for a leaf
    int32_t kept_off, kept_cnt;
    int32_t depth, pad;
};

struct KdbCtl {
    int32_t n_nodes, n_next, list_used, kept_used, overflow, pad[3];
};

__device__ __forceinline__ void kdb_extremes(const float* __restrict__ vpos, const int32_t* __restrict__ tri_vertex,
                                             int t, int axis, float& mn, float& mx)
{
    const float a = vpos[3 * tri_vertex[3 * t] + axis], b = vpos[3 * tri_vertex[3 * t + 1] + axis],
                c = vpos[3 * tri_vertex[3 * t + 2] + axis];
    mx = fmaxf(a, fmaxf(b, c));
    mn = fminf(a, fminf(b, c));
}

__global__ void __launch_bounds__(KDB_BLOCK)
k_kd_build_level(const float* __restrict__ vpos, const int32_t* __restrict__ tri_vertex,
                 const KdbItem* __restrict__ items, const int32_t* __restrict__ list_in, int32_t* __restrict__ list_out,
                 int32_t* __restrict__ kept, KdbNode* __restrict__ nodes, KdbItem* __restrict__ next, KdbCtl* ctl,
                 int max_nodes, int max_next, int list_cap)
{
    __shared__ float    s_max[KDB_CHUNK], s_min[KDB_CHUNK];
    __shared__ float    s_cost[KDB_BLOCK];
    __shared__ int      s_cand[KDB_BLOCK];
    __shared__ int      s_cnt[3][KDB_BLOCK / 64]; // per wave: triangles going left, right, staying
    __shared__ int      s_base[4];                // where the three lists go; [3]: the children's first id
    const KdbItem it = items[blockIdx.x];
    const int     tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int     axis = it.depth % 3;
    KdbNode*      nd = nodes + it.node;
    const float   lo = nd->b[2 * axis], hi = nd->b[2 * axis + 1];
    const int     n = it.cnt;
    const int32_t* tris = list_in + it.off;
    if (n < 11 || (double)(hi - lo) < 0.0001) { // KDtree.cpp:84-100: a leaf keeps its list
        if (tid == 0) s_base[2] = atomicAdd(&ctl->kept_used, n);
        __syncthreads();
        const int k0 = s_base[2];
        for (int i = tid; i < n; i += KDB_BLOCK) kept[k0 + i] = tris[i];
        if (tid == 0) nd->kept_off = k0, nd->kept_cnt = n, nd->left = -1, nd->right = -1;
        return;
    }
    // ---- calcBestSplit, KDtree.cpp:171-230 ----
    const float accuracy = 1000.0f;
    float       mid[KDB_CAND];
    unsigned    left[KDB_CAND], right[KDB_CAND];
    for (int j = 0; j < KDB_CAND; ++j) {
        const int i = tid + 1 + KDB_BLOCK * j;
        mid[j] = __fdiv_rn((hi - lo) * (float)i, accuracy) + lo;
        left[j] = right[j] = 0;
    }
    for (int c0 = 0; c0 < n; c0 += KDB_CHUNK) {
        const int m = min(KDB_CHUNK, n - c0);
        __syncthreads();
        for (int i = tid; i < m; i += KDB_BLOCK) kdb_extremes(vpos, tri_vertex, tris[c0 + i], axis, s_min[i], s_max[i]);
        __syncthreads();
        for (int i = 0; i < m; ++i) {
            const float mx = s_max[i], mn = s_min[i];
            for (int j = 0; j < KDB_CAND; ++j) left[j] += mx < mid[j] ? 1u : 0u, right[j] += mn > mid[j] ? 1u : 0u;
        }
    }
    float best_cost = 3.402823466e+38f; // FLT_MAX: a candidate has to be below it
    int   best_i = 0x7fffffff;
    float best_mid = 0.0f;
    for (int j = 0; j < KDB_CAND; ++j) {
        const int i = tid + 1 + KDB_BLOCK * j;
        if (i >= 1000) break;
        const unsigned both = (unsigned)n - left[j] - right[j];
        const float    leftSize = mid[j] - lo, rightSize = hi - mid[j], totalSize = hi - lo;
        const float    cost = __fdiv_rn((float)both * totalSize * 20.0f, (float)it.depth) + (float)left[j] * leftSize +
                           (float)right[j] * rightSize;
        if (cost < best_cost) best_cost = cost, best_i = i, best_mid = mid[j]; // (ascending i: the first minimum stays)
    }
    s_cost[tid] = best_cost, s_cand[tid] = best_i;
    __syncthreads();
    for (int s = KDB_BLOCK / 2; s > 0; s >>= 1) {
        if (tid < s) {
            const float c2 = s_cost[tid + s];
            const int   i2 = s_cand[tid + s];
            if (c2 < s_cost[tid] || (c2 == s_cost[tid] && i2 < s_cand[tid])) s_cost[tid] = c2, s_cand[tid] = i2;
        }
        __syncthreads();
    }
    const int   win = s_cand[0];
    const float plane = win == 0x7fffffff ? 0.0f : __fdiv_rn((hi - lo) * (float)win, accuracy) + lo;
    (void)best_mid;
    // ---- the three lists, KDtree.cpp:136-152, in the order of the node's list ----
    auto side = [&](int t) { // 0 left, 1 right, 2 stays
        const float a = vpos[3 * tri_vertex[3 * t] + axis], b = vpos[3 * tri_vertex[3 * t + 1] + axis],
                    c = vpos[3 * tri_vertex[3 * t + 2] + axis];
        if (a < plane && b < plane && c < plane) return 0;
        if (a > plane && b > plane && c > plane) return 1;
        return 2;
    };
    int tot[3] = {0, 0, 0};
    for (int i0 = 0; i0 < n; i0 += KDB_BLOCK) { // first the sizes
        const int i = i0 + tid;
        const int sd = i < n ? side(tris[i]) : -1;
        for (int k = 0; k < 3; ++k) tot[k] += __popcll(__ballot(sd == k));
    }
    for (int k = 0; k < 3; ++k) {
        if (lane == 0) s_cnt[k][wave] = tot[k];
    }
    __syncthreads();
    int total[3];
    for (int k = 0; k < 3; ++k) {
        total[k] = 0;
        for (int w = 0; w < KDB_BLOCK / 64; ++w) total[k] += s_cnt[k][w];
    }
    if (tid == 0) {
        const int l0 = atomicAdd(&ctl->list_used, total[0] + total[1]);
        const int k0 = atomicAdd(&ctl->kept_used, total[2]);
        const int c0 = atomicAdd(&ctl->n_nodes, 2);
        const int q0 = atomicAdd(&ctl->n_next, 2);
        const bool ok = l0 + total[0] + total[1] <= list_cap && c0 + 2 <= max_nodes && q0 + 2 <= max_next;
        if (!ok) atomicExch(&ctl->overflow, 1);
        s_base[0] = ok ? l0 : -1, s_base[1] = l0 + total[0], s_base[2] = k0, s_base[3] = c0;
        if (ok) {
            KdbNode cl = *nd, cr = *nd;
            cl.b[2 * axis + 1] = plane, cr.b[2 * axis] = plane;
            cl.left = cl.right = cr.left = cr.right = -1;
            cl.kept_off = cr.kept_off = 0, cl.kept_cnt = cr.kept_cnt = 0;
            cl.depth = cr.depth = it.depth + 1;
            nodes[c0] = cl, nodes[c0 + 1] = cr;
            next[q0] = KdbItem{c0, it.depth + 1, l0, total[0]};
            next[q0 + 1] = KdbItem{c0 + 1, it.depth + 1, l0 + total[0], total[1]};
            nd->left = c0, nd->right = c0 + 1, nd->kept_off = k0, nd->kept_cnt = total[2];
        }
    }
    __syncthreads();
    if (s_base[0] < 0) return;
    int run[3] = {0, 0, 0}; // placed so far, per list
    for (int i0 = 0; i0 < n; i0 += KDB_BLOCK) {
        const int i = i0 + tid;
        const int t = i < n ? tris[i] : -1;
        const int sd = i < n ? side(t) : -1;
        unsigned long long m[3];
        for (int k = 0; k < 3; ++k) m[k] = __ballot(sd == k);
        __syncthreads();
        if (lane == 0)
            for (int k = 0; k < 3; ++k) s_cnt[k][wave] = __popcll(m[k]);
        __syncthreads();
        if (sd >= 0) {
            int before = run[sd] + __popcll(m[sd] & ((1ull << lane) - 1ull));
            for (int w = 0; w < wave; ++w) before += s_cnt[sd][w];
            if (sd == 2) kept[s_base[2] + before] = t;
            else list_out[s_base[sd] + before] = t;
        }
        for (int k = 0; k < 3; ++k)
            for (int w = 0; w < KDB_BLOCK / 64; ++w) run[k] += s_cnt[k][w];
    }
}
