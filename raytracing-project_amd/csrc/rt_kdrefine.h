// rt_kdrefine.h - the two tree transformations of rt_scene_create on the GPU.
//
// rt_scene_create rearranges the reference's KD tree (KDtree.cpp:15-230) before it uploads it:
// triangles the reference keeps in inner nodes are handed down to every leaf whose cell they
// overlap, and leaves left with many triangles are split further by a surface-area heuristic.
// Both are loops over (node, triangle) pairs with an exact triangle/box test in double
// precision, O(triangles x nodes) on the host.  Here the same steps run level by level on the
// device: one block per node of a level, the per-triangle tests and the per-candidate costs
// spread over its threads.  The arithmetic (kd_overlaps, kd_split_cost) is shared with the host
// path as __host__ __device__ functions and every choice among equal costs follows the host
// loop's order, so the tree that comes out is the host's, node for node and slot for slot
// (RT_KD_GPU=0 selects the host path; a GPU test compares the two).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct KdBox {
    double lo[3], hi[3];
};

// Separating-axis test triangle vs box, the box grown a little: conservative.
__host__ __device__ inline bool kd_overlaps(const float* vpos, const int32_t* tri_vertex, int t, const KdBox& b)
{
    double c[3], h[3], v[3][3];
    for (int a = 0; a < 3; ++a) {
        c[a] = 0.5 * (b.lo[a] + b.hi[a]);
        h[a] = 0.5 * (b.hi[a] - b.lo[a]);
        h[a] += 1e-5 * fabs(h[a]) + 1e-5 * (fabs(c[a]) + 1.0);
    }
    for (int j = 0; j < 3; ++j)
        for (int a = 0; a < 3; ++a) v[j][a] = (double)vpos[3 * tri_vertex[3 * t + j] + a] - c[a];
    for (int a = 0; a < 3; ++a) {
        const double mn = fmin(v[0][a], fmin(v[1][a], v[2][a]));
        const double mx = fmax(v[0][a], fmax(v[1][a], v[2][a]));
        if (mn > h[a] || mx < -h[a]) return false;
    }
    double e[3][3];
    for (int a = 0; a < 3; ++a) {
        e[0][a] = v[1][a] - v[0][a];
        e[1][a] = v[2][a] - v[1][a];
        e[2][a] = v[0][a] - v[2][a];
    }
    for (int i = 0; i < 3; ++i)
        for (int a = 0; a < 3; ++a) { // axis = unit(a) x e[i]
            double    ax[3] = {0, 0, 0};
            const int a1 = (a + 1) % 3, a2 = (a + 2) % 3;
            ax[a1] = -e[i][a2], ax[a2] = e[i][a1];
            const double p0 = ax[0] * v[0][0] + ax[1] * v[0][1] + ax[2] * v[0][2];
            const double p1 = ax[0] * v[1][0] + ax[1] * v[1][1] + ax[2] * v[1][2];
            const double p2 = ax[0] * v[2][0] + ax[1] * v[2][1] + ax[2] * v[2][2];
            const double r = h[0] * fabs(ax[0]) + h[1] * fabs(ax[1]) + h[2] * fabs(ax[2]);
            const double mn = fmin(p0, fmin(p1, p2)), mx = fmax(p0, fmax(p1, p2));
            const double slack = 1e-9 * (r + fabs(mn) + fabs(mx)) + 1e-300;
            if (mn > r + slack || mx < -r - slack) return false;
        }
    double nrm[3] = {e[0][1] * e[1][2] - e[0][2] * e[1][1], e[0][2] * e[1][0] - e[0][0] * e[1][2],
                     e[0][0] * e[1][1] - e[0][1] * e[1][0]};
    const double d = nrm[0] * v[0][0] + nrm[1] * v[0][1] + nrm[2] * v[0][2];
    const double r = h[0] * fabs(nrm[0]) + h[1] * fabs(nrm[1]) + h[2] * fabs(nrm[2]);
    return fabs(d) <= r + 1e-9 * (r + fabs(d));
}

__host__ __device__ inline double kd_area(const KdBox& b)
{
    const double dx = fmax(b.hi[0] - b.lo[0], 0.0), dy = fmax(b.hi[1] - b.lo[1], 0.0), dz = fmax(b.hi[2] - b.lo[2], 0.0);
    return 2.0 * (dx * dy + dy * dz + dz * dx);
}

// extent of triangle t on axis a
__host__ __device__ inline void kd_tri_extent(const float* vpos, const int32_t* tri_vertex, int t, int a, double& mn,
                                              double& mx)
{
    mn = 1e300, mx = -1e300;
    for (int j = 0; j < 3; ++j) {
        const double c = vpos[3 * tri_vertex[3 * t + j] + a];
        mn = fmin(mn, c), mx = fmax(mx, c);
    }
}

// Cost of splitting the cell part `cb` (area sa) at plane `pl` of axis a, for the n triangles of
// `list`; false when the plane, rounded to the float the kernels compare against, is not
// strictly inside cb.
__host__ __device__ inline bool kd_split_cost(const float* vpos, const int32_t* tri_vertex, const int32_t* list, int n,
                                              const KdBox& cb, double sa, int a, double pl, double c_trav,
                                              double c_isect, double c_empty, double& cost, double& plane)
{
    const float plf = (float)pl;
    if (!((double)plf > cb.lo[a] && (double)plf < cb.hi[a])) return false;
    size_t nl = 0, nr = 0;
    for (int i = 0; i < n; ++i) {
        double mn, mx;
        kd_tri_extent(vpos, tri_vertex, list[i], a, mn, mx);
        if (mn < (double)plf || (mn == (double)plf && mx == (double)plf)) ++nl;
        if (mx > (double)plf) ++nr;
    }
    KdBox bl = cb, br = cb;
    bl.hi[a] = plf, br.lo[a] = plf;
    cost = c_trav + c_isect * (kd_area(bl) * (double)nl + kd_area(br) * (double)nr) / sa;
    if (nl == 0 || nr == 0) cost *= c_empty;
    plane = (double)plf;
    return true;
}

// ---- device side: work items in, results out; the host keeps the node arrays ----
struct KdHandItem {    // an inner node of the input tree: hand its list down to its two children
    KdBox   left, right; // the children's cells
    int32_t off, cnt;    // the node's list in the pool (its own triangles + what was handed down to it)
    int32_t own_off[2], own_cnt[2]; // the children's own lists
};
struct KdHandResult {
    int32_t off[2], cnt[2]; // the children's new lists: own triangles first, then the ones handed down
};
struct KdSplitItem { // a leaf: split it if that pays
    KdBox   cell;
    int32_t off, cnt;
};
struct KdSplitResult {
    int32_t axis;  // -1: left as it is
    float   plane;
    int32_t off[2], cnt[2];
};
struct KdSplitParams {
    KdBox  mesh_box;
    double c_trav, c_isect, c_empty;
    int    leaf_target;
};

#define KD_BLOCK 256

// order-preserving append of the flagged elements of one chunk (one element per thread);
// returns the position for this thread (valid when flag) and advances *base by the chunk's count
__device__ __forceinline__ int kd_block_scan(bool flag, int* s_wave, int& base)
{
    const unsigned long long m = __ballot(flag);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < KD_BLOCK / 64; ++w) {
        if (w < wave) before += s_wave[w];
        total += s_wave[w];
    }
    const int pos = base + before + __popcll(m & ((1ull << lane) - 1ull));
    base += total;
    __syncthreads();
    return pos;
}

// pool: int32 triangle ids; used: next free entry; a list that does not fit sets *overflow
__device__ __forceinline__ int kd_pool_alloc(int n, int* used, int pool_cap, int* overflow, int* s_off)
{
    if (threadIdx.x == 0) {
        int o = atomicAdd(used, n);
        if (o + n > pool_cap) *overflow = 1, o = -1;
        *s_off = o;
    }
    __syncthreads();
    const int o = *s_off;
    __syncthreads();
    return o;
}

__global__ void __launch_bounds__(KD_BLOCK)
k_kd_hand_down(const float* __restrict__ vpos, const int32_t* __restrict__ tri_vertex, const KdHandItem* items,
               KdHandResult* results, int32_t* pool, int* used, int pool_cap, int* overflow)
{
    __shared__ int   s_wave[KD_BLOCK / 64], s_off;
    const KdHandItem it = items[blockIdx.x];
    KdHandResult     res;
    for (int side = 0; side < 2; ++side) {
        const KdBox& mine = side == 0 ? it.left : it.right;
        const KdBox& other = side == 0 ? it.right : it.left;
        // count, then allocate own + handed down, then write in list order
        int n_in = 0;
        for (int i0 = 0; i0 < it.cnt; i0 += KD_BLOCK) {
            const int i = i0 + threadIdx.x;
            bool      f = false;
            if (i < it.cnt) {
                const int  t = pool[it.off + i];
                const bool in_me = kd_overlaps(vpos, tri_vertex, t, mine);
                f = in_me || !kd_overlaps(vpos, tri_vertex, t, other);
            }
            kd_block_scan(f, s_wave, n_in);
        }
        const int total = it.own_cnt[side] + n_in;
        const int o = kd_pool_alloc(total, used, pool_cap, overflow, &s_off);
        res.off[side] = o, res.cnt[side] = o < 0 ? 0 : total;
        if (o < 0) continue;
        for (int i = threadIdx.x; i < it.own_cnt[side]; i += KD_BLOCK) pool[o + i] = pool[it.own_off[side] + i];
        int base = o + it.own_cnt[side];
        for (int i0 = 0; i0 < it.cnt; i0 += KD_BLOCK) {
            const int i = i0 + threadIdx.x;
            bool      f = false;
            int       t = 0;
            if (i < it.cnt) {
                t = pool[it.off + i];
                const bool in_me = kd_overlaps(vpos, tri_vertex, t, mine);
                f = in_me || !kd_overlaps(vpos, tri_vertex, t, other);
            }
            const int pos = kd_block_scan(f, s_wave, base);
            if (f) pool[pos] = t;
        }
    }
    if (threadIdx.x == 0) results[blockIdx.x] = res;
}

__global__ void __launch_bounds__(KD_BLOCK)
k_kd_split(const float* __restrict__ vpos, const int32_t* __restrict__ tri_vertex, const KdSplitItem* items,
           KdSplitResult* results, KdSplitParams prm, int32_t* pool, int* used, int pool_cap, int* overflow)
{
    __shared__ int    s_wave[KD_BLOCK / 64], s_off, s_best_c[KD_BLOCK];
    __shared__ double s_best[KD_BLOCK], s_plane[KD_BLOCK];
    const KdSplitItem it = items[blockIdx.x];
    KdSplitResult     res;
    res.axis = -1, res.plane = 0.0f, res.off[0] = res.off[1] = -1, res.cnt[0] = res.cnt[1] = 0;
    const int      n = it.cnt;
    const int32_t* list = pool + it.off;
    KdBox          cb = it.cell; // the part of the cell rays can come from
    for (int a = 0; a < 3; ++a) {
        cb.lo[a] = fmax(cb.lo[a], prm.mesh_box.lo[a]);
        cb.hi[a] = fmin(cb.hi[a], prm.mesh_box.hi[a]);
    }
    const double sa = kd_area(cb);
    bool         go = n > prm.leaf_target && sa > 0.0;
    // candidates in the host loop's order: axis 0, 1, 2; per axis the triangles in list order, each
    // its lower then its upper extent; the first candidate of the lowest cost wins
    double best = prm.c_isect * (double)n, best_plane = 0.0;
    int    best_c = 0x7fffffff;
    if (go) {
        for (int c = threadIdx.x; c < 6 * n; c += KD_BLOCK) {
            const int a = c / (2 * n), i = (c - a * 2 * n) >> 1, upper = c & 1;
            double    mn, mx;
            kd_tri_extent(vpos, tri_vertex, list[i], a, mn, mx);
            const double pl = upper ? mx : mn;
            if (!(pl > cb.lo[a] && pl < cb.hi[a])) continue;
            double cost, plane;
            if (!kd_split_cost(vpos, tri_vertex, list, n, cb, sa, a, pl, prm.c_trav, prm.c_isect, prm.c_empty, cost, plane)) continue;
            if (cost < best) best = cost, best_c = c, best_plane = plane; // c ascends: ties keep the earlier one
        }
    }
    s_best[threadIdx.x] = best, s_best_c[threadIdx.x] = best_c, s_plane[threadIdx.x] = best_plane;
    __syncthreads();
    for (int s = KD_BLOCK / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const double ob = s_best[threadIdx.x + s];
            const int    oc = s_best_c[threadIdx.x + s];
            if (ob < s_best[threadIdx.x] || (ob == s_best[threadIdx.x] && oc < s_best_c[threadIdx.x]))
                s_best[threadIdx.x] = ob, s_best_c[threadIdx.x] = oc, s_plane[threadIdx.x] = s_plane[threadIdx.x + s];
        }
        __syncthreads();
    }
    best_c = s_best_c[0];
    best_plane = s_plane[0];
    __syncthreads();
    go = go && best_c != 0x7fffffff;
    if (go) {
        const int axis = best_c / (2 * n);
        KdBox     b[2] = {it.cell, it.cell};
        b[0].hi[axis] = best_plane, b[1].lo[axis] = best_plane;
        int cnt[2] = {0, 0};
        for (int side = 0; side < 2; ++side)
            for (int i0 = 0; i0 < n; i0 += KD_BLOCK) {
                const int i = i0 + threadIdx.x;
                bool      f = false;
                if (i < n) {
                    const bool in_me = kd_overlaps(vpos, tri_vertex, list[i], b[side]);
                    f = in_me || !kd_overlaps(vpos, tri_vertex, list[i], b[1 - side]);
                }
                kd_block_scan(f, s_wave, cnt[side]);
            }
        if (!(cnt[0] == n && cnt[1] == n)) { // something was separated
            res.axis = axis, res.plane = (float)best_plane;
            for (int side = 0; side < 2; ++side) {
                const int o = kd_pool_alloc(cnt[side], used, pool_cap, overflow, &s_off);
                res.off[side] = o, res.cnt[side] = o < 0 ? 0 : cnt[side];
                if (o < 0) continue;
                int base = o;
                for (int i0 = 0; i0 < n; i0 += KD_BLOCK) {
                    const int i = i0 + threadIdx.x;
                    bool      f = false;
                    if (i < n) {
                        const bool in_me = kd_overlaps(vpos, tri_vertex, list[i], b[side]);
                        f = in_me || !kd_overlaps(vpos, tri_vertex, list[i], b[1 - side]);
                    }
                    const int pos = kd_block_scan(f, s_wave, base);
                    if (f) pool[pos] = list[i];
                }
            }
        }
    }
    if (threadIdx.x == 0) results[blockIdx.x] = res;
}
