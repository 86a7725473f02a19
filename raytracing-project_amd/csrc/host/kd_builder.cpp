// kd_builder.cpp - builds the KD tree the reference builds (KDtree.cpp:15-230)
// as an index-linked node array.
//
// Rules taken from the reference, because the tree decides traversal cost and
// (through triangle order) hit ties:
//   * root box = mesh bounds grown by 5.1 on every side (KDtree.cpp:60);
//   * the split axis cycles x, y, z with depth; a node becomes a leaf below
//     11 triangles or when its extent on the axis is under 1e-4 (KDtree.cpp:84-100);
//   * 999 candidate planes at start + extent*i/1000; cost =
//     both*extent*20/depth + left*leftExtent + right*rightExtent, the first
//     minimum wins (KDtree.cpp:177-227).  At depth 0 the division by the depth
//     makes every cost inf or NaN, nothing beats FLT_MAX and the plane stays
//     at 0.0 (KDtree.cpp:174,219);
//   * triangles with all three vertices strictly below the plane go left,
//     strictly above go right, the rest stay in the node (KDtree.cpp:136-152).
// The candidate scan counts with two sorted arrays (largest / smallest vertex
// coordinate per triangle) instead of re-classifying every triangle per plane.
#include <algorithm>
#include <cfloat>
#include <cstdio>
#include <cstdlib>

#include "rt_host.hpp"

namespace rtamd {

namespace {

struct Work {
    int                  node;
    int                  depth;
    std::vector<int32_t> tris;
};

float choosePlane(const Mesh& mesh, const std::vector<int32_t>& tris, float lo, float hi, int axis,
                  int depth, std::vector<float>& tmax, std::vector<float>& tmin)
{
    tmax.clear(), tmin.clear();
    for (int32_t t : tris) {
        const float a = mesh.positions[3 * mesh.tri_vertex[3 * t] + axis];
        const float b = mesh.positions[3 * mesh.tri_vertex[3 * t + 1] + axis];
        const float c = mesh.positions[3 * mesh.tri_vertex[3 * t + 2] + axis];
        tmax.push_back(std::max(a, std::max(b, c)));
        tmin.push_back(std::min(a, std::min(b, c)));
    }
    std::sort(tmax.begin(), tmax.end());
    std::sort(tmin.begin(), tmin.end());
    const size_t n = tris.size();
    float        best = 0.0f, bestCost = FLT_MAX;
    const float  accuracy = 1000.0f; // KDtree.cpp:12
    for (int i = 1; i < accuracy; ++i) {
        const float  mid = ((hi - lo) * i / accuracy) + lo;
        const size_t left = (size_t)(std::lower_bound(tmax.begin(), tmax.end(), mid) - tmax.begin());
        const size_t right = n - (size_t)(std::upper_bound(tmin.begin(), tmin.end(), mid) - tmin.begin());
        const size_t both = n - left - right;
        const float  leftSize = mid - lo, rightSize = hi - mid, totalSize = hi - lo;
        const float  cost = both * totalSize * 20.0f / depth + left * leftSize + right * rightSize;
        if (cost < bestCost) {
            bestCost = cost;
            best = mid;
        }
    }
    return best;
}

} // namespace

void buildKDtree(const Mesh& mesh, KdTree& tree)
{
    tree = KdTree();
    const size_t nt = mesh.triangleCount();
    if (nt == 0) return;
    // RT_KD_BUILD_GPU=1: the same tree from the device (rt_build_kd_tree: one block per node of a level, the
    // 999 candidate planes of a node spread over its threads).  The loops below stay the default: the
    // reference's trees take them a few milliseconds, and a test holds the two against each other.
    if (const char* e = std::getenv("RT_KD_BUILD_GPU"); e && e[0] == '1') {
        const int32_t cap = 4 * (int32_t)nt + 64;
        tree.nodes.resize((size_t)cap);
        tree.node_tris.resize(nt);
        int32_t n_nodes = 0, depth = 0;
        const int rc = rt_build_kd_tree(0, (int32_t)(mesh.positions.size() / 3), mesh.positions.data(), (int32_t)nt,
                                        mesh.tri_vertex.data(), cap, tree.nodes.data(), &n_nodes, tree.node_tris.data(), &depth);
        if (rc == RT_OK) {
            tree.nodes.resize((size_t)n_nodes);
            tree.depth = depth;
            return;
        }
        std::fprintf(stderr, "buildKDtree: RT_KD_BUILD_GPU=1, but %s - building on the host\n", rt_last_error());
        tree = KdTree();
    }
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) lo[a] = hi[a] = mesh.positions[3 * mesh.tri_vertex[0] + a];
    for (size_t t = 0; t < nt; ++t)
        for (int j = 0; j < 3; ++j)
            for (int a = 0; a < 3; ++a) {
                const float c = mesh.positions[3 * mesh.tri_vertex[3 * t + j] + a];
                if (c < lo[a]) lo[a] = c;
                else if (c > hi[a]) hi[a] = c;
            }
    RtKdNode root{};
    for (int a = 0; a < 3; ++a) root.bounds[2 * a] = lo[a] - 5.1f, root.bounds[2 * a + 1] = hi[a] + 5.1f;
    root.left = root.right = -1;
    tree.nodes.push_back(root);

    std::vector<std::vector<int32_t>> kept(1); // triangles that stay in each node
    std::vector<Work>                 todo;
    {
        Work w{0, 0, {}};
        w.tris.resize(nt);
        for (size_t t = 0; t < nt; ++t) w.tris[t] = (int32_t)t;
        todo.push_back(std::move(w));
    }
    std::vector<float> tmax, tmin;
    while (!todo.empty()) {
        Work w = std::move(todo.back());
        todo.pop_back();
        tree.depth = std::max(tree.depth, w.depth);
        const int   axis = w.depth % 3;
        const float a0 = tree.nodes[w.node].bounds[2 * axis], a1 = tree.nodes[w.node].bounds[2 * axis + 1];
        if (w.tris.size() < 11 || (double)(a1 - a0) < 0.0001) {
            kept[w.node] = std::move(w.tris);
            continue;
        }
        const float mid = choosePlane(mesh, w.tris, a0, a1, axis, w.depth, tmax, tmin);
        Work        l{0, w.depth + 1, {}}, r{0, w.depth + 1, {}};
        for (int32_t t : w.tris) {
            const float a = mesh.positions[3 * mesh.tri_vertex[3 * t] + axis];
            const float b = mesh.positions[3 * mesh.tri_vertex[3 * t + 1] + axis];
            const float c = mesh.positions[3 * mesh.tri_vertex[3 * t + 2] + axis];
            if (a < mid && b < mid && c < mid) l.tris.push_back(t);
            else if (a > mid && b > mid && c > mid) r.tris.push_back(t);
            else kept[w.node].push_back(t);
        }
        RtKdNode ln = tree.nodes[w.node], rn = tree.nodes[w.node];
        ln.bounds[2 * axis + 1] = mid;
        rn.bounds[2 * axis] = mid;
        ln.left = ln.right = rn.left = rn.right = -1;
        l.node = (int)tree.nodes.size();
        r.node = l.node + 1;
        tree.nodes[w.node].left = l.node;
        tree.nodes[w.node].right = r.node;
        tree.nodes.push_back(ln);
        tree.nodes.push_back(rn);
        kept.emplace_back();
        kept.emplace_back();
        todo.push_back(std::move(r)); // left subtree is expanded first
        todo.push_back(std::move(l));
    }
    for (size_t n = 0; n < tree.nodes.size(); ++n) {
        tree.nodes[n].tri_offset = (int32_t)tree.node_tris.size();
        tree.nodes[n].tri_count = (int32_t)kept[n].size();
        tree.node_tris.insert(tree.node_tris.end(), kept[n].begin(), kept[n].end());
    }
}

} // namespace rtamd
