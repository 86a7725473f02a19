/*
 * ref_harness.cpp - drives the REAL reference sources (compiled where they lie
 * under /root/reference by oracle/Makefile, target "ref") and dumps their
 * answers so the CPU restatement in rt_oracle.c can be pinned against them.
 * TEST INFRASTRUCTURE, build container only: /root/reference does not exist
 * where the GPU tests run, which use the committed tests/golden/ vectors made
 * from this probe's output by tools/make_golden.py.
 *
 * What runs from the reference, unmodified: Mesh::computeVertexNormals
 * (mesh.cpp:32), buildKDtree/splitSpaces/calcBestSplit (KDtree.cpp:15-230),
 * findChildNode/findNextNode/contains (KDtree.cpp:235-605), checkHit and
 * performRayTracing (raytracing.cpp:61-386) with Vec3D.h/Matrix33.h.
 * What does not: Mesh::loadMesh (needs Corona), so the mesh is handed over
 * through the public members of Mesh/Material; main.cpp (GLUT) and
 * imageWriter.h (std::powf) are not compiled.
 *
 * Built as _ref/libref.so; _ref/refprobe (ref_driver.c) loads it with lazy
 * binding, because Mesh::loadMesh's call into Corona stays an undefined symbol.
 *
 * usage: refprobe <scene.bin> <rays.bin> <out.bin>
 */
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "config.h"
#include "raytracing.h"
#include "settings.h"

// KDtree.cpp is compiled as part of this unit so that its inline walk helpers
// get an out-of-line definition raytracing.cpp can link against.
#include REF_KDTREE_CPP

// the globals main.cpp:27-36 defines
Vec3Df MyCameraPosition;
std::vector<Vec3Df> MyLightPositions;
Mesh MyMesh;

extern float focusDepth;

typedef AccelTreeNode (*fcn_t)(const AccelTreeNode &, int, const Vec3Df &);
typedef AccelTreeNode (*fnn_t)(const AccelTreeNode &, const Vec3Df &, const Vec3Df &);
typedef bool (*cont_t)(const std::vector<AccelTreeNode *> &, const AccelTreeNode &);
void *keep_inline[] = {(void *)(fcn_t)&findChildNode, (void *)(fnn_t)&findNextNode,
                       (void *)(cont_t)&contains};

static void rd(FILE *f, void *p, size_t n)
{
    if (fread(p, 1, n, f) != n) {
        fprintf(stderr, "short read\n");
        exit(2);
    }
}

static void dumpTree(FILE *out, const AccelTreeNode &n)
{
    float b[6] = {n.xStart, n.xEnd, n.yStart, n.yEnd, n.zStart, n.zEnd};
    fwrite(b, 4, 6, out);
    int leaf = n.leftChild == nullptr, cnt = (int)n.triangles.size();
    fwrite(&leaf, 4, 1, out);
    fwrite(&cnt, 4, 1, out);
    if (cnt) fwrite(n.triangles.data(), 4, cnt, out);
    if (!leaf) {
        dumpTree(out, *n.leftChild);
        dumpTree(out, *n.rightChild);
    }
}

extern "C" int ref_main(int argc, char **argv)
{
    if (argc < 4) return 1;
    FILE *fs = fopen(argv[1], "rb");
    FILE *fr = fopen(argv[2], "rb");
    FILE *out = fopen(argv[3], "wb");
    if (!fs || !fr || !out) return 1;

    int nv, nt, ntc, nm, nl;
    rd(fs, &nv, 4);
    std::vector<float> vpos(3 * nv);
    rd(fs, vpos.data(), 12 * nv);
    rd(fs, &nt, 4);
    std::vector<int> tv(3 * nt), tt(3 * nt), tm(nt);
    rd(fs, tv.data(), 12 * nt);
    rd(fs, tt.data(), 12 * nt);
    rd(fs, tm.data(), 4 * nt);
    rd(fs, &ntc, 4);
    std::vector<float> tc(3 * ntc);
    if (ntc) rd(fs, tc.data(), 12 * ntc);
    rd(fs, &nm, 4);
    std::vector<float> mats(16 * nm);
    rd(fs, mats.data(), 64 * nm);
    rd(fs, &nl, 4);
    std::vector<float> lights(3 * nl);
    rd(fs, lights.data(), 12 * nl);
    fclose(fs);

    for (int i = 0; i < nv; ++i)
        MyMesh.vertices.push_back(Vertex(Vec3Df(vpos[3 * i], vpos[3 * i + 1], vpos[3 * i + 2])));
    for (int i = 0; i < ntc; ++i)
        MyMesh.texcoords.push_back(Vec3Df(tc[3 * i], tc[3 * i + 1], tc[3 * i + 2]));
    for (int i = 0; i < nt; ++i) {
        MyMesh.triangles.push_back(Triangle(tv[3 * i], tt[3 * i], tv[3 * i + 1], tt[3 * i + 1],
                                            tv[3 * i + 2], tt[3 * i + 2]));
        MyMesh.triangleMaterials.push_back(tm[i]);
    }
    for (int i = 0; i < nm; ++i) {
        const float *r = &mats[16 * i];
        Material m;
        m.set_Ni(r[7]);  // value first, then drop the flag if the .mtl never set it
        m.cleanup();
        m.set_Kd(r[0], r[1], r[2]);
        m.set_Ks(r[3], r[4], r[5]);
        m.set_Ns(r[6]);
        m.set_illum((int)r[8]);
        if (r[9] != 0.0f) m.set_Ni(r[7]);
        MyMesh.materials.push_back(m);
    }
    MyLightPositions.clear();
    for (int i = 0; i < nl; ++i)
        MyLightPositions.push_back(Vec3Df(lights[3 * i], lights[3 * i + 1], lights[3 * i + 2]));
    focusDepth = 0;

    MyMesh.computeVertexNormals();
    fwrite(&nv, 4, 1, out);
    for (int i = 0; i < nv; ++i) fwrite(MyMesh.vertices[i].n.p, 4, 3, out);

    buildKDtree();
    dumpTree(out, treeRoot);

    int nr;
    rd(fr, &nr, 4);
    std::vector<float> ro(3 * nr), rdst(3 * nr);
    rd(fr, ro.data(), 12 * nr);
    rd(fr, rdst.data(), 12 * nr);
    fclose(fr);
    fwrite(&nr, 4, 1, out);
    for (int i = 0; i < nr; ++i) {
        Vec3Df o(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]), d(rdst[3 * i], rdst[3 * i + 1], rdst[3 * i + 2]);
        float t1 = 0, t2 = 0;
        HitTriangle h = checkHit(o, d, t1);
        Vec3Df col = performRayTracing(o, d, 0, 1.0f, t2);
        int   bh = h.bHit ? 1 : 0;
        int   v[3] = {-1, -1, -1};
        float ab[2] = {0, 0}, hp[3] = {0, 0, 0};
        if (h.bHit) {
            v[0] = (int)h.triangle.v[0], v[1] = (int)h.triangle.v[1], v[2] = (int)h.triangle.v[2];
            ab[0] = h.a, ab[1] = h.b;
            hp[0] = h.hitPoint[0], hp[1] = h.hitPoint[1], hp[2] = h.hitPoint[2];
        }
        fwrite(&bh, 4, 1, out);
        fwrite(v, 4, 3, out);
        fwrite(ab, 4, 2, out);
        fwrite(&t1, 4, 1, out);
        fwrite(hp, 4, 3, out);
        fwrite(col.p, 4, 3, out);
        fwrite(&t2, 4, 1, out);
    }
    fclose(out);
    return 0;
}
