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
 * Mesh::loadMtl (mesh.cpp:354-480) runs unmodified too (mode --mtl): it never
 * touches the image library.  Material::getTexture / getTexture2
 * (mesh.cpp:485-553) run unmodified on pixels this harness hands over: the
 * members Corona would have filled (texturePixels, textureWidth, textureHeight,
 * containsTexture, mesh.h:164-168) are private, so this one translation unit
 * sees the reference's headers with `private` spelled `public` (the standard
 * headers are included before that; mesh.cpp itself is compiled as it lies).
 * What does not run: Mesh::loadMesh (needs Corona), so the mesh is handed over
 * through the members of Mesh/Material; main.cpp (GLUT) is not compiled.
 *
 * Built as _ref/libref.so; _ref/refprobe (ref_driver.c) loads it with lazy
 * binding, because Mesh::loadMesh's call into Corona stays an undefined symbol.
 *
 * usage: refprobe <scene.bin> <rays.bin> <out.bin>
 *        refprobe --mtl <file.mtl> <out.bin>
 */
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "config.h"
#define private public
#include "raytracing.h"
#undef private
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

static void putStr(FILE *out, const std::string &s)
{
    int n = (int)s.size();
    fwrite(&n, 4, 1, out);
    if (n) fwrite(s.data(), 1, n, out);
}

// --mtl: Mesh::loadMtl on one file; per material, in file order: name, Kd, Ks, Ns, Ni, illum, Tr,
// the has_* flags of its own block, the texture name.  Fields no line has set yet hold whatever
// the reference's uninitialised `Material mat` (mesh.cpp:369) holds: the reader of this dump
// compares only fields that some line at or above the material's block has set.
static int mtlMain(const char *path, const char *outPath)
{
    FILE *out = fopen(outPath, "wb");
    if (!out) return 1;
    Mesh                                mesh;
    std::map<std::string, unsigned int> index;
    const bool                          ok = mesh.loadMtl(path, index);
    int                                 n = ok ? (int)mesh.materials.size() : -1;
    fwrite(&n, 4, 1, out);
    for (int i = 0; i < n; ++i) {
        Material &m = mesh.materials[i];
        putStr(out, m.name());
        float v[9] = {m.Kd()[0], m.Kd()[1], m.Kd()[2], m.Ks()[0], m.Ks()[1], m.Ks()[2], m.Ns(), m.Ni(), m.Tr()};
        fwrite(v, 4, 9, out);
        int w[2] = {m.illum(), (m.has_Kd() ? 1 : 0) | (m.has_Ka() ? 2 : 0) | (m.has_Ks() ? 4 : 0) | (m.has_Ns() ? 8 : 0) |
                                   (m.has_Ni() ? 16 : 0) | (m.has_illum() ? 32 : 0) | (m.has_Tr() ? 64 : 0)};
        fwrite(w, 4, 2, out);
        putStr(out, m.textureName());
        int idx = -1;
        for (auto &kv : index)
            if ((int)kv.second == i) idx = 1;
        fwrite(&idx, 4, 1, out);
    }
    fclose(out);
    return 0;
}

extern "C" int ref_main(int argc, char **argv)
{
    if (argc == 4 && !strcmp(argv[1], "--mtl")) return mtlMain(argv[2], argv[3]);
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
    // optional: textures as the image library would have delivered them (RGBA8, top row first), each
    // followed by the two zero rows that make the over-read of mesh.cpp:499 defined; the material rows
    // name their texture in field 10 (-1: none)
    int ntex = 0;
    if (fread(&ntex, 4, 1, fs) != 1) ntex = 0;
    static std::vector<std::vector<unsigned char>> texPixels;
    std::vector<int>                               texW(ntex), texH(ntex);
    texPixels.resize(ntex);
    for (int i = 0; i < ntex; ++i) {
        rd(fs, &texW[i], 4);
        rd(fs, &texH[i], 4);
        texPixels[i].resize((size_t)texW[i] * (texH[i] + 2) * 4);
        rd(fs, texPixels[i].data(), texPixels[i].size());
    }
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
        const int tex = (int)r[10];
        if (ntex > 0 && tex >= 0 && tex < ntex) { // what Material::loadTexture (mesh.h:117-135) leaves behind
            m.containsTexture = true;
            m.texturePixels = texPixels[tex].data();
            m.textureWidth = texW[tex];
            m.textureHeight = texH[tex];
        }
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
