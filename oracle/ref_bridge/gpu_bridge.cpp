// gpu_bridge.cpp - the file a maintainer of the reference would add to bind its host code to
// librt_hip.so (include/rt_hip.h).  It is written against the reference's own headers and is
// compiled, syntax only, by `make -C oracle ref` where the reference checkout is present, so
// that this binding is known to fit raytracing.h / mesh.h / KDtree.h as they are.  Nothing else in this repository uses it.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rt_hip.h" // before settings.h: that header defines backgroundColor as a macro (settings.h:18),
                    // which is also the name of a field of RtParams
#include "config.h"
#include "raytracing.h"
#include "settings.h"

extern float focusDepth; // raytracing.cpp:17

static RtScene* gScene;

// treeRoot (KDtree.h:9-27) -> index-linked RtKdNode array, pre-order
static int flatten(AccelTreeNode* n, std::vector<RtKdNode>& nodes, std::vector<int32_t>& tris)
{
    const int me = (int)nodes.size();
    RtKdNode  k;
    const float b[6] = {n->xStart, n->xEnd, n->yStart, n->yEnd, n->zStart, n->zEnd};
    memcpy(k.bounds, b, sizeof b);
    k.left = k.right = -1;
    k.tri_offset = (int32_t)tris.size();
    k.tri_count = (int32_t)n->triangles.size();
    nodes.push_back(k);
    tris.insert(tris.end(), n->triangles.begin(), n->triangles.end());
    if (n->leftChild) {
        const int l = flatten(n->leftChild, nodes, tris), r = flatten(n->rightChild, nodes, tris);
        nodes[me].left = l;
        nodes[me].right = r;
    }
    return me;
}

// call at the end of init() (raytracing.cpp:23-53): MyMesh is loaded, normals computed, tree built
void gpuInit()
{
    std::vector<float>   pos, nrm, tc;
    std::vector<int32_t> tv, tt, tm;
    for (size_t i = 0; i < MyMesh.vertices.size(); ++i) {
        const Vertex& v = MyMesh.vertices[i];
        pos.insert(pos.end(), v.p.p, v.p.p + 3);
        nrm.insert(nrm.end(), v.n.p, v.n.p + 3);
    }
    for (size_t i = 0; i < MyMesh.triangles.size(); ++i)
        for (int j = 0; j < 3; ++j) {
            tv.push_back((int32_t)MyMesh.triangles[i].v[j]);
            tt.push_back((int32_t)MyMesh.triangles[i].t[j]);
        }
    tm.assign(MyMesh.triangleMaterials.begin(), MyMesh.triangleMaterials.end());
    for (size_t i = 0; i < MyMesh.texcoords.size(); ++i)
        tc.insert(tc.end(), MyMesh.texcoords[i].p, MyMesh.texcoords[i].p + 3);
    std::vector<RtMaterial> mats;
    std::vector<RtTexture>  tex;
    for (size_t i = 0; i < MyMesh.materials.size(); ++i) { // mesh.h:141-147
        Material&  m = MyMesh.materials[i];
        RtMaterial r;
        memset(&r, 0, sizeof r);
        for (int c = 0; c < 3; ++c) r.Kd[c] = m.Kd()[c], r.Ks[c] = m.Ks()[c];
        r.Ns = m.Ns();
        r.Ni = m.Ni();
        r.illum = m.illum();
        r.has_Ni = m.has_Ni();
        r.texture = -1; // a textured material needs accessors for Material's private texturePixels /
                        // textureWidth / textureHeight (mesh.h:164-167): push {pixels, width, height}
                        // into tex and store its index here
        mats.push_back(r);
    }
    std::vector<RtKdNode> nodes;
    std::vector<int32_t>  ntris;
    flatten(&treeRoot, nodes, ntris);
    RtSceneDesc d;
    memset(&d, 0, sizeof d);
    d.n_vertices = (int32_t)MyMesh.vertices.size(), d.vertex_pos = pos.data(), d.vertex_normal = nrm.data();
    d.n_triangles = (int32_t)MyMesh.triangles.size(), d.tri_vertex = tv.data(), d.tri_texcoord = tt.data();
    d.tri_material = tm.data();
    d.n_texcoords = (int32_t)MyMesh.texcoords.size(), d.texcoords = tc.data();
    d.n_materials = (int32_t)mats.size(), d.materials = mats.data();
    d.n_textures = (int32_t)tex.size(), d.textures = tex.data();
    d.n_nodes = (int32_t)nodes.size(), d.nodes = nodes.data();
    d.n_node_tris = (int32_t)ntris.size(), d.node_tris = ntris.data();
    if (rt_scene_create(&d, /*device*/ 0, &gScene) != RT_OK) {
        fprintf(stderr, "%s\n", rt_last_error());
        exit(1);
    }
}

// the body of RayTracer::doDaRayTracingShizz (main.cpp:322-381): `corners` are the eight points
// of main.cpp:52-55 in the order origin00 dest00 origin01 dest01 origin10 dest10 origin11 dest11;
// image / depth are Image::_image / Image::_depth (imageWriter.h:240-241)
bool gpuRenderFrame(const float corners[24], std::vector<float>& image, std::vector<float>& depth)
{
    RtParams p;
    rt_default_params(&p, WindowSize_X, WindowSize_Y);
    p.msaa = MSAA;             // settings.h:15
    p.focusDepth = focusDepth; // raytracing.cpp:17
    std::vector<float> lights;
    for (size_t i = 0; i < MyLightPositions.size(); ++i)
        lights.insert(lights.end(), MyLightPositions[i].p, MyLightPositions[i].p + 3);
    RtStats st;
    image.resize((size_t)3 * WindowSize_X * WindowSize_Y);
    depth.resize((size_t)WindowSize_X * WindowSize_Y);
    if (rt_render_frame(gScene, corners, lights.data(), (int32_t)MyLightPositions.size(), &p, &image[0], &depth[0],
                        NULL, &st) != RT_OK) {
        fprintf(stderr, "%s\n", rt_last_error());
        return false;
    }
    return true; // result.writeImagePPM / writeImageBMP / writeDepthBMP / writeBlurredBMP run unchanged
}

// performRayTracing(origin, dest, k, prev_Ni, minT) (raytracing.h:33) for single picks such as key 'f'
Vec3Df gpuPerformRayTracing(const Vec3Df& origin, const Vec3Df& dest, int k, float prev_Ni, float& minT)
{
    RtParams p;
    rt_default_params(&p, WindowSize_X, WindowSize_Y);
    std::vector<float> lights;
    for (size_t i = 0; i < MyLightPositions.size(); ++i)
        lights.insert(lights.end(), MyLightPositions[i].p, MyLightPositions[i].p + 3);
    Vec3Df rgb;
    rt_trace_rays(gScene, lights.data(), (int32_t)MyLightPositions.size(), &p, k, prev_Ni, 1, origin.p, dest.p, rgb.p,
                  &minT, NULL, NULL);
    return rgb;
}
