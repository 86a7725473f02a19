// rt_host.hpp - C++ host side of the gfx950 ray tracer.
//
// It keeps the surface the reference's host code programs against
// (raytracing.h:12-33 in the reference: the globals MyMesh, MyLightPositions,
// treeRoot, focusDepth and the functions init(), performRayTracing(),
// checkHit(), plus RayTracer::doDaRayTracingShizz() of main.cpp:322) and
// forwards the per-pixel work to librt_hip.so.  Scene loading, the KD build,
// the camera set-up and the image files stay on the host, as in the reference.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "rt_hip.h"

namespace rtamd {

struct Vec3Df {
    float p[3];
    Vec3Df() : p{0, 0, 0} {}
    Vec3Df(float x, float y, float z) : p{x, y, z} {}
    float&       operator[](int i) { return p[i]; }
    const float& operator[](int i) const { return p[i]; }
};

struct Texture {
    std::string          path;
    int                  width = 0, height = 0;
    std::vector<uint8_t> rgba; // width*height*4, top row first
};

// Mesh of mesh.h:258-291, stored as flat arrays ready for RtSceneDesc.
struct Mesh {
    std::vector<float>       positions; // 3 per vertex
    std::vector<float>       normals;   // 3 per vertex, filled by computeVertexNormals
    std::vector<float>       texcoords; // 3 per entry
    std::vector<int32_t>     tri_vertex, tri_texcoord, tri_material;
    std::vector<RtMaterial>  materials;
    std::vector<std::string> material_names;
    std::vector<Texture>     textures;

    size_t vertexCount() const { return positions.size() / 3; }
    size_t triangleCount() const { return tri_material.size(); }
    // mesh.cpp:96 (the flag is accepted and ignored there too: mesh.cpp:307)
    bool loadMesh(const char* filename, bool randomizeTriangulation = true, bool loadTextures = true);
    // mesh.cpp:32
    void computeVertexNormals();
};

// The tree of KDtree.h:9-27 as index-linked nodes.
struct KdTree {
    std::vector<RtKdNode> nodes;
    std::vector<int32_t>  node_tris;
    int                   depth = 0;
};
// KDtree.cpp:15
void buildKDtree(const Mesh& mesh, KdTree& tree);

// PNG -> RGBA8 as corona::OpenImage(path, PF_R8G8B8A8) delivers it (mesh.h:117-135).
bool decodePng(const uint8_t* data, size_t size, Texture& out, std::string& err);
bool readPng(const std::string& path, Texture& out, std::string& err);

// camera.sav of main.cpp:270-307: 16 floats of the column-major modelview,
// the number of lights, one light per line.
bool readCameraSav(const char* path, double modelview[16], std::vector<Vec3Df>& lights);
bool writeCameraSav(const char* path, const double modelview[16], const std::vector<Vec3Df>& lights);
// The eight corner points of main.cpp:52-55 for gluPerspective(50, w/h, 0.01, 10)
// (main.cpp:209), viewport (0,0,w,h) and the given modelview.
void cameraCorners(const double modelview[16], int width, int height, float corners[24]);

// Image of imageWriter.h:81-251 (float planes + the file writers).
struct Image {
    int                width = 0, height = 0;
    std::vector<float> rgb, depth, blurred;
    // The same three planes as the writers quantise them, when the GPU has already done that
    // (RayTracer::doDaRayTracingShizz): rgb8 w*h*3, depth8 w*h grey values of depth.bmp,
    // blurred8 (w-12)*(h-12)*3.  A writer uses the bytes when they are set, else the floats.
    const uint8_t *rgb8 = nullptr, *depth8 = nullptr, *blurred8 = nullptr;
    // imageWriter.h:239: the byte buffer the three BMP writers share; what one writer leaves
    // in it shows through in the next file where that one skips bytes
    mutable std::vector<unsigned char> imageC;
    Image() {}
    Image(int w, int h) : width(w), height(h), rgb((size_t)3 * w * h), depth((size_t)w * h) {}
    bool writeImagePPM(const char* filename) const;
    bool writeImageBMP(const char* filename) const;
    // imageWriter.h:472: rewrites rows y..y+h-1, columns x..x+w-1 of an existing result.bmp
    bool writeImageBMP(const char* filename, int x, int y, int w, int h) const;
    bool writeDepthBMP(const char* filename) const;   // expects normalised depth
    bool writeBlurredBMP(const char* filename) const; // expects `blurred` filled
};

struct HitTriangle { // raytracing.h:48-54
    bool    bHit = false;
    float   a = 0, b = 0;
    Vec3Df  hitPoint;
    int32_t triangle = -1; // index into Mesh triangles
};

// ---- the raytracing.h surface ----
extern Mesh                MyMesh;
extern std::vector<Vec3Df> MyLightPositions;
extern KdTree              MyTree;
extern float               focusDepth;
extern RtParams            MySettings; // settings.h + config.h values

// raytracing.h:24.  The reference compiles OBJ_PATH in; here it is an argument.
// Returns false (and leaves lastError()) when the scene cannot be set up.
bool init(const char* obj_path, int device = 0, bool loadTextures = true);
void shutdown();
const std::string& lastError();
void               setLastError(const std::string& message);
RtScene*           gpuScene();
// raytracing.h:33
Vec3Df performRayTracing(const Vec3Df& origin, const Vec3Df& dest, int k, float prev_Ni, float& minT);
// raytracing.h:76-77
HitTriangle checkHit(const Vec3Df& origin, const Vec3Df& dest);
HitTriangle checkHit(const Vec3Df& origin, const Vec3Df& dest, float& minT);

// RayTracer of main.cpp:41-80: corner rays from the camera, one GPU frame,
// then the four result files.
class RayTracer {
public:
    RayTracer(const double modelview[16], int width, int height);
    // main.cpp:322: renders and writes result.ppm, result.bmp, depth.bmp and
    // blurred.bmp into out_dir; returns the seconds the frame took, < 0 on error.
    // The frame, both depth normalisations and the blur stay on the GPU; what comes back is
    // the 8-bit planes of the four files (result.rgb8 / depth8 / blurred8).  floatPlanes = true
    // brings the float planes back instead (result.rgb / depth / blurred as the reference's
    // Image holds them) and post-processes those.  rowUpdates = true also writes the blank
    // result.bmp first and rewrites it row by row (main.cpp:325, 407) before the final files.
    double doDaRayTracingShizz(const char* out_dir = ".", bool floatPlanes = false, bool rowUpdates = false);
    double files_seconds = 0; // time the last call spent writing files
    // Several GPUs, one process each: every process calls shareFrame with its rank and a
    // communicator from rt_rccl_comm_create before doDaRayTracingShizz; each then renders its
    // tiles, rank 0 gathers them, post-processes and writes the files, the others write nothing.
    void shareFrame(void* comm, int rank, int n_ranks) { comm_ = comm, rank_ = rank, n_ranks_ = n_ranks; }

private:
    void* comm_ = nullptr;
    int   rank_ = 0, n_ranks_ = 1;
    bool  dealt_ = false; // rt_calibrate_deal has run for this tracer

public:
    // Key 'f' (raytracing.cpp:504-507): focusDepth = depth of what the ray through pixel (x, y)
    // hits, traced as performRayTracing(origin, dest, 0, 0.0f, focusDepth); the ray is the one
    // raytrace() builds for that pixel (main.cpp:514-529).  Returns false when the GPU call fails.
    bool setFocusAtPixel(double x, double y);
    Image   result;
    RtStats stats{};
    float   corners[24];
};

} // namespace rtamd
