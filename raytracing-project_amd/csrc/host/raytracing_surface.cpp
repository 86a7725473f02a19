// raytracing_surface.cpp - the reference's raytracing.h surface on top of the
// GPU library, and the 'r' render path.
//
//   init()                 raytracing.cpp:23-53   load, normals, default light, KD build,
//                                                 then hand the scene to the GPU
//   performRayTracing()    raytracing.cpp:61      one ray through rt_trace_rays
//   checkHit()             raytracing.cpp:276-351 one ray through rt_check_hits
//   RayTracer              main.cpp:41-80, 322-381
//
// The extern "C" rth_* functions at the end expose the host side to the Python
// tests and to bench.py (array getters, file writers, the render path).
#include <chrono>
#include <cstring>

#include "rt_host.hpp"

namespace rtamd {

Mesh                MyMesh;
std::vector<Vec3Df> MyLightPositions;
KdTree              MyTree;
float               focusDepth = 0.0f;
RtParams            MySettings;

namespace {
RtScene*    g_scene = nullptr;
int         g_device = 0;
std::string g_error;

bool fail(const std::string& what)
{
    g_error = what + ": " + rt_last_error();
    return false;
}
} // namespace

const std::string& lastError() { return g_error; }
void               setLastError(const std::string& message) { g_error = message; }
RtScene*           gpuScene() { return g_scene; }

// Fills an RtSceneDesc that points into mesh/tree (valid while they live).
static void describe(const Mesh& mesh, const KdTree& tree, std::vector<RtTexture>& tex, RtSceneDesc& d)
{
    memset(&d, 0, sizeof d);
    d.n_vertices = (int32_t)mesh.vertexCount();
    d.vertex_pos = mesh.positions.data();
    d.vertex_normal = mesh.normals.data();
    d.n_triangles = (int32_t)mesh.triangleCount();
    d.tri_vertex = mesh.tri_vertex.data();
    d.tri_texcoord = mesh.tri_texcoord.data();
    d.tri_material = mesh.tri_material.data();
    d.n_texcoords = (int32_t)(mesh.texcoords.size() / 3);
    d.texcoords = mesh.texcoords.data();
    d.n_materials = (int32_t)mesh.materials.size();
    d.materials = mesh.materials.data();
    tex.clear();
    for (const Texture& t : mesh.textures) tex.push_back(RtTexture{t.rgba.data(), t.width, t.height});
    d.n_textures = (int32_t)tex.size();
    d.textures = tex.data();
    d.n_nodes = (int32_t)tree.nodes.size();
    d.nodes = tree.nodes.data();
    d.n_node_tris = (int32_t)tree.node_tris.size();
    d.node_tris = tree.node_tris.data();
}

void shutdown()
{
    if (g_scene) rt_scene_destroy(g_scene);
    g_scene = nullptr;
}

bool init(const char* obj_path, int device, bool loadTextures)
{
    shutdown();
    rt_default_params(&MySettings, 800, 800); // config.h.dist:17-18
    if (!MyMesh.loadMesh(obj_path, true, loadTextures)) {
        g_error = std::string("cannot open ") + obj_path;
        return false;
    }
    MyMesh.computeVertexNormals();
    MyLightPositions.clear();
    MyLightPositions.push_back(Vec3Df(0, 10, 0)); // raytracing.cpp:38
    buildKDtree(MyMesh, MyTree);
    focusDepth = 0.0f;
    std::vector<RtTexture> tex;
    RtSceneDesc            d;
    describe(MyMesh, MyTree, tex, d);
    g_device = device;
    if (rt_scene_create(&d, device, &g_scene) != RT_OK) return fail("rt_scene_create");
    return true;
}

static std::vector<float> lightArray()
{
    std::vector<float> l;
    for (const Vec3Df& v : MyLightPositions) l.insert(l.end(), {v[0], v[1], v[2]});
    return l;
}

Vec3Df performRayTracing(const Vec3Df& origin, const Vec3Df& dest, int k, float prev_Ni, float& minT)
{
    Vec3Df rgb;
    if (k > 5) return rgb; // endOfReflection, raytracing.cpp:63
    std::vector<float> l = lightArray();
    if (!g_scene || rt_trace_rays(g_scene, l.data(), (int32_t)MyLightPositions.size(), &MySettings, k,
                                  prev_Ni, 1, origin.p, dest.p, rgb.p, &minT, nullptr, nullptr) != RT_OK)
        fail("performRayTracing");
    return rgb;
}

HitTriangle checkHit(const Vec3Df& origin, const Vec3Df& dest, float& minT)
{
    HitTriangle h;
    if (!g_scene || rt_check_hits(g_scene, 1, origin.p, dest.p, &h.triangle, &h.a, &h.b, &minT,
                                  h.hitPoint.p) != RT_OK) {
        fail("checkHit");
        return h;
    }
    h.bHit = h.triangle >= 0;
    return h;
}

HitTriangle checkHit(const Vec3Df& origin, const Vec3Df& dest)
{
    float d;
    return checkHit(origin, dest, d);
}

RayTracer::RayTracer(const double modelview[16], int width, int height) : result(width, height)
{
    cameraCorners(modelview, width, height, corners);
}

bool RayTracer::setFocusAtPixel(double x, double y)
{
    const int    W = result.width, H = result.height;
    const double xscale = 1.0 - x / (double)(W - 1), yscale = 1.0 - y / (double)(H - 1);
    const float  xs = (float)xscale, xs1 = (float)(1 - xscale), ys = (float)yscale, ys1 = (float)(1 - yscale);
    Vec3Df       o, d;
    for (int a = 0; a < 3; ++a) { // corners: origin00 dest00 origin01 dest01 origin10 dest10 origin11 dest11
        o.p[a] = (corners[a] * xs + corners[12 + a] * xs1) * ys + (corners[6 + a] * xs + corners[18 + a] * xs1) * ys1;
        d.p[a] = (corners[3 + a] * xs + corners[15 + a] * xs1) * ys + (corners[9 + a] * xs + corners[21 + a] * xs1) * ys1;
    }
    const std::string before = g_error;
    g_error.clear();
    float depth = focusDepth;
    performRayTracing(o, d, 0, 0.0f, depth);
    if (!g_error.empty()) return false;
    g_error = before;
    focusDepth = depth;
    fprintf(stderr, "Focus depth set to %f\n", focusDepth);
    return true;
}

double RayTracer::doDaRayTracingShizz(const char* out_dir, bool floatPlanes, bool rowUpdates)
{
    if (!g_scene) {
        g_error = "doDaRayTracingShizz: init() has not succeeded";
        return -1.0;
    }
    const int W = result.width, H = result.height;
    fprintf(stderr, "\nRaytracing a complete %i x %i image \n", W, H);
    const std::string dir = std::string(out_dir && *out_dir ? out_dir : ".") + "/";
    if (rowUpdates) Image(W, H).writeImageBMP((dir + "result.bmp").c_str()); // main.cpp:325
    RtParams p = MySettings;
    p.width = W, p.height = H;
    p.focusDepth = focusDepth;
    p.tile_w = p.tile_h = 0, p.rank = 0, p.world = 1;
    std::vector<float> l = lightArray();
    result.rgb8 = result.depth8 = result.blurred8 = nullptr;
    if (n_ranks_ > 1 && floatPlanes) {
        g_error = "doDaRayTracingShizz: a shared frame comes back as 8-bit planes only";
        return -1.0;
    }
    if (n_ranks_ > 1 && !dealt_) {
        // The reference's threads balance a frame while it renders (a thread that is through steals rows,
        // main.cpp:410-429); here the deal is settled once, before the frame that counts: rank 0, which
        // also post-processes, gets the share that makes all ranks finish together.
        int32_t shares = 0;
        if (rt_calibrate_deal(g_scene, comm_, rank_, n_ranks_, corners, l.data(), (int32_t)MyLightPositions.size(), &p,
                              &shares) != RT_OK) {
            fail("rt_calibrate_deal");
            return -1.0;
        }
        dealt_ = true;
        if (rank_ == 0) fprintf(stderr, "Deal: rank 0 renders %d of %d shares per GPU\n", shares, RT_SHARES_PER_RANK);
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (!floatPlanes) {
        if (rt_render_result_planes_shared(g_scene, comm_, rank_, n_ranks_, corners, l.data(),
                                           (int32_t)MyLightPositions.size(), &p, &result.rgb8, &result.depth8,
                                           &result.blurred8, &stats) != RT_OK) {
            fail("rt_render_result_planes_shared");
            return -1.0;
        }
        if (rank_ != 0) return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } else {
        if (rt_render_frame(g_scene, corners, l.data(), (int32_t)MyLightPositions.size(), &p, result.rgb.data(),
                            result.depth.data(), nullptr, &stats) != RT_OK) {
            fail("rt_render_frame");
            return -1.0;
        }
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "\nRaytracing comlete in %g seconds, storing result...\n", secs);
    const auto t1 = std::chrono::steady_clock::now();
    if (rowUpdates) // main.cpp:407, one update per finished row
        for (int y = 0; y < H; ++y) result.writeImageBMP((dir + "result.bmp").c_str(), 0, y, W, 1);
    bool ok = result.writeImagePPM((dir + "result.ppm").c_str());
    ok = result.writeImageBMP((dir + "result.bmp").c_str()) && ok;
    if (floatPlanes) {
        // writeDepthBMP normalises the depth plane, writeBlurredBMP normalises it a
        // second time before blurring (imageWriter.h:352, 416)
        const int64_t npx = (int64_t)W * H;
        if (rt_normalize_depth(g_device, npx, focusDepth, result.depth.data()) != RT_OK) {
            fail("rt_normalize_depth");
            return -1.0;
        }
    }
    ok = result.writeDepthBMP((dir + "depth.bmp").c_str()) && ok;
    if (W > 12 && H > 12) {
        if (floatPlanes) {
            std::vector<float> twice = result.depth;
            if (rt_normalize_depth(g_device, (int64_t)W * H, focusDepth, twice.data()) != RT_OK) {
                fail("rt_normalize_depth");
                return -1.0;
            }
            result.blurred.assign((size_t)(W - 12) * (H - 12) * 3, 0.0f);
            if (rt_dof_blur(g_device, W, H, p.depthPower, result.rgb.data(), twice.data(), result.blurred.data()) !=
                RT_OK) {
                fail("rt_dof_blur");
                return -1.0;
            }
        }
        ok = result.writeBlurredBMP((dir + "blurred.bmp").c_str()) && ok;
    }
    files_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    if (!ok) {
        g_error = "could not write the result files into " + dir;
        return -1.0;
    }
    fprintf(stderr, "Stored result in result.ppm, result.bmp, depth.bmp and blurred.bmp\n");
    return secs;
}

} // namespace rtamd

// ---------------------------------------------------------------------------
// C hooks for the Python tests / bench.py
// ---------------------------------------------------------------------------
using namespace rtamd;

struct RthScene {
    Mesh                   mesh;
    KdTree                 tree;
    std::vector<RtTexture> tex;
    RtSceneDesc            desc;
};

extern "C" {

RthScene* rth_scene_load(const char* obj_path, int load_textures)
{
    RthScene* s = new RthScene();
    if (!s->mesh.loadMesh(obj_path, true, load_textures != 0)) {
        delete s;
        return nullptr;
    }
    s->mesh.computeVertexNormals();
    buildKDtree(s->mesh, s->tree);
    describe(s->mesh, s->tree, s->tex, s->desc);
    return s;
}
void               rth_scene_free(RthScene* s) { delete s; }
const RtSceneDesc* rth_scene_desc(const RthScene* s) { return &s->desc; }
int                rth_tree_depth(const RthScene* s) { return s->tree.depth; }
const char*        rth_material_name(const RthScene* s, int i) { return s->mesh.material_names[i].c_str(); }

int rth_read_camera_sav(const char* path, double modelview[16], float* lights, int max_lights)
{
    std::vector<Vec3Df> l;
    if (!readCameraSav(path, modelview, l)) return -1;
    for (int i = 0; i < (int)l.size() && i < max_lights; ++i) memcpy(lights + 3 * i, l[i].p, 12);
    return (int)l.size();
}
void rth_camera_corners(const double modelview[16], int w, int h, float corners[24])
{
    cameraCorners(modelview, w, h, corners);
}

static Image asImage(const float* rgb, const float* depth, const float* blurred, int w, int h)
{
    Image im;
    im.width = w, im.height = h;
    if (rgb) im.rgb.assign(rgb, rgb + (size_t)3 * w * h);
    if (depth) im.depth.assign(depth, depth + (size_t)w * h);
    if (blurred) im.blurred.assign(blurred, blurred + (size_t)3 * (w - 12) * (h - 12));
    return im;
}
int rth_write_ppm(const char* path, const float* rgb, int w, int h)
{
    return asImage(rgb, nullptr, nullptr, w, h).writeImagePPM(path);
}
int rth_write_bmp(const char* path, const float* rgb, int w, int h)
{
    return asImage(rgb, nullptr, nullptr, w, h).writeImageBMP(path);
}
int rth_write_depth_bmp(const char* path, const float* depth_n, int w, int h)
{
    return asImage(nullptr, depth_n, nullptr, w, h).writeDepthBMP(path);
}
int rth_write_blurred_bmp(const char* path, const float* blurred, int w, int h)
{
    return asImage(nullptr, nullptr, blurred, w, h).writeBlurredBMP(path);
}

// The file sequence of the 'r' path on ONE Image, as main.cpp:325, 404-407 and 371-377 run
// it: blank progress.bmp, one per-row update per row, result.ppm, result.bmp, depth.bmp
// (depth_n = depth after the first normalisation), blurred.bmp (blurred = the blur's output).
int rth_write_result_files(const char* out_dir, const float* rgb, const float* depth_n, const float* blurred,
                           int w, int h)
{
    Image             blank(w, h);
    const std::string dir = std::string(out_dir) + "/";
    const std::string progress = dir + "progress.bmp";
    bool              ok = blank.writeImageBMP(progress.c_str());
    Image             im = asImage(rgb, depth_n, (w > 12 && h > 12) ? blurred : nullptr, w, h);
    im.imageC = blank.imageC;
    for (int y = 0; y < h; ++y) ok = im.writeImageBMP(progress.c_str(), 0, y, w, 1) && ok;
    ok = im.writeImagePPM((dir + "result.ppm").c_str()) && ok;
    ok = im.writeImageBMP((dir + "result.bmp").c_str()) && ok;
    ok = im.writeDepthBMP((dir + "depth.bmp").c_str()) && ok;
    if (w > 12 && h > 12) ok = im.writeBlurredBMP((dir + "blurred.bmp").c_str()) && ok;
    return ok;
}

// init() + camera.sav + the 'r' path; returns seconds or < 0
double rth_render_to_files(const char* obj_path, const char* camera_sav, int w, int h, int msaa,
                           const char* out_dir, int device, RtStats* stats)
{
    if (!init(obj_path, device, true)) return -1.0;
    double modelview[16];
    if (!readCameraSav(camera_sav, modelview, MyLightPositions)) return -2.0;
    MySettings.msaa = msaa;
    RayTracer    r(modelview, w, h);
    const double secs = r.doDaRayTracingShizz(out_dir);
    if (stats) *stats = r.stats;
    shutdown();
    return secs;
}
// the same through the float planes (result.rgb / depth / blurred filled), optionally with the
// per-row updates of result.bmp
double rth_render_to_files_float(const char* obj_path, const char* camera_sav, int w, int h, int msaa,
                                 const char* out_dir, int device, int row_updates)
{
    if (!init(obj_path, device, true)) return -1.0;
    double modelview[16];
    if (!readCameraSav(camera_sav, modelview, MyLightPositions)) return -2.0;
    MySettings.msaa = msaa;
    RayTracer    r(modelview, w, h);
    const double secs = r.doDaRayTracingShizz(out_dir, true, row_updates != 0);
    shutdown();
    return secs;
}
// `runs` times the 'r' path on a resident scene; out[0] = seconds of the last run's frame call
// (GPU frame + post-process + planes to the host), out[1] = seconds it spent writing the files
int rth_time_r_path(const char* obj_path, const char* camera_sav, int w, int h, int msaa, const char* out_dir,
                    int device, int runs, double out[2])
{
    if (!init(obj_path, device, true)) return -1;
    double modelview[16];
    if (!readCameraSav(camera_sav, modelview, MyLightPositions)) return -2;
    MySettings.msaa = msaa;
    RayTracer r(modelview, w, h);
    for (int i = 0; i < runs; ++i) {
        out[0] = r.doDaRayTracingShizz(out_dir);
        out[1] = r.files_seconds;
        if (out[0] < 0) return -3;
    }
    shutdown();
    return 0;
}
// the same after key 'f' at pixel (fx, fy); *focus receives the focus depth used
double rth_render_to_files_focus(const char* obj_path, const char* camera_sav, int w, int h, int msaa,
                                 const char* out_dir, int device, double fx, double fy, float* focus,
                                 RtStats* stats)
{
    if (!init(obj_path, device, true)) return -1.0;
    double modelview[16];
    if (!readCameraSav(camera_sav, modelview, MyLightPositions)) return -2.0;
    MySettings.msaa = msaa;
    RayTracer r(modelview, w, h);
    focusDepth = 0.0f;
    if (!r.setFocusAtPixel(fx, fy)) return -3.0;
    if (focus) *focus = focusDepth;
    const double secs = r.doDaRayTracingShizz(out_dir);
    if (stats) *stats = r.stats;
    focusDepth = 0.0f;
    shutdown();
    return secs;
}
const char* rth_last_error(void) { return lastError().c_str(); }

} // extern "C"
