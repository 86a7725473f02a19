/*
 * rt_oracle.h - CPU restatement of the reference ray tracer's per-pixel render
 * loop.  TEST INFRASTRUCTURE ONLY: this library is the parity checker and the
 * reported CPU baseline.  Nothing in the product path (the HIP library, its
 * host-side C++ mirror, the CLI) may include, link or call it.
 *
 * Every function cites the reference file:line it restates
 * (paths are relative to the reference checkout).
 *
 * Parity status: the KD build, the leaf walk, the ray/triangle test, the
 * shading recursion (raytracing.cpp, KDtree.cpp, mesh.cpp's normals), the
 * material loader (Mesh::loadMtl, mesh.cpp:354-480, with the fields that leak
 * from one newmtl block into the next), the texture lookup
 * (Material::getTexture / getTexture2, mesh.cpp:485-553, on the .rtex pixels
 * + two zero rows) and the image code (imageWriter.h: RGBValue clamp, depth
 * normalisation, the 13x13 depth-of-field kernel, the blur, the PPM/BMP
 * writers and the per-row update) are pinned against the reference's own
 * sources compiled in the build container (oracle/_ref, see oracle/Makefile;
 * vectors in tests/golden/).  The OBJ geometry parser (Mesh::loadMesh,
 * mesh.cpp:96-350: it calls into the Corona image library) and the frame
 * driver (corner rays, per-pixel ray, MSAA composition; main.cpp needs
 * GLU/GLUT) are restated from reading the source only - PARITY UNPINNED for
 * those two, because the reference files that hold them need libraries this
 * image lacks.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct or_scene or_scene;

/* settings.h:7-19 as run-time fields (defaults = the reference's constants) */
typedef struct or_settings {
    float backgroundlighting; /* 0.34 */
    float diffusePower;       /* 0.5  */
    float ShadowFactor;       /* 0.5  */
    int   MSAA;               /* 1, 4 or 16 */
    float depthPower;         /* 2.5  */
    float backgroundColor[3]; /* 0.2 0.5 1.0 */
    float focusDepth;         /* raytracing.cpp:17, 0 unless key 'f' */
    int   shadowSamples;      /* 1: hard shadows; n > 1: n^3 jittered shadow rays per light (settings.h:13) */
    float shadowRadius;       /* 0.9 (settings.h:12) */
} or_settings;

/* work counters (one closest-hit or shadow query = one "ray") */
typedef struct or_counters {
    uint64_t rays_closest;   /* checkHit calls from performRayTracing line 69 */
    uint64_t rays_shadow;    /* checkHit calls from line 142 */
    uint64_t tri_tests;      /* T: ray/triangle tests executed by the leaf walk */
    uint64_t leaf_steps;     /* S: findNextNode calls */
    uint64_t shaded_hits;    /* H */
    uint64_t hit_lights;     /* Hl: (shaded hit, light) pairs */
    uint64_t child_rays;     /* C: reflection / refraction rays spawned */
    uint64_t camera_rays;    /* raytrace() calls of the frame driver */
    uint64_t walk_guard;     /* leaf walks stopped by the iteration guard */
    /* Shadow queries whose answer cannot change the colour: shadowMax is 1 before the
     * query whatever the earlier queries of this hit answered (raytracing.cpp:127-143).
     * Included in rays_shadow / tri_tests / leaf_steps above; counted again here so the
     * work that reaches the image can be stated separately. */
    uint64_t moot_shadow_rays, moot_tri_tests, moot_leaf_steps;
} or_counters;

typedef struct or_hit {
    int   bHit;
    float a, b;
    float hitPoint[3];
    int   tri;        /* triangle index, -1 when no hit */
    float minT;
} or_hit;

void or_default_settings(or_settings *s);

/* ---- scene: mesh.cpp:96-480 (loader), mesh.cpp:32-52 (normals) ---- */
/* load_textures: 0 = never mark a material textured (matches a build without
 * an image library), 1 = read <dir>/<name>.rtex files */
or_scene *or_scene_load(const char *obj_path, int load_textures);
/* build a scene from raw arrays (no loader involved); materials are given as
 * rows of 16 floats: Kd[3] Ks[3] Ns Ni illum has_Ni texId(-1 none) pad[5] */
or_scene *or_scene_from_arrays(int nv, const float *vpos, int nt, const int32_t *tri_v,
                               const int32_t *tri_t, const int32_t *tri_mat, int ntc,
                               const float *texcoords, int nm, const float *materials);
void or_scene_free(or_scene *s);
void or_compute_vertex_normals(or_scene *s);
void or_set_lights(or_scene *s, int n, const float *xyz);
or_settings *or_scene_settings(or_scene *s);

int or_num_vertices(const or_scene *s);
int or_num_triangles(const or_scene *s);
int or_num_texcoords(const or_scene *s);
int or_num_materials(const or_scene *s);
int or_num_textures(const or_scene *s);
const float *or_vertex_pos(const or_scene *s);
const float *or_vertex_nrm(const or_scene *s);
void or_get_triangles(const or_scene *s, int32_t *v, int32_t *t, int32_t *mat);
const float *or_texcoords(const or_scene *s);
/* 16 floats per material, layout as in or_scene_from_arrays */
void or_get_materials(const or_scene *s, float *out);
const char *or_material_name(const or_scene *s, int i);
/* texture i: padded RGBA8 (two extra zero rows), width, height */
const unsigned char *or_texture(const or_scene *s, int i, int *w, int *h);

/* ---- KD tree: KDtree.cpp:15-230 ---- */
void or_build_kdtree(or_scene *s);
int or_num_nodes(const or_scene *s);
int or_tree_depth(const or_scene *s);
int or_num_node_tris(const or_scene *s);
/* per node: bounds[6] = xS xE yS yE zS zE; left/right/parent (-1 none); depth;
 * tri_off/tri_cnt into the node-triangle index list */
void or_get_nodes(const or_scene *s, float *bounds, int32_t *left, int32_t *right,
                  int32_t *parent, int32_t *depth, int32_t *tri_off, int32_t *tri_cnt);
const int32_t *or_node_tris(const or_scene *s);

/* ---- tracing: raytracing.cpp:61-386, KDtree.cpp:235-605 ---- */
void or_check_hit(const or_scene *s, const float o[3], const float d[3], or_hit *out,
                  or_counters *c);
void or_perform_ray_tracing(const or_scene *s, const float o[3], const float d[3], int k,
                            float prev_Ni, float *minT, float rgb[3], or_counters *c);
/* batched form; rgb[n*3], minT[n], tri[n] (tri of the first hit, -1 on miss) */
void or_trace_rays(const or_scene *s, int n, const float *origins, const float *dests,
                   float *rgb, float *minT, int32_t *tri, int threads, or_counters *c);
/* batched closest hit only */
void or_check_hits(const or_scene *s, int n, const float *origins, const float *dests,
                   int32_t *tri, float *a, float *b, float *t, float *hitPoint, int threads,
                   or_counters *c);
/* rays whose leaf walk was stopped by the step guard since the last reset (origin, destination; at most
 * 64 are kept); returns how many there were */
int  or_guard_rays(float *out6, int max);
void or_guard_reset(void);
/* brute force over all triangles with the same triangle test (diagnostic) */
void or_brute_hits(const or_scene *s, int n, const float *origins, const float *dests,
                   int32_t *tri, float *t);

/* the counter-based stand-in for rand() in the soft-shadow loop (see rt_oracle.c) */
uint32_t or_soft_rand(uint64_t id, uint32_t path, uint32_t light, uint32_t sample, uint32_t comp);

/* ---- frame driver: main.cpp:43-56, 435-552 ---- */
/* corners[8][3]: origin00 dest00 origin01 dest01 origin10 dest10 origin11 dest11 */
void or_camera_corners(const double modelview[16], int W, int H, float corners[24]);
void or_pixel_ray(const float corners[24], int W, int H, double x, double y, float o[3],
                  float d[3]);
/* renders rows [y0,y1); outputs are full-frame W*H arrays, only the rows in
 * range are written.  rgb is the clamped RGBValue (imageWriter.h:23-44). */
void or_render_rows(const or_scene *s, const float corners[24], int W, int H, int y0, int y1,
                    float *rgb, float *depth, int32_t *primary_tri, int threads,
                    or_counters *c);

/* same; `centre` (may be NULL) additionally receives the work of the pixel-centre
 * rays alone (the first raytrace() call of main.cpp:440) */
void or_render_rows2(const or_scene *s, const float corners[24], int W, int H, int y0, int y1,
                     float *rgb, float *depth, int32_t *primary_tri, int threads, or_counters *c,
                     or_counters *centre);

/* ---- image: imageWriter.h ---- */
void or_normalize_depth(float *depth, int n, float focusDepth);
void or_gaussian_map(float intensity, float depthPower, float map[169]);
/* out: (W-12)*(H-12)*3 floats */
void or_blur(const float *rgb, const float *depthN, int W, int H, float depthPower, float *out,
             int threads);
void or_to_bytes(const float *v, int n, unsigned char *out);
int or_write_bmp(const char *path, const float *rgb, int W, int H);
int or_write_ppm(const char *path, const float *rgb, int W, int H);
int or_write_depth_bmp(const char *path, const float *rgb_unused, const float *depthN, int W,
                       int H);
int or_write_blurred_bmp(const char *path, const float *blur, int W, int H);
/* prev: what the previous writer left in the shared byte buffer (NULL = zeros) */
int or_write_blurred_bmp2(const char *path, const float *blur, int W, int H, const unsigned char *prev,
                          size_t n_prev);
/* Image::writeImageBMP(filename, x, y, w, h), imageWriter.h:472-488 */
int or_update_bmp_rows(const char *path, const float *rgb, int W, int x, int y, int w, int h);

#ifdef __cplusplus
}
#endif
#endif
