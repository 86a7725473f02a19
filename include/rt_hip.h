/*
 * rt_hip.h - C ABI of the MI355X (gfx950) ray-tracing library, librt_hip.so.
 *
 * The library is the GPU replacement for the per-pixel render loop of the
 * reference CPU ray tracer: everything that runs per pixel after key 'r'.
 * Each entry point names the reference interface it stands in for
 * (file:line in the reference checkout).  All entry points return 0 on
 * success or a negative RtError; rt_last_error() gives the message of the last
 * failure on the calling thread.  No C++ exception crosses this boundary.
 *
 * Threading: calls on one RtScene are serialised by the caller.  Entry points
 * without a stream argument are synchronous; the *_device entry points enqueue
 * on the given HIP stream and return without waiting unless a stats pointer is
 * passed.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum RtError {
    RT_OK = 0,
    RT_ERR_INVALID = -1,     /* bad argument (null pointer, size out of range, bad index) */
    RT_ERR_HIP = -2,         /* a HIP runtime call failed */
    RT_ERR_UNSUPPORTED = -3, /* valid request the library cannot serve (tree too deep, ...) */
    RT_ERR_NO_DEVICE = -4
} RtError;

#define RT_MAX_LIGHTS 64

typedef struct RtScene RtScene;

/* One material as the shading loop reads it (mesh.h:141-147, 170-178).
 * has_Ni mirrors Material::has_Ni(); Ni carries the stored value even when
 * has_Ni is 0 because raytracing.cpp:206 tests Ni() without the flag. */
typedef struct RtMaterial {
    float   Kd[3];
    float   Ks[3];
    float   Ns;
    float   Ni;
    int32_t illum;
    int32_t has_Ni;
    int32_t texture; /* index into RtSceneDesc.textures, -1 = hasTexture() false */
    int32_t reserved[5];
} RtMaterial;

/* RGBA8 image, row major, top row first (corona::PF_R8G8B8A8, mesh.h:121).
 * The library appends two rows of zero bytes: mesh.cpp:499 can index one row
 * past the image. */
typedef struct RtTexture {
    const uint8_t *rgba;
    int32_t        width, height;
} RtTexture;

/* One AccelTreeNode (KDtree.h:9-27) with pointers replaced by indices.
 * Node 0 is treeRoot; left == right == -1 marks a leaf. */
typedef struct RtKdNode {
    float   bounds[6]; /* xStart xEnd yStart yEnd zStart zEnd */
    int32_t left, right;
    int32_t tri_offset, tri_count; /* range in RtSceneDesc.node_tris */
} RtKdNode;

/* The globals the reference's tracer reads (raytracing.h:12-21): MyMesh and
 * treeRoot, flattened by the host. */
typedef struct RtSceneDesc {
    int32_t           n_vertices;
    const float      *vertex_pos;    /* 3 per vertex: Vertex::p */
    const float      *vertex_normal; /* 3 per vertex: Vertex::n after computeVertexNormals */
    int32_t           n_triangles;
    const int32_t    *tri_vertex;    /* 3 per triangle: Triangle::v */
    const int32_t    *tri_texcoord;  /* 3 per triangle: Triangle::t */
    const int32_t    *tri_material;  /* Mesh::triangleMaterials */
    int32_t           n_texcoords;
    const float      *texcoords;     /* 3 per entry (Vec3Df, z unused) */
    int32_t           n_materials;
    const RtMaterial *materials;
    int32_t           n_textures;
    const RtTexture  *textures;
    int32_t           n_nodes;
    const RtKdNode   *nodes;
    int32_t           n_node_tris;
    const int32_t    *node_tris;     /* concatenated AccelTreeNode::triangles */
} RtSceneDesc;

/* settings.h:7-19 and config.h's WindowSize as run-time values, plus the
 * screen-space partition used when several GPUs share a frame. */
typedef struct RtParams {
    int32_t width, height;     /* WindowSize_X / WindowSize_Y */
    int32_t msaa;              /* 1, 4 or 16 (settings.h:15) */
    float   backgroundlighting; /* settings.h:7 */
    float   diffusePower;      /* settings.h:8 */
    float   ShadowFactor;      /* settings.h:10 */
    float   depthPower;        /* settings.h:16 */
    float   backgroundColor[3]; /* settings.h:18 */
    float   focusDepth;        /* raytracing.cpp:17 */
    /* Tiles of tile_w x tile_h pixels (multiples of 8) are dealt round-robin
     * over `world` shares: this call renders the tiles t whose share t % world
     * lies in [rank, rank + rank_span).  rank_span <= 0 means 1, i.e. the plain
     * round-robin "t % world == rank"; a caller that wants one GPU to carry
     * less of the frame than the others (it also runs the post-process, say)
     * uses more shares than GPUs and hands them out unevenly.  tile_w == 0
     * means the whole frame (rank 0 of 1). */
    int32_t tile_w, tile_h;
    int32_t rank, world;
    int32_t rank_span;
    /* settings.h:12-13.  shadowSamples <= 1: one shadow ray per light (the reference's default);
     * n = 2..6: the soft shadows of raytracing.cpp:145-160, up to n^3 jittered shadow rays per
     * light towards points around it (the float loops of the reference decide how many).  The
     * reference jitters with rand(), whose order over its threads is not reproducible; here the
     * three numbers of a sample come from a counter-based generator keyed by pixel, sample,
     * position in the ray tree, light and sample ordinal, so frames are repeatable. */
    int32_t shadowSamples;
    float   shadowRadius;
} RtParams;

typedef struct RtStats {
    uint64_t rays_closest; /* closest-hit queries traced on the GPU */
    uint64_t rays_shadow;  /* shadow (any-hit) queries traced on the GPU */
    uint64_t tri_tests;    /* ray/triangle tests executed */
    uint64_t node_visits;  /* KD nodes visited */
    uint64_t camera_rays;  /* camera samples evaluated (incl. centre visibility rays) */
    float    kernel_ms;    /* HIP-event time of the call's kernels */
    float    reserved;
    uint64_t shaded_hits;  /* closest hits that were shaded (raytracing.cpp:76-164) */
    uint64_t child_rays;   /* reflection / refraction rays spawned (raytracing.cpp:166-271) */
} RtStats;

/* The scene as laid out on the device: the reference's tree with its inner-node
 * triangles handed down to the leaves and large leaves split further, so
 * n_nodes / tree_depth can exceed the input's; n_slots counts triangle
 * references (a triangle is listed by every leaf it touches). */
typedef struct RtSceneInfo {
    int32_t  n_nodes, n_triangles, tree_depth, n_slots;
    uint64_t device_bytes;
    uint64_t tree_hash;  /* FNV-1a over the device tree's nodes and slots: equal trees, equal hash */
    float    create_ms;  /* wall time of rt_scene_create */
    float    tree_ms;    /* of which: handing triangles down + splitting leaves */
    int32_t  tree_on_gpu; /* 1: those two steps ran on the device, 0: on the host (RT_KD_GPU=0) */
    int32_t  render_path; /* what rt_render_frame* runs for this scene: 0 the single persistent launch over ray
                           * queues (the product path), 1 the same stages as one launch each
                           * (RT_WF_MULTILAUNCH=1), 2 one thread per camera sample (RT_MEGAKERNEL=1 or
                           * RT_KD_PUSHDOWN=0: ablation switches; an order of magnitude slower) */
} RtSceneInfo;

/* Per-kernel device time of the render calls since timing was switched on or
 * last collected, from HIP events recorded on the render stream around each
 * launch (no host synchronisation inside the render call). */
typedef struct RtKernelTimes {
    double  trace_closest_ms; /* closest-hit trace kernel launches */
    double  trace_any_ms;     /* any-hit (shadow) trace kernel launches */
    double  frame_ms;         /* first to last kernel of each render call */
    int32_t trace_closest_launches, trace_any_launches, frames;
    int32_t reserved;
} RtKernelTimes;

/* RtParams with the reference's constants. */
void rt_default_params(RtParams *p, int32_t width, int32_t height);

int         rt_device_count(void);
const char *rt_last_error(void);

/* buildKDtree (KDtree.cpp:15-230: splitSpaces and the 999-candidate scan of calcBestSplit) on the device:
 * the tree the reference builds for these triangles, node for node - bounds, children, the triangles every
 * node keeps and their order - numbered as the reference's recursion creates nodes (children when their
 * parent is expanded, left subtree first).  nodes: room for max_nodes (4 * n_triangles + 64 always
 * suffices); node_tris: room for n_triangles; depth: the deepest node (may be null).  What comes out is what
 * RtSceneDesc.nodes / node_tris take. */
int  rt_build_kd_tree(int device, int32_t n_vertices, const float *vertex_pos, int32_t n_triangles,
                      const int32_t *tri_vertex, int32_t max_nodes, RtKdNode *nodes, int32_t *n_nodes,
                      int32_t *node_tris, int32_t *depth);

/* Takes over from the end of init() (raytracing.cpp:23-53): the mesh is
 * loaded, normals computed, the KD tree built; this call lays both out for
 * the GPU and uploads them to `device`.  The host arrays are not referenced
 * after the call returns. */
int  rt_scene_create(const RtSceneDesc *desc, int device, RtScene **out);
void rt_scene_destroy(RtScene *scene);
int  rt_scene_info(const RtScene *scene, RtSceneInfo *info);
/* Bench instrumentation, no counterpart in the reference (its only timer is the
 * clock() pair of main.cpp:327,368).  rt_scene_set_timing switches event
 * recording on or off; rt_scene_collect_timing waits for the device, sums the
 * recorded intervals per kernel and resets them.  At most 4096 launches are
 * kept between two collections; later ones are not timed. */
int  rt_scene_set_timing(RtScene *scene, int enable);
int  rt_scene_collect_timing(RtScene *scene, RtKernelTimes *out);

/* checkHit(origin, dest, minT) (raytracing.h:77, raytracing.cpp:281-351) for n
 * rays.  tri is the index into Mesh::triangles of the closest hit, -1 for a
 * miss; a, b, t, hitpoint as HitTriangle/minT.  Any output may be NULL. */
int rt_check_hits(RtScene *scene, int64_t n, const float *origins, const float *dests,
                  int32_t *tri, float *a, float *b, float *t, float *hitpoint);

/* performRayTracing(origin, dest, k, prev_Ni, minT) (raytracing.h:33,
 * raytracing.cpp:61-274) for n rays sharing k (0..5) and prev_Ni; the frame
 * driver calls it with k = 0, prev_Ni = 1.0f (main.cpp:528).  rgb is the
 * returned Vec3Df (unclamped), min_t the depth of the first hit (FLT_MAX on a
 * miss), tri its triangle. */
int rt_trace_rays(RtScene *scene, const float *lights, int32_t n_lights, const RtParams *params,
                  int32_t k, float prev_Ni, int64_t n, const float *origins, const float *dests,
                  float *rgb, float *min_t, int32_t *tri, RtStats *stats);

/* One frame of the 'r' path: RayTracer::threadmethod over every pixel
 * (main.cpp:391-408) = raytraceMSAA (main.cpp:435-511) + Image::setPixel with
 * the RGBValue clamp (imageWriter.h:23-44, 99-105).  camera = the eight corner
 * points of main.cpp:52-55 in the order origin00 dest00 origin01 dest01
 * origin10 dest10 origin11 dest11.  Outputs are width*height, row major:
 * rgb 3 floats per pixel, depth 1 float, tri (may be NULL) the triangle hit by
 * the pixel-centre ray.  Host pointers. */
int rt_render_frame(RtScene *scene, const float camera[24], const float *lights, int32_t n_lights,
                    const RtParams *params, float *rgb, float *depth, int32_t *tri,
                    RtStats *stats);

/* The whole 'r' path (main.cpp:322-378 without the file writes) with the frame kept on the
 * device: the frame, Image::normailzeDepthRange twice and the blur, then the three 8-bit
 * planes the reference's writers derive from them - rgb8: (unsigned char)(v*255.0f) of the
 * colour plane, width*height*3, RGB order (imageWriter.h:270, 322-324); depth8: the grey value
 * of depth.bmp, (unsigned char)((1-depth)*255.0f) of the plane normalised once, width*height
 * (imageWriter.h:387); blurred8: the blurred plane quantised the same way, (width-12)*
 * (height-12)*3, NULL for frames too small to blur (imageWriter.h:452-454).  The pointers
 * returned lie in pinned host memory owned by the scene; they stay valid until the next call
 * on this scene or rt_scene_destroy.  Whole frames only (params->tile_w == 0). */
int rt_render_result_planes(RtScene *scene, const float camera[24], const float *lights,
                            int32_t n_lights, const RtParams *params, const uint8_t **rgb8,
                            const uint8_t **depth8, const uint8_t **blurred8, RtStats *stats);

/* Same with device pointers on the scene's device, enqueued on `stream`
 * (a hipStream_t).  Pixels outside this rank's tiles are not written.  With
 * stats != NULL the call waits for completion and fills it. */
int rt_render_frame_device(RtScene *scene, const float camera[24], const float *lights,
                           int32_t n_lights, const RtParams *params, float *d_rgb, float *d_depth,
                           int32_t *d_tri, void *stream, RtStats *stats);

/* Image::normailzeDepthRange (imageWriter.h:113-135), in place, n values. */
int rt_normalize_depth(int device, int64_t n, float focus_depth, float *depth);
int rt_normalize_depth_device(int device, int64_t n, float focus_depth, float *d_depth,
                              void *stream);

/* No counterpart in the reference.  For callers that cannot allocate device
 * memory - code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc: rt_normalize_depth_device with its scratch (the per-block
 * partial minima and maxima, 8 KB) in a fixed device array of the library
 * instead of the stream's memory pool.  Calls no hipMalloc / hipFree or any
 * other allocating routine.  n may not exceed RT_NOALLOC_MAX_DEPTH_VALUES.
 * The array's address is looked up once per device (hipGetSymbolAddress, which
 * may make the runtime load the library's code object): by rt_scene_create on
 * that device, else by the first call here - a caller that has made no scene
 * on the device makes one warm-up call before it enters the region in which it
 * may not allocate.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU
 * (different streams or threads) race; calls on one stream are safe. */
#define RT_NOALLOC_MAX_DEPTH_VALUES ((int64_t)1 << 28) /* a 16384 x 16384 frame */
int rt_normalize_depth_device_noalloc(int device, int64_t n, float focus_depth, float *d_depth,
                                      void *stream);

/* Image::blurrAllLines (imageWriter.h:137-237): depth-dependent 13x13 blur of
 * the float image; depth_n is the normalised depth.  out is
 * (width-12)*(height-12)*3 floats. */
int rt_dof_blur(int device, int32_t width, int32_t height, float depth_power, const float *rgb,
                const float *depth_n, float *out);
int rt_dof_blur_device(int device, int32_t width, int32_t height, float depth_power,
                       const float *d_rgb, const float *d_depth_n, float *d_out, void *stream);

/* The arithmetic of writeDepthBMP + writeBlurredBMP (imageWriter.h:352, 416-418) on a frame in
 * device memory, in one call: d_depth (width*height, FLT_MAX = miss) is normalised in place -
 * afterwards it is the plane depth.bmp shows - and d_blurred (may be NULL: no blur) receives
 * the (width-12)*(height-12)*3 floats of blurred.bmp, computed from the plane normalised a second
 * time as the reference does; that second normalisation is applied as the blur reads the plane,
 * it is not stored.  Same values as rt_normalize_depth_device twice + rt_dof_blur_device. */
int rt_postprocess_device(int device, int32_t width, int32_t height, float focus_depth, float depth_power,
                          const float *d_rgb, float *d_depth, float *d_blurred, void *stream);

/* (unsigned char)(v * 255.0f) of imageWriter.h:270,322-324 for n floats. */
int rt_quantize(int device, int64_t n, const float *v, uint8_t *out);

/* Screen-space partition helpers.  They replace the row bands of
 * main.cpp:329-339 when a frame is shared by several GPUs.  Where a rank is
 * passed explicitly it replaces params->rank; params->rank_span is the span of
 * THAT rank (the caller sets it to the source's span before unpacking).
 * rt_tile_pixels: number of pixel slots (whole tiles) rank owns.
 * rt_pack_tiles_device: copy the rank's tiles of the frame buffers into a
 *   payload of rt_tile_pixels * 4 floats (r g b depth per pixel slot).
 * rt_unpack_tiles_device: the inverse, for the tiles of `src_rank`. */
int64_t rt_tile_pixels(const RtParams *params, int32_t rank);
int     rt_pack_tiles_device(int device, const RtParams *params, const float *d_rgb,
                             const float *d_depth, float *d_payload, void *stream);
int     rt_unpack_tiles_device(int device, const RtParams *params, int32_t src_rank,
                               const float *d_payload, float *d_rgb, float *d_depth, void *stream);

/* The exchange step of a frame shared by several GPUs, one process per GPU: every rank's tile
 * payload (rt_pack_tiles_device) is gathered on rank 0 over RCCL - ncclGroupStart, one ncclRecv
 * per peer on rank 0 / one ncclSend on every other rank, ncclGroupEnd - so each peer uses its own
 * xGMI link to rank 0.  It replaces the shared Image the reference's threads write into
 * (main.cpp:329-339, 404).  comm is an ncclComm_t, from rt_rccl_comm_create or the caller's own.
 * floats_per_rank[r] = rt_tile_pixels(params of rank r) * 4; on rank 0 d_gathered receives the
 * payloads one after the other in rank order (d_gathered may be NULL elsewhere).  Enqueued on
 * `stream`; RCCL is loaded on first use (librccl.so.1), RT_ERR_UNSUPPORTED if it is not there. */
#define RT_RCCL_ID_BYTES 128
int  rt_rccl_unique_id(void *id128);            /* ncclGetUniqueId, on rank 0; hand the bytes to every rank */
int  rt_rccl_comm_create(const void *id128, int32_t n_ranks, int32_t rank, int device, void **comm);
void rt_rccl_comm_destroy(void *comm);
int  rt_gather_tiles_rccl(void *comm, int32_t rank, int32_t n_ranks, const float *d_payload,
                          const int64_t *floats_per_rank, float *d_gathered, void *stream);

/* The deal of a shared frame.  The reference gives every thread an equal band of rows and lets a
 * thread that is through steal rows from the others (main.cpp:329-339, 410-429); here the tiles are
 * dealt once, unevenly: 32 x 32 tiles go round-robin over first_rank_shares + RT_SHARES_PER_RANK *
 * (n_ranks - 1) shares, rank 0 renders the first first_rank_shares of them (1..RT_SHARES_PER_RANK: it
 * also unpacks the gathered tiles and runs the post-process), every other rank RT_SHARES_PER_RANK.
 * rt_deal_params: `frame` with tile size, rank, world and rank_span filled in for `rank` of that deal
 *   (what rt_render_frame_device, rt_tile_pixels and the pack / unpack calls take).
 * rt_scene_set_deal: the first_rank_shares later rt_render_result_planes_shared calls on this scene
 *   use (RT_SHARES_PER_RANK, an even deal, until it is set); every rank must set the same value.
 * rt_calibrate_deal: measures and sets it, a collective call of all ranks before the frames that
 *   count.  Rank 0 times its render under an even deal and under a halved share (time = slope * share
 *   of the frame + tail), renders the whole frame once to time its post-process on it, and takes the
 *   share f0 = (1 - (n_ranks - 1) * (t_post + 0.1 ms) / slope) / n_ranks, clamped to [0.2, 1] / n_ranks;
 *   the result travels to the other ranks over the communicator.  Returns it in *first_rank_shares. */
#define RT_SHARES_PER_RANK 16
int rt_deal_params(const RtParams *frame, int32_t n_ranks, int32_t first_rank_shares, int32_t rank,
                   RtParams *out);
int rt_scene_set_deal(RtScene *scene, int32_t first_rank_shares);
int rt_calibrate_deal(RtScene *scene, void *comm, int32_t rank, int32_t n_ranks, const float camera[24],
                      const float *lights, int32_t n_lights, const RtParams *params,
                      int32_t *first_rank_shares);

/* The order in which a frame's regions (the units of 64 to 256 pixels a block of the render launch takes)
 * are handed out.  The reference's threads take rows top to bottom and steal from each other when they run
 * dry (main.cpp:410-429); here a launch ends when its last region does, and a region that mirrors keep busy
 * takes ten to twenty times the median.  from_last_frame = 1 (the default; RT_WF_FEEDBACK=0 in the
 * environment at rt_scene_create makes 0 the default): every launch leaves the duration of each region
 * behind, and the next launch of the same shape (frame size, sampling, this rank's tiles) starts the regions
 * it would otherwise end up waiting for first.  A hint about when a region runs, never about what it
 * renders: the frame is the same bit for bit.  0: always by index (what the first frame of a shape gets). */
int rt_scene_set_region_order(RtScene *scene, int32_t from_last_frame);

/* rt_render_result_planes for a frame shared by n_ranks processes, one per GPU (the C++ host's
 * replacement of the thread bands of main.cpp:329-339): this rank renders its tiles of the scene's
 * deal (above), the payloads are gathered on rank 0 (rt_gather_tiles_rccl), and rank 0 rebuilds the
 * frame, post-processes it and receives the planes; on the other ranks the three pointers come
 * back NULL.  n_ranks == 1 needs no communicator.  Every rank calls it with the same camera,
 * lights and params. */
int rt_render_result_planes_shared(RtScene *scene, void *comm, int32_t rank, int32_t n_ranks,
                                   const float camera[24], const float *lights, int32_t n_lights,
                                   const RtParams *params, const uint8_t **rgb8, const uint8_t **depth8,
                                   const uint8_t **blurred8, RtStats *stats);

/* The shared frame rehearsed on ONE GPU: the parts of all n_ranks ranks of the scene's deal are
 * rendered here one after the other, each but rank 0's into a frame of its own, packed, laid out as
 * the gather would lay them out, unpacked into rank 0's frame and post-processed - every step of
 * rt_render_result_planes_shared but the transfer, for machines with fewer GPUs than ranks.  The
 * planes equal those of rt_render_result_planes byte for byte (a frame does not depend on its
 * partition). */
int rt_render_result_planes_rehearsal(RtScene *scene, int32_t n_ranks, const float camera[24],
                                      const float *lights, int32_t n_lights, const RtParams *params,
                                      const uint8_t **rgb8, const uint8_t **depth8,
                                      const uint8_t **blurred8, RtStats *stats);

#ifdef __cplusplus
}
#endif
#endif
