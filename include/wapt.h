/*
 * wapt.h — C ABI of the MI355X radiance-loop library (libwapt.so).
 *
 * The reference program has no plugin or FFI seam; the narrowest seam that
 * contains its whole per-pixel radiance loop is the pair
 *     udp_renderer::udp_renderer(...)      src/unidirectionalPathTracer.h:30
 *     udp_renderer::render(float*)         src/unidirectionalPathTracer.h:33
 * called from main() at src/myRenderer.cpp:149 and :156, with the scene
 * coming from wa_content (src/content.h:18-57).  The inner seam is the
 * closest-hit query wa_content::traceFAST (src/content.h:26,
 * src/content.cpp:91-123).  Every entry point below names the reference
 * interface it stands in for.
 *
 * Conventions
 *  - plain C types only; every pointer is a host pointer unless the name
 *    starts with d_ (device pointer on the scene's GPU);
 *  - every function returns 0 on success, a WAPT_E_* code otherwise, and
 *    never lets a C++ exception out; wapt_last_error() gives the text of
 *    the calling thread's last failure;
 *  - a wapt_scene owns its stream, queues and counters: calls on ONE scene
 *    must not overlap (wapt_progressive_abort excepted); different scenes,
 *    on the same or on different GPUs, may be used from different threads;
 *  - framebuffers are linear, un-tone-mapped mean radiance, float RGB,
 *    row-major, fb[(j*W+i)*3+c], row j=0 = bottom of the picture, exactly
 *    what udp_renderer::render leaves in the caller's buffer
 *    (src/unidirectionalPathTracer.cpp:141-161, src/fundamentals.cpp:82-87).
 */
#ifndef WAPT_H
#define WAPT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    WAPT_OK = 0,
    WAPT_E_INVALID = 1,   /* bad argument / inconsistent scene description */
    WAPT_E_IO = 2,        /* file could not be read / parsed / written     */
    WAPT_E_HIP = 3,       /* a HIP runtime call failed                     */
    WAPT_E_NO_DEVICE = 4, /* no usable gfx950 device                       */
    WAPT_E_LIMIT = 5,     /* a documented capacity limit was exceeded      */
    WAPT_E_ABORTED = 6    /* wapt_progressive_abort() ended the call       */
};

/* Material flag bits: the booleans of struct wa_shader, src/objects.h:32-39. */
enum {
    WAPT_MAT_OPAQUE = 1u << 0,        /* isOpaque      */
    WAPT_MAT_CLEAR = 1u << 1,         /* isClear       */
    WAPT_MAT_LIGHT = 1u << 2,         /* isLight       */
    WAPT_MAT_DIFFUSE = 1u << 3,       /* diffuse       */
    WAPT_MAT_SPECULAR = 1u << 4,      /* specular      */
    WAPT_MAT_SSS = 1u << 5,           /* sss           */
    WAPT_MAT_DIELECTRIC = 1u << 6,    /* dielectric    */
    WAPT_MAT_DIELECTRICSSS = 1u << 7  /* dielectricsss */
};

/* wa_camera constructor arguments, src/fundamentals.cpp:227. */
typedef struct wapt_camera {
    int32_t type;        /* 0 perspective, 1 orthographic */
    float P[3];          /* "from" */
    float D[3];          /* normalized(at - from), src/myRenderer.cpp:342 */
    float up[3];         /* "up" as given in the file (up_simple) */
    float fov_deg;       /* horizontal field of view, degrees */
    int32_t resI, resJ;  /* image width, height */
    float sensor_width;
} wapt_camera;

/* The fields of wa_shader that the radiance loop reads, src/objects.h:13-41. */
typedef struct wapt_material {
    float kd[3], ks[3], kt[3];
    float sa[3], ss[3], st[3];
    float ior;
    uint32_t flags; /* WAPT_MAT_* */
} wapt_material;

/* wa_areaLight (src/lights.cpp:24-34) / wa_sphereLight (src/lights.cpp:136-143). */
typedef struct wapt_light {
    int32_t type;    /* 0 area (LIGHT_AREA_TYPE), 1 sphere (LIGHT_SPHERE_TYPE) */
    float P[3];      /* corner (area) or centre (sphere) */
    float U[3], V[3];/* edge vectors (area only) */
    float radius;    /* sphere only */
    float power[3];
} wapt_light;

/*
 * Flattened scene = what wa_content holds after init() (src/content.cpp:13-40):
 * objects[0..n_lights-1] are the light placeholder objects, meshes follow;
 * a triangle's obj index is the Embree geomID of the reference and selects
 * materials[obj].  Area lights contribute their quad as two triangles
 * (v0,v1,v3),(v2,v3,v1) with prim 0; sphere lights contribute no geometry.
 */
typedef struct wapt_scene_desc {
    wapt_camera camera;
    wapt_material global_medium;      /* globalShader, src/myRenderer.cpp:20,666 */
    int32_t n_lights;
    const wapt_light* lights;
    int32_t n_objects;                /* lights + meshes */
    const wapt_material* materials;   /* [n_objects] */
    uint64_t n_tris;
    const float* tri_verts;           /* [n_tris*9]  v0 v1 v2, CCW = front */
    const int32_t* tri_obj;           /* [n_tris]    geomID                */
    const int32_t* tri_prim;          /* [n_tris]    primID inside geomID  */
    float world_center[3];            /* worldBoundSphere_P   (275,275,275), cpp:36 */
    float world_radius;               /* worldBoundSphere_rad 1100,          cpp:37 */
} wapt_scene_desc;

/* Constructor arguments of udp_renderer, src/unidirectionalPathTracer.cpp:26. */
typedef struct wapt_render_params {
    int32_t psamples;      /* samples per pixel of the WHOLE image (normalisation) */
    int32_t hsamples;
    int32_t lsamples;
    int32_t max_depth;     /* maxTraceDepth ("td") */
    float me;              /* rayMarchDistanceDistributionExponent */
    float mr;              /* rayMarchDistanceSampleRate */
    uint64_t seed;         /* counter-based RNG seed */
    int32_t sample_begin;  /* this call renders global sample indices      */
    int32_t sample_count;  /*   [sample_begin, sample_begin+sample_count)  */
    int32_t profile;       /* bit 0: bracket every kernel launch with HIP events;
                              bit 1: run the counting build of the closest-hit kernel */
    int32_t reserved;
} wapt_render_params;

typedef struct wapt_stats {
    uint64_t paths;           /* top-level radiance() calls = pixel samples      */
    uint64_t rays;            /* closest-hit queries (every traceFAST equivalent) */
    uint64_t stack_overflows; /* traversal-stack pushes dropped (must be 0)      */
    double seconds;           /* wall time of the render() equivalent            */
    /* filled when profile=1: per kernel family, HIP-event milliseconds and launches
     * (ms_head stays 0: the hit-record pass it timed is part of the surface kernel, ms_shade; ms_volume covers
     * the march kernel and the two small kernels that count and list its samples) */
    double ms_raygen, ms_trace, ms_head, ms_shade, ms_volume, ms_shadow, ms_probe;
    uint64_t n_trace_launches, n_trace_rays;
    uint64_t n_shadow_launches, n_shadow_rays;
    /* filled when profile bit 1 is set: totals over the closest-hit kernel's rays */
    uint64_t trace_node_visits, trace_tri_tests;
    /* same build, per wave: loop iterations, iterations whose node phase ran, trips of the triangle
     * loop (lane utilisation of a phase = visits or tests / (64 x its count)) */
    uint64_t trace_wave_steps, trace_node_steps, trace_tri_trips;
    /* the same five tallies for the rays of the shadow walks */
    uint64_t shadow_node_visits, shadow_tri_tests, shadow_wave_steps, shadow_node_steps, shadow_tri_trips;
    /* closest-hit queries of the reference's estimator that were answered without a search and are not in `rays`:
     * directLightHit / directLightingSingleScatter probes (cpp:631-687) only ask whether the closest hit is one given
     * light; a ray that misses that light's quad (or a light without geometry) cannot have it */
    uint64_t queries_decided;
    /* of `rays`: the searches of those two probe estimators (the rest of rays - n_trace_rays are shadow-walk steps) */
    uint64_t n_probe_rays;
    /* profile bit 1: the five tallies for the probe searches */
    uint64_t probe_node_visits, probe_tri_tests, probe_wave_steps, probe_node_steps, probe_tri_trips;
    /* device memory held by the scene's work queues after this call (their size follows the call's work, the free
     * memory and WAPT_QUEUE_BUDGET_GB), and the tasks per depth that one closest-hit launch can take */
    uint64_t queue_bytes, queue_tasks;
} wapt_stats;

typedef struct wapt_ray {
    float org[3];
    float tnear;
    float dir[3];
    float tfar;
} wapt_ray; /* 32 B, the fields of RTCRay that content.cpp:95 sets */

typedef struct wapt_hit {
    float t;        /* tfar after the query; unchanged input tfar on a miss */
    int32_t prim;   /* primID, -1 on a miss */
    int32_t obj;    /* geomID, -1 on a miss */
    float Ng[3];    /* un-normalised (v1-v0)x(v2-v0) */
} wapt_hit;

typedef struct wapt_bvh_info {
    uint64_t n_nodes, n_tris;
    uint64_t node_bytes, tri_bytes; /* total bytes of each array in HBM */
    uint32_t node_size, tri_size;   /* bytes per record */
    uint32_t max_depth;
    float sah_cost;
    uint32_t builder;     /* WAPT_BUILDER_* that produced the tree */
    float build_seconds;  /* build + upload */
} wapt_bvh_info;

typedef struct wapt_trace_counters {
    uint64_t rays, node_visits, tri_tests, stack_spills;
    uint64_t wave_steps, node_steps, tri_trips; /* per wave, as in wapt_stats */
} wapt_trace_counters;

typedef struct wapt_scene wapt_scene;        /* scene resident on one GPU */
typedef struct wapt_waff wapt_waff;          /* parsed .waff file (host) */

/* ---- host side of the reference kept as is -------------------------------- */

/* loadContentFromFile_waff, src/myRenderer.cpp:649-683 (+ the mesh readers of
 * src/raytracer.cpp:39-165).  Mesh paths are resolved against the current
 * directory first (as the reference does) and then against the directory of
 * the .waff file. */
int wapt_waff_load(const char* path, wapt_waff** out);
/* The flattened scene of a parsed file; pointers live as long as the wapt_waff. */
int wapt_waff_desc(const wapt_waff* w, wapt_scene_desc* desc_out);
/* traceDepth / rayMarch* globals, src/myRenderer.cpp:17-19, 350-365. */
int wapt_waff_settings(const wapt_waff* w, int32_t* td, float* me, float* mr);
void wapt_waff_free(wapt_waff* w);

/* --convertOBJ mode, src/myRenderer.cpp:90-119. */
int wapt_convert_obj_to_fast(const char* obj_path, const char* fast_path);
/* Tone map in place, src/myRenderer.cpp:174-185. */
int wapt_tonemap(float* fb, int32_t w, int32_t h);
/* writeImage, src/myRenderer.cpp:686-726 (expects tone-mapped values). */
int wapt_write_bmp(const char* path, int32_t w, int32_t h, const float* fb);

/* ---- the radiance loop on the GPU ----------------------------------------- */

/* wa_content::init + rtcCommitScene (src/content.cpp:13-40,
 * src/raytracer.cpp:172-175): builds the wide BVH on the host and uploads
 * scene, BVH and work queues to GPU `device`. */
int wapt_scene_create(const wapt_scene_desc* desc, int32_t device, wapt_scene** out);
/* Same with the BVH builder chosen by the caller.  The reference has one builder (Embree's, on the
 * CPU, inside rtcCommitScene).  Here: a device builder (Morton order, parallel locally-ordered
 * clustering, collapse to 8-wide nodes on the GPU: 0.2 s for 5.6e7 triangles) and a host builder
 * (binned SAH, threaded: 26 s for the same).  AUTO takes the device builder, whose trees trace as
 * fast or faster on all but one of the benchmark scenes (WAPT_BVH_BUILDER=host|device overrides;
 * WAPT_DEVICE_BVH=lbvh selects the older radix-tree construction inside the device builder). */
enum { WAPT_BUILDER_AUTO = 0, WAPT_BUILDER_HOST_SAH = 1, WAPT_BUILDER_DEVICE_LBVH = 2 };
int wapt_scene_create_with(const wapt_scene_desc* desc, int32_t device, int32_t builder, wapt_scene** out);
void wapt_scene_destroy(wapt_scene* s);
int wapt_scene_bvh_info(const wapt_scene* s, wapt_bvh_info* out);

/* udp_renderer::render(float*), src/unidirectionalPathTracer.cpp:78-103:
 * synchronous; overwrites all W*H*3 floats of the caller's host buffer with
 * mean radiance over params->psamples samples (sample_begin/sample_count must
 * cover 0..psamples here). */
int wapt_render(wapt_scene* s, const wapt_render_params* params, float* fb, wapt_stats* stats);

/* Same loop, for spp sharding across GPUs: ADDS the radiance sum (not mean) of
 * the samples [sample_begin, sample_begin+sample_count) into the caller's
 * W*H*3 float buffer in device memory and returns after the work has finished.
 * The caller zeroes the buffer, sums buffers of different GPUs and divides by
 * psamples. */
int wapt_render_accumulate(wapt_scene* s, const wapt_render_params* params, float* d_sum, wapt_stats* stats);

/* wa_content::traceFAST, src/content.cpp:91-123, for a batch of rays:
 * closest hit in (tnear, tfar], two-sided. */
int wapt_intersect(wapt_scene* s, const wapt_ray* rays, uint32_t n, wapt_hit* hits);
/* Same with rays and hits already in device memory; `counters` (host, may be
 * NULL) switches to the counting build of the same traversal and `ms` (host,
 * may be NULL) receives the HIP-event time of the kernel alone. */
int wapt_intersect_device(wapt_scene* s, const wapt_ray* d_rays, uint32_t n, wapt_hit* d_hits,
                          wapt_trace_counters* counters, float* ms);

/* No counterpart in the reference.  wapt_intersect_device for callers that cannot allocate device memory
 * (tight serving loops, harnesses that forbid hipMalloc, code that runs while another stream of the process is
 * under HIP graph capture, where allocation calls are forbidden): the same closest-hit batch, with the launch's
 * work-feed counters, result counters and traversal-stack spill area in a 32 MiB array that is part of the
 * library's code object (one copy per GPU, held until the process ends) instead of memory the library allocates.
 * Calls no hipMalloc / hipFree.  The call is NOT itself capturable: it runs at once on the scene's own stream,
 * copies its overflow counter back and blocks until the batch is done; d_rays must be complete before the call
 * (it is not ordered after work on the caller's streams).  At most 2^30 rays per call (WAPT_E_LIMIT beyond); a tree
 * deeper than the spill area holds at full width runs with fewer blocks.  All calls on one GPU share that GPU's
 * array, whatever scene they use: calls that overlap on one GPU race. */
int wapt_intersect_device_static(wapt_scene* s, const wapt_ray* d_rays, uint32_t n, wapt_hit* d_hits);

/* ---- progressive rendering: the interactive-preview interface of udp_renderer -------------- */

typedef struct wapt_progressive wapt_progressive;
/* struct tile, src/unidirectionalPathTracer.h:14-18 */
typedef struct wapt_tile {
    int32_t l, r, b, t;
    int32_t w, h;
} wapt_tile;

/* The accumulating state of udp_renderer (internalBuf + itCnt + render resolution + region,
 * src/unidirectionalPathTracer.cpp:50-62): a display-sized running-average buffer on the scene's GPU.
 * params->psamples is the number of samples one iteration adds; sample_begin/sample_count are ignored. */
int wapt_progressive_create(wapt_scene* s, const wapt_render_params* params, wapt_progressive** out);
void wapt_progressive_destroy(wapt_progressive* p);
/* udp_renderer::reset, cpp:907-917: iteration count 0, buffer black. */
int wapt_progressive_reset(wapt_progressive* p);
/* udp_renderer::setIterationCount, cpp:919-922 (render() itself never changes the count, as in the
 * reference: the caller advances it). */
int wapt_progressive_set_iteration(wapt_progressive* p, int32_t iteration);
/* udp_renderer::setRenderResolution + lowerResolution, cpp:68-76, 924-934: render at 1/2^k of the
 * display resolution and replicate every pixel into a 2^k x 2^k block; resets the region. */
int wapt_progressive_set_resolution(wapt_progressive* p, int32_t downres_pow2);
/* udp_renderer::setRegion, cpp:962-972, in pixels of the render resolution: [l, r) x [b, t).  The
 * reference stores the region but its render loop ignores it; here only the region is rendered. */
int wapt_progressive_set_region(wapt_progressive* p, int32_t l, int32_t r, int32_t b, int32_t t);
/* One iteration = udp_renderer::render with itCnt = n: psamples new samples per pixel of the region,
 * internal = n/(n+1) internal + 1/(n+1) new (cpp:141-145), then the frame buffer (display size, host)
 * receives the internal buffer with pixel replication (cpp:152-159).  Iteration n uses the global sample
 * indices [n psamples, (n+1) psamples). */
int wapt_progressive_render(wapt_progressive* p, float* fb, wapt_stats* stats);
/* udp_renderer::abort, cpp:936-939, callable from another thread: the running iteration finishes the
 * paths it has started, starts no more, leaves the running average untouched and returns WAPT_E_ABORTED. */
int wapt_progressive_abort(wapt_progressive* p);
/* udp_renderer::generateTiles, cpp:941-960: fills 4^level tiles, *n_out = their number. */
int wapt_generate_tiles(int32_t renW, int32_t renH, int32_t subdiv_level, wapt_tile* tiles, int32_t* n_out);

const char* wapt_last_error(void);
const char* wapt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* WAPT_H */
