/*
 * pathed_hip.h — C ABI of the MI355X path-tracing integrator (libpathed_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of chellmuth/pathed: the per-pixel
 * radiance loop  Integrator::run -> SampleIntegrator::sampleImage/samplePixel ->
 * PathTracer::L / direct  (reference src/integrator.cpp:19-106,
 * src/sample_integrator.cpp:10-113, src/path_tracer.cpp:19-216) together with the
 * ray queries it makes through Scene (reference src/scene.cpp:91-223, 355-381,
 * 446-502), which the reference forwards to Embree (rtcIntersect1 / rtcOccluded1).
 *
 * The reference has no FFI: its plug-in surface is the C++ virtual class
 * `Integrator` (reference include/integrator.h:16-55) chosen by the "integrator"
 * string of job.json (reference src/job.cpp:65-97).  A GPU integrator subclasses
 * `Integrator`, overrides run(), and calls the functions below.  The binding a
 * reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every call returns 0 on success
 *     or a negative PATHED_E_* code, message via pathed_hip_last_error()
 *     (reference error behaviour: throw "Unimplemented" / std::runtime_error /
 *     exit(1), src/job.cpp:96, src/scene_parser.cpp:665, src/glass.cpp:69-72).
 *   - caller owns every array in PathedSceneDesc; scene_create copies what it needs.
 *   - a PathedScene is used by one host thread at a time; render calls block.
 *   - all arithmetic is IEEE fp32; indices are int32/uint32.
 */
#ifndef PATHED_HIP_H
#define PATHED_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PATHED_ABI_VERSION 4   /* 2: image-texture albedo (PathedTexture, PathedMaterial.texture)
                                * 3: participating media (PathedMedium, PathedGeom.medium, PATHED_MAT_PASSTHROUGH)
                                * 4: the process-global pathed_hip_set_bvh_builder is gone (PathedSceneOptions.bvh_builder);
                                *    PathedSceneOptions.build_threads; pathed_hip_comm_* (RCCL reduce of the radiance sums) */
/* Additions that leave PathedSceneDesc unchanged (no version bump): PathedSceneOptions /
 * pathed_hip_scene_create_ex (per-scene device and tuning), pathed_hip_measure_valu,
 * pathed_hip_accum_add / pathed_hip_accum_copy_peer (multi-GPU fan-in of the radiance sums). */

/* error codes */
#define PATHED_OK            0
#define PATHED_E_INVALID    -1   /* bad argument / malformed scene description   */
#define PATHED_E_DEVICE     -2   /* HIP runtime error (message has the HIP text)  */
#define PATHED_E_NO_DEVICE  -3   /* no usable gfx950 device                        */
#define PATHED_E_UNSUPPORTED -4  /* feature outside the hot-path scope             */
#define PATHED_E_NOMEM      -5

/* material kinds — reference Material subclasses on the hot path
 * (src/lambertian.cpp, src/oren_nayar.cpp, src/microfacet.cpp + src/beckmann.cpp,
 *  src/plastic.cpp, src/glass.cpp, src/mirror.cpp) */
#define PATHED_MAT_LAMBERTIAN 0
#define PATHED_MAT_OREN_NAYAR 1
#define PATHED_MAT_MICROFACET 2
#define PATHED_MAT_PLASTIC    3
#define PATHED_MAT_GLASS      4
#define PATHED_MAT_MIRROR     5
#define PATHED_MAT_PASSTHROUGH 6 /* reference src/passthrough.cpp: the boundary of a participating medium ("container") */

/* Lambertian albedo source (reference include/albedo.h, src/checkerboard.cpp:9-20) */
#define PATHED_ALBEDO_CONSTANT     0
#define PATHED_ALBEDO_CHECKERBOARD 1
#define PATHED_ALBEDO_TEXTURE      2   /* reference src/texture.cpp:33-49 */

/* microfacet distribution (reference src/scene_parser.cpp:669-683) */
#define PATHED_DIST_BECKMANN 0
#define PATHED_DIST_GGX      1   /* reference src/ggx.cpp */

/* geometry kinds in model order (reference src/scene_parser.cpp:251-291) */
#define PATHED_GEOM_MESH   0
#define PATHED_GEOM_SPHERE 1

/* Camera — reference Camera ctor arguments (src/camera.cpp:13-30), as parsed by
 * src/scene_parser.cpp:146-158.  vertical_fov is already in radians. */
typedef struct PathedCamera {
    float origin[3];
    float target[3];
    float up[3];
    float vertical_fov;      /* radians */
    int32_t width;           /* job.json "width"  (reference include/job.h:24) */
    int32_t height;          /* job.json "height" (reference include/job.h:25) */
    int32_t flip_handedness; /* scene "sensor.flipHandedness"                  */
} PathedCamera;

/* One BSDF.  Only the fields of `type` are read.
 * emit != 0 makes every surface with this material an area light
 * (reference src/scene_parser.cpp:173-183). */
typedef struct PathedMaterial {
    int32_t type;            /* PATHED_MAT_*                                   */
    int32_t albedo_type;     /* PATHED_ALBEDO_* (Lambertian / Plastic diffuse) */
    float diffuse[3];        /* Lambertian, OrenNayar, Plastic                 */
    float emit[3];           /* Material::emit()                               */
    float checker_on[3];     /* Checkerboard onColor                           */
    float checker_off[3];    /* Checkerboard offColor                          */
    float checker_res[2];    /* Checkerboard resolution (u, v)                 */
    float sigma;             /* OrenNayar sigma (A, B derived as oren_nayar.cpp:11-18) */
    float alpha;             /* Beckmann alpha                                 */
    float ior;               /* Glass ior (reference default 1.4, glass.cpp:16-18) */
    int32_t distribution;    /* PATHED_DIST_*                                  */
    int32_t texture;         /* PATHED_ALBEDO_TEXTURE: index into PathedSceneDesc.textures */
} PathedMaterial;

/* Image texture — reference Texture (src/texture.cpp): what stbi_load(path, .., 3) returns,
 * 8-bit RGB, row 0 = first row of the file.  The lookup wraps uv, flips v, rounds to the
 * nearest texel and applies powf(x / 255, 2.2) (texture.cpp:33-49). */
typedef struct PathedTexture {
    int32_t width;           /* 1 .. 65535                                     */
    int32_t height;
    const uint8_t *rgb;      /* 3*width*height                                 */
} PathedTexture;

/* Sphere — reference Sphere (src/sphere.cpp).  The reference intersects the
 * TRANSFORMED centre (sphere.cpp:30-35) but samples the UNTRANSFORMED m_center
 * (sphere.cpp:77-128); both are carried so that quirk is reproducible. */
typedef struct PathedSphere {
    float center_world[3];
    float radius;
    float center_sample[3];
    int32_t material;
} PathedSphere;

/* One entry per model of the scene JSON, in file order (= Embree geomID order,
 * reference src/rtc_manager.cpp:12-27).  Light order follows it
 * (reference src/scene_parser.cpp:173-183). */
typedef struct PathedGeom {
    int32_t type;   /* PATHED_GEOM_*                                            */
    int32_t first;  /* MESH: first triangle index; SPHERE: index into spheres  */
    int32_t count;  /* MESH: triangle count;       SPHERE: 1                   */
    int32_t medium; /* the model's "internal_medium" (reference src/scene_parser.cpp:324-337, Surface::getInternalMedium):
                       index into PathedSceneDesc.media, -1 = none               */
} PathedGeom;

/* Homogeneous participating medium — reference HomogeneousMedium (src/homogeneous_medium.cpp; scene JSON "media",
 * src/scene_parser.cpp:221-226).  The reference's transmittance uses all three channels of sigma_t, its distance
 * sampling the red one (it asserts the three are equal). */
typedef struct PathedMedium {
    float sigma_t[3];
    float sigma_s[3];
} PathedMedium;

/* Environment light — reference EnvironmentLight (src/environment_light.cpp).
 * rgba is the float RGBA image LoadEXR returns (row 0 = theta 0), map_to_world /
 * world_to_map are the 4x4 row-major matrices parseTransform builds
 * (src/scene_parser.cpp:690-793). */
typedef struct PathedEnvLight {
    int32_t width;
    int32_t height;
    const float *rgba;        /* 4*width*height                                 */
    float scale;
    float map_to_world[16];
    float world_to_map[16];
} PathedEnvLight;

/* Flat scene description: what the reference hands Embree
 * (src/geometry_parser.cpp:12-96: vertex / index / uv / normal buffers per mesh)
 * plus its material, light and camera objects. */
typedef struct PathedSceneDesc {
    uint32_t abi_version;     /* PATHED_ABI_VERSION                             */

    PathedCamera camera;

    /* triangle soup, all meshes concatenated in geom order */
    uint32_t n_vertices;
    const float *positions;   /* 3*n_vertices, world space                      */
    const float *normals;     /* 3*n_vertices, zero vector = "no normal"        */
    const float *uvs;         /* 2*n_vertices                                   */
    uint32_t n_triangles;
    const uint32_t *indices;  /* 3*n_triangles, into the vertex arrays          */
    const int32_t *tri_material; /* n_triangles, index into materials           */

    uint32_t n_spheres;
    const PathedSphere *spheres;

    uint32_t n_geoms;
    const PathedGeom *geoms;

    uint32_t n_materials;
    const PathedMaterial *materials;

    const PathedEnvLight *env; /* NULL = no environment light                   */

    uint32_t n_textures;
    const PathedTexture *textures;

    uint32_t n_media;
    const PathedMedium *media;
} PathedSceneDesc;

typedef struct PathedScene PathedScene;   /* opaque; owns all device memory     */

/* Counters filled by the instrumented ("stats") variants of the kernels and by the
 * timed renders; see DESIGN.md "Algorithmic bytes". */
typedef struct PathedStats {
    uint64_t camera_samples;       /* paths started                              */
    uint64_t closest_rays;         /* closest-hit queries traced                 */
    uint64_t shadow_rays;          /* any-hit queries traced                     */
    uint64_t nodes_visited;        /* child boxes tested (32 B each: a 128-B node holds four) — stats mode */
    uint64_t tris_tested;          /* triangles tested (48 B each)  — stats mode */
    uint64_t dropped_samples;      /* non-finite samples dropped                 */
    uint64_t iterations;           /* wavefront iterations launched              */
    double   trace_ms;             /* HIP-event time inside the trace kernel     */
    double   shade_ms;             /* HIP-event time inside the shade kernel     */
    uint64_t trace_launches;
    uint64_t bvh_nodes;            /* 4-wide inner nodes (128 B each)            */
    uint64_t bvh_bytes;            /* nodes + leaf triangles resident in HBM     */
    uint32_t bvh_max_depth;
    uint32_t scene_in_lds;         /* 0 BVH in HBM, 1 BVH staged in LDS, 2 tiny scene: all triangles tested (scalar loads) */
    uint64_t max_boxes_per_ray;    /* most child boxes a single closest-hit ray tested — stats mode */
    uint64_t parked_rays;          /* rays a trace launch handed on to the next one unfinished — stats mode */
    double   bvh_build_ms;         /* scene_create's BVH build: host wall time (SAH) or HIP-event time (LBVH) */
    uint32_t bvh_builder;          /* PATHED_BVH_* the scene was built with                 */
    uint32_t trace_launches_all;   /* trace launches since reset_stats, timed or not (trace_launches counts the timed ones) */
    uint32_t path_kernel;          /* 1 wavefront with the per-slot shade kernel, 2 wavefront with the staged shade kernel,
                                      3 fused path kernel (tiny scenes: one persistent launch per pass, timed as trace_ms),
                                      4 volume path kernel (PATHED_INTEGRATOR_VOLUME_PATH_TRACER),
                                      5 wavefront with the split shade stage (k_vertex + k_regen over the trace kernel's lists),
                                      6 wave path kernel (BVH scenes, the last render call),
                                      7 hybrid path kernel (scenes of 65 .. 4096 triangles, the last render call) */
    uint32_t reserved0;
    uint64_t local_closest_rays;   /* closest-hit queries the shade kernel resolved itself (rays that cannot meet anything but the
                                      scene's few large triangles, PathedSceneOptions.local_rays) -- stats mode; NOT in closest_rays */
    uint64_t local_shadow_rays;    /* ... and any-hit queries -- stats mode; NOT in shadow_rays */
} PathedStats;

/* ---- life cycle ---------------------------------------------------------- */

/* Select the device this process renders on (one process per GPU).
 * Replaces the reference's rtcNewDevice / rtcNewScene (app/main.cpp:46-52). */
int pathed_hip_init(int device_id);

/* Which builder stands in for rtcCommitScene (reference src/scene.cpp:39): PathedSceneOptions.bvh_builder:
 *   PATHED_BVH_SAH_HOST     binned-SAH tree built on the host cores (default: cheapest to traverse)
 *   PATHED_BVH_LBVH_DEVICE  Morton-code linear BVH built on the GPU in milliseconds;
 *                           same node format, so hits and images are bit-identical, traversal costs more.
 *   PATHED_BVH_PLOC_DEVICE  bottom-up clustering along the Morton order on the GPU (PLOC): a few times
 *                           the LBVH's build time, tree quality close to the SAH build's.
 * Meshes of <= 64 triangles always take the host path (they are not traversed at all). */
#define PATHED_BVH_SAH_HOST    0
#define PATHED_BVH_LBVH_DEVICE 1
#define PATHED_BVH_PLOC_DEVICE 2

/* Flatten + upload once: leaf-ordered 48-B triangles, flattened 4-wide BVH (128-B nodes),
 * spheres, material table, light table, env map + CDFs, camera.
 * Replaces rtcCommitScene (reference src/scene.cpp:39) and the light list
 * construction (src/scene_parser.cpp:173-190). */
int pathed_hip_scene_create(const PathedSceneDesc *desc, PathedScene **out);
void pathed_hip_scene_destroy(PathedScene *scene);

/* Refit: new vertex positions -- and optionally normals -- over an UNCHANGED topology, what
 * rtcCommitScene (reference src/scene.cpp:39) repeats for a deforming mesh.  The tree keeps its shape: leaf triangle
 * records, shading records and every node's child boxes are recomputed bottom-up on the device (milliseconds for millions
 * of triangles), boxes padded as the builders pad them.  positions: 3 * n_vertices floats, normals: the same or NULL (keep),
 * host memory; n_vertices must be the scene's.  device_ms (may be NULL): HIP-event time of the refit kernels, uploads excluded.
 * Needs PathedSceneOptions.refittable at creation; PATHED_E_UNSUPPORTED for scenes of <= 64 triangles (recreate those). */
int pathed_hip_scene_refit(PathedScene *scene, const float *positions, const float *normals, uint32_t n_vertices, float *device_ms);

/* Per-scene options.  Zero-initialise, set struct_size = sizeof(PathedSceneOptions), then set only
 * what you need: every field's "automatic" value is 0 except where stated, so a zeroed struct
 * means "all defaults" -- EXCEPT `device`, where 0 is device 0; use PATHED_DEVICE_CURRENT (-1) for
 * the device of pathed_hip_init.  The device is part of the scene: every call that takes the
 * PathedScene selects it first (hipSetDevice is per host thread), so a host may drive several
 * scenes on several GPUs from one thread per GPU (reference app/main.cpp:93-98 runs the
 * integrator on its own std::thread).  This struct is the ONLY way to tune the product library: it
 * reads no PATHED_* environment variable that could change a kernel, a slot count or a builder
 * (the experiments build, `make experiments`, still honours them for A/B scripts; two debug prints,
 * PATHED_DEBUG_ALLOC and PATHED_DEBUG_STATS, exist in both and change no result). */
#define PATHED_DEVICE_CURRENT (-1)
typedef struct PathedSceneOptions {
    uint32_t struct_size;       /* sizeof(PathedSceneOptions)                                  */
    int32_t device;             /* HIP device id, or PATHED_DEVICE_CURRENT                     */
    int32_t bvh_builder;        /* PATHED_BVH_* + 1 (0 = PATHED_BVH_SAH_HOST)                  */
    int32_t stack_rows;         /* LDS rows of the traversal stack: 8, 16 or 22 (0 = by tree depth); deeper entries spill to HBM */
    int32_t pools;              /* independent slot pools, 1..4 (0 = 2)                        */
    int32_t suspend_lanes;      /* park a trace wave's tail below this many rays, 1..64; -1 = never (0 = 32) */
    int32_t suspend_patience;   /* ... after this many steps without a new card, >= 1; -1 = none (0 = 24)     */
    int32_t park_min_cards;     /* ... while the pool has this many cards per wave, >= 1; -1 = always (0 = 1)  */
    int32_t max_slots;          /* path slots (0 = 1 Mi for all-triangles scenes; BVH scenes 8 Mi, fewer for short calls) */
    int32_t intersector;        /* 0 automatic, 1 always walk the BVH (never the all-triangles kernel)         */
    int32_t trace_blocks_per_cu;/* persistent trace blocks per CU (0 = automatic)              */
    int32_t shade_kernel;       /* 0 automatic; 1 wavefront, k_shade (one lane per slot); 2 wavefront, k_shade_staged (dense,
                                   state-sorted stages per block); 3 k_path_small (fused: whole paths in registers; scenes of
                                   <= 64 triangles only, their default); 4 wavefront, k_vertex + k_regen over the hit / miss
                                   lists the trace kernel writes (BVH scenes only); 5 k_path_wave (BVH scenes of <= 96 materials:
                                   paths in registers, the wave's rays shared through LDS, no path state in HBM) for every call.
                                   6 k_path_hybrid (sphere-free scenes of 65 .. 4096 triangles, <= 96 materials: the <= 64 largest
                                   triangles through the all-items intersector, the rest through a tree of their own, paths in
                                   registers).
                                   0 on a BVH scene: k_path_hybrid where it applies; else k_path_wave for calls of fewer than 48 Mi
                                   camera samples, whose rate hardly depends on the call's size, the wavefront (1) for longer ones
                                   (trees of up to 36 KB, which the wavefront would copy into LDS: k_path_wave at every length);
                                   images are identical */
    int32_t stage_slots;        /* slots per block of the staged kernel: 512 or 1024 (0 = automatic)           */
    int32_t unit_order;         /* order work units are handed out in (scheduling only, results identical):
                                 * 0 automatic = 1; 1 chunk stripes, rows; 2 chunk stripes, 32 x 8 tiles; 3 pixel tiles */
    int32_t build_threads;      /* host threads of the SAH builder (0 = all cores; the tree does not depend on it) */
    int32_t generic_kernels;    /* 1: never pick a scene-specialised kernel instantiation (k_shade<.., ENV_ONLY> for scenes whose
                                   one light is the environment and whose materials do not emit): A/B runs and tests */
    int32_t node_format;        /* the tree the trace kernel walks: 0 automatic; 1 128-byte nodes (four float boxes); 2 compressed
                                   64-byte nodes (the boxes on an 8-bit grid over their union, rounded outward: same hits);
                                   3 compressed 8-wide nodes (128 bytes, up to eight children: grandchildren pulled up) --
                                   2 and 3: sphere-free scenes whose tree stays in HBM, per-slot pipeline; an error elsewhere */
    int32_t small_phase1;       /* the fused kernel's conservative all-triangles pass (phase 1 of rtcIntersect1 / rtcOccluded1,
                                   reference src/scene.cpp:113,374): 0 automatic; 1 on the VALU (packed Moeller-Trumbore);
                                   2 on the matrix pipe (v_mfma_f32_32x32x2_f32 over Pluecker rows, mfma_candidates.h) --
                                   phase 2 decides either way: hits and images are bit-identical */
    int32_t refittable;         /* 1: keep the triangle soup (positions, normals, uvs, indices: 44 bytes per triangle or so) on the
                                   device so that pathed_hip_scene_refit can move the vertices later; BVH scenes only */
    int32_t wave_max_ksamples;  /* shade_kernel 0 on a BVH scene: render calls of fewer than this many x 1024 camera samples run
                                   k_path_wave, longer ones the wavefront (0 = 49 152, i.e. 48 Mi samples) */
    int32_t wave_stragglers;    /* k_path_wave / k_path_hybrid: a traversal burst ends once the wave's list is dealt and fewer rays
                                   than this are still in flight, 1..64; -1 = every ray is finished first (0 = 24 / 16).  Scheduling only */
    int32_t wave_refill;        /* k_path_wave / k_path_hybrid: idle lanes draw from the wave's ray list once fewer than this many are
                                   busy, 1..64 (0 = 48 / 40).  Scheduling only */
    int32_t chunks_per_pass;    /* work units per pixel of one internal pass, 1..4096 (0 = 256, fewer at resolutions whose partial
                                   sums would not fit): longer passes amortise a pass's ramp-up and drain, at 16 bytes per unit */
    int32_t local_rays;         /* the wavefront's local rays: a sphere-free BVH scene with at most 8 LARGE triangles (each >= 1 / 256 of
                                   its surface: a floor, a backdrop) lets the shade kernel resolve the rays that cannot meet the
                                   bounds of everything else -- they skip the trace kernel; same hits.  0 automatic (on), 1 off */
    int32_t shade_chain;        /* environment-lit scenes: the shade kernel that ends a sample starts the next one at once and, when its
                                   camera ray is a local ray that hits a large triangle, shades its first vertex in the same launch
                                   (k_shade_env).  0 automatic (on), 1 off.  Scheduling only */
    int32_t shade_launches;     /* ... shade launches per trace launch on such a scene, 1..16 (0 = 1: more were measured and lose): in the further ones the slots
                                   whose rays were all local advance another vertex, the others wait for the trace kernel, which
                                   then finds the tree-walking rays of several vertices in one launch.  Scheduling only */
    int32_t hybrid_batch;       /* k_path_hybrid: a wave walks its tree part once this many of its rays wait or are in flight,
                                   1..128 (1 = in every iteration; 0 = 24).  Scheduling only */
    int32_t hybrid_ready;       /* ... or once fewer of its paths than this can go on without a result, 1..64; -1 = only when
                                   none can (0 = 28).  Scheduling only */
} PathedSceneOptions;
int pathed_hip_scene_create_ex(const PathedSceneDesc *desc, const PathedSceneOptions *options, PathedScene **out);
int pathed_hip_scene_device(const PathedScene *scene);   /* the HIP device the scene lives on, or a negative error */
/* Another view of the same scene: replaces the camera (reference Camera, src/camera.cpp:13-30) without rebuilding or
 * re-uploading anything; the resolution must stay (the caller's radiance sums are per pixel). */
int pathed_hip_scene_set_camera(PathedScene *scene, const PathedCamera *camera);

/* ---- the hot path -------------------------------------------------------- */

/* Render camera samples [spp_begin, spp_begin+spp_count) of every pixel and ADD
 * the per-pixel radiance sums into accum_rgb_sum (host memory, 3*W*H floats,
 * index 3*(row*W+col)+c, row 0 = bottom scanline) — exactly what
 * SampleIntegrator::sampleImage does to radianceLookup, spp_count times
 * (reference src/sample_integrator.cpp:61-63, src/integrator.cpp:42-51).
 * Bounce window as BounceController (reference src/bounce_controller.cpp:14-25);
 * last_bounce = -1 means unbounded, as in the reference (paths end on a miss or when the
 * throughput becomes exactly black). */
int pathed_hip_render(PathedScene *scene, uint64_t seed,
                      uint32_t spp_begin, uint32_t spp_count,
                      int start_bounce, int last_bounce,
                      float *accum_rgb_sum);

/* Same, but the sum buffer is DEVICE memory owned by the caller (e.g. a torch
 * tensor) and the work is enqueued on `stream` (a hipStream_t, NULL = default
 * stream).  The per-pixel sum CONTINUES from the buffer's current contents, so successive
 * calls whose lengths are multiples of the samples-per-unit setting are bit-identical to
 * one long call.  The call returns when the device work has completed (`blocking` is
 * reserved; the iteration loop polls a device counter). */
int pathed_hip_render_device(PathedScene *scene, uint64_t seed,
                             uint32_t spp_begin, uint32_t spp_count,
                             int start_bounce, int last_bounce,
                             float *d_accum_rgb_sum, void *stream, int blocking);

/* Which of the reference's SampleIntegrator subclasses the render calls run (job.json "integrator", src/job.cpp:65-97):
 *   PATHED_INTEGRATOR_PATH_TRACER         PathTracer::L (src/path_tracer.cpp:19-216), the default
 *   PATHED_INTEGRATOR_VOLUME_PATH_TRACER  VolumePathTracer::L (src/volume_path_tracer.cpp:14-131) with
 *                                         DirectLightingHelper::Ld (src/direct_lighting_helper.cpp:37-187): participating
 *                                         media behind PATHED_MAT_PASSTHROUGH containers, single scattering per segment.
 * Scenes that contain PATHED_MAT_PASSTHROUGH materials render with the volume integrator only. */
#define PATHED_INTEGRATOR_PATH_TRACER 0
#define PATHED_INTEGRATOR_VOLUME_PATH_TRACER 1
int pathed_hip_set_integrator(PathedScene *scene, int integrator);

/* Summation granularity.  A pixel's samples are summed in sample order in groups of
 * `samples` (a work unit); the group sums are then added to the pixel in group order.
 * The result is deterministic for a given value.  The default, samples == 1, reproduces the
 * reference's order exactly (radianceLookup += one sample per wave, src/integrator.cpp:42-51);
 * it is also the finest grain of the work queue (a render call drains its last UNITS).
 * Larger groups write fewer partial sums (16 bytes per unit).  Range [1, 128]. */
int pathed_hip_set_samples_per_unit(PathedScene *scene, int samples);

/* Test hook onto the intersector that stands in for Embree.
 * rays: n * 8 floats (ox,oy,oz,tnear, dx,dy,dz,tfar), host memory.
 * any_hit == 0: closest hit (rtcIntersect1, reference src/scene.cpp:91-117):
 *               hits = n * 4 x 32-bit (t, u, v as float; prim as int32, -1 = miss;
 *               prim < n_triangles: triangle, else sphere n_triangles+i).
 * any_hit != 0: occlusion (rtcOccluded1, reference src/scene.cpp:355-381):
 *               hits = n * int32 (1 = occluded). */
int pathed_hip_trace(PathedScene *scene, const float *rays, size_t n,
                     int any_hit, void *hits);

/* Test hook onto phase 1 of the all-triangles intersector (scenes of <= 64 triangles; the stand-in for
 * rtcIntersect1 / rtcOccluded1 on such scenes, reference src/scene.cpp:113,374).
 * rays: n * 10 floats (origin.xyz, continuation direction.xyz, shadow direction.xyz, shadow tfar), host memory.
 * out:  n * 8 x uint64, bit p = ORIGINAL primitive id p (triangle ids of such scenes are < 64):
 *       candidates of the pair-of-triangles phase 1 (continuation, shadow), of the matrix-pipe phase 1 (continuation,
 *       shadow; zero in the product library), triangles phase 2 accepts (continuation: t in (1e-3, 1e5]; shadow: t in
 *       (1e-3, tfar]), candidates of the ITEM phase 1 the fused kernel runs -- parallelograms for the triangle pairs that
 *       form one (pathed_amd/csrc/small_items.h) -- (continuation, shadow).
 * Phase 1 is correct iff accepted is a subset of candidates for every ray. */
int pathed_hip_debug_small_candidates(PathedScene *scene, const float *rays, size_t n, uint64_t *out);

/* 1 in libpathed_hip_experiments.so (`make experiments`), 0 in the product library.  The experiments build adds the
 * kernel organisations that were measured and rejected: shade_kernel 2 (staged) and 4 (split),
 * node_format 2 / 3 (compressed nodes), small_phase1 2 (matrix pipe) and pathed_hip_measure_valu_clocks; the product
 * library answers PATHED_E_UNSUPPORTED to each of them.  (pathed_hip_debug_small_candidates runs in both: the product reports
 * zero for the matrix-pipe form's two words.)  The experiments build also honours the PATHED_* tuning variables. */
int pathed_hip_has_experiments(void);

/* Bit 0: the counting variants of the trace kernel (nodes_visited / tris_tested); off by default, the
 * timed path never counts.  Bit 1: HIP-event pairs around every trace and shade launch (trace_ms,
 * shade_ms, trace_launches).  Bit 2: ... around every 8th launch only: the event pairs keep a pool's
 * kernels from running back to back and cost ~6 % of the rate when every launch is timed. */
int pathed_hip_set_stats_mode(PathedScene *scene, int enabled);
int pathed_hip_get_stats(PathedScene *scene, PathedStats *out);
int pathed_hip_reset_stats(PathedScene *scene);

/* Export the flattened BVH so a checker can walk the SAME tree.
 * nodes: 32 floats per 4-wide node, the four children in the components of
 *   (lo.x[4]) (lo.y[4]) (lo.z[4]) (hi.x[4]) (hi.y[4]) (hi.z[4]) (ref[4]) (unused);
 *   ref is an int32: >= 0 inner node index, <= -2 leaf with -ref - 1 =
 *   (first leaf triangle << 3) | triangle count, INT32_MIN empty slot.
 * tris: 12 floats per leaf triangle (v0.xyz, prim) (e1.xyz, -) (e2.xyz, -).
 * Pass NULL to query sizes. */
int pathed_hip_scene_export_bvh(PathedScene *scene,
                                float *nodes, size_t *n_nodes,
                                float *tris, size_t *n_tris);
/* The compressed form of the same nodes, same indices.
 * node_format 2, 16 words per node, same refs:
 *   (origin.xyz, scale.x) (scale.y, scale.z, qlo.x, qlo.y) (qlo.z, qhi.x, qhi.y, qhi.z) (ref[4])
 * node_format 3, 32 words per node, up to eight children (children of children pulled up; nodes no ref reaches are dead):
 *   (origin.xyz, scale.x) (scale.y, scale.z, qlo.x[2]) (qlo.y[2], qlo.z[2]) (qhi.x[2], qhi.y[2]) (qhi.z[2], -, -) (ref[0..3]) (ref[4..7]) (-)
 * origin / scale are floats, a q word holds four children's 8-bit grid indices (child c in bits 8(c mod 4) .. of word c / 4):
 * child c spans [origin + qlo * scale, origin + qhi * scale] per axis and CONTAINS its float box.
 * *n_nodes comes back 0 for a scene that does not carry a compressed form.  Pass NULL to query the sizes. */
int pathed_hip_scene_export_compressed_nodes(PathedScene *scene, uint32_t *nodes, size_t *n_nodes, size_t *words_per_node);

/* Measurement aid: what a plain streaming kernel reaches on THIS device, as a
 * second denominator beside the 8 TB/s HBM3E spec figure.  Allocates two probe buffers of `bytes`
 * (use >= 1 GiB: beyond the 256 MB Infinity Cache), times `repeats` passes of a 16-byte-per-lane
 * read kernel and of a copy kernel with HIP events.  read_gbs = bytes read / s; copy_gbs counts
 * bytes read + bytes written. */
int pathed_hip_measure_bandwidth(size_t bytes, int repeats, double *read_gbs, double *copy_gbs);

/* Measurement aid: the VALU issue rate THIS device sustains, in wave-instructions per second, as the
 * denominator of the "VALU issue" bound quoted for the scenes whose ray queries never leave the
 * registers (<= 64 triangles).  Runs `waves_per_simd` waves (1..8) on every SIMD of the chip, each
 * issuing `instructions_per_wave` INDEPENDENT v_fma_f32 (eight accumulator chains per lane, so no
 * wave ever waits on its own result), timed with HIP events over `repeats` launches.
 * fma_rate = v_fma_f32 wave-instructions / s (one VGPR source, scalar multiplier and addend); mixed_rate = the
 * same with one v_rcp_f32 / v_sqrt_f32 pair per six v_fma_f32 (what a path tracer's normalisations issue). */
int pathed_hip_measure_valu(int waves_per_simd, int repeats, double *fma_rate, double *mixed_rate);
/* The same probe over n_modes (<= 5) instruction mixes, rates[mode] in wave-instructions / s:
 * 0 v_fma_f32 with one VGPR source (scalar multiplier / addend: no register-bank conflicts), 1 six of those + v_rcp_f32 +
 * v_sqrt_f32, 2 v_fma_f32 with three VGPR sources, 3 v_pk_fma_f32 (two FMAs per lane), 4 v_mul_lo_u32. */
int pathed_hip_measure_valu_modes(int waves_per_simd, int repeats, double *rates, int n_modes);

/* The probe with its own clocks: `chains` (8 or 16) independent v_fma_f32 chains per lane on three VGPR operands; every
 * wave reads the shader clock (s_memtime) and the constant-rate wall clock (s_memrealtime) around its loop, so the issue
 * rate comes out in CYCLES PER INSTRUCTION at the frequency the chip actually ran at under this load, beside the rate
 * in instructions per second from HIP events.  (The guide's constants table has v_fma_f32 at 2 cycles per wave64
 * instruction: 1 228.8 G wave-instr/s at 2.4 GHz on 1 024 SIMDs.) */
typedef struct PathedValuClocks {
    double rate;                          /* wave-instructions / s, HIP events over `repeats` launches              */
    double shader_clock_mhz;              /* s_memtime ticks / s_memrealtime ticks x wall_clock_mhz                   */
    double wall_clock_mhz;                /* hipDeviceAttributeWallClockRate                                         */
    double peak_clock_mhz;                /* hipDeviceAttributeClockRate (what the device advertises)                */
    double wave_ticks_per_instruction;    /* shader-clock ticks one wave needs per instruction it issues             */
    double cycles_per_instruction;        /* ... divided by the waves that share its SIMD: the SIMD's issue interval */
    double cycles_per_instruction_events; /* SIMDs x shader clock / rate: the same from the event time               */
} PathedValuClocks;
int pathed_hip_measure_valu_clocks(int waves_per_simd, int chains, int repeats, PathedValuClocks *out);

/* ---- multi-GPU fan-in ------------------------------------------------------ */
/* The path's one exchange step: per-GPU radiance sums -> one buffer (the reference's
 * waves are additive, src/integrator.cpp:42-51).  A host that drives several scenes in one process
 * (one worker thread per GPU) uses these two instead of RCCL:
 *   pathed_hip_accum_copy_peer  copies `count` floats from src (on src_scene's device) into dst (on
 *                               dst_scene's device) over xGMI (hipMemcpyPeer); blocking.
 *   pathed_hip_accum_add        dst[i] += src[i] on dst_scene's device; blocking. */
int pathed_hip_accum_copy_peer(PathedScene *dst_scene, float *dst, PathedScene *src_scene, const float *src, size_t count);
int pathed_hip_accum_add(PathedScene *dst_scene, float *dst, const float *src, size_t count);
/* Device memory for such buffers, on the scene's device (hipMalloc / hipFree / hipMemcpy wrappers so a
 * plain C++ host needs no HIP headers). */
int pathed_hip_accum_alloc(PathedScene *scene, size_t count, float **out);   /* zero-filled */
int pathed_hip_accum_free(PathedScene *scene, float *buffer);
int pathed_hip_accum_download(PathedScene *scene, const float *buffer, size_t count, float *host);
int pathed_hip_accum_upload(PathedScene *scene, float *buffer, size_t count, const float *host);

/* The same exchange step as ONE collective: ncclReduce(sum, fp32, count, root = replica 0) over RCCL / xGMI
 * (the reference adds its waves in src/integrator.cpp:42-51).  A communicator spans the
 * devices of one process's replicas (ncclCommInitAll); librccl is loaded when the first communicator is made,
 * so a single-GPU host never pays for it.  Devices must be distinct (RCCL refuses two ranks on one device:
 * replicas that share a GPU use the peer-copy calls above); n_devices = 1 is a valid communicator.
 *   pathed_hip_comm_reduce  send[r] holds `count` floats on device_ids[r]; their sum lands in recv_root on
 *                           device_ids[0] (may equal send[0]: in place); blocking. */
typedef struct PathedComm PathedComm;
int pathed_hip_comm_init(int n_devices, const int *device_ids, PathedComm **out);
int pathed_hip_comm_reduce(PathedComm *comm, const float *const *send, float *recv_root, size_t count);
void pathed_hip_comm_destroy(PathedComm *comm);

const char *pathed_hip_last_error(void);
const char *pathed_hip_version(void);   /* "pathed_hip <version> (gfx950, abi <PATHED_ABI_VERSION>)" */

#ifdef __cplusplus
}
#endif
#endif /* PATHED_HIP_H */
