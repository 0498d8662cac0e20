// libpathed_hip.so — implementation of include/pathed_hip.h.
// Host side: scene flattening (BVH, shading records, light table, env CDFs), upload,
// the wavefront iteration loop, HIP-event timing and statistics.
#include "pathed_hip.h"

#include "bvh_build.h"
#include "kernels.h"
#if PATHED_EXPERIMENTS
#include "kernels_experiments.h"
#endif
#include "lbvh.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

using namespace pathed;

namespace {

thread_local std::string g_error;
int g_device = -1;

int fail(int code, const std::string &message)
{
    g_error = message;
    return code;
}

// Tuning switches read from the environment exist in the EXPERIMENTS build only (`make experiments`): the product library
// takes every setting from PathedSceneOptions, so a stray PATHED_* variable on a bench box cannot change a kernel, a slot
// count or a builder behind the caller's back (bench.py also refuses to run with one set).  Two debug PRINTS stay in both
// builds: PATHED_DEBUG_ALLOC and PATHED_DEBUG_STATS (they change no result and no timing path).
inline const char *tuningEnv(const char *name)
{
#if PATHED_EXPERIMENTS
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t status_ = (expr);                                                       \
        if (status_ != hipSuccess) {                                                       \
            return fail(PATHED_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(status_)); \
        }                                                                                  \
    } while (0)

// hipSetDevice is per host thread: every entry point that takes a scene selects the scene's device first
#define SELECT_DEVICE(scene) HIP_TRY(hipSetDevice((scene)->deviceId))

template <typename T>
struct DeviceBuffer {
    T *ptr = nullptr;
    size_t count = 0;

    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer &) = delete;              // owns its allocation: a member added to PathedScene
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;   // is freed with it, whether or not ~PathedScene names it
    ~DeviceBuffer() { release(); }

    hipError_t allocate(size_t n)
    {
        const bool report = getenv("PATHED_DEBUG_ALLOC") != nullptr && n * sizeof(T) >= ((size_t)1 << 28);
        const auto t0 = std::chrono::steady_clock::now();
        release();
        const auto t1 = std::chrono::steady_clock::now();
        count = n;
        if (n == 0) { return hipSuccess; }
        const hipError_t status = hipMalloc((void **)&ptr, n * sizeof(T));
        if (report) {
            const auto t2 = std::chrono::steady_clock::now();
            fprintf(stderr, "[pathed] device buffer of %.2f GB: hipFree of the old one %.3f s, hipMalloc %.3f s\n", (double)(n * sizeof(T)) / 1e9,
                    std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count());
        }
        return status;
    }

    hipError_t upload(const std::vector<T> &host)
    {
        hipError_t status = allocate(host.size());
        if (status != hipSuccess || host.empty()) { return status; }
        return hipMemcpy(ptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice);
    }

    void release()
    {
        if (ptr) { (void)hipFree(ptr); }
        ptr = nullptr;
        count = 0;
    }
};

struct EventRing {
    static const int kPairs = 512;
    hipEvent_t start[kPairs];
    hipEvent_t stop[kPairs];
    bool used[kPairs];
    int next = 0;
    bool created = false;
    double totalMs = 0.0;
    unsigned long long launches = 0;

    hipError_t create()
    {
        if (created) { return hipSuccess; }
        for (int i = 0; i < kPairs; i++) {
            hipError_t status = hipEventCreate(&start[i]);
            if (status != hipSuccess) { return status; }
            status = hipEventCreate(&stop[i]);
            if (status != hipSuccess) { return status; }
            used[i] = false;
        }
        created = true;
        return hipSuccess;
    }

    void harvest(int i)
    {
        if (!used[i]) { return; }
        float ms = 0.f;
        (void)hipEventSynchronize(stop[i]);
        if (hipEventElapsedTime(&ms, start[i], stop[i]) == hipSuccess) {
            totalMs += ms;
            launches++;
        }
        used[i] = false;
    }

    int acquire()
    {
        const int i = next;
        next = (next + 1) % kPairs;
        harvest(i);
        used[i] = true;
        return i;
    }

    void harvestAll()
    {
        if (!created) { return; }
        for (int i = 0; i < kPairs; i++) { harvest(i); }
    }

    void destroy()
    {
        if (!created) { return; }
        for (int i = 0; i < kPairs; i++) {
            (void)hipEventDestroy(start[i]);
            (void)hipEventDestroy(stop[i]);
        }
        created = false;
    }
};

}  // namespace

static const int kMaxPools = 4;
static const int kDefaultShadeLaunches = 1;   // shade launches per trace launch of a scene with local rays (PathedSceneOptions.shade_launches = 0): more were
// measured and lose -- the shade kernel is the slower partner, profiles/r5_ab_local_rays.log
static const int kDefaultSmallPhase1 = 1;   // PathedSceneOptions.small_phase1 = 0: 1 VALU, 2 matrix pipe


struct PathedScene {
    DScene device;
    int width = 0, height = 0;
    int deviceId = 0;             // the HIP device every buffer of this scene lives on
    PathedSceneOptions options;   // as passed to scene_create_ex (zeroed = defaults)

    // host copy kept for export / introspection (a device-built tree is downloaded on demand)
    FlatBvh bvh;
    bool bvhOnHost = true;
    int bvhBuilder = PATHED_BVH_SAH_HOST;
    double bvhBuildMs = 0.0;

    DeviceBuffer<float4> nodes, nodesQ, leafTris, triShade, triCompact, envRgba, texels;
    DeviceBuffer<DSphere> spheres;
    float spherePairs[8][4][2] = {};   // RenderParams.spherePairs: the spheres two by two for the fused kernel's packed pre-test
    DeviceBuffer<DMaterial> materials;
    DeviceBuffer<DLight> lights;
    DeviceBuffer<float> thetaCdf, phiCdf;
    DeviceBuffer<int> phiEmpty, thetaGuide, phiGuide;
    DeviceBuffer<float4> thetaRecords, phiRecords;
    DeviceBuffer<DMedium> media;
    DeviceBuffer<int> primMedium;
    DeviceBuffer<int> volumeOverflow;   // the volume kernel's traversal-stack spill, per thread

    // render state, allocated on first use
    int nSlots = 0;               // capacity of the per-slot state buffers
    bool adaptiveSlots = false;   // BVH scenes without an explicit max_slots: short calls use fewer slots
    int unitOrder = kOrderStripes;   // kernels.h: THE UNIT ORDER
    size_t chunkCapacity = 0;     // float4 entries of chunkBuf
    int samplesPerUnit = 1;       // "chunk": samples a slot sums before it publishes a partial (1 = the reference's summation order)
    int maxSlots = 1 << 20;
    int pools = 2;                // independent slot pools on separate streams (trace || shade); PATHED_POOLS = 1..kMaxPools
    hipStream_t poolStreams[kMaxPools] = { nullptr, nullptr, nullptr, nullptr };
    hipEvent_t poolDone[kMaxPools] = { nullptr, nullptr, nullptr, nullptr };
    hipEvent_t callerReady = nullptr;
    DeviceBuffer<float4> rayO, rayD, hit, mod, thr, res, pend, acc, shO, shD, chunkBuf;
    DeviceBuffer<unsigned int> counters;
    DeviceBuffer<unsigned long long> suspendMask;  // per pool, per trace wave (kernels.h: tail suspension)
    DeviceBuffer<int> suspendData;
    DeviceBuffer<int> stackOverflow;  // per pool, per trace thread, (maxStack - stackRows) rows
    DeviceBuffer<unsigned long long> stats;
    unsigned int *hostRemaining = nullptr;  // pinned

    int integrator = PATHED_INTEGRATOR_PATH_TRACER;   // pathed_hip_set_integrator
    bool hasContainers = false;   // some surface carries the passthrough material: only the volume integrator renders the scene
    bool spheresInTree = false;   // the host builder put the spheres into leaves (else they are tested one by one after the traversal)
    bool fusedPath = false;   // tiny scenes: k_path_small, whole paths in registers, no wavefront buffers
    // BVH scenes: k_path_wave (path_wave.h: paths in registers, no state in HBM) instead of the wavefront kernels.  It renders
    // at a rate that hardly depends on the size of the call, the wavefront needs some 50 M camera samples to fill and drain its
    // 8 Mi slots and is 9-20 % faster beyond (profiles/r4_ab_wave.log): 0 by the size of the call, 1 never, 2 always
    int waveMode = 0;
    bool waveAvailable = false;   // ... and the scene is one it serves
    bool lastCallWave = false;    // what the last render call ran (PathedStats.path_kernel)
    bool lastCallHybrid = false;
    unsigned long long waveMaxSamples = 48ull << 20;   // calls of fewer camera samples than this take it (waveMode 0)
    int waveStragglers = 24;      // its traversal bursts end once fewer rays than this are in flight
    bool waveBlock = false;       // the block's waves share one ray ring (k_path_wave<.., BLOCK>; PATHED_WAVE_BLOCK=1)
    int waveShadeReady = 40;      // ... and a wave shades once this many of its paths have their rays back
    int waveRefill = kRefillThreshold;   // ... and idle lanes draw from the wave's list once fewer than this many are busy
    bool stagedShade = true;  // k_shade_staged (dense, state-sorted stages inside a block) or k_shade (one lane per slot)
    bool lambertianTriangles = false;   // constant-albedo Lambertian surfaces, triangle lights, no spheres, no environment: k_path_small<.., TraitsLambertianTriangles>
    bool lambertianPlasticSpheres = false;   // the Veach scene's set (shading.h)
    bool triangleLit = false;                // any BSDF, constant albedo, triangle lights only, no spheres, no environment
    // [r5] ... narrowed further (shading.h: the ladder): rough BSDFs over one microfacet distribution / Lambertian + glass + mirror
    bool roughBeckmann = false, roughGgx = false, smoothSet = false;
    bool lambertianGlassContainer = false;   // the reference's volume scene's set
    int nodeFormat = 0;                      // what k_trace walks (trace.h): 0 the 128-byte float nodes, 1 nodeQ, 2 node8
    bool envOnly = false;     // the one light is the environment and no material emits: k_shade<.., ENV_ONLY> (kernels.h)
    bool splitShade = false;  // k_vertex + k_regen over the hit / miss lists the trace kernel writes (kernels.h: split shade stage)
    int vertexGrid = 0, regenGrid = 0;   // their persistent grids, blocks
    unsigned int listCap = 0;            // list blocks per shard
    DeviceBuffer<unsigned int> slotLists, deferredLists;
    size_t deferredSlots = 0;            // slots deferredLists was sized for
    int stageRounds = 2;      // staged kernel: a block owns stageRounds x 256 slots

    int stackRows = 8;    // LDS rows of the per-lane traversal stack (8 / 16 / 22)
    int maxStack = 0;     // the tree's bound on stack entries; entries beyond stackRows spill to stackOverflow
    bool sceneInLds = false;
    bool bruteForce = false;      // <= kBruteForceMaxTris triangles: test them all, no BVH walk
    SmallTris smallTris;          // their records, passed to k_trace_small as a kernel argument
    // the fused kernel's phase-1 records (small_items.h): parallelograms first, then the triangles without a partner; phase 2
    // indexes itemTris, the triangle records in the same (item) order
    SmallTris smallItems;
    SmallItemsLayout smallLayout;
    DeviceBuffer<float4> itemTris;
    std::vector<float> itemTrisHost;       // host copy of itemTris (12 floats per triangle)
    std::vector<float> smallExtraPoints;   // sphere bounds: where else a ray may start (the camera is added when the records are built)
    // PathedSceneOptions.refittable: the triangle soup stays on the device for pathed_hip_scene_refit
    bool refittable = false;
    DeviceBuffer<float> soupPositions, soupNormals, soupUvs;
    DeviceBuffer<uint32_t> soupIndices;
    DeviceBuffer<int> soupTriMaterial;
    DeviceBuffer<float4> refitLo, refitHi;          // unpadded bounds per node (refit working memory, allocated on first use)
    DeviceBuffer<unsigned char> refitReady;         // two flag arrays
    size_t soupVertices = 0;
    // [r5] local rays of the wavefront (kernels.h: RenderParams::localTris): the scene's few large triangles and the bounds of the rest
    int localCount = 0;
    float4 localTris[3 * 8];
    float localLo[3] = { 0.f, 0.f, 0.f }, localHi[3] = { 0.f, 0.f, 0.f }, localSphere[4] = { 0.f, 0.f, 0.f, 0.f };
    // [r5] k_path_hybrid (path_hybrid.h): scenes of 65 .. kHybridMaxTris triangles split into a DIRECT set of <= 64 large
    // triangles (all-items intersector) and a TREE part with a 4-wide BVH of its own
    bool hybridAvailable = false;  // the split exists
    bool hybridPath = false;       // ... and render calls take it
    SmallTris hybridItems;         // phase-1 records of the direct set
    SmallItemsLayout hybridLayout;
    std::vector<float> hybridDirectLeaf;   // the direct set as the builder's 12-float records (v0, prim) (e1, -) (e2, -)
    std::vector<float> hybridBounds;       // corners of the scene's bounding box: where else rays start (buildSmallItems' extra points)
    DeviceBuffer<float4> hybridItemTris, hybridNodes, hybridTris;
    int hybridNodeCount = 0, hybridTreeTris = 0, hybridDirectTris = 0, hybridMaxStack = 1;
    float hybridLo[3] = { 0.f, 0.f, 0.f }, hybridHi[3] = { 0.f, 0.f, 0.f }, hybridSphere[4] = { 0.f, 0.f, 0.f, 0.f };
    bool mfmaPhase1 = false;      // k_path_small<.., MFMA>: phase 1 on the matrix pipe (mfma_candidates.h)
    DeviceBuffer<float> mfmaTable;   // its A-side rows
    MfmaFrame mfmaFrame;
    size_t traceLdsBytes = 0;
    int traceGrid = 0;
    int suspendLanes = kSuspendLanes;  // PATHED_SUSPEND_LANES overrides (0 = off)
    int suspendPatience = kSuspendPatience;  // PATHED_SUSPEND_PATIENCE
    int parkMinCards = 1;                    // PATHED_PARK_MIN_CARDS
    int computeUnits = 256;

    bool countMode = false;
    bool timeKernels = false;
    int timeInterval = 1;                 // HIP-event pairs around every n-th launch pair of a pool
    unsigned long long traceLaunchesAll = 0;   // trace launches since reset_stats, timed or not
    EventRing traceEvents, shadeEvents;
    unsigned long long iterations = 0;
    unsigned long long cameraSamples = 0;

    ~PathedScene()
    {
        nodes.release(); nodesQ.release(); leafTris.release(); triShade.release(); triCompact.release(); envRgba.release(); texels.release();
        spheres.release(); materials.release(); lights.release();
        thetaCdf.release(); phiCdf.release(); phiEmpty.release(); thetaGuide.release(); phiGuide.release();
        media.release(); primMedium.release(); volumeOverflow.release();
        thetaRecords.release(); phiRecords.release();
        rayO.release(); rayD.release(); hit.release(); mod.release(); thr.release();
        res.release(); pend.release(); acc.release(); shO.release(); shD.release(); chunkBuf.release();
        counters.release(); stats.release();
        suspendMask.release(); suspendData.release(); stackOverflow.release();
        slotLists.release(); deferredLists.release();
        if (hostRemaining) { (void)hipHostFree(hostRemaining); }
        for (int h = 0; h < kMaxPools; h++) {
            if (poolStreams[h]) { (void)hipStreamDestroy(poolStreams[h]); }
            if (poolDone[h]) { (void)hipEventDestroy(poolDone[h]); }
        }
        if (callerReady) { (void)hipEventDestroy(callerReady); }
        traceEvents.destroy();
        shadeEvents.destroy();
    }
};

namespace {

V3 hv(const float *p) { return v3(p[0], p[1], p[2]); }

// lookAt, reference src/transform.cpp:138-164
void buildCamera(const PathedCamera &desc, DCamera *camera)
{
    const V3 source = hv(desc.origin);
    const V3 target = hv(desc.target);
    const V3 up = hv(desc.up);

    const V3 direction = normalized(source - target);
    const V3 xAxis = normalized(cross(normalized(up), direction));
    const V3 yAxis = cross(direction, xAxis);
    const float sign = desc.flip_handedness ? -1.f : 1.f;

    camera->m[0] = sign * xAxis.x; camera->m[1] = yAxis.x; camera->m[2] = direction.x;
    camera->m[3] = sign * xAxis.y; camera->m[4] = yAxis.y; camera->m[5] = direction.y;
    camera->m[6] = sign * xAxis.z; camera->m[7] = yAxis.z; camera->m[8] = direction.z;
    camera->origin[0] = source.x; camera->origin[1] = source.y; camera->origin[2] = source.z;

    // src/camera.cpp:34-35
    const float zNear = 0.01f;
    camera->filmHeight = 2 * tanf(desc.vertical_fov / 2) * zNear;
    camera->filmWidth = camera->filmHeight * desc.width / desc.height;
    camera->resX = desc.width;
    camera->resY = desc.height;
}

DMaterial buildMaterial(const PathedMaterial &m)
{
    DMaterial out;
    std::memset(&out, 0, sizeof out);
    out.type = m.type;
    out.albedoType = m.albedo_type;
    // OrenNayar ctor, reference src/oren_nayar.cpp:11-18
    const float sigma2 = m.sigma * m.sigma;
    out.orenA = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
    out.orenB = (0.45f * sigma2) / (sigma2 + 0.09f);
    for (int i = 0; i < 3; i++) {
        out.diffuse[i] = m.diffuse[i];
        // only Lambertian carries emission through the reference's parser
        // (src/scene_parser.cpp:574-667: every other ctor passes Color(0))
        out.emit[i] = (m.type == PATHED_MAT_LAMBERTIAN) ? m.emit[i] : 0.f;
        out.checkerOn[i] = m.checker_on[i];
        out.checkerOff[i] = m.checker_off[i];
    }
    out.alpha = m.alpha;
    out.ior = m.ior;
    out.distribution = m.distribution;
    out.checkerResU = m.checker_res[0];
    out.checkerResV = m.checker_res[1];
    return out;
}

bool emits(const DMaterial &m) { return !(m.emit[0] == 0.f && m.emit[1] == 0.f && m.emit[2] == 0.f); }

// Distribution ctor, reference src/distribution.cpp:6-33 (sequential float accumulation)
bool buildCdf(const float *values, size_t size, float *cdf)
{
    float sum = 0.f;
    for (size_t i = 0; i < size; i++) { sum += values[i]; cdf[i] = 0.f; }
    if (sum == 0.f) { return true; }  // empty
    for (size_t i = 0; i < size; i++) {
        cdf[i] = values[i] / sum;
        if (i > 0) { cdf[i] += cdf[i - 1]; }
    }
    cdf[size - 1] = 1.f;
    return false;
}

int validate(const PathedSceneDesc *desc)
{
    if (!desc) { return fail(PATHED_E_INVALID, "scene description is null"); }
    if (desc->abi_version != PATHED_ABI_VERSION) { return fail(PATHED_E_INVALID, "abi_version mismatch"); }
    if (desc->camera.width <= 0 || desc->camera.height <= 0) { return fail(PATHED_E_INVALID, "camera resolution must be positive"); }
    if ((uint64_t)desc->camera.width * (uint64_t)desc->camera.height > (1ull << 28)) { return fail(PATHED_E_INVALID, "image too large"); }
    if (desc->n_triangles && (!desc->positions || !desc->indices || !desc->tri_material || !desc->normals || !desc->uvs)) {
        return fail(PATHED_E_INVALID, "triangle arrays missing");
    }
    if (desc->n_spheres && !desc->spheres) { return fail(PATHED_E_INVALID, "sphere array missing"); }
    if (desc->n_geoms && !desc->geoms) { return fail(PATHED_E_INVALID, "geom array missing"); }
    if (desc->n_materials == 0 || !desc->materials) { return fail(PATHED_E_INVALID, "scene needs at least one material"); }
    for (uint32_t i = 0; i < desc->n_triangles; i++) {
        for (int k = 0; k < 3; k++) {
            if (desc->indices[3 * i + k] >= desc->n_vertices) { return fail(PATHED_E_INVALID, "vertex index out of range"); }
        }
        if (desc->tri_material[i] < 0 || (uint32_t)desc->tri_material[i] >= desc->n_materials) {
            return fail(PATHED_E_INVALID, "triangle material index out of range");
        }
    }
    for (uint32_t i = 0; i < desc->n_spheres; i++) {
        if (desc->spheres[i].material < 0 || (uint32_t)desc->spheres[i].material >= desc->n_materials) {
            return fail(PATHED_E_INVALID, "sphere material index out of range");
        }
    }
    for (uint32_t i = 0; i < desc->n_materials; i++) {
        const int type = desc->materials[i].type;
        if (type < PATHED_MAT_LAMBERTIAN || type > PATHED_MAT_PASSTHROUGH) { return fail(PATHED_E_UNSUPPORTED, "unknown material type"); }
        const int albedo = desc->materials[i].albedo_type;
        if (albedo < PATHED_ALBEDO_CONSTANT || albedo > PATHED_ALBEDO_TEXTURE) { return fail(PATHED_E_UNSUPPORTED, "unknown albedo type"); }
        if (albedo == PATHED_ALBEDO_TEXTURE) {
            const int texture = desc->materials[i].texture;
            if (texture < 0 || (uint32_t)texture >= desc->n_textures || !desc->textures) { return fail(PATHED_E_INVALID, "material texture index out of range"); }
        }
        const int distribution = desc->materials[i].distribution;
        if ((type == PATHED_MAT_MICROFACET || type == PATHED_MAT_PLASTIC)
            && distribution != PATHED_DIST_BECKMANN && distribution != PATHED_DIST_GGX) {
            return fail(PATHED_E_UNSUPPORTED, "unknown microfacet distribution");
        }
    }
    if (desc->n_textures && !desc->textures) { return fail(PATHED_E_INVALID, "texture array missing"); }
    for (uint32_t i = 0; i < desc->n_textures; i++) {
        const PathedTexture &texture = desc->textures[i];
        if (texture.width < 1 || texture.height < 1 || texture.width > 65535 || texture.height > 65535 || !texture.rgb) {
            return fail(PATHED_E_INVALID, "texture size must be 1..65535 and its texels present");
        }
    }
    for (uint32_t i = 0; i < desc->n_geoms; i++) {
        const PathedGeom &geom = desc->geoms[i];
        if (geom.type == PATHED_GEOM_MESH) {
            if (geom.first < 0 || geom.count < 0 || (uint64_t)geom.first + (uint64_t)geom.count > desc->n_triangles) {
                return fail(PATHED_E_INVALID, "mesh geom range out of bounds");
            }
        } else if (geom.type == PATHED_GEOM_SPHERE) {
            if (geom.first < 0 || (uint32_t)geom.first >= desc->n_spheres) { return fail(PATHED_E_INVALID, "sphere geom out of bounds"); }
        } else {
            return fail(PATHED_E_INVALID, "unknown geom type");
        }
    }
    if (desc->env) {
        if (desc->env->width <= 0 || desc->env->height <= 0 || !desc->env->rgba) { return fail(PATHED_E_INVALID, "environment map missing"); }
    }
    if (desc->n_media && !desc->media) { return fail(PATHED_E_INVALID, "media array missing"); }
    for (uint32_t i = 0; i < desc->n_media; i++) {
        // HomogeneousMedium asserts sigma_t.r == g == b (reference src/homogeneous_medium.cpp): distances are sampled with
        // one channel and attenuated with all three, so unequal channels would render a biased image silently
        const float *t = desc->media[i].sigma_t, *sc = desc->media[i].sigma_s;
        if (!(t[0] == t[1] && t[1] == t[2])) { return fail(PATHED_E_INVALID, "homogeneous medium: the three sigma_t channels must be equal"); }
        for (int k = 0; k < 3; k++) {
            if (!(t[k] >= 0.f && t[k] < 3e38f && sc[k] >= 0.f && sc[k] < 3e38f)) { return fail(PATHED_E_INVALID, "homogeneous medium: sigma_t and sigma_s must be finite and not negative"); }
        }
    }
    for (uint32_t i = 0; i < desc->n_geoms; i++) {
        if (desc->geoms[i].medium < -1 || (desc->geoms[i].medium >= 0 && (uint32_t)desc->geoms[i].medium >= desc->n_media)) {
            return fail(PATHED_E_INVALID, "geom medium index out of range");
        }
    }
    return PATHED_OK;
}

// chunks (units per pixel) rendered per internal pass: bounds chunkBuf to 256 float4 per pixel (4.3 GB at 1024^2, 8.5 GB
// at 1080p; further capped at 2^30 units).  Every pass pays the slot pool's ramp-up and drain once -- with one sample
// per unit and the queues in step that is a few milliseconds on a BVH scene (256 against 1 024 units per pixel: -0.3 %
// on Cornell, -1.8 % on the 5.2 M-triangle mesh, -4 % on the teapot with 8 Mi slots, tools/pass_length_sweep.sh) -- but
// hipMalloc of a large buffer is not free here: 17 GB take 0.5 s, 1.0 s when another 17 GB were freed just before
// (PATHED_DEBUG_ALLOC=1 prints it), which would add a quarter to a 2 s job.  PATHED_CHUNKS_PER_PASS overrides for
// experiments and long-lived processes.
const int kMaxChunksPerPass = 256;

int ensureRenderState(PathedScene *scene, int nSlots, size_t chunkEntries)
{
    if (scene->nSlots < nSlots || !scene->rayO.ptr) {   // capacity: a call that wants fewer slots uses a prefix
        scene->nSlots = nSlots;
        const size_t n = (size_t)nSlots;
        HIP_TRY(scene->rayO.allocate(n));
        HIP_TRY(scene->rayD.allocate(n));
        HIP_TRY(scene->hit.allocate(n));
        HIP_TRY(scene->mod.allocate(n));
        HIP_TRY(scene->thr.allocate(n));
        HIP_TRY(scene->res.allocate(n));
        HIP_TRY(scene->pend.allocate(n));
        HIP_TRY(scene->acc.allocate(n));
        HIP_TRY(scene->shO.allocate(n));
        HIP_TRY(scene->shD.allocate(n));
    }
    if (scene->chunkCapacity < chunkEntries) {
        HIP_TRY(scene->chunkBuf.allocate(chunkEntries));
        scene->chunkCapacity = chunkEntries;
    }
    if (scene->splitShade) {
        // per pool and list: every slot once, plus one partly filled block per wave that writes (trace and k_vertex
        // waves); sized for the slot capacity, so a call that uses fewer slots fits as well
        const size_t writerWaves = (size_t)(scene->traceGrid + scene->vertexGrid) * kWavesPerBlock;
        const size_t blocks = (size_t)scene->nSlots / 64 + writerWaves;
        // a block's shard is picked from the shader clock: an eighth of slack keeps a shard from ever running full
        const unsigned int cap = (unsigned int)((blocks + kListShards - 1) / kListShards) * 9u / 8u + 8u;
        if (cap > scene->listCap || !scene->slotLists.ptr) {
            scene->listCap = cap;
            HIP_TRY(scene->slotLists.allocate((size_t)kMaxPools * 2 * (size_t)cap * kListShards * 64));
        }
        // sized and indexed with the slot count itself (cap rounds per 32 blocks: nSlots can grow without cap growing)
        if ((size_t)scene->nSlots > scene->deferredSlots || !scene->deferredLists.ptr) {
            HIP_TRY(scene->deferredLists.allocate((size_t)kMaxPools * 4 * (size_t)scene->nSlots));
            scene->deferredSlots = (size_t)scene->nSlots;
        }
    }
    if (!scene->counters.ptr) { HIP_TRY(scene->counters.allocate(kMaxPools * kCtrCount)); }
    if (!scene->suspendMask.ptr && !scene->bruteForce) {
        const size_t waves = (size_t)scene->traceGrid * kWavesPerBlock;
        HIP_TRY(scene->suspendMask.allocate((size_t)scene->pools * waves));
        HIP_TRY(scene->suspendData.allocate((size_t)scene->pools * waves * (size_t)(kSaveWords + scene->maxStack) * 64));
        const size_t overflowRows = (size_t)(scene->maxStack > scene->stackRows ? scene->maxStack - scene->stackRows : 0);
        HIP_TRY(scene->stackOverflow.allocate((size_t)scene->pools * (size_t)scene->traceGrid * kBlock * (overflowRows ? overflowRows : 1)));
    }
    if (!scene->stats.ptr) {
        HIP_TRY(scene->stats.allocate(kStatCount));
        HIP_TRY(hipMemset(scene->stats.ptr, 0, kStatCount * sizeof(unsigned long long)));
    }
    if (!scene->hostRemaining) {
        HIP_TRY(hipHostMalloc((void **)&scene->hostRemaining, kMaxPools * 32 * sizeof(unsigned int), hipHostMallocDefault));
    }
    return PATHED_OK;
}

template <int STACK>
void launchTraceStack(PathedScene *scene, const RenderParams &params, hipStream_t stream)
{
    const dim3 grid((unsigned)scene->traceGrid), block(kBlock);
    const size_t lds = scene->traceLdsBytes;
#if PATHED_EXPERIMENTS
    if (scene->splitShade) {
        if (scene->sceneInLds) {
            if (scene->countMode) { hipLaunchKernelGGL((k_trace<STACK, true, true, true>), grid, block, lds, stream, params); }
            else { hipLaunchKernelGGL((k_trace<STACK, true, false, true>), grid, block, lds, stream, params); }
        } else {
            if (scene->countMode) { hipLaunchKernelGGL((k_trace<STACK, false, true, true>), grid, block, lds, stream, params); }
            else { hipLaunchKernelGGL((k_trace<STACK, false, false, true>), grid, block, lds, stream, params); }
        }
        return;
    }
    if (scene->nodeFormat == 1) {
        if (scene->countMode) { hipLaunchKernelGGL((k_trace<STACK, false, true, false, false, 1>), grid, block, lds, stream, params); }
        else { hipLaunchKernelGGL((k_trace<STACK, false, false, false, false, 1>), grid, block, lds, stream, params); }
        return;
    }
    if (scene->nodeFormat == 2) {
        if (scene->countMode) { hipLaunchKernelGGL((k_trace<STACK, false, true, false, false, 2>), grid, block, lds, stream, params); }
        else { hipLaunchKernelGGL((k_trace<STACK, false, false, false, false, 2>), grid, block, lds, stream, params); }
        return;
    }
#endif
    if (!scene->sceneInLds && scene->device.nSpheres == 0 && !scene->options.generic_kernels && !tuningEnv("PATHED_NO_SCENE_TRAITS")) {
        if (scene->countMode) { hipLaunchKernelGGL((k_trace<STACK, false, true, false, false>), grid, block, lds, stream, params); }
        else { hipLaunchKernelGGL((k_trace<STACK, false, false, false, false>), grid, block, lds, stream, params); }
        return;
    }
    if (scene->sceneInLds) {
        if (scene->countMode) { hipLaunchKernelGGL((k_trace<STACK, true, true, false>), grid, block, lds, stream, params); }
        else { hipLaunchKernelGGL((k_trace<STACK, true, false, false>), grid, block, lds, stream, params); }
    } else {
        if (scene->countMode) { hipLaunchKernelGGL((k_trace<STACK, false, true, false>), grid, block, lds, stream, params); }
        else { hipLaunchKernelGGL((k_trace<STACK, false, false, false>), grid, block, lds, stream, params); }
    }
}

void launchTrace(PathedScene *scene, const RenderParams &params, hipStream_t stream)
{
    if (scene->bruteForce) {
        // uniform cost per batch and no per-block setup: one 64-ray batch per wave, the hardware
        // dispatcher balances (a persistent grid quantises 3.3 batches per wave to 4 rounds)
        const dim3 grid((unsigned)(2 * params.nSlots / kBlock)), block(kBlock);
        if (scene->countMode) { hipLaunchKernelGGL((k_trace_small<true>), grid, block, 0, stream, params, scene->smallTris); }
        else { hipLaunchKernelGGL((k_trace_small<false>), grid, block, 0, stream, params, scene->smallTris); }
        return;
    }
    switch (scene->stackRows) {
    case 8: launchTraceStack<8>(scene, params, stream); break;
    case 16: launchTraceStack<16>(scene, params, stream); break;
    default: launchTraceStack<22>(scene, params, stream); break;
    }
}

#if defined(PATHED_EXPERIMENTS) && PATHED_EXPERIMENTS
// [r5, experiments build] What would SORTING the ray queue buy the production trace kernel?  (Round 1's answer came from the
// one-ray-per-thread test kernel.)  With PATHED_SORT_PROBE=<iteration> the wavefront stops before pool 0's trace launch of that
// iteration, copies the pool's rays, and times k_trace -- this very kernel, cards, refill, LDS stack rows, parking off -- over
// (a) the copy in slot order and (b) the same rays permuted so that the ones to be traced come first, ordered by (direction octant,
// 30-bit Morton code of the origin); hits go to a scratch buffer, the render's own state is not touched.  tools/sort_probe.py.
static void sortProbe(PathedScene *scene, const RenderParams &q, hipStream_t stream)
{
    (void)hipDeviceSynchronize();
    const size_t n = (size_t)q.nSlots;
    std::vector<float4> rayO(n), rayD(n);
    if (hipMemcpy(rayO.data(), q.state.rayO, n * sizeof(float4), hipMemcpyDeviceToHost) != hipSuccess
        || hipMemcpy(rayD.data(), q.state.rayD, n * sizeof(float4), hipMemcpyDeviceToHost) != hipSuccess) { return; }
    std::vector<uint32_t> traced, rest;
    float lo[3] = { 3e38f, 3e38f, 3e38f }, hi[3] = { -3e38f, -3e38f, -3e38f };
    for (size_t i = 0; i < n; i++) {
        int flags; std::memcpy(&flags, &rayD[i].w, 4);
        if (flags & (kStDone | kStHold | kStLocal)) { rest.push_back((uint32_t)i); continue; }
        traced.push_back((uint32_t)i);
        const float o[3] = { rayO[i].x, rayO[i].y, rayO[i].z };
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], o[a]); hi[a] = std::max(hi[a], o[a]); }
    }
    auto spread = [](uint32_t v) { uint64_t x = v & 0x3FFu; x = (x | (x << 16)) & 0x30000FFull; x = (x | (x << 8)) & 0x300F00Full; x = (x | (x << 4)) & 0x30C30C3ull; x = (x | (x << 2)) & 0x9249249ull; return x; };
    // orders: 0 slot order; 1 by direction octant only (stable: what eight counters in the shade kernel could produce);
    // 2 by (octant, 24-bit Morton code of the origin, 18-bit Morton code of the direction), stable
    const int kOrders = 3;
    std::vector<float4> orderedO[kOrders], orderedD[kOrders];
    for (int v = 0; v < kOrders; v++) {
        std::vector<std::pair<uint64_t, uint32_t>> keyed(traced.size());
        for (size_t k = 0; k < traced.size(); k++) {
            const uint32_t i = traced[k];
            const float o[3] = { rayO[i].x, rayO[i].y, rayO[i].z }, d[3] = { rayD[i].x, rayD[i].y, rayD[i].z };
            uint32_t cell[3], turn[3];
            for (int a = 0; a < 3; a++) {
                const float t = hi[a] > lo[a] ? (o[a] - lo[a]) / (hi[a] - lo[a]) : 0.f;
                cell[a] = (uint32_t)std::min(255.f, std::max(0.f, t * 255.f));
                turn[a] = (uint32_t)std::min(63.f, std::max(0.f, (d[a] * 0.5f + 0.5f) * 63.f));
            }
            const uint64_t octant = (d[0] < 0.f ? 4u : 0u) | (d[1] < 0.f ? 2u : 0u) | (d[2] < 0.f ? 1u : 0u);
            const uint64_t place = (spread(cell[0]) << 2) | (spread(cell[1]) << 1) | spread(cell[2]);
            const uint64_t heading = (spread(turn[0]) << 2) | (spread(turn[1]) << 1) | spread(turn[2]);
            keyed[k] = { v == 0 ? 0ull : v == 1 ? octant : ((octant << 42) | (place << 18) | heading), i };
        }
        std::stable_sort(keyed.begin(), keyed.end(), [](const std::pair<uint64_t, uint32_t> &a, const std::pair<uint64_t, uint32_t> &b) { return a.first < b.first; });
        orderedO[v].resize(n); orderedD[v].resize(n);
        size_t at = 0;
        for (const auto &entry : keyed) { orderedO[v][at] = rayO[entry.second]; orderedD[v][at] = rayD[entry.second]; at++; }
        for (uint32_t i : rest) { orderedO[v][at] = rayO[i]; orderedD[v][at] = rayD[i]; at++; }
    }
    float4 *dO[kOrders] = { nullptr, nullptr, nullptr }, *dD[kOrders] = { nullptr, nullptr, nullptr }, *dHit = nullptr;
    unsigned int *dCounters = nullptr;
    bool ok = hipMalloc(&dHit, n * sizeof(float4)) == hipSuccess && hipMalloc(&dCounters, kCtrCount * sizeof(unsigned int)) == hipSuccess;
    for (int v = 0; v < kOrders && ok; v++) {
        ok = hipMalloc(&dO[v], n * sizeof(float4)) == hipSuccess && hipMalloc(&dD[v], n * sizeof(float4)) == hipSuccess
            && hipMemcpy(dO[v], orderedO[v].data(), n * sizeof(float4), hipMemcpyHostToDevice) == hipSuccess
            && hipMemcpy(dD[v], orderedD[v].data(), n * sizeof(float4), hipMemcpyHostToDevice) == hipSuccess;
    }
    hipEvent_t start = nullptr, stop = nullptr;
    ok = ok && hipEventCreate(&start) == hipSuccess && hipEventCreate(&stop) == hipSuccess;
    if (ok) {
        fprintf(stderr, "[pathed] sort probe: %zu slots, %zu rays for the trace kernel (%.3f of the slots), shadow list left out\n", n, traced.size(), (double)traced.size() / (double)n);
        for (int repeat = 0; repeat < 3; repeat++) {
            for (int v = -1; v < kOrders; v++) {
                RenderParams probe = q;
                if (v >= 0) { probe.state.rayO = dO[v]; probe.state.rayD = dD[v]; }   // (-1: the pool's own arrays, rays where their slots are)
                probe.state.hit = dHit;
                probe.counters = dCounters;
                probe.suspendLanes = 0;   // no parking: the whole pool in one launch
                (void)hipMemsetAsync(dCounters, 0, kCtrCount * sizeof(unsigned int), stream);
                (void)hipEventRecord(start, stream);
                launchTrace(scene, probe, stream);
                (void)hipEventRecord(stop, stream);
                (void)hipEventSynchronize(stop);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, start, stop);
                fprintf(stderr, "[pathed] sort probe: k_trace over the rays %s: %.1f us (%.2f Grays/s)\n",
                        v < 0 ? "as they lie in the pool" : v == 0 ? "compacted, in slot order" : v == 1 ? "by direction octant" : "by (octant, origin cell, direction cell)", 1e3 * ms, (double)traced.size() / (1e6 * ms));
            }
        }
    }
    if (start) { (void)hipEventDestroy(start); }
    if (stop) { (void)hipEventDestroy(stop); }
    for (int v = 0; v < kOrders; v++) { if (dO[v]) { (void)hipFree(dO[v]); } if (dD[v]) { (void)hipFree(dD[v]); } }
    if (dHit) { (void)hipFree(dHit); }
    if (dCounters) { (void)hipFree(dCounters); }
}
#endif

void launchShade(PathedScene *scene, const RenderParams &params, hipStream_t stream)
{
#if PATHED_EXPERIMENTS
    if (scene->splitShade && !PATHED_EXP_LISTS_ONLY) {
        const bool ldsMaterials = scene->device.nMaterials <= kMaxLdsMaterials;
        const size_t lds = ldsMaterials ? (size_t)scene->device.nMaterials * sizeof(DMaterial) : 0;
        const int slotBlocks = params.nSlots / kBlock;
        const dim3 vertexGrid((unsigned)(scene->vertexGrid < slotBlocks ? scene->vertexGrid : slotBlocks));
        const dim3 regenGrid((unsigned)(scene->regenGrid < slotBlocks ? scene->regenGrid : slotBlocks));
        if (ldsMaterials) { hipLaunchKernelGGL((k_vertex<true>), vertexGrid, dim3(kBlock), lds, stream, params); }
        else { hipLaunchKernelGGL((k_vertex<false>), vertexGrid, dim3(kBlock), lds, stream, params); }
        hipLaunchKernelGGL(k_regen, regenGrid, dim3(kBlock), 0, stream, params);
        return;
    }
    if (scene->stagedShade) {
        const dim3 grid((unsigned)(params.nSlots / (kBlock * scene->stageRounds))), block(kBlock);
        const bool ldsMaterials = scene->device.nMaterials <= kMaxLdsMaterials;
        const size_t lds = ldsMaterials ? (size_t)scene->device.nMaterials * sizeof(DMaterial) : 0;
        if (scene->stageRounds == 4) {
            if (ldsMaterials) { hipLaunchKernelGGL((k_shade_staged<true, 4>), grid, block, lds, stream, params); }
            else { hipLaunchKernelGGL((k_shade_staged<false, 4>), grid, block, lds, stream, params); }
        } else {
            if (ldsMaterials) { hipLaunchKernelGGL((k_shade_staged<true, 2>), grid, block, lds, stream, params); }
            else { hipLaunchKernelGGL((k_shade_staged<false, 2>), grid, block, lds, stream, params); }
        }
        return;
    }
#endif
    const dim3 grid((unsigned)(params.nSlots / kBlock)), block(kBlock);
    if (scene->device.nMaterials <= kMaxLdsMaterials) {
        // (k_shade narrowed further to {Lambertian, plastic, environment} -- 88 VGPRs against 92 -- shortens the shade launches of
        // the 5.2 M-triangle mesh by 13 % and LENGTHENS the other pool's trace launches by 23 %: -4.8 % overall, not adopted,
        // profiles/r3_ab_scene_traits.log)
        // ([r5] again with local rays, when the trace kernel has a fifth of its rays left: {Lambertian, plastic, environment} 2 312
        // against 2 456 Msamples/s on the same mesh, the sphere-free environment set +-0: profiles/r5_ab_local_rays.log)
        // [r5] environment-only scenes: k_shade_env -- the same vertex code, the next sample's camera ray started (and, if it is a
        // local ray that hits a large triangle, its first vertex shaded) in the launch that ends a sample; shade_chain = 1: the
        // plain k_shade<.., ENV_ONLY> (A/B runs, same floats)
        if (scene->envOnly && scene->options.shade_chain != 1) { hipLaunchKernelGGL((k_shade_env<true>), grid, block, 0, stream, params); }
        else if (scene->envOnly) { hipLaunchKernelGGL((k_shade<true, true>), grid, block, 0, stream, params); }
        else { hipLaunchKernelGGL((k_shade<true, false>), grid, block, 0, stream, params); }
    } else {
        if (scene->envOnly && scene->options.shade_chain != 1) { hipLaunchKernelGGL((k_shade_env<false>), grid, block, 0, stream, params); }
        else if (scene->envOnly) { hipLaunchKernelGGL((k_shade<false, true>), grid, block, 0, stream, params); }
        else { hipLaunchKernelGGL((k_shade<false, false>), grid, block, 0, stream, params); }
    }
}

template <int STACK, bool SMALL>
void launchVolumeStack(const RenderParams &params, const SmallTris &smallTris, dim3 grid, size_t lds, bool ldsMaterials, hipStream_t stream)
{
    if (ldsMaterials) { hipLaunchKernelGGL((k_path_volume<true, STACK, SMALL>), grid, dim3(kBlock), lds, stream, params, smallTris); }
    else { hipLaunchKernelGGL((k_path_volume<false, STACK, SMALL>), grid, dim3(kBlock), lds, stream, params, smallTris); }
}

// small: the scene's triangles go through the all-triangles intersector (kernarg pair records), no tree walk
void launchVolume(int stackRows, bool small, bool narrowed, const RenderParams &params, const SmallTris &smallTris, dim3 grid, size_t lds, bool ldsMaterials,
                  hipStream_t stream)
{
    if (small && narrowed && ldsMaterials) {   // the reference's own volume scene kinds (shading.h: TraitsLambertianGlassContainer)
        if (params.smallQuads > 0) { hipLaunchKernelGGL((k_path_volume<true, 8, true, TraitsLambertianGlassContainer, true>), grid, dim3(kBlock), lds, stream, params, smallTris); }
        else { hipLaunchKernelGGL((k_path_volume<true, 8, true, TraitsLambertianGlassContainer>), grid, dim3(kBlock), lds, stream, params, smallTris); }
        return;
    }
    if (small && ldsMaterials && params.smallQuads > 0) {
        hipLaunchKernelGGL((k_path_volume<true, 8, true, TraitsAll, true>), grid, dim3(kBlock), lds, stream, params, smallTris);
        return;
    }
    if (small) { launchVolumeStack<8, true>(params, smallTris, grid, lds, ldsMaterials, stream); return; }
    switch (stackRows) {
    case 8: launchVolumeStack<8, false>(params, smallTris, grid, lds, ldsMaterials, stream); break;
    case 16: launchVolumeStack<16, false>(params, smallTris, grid, lds, ldsMaterials, stream); break;
    default: launchVolumeStack<22, false>(params, smallTris, grid, lds, ldsMaterials, stream); break;
    }
}

// what node_format 0 picks for the scenes the compressed trees serve: 0 float nodes, 1 nodeQ, 2 node8 (DESIGN.md has the measurements)
#ifndef PATHED_DEFAULT_NODE_FORMAT
#define PATHED_DEFAULT_NODE_FORMAT 0
#endif

// PathedSceneOptions.node_format with the experiments' override: 0 automatic, 1 float nodes, 2 compressed, 3 compressed 8-wide
int requestedNodeFormat(const PathedSceneOptions &options)
{
    if (const char *text = tuningEnv("PATHED_NODE_FORMAT")) {
        if (!strcmp(text, "wide")) { return 1; }
        if (!strcmp(text, "compressed")) { return 2; }
        if (!strcmp(text, "compressed8")) { return 3; }
    }
    return options.node_format;
}

void configureTrace(PathedScene *scene)
{
    // a 4-wide node stacks up to three children: 3 entries per level bound the stack.  22 rows
    // (+1 scratch, +8 KB staging = 31 KB per block) keep five blocks per CU possible; the rare
    // deeper entries spill to HBM.
    // (an 8-wide node stacks all of its up to eight hits and pops one back: 7 per level, 8 for a moment)
    const PathedSceneOptions &options = scene->options;
    const int requested = requestedNodeFormat(options);
    const bool mayWalkNode8 = requested == 3 || (requested == 0 && PATHED_DEFAULT_NODE_FORMAT == 2);
    scene->maxStack = mayWalkNode8 ? 7 * scene->bvh.maxDepth + 9 : 3 * scene->bvh.maxDepth + 1;
    scene->stackRows = scene->maxStack <= 8 ? 8 : scene->maxStack <= 16 ? 16 : 22;
    if (options.stack_rows == 8 || options.stack_rows == 16 || options.stack_rows == 22) { scene->stackRows = options.stack_rows; }
    if (const char *override = tuningEnv("PATHED_STACK_ROWS")) {   // experiments: force the HBM spill path
        const int value = atoi(override);
        if (value == 8 || value == 16 || value == 22) { scene->stackRows = value; }
    }

    // per-thread traversal stacks + the waves' ray staging rows (2 float4 per thread)
    const size_t stackBytes = (size_t)(scene->stackRows + 1) * kBlock * sizeof(int) + (size_t)2 * kBlock * kCardRounds * sizeof(float4)
        + (scene->splitShade ? (size_t)kWavesPerBlock * 2 * 128 * sizeof(unsigned int) : 0);
    const size_t sceneBytes = (size_t)scene->device.nNodes * 128 + (size_t)scene->device.nTris * 48;
    // stage the BVH in LDS when it is small enough to leave >= 4 blocks per CU
    scene->sceneInLds = scene->device.nNodes > 0 && (stackBytes + sceneBytes) <= 36 * 1024;
    scene->traceLdsBytes = stackBytes + (scene->sceneInLds ? sceneBytes : 0);

    int blocksPerCu = (int)((160 * 1024) / (scene->traceLdsBytes ? scene->traceLdsBytes : 1));
    if (blocksPerCu > 8) { blocksPerCu = 8; }
    // measured on MI355X (4-wide tree; teapot / 5.2 M-triangle mesh / the same mesh filling the frame, Msamples/s, 8 Mi
    // slots, queues in step): with the other pool's k_shade sharing the CUs TWO blocks per CU are best -- 1 786 / 2 172 /
    // 1 417 against 1 725 / 2 093 / 1 378 with three (the round-1 optimum, when k_shade was the slower partner by less)
    // and 1 411 / 1 829 / 1 106 with one: two 96-VGPR trace waves leave a SIMD room for THREE 104-VGPR k_shade waves.
    // 448 .. 640 blocks are within 2 % (PATHED_TRACE_GRID).  A single pool wants at least four (tools/occupancy_sweep.sh).
    if (!scene->sceneInLds) { blocksPerCu = scene->pools > 1 ? (blocksPerCu < 2 ? blocksPerCu : 2) : (blocksPerCu < 5 ? blocksPerCu : 5); }
    if (blocksPerCu < 1) { blocksPerCu = 1; }
    if (options.trace_blocks_per_cu >= 1 && options.trace_blocks_per_cu <= 16) { blocksPerCu = options.trace_blocks_per_cu; }
    if (const char *override = tuningEnv("PATHED_TRACE_BLOCKS_PER_CU")) {
        const int value = atoi(override);
        if (value >= 1 && value <= 16) { blocksPerCu = value; }
    }
    scene->traceGrid = scene->computeUnits * blocksPerCu;
    if (const char *override = tuningEnv("PATHED_TRACE_GRID")) {   // experiments: the persistent grid in blocks, any number
        const int value = atoi(override);
        if (value >= 1 && value <= scene->computeUnits * 16) { scene->traceGrid = value; }
    }
    if (options.park_min_cards != 0) { scene->parkMinCards = options.park_min_cards < 0 ? 0 : (options.park_min_cards > 1024 ? 1024 : options.park_min_cards); }
    if (options.suspend_patience != 0) { scene->suspendPatience = options.suspend_patience < 0 ? 0 : (options.suspend_patience > 4096 ? 4096 : options.suspend_patience); }
    if (options.suspend_lanes != 0) { scene->suspendLanes = options.suspend_lanes < 0 ? 0 : (options.suspend_lanes > 64 ? 64 : options.suspend_lanes); }
    if (const char *override = tuningEnv("PATHED_PARK_MIN_CARDS")) {
        const int value = atoi(override);
        if (value >= 0 && value <= 1024) { scene->parkMinCards = value; }
    }
    if (const char *override = tuningEnv("PATHED_SUSPEND_PATIENCE")) {
        const int value = atoi(override);
        if (value >= 0 && value <= 4096) { scene->suspendPatience = value; }
    }
    if (const char *override = tuningEnv("PATHED_SUSPEND_LANES")) {
        const int value = atoi(override);
        if (value >= 0 && value <= 64) { scene->suspendLanes = value; }
    }
}

}  // namespace

extern "C" {

const char *pathed_hip_last_error(void) { return g_error.c_str(); }

#define PATHED_STRINGIFY2(x) #x
#define PATHED_STRINGIFY(x) PATHED_STRINGIFY2(x)
const char *pathed_hip_version(void) { return "pathed_hip 0.3.0 (gfx950, abi " PATHED_STRINGIFY(PATHED_ABI_VERSION) ")"; }

int pathed_hip_init(int device_id)
{
    int count = 0;
    hipError_t status = hipGetDeviceCount(&count);
    if (status != hipSuccess || count == 0) {
        return fail(PATHED_E_NO_DEVICE, std::string("no HIP device: ") + hipGetErrorString(status));
    }
    if (device_id < 0 || device_id >= count) { return fail(PATHED_E_INVALID, "device id out of range"); }
    HIP_TRY(hipSetDevice(device_id));
    g_device = device_id;
    return PATHED_OK;
}

int pathed_hip_measure_bandwidth(size_t bytes, int repeats, double *read_gbs, double *copy_gbs)
{
    if (!read_gbs || !copy_gbs || repeats < 1 || repeats > 1000) { return fail(PATHED_E_INVALID, "bad argument"); }
    if (bytes < (1u << 20) || bytes > ((size_t)64 << 30)) { return fail(PATHED_E_INVALID, "probe size must be 1 MiB .. 64 GiB"); }
    if (g_device < 0) {
        const int code = pathed_hip_init(0);
        if (code != PATHED_OK) { return code; }
    }
    const size_t count = bytes / sizeof(float4);
    DeviceBuffer<float4> source, target;
    DeviceBuffer<float> sink;
    hipEvent_t start = nullptr, stop = nullptr;
    auto cleanup = [&]() {
        source.release(); target.release(); sink.release();
        if (start) { (void)hipEventDestroy(start); }
        if (stop) { (void)hipEventDestroy(stop); }
    };
    hipError_t status = source.allocate(count);
    if (status == hipSuccess) { status = target.allocate(count); }
    if (status == hipSuccess) { status = sink.allocate(1); }
    if (status == hipSuccess) { status = hipMemset(source.ptr, 0, count * sizeof(float4)); }
    if (status == hipSuccess) { status = hipMemset(target.ptr, 0, count * sizeof(float4)); }
    if (status == hipSuccess) { status = hipEventCreate(&start); }
    if (status == hipSuccess) { status = hipEventCreate(&stop); }
    if (status != hipSuccess) { cleanup(); return fail(PATHED_E_DEVICE, std::string("bandwidth probe: ") + hipGetErrorString(status)); }

    hipDeviceProp_t properties;
    int units = 256;
    if (hipGetDeviceProperties(&properties, g_device) == hipSuccess && properties.multiProcessorCount > 0) { units = properties.multiProcessorCount; }
    const dim3 grid((unsigned)(units * 16)), block(kBlock);
    float readMs = 0.f, copyMs = 0.f;
    hipLaunchKernelGGL(k_stream_read, grid, block, 0, nullptr, source.ptr, count, sink.ptr);   // warm-up
    hipLaunchKernelGGL(k_stream_copy, grid, block, 0, nullptr, source.ptr, target.ptr, count);
    (void)hipEventRecord(start, nullptr);
    for (int r = 0; r < repeats; r++) { hipLaunchKernelGGL(k_stream_read, grid, block, 0, nullptr, source.ptr, count, sink.ptr); }
    (void)hipEventRecord(stop, nullptr);
    status = hipEventSynchronize(stop);
    if (status == hipSuccess) { status = hipEventElapsedTime(&readMs, start, stop); }
    (void)hipEventRecord(start, nullptr);
    for (int r = 0; r < repeats; r++) { hipLaunchKernelGGL(k_stream_copy, grid, block, 0, nullptr, source.ptr, target.ptr, count); }
    (void)hipEventRecord(stop, nullptr);
    if (status == hipSuccess) { status = hipEventSynchronize(stop); }
    if (status == hipSuccess) { status = hipEventElapsedTime(&copyMs, start, stop); }
    cleanup();
    if (status != hipSuccess || !(readMs > 0.f) || !(copyMs > 0.f)) {
        return fail(PATHED_E_DEVICE, std::string("bandwidth probe: ") + hipGetErrorString(status));
    }
    const double moved = (double)count * sizeof(float4) * repeats;
    *read_gbs = moved / (readMs * 1e-3) / 1e9;
    *copy_gbs = 2.0 * moved / (copyMs * 1e-3) / 1e9;   // bytes read + bytes written
    return PATHED_OK;
}

int pathed_hip_measure_valu(int waves_per_simd, int repeats, double *fma_rate, double *mixed_rate)
{
    if (!fma_rate || !mixed_rate) { return fail(PATHED_E_INVALID, "bad argument"); }
    double rates[5] = { 0.0, 0.0, 0.0, 0.0, 0.0 };
    const int code = pathed_hip_measure_valu_modes(waves_per_simd, repeats, rates, 2);
    *fma_rate = rates[0];
    *mixed_rate = rates[1];
    return code;
}

int pathed_hip_measure_valu_modes(int waves_per_simd, int repeats, double *rates, int n_modes)
{
    if (!rates || repeats < 1 || repeats > 1000 || n_modes < 1 || n_modes > 5) { return fail(PATHED_E_INVALID, "bad argument"); }
    if (waves_per_simd < 1 || waves_per_simd > 8) { return fail(PATHED_E_INVALID, "waves_per_simd must be 1..8"); }
    if (g_device < 0) {
        const int code = pathed_hip_init(0);
        if (code != PATHED_OK) { return code; }
    }
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    hipDeviceProp_t properties;
    int units = 256;
    if (hipGetDeviceProperties(&properties, device) == hipSuccess && properties.multiProcessorCount > 0) { units = properties.multiProcessorCount; }
    float *sink = nullptr;
    hipEvent_t start = nullptr, stop = nullptr;
    hipError_t status = hipMalloc((void **)&sink, sizeof(float));
    if (status == hipSuccess) { status = hipEventCreate(&start); }
    if (status == hipSuccess) { status = hipEventCreate(&stop); }
    // one 256-thread block = one wave on each of a CU's four SIMDs; k blocks per CU = k waves per SIMD
    const dim3 grid((unsigned)(units * waves_per_simd)), block(kBlock);
    const int iterations = 8192;   // 393 216 instructions per wave and launch
    auto launch = [&](int mode) {
        switch (mode) {
        case 0: hipLaunchKernelGGL((k_valu_probe<0>), grid, block, 0, nullptr, iterations, 1.f, sink); break;
        case 1: hipLaunchKernelGGL((k_valu_probe<1>), grid, block, 0, nullptr, iterations, 1.f, sink); break;
        case 2: hipLaunchKernelGGL((k_valu_probe<2>), grid, block, 0, nullptr, iterations, 1.f, sink); break;
        case 3: hipLaunchKernelGGL((k_valu_probe<3>), grid, block, 0, nullptr, iterations, 1.f, sink); break;
        default: hipLaunchKernelGGL((k_valu_probe<4>), grid, block, 0, nullptr, iterations, 1.f, sink); break;
        }
    };
    const double issued = (double)grid.x * kWavesPerBlock * (double)iterations * kValuProbeUnroll * repeats;
    for (int mode = 0; mode < n_modes && status == hipSuccess; mode++) {
        float ms = 0.f;
        launch(mode);   // warm-up
        (void)hipEventRecord(start, nullptr);
        for (int r = 0; r < repeats; r++) { launch(mode); }
        (void)hipEventRecord(stop, nullptr);
        status = hipEventSynchronize(stop);
        if (status == hipSuccess) { status = hipEventElapsedTime(&ms, start, stop); }
        if (status == hipSuccess) { status = hipGetLastError(); }
        if (status == hipSuccess && !(ms > 0.f)) { status = hipErrorUnknown; }
        if (status == hipSuccess) { rates[mode] = issued / (ms * 1e-3); }
    }
    if (sink) { (void)hipFree(sink); }
    if (start) { (void)hipEventDestroy(start); }
    if (stop) { (void)hipEventDestroy(stop); }
    if (status != hipSuccess) { return fail(PATHED_E_DEVICE, std::string("VALU probe: ") + hipGetErrorString(status)); }
    return PATHED_OK;
}

int pathed_hip_measure_valu_clocks(int waves_per_simd, int chains, int repeats, PathedValuClocks *out)
{
#if !PATHED_EXPERIMENTS
    (void)waves_per_simd; (void)chains; (void)repeats; (void)out;
    return fail(PATHED_E_UNSUPPORTED, "the clocked VALU probe lives in libpathed_hip_experiments.so (`make experiments`)");
#else
    if (!out || repeats < 1 || repeats > 1000) { return fail(PATHED_E_INVALID, "bad argument"); }
    if (waves_per_simd < 1 || waves_per_simd > 8) { return fail(PATHED_E_INVALID, "waves_per_simd must be 1..8"); }
    if (chains != 8 && chains != 16) { return fail(PATHED_E_INVALID, "chains must be 8 or 16"); }
    if (g_device < 0) {
        const int code = pathed_hip_init(0);
        if (code != PATHED_OK) { return code; }
    }
    std::memset(out, 0, sizeof *out);
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    hipDeviceProp_t properties;
    int units = 256;
    if (hipGetDeviceProperties(&properties, device) == hipSuccess && properties.multiProcessorCount > 0) { units = properties.multiProcessorCount; }
    int wallKhz = 0;
    if (hipDeviceGetAttribute(&wallKhz, hipDeviceAttributeWallClockRate, device) != hipSuccess || wallKhz <= 0) { wallKhz = 100000; }   // 100 MHz on CDNA
    int peakKhz = 0;
    (void)hipDeviceGetAttribute(&peakKhz, hipDeviceAttributeClockRate, device);
    const dim3 grid((unsigned)(units * waves_per_simd)), block(kBlock);
    const size_t waves = (size_t)grid.x * kWavesPerBlock;
    const int iterations = 8192;   // 393 216 instructions per wave and launch
    float *sink = nullptr;
    unsigned long long *clocks = nullptr;
    hipEvent_t start = nullptr, stop = nullptr;
    hipError_t status = hipMalloc((void **)&sink, sizeof(float));
    if (status == hipSuccess) { status = hipMalloc((void **)&clocks, 2 * waves * sizeof(unsigned long long)); }
    if (status == hipSuccess) { status = hipEventCreate(&start); }
    if (status == hipSuccess) { status = hipEventCreate(&stop); }
    auto launch = [&]() {
        if (chains == 16) { hipLaunchKernelGGL((k_valu_clock_probe<16>), grid, block, 0, nullptr, iterations, 1.f, sink, clocks); }
        else { hipLaunchKernelGGL((k_valu_clock_probe<8>), grid, block, 0, nullptr, iterations, 1.f, sink, clocks); }
    };
    float ms = 0.f;
    if (status == hipSuccess) {
        launch();   // warm-up
        (void)hipEventRecord(start, nullptr);
        for (int r = 0; r < repeats; r++) { launch(); }
        (void)hipEventRecord(stop, nullptr);
        status = hipEventSynchronize(stop);
    }
    if (status == hipSuccess) { status = hipEventElapsedTime(&ms, start, stop); }
    std::vector<unsigned long long> host(2 * waves);
    if (status == hipSuccess) { status = hipMemcpy(host.data(), clocks, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost); }
    if (status == hipSuccess && ms > 0.f) {
        const double perWave = (double)iterations * kValuProbeUnroll;   // instructions of the last launch's waves
        double shaderTicks = 0.0, wallTicks = 0.0;
        for (size_t w = 0; w < waves; w++) { shaderTicks += (double)host[2 * w]; wallTicks += (double)host[2 * w + 1]; }
        out->rate = (double)waves * perWave * repeats / (ms * 1e-3);
        out->wall_clock_mhz = wallKhz / 1e3;
        out->peak_clock_mhz = peakKhz / 1e3;
        out->shader_clock_mhz = wallTicks > 0.0 ? shaderTicks / wallTicks * out->wall_clock_mhz : 0.0;
        // a wave issues its instructions while waves_per_simd - 1 others share its SIMD: the SIMD's cycles per instruction
        // are the wave's ticks per instruction divided by the waves that share it
        out->wave_ticks_per_instruction = shaderTicks / ((double)waves * perWave);
        out->cycles_per_instruction = out->wave_ticks_per_instruction / waves_per_simd;
        // ... and from the event time alone, against the frequency measured in the same launches
        const double simds = (double)units * 4.0;
        out->cycles_per_instruction_events = out->shader_clock_mhz > 0.0 ? simds * out->shader_clock_mhz * 1e6 / out->rate : 0.0;
    }
    if (sink) { (void)hipFree(sink); }
    if (clocks) { (void)hipFree(clocks); }
    if (start) { (void)hipEventDestroy(start); }
    if (stop) { (void)hipEventDestroy(stop); }
    if (status != hipSuccess) { return fail(PATHED_E_DEVICE, std::string("VALU clock probe: ") + hipGetErrorString(status)); }
    return PATHED_OK;
#endif
}

int pathed_hip_accum_alloc(PathedScene *scene, size_t count, float **out)
{
    if (!scene || !out || count == 0) { return fail(PATHED_E_INVALID, "bad argument"); }
    SELECT_DEVICE(scene);
    float *buffer = nullptr;
    HIP_TRY(hipMalloc((void **)&buffer, count * sizeof(float)));
    const hipError_t status = hipMemset(buffer, 0, count * sizeof(float));
    if (status != hipSuccess) { (void)hipFree(buffer); return fail(PATHED_E_DEVICE, hipGetErrorString(status)); }
    *out = buffer;
    return PATHED_OK;
}

int pathed_hip_accum_free(PathedScene *scene, float *buffer)
{
    if (!scene) { return fail(PATHED_E_INVALID, "null scene"); }
    SELECT_DEVICE(scene);
    if (buffer) { HIP_TRY(hipFree(buffer)); }
    return PATHED_OK;
}

int pathed_hip_accum_download(PathedScene *scene, const float *buffer, size_t count, float *host)
{
    if (!scene || !buffer || !host) { return fail(PATHED_E_INVALID, "null argument"); }
    SELECT_DEVICE(scene);
    HIP_TRY(hipMemcpy(host, buffer, count * sizeof(float), hipMemcpyDeviceToHost));
    return PATHED_OK;
}

int pathed_hip_accum_upload(PathedScene *scene, float *buffer, size_t count, const float *host)
{
    if (!scene || !buffer || !host) { return fail(PATHED_E_INVALID, "null argument"); }
    SELECT_DEVICE(scene);
    HIP_TRY(hipMemcpy(buffer, host, count * sizeof(float), hipMemcpyHostToDevice));
    return PATHED_OK;
}

int pathed_hip_accum_copy_peer(PathedScene *dst_scene, float *dst, PathedScene *src_scene, const float *src, size_t count)
{
    if (!dst_scene || !src_scene || !dst || !src) { return fail(PATHED_E_INVALID, "null argument"); }
    SELECT_DEVICE(dst_scene);
    if (dst_scene->deviceId == src_scene->deviceId) {
        HIP_TRY(hipMemcpy(dst, src, count * sizeof(float), hipMemcpyDeviceToDevice));
    } else {
        HIP_TRY(hipMemcpyPeer(dst, dst_scene->deviceId, src, src_scene->deviceId, count * sizeof(float)));
    }
    return PATHED_OK;
}

int pathed_hip_accum_add(PathedScene *dst_scene, float *dst, const float *src, size_t count)
{
    if (!dst_scene || !dst || !src) { return fail(PATHED_E_INVALID, "null argument"); }
    SELECT_DEVICE(dst_scene);
    if (count == 0) { return PATHED_OK; }
    size_t blocks = (count + kBlock - 1) / kBlock;
    if (blocks > (size_t)dst_scene->computeUnits * 16) { blocks = (size_t)dst_scene->computeUnits * 16; }
    hipLaunchKernelGGL(k_accum_add, dim3((unsigned)blocks), dim3(kBlock), 0, nullptr, dst, src, count);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return PATHED_OK;
}

// ---- RCCL reduce of the per-device radiance sums (include/pathed_hip.h: multi-GPU fan-in) ----
// librccl is opened on first use: its types are restated here so that neither this library nor a single-GPU
// host links against it (rccl.h: ncclFloat32 = 7, ncclSum = 0, ncclSuccess = 0).
namespace {
struct RcclApi {
    void *library = nullptr;
    int (*commInitAll)(void **, int, const int *) = nullptr;
    int (*commDestroy)(void *) = nullptr;
    int (*reduce)(const void *, void *, size_t, int, int, int, void *, hipStream_t) = nullptr;
    int (*groupStart)() = nullptr;
    int (*groupEnd)() = nullptr;
    const char *(*errorString)(int) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rcclMutex;

bool loadRccl(std::string *why)
{
    std::lock_guard<std::mutex> guard(g_rcclMutex);
    if (g_rccl.library) { return true; }
    void *library = nullptr;
    for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
        library = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (library) { break; }
    }
    if (!library) { *why = std::string("cannot load librccl: ") + dlerror(); return false; }
    RcclApi api;
    api.library = library;
    api.commInitAll = reinterpret_cast<int (*)(void **, int, const int *)>(dlsym(library, "ncclCommInitAll"));
    api.commDestroy = reinterpret_cast<int (*)(void *)>(dlsym(library, "ncclCommDestroy"));
    api.reduce = reinterpret_cast<int (*)(const void *, void *, size_t, int, int, int, void *, hipStream_t)>(dlsym(library, "ncclReduce"));
    api.groupStart = reinterpret_cast<int (*)()>(dlsym(library, "ncclGroupStart"));
    api.groupEnd = reinterpret_cast<int (*)()>(dlsym(library, "ncclGroupEnd"));
    api.errorString = reinterpret_cast<const char *(*)(int)>(dlsym(library, "ncclGetErrorString"));
    if (!api.commInitAll || !api.commDestroy || !api.reduce || !api.groupStart || !api.groupEnd || !api.errorString) {
        *why = "librccl lacks one of ncclCommInitAll / ncclCommDestroy / ncclReduce / ncclGroupStart / ncclGroupEnd / ncclGetErrorString";
        dlclose(library);
        return false;
    }
    g_rccl = api;
    return true;
}
}  // namespace

struct PathedComm {
    std::vector<int> devices;
    std::vector<void *> comms;          // ncclComm_t per device
    std::vector<hipStream_t> streams;   // one per device
};

int pathed_hip_comm_init(int n_devices, const int *device_ids, PathedComm **out)
{
    if (!out) { return fail(PATHED_E_INVALID, "out pointer is null"); }
    *out = nullptr;
    if (n_devices < 1 || !device_ids) { return fail(PATHED_E_INVALID, "a communicator needs at least one device"); }
    int visible = 0;
    HIP_TRY(hipGetDeviceCount(&visible));
    for (int i = 0; i < n_devices; i++) {
        if (device_ids[i] < 0 || device_ids[i] >= visible) { return fail(PATHED_E_INVALID, "device id out of range"); }
        for (int k = 0; k < i; k++) {
            if (device_ids[k] == device_ids[i]) { return fail(PATHED_E_UNSUPPORTED, "RCCL takes one rank per device: replicas that share a GPU use pathed_hip_accum_copy_peer"); }
        }
    }
    std::string why;
    if (!loadRccl(&why)) { return fail(PATHED_E_UNSUPPORTED, why); }
    PathedComm *comm = new PathedComm();
    comm->devices.assign(device_ids, device_ids + n_devices);
    comm->comms.assign((size_t)n_devices, nullptr);
    comm->streams.assign((size_t)n_devices, nullptr);
    const int status = g_rccl.commInitAll(comm->comms.data(), n_devices, device_ids);
    if (status != 0) {
        const std::string message = std::string("ncclCommInitAll: ") + g_rccl.errorString(status);
        delete comm;
        return fail(PATHED_E_DEVICE, message);
    }
    for (int i = 0; i < n_devices; i++) {
        hipError_t error = hipSetDevice(device_ids[i]);
        if (error == hipSuccess) { error = hipStreamCreateWithFlags(&comm->streams[(size_t)i], hipStreamNonBlocking); }
        if (error != hipSuccess) {
            const std::string message = std::string("communicator stream: ") + hipGetErrorString(error);
            pathed_hip_comm_destroy(comm);
            return fail(PATHED_E_DEVICE, message);
        }
    }
    *out = comm;
    return PATHED_OK;
}

int pathed_hip_comm_reduce(PathedComm *comm, const float *const *send, float *recv_root, size_t count)
{
    if (!comm || !send || !recv_root) { return fail(PATHED_E_INVALID, "null communicator or buffer"); }
    const size_t n = comm->devices.size();
    for (size_t r = 0; r < n; r++) { if (!send[r]) { return fail(PATHED_E_INVALID, "null send buffer"); } }
    // whatever the caller queued on the devices' null streams (its renders are blocking) is done; the collective runs on
    // the communicator's own streams
    int status = g_rccl.groupStart();
    for (size_t r = 0; r < n && status == 0; r++) {
        status = g_rccl.reduce(send[r], r == 0 ? recv_root : nullptr, count, /* ncclFloat32 */ 7, /* ncclSum */ 0, /* root */ 0,
                               comm->comms[r], comm->streams[r]);
    }
    const int ended = g_rccl.groupEnd();
    if (status == 0) { status = ended; }
    if (status != 0) { return fail(PATHED_E_DEVICE, std::string("ncclReduce: ") + g_rccl.errorString(status)); }
    for (size_t r = 0; r < n; r++) {
        HIP_TRY(hipSetDevice(comm->devices[r]));
        HIP_TRY(hipStreamSynchronize(comm->streams[r]));
    }
    return PATHED_OK;
}

void pathed_hip_comm_destroy(PathedComm *comm)
{
    if (!comm) { return; }
    for (size_t r = 0; r < comm->devices.size(); r++) {
        if (comm->streams[r]) { (void)hipSetDevice(comm->devices[r]); (void)hipStreamDestroy(comm->streams[r]); }
        if (comm->comms[r] && g_rccl.commDestroy) { (void)g_rccl.commDestroy(comm->comms[r]); }
    }
    delete comm;
}

int pathed_hip_scene_create(const PathedSceneDesc *desc, PathedScene **out)
{
    return pathed_hip_scene_create_ex(desc, nullptr, out);
}

int pathed_hip_scene_device(const PathedScene *scene)
{
    if (!scene) { return fail(PATHED_E_INVALID, "null scene"); }
    return scene->deviceId;
}

static int unitOrderFromEnvironment(int fallback)
{
    if (const char *text = tuningEnv("PATHED_UNIT_ORDER")) {   // experiments
        if (!strcmp(text, "tiles")) { return kOrderTiles; }
        if (!strcmp(text, "stripes-tiled")) { return kOrderStripesTiled; }
        if (!strcmp(text, "stripes")) { return kOrderStripes; }
    }
    return fallback;
}

// The fused kernel's phase-1 records depend on the camera (the tolerances of the parallelograms scale with the distance a ray
// origin may have from a triangle): built at scene_create and again when the camera moves.
static hipError_t rebuildSmallItems(PathedScene *scene)
{
    if (!scene->bruteForce || scene->device.nTris <= 0) { return hipSuccess; }
    std::vector<float> points = scene->smallExtraPoints;
    for (int a = 0; a < 3; a++) { points.push_back(scene->device.camera.origin[a]); }
    // (the parallelogram instantiations of k_path_small keep the material table in LDS beside the path-state stash and the
    // shared resolve's lists, 33.5 KiB per block: up to 64 materials -- 6 KiB -- four blocks fit a CU's 160 KiB; scenes with more
    // pair nothing)
    const int kMaxQuadMaterials = 64;
    const bool pairQuads = scene->options.generic_kernels == 0 && !tuningEnv("PATHED_NO_QUADS") && scene->device.nMaterials <= kMaxQuadMaterials;
    std::vector<float> ordered;
    scene->smallLayout = buildSmallItems(scene->bvh.leafTris.data(), scene->device.nTris, points.data(), (int)(points.size() / 3), pairQuads, PATHED_TNEAR,
                                         reinterpret_cast<float *>(scene->smallItems.data), &ordered);
    std::vector<float4> records(ordered.size() / 4);
    std::memcpy(records.data(), ordered.data(), ordered.size() * sizeof(float));
    scene->itemTrisHost = ordered;
    const hipError_t status = scene->itemTris.upload(records);
    if (status == hipSuccess && scene->mfmaPhase1) {
        // (experiments build) the matrix-pipe rows are expressed in a frame centred on the camera: a new camera is a new table,
        // or phase 1 stops being conservative for rays that start far from the old one
        std::vector<float> table;
        buildMfmaTable(scene->itemTrisHost.data(), scene->device.nTris, scene->device.camera.origin, 1, &table, &scene->mfmaFrame);
        return scene->mfmaTable.upload(table);
    }
    return status;
}

// [r5] the phase-1 records of the hybrid kernel's direct set (they depend on the camera only through the scene's extent)
static hipError_t rebuildHybridItems(PathedScene *scene)
{
    if (!scene->hybridAvailable || scene->hybridDirectTris <= 0) { return hipSuccess; }
    std::vector<float> points = scene->hybridBounds;
    for (int a = 0; a < 3; a++) { points.push_back(scene->device.camera.origin[a]); }
    std::vector<float> ordered;
    scene->hybridLayout = buildSmallItems(scene->hybridDirectLeaf.data(), scene->hybridDirectTris, points.data(), (int)(points.size() / 3), true, PATHED_TNEAR,
                                          reinterpret_cast<float *>(scene->hybridItems.data), &ordered);
    std::vector<float4> records(ordered.size() / 4);
    std::memcpy(records.data(), ordered.data(), ordered.size() * sizeof(float));
    return scene->hybridItemTris.upload(records);
}

// [r5] The wavefront's local rays: a sphere-free BVH scene with at most kMaxLocalTris LARGE triangles (each at least 1 / 256 of
// the scene's surface: a floor, a backdrop) keeps their records for k_shade, with the bounds of everything else -- a box and
// a sphere about its centre, padded: the tests that use them only cull.
static void buildLocalSet(PathedScene *scene, const PathedSceneDesc *desc)
{
    scene->localCount = 0;
    const uint32_t n = desc->n_triangles;
    if (n == 0 || desc->n_spheres != 0) { return; }
    std::vector<double> area(n);
    double total = 0.0;
    for (uint32_t i = 0; i < n; i++) {
        const float *a = desc->positions + 3 * (size_t)desc->indices[3 * i], *b = desc->positions + 3 * (size_t)desc->indices[3 * i + 1], *c = desc->positions + 3 * (size_t)desc->indices[3 * i + 2];
        const double e1[3] = { (double)b[0] - a[0], (double)b[1] - a[1], (double)b[2] - a[2] }, e2[3] = { (double)c[0] - a[0], (double)c[1] - a[1], (double)c[2] - a[2] };
        const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
        area[i] = 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
        total += area[i];
    }
    std::vector<uint32_t> large;
    for (uint32_t i = 0; i < n; i++) {
        if (area[i] * 256.0 >= total) {
            large.push_back(i);
            if (large.size() > (size_t)kMaxLocalTris) { return; }   // a scene of many large faces: no "rest" worth culling against
        }
    }
    if (large.empty() || large.size() == n) { return; }
    std::vector<char> isLarge(n, 0);
    for (uint32_t i : large) { isLarge[i] = 1; }
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    for (uint32_t i = 0; i < n; i++) {
        if (isLarge[i]) { continue; }
        for (int k = 0; k < 3; k++) {
            const float *v = desc->positions + 3 * (size_t)desc->indices[3 * i + k];
            for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], (double)v[a]); hi[a] = std::max(hi[a], (double)v[a]); }
        }
    }
    double extent = 0.0, centre[3], radius2 = 0.0;
    for (int a = 0; a < 3; a++) { extent = std::max(extent, std::max(hi[a] - lo[a], std::max(std::fabs(lo[a]), std::fabs(hi[a])))); }
    for (int a = 0; a < 3; a++) { centre[a] = (double)(float)(0.5 * (lo[a] + hi[a])); }
    for (uint32_t i = 0; i < n; i++) {
        if (isLarge[i]) { continue; }
        for (int k = 0; k < 3; k++) {
            const float *v = desc->positions + 3 * (size_t)desc->indices[3 * i + k];
            double d2 = 0.0;
            for (int a = 0; a < 3; a++) { const double d = (double)v[a] - centre[a]; d2 += d * d; }
            radius2 = std::max(radius2, d2);
        }
    }
    const double radius = std::sqrt(radius2) * 1.0001 + 1e-4 * extent + 1e-30;
    for (int a = 0; a < 3; a++) {
        scene->localLo[a] = (float)(lo[a] - 1e-4 * extent - 1e-30);
        scene->localHi[a] = (float)(hi[a] + 1e-4 * extent + 1e-30);
        scene->localSphere[a] = (float)centre[a];
    }
    scene->localSphere[3] = (float)(radius * radius * 1.00001);
    for (size_t k = 0; k < large.size(); k++) {
        const uint32_t i = large[k];
        const float *v0 = desc->positions + 3 * (size_t)desc->indices[3 * i], *v1 = desc->positions + 3 * (size_t)desc->indices[3 * i + 1], *v2 = desc->positions + 3 * (size_t)desc->indices[3 * i + 2];
        float prim;
        const int id = (int)i;
        std::memcpy(&prim, &id, 4);
        // the record the tree's leaves hold for the triangle (bvh_build.h): v0, e1 = v1 - v0, e2 = v2 - v0 in fp32
        scene->localTris[3 * k + 0] = make_float4(v0[0], v0[1], v0[2], prim);
        scene->localTris[3 * k + 1] = make_float4(v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2], 0.f);
        scene->localTris[3 * k + 2] = make_float4(v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2], 0.f);
    }
    scene->localCount = (int)large.size();
}

// [r5] Splits a scene of 65 .. kHybridMaxTris triangles for k_path_hybrid: the (up to 64) largest triangles -- each at least
// 1 / 256 of the scene's surface: walls, floors, boxes, what most rays hit and what no box around it would cull -- are tested
// directly; everything else gets a tree of its own (the host SAH builder, the node format of the scene's tree).
static hipError_t buildHybrid(PathedScene *scene, const PathedSceneDesc *desc)
{
    const uint32_t n = desc->n_triangles;
    std::vector<double> area(n);
    double total = 0.0;
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    for (uint32_t i = 0; i < n; i++) {
        const float *a = desc->positions + 3 * (size_t)desc->indices[3 * i], *b = desc->positions + 3 * (size_t)desc->indices[3 * i + 1], *c = desc->positions + 3 * (size_t)desc->indices[3 * i + 2];
        const double e1[3] = { (double)b[0] - a[0], (double)b[1] - a[1], (double)b[2] - a[2] }, e2[3] = { (double)c[0] - a[0], (double)c[1] - a[1], (double)c[2] - a[2] };
        const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
        area[i] = 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
        total += area[i];
        for (const float *v : { a, b, c }) { for (int x = 0; x < 3; x++) { lo[x] = std::min(lo[x], (double)v[x]); hi[x] = std::max(hi[x], (double)v[x]); } }
    }
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; i++) { order[i] = i; }
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return area[x] > area[y]; });
    std::vector<char> isDirect(n, 0);
    int nDirect = 0;
    for (uint32_t k = 0; k < n && nDirect < kBruteForceMaxTris; k++) {
        if (!(area[order[k]] * 256.0 >= total)) { break; }
        isDirect[order[k]] = 1;
        nDirect++;
    }
    // direct set, in primitive order
    scene->hybridDirectLeaf.clear();
    std::vector<uint32_t> treeIndices, treePrim;
    for (uint32_t i = 0; i < n; i++) {
        if (!isDirect[i]) {
            for (int k = 0; k < 3; k++) { treeIndices.push_back(desc->indices[3 * i + k]); }
            treePrim.push_back(i);
            continue;
        }
        const float *v0 = desc->positions + 3 * (size_t)desc->indices[3 * i], *v1 = desc->positions + 3 * (size_t)desc->indices[3 * i + 1], *v2 = desc->positions + 3 * (size_t)desc->indices[3 * i + 2];
        float record[12];
        for (int a = 0; a < 3; a++) { record[a] = v0[a]; record[4 + a] = v1[a] - v0[a]; record[8 + a] = v2[a] - v0[a]; }   // as bvh_build.h writes them
        bvh_detail::putInt(record + 3, (int)i);
        record[7] = 0.f; record[11] = 0.f;
        scene->hybridDirectLeaf.insert(scene->hybridDirectLeaf.end(), record, record + 12);
    }
    scene->hybridDirectTris = nDirect;
    scene->hybridTreeTris = (int)treePrim.size();
    scene->hybridBounds.clear();
    for (int corner = 0; corner < 8; corner++) {
        for (int a = 0; a < 3; a++) { scene->hybridBounds.push_back((float)(((corner >> a) & 1) ? hi[a] : lo[a])); }
    }
    scene->hybridNodeCount = 0;
    scene->hybridMaxStack = 1;
    hipError_t status = hipSuccess;
    if (!treePrim.empty()) {
        FlatBvh tree = buildBvh(desc->positions, treeIndices.data(), (uint32_t)treePrim.size(), nullptr, 0, scene->options.build_threads, 4u);   // (leaves of <= 4: bvh_build.h)
        // the builder numbered the part's triangles 0 .. : back to the scene's primitive ids (shading records, lights)
        for (size_t k = 0; k < treePrim.size(); k++) {
            int sub;
            std::memcpy(&sub, &tree.leafTris[12 * k + 3], 4);
            bvh_detail::putInt(&tree.leafTris[12 * k + 3], (int)treePrim[(size_t)sub]);
        }
        std::vector<float4> nodes(tree.nodes.size() / 4), tris(tree.leafTris.size() / 4);
        std::memcpy(nodes.data(), tree.nodes.data(), tree.nodes.size() * sizeof(float));
        std::memcpy(tris.data(), tree.leafTris.data(), tree.leafTris.size() * sizeof(float));
        if ((status = scene->hybridNodes.upload(nodes)) != hipSuccess) { return status; }
        if ((status = scene->hybridTris.upload(tris)) != hipSuccess) { return status; }
        scene->hybridNodeCount = tree.nodeCount;
        scene->hybridMaxStack = 3 * tree.maxDepth + 1;
        double tlo[3] = { 1e300, 1e300, 1e300 }, thi[3] = { -1e300, -1e300, -1e300 };
        for (uint32_t index : treeIndices) {
            for (int a = 0; a < 3; a++) { tlo[a] = std::min(tlo[a], (double)desc->positions[3 * (size_t)index + a]); thi[a] = std::max(thi[a], (double)desc->positions[3 * (size_t)index + a]); }
        }
        double extent = 0.0;
        for (int a = 0; a < 3; a++) { extent = std::max(extent, std::max(thi[a] - tlo[a], std::max(std::fabs(tlo[a]), std::fabs(thi[a])))); }
        for (int a = 0; a < 3; a++) {   // padded: the proxy test is a cull, the walk's own slab tests decide
            scene->hybridLo[a] = (float)(tlo[a] - 1e-4 * extent - 1e-30);
            scene->hybridHi[a] = (float)(thi[a] + 1e-4 * extent + 1e-30);
        }
        // a bounding sphere about the box's centre (not the smallest one: the vertex farthest from that centre decides)
        double centre[3], radius2 = 0.0;
        for (int a = 0; a < 3; a++) { centre[a] = (double)(float)(0.5 * (tlo[a] + thi[a])); }
        for (uint32_t index : treeIndices) {
            double d2 = 0.0;
            for (int a = 0; a < 3; a++) { const double d = (double)desc->positions[3 * (size_t)index + a] - centre[a]; d2 += d * d; }
            radius2 = std::max(radius2, d2);
        }
        const double radius = std::sqrt(radius2) * 1.0001 + 1e-4 * extent + 1e-30;
        for (int a = 0; a < 3; a++) { scene->hybridSphere[a] = (float)centre[a]; }
        scene->hybridSphere[3] = (float)(radius * radius * 1.00001);
    }
    scene->hybridAvailable = true;
    return rebuildHybridItems(scene);
}

int pathed_hip_scene_create_ex(const PathedSceneDesc *desc, const PathedSceneOptions *optionsIn, PathedScene **out)
{
    if (!out) { return fail(PATHED_E_INVALID, "out pointer is null"); }
    *out = nullptr;
    int code = validate(desc);
    if (code != PATHED_OK) { return code; }
    PathedSceneOptions options;
    std::memset(&options, 0, sizeof options);
    options.device = PATHED_DEVICE_CURRENT;
    if (optionsIn) {
        if (optionsIn->struct_size != sizeof(PathedSceneOptions)) { return fail(PATHED_E_INVALID, "PathedSceneOptions.struct_size mismatch"); }
        options = *optionsIn;
        if (options.generic_kernels != 0 && options.generic_kernels != 1) { return fail(PATHED_E_INVALID, "generic_kernels must be 0 or 1"); }
        if (options.node_format < 0 || options.node_format > 3) { return fail(PATHED_E_INVALID, "node_format must be 0 (automatic), 1 (128-byte float nodes), 2 (compressed 64-byte nodes) or 3 (compressed 8-wide nodes)"); }
        if (options.build_threads < 0 || options.build_threads > 4096) { return fail(PATHED_E_INVALID, "build_threads must be 0..4096"); }
        if (options.unit_order < 0 || options.unit_order > 3) { return fail(PATHED_E_INVALID, "unit_order must be 0..3"); }
        if (options.bvh_builder < 0 || options.bvh_builder > PATHED_BVH_PLOC_DEVICE + 1) { return fail(PATHED_E_INVALID, "unknown BVH builder"); }
        if (options.pools < 0 || options.pools > kMaxPools) { return fail(PATHED_E_INVALID, "pools must be 0..4"); }
        if (options.stack_rows != 0 && options.stack_rows != 8 && options.stack_rows != 16 && options.stack_rows != 22) {
            return fail(PATHED_E_INVALID, "stack_rows must be 0, 8, 16 or 22");
        }
        if (options.max_slots < 0 || (options.max_slots != 0 && options.max_slots < kBlock)) { return fail(PATHED_E_INVALID, "max_slots must be 0 or >= 256"); }
        if (options.shade_kernel < 0 || options.shade_kernel > 6) { return fail(PATHED_E_INVALID, "shade_kernel must be 0..6"); }
        if (options.stage_slots != 0 && options.stage_slots != 512 && options.stage_slots != 1024) { return fail(PATHED_E_INVALID, "stage_slots must be 0, 512 or 1024"); }
        if (options.refittable != 0 && options.refittable != 1) { return fail(PATHED_E_INVALID, "refittable must be 0 or 1"); }
        if (options.small_phase1 < 0 || options.small_phase1 > 2) { return fail(PATHED_E_INVALID, "small_phase1 must be 0 (automatic), 1 (VALU) or 2 (matrix pipe)"); }
        if (options.wave_max_ksamples < 0) { return fail(PATHED_E_INVALID, "wave_max_ksamples must be >= 0"); }
        if (options.wave_stragglers < -1 || options.wave_stragglers > 64) { return fail(PATHED_E_INVALID, "wave_stragglers must be -1 (none), 0 (default) or 1..64"); }
        if (options.wave_refill < 0 || options.wave_refill > 64) { return fail(PATHED_E_INVALID, "wave_refill must be 0 (default) or 1..64"); }
        if (options.chunks_per_pass < 0 || options.chunks_per_pass > 4096) { return fail(PATHED_E_INVALID, "chunks_per_pass must be 0 (default) or 1..4096"); }
        if (options.shade_chain != 0 && options.shade_chain != 1) { return fail(PATHED_E_INVALID, "shade_chain must be 0 (automatic) or 1 (off)"); }
        if (options.shade_launches < 0 || options.shade_launches > 16) { return fail(PATHED_E_INVALID, "shade_launches must be 0 (automatic) or 1..16"); }
        if (options.local_rays != 0 && options.local_rays != 1) { return fail(PATHED_E_INVALID, "local_rays must be 0 (automatic) or 1 (off)"); }
        if (options.hybrid_batch < 0 || options.hybrid_batch > 128) { return fail(PATHED_E_INVALID, "hybrid_batch must be 0 (default) or 1..128"); }
        if (options.hybrid_ready < -1 || options.hybrid_ready > 64) { return fail(PATHED_E_INVALID, "hybrid_ready must be -1 (never), 0 (default) or 1..64"); }
    }
    int deviceId = options.device;
    if (deviceId == PATHED_DEVICE_CURRENT) {
        if (g_device < 0) {
            code = pathed_hip_init(0);
            if (code != PATHED_OK) { return code; }
        }
        deviceId = g_device;
    } else {
        int count = 0;
        const hipError_t counted = hipGetDeviceCount(&count);
        if (counted != hipSuccess || count == 0) { return fail(PATHED_E_NO_DEVICE, std::string("no HIP device: ") + hipGetErrorString(counted)); }
        if (deviceId < 0 || deviceId >= count) { return fail(PATHED_E_INVALID, "device id out of range"); }
    }
    HIP_TRY(hipSetDevice(deviceId));

    PathedScene *scene = new PathedScene();
    std::memset(&scene->device, 0, sizeof scene->device);
    scene->deviceId = deviceId;
    scene->options = options;
    scene->width = desc->camera.width;
    scene->height = desc->camera.height;

    hipDeviceProp_t properties;
    if (hipGetDeviceProperties(&properties, deviceId) == hipSuccess) {
        scene->computeUnits = properties.multiProcessorCount > 0 ? properties.multiProcessorCount : 256;
    }

    DScene &d = scene->device;
    buildCamera(desc->camera, &d.camera);

    // image textures: one float4 array, texels already through Texture::lookup's
    // powf(x / 255.f, 2.2f) (reference src/texture.cpp:44-48; 256 possible values per channel)
    std::vector<size_t> textureOffset(desc->n_textures);
    {
        float gammaTable[256];
        for (int value = 0; value < 256; value++) { gammaTable[value] = powf((unsigned char)value / 255.f, 2.2f); }
        std::vector<float4> texels;
        for (uint32_t t = 0; t < desc->n_textures; t++) {
            const PathedTexture &texture = desc->textures[t];
            textureOffset[t] = texels.size();
            const size_t count = (size_t)texture.width * texture.height;
            for (size_t k = 0; k < count; k++) {
                texels.push_back(make_float4(gammaTable[texture.rgb[3 * k + 0]], gammaTable[texture.rgb[3 * k + 1]], gammaTable[texture.rgb[3 * k + 2]], 0.f));
            }
        }
        const hipError_t uploaded = scene->texels.upload(texels);
        if (uploaded != hipSuccess) {
            delete scene;
            return fail(PATHED_E_DEVICE, std::string("upload textures: ") + hipGetErrorString(uploaded));
        }
    }

    // materials
    std::vector<DMaterial> materials(desc->n_materials);
    for (uint32_t i = 0; i < desc->n_materials; i++) {
        materials[i] = buildMaterial(desc->materials[i]);
        if (desc->materials[i].albedo_type == PATHED_ALBEDO_TEXTURE) {
            const PathedTexture &texture = desc->textures[desc->materials[i].texture];
            materials[i].texSize = texture.width | (texture.height << 16);
            materials[i].texels = scene->texels.ptr + textureOffset[(size_t)desc->materials[i].texture];
        }
    }

    std::vector<DSphere> spheres(desc->n_spheres);
    for (uint32_t i = 0; i < desc->n_spheres; i++) {
        const PathedSphere &s = desc->spheres[i];
        for (int k = 0; k < 3; k++) {
            spheres[i].centerWorld[k] = s.center_world[k];
            spheres[i].centerSample[k] = s.center_sample[k];
        }
        spheres[i].radius = s.radius;
        spheres[i].material = s.material;
    }
    std::memset(scene->spherePairs, 0, sizeof scene->spherePairs);
    for (int pair = 0; pair < 8; pair++) { scene->spherePairs[pair][3][0] = -1.f; scene->spherePairs[pair][3][1] = -1.f; }   // radius^2 < 0: nothing touches it
    for (uint32_t i = 0; i < desc->n_spheres && i < 16u; i++) {
        for (int k = 0; k < 3; k++) { scene->spherePairs[i / 2][k][i % 2] = spheres[i].centerWorld[k]; }
        scene->spherePairs[i / 2][3][i % 2] = spheres[i].radius * spheres[i].radius;
    }

    // lights: emissive surfaces in model order, environment light last
    // (reference src/scene_parser.cpp:173-190)
    std::vector<DLight> lights;
    for (uint32_t g = 0; g < desc->n_geoms; g++) {
        const PathedGeom &geom = desc->geoms[g];
        if (geom.type == PATHED_GEOM_MESH) {
            for (int i = 0; i < geom.count; i++) {
                const int tri = geom.first + i;
                if (emits(materials[(size_t)desc->tri_material[tri]])) { lights.push_back({ 0, tri }); }
            }
        } else if (emits(materials[(size_t)desc->spheres[geom.first].material])) {
            lights.push_back({ 1, geom.first });
        }
    }

    // environment light: EnvironmentLight ctor, reference src/environment_light.cpp:14-53
    std::vector<float4> envRgba;
    std::vector<float> thetaCdf, phiCdf;
    std::vector<int> phiEmpty, thetaGuide, phiGuide;
    std::vector<float4> thetaRecords, phiRecords;
    if (desc->env) {
        const PathedEnvLight &env = *desc->env;
        const size_t texels = (size_t)env.width * env.height;
        envRgba.resize(texels);
        std::vector<float> luminance(texels, 0.f);
        for (size_t i = 0; i < texels; i++) {
            envRgba[i] = make_float4(env.rgba[4 * i], env.rgba[4 * i + 1], env.rgba[4 * i + 2], env.rgba[4 * i + 3]);
            luminance[i] += env.rgba[4 * i + 0];
            luminance[i] += env.rgba[4 * i + 1];
            luminance[i] += env.rgba[4 * i + 2];
        }
        thetaCdf.resize((size_t)env.height);
        phiCdf.resize(texels);
        phiEmpty.resize((size_t)env.height);
        std::vector<float> thetaData((size_t)env.height, 0.f);
        for (int row = 0; row < env.height; row++) {
            float thetaSum = 0.f;
            for (int col = 0; col < env.width; col++) { thetaSum += luminance[(size_t)row * env.width + col]; }
            phiEmpty[(size_t)row] = buildCdf(&luminance[(size_t)row * env.width], (size_t)env.width, &phiCdf[(size_t)row * env.width]) ? 1 : 0;
            thetaData[(size_t)row] = thetaSum;
        }
        d.env.thetaEmpty = buildCdf(thetaData.data(), (size_t)env.height, thetaCdf.data()) ? 1 : 0;
        // guide[j] = first i with cdf[i] >= j / size (the last entry when there is none)
        auto buildGuide = [](const float *cdf, size_t size, int *guide) {
            size_t i = 0;
            for (size_t j = 0; j <= size; j++) {
                const float threshold = (float)j / (float)size;
                while (i + 1 < size && !(cdf[i] >= threshold)) { i++; }
                guide[j] = (int)i;
            }
        };
        thetaGuide.resize((size_t)env.height + 1);
        buildGuide(thetaCdf.data(), (size_t)env.height, thetaGuide.data());
        phiGuide.resize((size_t)env.height * ((size_t)env.width + 1));
        for (int row = 0; row < env.height; row++) {
            buildGuide(&phiCdf[(size_t)row * env.width], (size_t)env.width, &phiGuide[(size_t)row * (env.width + 1)]);
        }
        // sampling records (device_scene.h): for the guide cell `bucket` the search window of cdfSample is
        // [guide[bucket - 1], guide[bucket + 2]]; the record carries its start, its end and the CDF around the start
        auto buildRecords = [](const float *cdf, const int *guide, size_t size, bool empty, float4 *records) {
            for (size_t bucket = 0; bucket <= size; bucket++) {
                const size_t below = bucket == 0 ? 0 : bucket - 1;
                const size_t above = bucket + 2 > size ? size : bucket + 2;
                const int lo = guide[below], hi = guide[above];
                auto at = [&](int index) { return cdf[(size_t)(index < 0 ? 0 : (index > (int)size - 1 ? (int)size - 1 : index))]; };
                int loBits = empty ? -1 : lo;
                float loAsFloat, hiAsFloat;
                std::memcpy(&loAsFloat, &loBits, 4);
                std::memcpy(&hiAsFloat, &hi, 4);
                records[2 * bucket + 0] = make_float4(loAsFloat, hiAsFloat, lo > 0 ? at(lo - 1) : 0.f, at(lo));
                records[2 * bucket + 1] = make_float4(at(lo + 1), at(lo + 2), 0.f, 0.f);
            }
        };
        thetaRecords.resize(2 * ((size_t)env.height + 1));
        buildRecords(thetaCdf.data(), thetaGuide.data(), (size_t)env.height, d.env.thetaEmpty != 0, thetaRecords.data());
        phiRecords.resize(2 * (size_t)env.height * ((size_t)env.width + 1));
        for (int row = 0; row < env.height; row++) {
            buildRecords(&phiCdf[(size_t)row * env.width], &phiGuide[(size_t)row * (env.width + 1)], (size_t)env.width,
                         phiEmpty[(size_t)row] != 0, &phiRecords[2 * (size_t)row * ((size_t)env.width + 1)]);
        }
        d.env.width = env.width;
        d.env.height = env.height;
        d.env.scale = env.scale;
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) {
                d.env.mapToWorld[3 * r + c] = env.map_to_world[4 * r + c];
                d.env.worldToMap[3 * r + c] = env.world_to_map[4 * r + c];
            }
        }
        d.hasEnv = 1;
        lights.push_back({ 2, 0 });
    }

    auto fail_cleanup = [&](hipError_t status, const char *what) {
        delete scene;
        return fail(PATHED_E_DEVICE, std::string(what) + ": " + hipGetErrorString(status));
    };

    hipError_t status;
    int builder = options.bvh_builder > 0 ? options.bvh_builder - 1 : PATHED_BVH_SAH_HOST;
    if (const char *text = tuningEnv("PATHED_BVH_BUILDER")) {
        if (!strcmp(text, "lbvh")) { builder = PATHED_BVH_LBVH_DEVICE; }
        else if (!strcmp(text, "ploc")) { builder = PATHED_BVH_PLOC_DEVICE; }
        else if (!strcmp(text, "sah")) { builder = PATHED_BVH_SAH_HOST; }
    }
    // tiny meshes take the all-triangles kernel, which wants the host copy of the records
    if (desc->n_triangles <= (uint32_t)kBruteForceMaxTris) { builder = PATHED_BVH_SAH_HOST; }
    scene->bvhBuilder = builder;
    // the triangle soup on the device: input of the device builders and of the shading-record gather
    DeviceBuffer<float> devicePositions, deviceNormals, deviceUvs;
    DeviceBuffer<uint32_t> deviceIndices;
    DeviceBuffer<int> deviceTriMaterial;
    auto releaseSoup = [&]() {
        devicePositions.release(); deviceNormals.release(); deviceUvs.release(); deviceIndices.release(); deviceTriMaterial.release();
    };
    if (desc->n_triangles > 0) {
        const size_t nv = desc->n_vertices, nt = desc->n_triangles;
        status = devicePositions.allocate(3 * nv);
        if (status == hipSuccess) { status = deviceNormals.allocate(3 * nv); }
        if (status == hipSuccess) { status = deviceUvs.allocate(2 * nv); }
        if (status == hipSuccess) { status = deviceIndices.allocate(3 * nt); }
        if (status == hipSuccess) { status = deviceTriMaterial.allocate(nt); }
        if (status == hipSuccess) { status = scene->triShade.allocate((size_t)kTriShadeQuads * nt); }
        if (status == hipSuccess) { status = scene->triCompact.allocate(nt); }
        if (status == hipSuccess) { status = hipMemcpy(devicePositions.ptr, desc->positions, 3 * nv * sizeof(float), hipMemcpyHostToDevice); }
        if (status == hipSuccess) { status = hipMemcpy(deviceNormals.ptr, desc->normals, 3 * nv * sizeof(float), hipMemcpyHostToDevice); }
        if (status == hipSuccess) { status = hipMemcpy(deviceUvs.ptr, desc->uvs, 2 * nv * sizeof(float), hipMemcpyHostToDevice); }
        if (status == hipSuccess) { status = hipMemcpy(deviceIndices.ptr, desc->indices, 3 * nt * sizeof(uint32_t), hipMemcpyHostToDevice); }
        if (status == hipSuccess) { status = hipMemcpy(deviceTriMaterial.ptr, desc->tri_material, nt * sizeof(int), hipMemcpyHostToDevice); }
        if (status == hipSuccess) {
            hipLaunchKernelGGL(k_build_tri_shade, dim3((unsigned)((8 * nt + kBlock - 1) / kBlock)), dim3(kBlock), 0, nullptr,
                               devicePositions.ptr, deviceNormals.ptr, deviceUvs.ptr, deviceIndices.ptr, deviceTriMaterial.ptr,
                               (uint32_t)nt, scene->triShade.ptr, scene->triCompact.ptr);
            status = hipGetLastError();
        }
        if (status != hipSuccess) { releaseSoup(); return fail_cleanup(status, "upload triangle soup"); }
    }

    if (builder == PATHED_BVH_LBVH_DEVICE || builder == PATHED_BVH_PLOC_DEVICE) {
        // rtcCommitScene's stand-in on the device (lbvh.h): build, keep the result in place
        DeviceBvh built;
        std::string message;
        // spheres join the tree as one-sphere leaves, as on the host (reference src/sphere.cpp:16-48: each is a geometry of its own)
        DeviceBuffer<float4> sphereBounds;
        if (desc->n_spheres > 0) {
            std::vector<float4> bounds(desc->n_spheres);
            for (uint32_t i = 0; i < desc->n_spheres; i++) {
                bounds[i] = make_float4(desc->spheres[i].center_world[0], desc->spheres[i].center_world[1], desc->spheres[i].center_world[2], desc->spheres[i].radius);
            }
            if ((status = sphereBounds.upload(bounds)) != hipSuccess) { releaseSoup(); return fail_cleanup(status, "upload sphere bounds"); }
            scene->spheresInTree = true;
        }
        status = buildBvhOnDevice(builder == PATHED_BVH_PLOC_DEVICE ? kDeviceBuilderPloc : kDeviceBuilderLbvh,
                                  devicePositions.ptr, deviceIndices.ptr, desc->n_triangles, sphereBounds.ptr, desc->n_spheres, nullptr, &built, &message);
        if (status != hipSuccess) { releaseSoup(); return fail_cleanup(status, message.empty() ? "device BVH build" : message.c_str()); }
        scene->nodes.ptr = built.nodes;
        scene->nodes.count = built.nodeCapacity * 8;
        scene->leafTris.ptr = built.leafTris;
        scene->leafTris.count = (size_t)desc->n_triangles * 3;
        scene->bvh.nodeCount = built.nodeCount;
        scene->bvh.maxDepth = built.maxDepth;
        scene->bvhOnHost = false;
        scene->bvhBuildMs = built.buildMs;
    } else {
        const auto buildStart = std::chrono::steady_clock::now();
        // Spheres join the tree as one-sphere leaves (the reference gives each one to Embree as a geometry of its own,
        // src/sphere.cpp:16-48) unless the scene is small enough for the all-triangles kernel, which tests them one by one.
        std::vector<float> sphereBounds;
        const bool tiny = desc->n_triangles <= (uint32_t)kBruteForceMaxTris && desc->n_spheres <= (uint32_t)kBruteForceMaxSpheres
            && scene->options.intersector != 1 && !tuningEnv("PATHED_NO_BRUTE_FORCE");
        if (!tiny) {
            for (uint32_t i = 0; i < desc->n_spheres; i++) {
                for (int a = 0; a < 3; a++) { sphereBounds.push_back(desc->spheres[i].center_world[a]); }
                sphereBounds.push_back(desc->spheres[i].radius);
            }
        }
        scene->spheresInTree = !sphereBounds.empty();
        scene->bvh = buildBvh(desc->positions, desc->indices, desc->n_triangles, sphereBounds.data(), (uint32_t)(sphereBounds.size() / 4),
                              scene->options.build_threads);
        scene->bvhBuildMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - buildStart).count();
        std::vector<float4> nodes(scene->bvh.nodes.size() / 4), tris(scene->bvh.leafTris.size() / 4);
        std::memcpy(nodes.data(), scene->bvh.nodes.data(), scene->bvh.nodes.size() * sizeof(float));
        std::memcpy(tris.data(), scene->bvh.leafTris.data(), scene->bvh.leafTris.size() * sizeof(float));
        if ((status = scene->nodes.upload(nodes)) != hipSuccess) { releaseSoup(); return fail_cleanup(status, "upload nodes"); }
        if ((status = scene->leafTris.upload(tris)) != hipSuccess) { releaseSoup(); return fail_cleanup(status, "upload triangles"); }
    }
    status = hipDeviceSynchronize();   // the shading-record gather reads the soup
    if (status == hipSuccess && options.refittable && desc->n_triangles > 0) {
        // pathed_hip_scene_refit moves the vertices later: the soup stays (the buffers change owner, nothing is copied)
        auto adopt = [](auto &to, auto &from) { to.ptr = from.ptr; to.count = from.count; from.ptr = nullptr; from.count = 0; };
        adopt(scene->soupPositions, devicePositions); adopt(scene->soupNormals, deviceNormals); adopt(scene->soupUvs, deviceUvs);
        adopt(scene->soupIndices, deviceIndices); adopt(scene->soupTriMaterial, deviceTriMaterial);
        scene->soupVertices = desc->n_vertices;
        scene->refittable = true;
    }
    releaseSoup();
    if (status != hipSuccess) { return fail_cleanup(status, "build shading records"); }
    if ((status = scene->spheres.upload(spheres)) != hipSuccess) { return fail_cleanup(status, "upload spheres"); }
    if ((status = scene->materials.upload(materials)) != hipSuccess) { return fail_cleanup(status, "upload materials"); }
    if ((status = scene->lights.upload(lights)) != hipSuccess) { return fail_cleanup(status, "upload lights"); }
    {
        // participating media and every primitive's internal medium (Surface::getInternalMedium), triangles then spheres
        std::vector<DMedium> media(desc->n_media);
        for (uint32_t i = 0; i < desc->n_media; i++) {
            std::memset(&media[i], 0, sizeof(DMedium));
            for (int k = 0; k < 3; k++) { media[i].sigmaT[k] = desc->media[i].sigma_t[k]; media[i].sigmaS[k] = desc->media[i].sigma_s[k]; }
        }
        std::vector<int> primMedium((size_t)desc->n_triangles + desc->n_spheres, -1);
        for (uint32_t g = 0; g < desc->n_geoms; g++) {
            const PathedGeom &geom = desc->geoms[g];
            // (a medium with sigma_t = 0 stays a medium: its surface is still a container the volumetric queries skip; its
            // distance sample -log(1 - xi) / 0 is +inf, "no event", as in the reference)
            const int medium = geom.medium;
            if (geom.type == PATHED_GEOM_MESH) {
                for (int i = 0; i < geom.count; i++) { primMedium[(size_t)(geom.first + i)] = medium; }
            } else {
                primMedium[(size_t)desc->n_triangles + (size_t)geom.first] = medium;
            }
        }
        for (uint32_t i = 0; i < desc->n_materials; i++) { scene->hasContainers = scene->hasContainers || desc->materials[i].type == PATHED_MAT_PASSTHROUGH; }
        if ((status = scene->media.upload(media)) != hipSuccess) { return fail_cleanup(status, "upload media"); }
        if ((status = scene->primMedium.upload(primMedium)) != hipSuccess) { return fail_cleanup(status, "upload media"); }
    }
    if ((status = scene->envRgba.upload(envRgba)) != hipSuccess) { return fail_cleanup(status, "upload env map"); }
    if ((status = scene->thetaCdf.upload(thetaCdf)) != hipSuccess) { return fail_cleanup(status, "upload env cdf"); }
    if ((status = scene->phiCdf.upload(phiCdf)) != hipSuccess) { return fail_cleanup(status, "upload env cdf"); }
    if ((status = scene->phiEmpty.upload(phiEmpty)) != hipSuccess) { return fail_cleanup(status, "upload env cdf"); }
    if ((status = scene->thetaGuide.upload(thetaGuide)) != hipSuccess) { return fail_cleanup(status, "upload env cdf"); }
    if ((status = scene->phiGuide.upload(phiGuide)) != hipSuccess) { return fail_cleanup(status, "upload env cdf"); }
    if ((status = scene->thetaRecords.upload(thetaRecords)) != hipSuccess) { return fail_cleanup(status, "upload env cdf"); }
    if ((status = scene->phiRecords.upload(phiRecords)) != hipSuccess) { return fail_cleanup(status, "upload env cdf"); }

    d.nodes = scene->nodes.ptr;
    d.nodesQ = nullptr;
    d.leafTris = scene->leafTris.ptr;
    d.nNodes = scene->bvh.nodeCount;
    d.nTris = (int)desc->n_triangles;
    d.spheres = scene->spheres.ptr;
    d.nSpheres = (int)desc->n_spheres;
    d.nLinearSpheres = scene->spheresInTree ? 0 : (int)desc->n_spheres;
    d.triShade = scene->triShade.ptr;
    d.triCompact = scene->triCompact.ptr;
    {
        // plain ranges (device_scene.h): meshes all of whose vertices carry zero normals and zero uvs, e.g. an OBJ without
        // vn / vt or a PLY of positions
        d.nPlainRanges = 0;
        for (uint32_t g = 0; g < desc->n_geoms; g++) {
            const PathedGeom &geom = desc->geoms[g];
            if (geom.type != PATHED_GEOM_MESH || geom.count <= 0) { continue; }
            bool plain = true;
            for (int i = 0; i < geom.count && plain; i++) {
                for (int k = 0; k < 3 && plain; k++) {
                    const size_t v = desc->indices[3 * (size_t)(geom.first + i) + k];
                    plain = desc->normals[3 * v] == 0.f && desc->normals[3 * v + 1] == 0.f && desc->normals[3 * v + 2] == 0.f
                        && desc->uvs[2 * v] == 0.f && desc->uvs[2 * v + 1] == 0.f;
                }
            }
            if (!plain) { continue; }
            if (d.nPlainRanges > 0 && d.plainEnd[d.nPlainRanges - 1] == geom.first) { d.plainEnd[d.nPlainRanges - 1] = geom.first + geom.count; }
            else if (d.nPlainRanges < kMaxPlainRanges) {
                d.plainBegin[d.nPlainRanges] = geom.first;
                d.plainEnd[d.nPlainRanges] = geom.first + geom.count;
                d.nPlainRanges++;
            }
        }
    }
    d.materials = scene->materials.ptr;
    d.nMaterials = (int)desc->n_materials;
    d.lights = scene->lights.ptr;
    d.nLights = (int)lights.size();
    {
        bool emissive = false;
        for (const DMaterial &material : materials) { emissive = emissive || emits(material); }
        scene->envOnly = desc->env != nullptr && lights.size() == 1 && !emissive && options.generic_kernels == 0 && !tuningEnv("PATHED_NO_ENV_ONLY");
        bool plainLambertian = desc->env == nullptr && desc->n_spheres == 0;
        for (uint32_t i = 0; i < desc->n_materials; i++) {
            plainLambertian = plainLambertian && desc->materials[i].type == PATHED_MAT_LAMBERTIAN && desc->materials[i].albedo_type == PATHED_ALBEDO_CONSTANT;
        }
        const bool narrow = options.generic_kernels == 0 && !tuningEnv("PATHED_NO_SCENE_TRAITS");
        scene->lambertianTriangles = plainLambertian && narrow;
        bool lambertianPlastic = true;
        for (uint32_t i = 0; i < desc->n_materials; i++) {
            lambertianPlastic = lambertianPlastic && (desc->materials[i].type == PATHED_MAT_LAMBERTIAN || desc->materials[i].type == PATHED_MAT_PLASTIC)
                && desc->materials[i].albedo_type == PATHED_ALBEDO_CONSTANT;
        }
        bool triangleLights = false;
        for (const DLight &light : lights) { triangleLights = triangleLights || light.kind == 0; }
        scene->lambertianPlasticSpheres = lambertianPlastic && desc->env == nullptr && !triangleLights && narrow;
        bool constantAlbedo = true, noContainer = true;
        for (uint32_t i = 0; i < desc->n_materials; i++) {
            constantAlbedo = constantAlbedo && desc->materials[i].albedo_type == PATHED_ALBEDO_CONSTANT;
            noContainer = noContainer && desc->materials[i].type != PATHED_MAT_PASSTHROUGH;
        }
        scene->triangleLit = constantAlbedo && noContainer && desc->env == nullptr && desc->n_spheres == 0 && narrow;
        {
            unsigned kinds = 0u, distributions = 0u;
            for (uint32_t i = 0; i < desc->n_materials; i++) {
                const int type = desc->materials[i].type;
                kinds |= 1u << type;
                if (type == PATHED_MAT_MICROFACET || type == PATHED_MAT_PLASTIC) { distributions |= desc->materials[i].distribution == PATHED_DIST_GGX ? 2u : 1u; }
            }
            const unsigned rough = (1u << PATHED_MAT_LAMBERTIAN) | (1u << PATHED_MAT_OREN_NAYAR) | (1u << PATHED_MAT_MICROFACET) | (1u << PATHED_MAT_PLASTIC);
            const unsigned smooth = (1u << PATHED_MAT_LAMBERTIAN) | (1u << PATHED_MAT_GLASS) | (1u << PATHED_MAT_MIRROR);
            scene->roughBeckmann = scene->triangleLit && (kinds & ~rough) == 0u && (distributions & 2u) == 0u;
            scene->roughGgx = scene->triangleLit && (kinds & ~rough) == 0u && distributions == 2u;
            scene->smoothSet = scene->triangleLit && (kinds & ~smooth) == 0u;
        }
        bool glassContainer = desc->env == nullptr;
        for (uint32_t i = 0; i < desc->n_materials; i++) {
            const int type = desc->materials[i].type;
            glassContainer = glassContainer && (type == PATHED_MAT_LAMBERTIAN || type == PATHED_MAT_GLASS || type == PATHED_MAT_PASSTHROUGH)
                && desc->materials[i].albedo_type == PATHED_ALBEDO_CONSTANT;
        }
        scene->lambertianGlassContainer = glassContainer && narrow;
    }
    d.media = scene->media.ptr;
    d.primMedium = scene->primMedium.ptr;
    d.nMedia = (int)desc->n_media;
    d.env.rgba = scene->envRgba.ptr;
    d.env.thetaCdf = scene->thetaCdf.ptr;
    d.env.phiCdf = scene->phiCdf.ptr;
    d.env.phiEmpty = scene->phiEmpty.ptr;
    d.env.thetaGuide = scene->thetaGuide.ptr;
    d.env.phiGuide = scene->phiGuide.ptr;
    d.env.thetaRecords = scene->thetaRecords.ptr;
    d.env.phiRecords = scene->phiRecords.ptr;

    if (options.pools > 0) { scene->pools = options.pools; }
    if (const char *poolCount = tuningEnv("PATHED_POOLS")) {
        const int value = atoi(poolCount);
        scene->pools = value < 1 ? 1 : value > kMaxPools ? kMaxPools : value;
    }
    scene->bruteForce = scene->device.nTris <= kBruteForceMaxTris && scene->device.nSpheres <= kBruteForceMaxSpheres
        && options.intersector != 1 && !tuningEnv("PATHED_NO_BRUTE_FORCE");
    // BVH scenes: more slots = more rays per persistent wave to refill finished lanes from (ray cost is heavy-tailed) and
    // fewer, longer launches: 8 Mi slots (1.3 GB of path state) against 4 Mi: +3.6 % on the teapot, +6.1 % on the 5.2 M-
    // triangle mesh at >= 256 spp per call, 16 Mi +1 % / +7.6 %, 32 Mi less again (tools/slots_sweep.py); short calls want
    // fewer (renderPass: at least 16 units per slot).  The all-triangles wavefront kernel has uniform cost and prefers the
    // smaller, Infinity-Cache-resident state.
    scene->maxSlots = scene->bruteForce ? (1 << 20) : (1 << 23);
    scene->adaptiveSlots = !scene->bruteForce;
    // which kernels carry the radiance loop: scenes of <= 64 triangles default to the fused in-register path
    // kernel, everything else to the wavefront with the per-slot shade kernel; the staged shade kernel and the split
    // shade stage (k_vertex + k_regen over lists the trace kernel writes) are selectable: both issue fewer
    // instructions on fuller waves and both lose to the coalesced per-slot kernel (DESIGN.md has the measurements)
    int shadeKernel = options.shade_kernel;
    if (const char *text = tuningEnv("PATHED_SHADE_KERNEL")) {   // experiments: "per-slot" | "staged" | "fused" | "split"
        if (!strcmp(text, "per-slot")) { shadeKernel = 1; }
        else if (!strcmp(text, "staged")) { shadeKernel = 2; }
        else if (!strcmp(text, "fused")) { shadeKernel = 3; }
        else if (!strcmp(text, "split")) { shadeKernel = 4; }
        else if (!strcmp(text, "wave")) { shadeKernel = 5; }
    }
    if (shadeKernel == 3 && !scene->bruteForce) {
        delete scene;
        return fail(PATHED_E_INVALID, "the fused path kernel serves scenes of at most 64 triangles that take the all-triangles intersector");
    }
#if !PATHED_EXPERIMENTS
    if (shadeKernel == 2 || shadeKernel == 4 || requestedNodeFormat(options) >= 2 || options.small_phase1 == 2) {
        delete scene;
        return fail(PATHED_E_UNSUPPORTED, "the staged and split shade stages, the compressed node formats and the matrix-pipe phase 1 are measured-and-rejected "
                                          "experiments: build libpathed_hip_experiments.so (`make experiments`) and load it instead");
    }
#endif
    if (shadeKernel == 4 && scene->bruteForce) {
        delete scene;
        return fail(PATHED_E_INVALID, "the split shade stage follows the BVH trace kernel: scenes of at most 64 triangles take the fused, per-slot or staged kernels");
    }
    scene->splitShade = shadeKernel == 4;
    configureTrace(scene);
    {
        // the compressed tree: sphere-free scenes whose tree stays in HBM, walked by the per-slot pipeline's trace kernel
        const int nodeFormat = requestedNodeFormat(options);
        const bool eligible = !scene->bruteForce && !scene->sceneInLds && !scene->splitShade && scene->device.nSpheres == 0
            && scene->device.nNodes > 0 && options.generic_kernels == 0;
        if (nodeFormat >= 2 && !eligible) {
            delete scene;
            return fail(PATHED_E_INVALID, "compressed nodes serve sphere-free scenes whose tree is walked in HBM by the per-slot pipeline (not generic_kernels, not the split stage)");
        }
        const int chosen = !eligible || nodeFormat == 1 ? 0 : nodeFormat == 0 ? PATHED_DEFAULT_NODE_FORMAT : nodeFormat - 1;
#if PATHED_EXPERIMENTS
        if (chosen != 0) {
            const size_t nNodes = (size_t)scene->device.nNodes;
            const unsigned blocks = (unsigned)((nNodes + kBlock - 1) / kBlock);
            if ((status = scene->nodesQ.allocate((chosen == 2 ? 8 : 4) * nNodes)) != hipSuccess) { return fail_cleanup(status, "allocate compressed nodes"); }
            if (chosen == 2) { hipLaunchKernelGGL(k_widen_nodes, dim3(blocks), dim3(kBlock), 0, nullptr, scene->nodes.ptr, (int)nNodes, scene->nodesQ.ptr); }
            else { hipLaunchKernelGGL(k_compress_nodes, dim3(blocks), dim3(kBlock), 0, nullptr, scene->nodes.ptr, (int)nNodes, scene->nodesQ.ptr); }
            if ((status = hipDeviceSynchronize()) != hipSuccess) { return fail_cleanup(status, "compress nodes"); }
            scene->device.nodesQ = scene->nodesQ.ptr;
            scene->nodeFormat = chosen;
        }
#else
        (void)chosen;
#endif
    }
    // persistent grids of the split stage: k_vertex at PATHED_VERTEX_WAVES blocks per CU, k_regen at PATHED_REGEN_WAVES
#if PATHED_EXPERIMENTS
    scene->vertexGrid = scene->computeUnits * PATHED_VERTEX_WAVES;
    scene->regenGrid = scene->computeUnits * PATHED_REGEN_WAVES;
#endif
    if (const char *text = tuningEnv("PATHED_VERTEX_GRID")) { const int value = atoi(text); if (value >= 1 && value <= 65536) { scene->vertexGrid = value; } }
    if (const char *text = tuningEnv("PATHED_REGEN_GRID")) { const int value = atoi(text); if (value >= 1 && value <= 65536) { scene->regenGrid = value; } }
    scene->fusedPath = scene->bruteForce && (shadeKernel == 0 || shadeKernel == 3);
    scene->waveMode = shadeKernel == 5 ? 2 : (shadeKernel == 0 || shadeKernel == 6) ? 0 : 1;
    scene->waveAvailable = !scene->bruteForce && scene->device.nMaterials <= kMaxLdsMaterials && scene->nodeFormat == 0;
    {
        // [r5] k_path_hybrid: BVH scenes small enough that a handful of large triangles carry most of the hits (path_hybrid.h)
        // (shade_kernel 6 asks for it on a larger mesh as well: the walk is the same, the tree just stops being cache-resident)
        const bool eligible = !scene->bruteForce && desc->n_spheres == 0
            && scene->device.nMaterials <= kMaxLdsMaterials && scene->nodeFormat == 0 && options.intersector != 1 && !scene->hasContainers
            && !options.refittable;   // (a refittable scene keeps ONE tree, the one pathed_hip_scene_refit moves)
        if (shadeKernel == 6 && !eligible) {
            delete scene;
            return fail(PATHED_E_INVALID, "the hybrid path kernel serves sphere-free scenes of more than 64 triangles and at most 96 materials (intersector 0, not refittable)");
        }
        if (eligible && ((shadeKernel == 0 && desc->n_triangles <= (uint32_t)kHybridMaxTris) || shadeKernel == 6)) {
            if ((status = buildHybrid(scene, desc)) != hipSuccess) { return fail_cleanup(status, "build the hybrid kernel's scene split"); }
            scene->hybridPath = true;
        }
    }
    // (not for refittable scenes: their vertices move, the large triangles' records and the bounds of the rest would go stale)
    if (!scene->bruteForce && options.local_rays != 1 && scene->nodeFormat == 0 && !options.refittable) { buildLocalSet(scene, desc); }
    if (shadeKernel == 5 && !scene->waveAvailable) {
        delete scene;
        return fail(PATHED_E_INVALID, "the wave path kernel serves BVH scenes (more than 64 triangles or intersector 1) of at most 96 materials over the float nodes");
    }
    if (options.wave_max_ksamples > 0) { scene->waveMaxSamples = (unsigned long long)options.wave_max_ksamples << 10; }
    if (options.wave_stragglers != 0) { scene->waveStragglers = options.wave_stragglers < 0 ? 0 : options.wave_stragglers; }
    if (options.wave_refill != 0) { scene->waveRefill = options.wave_refill; }
    if (const char *text = tuningEnv("PATHED_WAVE_MAX_SAMPLES")) { scene->waveMaxSamples = strtoull(text, nullptr, 10); }
    if (const char *text = tuningEnv("PATHED_WAVE_BLOCK")) { scene->waveBlock = atoi(text) != 0; }
    if (const char *text = tuningEnv("PATHED_WAVE_SHADE_READY")) { const int value = atoi(text); if (value >= 1 && value <= 64) { scene->waveShadeReady = value; } }
    if (const char *text = tuningEnv("PATHED_WAVE_REFILL")) { const int value = atoi(text); if (value >= 1 && value <= 64) { scene->waveRefill = value; } }
    if (const char *text = tuningEnv("PATHED_WAVE_STRAGGLERS")) { const int value = atoi(text); if (value >= 0 && value <= 64) { scene->waveStragglers = value; } }
    scene->stagedShade = shadeKernel == 2;
    // more slots per block = fuller last waves of the dense stages, fewer blocks to fill the chip with:
    // the 0.5 Mi-slot pools of the all-triangles scenes take 512, the 2 Mi-slot pools of the BVH scenes 1024
    scene->stageRounds = scene->bruteForce ? 2 : 4;
    if (options.stage_slots != 0) { scene->stageRounds = options.stage_slots / kBlock; }
    if (const char *text = tuningEnv("PATHED_STAGE_SLOTS")) {
        const int value = atoi(text);
        if (value == 512 || value == 1024) { scene->stageRounds = value / kBlock; }
    }
    std::memset(&scene->smallTris, 0, sizeof scene->smallTris);
    if (scene->bruteForce) {
        // pairs of leaf-ordered triangles, component-interleaved (kernels.h: SmallTris); the odd
        // one out is paired with an all-zero triangle, masked off in the kernel
        const float *tris = scene->bvh.leafTris.data();
        for (int k = 0; k < scene->device.nTris; k++) {
            const float *tri = tris + (size_t)12 * k;  // (v0, prim) (e1, -) (e2, -)
            f2 *record = scene->smallTris.data + (size_t)kSmallPairWords * (k / 2);
            for (int row = 0; row < 3; row++) {
                for (int axis = 0; axis < 3; axis++) { record[3 * row + axis][k & 1] = tri[4 * row + axis]; }
            }
        }
    }
    std::memset(&scene->smallItems, 0, sizeof scene->smallItems);
    scene->smallExtraPoints.clear();
    for (uint32_t i = 0; i < desc->n_spheres; i++) {
        for (int sign = -1; sign <= 1; sign += 2) {
            for (int a = 0; a < 3; a++) { scene->smallExtraPoints.push_back(desc->spheres[i].center_world[a] + sign * std::fabs(desc->spheres[i].radius) * 1.001f); }
        }
    }
    if ((status = rebuildSmallItems(scene)) != hipSuccess) { return fail_cleanup(status, "upload the phase-1 records"); }
    scene->mfmaPhase1 = false;
    std::memset(&scene->mfmaFrame, 0, sizeof scene->mfmaFrame);
    if (scene->bruteForce && scene->fusedPath) {
        int phase1 = options.small_phase1;
        if (const char *text = tuningEnv("PATHED_SMALL_PHASE1")) {
            if (!strcmp(text, "valu")) { phase1 = 1; } else if (!strcmp(text, "mfma")) { phase1 = 2; }
        }
        if (phase1 == 0) { phase1 = kDefaultSmallPhase1; }
        if (phase1 == 2 && scene->device.nTris > 0) {
            std::vector<float> table;
            // rows in ITEM order: phase 2 of the fused kernel indexes the item-ordered triangle records
            buildMfmaTable(scene->itemTrisHost.data(), scene->device.nTris, scene->device.camera.origin, 1, &table, &scene->mfmaFrame);
            if ((status = scene->mfmaTable.upload(table)) != hipSuccess) { return fail_cleanup(status, "upload the matrix-pipe rows"); }
            scene->mfmaPhase1 = true;
        }
    }
    scene->unitOrder = unitOrderFromEnvironment(options.unit_order == 2 ? kOrderStripesTiled : options.unit_order == 3 ? kOrderTiles : kOrderStripes);
    if (options.max_slots >= kBlock) { scene->maxSlots = options.max_slots; scene->adaptiveSlots = false; }
    if (const char *slots = tuningEnv("PATHED_MAX_SLOTS")) {
        const long value = atol(slots);
        if (value >= kBlock) { scene->maxSlots = (int)value; scene->adaptiveSlots = false; }
    }
    *out = scene;
    return PATHED_OK;
}

void pathed_hip_scene_destroy(PathedScene *scene)
{
    if (scene) { (void)hipSetDevice(scene->deviceId); }
    delete scene;
}

// The unit order of one pool's launch parameters (kernels.h: THE UNIT ORDER).  The (nQueues x pools) queues share out
// ITEMS round-robin -- the chunks of the pass (stripes: an item is one whole image of units) or groups of kUnitGroup pixels
// (tiles: an item is all chunks of the group) -- queue q of pool `pool` owns the items (q * pools + pool) + j * queues.
// Returns false when the unit ids of the pass do not fit 32 bits.
static bool fillUnitOrder(RenderParams &q, int order, int width, int height, int chunksPerPixel, int pool, int pools, int nQueues)
{
    const unsigned long long nPixels = (unsigned long long)width * (unsigned long long)height;
    const unsigned long long allQueues = (unsigned long long)nQueues * (unsigned long long)pools;
    unsigned long long items, itemUnits, lastItemShortfall = 0;
    if (order == kOrderTiles) {
        items = (nPixels + kUnitGroup - 1ull) / kUnitGroup;
        itemUnits = (unsigned long long)kUnitGroup * (unsigned long long)chunksPerPixel;
        q.lastGroupPixels = (unsigned int)(nPixels - (items - 1ull) * kUnitGroup);
        lastItemShortfall = (kUnitGroup - (unsigned long long)q.lastGroupPixels) * (unsigned long long)chunksPerPixel;
    } else {
        items = (unsigned long long)chunksPerPixel;
        itemUnits = nPixels;
        q.lastGroupPixels = kUnitGroup;
    }
    const unsigned long long stride = ((items + allQueues - 1ull) / allQueues) * itemUnits;   // the longest queue
    if (stride * (unsigned long long)nQueues >= 0xFFFFFFF0ull || nPixels + kUnitGroup >= 0xFFFFFFF0ull) { return false; }
    q.unitOrder = order;
    q.pool = (unsigned int)pool;
    q.pools = (unsigned int)pools;
    q.nQueues = nQueues;
    q.nGroups = (unsigned int)items;
    q.groupUnits = (unsigned int)itemUnits;
    q.unitsPerQueue = (unsigned int)stride;
    q.divStride = makeFastDiv((unsigned int)stride);
    q.divGroupUnits = makeFastDiv((unsigned int)itemUnits);
    q.divBand = makeFastDiv(8u * (unsigned int)width);
    q.divWidth = makeFastDiv((unsigned int)width);
    unsigned long long total = 0;
    for (int k = 0; k < kUnitQueues; k++) {
        unsigned long long units = 0;
        const unsigned long long first = (unsigned long long)k * (unsigned long long)pools + (unsigned long long)pool;
        if (k < nQueues && first < items) {
            units = ((items - 1ull - first) / allQueues + 1ull) * itemUnits;
            if ((items - 1ull - first) % allQueues == 0ull) { units -= lastItemShortfall; }   // owns the ragged last group
        }
        q.queueUnits[k] = (unsigned int)units;
        total += units;
    }
    q.nUnits = (unsigned int)total;
    return true;
}

// VolumePathTracer on a scene WITHOUT media and without passthrough containers is PathTracer: the reference's two integrators
// share their direct-lighting arithmetic statement for statement (src/direct_lighting_helper.cpp:37-187 against
// src/path_tracer.cpp:79-216), every volumetric query degenerates to the regular one and no segment scatters -- the oracle's
// two integrators and k_path_volume / the path-tracer kernels give the same floats there (tests/test_gpu_volume.py).  Such a
// scene therefore runs the path tracer's kernels (fused or wavefront: one path per lane with no refill is 0.5-0.6x of them);
// PathedSceneOptions.generic_kernels = 1 keeps k_path_volume, for that very comparison.
static bool usesVolumeKernel(const PathedScene *scene)
{
    if (scene->integrator != PATHED_INTEGRATOR_VOLUME_PATH_TRACER) { return false; }
    return scene->hasContainers || scene->device.nMedia > 0 || scene->options.generic_kernels != 0;
}

// One internal pass of the fused path kernel (scenes of <= 64 triangles): a single persistent launch
// renders every unit of the pass; no slot pool, no iteration loop, no polling.
static int renderPassFused(PathedScene *scene, uint64_t seed, uint32_t begin, uint32_t count,
                           int start_bounce, int last_bounce, float *d_accum, hipStream_t stream)
{
    const int nPixels = scene->width * scene->height;
    const int chunk = scene->samplesPerUnit;
    const int chunksPerPixel = (int)((count + (uint32_t)chunk - 1) / (uint32_t)chunk);
    const unsigned long long nUnits64 = (unsigned long long)nPixels * (unsigned long long)chunksPerPixel;
    if (nUnits64 >= 0xFFFFFFF0ull) { return fail(PATHED_E_INVALID, "too many work units in one pass"); }
    const unsigned int nUnits = (unsigned int)nUnits64;

    if (scene->chunkCapacity < (size_t)nUnits) {
        HIP_TRY(scene->chunkBuf.allocate((size_t)nUnits));
        scene->chunkCapacity = (size_t)nUnits;
    }
    if (!scene->counters.ptr) { HIP_TRY(scene->counters.allocate(kMaxPools * kCtrCount)); }
    if (!scene->stats.ptr) {
        HIP_TRY(scene->stats.allocate(kStatCount));
        HIP_TRY(hipMemset(scene->stats.ptr, 0, kStatCount * sizeof(unsigned long long)));
    }

    // persistent grid: what the register budget keeps resident (PATHED_FUSED_WAVES waves per SIMD = blocks per CU),
    // or fewer when the pass has fewer than 64 units per wave
    unsigned long long blocks = (unsigned long long)scene->computeUnits * PATHED_FUSED_WAVES;
    const unsigned long long blocksNeeded = (nUnits64 + (unsigned long long)kBlock - 1) / kBlock;
    if (blocks > blocksNeeded) { blocks = blocksNeeded; }
    if (blocks < 1) { blocks = 1; }
    const unsigned int waves = (unsigned int)blocks * kWavesPerBlock;

    RenderParams params;
    std::memset(&params, 0, sizeof params);
    params.scene = scene->device;
    params.state.chunkBuf = scene->chunkBuf.ptr;
    params.counters = scene->counters.ptr;
    params.stats = scene->stats.ptr;
    params.accum = d_accum;
    params.nPixels = nPixels;
    if (!fillUnitOrder(params, scene->unitOrder, scene->width, scene->height, chunksPerPixel, 0, 1,
                       (int)(waves < (unsigned int)kUnitQueues ? waves : (unsigned int)kUnitQueues))) {
        return fail(PATHED_E_INVALID, "too many work units in one pass");
    }
    {
        // a wave reserves up to one unit per lane at a time; passes too small for that hand out less per atomic,
        // so that the last reservations of a queue do not leave most waves idle
        const unsigned int wavesPerQueue = (waves + (unsigned int)params.nQueues - 1) / (unsigned int)params.nQueues;
        unsigned int grab = (nUnits / (unsigned int)params.nQueues) / (wavesPerQueue * 4u);
        params.unitGrab = (int)(grab < 1u ? 1u : grab > 64u ? 64u : grab);
    }
    params.chunk = chunk;
    params.chunksPerPixel = chunksPerPixel;
    params.seedLo = (uint32_t)seed;
    params.seedHi = (uint32_t)(seed >> 32);
    params.sppBegin = begin;
    params.sppEnd = begin + count;
    params.startBounce = start_bounce;
    params.lastBounce = last_bounce;

    HIP_TRY(hipMemsetAsync(params.counters, 0, kCtrCount * sizeof(unsigned int), stream));
    const dim3 grid((unsigned)blocks), block(kBlock);
    const bool ldsMaterials = scene->device.nMaterials <= kMaxLdsMaterials;
    const size_t lds = ldsMaterials ? (size_t)scene->device.nMaterials * sizeof(DMaterial) : 0;
    int timed = -1;
    if (scene->timeKernels) {
        timed = scene->traceEvents.acquire();
        (void)hipEventRecord(scene->traceEvents.start[timed], stream);
    }
    // the narrowest instantiation whose compile-time scene set contains this scene's (shading.h: SceneTraits)
    params.mfmaTable = scene->mfmaTable.ptr;
    params.mfmaFrame = scene->mfmaFrame;
    params.smallQuads = scene->smallLayout.nQuads;
    params.smallKappaT = scene->smallLayout.kappaT;
    params.scene.leafTris = scene->itemTris.ptr;   // phase 2 indexes the triangles in the order phase 1's bits come in
    std::memcpy(params.spherePairs, scene->spherePairs, sizeof params.spherePairs);
#if PATHED_EXPERIMENTS
    if (scene->mfmaPhase1 && ldsMaterials) {
        // phase 1 on the matrix pipe: the same instantiations with MFMA = true
        if (scene->lambertianTriangles) {
            if (scene->countMode) { hipLaunchKernelGGL((k_path_small<true, true, TraitsLambertianTriangles, true>), grid, block, lds, stream, params, scene->smallItems); }
            else { hipLaunchKernelGGL((k_path_small<true, false, TraitsLambertianTriangles, true>), grid, block, lds, stream, params, scene->smallItems); }
        } else if (scene->lambertianPlasticSpheres) {
            if (scene->countMode) { hipLaunchKernelGGL((k_path_small<true, true, TraitsLambertianPlasticSpheres, true>), grid, block, lds, stream, params, scene->smallItems); }
            else { hipLaunchKernelGGL((k_path_small<true, false, TraitsLambertianPlasticSpheres, true>), grid, block, lds, stream, params, scene->smallItems); }
        } else {
            if (scene->countMode) { hipLaunchKernelGGL((k_path_small<true, true, TraitsAll, true>), grid, block, lds, stream, params, scene->smallItems); }
            else { hipLaunchKernelGGL((k_path_small<true, false, TraitsAll, true>), grid, block, lds, stream, params, scene->smallItems); }
        }
    } else
#endif
    if (scene->smallLayout.nQuads > 0 && ldsMaterials) {
        // some triangles are halves of parallelograms: phase 1 tests those as parallelograms (small_items.h)
        if (scene->lambertianTriangles) {
            if (scene->countMode) { hipLaunchKernelGGL((k_path_small<true, true, TraitsLambertianTriangles, false, true>), grid, block, lds, stream, params, scene->smallItems); }
            else { hipLaunchKernelGGL((k_path_small<true, false, TraitsLambertianTriangles, false, true>), grid, block, lds, stream, params, scene->smallItems); }
        } else if (scene->lambertianPlasticSpheres) {
            if (scene->countMode) { hipLaunchKernelGGL((k_path_small<true, true, TraitsLambertianPlasticSpheres, false, true>), grid, block, lds, stream, params, scene->smallItems); }
            else { hipLaunchKernelGGL((k_path_small<true, false, TraitsLambertianPlasticSpheres, false, true>), grid, block, lds, stream, params, scene->smallItems); }
        } else if (scene->roughBeckmann && !scene->countMode) {
            hipLaunchKernelGGL((k_path_small<true, false, TraitsRoughBeckmann, false, true>), grid, block, lds, stream, params, scene->smallItems);
        } else if (scene->roughGgx && !scene->countMode) {
            hipLaunchKernelGGL((k_path_small<true, false, TraitsRoughGgx, false, true>), grid, block, lds, stream, params, scene->smallItems);
        } else if (scene->smoothSet && !scene->countMode) {
            hipLaunchKernelGGL((k_path_small<true, false, TraitsSmooth, false, true>), grid, block, lds, stream, params, scene->smallItems);
        } else if (scene->triangleLit && !scene->countMode) {
            hipLaunchKernelGGL((k_path_small<true, false, TraitsTriangleLit, false, true>), grid, block, lds, stream, params, scene->smallItems);
        } else {
            if (scene->countMode) { hipLaunchKernelGGL((k_path_small<true, true, TraitsAll, false, true>), grid, block, lds, stream, params, scene->smallItems); }
            else { hipLaunchKernelGGL((k_path_small<true, false, TraitsAll, false, true>), grid, block, lds, stream, params, scene->smallItems); }
        }
    } else
    if (scene->lambertianTriangles) {
        if (ldsMaterials) {
            if (scene->countMode) { hipLaunchKernelGGL((k_path_small<true, true, TraitsLambertianTriangles>), grid, block, lds, stream, params, scene->smallItems); }
            else { hipLaunchKernelGGL((k_path_small<true, false, TraitsLambertianTriangles>), grid, block, lds, stream, params, scene->smallItems); }
        } else {
            if (scene->countMode) { hipLaunchKernelGGL((k_path_small<false, true, TraitsLambertianTriangles>), grid, block, lds, stream, params, scene->smallItems); }
            else { hipLaunchKernelGGL((k_path_small<false, false, TraitsLambertianTriangles>), grid, block, lds, stream, params, scene->smallItems); }
        }
    } else if (scene->lambertianPlasticSpheres && ldsMaterials) {
        if (scene->countMode) { hipLaunchKernelGGL((k_path_small<true, true, TraitsLambertianPlasticSpheres>), grid, block, lds, stream, params, scene->smallItems); }
        else { hipLaunchKernelGGL((k_path_small<true, false, TraitsLambertianPlasticSpheres>), grid, block, lds, stream, params, scene->smallItems); }
    } else if (ldsMaterials) {
        if (scene->countMode) { hipLaunchKernelGGL((k_path_small<true, true, TraitsAll>), grid, block, lds, stream, params, scene->smallItems); }
        else { hipLaunchKernelGGL((k_path_small<true, false, TraitsAll>), grid, block, lds, stream, params, scene->smallItems); }
    } else {
        if (scene->countMode) { hipLaunchKernelGGL((k_path_small<false, true, TraitsAll>), grid, block, lds, stream, params, scene->smallItems); }
        else { hipLaunchKernelGGL((k_path_small<false, false, TraitsAll>), grid, block, lds, stream, params, scene->smallItems); }
    }
    if (timed >= 0) { (void)hipEventRecord(scene->traceEvents.stop[timed], stream); }
    scene->traceLaunchesAll++;
    const dim3 pixelGrid((unsigned)((nPixels + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(k_resolve, pixelGrid, block, 0, stream, params);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));

    scene->iterations += 1;
    scene->cameraSamples += (unsigned long long)count * (unsigned long long)nPixels;
    return PATHED_OK;
}

// One internal pass of the volume integrator (k_path_volume): like renderPassFused, one persistent launch.
static int renderPassVolume(PathedScene *scene, uint64_t seed, uint32_t begin, uint32_t count,
                            int start_bounce, int last_bounce, float *d_accum, hipStream_t stream)
{
    const int nPixels = scene->width * scene->height;
    const int chunk = scene->samplesPerUnit;
    const int chunksPerPixel = (int)((count + (uint32_t)chunk - 1) / (uint32_t)chunk);
    const unsigned long long nUnits64 = (unsigned long long)nPixels * (unsigned long long)chunksPerPixel;
    if (nUnits64 >= 0xFFFFFFF0ull) { return fail(PATHED_E_INVALID, "too many work units in one pass"); }
    const unsigned int nUnits = (unsigned int)nUnits64;

    if (scene->chunkCapacity < (size_t)nUnits) {
        HIP_TRY(scene->chunkBuf.allocate((size_t)nUnits));
        scene->chunkCapacity = (size_t)nUnits;
    }
    if (!scene->counters.ptr) { HIP_TRY(scene->counters.allocate(kMaxPools * kCtrCount)); }
    if (!scene->stats.ptr) {
        HIP_TRY(scene->stats.allocate(kStatCount));
        HIP_TRY(hipMemset(scene->stats.ptr, 0, kStatCount * sizeof(unsigned long long)));
    }

    const bool ldsMaterials = scene->device.nMaterials <= kMaxLdsMaterials;
    const bool small = scene->bruteForce;   // <= 64 triangles, <= 16 spheres: the all-triangles intersector
    const size_t lds = (size_t)((small ? 8 : scene->stackRows) + 1) * kBlock * sizeof(int)
        + (ldsMaterials ? (size_t)scene->device.nMaterials * sizeof(DMaterial) : 0);
    unsigned long long blocks = (unsigned long long)scene->computeUnits * PATHED_VOLUME_WAVES;   // what the kernel's register budget keeps resident
    const unsigned long long blocksNeeded = (nUnits64 + (unsigned long long)kBlock - 1) / kBlock;
    if (blocks > blocksNeeded) { blocks = blocksNeeded; }
    if (blocks < 1) { blocks = 1; }
    const unsigned int waves = (unsigned int)blocks * kWavesPerBlock;
    const size_t overflowRows = (size_t)(scene->maxStack > scene->stackRows ? scene->maxStack - scene->stackRows : 0);
    const size_t overflowInts = (size_t)blocks * kBlock * (overflowRows ? overflowRows : 1);
    if (scene->volumeOverflow.count < overflowInts) { HIP_TRY(scene->volumeOverflow.allocate(overflowInts)); }

    RenderParams params;
    std::memset(&params, 0, sizeof params);
    params.scene = scene->device;
    params.state.chunkBuf = scene->chunkBuf.ptr;
    params.counters = scene->counters.ptr;
    params.stats = scene->stats.ptr;
    params.stackOverflow = scene->volumeOverflow.ptr;
    params.maxStack = scene->maxStack;
    params.accum = d_accum;
    params.nPixels = nPixels;
    if (!fillUnitOrder(params, scene->unitOrder, scene->width, scene->height, chunksPerPixel, 0, 1,
                       (int)(waves < (unsigned int)kUnitQueues ? waves : (unsigned int)kUnitQueues))) {
        return fail(PATHED_E_INVALID, "too many work units in one pass");
    }
    {
        const unsigned int wavesPerQueue = (waves + (unsigned int)params.nQueues - 1) / (unsigned int)params.nQueues;
        unsigned int grab = (nUnits / (unsigned int)params.nQueues) / (wavesPerQueue * 4u);
        params.unitGrab = (int)(grab < 1u ? 1u : grab > 64u ? 64u : grab);
    }
    params.chunk = chunk;
    params.chunksPerPixel = chunksPerPixel;
    params.seedLo = (uint32_t)seed;
    params.seedHi = (uint32_t)(seed >> 32);
    params.sppBegin = begin;
    params.sppEnd = begin + count;
    params.startBounce = start_bounce;
    params.lastBounce = last_bounce;

    HIP_TRY(hipMemsetAsync(params.counters, 0, kCtrCount * sizeof(unsigned int), stream));
    const dim3 grid((unsigned)blocks);
    int timed = -1;
    if (scene->timeKernels) {
        timed = scene->traceEvents.acquire();
        (void)hipEventRecord(scene->traceEvents.start[timed], stream);
    }
    // scenes made of quads: phase 1 over the item records (small_items.h), phase 2 over the item-ordered triangles
    const bool quads = small && ldsMaterials && scene->smallLayout.nQuads > 0;
    params.smallQuads = quads ? scene->smallLayout.nQuads : 0;
    params.smallKappaT = scene->smallLayout.kappaT;
    if (quads) { params.scene.leafTris = scene->itemTris.ptr; }
    launchVolume(scene->stackRows, small, scene->lambertianGlassContainer, params, quads ? scene->smallItems : scene->smallTris, grid, lds, ldsMaterials, stream);
    if (timed >= 0) { (void)hipEventRecord(scene->traceEvents.stop[timed], stream); }
    scene->traceLaunchesAll++;
    const dim3 pixelGrid((unsigned)((nPixels + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(k_resolve, pixelGrid, dim3(kBlock), 0, stream, params);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));

    scene->iterations += 1;
    scene->cameraSamples += (unsigned long long)count * (unsigned long long)nPixels;
    return PATHED_OK;
}

// One internal pass of the path tracer over a BVH with the paths on chip (k_path_wave, path_wave.h): one persistent launch.
static int renderPassWave(PathedScene *scene, uint64_t seed, uint32_t begin, uint32_t count,
                          int start_bounce, int last_bounce, float *d_accum, hipStream_t stream)
{
    const int nPixels = scene->width * scene->height;
    const int chunk = scene->samplesPerUnit;
    const int chunksPerPixel = (int)((count + (uint32_t)chunk - 1) / (uint32_t)chunk);
    const unsigned long long nUnits64 = (unsigned long long)nPixels * (unsigned long long)chunksPerPixel;
    if (nUnits64 >= 0xFFFFFFF0ull) { return fail(PATHED_E_INVALID, "too many work units in one pass"); }
    const unsigned int nUnits = (unsigned int)nUnits64;

    if (scene->chunkCapacity < (size_t)nUnits) {
        HIP_TRY(scene->chunkBuf.allocate((size_t)nUnits));
        scene->chunkCapacity = (size_t)nUnits;
    }
    if (!scene->counters.ptr) { HIP_TRY(scene->counters.allocate(kMaxPools * kCtrCount)); }
    if (!scene->stats.ptr) {
        HIP_TRY(scene->stats.allocate(kStatCount));
        HIP_TRY(hipMemset(scene->stats.ptr, 0, kStatCount * sizeof(unsigned long long)));
    }

    const int stackRows = 22;
    const size_t lds = pathWaveLdsBytes(stackRows, scene->device.nMaterials, PATHED_EXPERIMENTS && scene->waveBlock);
    unsigned long long blocks = (unsigned long long)scene->computeUnits * PATHED_WAVE_WAVES;
    const unsigned long long blocksNeeded = (nUnits64 + (unsigned long long)kBlock - 1) / kBlock;
    if (blocks > blocksNeeded) { blocks = blocksNeeded; }
    if (blocks < 1) { blocks = 1; }
    const unsigned int waves = (unsigned int)blocks * kWavesPerBlock;
    const size_t overflowRows = (size_t)(scene->maxStack > stackRows ? scene->maxStack - stackRows : 0);
    const size_t overflowInts = (size_t)blocks * kBlock * (overflowRows ? overflowRows : 1);
    if (scene->volumeOverflow.count < overflowInts) { HIP_TRY(scene->volumeOverflow.allocate(overflowInts)); }

    RenderParams params;
    std::memset(&params, 0, sizeof params);
    params.scene = scene->device;
    params.state.chunkBuf = scene->chunkBuf.ptr;
    params.counters = scene->counters.ptr;
    params.stats = scene->stats.ptr;
    params.stackOverflow = scene->volumeOverflow.ptr;
    params.maxStack = scene->maxStack;
    params.suspendLanes = scene->waveStragglers;
    params.suspendPatience = scene->waveRefill;   // k_path_wave: idle lanes are refilled from the wave's list once fewer than this many are busy
    params.accum = d_accum;
    params.nPixels = nPixels;
    if (!fillUnitOrder(params, scene->unitOrder, scene->width, scene->height, chunksPerPixel, 0, 1,
                       (int)(waves < (unsigned int)kUnitQueues ? waves : (unsigned int)kUnitQueues))) {
        return fail(PATHED_E_INVALID, "too many work units in one pass");
    }
    {
        const unsigned int wavesPerQueue = (waves + (unsigned int)params.nQueues - 1) / (unsigned int)params.nQueues;
        unsigned int grab = (nUnits / (unsigned int)params.nQueues) / (wavesPerQueue * 4u);
        params.unitGrab = (int)(grab < 1u ? 1u : grab > 64u ? 64u : grab);
    }
    params.chunk = chunk;
    params.chunksPerPixel = chunksPerPixel;
    params.seedLo = (uint32_t)seed;
    params.seedHi = (uint32_t)(seed >> 32);
    params.sppBegin = begin;
    params.sppEnd = begin + count;
    params.startBounce = start_bounce;
    params.lastBounce = last_bounce;

    HIP_TRY(hipMemsetAsync(params.counters, 0, kCtrCount * sizeof(unsigned int), stream));
    const dim3 grid((unsigned)blocks);
    int timed = -1;
    if (scene->timeKernels) {
        timed = scene->traceEvents.acquire();
        (void)hipEventRecord(scene->traceEvents.start[timed], stream);
    }
    params.parkMinCardsPerWave = scene->waveShadeReady;   // k_path_wave<BLOCK>: a wave shades once this many of its paths have their rays back
#if PATHED_EXPERIMENTS
    // the block's waves sharing one ray ring: measured 13-20 % slower than a list per wave (profiles/r4_ab_wave.log)
    if (scene->waveBlock && scene->envOnly && scene->device.nSpheres == 0 && !scene->hasContainers) { hipLaunchKernelGGL((k_path_wave<true, 22, TraitsEnvironmentOnly, false, true>), grid, dim3(kBlock), lds, stream, params); }
    else
#endif
    if (scene->envOnly && scene->device.nSpheres == 0 && !scene->hasContainers) { hipLaunchKernelGGL((k_path_wave<true, 22, TraitsEnvironmentOnly, false>), grid, dim3(kBlock), lds, stream, params); }
    else if (scene->smoothSet) { hipLaunchKernelGGL((k_path_wave<true, 22, TraitsSmooth, false>), grid, dim3(kBlock), lds, stream, params); }
    else if (scene->triangleLit) { hipLaunchKernelGGL((k_path_wave<true, 22, TraitsTriangleLit, false>), grid, dim3(kBlock), lds, stream, params); }
    else if (scene->device.nSpheres == 0) { hipLaunchKernelGGL((k_path_wave<true, 22, TraitsAll, false>), grid, dim3(kBlock), lds, stream, params); }
    else { hipLaunchKernelGGL((k_path_wave<true, 22, TraitsAll, true>), grid, dim3(kBlock), lds, stream, params); }
    if (timed >= 0) { (void)hipEventRecord(scene->traceEvents.stop[timed], stream); }
    scene->traceLaunchesAll++;
    const dim3 pixelGrid((unsigned)((nPixels + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(k_resolve, pixelGrid, dim3(kBlock), 0, stream, params);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));

    scene->iterations += 1;
    scene->cameraSamples += (unsigned long long)count * (unsigned long long)nPixels;
    return PATHED_OK;
}

// One internal pass of the hybrid path kernel (k_path_hybrid, path_hybrid.h): as renderPassFused, one persistent launch.
static int renderPassHybrid(PathedScene *scene, uint64_t seed, uint32_t begin, uint32_t count,
                            int start_bounce, int last_bounce, float *d_accum, hipStream_t stream)
{
    const int nPixels = scene->width * scene->height;
    const int chunk = scene->samplesPerUnit;
    const int chunksPerPixel = (int)((count + (uint32_t)chunk - 1) / (uint32_t)chunk);
    const unsigned long long nUnits64 = (unsigned long long)nPixels * (unsigned long long)chunksPerPixel;
    if (nUnits64 >= 0xFFFFFFF0ull) { return fail(PATHED_E_INVALID, "too many work units in one pass"); }
    const unsigned int nUnits = (unsigned int)nUnits64;

    if (scene->chunkCapacity < (size_t)nUnits) {
        HIP_TRY(scene->chunkBuf.allocate((size_t)nUnits));
        scene->chunkCapacity = (size_t)nUnits;
    }
    if (!scene->counters.ptr) { HIP_TRY(scene->counters.allocate(kMaxPools * kCtrCount)); }
    if (!scene->stats.ptr) {
        HIP_TRY(scene->stats.allocate(kStatCount));
        HIP_TRY(hipMemset(scene->stats.ptr, 0, kStatCount * sizeof(unsigned long long)));
    }

    unsigned long long blocks = (unsigned long long)scene->computeUnits * PATHED_HYBRID_WAVES;
    const unsigned long long blocksNeeded = (nUnits64 + (unsigned long long)kBlock - 1) / kBlock;
    if (blocks > blocksNeeded) { blocks = blocksNeeded; }
    if (blocks < 1) { blocks = 1; }
    const unsigned int waves = (unsigned int)blocks * kWavesPerBlock;
    const size_t overflowRows = (size_t)(scene->hybridMaxStack > kHybridStackRows ? scene->hybridMaxStack - kHybridStackRows : 0);
    const size_t overflowInts = (size_t)blocks * kBlock * (overflowRows ? overflowRows : 1);
    if (scene->volumeOverflow.count < overflowInts) { HIP_TRY(scene->volumeOverflow.allocate(overflowInts)); }

    RenderParams params;
    std::memset(&params, 0, sizeof params);
    params.scene = scene->device;
    params.state.chunkBuf = scene->chunkBuf.ptr;
    params.counters = scene->counters.ptr;
    params.stats = scene->stats.ptr;
    params.stackOverflow = scene->volumeOverflow.ptr;
    params.maxStack = scene->hybridMaxStack;
    params.accum = d_accum;
    params.nPixels = nPixels;
    if (!fillUnitOrder(params, scene->unitOrder, scene->width, scene->height, chunksPerPixel, 0, 1,
                       (int)(waves < (unsigned int)kUnitQueues ? waves : (unsigned int)kUnitQueues))) {
        return fail(PATHED_E_INVALID, "too many work units in one pass");
    }
    {
        const unsigned int wavesPerQueue = (waves + (unsigned int)params.nQueues - 1) / (unsigned int)params.nQueues;
        unsigned int grab = (nUnits / (unsigned int)params.nQueues) / (wavesPerQueue * 4u);
        params.unitGrab = (int)(grab < 1u ? 1u : grab > 64u ? 64u : grab);
    }
    params.chunk = chunk;
    params.chunksPerPixel = chunksPerPixel;
    params.seedLo = (uint32_t)seed;
    params.seedHi = (uint32_t)(seed >> 32);
    params.sppBegin = begin;
    params.sppEnd = begin + count;
    params.startBounce = start_bounce;
    params.lastBounce = last_bounce;
    params.smallQuads = scene->hybridLayout.nQuads;
    params.smallKappaT = scene->hybridLayout.kappaT;
    params.scene.leafTris = scene->hybridItemTris.ptr;   // the direct set in item order
    params.hybridNodes = scene->hybridNodes.ptr;
    params.hybridTris = scene->hybridTris.ptr;
    params.hybridNodeCount = scene->hybridNodeCount;
    params.hybridTreeTris = scene->hybridTreeTris;
    params.hybridDirectTris = scene->hybridDirectTris;
    for (int a = 0; a < 3; a++) { params.hybridLo[a] = scene->hybridLo[a]; params.hybridHi[a] = scene->hybridHi[a]; }
    for (int a = 0; a < 4; a++) { params.hybridSphere[a] = scene->hybridSphere[a]; }
    // scheduling of its bursts (PathedSceneOptions.wave_stragglers / wave_refill): results do not depend on them
    params.suspendLanes = scene->options.wave_stragglers != 0 ? (scene->options.wave_stragglers < 0 ? 0 : scene->options.wave_stragglers) : kHybridStragglers;
    params.suspendPatience = scene->options.wave_refill != 0 ? scene->options.wave_refill : kHybridRefill;
    params.hybridBatch = scene->options.hybrid_batch != 0 ? scene->options.hybrid_batch : kHybridBatch;
    params.hybridReady = scene->options.hybrid_ready != 0 ? (scene->options.hybrid_ready < 0 ? 1 : scene->options.hybrid_ready) : kHybridReady;

    HIP_TRY(hipMemsetAsync(params.counters, 0, kCtrCount * sizeof(unsigned int), stream));
    const dim3 grid((unsigned)blocks), block(kBlock);
    const size_t lds = (size_t)scene->device.nMaterials * sizeof(DMaterial);
    int timed = -1;
    if (scene->timeKernels) {
        timed = scene->traceEvents.acquire();
        (void)hipEventRecord(scene->traceEvents.start[timed], stream);
    }
    // the narrowest instantiation whose compile-time scene set contains this scene's (shading.h: SceneTraits)
    if (scene->envOnly && !scene->hasContainers) { hipLaunchKernelGGL((k_path_hybrid<TraitsEnvironmentOnly>), grid, block, lds, stream, params, scene->hybridItems); }
    else if (scene->smoothSet) { hipLaunchKernelGGL((k_path_hybrid<TraitsSmooth>), grid, block, lds, stream, params, scene->hybridItems); }
    else if (scene->triangleLit) { hipLaunchKernelGGL((k_path_hybrid<TraitsTriangleLit>), grid, block, lds, stream, params, scene->hybridItems); }
    else { hipLaunchKernelGGL((k_path_hybrid<TraitsAll>), grid, block, lds, stream, params, scene->hybridItems); }
    if (timed >= 0) { (void)hipEventRecord(scene->traceEvents.stop[timed], stream); }
    scene->traceLaunchesAll++;
    const dim3 pixelGrid((unsigned)((nPixels + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(k_resolve, pixelGrid, block, 0, stream, params);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));

    scene->iterations += 1;
    scene->cameraSamples += (unsigned long long)count * (unsigned long long)nPixels;
    return PATHED_OK;
}

// One internal pass: samples [begin, begin+count), count <= chunk * kMaxChunksPerPass.
// The slot pool is split into `pools` independent halves, each with its own unit range,
// counters and HIP stream: while one half runs its (ALU-bound) trace kernel the other runs its
// (memory-bound) shade kernel, so the two overlap instead of alternating.
static int renderPass(PathedScene *scene, uint64_t seed, uint32_t begin, uint32_t count,
                      int start_bounce, int last_bounce, float *d_accum, hipStream_t stream)
{
    const int nPixels = scene->width * scene->height;
    const int chunk = scene->samplesPerUnit;
    const int chunksPerPixel = (int)((count + (uint32_t)chunk - 1) / (uint32_t)chunk);
    const unsigned long long nUnits64 = (unsigned long long)nPixels * (unsigned long long)chunksPerPixel;
    if (nUnits64 >= 0xFFFFFFF0ull) { return fail(PATHED_E_INVALID, "too many work units in one pass"); }
    const unsigned int nUnits = (unsigned int)nUnits64;

    // small jobs use one pool; otherwise split slots and units evenly
    int pools = scene->pools;
    if (nUnits64 < 4ull * kBlock * (unsigned long long)pools) { pools = 1; }

    unsigned long long wanted = nUnits64 < (unsigned long long)scene->maxSlots ? nUnits64 : (unsigned long long)scene->maxSlots;
    if (scene->adaptiveSlots) {
        // a slot should render at least 16 units in a call, or the ramp-up and the drain of the pool outweigh its size
        // (16 spp at 1024^2: 4 Mi slots 899, 8 Mi 794 Msamples/s on the teapot), but never fewer than 4 Mi
        const unsigned long long floorSlots = 1ull << 22;
        unsigned long long byWork = nUnits64 / 16ull;
        if (byWork < floorSlots) { byWork = floorSlots; }
        if (wanted > byWork) { wanted = byWork; }
    }
    // a block of the staged shade kernel owns stageRounds x 256 slots: pools are whole blocks
    const unsigned long long slotQuantum = (unsigned long long)kBlock * (scene->stagedShade ? scene->stageRounds : 1);
    const int slotsPerPool = (int)((wanted / pools + slotQuantum - 1) / slotQuantum * slotQuantum);
    const int nSlots = slotsPerPool * pools;
    int code = ensureRenderState(scene, nSlots, (size_t)nUnits);
    if (code != PATHED_OK) { return code; }
    if (pools > 1 && !scene->poolStreams[0]) {
        for (int h = 0; h < kMaxPools; h++) {
            HIP_TRY(hipStreamCreateWithFlags(&scene->poolStreams[h], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&scene->poolDone[h], hipEventDisableTiming));
        }
        HIP_TRY(hipEventCreateWithFlags(&scene->callerReady, hipEventDisableTiming));
    }

    const int blocksPerPool = slotsPerPool / kBlock;
    // every unit queue needs a consumer among the blocks of the shade kernel (a staged block owns stageRounds x 256 slots)
    const int shadeBlocks = blocksPerPool / (scene->stagedShade ? scene->stageRounds : 1);
    const int nQueues = shadeBlocks < kUnitQueues ? shadeBlocks : kUnitQueues;

    RenderParams params[kMaxPools];
    hipStream_t streams[kMaxPools] = { stream, stream, stream, stream };
    for (int h = 0; h < pools; h++) {
        RenderParams &q = params[h];
        const size_t slotBase = (size_t)h * slotsPerPool;
        q.scene = scene->device;
        q.state.rayO = scene->rayO.ptr + slotBase;
        q.state.rayD = scene->rayD.ptr + slotBase;
        q.state.hit = scene->hit.ptr + slotBase;
        q.state.mod = scene->mod.ptr + slotBase;
        q.state.thr = scene->thr.ptr + slotBase;
        q.state.res = scene->res.ptr + slotBase;
        q.state.pend = scene->pend.ptr + slotBase;
        q.state.acc = scene->acc.ptr + slotBase;
        q.state.shO = scene->shO.ptr + slotBase;
        q.state.shD = scene->shD.ptr + slotBase;
        q.state.chunkBuf = scene->chunkBuf.ptr;
        if (scene->splitShade) {
            const size_t listEntries = (size_t)scene->listCap * kListShards * 64;
            for (int list = 0; list < 2; list++) {
                q.state.lists[list] = scene->slotLists.ptr + ((size_t)h * 2 + list) * listEntries;
                for (int parity = 0; parity < 2; parity++) {
                    q.state.deferred[list][parity] = scene->deferredLists.ptr + (((size_t)h * 2 + list) * 2 + parity) * (size_t)scene->nSlots;
                }
            }
            q.listCap = scene->listCap;
        }
        q.counters = scene->counters.ptr + (size_t)h * kCtrCount;
        const size_t traceWaves = (size_t)scene->traceGrid * kWavesPerBlock;
        q.suspendLanes = scene->bruteForce ? 0 : scene->suspendLanes;
        q.suspendPatience = scene->suspendPatience;
        q.parkMinCardsPerWave = scene->parkMinCards;
        q.suspendMask = scene->bruteForce ? nullptr : scene->suspendMask.ptr + (size_t)h * traceWaves;
        q.suspendData = scene->bruteForce ? nullptr
            : scene->suspendData.ptr + (size_t)h * traceWaves * (size_t)(kSaveWords + scene->maxStack) * 64;
        {
            const size_t overflowRows = (size_t)(scene->maxStack > scene->stackRows ? scene->maxStack - scene->stackRows : 0);
            q.stackOverflow = scene->bruteForce ? nullptr
                : scene->stackOverflow.ptr + (size_t)h * (size_t)scene->traceGrid * kBlock * (overflowRows ? overflowRows : 1);
            q.maxStack = scene->maxStack;
        }
        q.stats = scene->stats.ptr;
        q.accum = d_accum;
        q.nSlots = slotsPerPool;
        q.nPixels = nPixels;
        if (!fillUnitOrder(q, scene->unitOrder, scene->width, scene->height, chunksPerPixel, h, pools, nQueues)) {
            return fail(PATHED_E_INVALID, "too many work units in one pass");
        }
        q.chunk = chunk;
        q.chunksPerPixel = chunksPerPixel;
        q.seedLo = (uint32_t)seed;
        q.seedHi = (uint32_t)(seed >> 32);
        q.sppBegin = begin;
        q.sppEnd = begin + count;
        q.startBounce = start_bounce;
        q.lastBounce = last_bounce;
        // local rays (per-slot shade kernel only: the staged and split stages of the experiments build do not know them)
        q.localCount = (!scene->stagedShade && !scene->splitShade) ? scene->localCount : 0;
        q.localCounting = scene->countMode ? 1 : 0;
        q.afterTrace = 1;
        if (q.localCount > 0) {
            std::memcpy(q.localTris, scene->localTris, sizeof q.localTris);
            for (int a = 0; a < 3; a++) { q.hybridLo[a] = scene->localLo[a]; q.hybridHi[a] = scene->localHi[a]; }
            for (int a = 0; a < 4; a++) { q.hybridSphere[a] = scene->localSphere[a]; }
        }
        if (pools > 1) { streams[h] = scene->poolStreams[h]; }
    }

    if (pools > 1) {
        // the pool streams start after whatever the caller queued on its stream
        HIP_TRY(hipEventRecord(scene->callerReady, stream));
        for (int h = 0; h < pools; h++) { HIP_TRY(hipStreamWaitEvent(streams[h], scene->callerReady, 0)); }
    }

    const dim3 slotGrid((unsigned)blocksPerPool), block(kBlock);
    for (int h = 0; h < pools; h++) {
        HIP_TRY(hipMemsetAsync(params[h].counters, 0, kCtrCount * sizeof(unsigned int), streams[h]));
        if (params[h].suspendMask) {
            HIP_TRY(hipMemsetAsync(params[h].suspendMask, 0, (size_t)scene->traceGrid * kWavesPerBlock * sizeof(unsigned long long), streams[h]));
        }
        hipLaunchKernelGGL(k_init, slotGrid, block, 0, streams[h], params[h]);
    }

    // Iterate until every slot of every pool has run out of units.  `remaining` is polled with
    // a lag of one chunk of launches so the GPU never waits for the host.
    const int launchChunk = 8;
    const int ringSize = 32;
    // the poll events are destroyed and the pool streams drained on every way out, error or not
    struct PollGuard {
        hipEvent_t events[kMaxPools][2] = {};
        hipStream_t *streams = nullptr;
        int pools = 0;
        ~PollGuard()
        {
            for (int h = 0; h < pools; h++) { (void)hipStreamSynchronize(streams[h]); }
            for (int h = 0; h < kMaxPools; h++) {
                for (int k = 0; k < 2; k++) { if (events[h][k]) { (void)hipEventDestroy(events[h][k]); } }
            }
        }
    } guard;
    guard.streams = streams;
    guard.pools = pools;
    hipEvent_t (&pollEvents)[kMaxPools][2] = guard.events;
    for (int h = 0; h < pools; h++) {
        HIP_TRY(hipEventCreateWithFlags(&pollEvents[h][0], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&pollEvents[h][1], hipEventDisableTiming));
    }
    const int shadeLaunches = params[0].localCount > 0 ? (scene->options.shade_launches > 0 ? scene->options.shade_launches : kDefaultShadeLaunches) : 1;
    unsigned long long iteration = 0;
    int pollIndex = 0;
    bool havePending = false;
    bool poolDone[kMaxPools];
    for (int h = 0; h < kMaxPools; h++) { poolDone[h] = h >= pools; }
    while (!(poolDone[0] && poolDone[1] && poolDone[2] && poolDone[3])) {
        for (int k = 0; k < launchChunk; k++) {
            for (int h = 0; h < pools; h++) {
                if (poolDone[h]) { continue; }
                params[h].parity = (int)(iteration & 1ull);
                scene->traceLaunchesAll++;
#if defined(PATHED_EXPERIMENTS) && PATHED_EXPERIMENTS
                if (h == 0) {
                    if (const char *text = tuningEnv("PATHED_SORT_PROBE")) {
                        if (iteration == (unsigned long long)atoll(text)) { sortProbe(scene, params[h], streams[h]); }
                    }
                }
#endif
                if (scene->timeKernels && iteration % (unsigned long long)scene->timeInterval == 0ull) {
                    const int e = scene->traceEvents.acquire();
                    (void)hipEventRecord(scene->traceEvents.start[e], streams[h]);
                    launchTrace(scene, params[h], streams[h]);
                    (void)hipEventRecord(scene->traceEvents.stop[e], streams[h]);
                    const int s = scene->shadeEvents.acquire();
                    (void)hipEventRecord(scene->shadeEvents.start[s], streams[h]);
                    for (int round = 0; round < shadeLaunches; round++) {
                        params[h].afterTrace = round == 0 ? 1 : 0;
                        launchShade(scene, params[h], streams[h]);
                    }
                    (void)hipEventRecord(scene->shadeEvents.stop[s], streams[h]);
                } else {
                    launchTrace(scene, params[h], streams[h]);
                    // with local rays: several shade launches per trace launch -- the slots whose rays were all local advance
                    // another vertex each time, the others wait; the trace kernel then finds the hard rays of several vertices
                    // in ONE launch
                    for (int round = 0; round < shadeLaunches; round++) {
                        params[h].afterTrace = round == 0 ? 1 : 0;
                        launchShade(scene, params[h], streams[h]);
                    }
                }
            }
            iteration++;
        }
        const int slotIndex = pollIndex % ringSize;
        for (int h = 0; h < pools; h++) {
            if (poolDone[h]) { continue; }
            HIP_TRY(hipMemcpyAsync(scene->hostRemaining + h * ringSize + slotIndex, params[h].counters + kCtrRemaining,
                                   sizeof(unsigned int), hipMemcpyDeviceToHost, streams[h]));
            HIP_TRY(hipEventRecord(pollEvents[h][pollIndex & 1], streams[h]));
        }
        if (havePending) {
            const int previous = pollIndex - 1;
            for (int h = 0; h < pools; h++) {
                if (poolDone[h]) { continue; }
                HIP_TRY(hipEventSynchronize(pollEvents[h][previous & 1]));
                if (scene->hostRemaining[h * ringSize + previous % ringSize] == 0) { poolDone[h] = true; }
            }
        }
        havePending = true;
        pollIndex++;
    }
    if (pools > 1) {
        for (int h = 0; h < pools; h++) {
            HIP_TRY(hipEventRecord(scene->poolDone[h], streams[h]));
            HIP_TRY(hipStreamWaitEvent(stream, scene->poolDone[h], 0));
        }
    }
    const dim3 pixelGrid((unsigned)((nPixels + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(k_resolve, pixelGrid, block, 0, stream, params[0]);
    for (int h = 0; h < pools; h++) { HIP_TRY(hipStreamSynchronize(streams[h])); }
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipGetLastError());

    scene->iterations += iteration;
    scene->cameraSamples += (unsigned long long)count * (unsigned long long)nPixels;
    return PATHED_OK;
}

int pathed_hip_render_device(PathedScene *scene, uint64_t seed,
                             uint32_t spp_begin, uint32_t spp_count,
                             int start_bounce, int last_bounce,
                             float *d_accum_rgb_sum, void *stream_handle, int blocking)
{
    (void)blocking;  // the iteration loop polls a device counter, so the call always completes
    if (!scene || !d_accum_rgb_sum) { return fail(PATHED_E_INVALID, "null scene or accumulation buffer"); }
    if (start_bounce < 0 || (last_bounce != -1 && start_bounce > last_bounce)) {
        return fail(PATHED_E_INVALID, "bounce window: need 0 <= startBounce <= lastBounce (or lastBounce == -1)");
    }
    if (spp_count == 0) { return PATHED_OK; }
    if ((uint64_t)spp_begin + spp_count > 0x7fffffffull) { return fail(PATHED_E_INVALID, "sample index overflow"); }
    if (scene->hasContainers && scene->integrator != PATHED_INTEGRATOR_VOLUME_PATH_TRACER) {
        return fail(PATHED_E_UNSUPPORTED, "the scene has passthrough (medium container) surfaces: select the VolumePathTracer integrator (pathed_hip_set_integrator)");
    }
    SELECT_DEVICE(scene);

    if (scene->timeKernels) {
        HIP_TRY(scene->traceEvents.create());
        HIP_TRY(scene->shadeEvents.create());
    }
    hipStream_t stream = (hipStream_t)stream_handle;

    // passes of at most chunk * kMaxChunksPerPass samples keep the partial-sum buffer bounded
    int chunksPerPass = kMaxChunksPerPass;
    {
        const unsigned long long pixels = (unsigned long long)scene->width * (unsigned long long)scene->height;
        const unsigned long long cap = (1ull << 30) / (pixels ? pixels : 1ull);   // 2^30 partial sums = 17 GB
        if ((unsigned long long)chunksPerPass > cap) { chunksPerPass = cap >= 1ull ? (int)cap : 1; }
    }
    if (scene->options.chunks_per_pass > 0) { chunksPerPass = scene->options.chunks_per_pass; }
    if (const char *text = tuningEnv("PATHED_CHUNKS_PER_PASS")) {   // tuning: fewer, longer passes at the cost of a larger partial-sum buffer
        const int value = atoi(text);
        if (value >= 1 && value <= 4096) { chunksPerPass = value; }
    }
    const uint32_t perPass = (uint32_t)scene->samplesPerUnit * (uint32_t)chunksPerPass;
    // BVH scenes: short calls on the wave path kernel, long ones on the wavefront (the counting instantiations are the wavefront's)
    const unsigned long long callSamples = (unsigned long long)scene->width * (unsigned long long)scene->height * (unsigned long long)spp_count;
    const bool wavePath = scene->waveAvailable && !usesVolumeKernel(scene) && !scene->countMode
        && (scene->waveMode == 2 || (scene->waveMode == 0 && (callSamples < scene->waveMaxSamples || scene->sceneInLds)));
    // (trees small enough for the trace kernel's LDS copy -- the Cornell box with its two meshes: k_path_wave, whose node reads
    // hit L1 / L2, is level with the wavefront or 11 % ahead at every length, profiles/r4_ab_wave.log)
    // [r5] ... and scenes of 65 .. 4096 triangles on the hybrid kernel at every length (counting is the wavefront's)
    const bool hybridPath = scene->hybridPath && !usesVolumeKernel(scene) && !scene->countMode;
    scene->lastCallWave = wavePath && !hybridPath;
    scene->lastCallHybrid = hybridPath;
    uint32_t done = 0;
    while (done < spp_count) {
        const uint32_t count = (spp_count - done < perPass) ? (spp_count - done) : perPass;
        const int code = usesVolumeKernel(scene)
            ? renderPassVolume(scene, seed, spp_begin + done, count, start_bounce, last_bounce, d_accum_rgb_sum, stream)
            : scene->fusedPath
                ? renderPassFused(scene, seed, spp_begin + done, count, start_bounce, last_bounce, d_accum_rgb_sum, stream)
            : hybridPath
                ? renderPassHybrid(scene, seed, spp_begin + done, count, start_bounce, last_bounce, d_accum_rgb_sum, stream)
            : wavePath
                ? renderPassWave(scene, seed, spp_begin + done, count, start_bounce, last_bounce, d_accum_rgb_sum, stream)
                : renderPass(scene, seed, spp_begin + done, count, start_bounce, last_bounce, d_accum_rgb_sum, stream);
        if (code != PATHED_OK) { return code; }
        done += count;
    }
    if (scene->timeKernels) {
        scene->traceEvents.harvestAll();
        scene->shadeEvents.harvestAll();
    }
    return PATHED_OK;
}

int pathed_hip_scene_set_camera(PathedScene *scene, const PathedCamera *camera)
{
    if (!scene || !camera) { return fail(PATHED_E_INVALID, "null scene or camera"); }
    if (camera->width != scene->width || camera->height != scene->height) {
        return fail(PATHED_E_INVALID, "the new camera must keep the scene's resolution (the radiance sums are per pixel)");
    }
    buildCamera(*camera, &scene->device.camera);
    SELECT_DEVICE(scene);
    HIP_TRY(rebuildSmallItems(scene));   // tiny scenes: the phase-1 tolerances depend on how far away a ray may start
    HIP_TRY(rebuildHybridItems(scene));  // ... and so do the hybrid kernel's direct set's
    return PATHED_OK;
}

int pathed_hip_set_integrator(PathedScene *scene, int integrator)
{
    if (!scene) { return fail(PATHED_E_INVALID, "null scene"); }
    if (integrator != PATHED_INTEGRATOR_PATH_TRACER && integrator != PATHED_INTEGRATOR_VOLUME_PATH_TRACER) {
        return fail(PATHED_E_INVALID, "unknown integrator");
    }
    scene->integrator = integrator;
    return PATHED_OK;
}

int pathed_hip_set_samples_per_unit(PathedScene *scene, int samples)
{
    if (!scene) { return fail(PATHED_E_INVALID, "null scene"); }
    if (samples < 1 || samples > kMaxChunk) { return fail(PATHED_E_INVALID, "samples per unit must be in [1, 128]"); }
    scene->samplesPerUnit = samples;
    return PATHED_OK;
}

int pathed_hip_render(PathedScene *scene, uint64_t seed,
                      uint32_t spp_begin, uint32_t spp_count,
                      int start_bounce, int last_bounce,
                      float *accum_rgb_sum)
{
    if (!scene || !accum_rgb_sum) { return fail(PATHED_E_INVALID, "null scene or accumulation buffer"); }
    SELECT_DEVICE(scene);
    const size_t count = (size_t)3 * scene->width * scene->height;
    float *deviceAccum = nullptr;
    HIP_TRY(hipMalloc((void **)&deviceAccum, count * sizeof(float)));
    hipError_t status = hipMemset(deviceAccum, 0, count * sizeof(float));
    if (status != hipSuccess) { (void)hipFree(deviceAccum); return fail(PATHED_E_DEVICE, hipGetErrorString(status)); }

    const int code = pathed_hip_render_device(scene, seed, spp_begin, spp_count, start_bounce, last_bounce, deviceAccum, nullptr, 1);
    if (code != PATHED_OK) { (void)hipFree(deviceAccum); return code; }

    std::vector<float> host(count);
    status = hipMemcpy(host.data(), deviceAccum, count * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(deviceAccum);
    if (status != hipSuccess) { return fail(PATHED_E_DEVICE, hipGetErrorString(status)); }
    for (size_t i = 0; i < count; i++) { accum_rgb_sum[i] += host[i]; }
    return PATHED_OK;
}

int pathed_hip_trace(PathedScene *scene, const float *rays, size_t n, int any_hit, void *hits)
{
    if (!scene) { return fail(PATHED_E_INVALID, "null scene"); }
    if (n == 0) { return PATHED_OK; }
    if (!rays || !hits) { return fail(PATHED_E_INVALID, "null ray or hit buffer"); }
    if (n > (size_t)1 << 28) { return fail(PATHED_E_INVALID, "too many rays in one call"); }
    SELECT_DEVICE(scene);

    float4 *deviceRays = nullptr;
    float4 *deviceHits = nullptr;
    int *deviceOccluded = nullptr;
    int *deviceOverflow = nullptr;
    HIP_TRY(hipMalloc((void **)&deviceRays, n * 2 * sizeof(float4)));
    hipError_t status = hipMemcpy(deviceRays, rays, n * 8 * sizeof(float), hipMemcpyHostToDevice);
    if (status == hipSuccess) {
        status = any_hit ? hipMalloc((void **)&deviceOccluded, n * sizeof(int)) : hipMalloc((void **)&deviceHits, n * sizeof(float4));
    }
    if (status == hipSuccess) {
        const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
        const size_t lds = (size_t)(scene->stackRows + 1) * kBlock * sizeof(int);
        const size_t overflowRows = (size_t)(scene->maxStack > scene->stackRows ? scene->maxStack - scene->stackRows : 0);
        status = hipMalloc((void **)&deviceOverflow, (size_t)grid.x * kBlock * (overflowRows ? overflowRows : 1) * sizeof(int));
        if (status == hipSuccess) {
            #define PATHED_HOOK(STACK, FORMAT) hipLaunchKernelGGL((k_trace_rays<STACK, FORMAT>), grid, block, lds, 0, scene->device, deviceRays, (int)n, any_hit, deviceHits, deviceOccluded, deviceOverflow, scene->maxStack)
            #define PATHED_HOOK_ROWS(FORMAT) switch (scene->stackRows) { \
                case 8: PATHED_HOOK(8, FORMAT); break; \
                case 16: PATHED_HOOK(16, FORMAT); break; \
                default: PATHED_HOOK(22, FORMAT); break; }
#if PATHED_EXPERIMENTS
            if (scene->nodeFormat == 2) { PATHED_HOOK_ROWS(2) }
            else if (scene->nodeFormat == 1) { PATHED_HOOK_ROWS(1) }
            else
#endif
            { PATHED_HOOK_ROWS(0) }
            #undef PATHED_HOOK_ROWS
            #undef PATHED_HOOK
            status = hipGetLastError();
            if (status == hipSuccess) { status = hipDeviceSynchronize(); }
        }
    }
    if (status == hipSuccess) {
        status = any_hit
            ? hipMemcpy(hits, deviceOccluded, n * sizeof(int), hipMemcpyDeviceToHost)
            : hipMemcpy(hits, deviceHits, n * sizeof(float4), hipMemcpyDeviceToHost);
    }
    (void)hipFree(deviceRays);
    if (deviceHits) { (void)hipFree(deviceHits); }
    if (deviceOccluded) { (void)hipFree(deviceOccluded); }
    if (deviceOverflow) { (void)hipFree(deviceOverflow); }
    if (status != hipSuccess) { return fail(PATHED_E_DEVICE, hipGetErrorString(status)); }
    return PATHED_OK;
}

int pathed_hip_has_experiments(void) { return PATHED_EXPERIMENTS ? 1 : 0; }

int pathed_hip_debug_small_candidates(PathedScene *scene, const float *rays, size_t n, uint64_t *out)
{
    if (!scene) { return fail(PATHED_E_INVALID, "null scene"); }
    if (!scene->bruteForce || scene->device.nTris < 1) { return fail(PATHED_E_UNSUPPORTED, "the all-triangles intersector serves scenes of 1..64 triangles"); }
    if (n == 0) { return PATHED_OK; }
    if (!rays || !out) { return fail(PATHED_E_INVALID, "null ray or output buffer"); }
    if (n > (size_t)1 << 24) { return fail(PATHED_E_INVALID, "too many rays in one call"); }
    SELECT_DEVICE(scene);
    // the matrix-pipe rows over the LEAF order (the render's own table, if any, follows the item order)
    DeviceBuffer<float> table;
    MfmaFrame frame;
    std::memset(&frame, 0, sizeof frame);
#if PATHED_EXPERIMENTS
    {
        std::vector<float> rows;
        buildMfmaTable(scene->bvh.leafTris.data(), scene->device.nTris, scene->device.camera.origin, 1, &rows, &frame);
        HIP_TRY(table.upload(rows));
    }
#endif
    const size_t padded = (n + kBlock - 1) / kBlock * kBlock;   // whole blocks: every lane of a wave issues the matrix instructions
    std::vector<float> hostRays(padded * 10, 0.f);
    std::memcpy(hostRays.data(), rays, n * 10 * sizeof(float));
    for (size_t i = n; i < padded; i++) { hostRays[10 * i + 5] = 1.f; hostRays[10 * i + 8] = 1.f; }
    DeviceBuffer<float> deviceRays;
    DeviceBuffer<unsigned long long> deviceOut;
    HIP_TRY(deviceRays.upload(hostRays));
    HIP_TRY(deviceOut.allocate(n * 8));
    hipLaunchKernelGGL(k_debug_small_candidates, dim3((unsigned)(padded / kBlock)), dim3(kBlock), 0, nullptr, scene->device, scene->smallTris,
                       scene->smallItems, scene->smallLayout.nQuads, scene->smallLayout.kappaT, scene->itemTris.ptr, table.ptr, frame,
                       deviceRays.ptr, (int)n, deviceOut.ptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, deviceOut.ptr, n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return PATHED_OK;
}

int pathed_hip_scene_refit(PathedScene *scene, const float *positions, const float *normals, uint32_t n_vertices, float *device_ms)
{
    if (!scene || !positions) { return fail(PATHED_E_INVALID, "null scene or positions"); }
    if (!scene->refittable) { return fail(PATHED_E_INVALID, "create the scene with PathedSceneOptions.refittable = 1: the refit needs the triangle soup on the device"); }
    if ((size_t)n_vertices != scene->soupVertices) { return fail(PATHED_E_INVALID, "refit keeps the topology: the vertex count must be the scene's"); }
    if (scene->bruteForce) { return fail(PATHED_E_UNSUPPORTED, "scenes of at most 64 triangles carry no tree to refit: create the scene again (microseconds)"); }
    if (scene->nodeFormat != 0) { return fail(PATHED_E_UNSUPPORTED, "refit serves the 128-byte float nodes"); }
    SELECT_DEVICE(scene);
    const int nNodes = scene->device.nNodes;
    const size_t nTris = (size_t)scene->device.nTris;
    if (nNodes <= 0 || nTris == 0) { return fail(PATHED_E_INVALID, "the scene has no tree"); }
    if (!scene->refitLo.ptr) {
        HIP_TRY(scene->refitLo.allocate((size_t)nNodes));
        HIP_TRY(scene->refitHi.allocate((size_t)nNodes));
        HIP_TRY(scene->refitReady.allocate(2 * (size_t)nNodes));
    }
    HIP_TRY(hipMemcpy(scene->soupPositions.ptr, positions, 3 * (size_t)n_vertices * sizeof(float), hipMemcpyHostToDevice));
    if (normals) { HIP_TRY(hipMemcpy(scene->soupNormals.ptr, normals, 3 * (size_t)n_vertices * sizeof(float), hipMemcpyHostToDevice)); }
    hipEvent_t start = nullptr, stop = nullptr;
    HIP_TRY(hipEventCreate(&start));
    hipError_t status = hipEventCreate(&stop);
    if (status == hipSuccess) { status = hipEventRecord(start, nullptr); }
    if (status == hipSuccess) {
        // shading records (corners, normals, uvs) of every primitive, then the tree bottom-up
        hipLaunchKernelGGL(k_build_tri_shade, dim3((unsigned)((8 * nTris + kBlock - 1) / kBlock)), dim3(kBlock), 0, nullptr,
                           scene->soupPositions.ptr, scene->soupNormals.ptr, scene->soupUvs.ptr, scene->soupIndices.ptr, scene->soupTriMaterial.ptr,
                           (uint32_t)nTris, scene->triShade.ptr, scene->triCompact.ptr);
        status = hipMemsetAsync(scene->refitReady.ptr, 0, 2 * (size_t)nNodes, nullptr);
    }
    int passes = 0;
    unsigned char rootReady = 0;
    const dim3 grid((unsigned)((nNodes + kBlock - 1) / kBlock)), block(kBlock);
    auto runPasses = [&](int count) {
        for (int k = 0; k < count; k++, passes++) {
            unsigned char *in = scene->refitReady.ptr + (size_t)(passes & 1) * nNodes, *out = scene->refitReady.ptr + (size_t)((passes + 1) & 1) * nNodes;
            hipLaunchKernelGGL(k_refit_pass, grid, block, 0, nullptr, scene->nodes.ptr, nNodes, scene->leafTris.ptr, scene->soupPositions.ptr,
                               scene->soupIndices.ptr, scene->spheres.ptr, scene->refitLo.ptr, scene->refitHi.ptr, in, out);
        }
    };
    // one pass per 4-wide level, back to back (no look at the flags in between: a host round trip costs more than a pass)
    if (status == hipSuccess) { runPasses(scene->bvh.maxDepth > 0 ? scene->bvh.maxDepth + 1 : 16); status = hipGetLastError(); }
    float ms = 0.f;
    if (status == hipSuccess) { status = hipEventRecord(stop, nullptr); }
    if (status == hipSuccess) { status = hipEventSynchronize(stop); }
    if (status == hipSuccess) { status = hipEventElapsedTime(&ms, start, stop); }
    if (status == hipSuccess) { status = hipMemcpy(&rootReady, scene->refitReady.ptr + (size_t)(passes & 1) * nNodes, 1, hipMemcpyDeviceToHost); }
    while (status == hipSuccess && !rootReady && passes < 4096) {
        // (a tree deeper than its recorded depth: keep going, four passes per look at the root's flag; their device time
        // joins device_ms)
        float extra = 0.f;
        status = hipEventRecord(start, nullptr);
        if (status == hipSuccess) { runPasses(4); status = hipGetLastError(); }
        if (status == hipSuccess) { status = hipEventRecord(stop, nullptr); }
        if (status == hipSuccess) { status = hipEventSynchronize(stop); }
        if (status == hipSuccess) { status = hipEventElapsedTime(&extra, start, stop); }
        ms += extra;
        if (status == hipSuccess) { status = hipMemcpy(&rootReady, scene->refitReady.ptr + (size_t)(passes & 1) * nNodes, 1, hipMemcpyDeviceToHost); }
    }
    (void)hipEventDestroy(start);
    if (stop) { (void)hipEventDestroy(stop); }
    if (status != hipSuccess) { return fail(PATHED_E_DEVICE, std::string("refit: ") + hipGetErrorString(status)); }
    if (!rootReady) { return fail(PATHED_E_DEVICE, "refit: the root was never reached (a cycle in the tree?)"); }
    scene->bvhOnHost = false;   // exports download the refitted tree
    if (device_ms) { *device_ms = ms; }
    return PATHED_OK;
}

int pathed_hip_set_stats_mode(PathedScene *scene, int enabled)
{
    if (!scene) { return fail(PATHED_E_INVALID, "null scene"); }
    scene->countMode = (enabled & 1) != 0;   // bit 0: count child boxes / triangles tested
    scene->timeKernels = (enabled & (2 | 4)) != 0; // bit 1: HIP-event timing of every trace / shade launch
    scene->timeInterval = (enabled & 4) ? 8 : 1;   // bit 2: ... of every 8th launch only (the event pairs cost ~6 % of the rate)
    return PATHED_OK;
}

int pathed_hip_reset_stats(PathedScene *scene)
{
    if (!scene) { return fail(PATHED_E_INVALID, "null scene"); }
    SELECT_DEVICE(scene);
    if (scene->stats.ptr) { HIP_TRY(hipMemset(scene->stats.ptr, 0, kStatCount * sizeof(unsigned long long))); }
    scene->iterations = 0;
    scene->traceLaunchesAll = 0;
    scene->cameraSamples = 0;
    scene->traceEvents.totalMs = 0.0;
    scene->traceEvents.launches = 0;
    scene->shadeEvents.totalMs = 0.0;
    scene->shadeEvents.launches = 0;
    return PATHED_OK;
}

int pathed_hip_get_stats(PathedScene *scene, PathedStats *out)
{
    if (!scene || !out) { return fail(PATHED_E_INVALID, "null argument"); }
    SELECT_DEVICE(scene);
    std::memset(out, 0, sizeof *out);
    unsigned long long device[kStatCount] = { 0 };
    if (scene->stats.ptr) {
        HIP_TRY(hipMemcpy(device, scene->stats.ptr, sizeof device, hipMemcpyDeviceToHost));
    }
    out->camera_samples = scene->cameraSamples;
    out->closest_rays = device[kStatClosest];
    out->shadow_rays = device[kStatShadow];
    out->nodes_visited = device[kStatBoxes];
    out->tris_tested = device[kStatTris];
    out->dropped_samples = device[kStatDropped];
    out->local_closest_rays = device[kStatLocalClosest];
    out->local_shadow_rays = device[kStatLocalShadow];
    out->iterations = scene->iterations;
    out->trace_ms = scene->traceEvents.totalMs;
    out->shade_ms = scene->shadeEvents.totalMs;
    out->trace_launches = scene->traceEvents.launches;
    out->bvh_nodes = (uint64_t)scene->bvh.nodeCount;
    out->bvh_bytes = (uint64_t)scene->bvh.nodeCount * 128 + (uint64_t)scene->device.nTris * 48;
    out->bvh_max_depth = (uint32_t)scene->bvh.maxDepth;
    out->scene_in_lds = scene->bruteForce ? 2u : (scene->sceneInLds ? 1u : 0u);
    out->max_boxes_per_ray = device[kStatMaxBoxes];
    out->parked_rays = device[kStatParked];
    out->bvh_build_ms = scene->bvhBuildMs;
    out->bvh_builder = (uint32_t)scene->bvhBuilder;
    out->trace_launches_all = (uint32_t)scene->traceLaunchesAll;
    out->path_kernel = usesVolumeKernel(scene) ? 4u : scene->fusedPath ? 3u : scene->lastCallHybrid ? 7u : scene->lastCallWave ? 6u : scene->splitShade ? 5u : (scene->stagedShade ? 2u : 1u);
    if (tuningEnv("PATHED_SHADE_PROFILE")) {   // counters exist in -DPATHED_SHADE_PROFILE builds only
        static const char *regions[11] = { "all waves", "active slots", "makeIsect (hit)", "camera-ray vertex", "finish previous MIS term",
                                           "new vertex: BSDF sample", "light sampling", "sample finished", "startSample (regeneration)", "shadow ray pushed",
                                           "BSDF sample with black throughput" };
        for (int r = 0; r < 11; r++) {
            const unsigned long long waves = device[kStatShadeProfile + 2 * r], lanes = device[kStatShadeProfile + 2 * r + 1];
            fprintf(stderr, "[pathed] k_shade region %-28s waves %12llu  (%.3f of all)  lanes per wave %.1f\n", regions[r], waves,
                    device[kStatShadeProfile] ? (double)waves / (double)device[kStatShadeProfile] : 0.0, waves ? (double)lanes / (double)waves : 0.0);
        }
    }
    if (tuningEnv("PATHED_FUSED_PROFILE")) {   // the same counters in k_path_small (-DPATHED_SHADE_PROFILE builds)
        static const char *regions[9] = { "iterations (live lanes)", "camera ray", "passes with shadow rays (lanes with one)", "makeIsect (hit)",
                                          "camera-ray vertex", "BSDF sample met an emitter: lightsPDF", "new vertex: BSDF sample", "light sampling",
                                          "sample finished" };
        for (int r = 0; r < 9; r++) {
            const unsigned long long waves = device[kStatShadeProfile + 2 * r], lanes = device[kStatShadeProfile + 2 * r + 1];
            fprintf(stderr, "[pathed] k_path_small %-44s waves %12llu  (%.3f of the iterations)  lanes per wave %.1f\n", regions[r], waves,
                    device[kStatShadeProfile] ? (double)waves / (double)device[kStatShadeProfile] : 0.0, waves ? (double)lanes / (double)waves : 0.0);
        }
        static const char *loops[2] = { "resolve loop of the path's ray", "resolve loop of the shadow ray" };
        for (int r = 0; r < 2; r++) {
            const unsigned long long turns = device[kStatShadeProfile + 2 * (9 + r)], candidates = device[kStatShadeProfile + 2 * (9 + r) + 1];
            fprintf(stderr, "[pathed] k_path_small %-44s %.2f turns per iteration, %.2f candidates per turn (of 64 lanes)\n", loops[r],
                    device[kStatShadeProfile] ? (double)turns / (double)device[kStatShadeProfile] : 0.0, turns ? (double)candidates / (double)turns : 0.0);
        }
    }
    if (tuningEnv("PATHED_WAVE_PROFILE")) {   // k_path_wave (-DPATHED_SHADE_PROFILE builds, tools/wave_profile.py)
        const unsigned long long *v = device + kStatShadeProfile;
        const double iterations = v[0] ? (double)v[0] : 1.0, cycles = v[8] ? (double)v[8] : 1.0;
        fprintf(stderr, "[pathed] k_path_wave: %llu waves, %.0f iterations each, %.1f live paths per iteration\n", v[11], iterations / (double)(v[11] ? v[11] : 1), (double)v[1] / iterations);
        fprintf(stderr, "[pathed] k_path_wave traversal: %.1f rays posted per iteration, %.1f steps per iteration at %.1f of 64 lanes, %.1f rays left in flight at the end of a burst\n",
                (double)v[6] / iterations, (double)v[2] / iterations, v[2] ? (double)v[3] / (double)v[2] : 0.0, (double)v[7] / iterations);
        fprintf(stderr, "[pathed] k_path_wave shade: %.3f of the iterations shade, %.1f paths each\n", (double)v[4] / iterations, v[4] ? (double)v[5] / (double)v[4] : 0.0);
        fprintf(stderr, "[pathed] k_path_wave wave cycles: traversal bursts %.3f, shade + post + regeneration %.3f of the waves' lifetimes\n", (double)v[9] / cycles, (double)v[10] / cycles);
    }
#ifdef PATHED_SHADE_PROFILE
    if (scene->lastCallHybrid) {   // k_path_hybrid (tools/hybrid_profile.py)
        const unsigned long long *v = device + kStatShadeProfile;
        const double iterations = v[0] ? (double)v[0] : 1.0, cycles = v[8] ? (double)v[8] : 1.0, bursts = v[3] ? (double)v[3] : 1.0;
        fprintf(stderr, "[pathed] k_path_hybrid: %llu waves, %.0f iterations each, %.1f live paths per iteration, %.1f of them at a vertex\n", v[12], iterations / (double)(v[12] ? v[12] : 1),
                (double)v[1] / iterations, (double)v[13] / iterations);
        fprintf(stderr, "[pathed] k_path_hybrid tree part: %.3f of the iterations run a burst, %.1f rays posted per burst, %.1f steps per burst (%.1f of them triangle phases) at %.1f of 64 lanes, %.2f refill rounds, %.1f rays parked at its end\n",
                (double)v[3] / iterations, (double)v[2] / bursts, (double)v[4] / bursts, (double)v[7] / bursts, v[4] ? (double)v[5] / (double)v[4] : 0.0, (double)v[6] / bursts, (double)v[14] / bursts);
        fprintf(stderr, "[pathed] k_path_hybrid wave cycles: direct pass + resolve %.3f, proxy + burst %.3f, vertex + regeneration %.3f of the waves' lifetimes\n",
                (double)v[9] / cycles, (double)v[10] / cycles, (double)v[11] / cycles);
    }
#endif
    if (tuningEnv("PATHED_VOLUME_PROFILE")) {   // the same counters in k_path_volume (-DPATHED_SHADE_PROFILE builds)
        static const char *regions[9] = { "samples", "camera-ray query", "bounce-loop iterations", "segment query (no direct lighting before)",
                                          "medium event: occlusion query", "direct lighting at a vertex", "light sample: occlusion query",
                                          "BSDF sample: closest query", "seen through a container: query" };
        for (int r = 0; r < 9; r++) {
            const unsigned long long waves = device[kStatShadeProfile + 2 * r], lanes = device[kStatShadeProfile + 2 * r + 1];
            fprintf(stderr, "[pathed] k_path_volume %-44s waves %12llu  (%.3f per sample-wave)  lanes per wave %.1f\n", regions[r], waves,
                    device[kStatShadeProfile] ? (double)waves / (double)device[kStatShadeProfile] : 0.0, waves ? (double)lanes / (double)waves : 0.0);
        }
    }
    if (getenv("PATHED_DEBUG_STATS")) {
        fprintf(stderr, "[pathed] wave steps %llu lane steps %llu (lane utilisation %.3f) refill rounds %llu\n",
                device[kStatWaveSteps], device[kStatLaneSteps],
                device[kStatWaveSteps] ? (double)device[kStatLaneSteps] / (64.0 * (double)device[kStatWaveSteps]) : 0.0,
                device[kStatRefills]);
        fprintf(stderr, "[pathed] sum of wave lifetimes %.3e cycles, longest single wave %.3e cycles\n",
                (double)device[kStatWaveCycles], (double)device[kStatWaveCyclesMax]);
        fprintf(stderr, "[pathed] wave cycles: refill %.3e, inner phases %.3e (%llu steps), triangle phases %.3e\n",
                (double)device[kStatRefillCycles], (double)device[kStatInnerCycles], device[kStatInnerSteps], (double)device[kStatLeafCycles]);
        fprintf(stderr, "[pathed] tail (no cards left): %llu wave steps, %llu lane steps, %.3e cycles; %llu rays parked\n",
                device[kStatTailSteps], device[kStatTailLaneSteps], (double)device[kStatTailCycles], device[kStatParked]);
    }
    return PATHED_OK;
}

int pathed_hip_scene_export_bvh(PathedScene *scene, float *nodes, size_t *n_nodes, float *tris, size_t *n_tris)
{
    if (!scene || !n_nodes || !n_tris) { return fail(PATHED_E_INVALID, "null argument"); }
    SELECT_DEVICE(scene);
    if (!scene->bvhOnHost) {
        // a device-built tree: fetch it once
        scene->bvh.nodes.resize((size_t)scene->bvh.nodeCount * kNodeFloats);
        scene->bvh.leafTris.resize((size_t)scene->device.nTris * 12);
        HIP_TRY(hipMemcpy(scene->bvh.nodes.data(), scene->nodes.ptr, scene->bvh.nodes.size() * sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(scene->bvh.leafTris.data(), scene->leafTris.ptr, scene->bvh.leafTris.size() * sizeof(float), hipMemcpyDeviceToHost));
        scene->bvhOnHost = true;
    }
    const size_t nodeCount = (size_t)scene->bvh.nodeCount;
    const size_t triCount = scene->bvh.leafTris.size() / 12;
    if (nodes) {
        if (*n_nodes < nodeCount) { return fail(PATHED_E_INVALID, "node buffer too small"); }
        std::memcpy(nodes, scene->bvh.nodes.data(), nodeCount * kNodeFloats * sizeof(float));
    }
    if (tris) {
        if (*n_tris < triCount) { return fail(PATHED_E_INVALID, "triangle buffer too small"); }
        std::memcpy(tris, scene->bvh.leafTris.data(), triCount * 12 * sizeof(float));
    }
    *n_nodes = nodeCount;
    *n_tris = triCount;
    return PATHED_OK;
}

int pathed_hip_scene_export_compressed_nodes(PathedScene *scene, uint32_t *nodes, size_t *n_nodes, size_t *words_per_node)
{
    if (!scene || !n_nodes) { return fail(PATHED_E_INVALID, "null argument"); }
    SELECT_DEVICE(scene);
    const size_t nodeCount = scene->nodeFormat != 0 ? (size_t)scene->bvh.nodeCount : 0;
    const size_t words = scene->nodeFormat == 2 ? 32 : 16;
    if (nodes && nodeCount > 0) {
        if (*n_nodes < nodeCount) { return fail(PATHED_E_INVALID, "node buffer too small"); }
        HIP_TRY(hipMemcpy(nodes, scene->nodesQ.ptr, nodeCount * words * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    *n_nodes = nodeCount;
    if (words_per_node) { *words_per_node = nodeCount > 0 ? words : 0; }
    return PATHED_OK;
}

}  // extern "C"
