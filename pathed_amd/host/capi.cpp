// C entry points of libpathed_host.so for the Python test / bench harness (ctypes).
// These are plumbing around the C++ host classes; the product boundary is
// include/pathed_hip.h.
#include "scene_loader.h"

#include <cstring>
#include <string>

namespace {
thread_local std::string g_hostError;

struct LoadedScene {
    pathed::FlatScene scene;
    PathedSceneDesc desc;
};
}  // namespace

extern "C" {

const char *pathed_host_last_error(void)
{
    return g_hostError.c_str();
}

void *pathed_host_load_scene(const char *scenePath, int width, int height, const char *assetRoot)
{
    try {
        LoadedScene *loaded = new LoadedScene();
        loaded->scene = pathed::loadScene(scenePath, width, height, assetRoot ? assetRoot : "");
        loaded->desc = loaded->scene.desc();
        return loaded;
    } catch (const std::exception &error) {
        g_hostError = error.what();
        return nullptr;
    }
}

const PathedSceneDesc *pathed_host_scene_desc(void *handle)
{
    if (!handle) { return nullptr; }
    return &((LoadedScene *)handle)->desc;
}

void pathed_host_free_scene(void *handle)
{
    delete (LoadedScene *)handle;
}

// tokens of one line, '\n'-separated (reference tokenize / lTrim); returns the token count, -1 if `out` is too small
int pathed_host_tokenize(const char *line, char *out, size_t capacity)
{
    const std::vector<std::string> tokens = pathed::tokenizeLine(line ? line : "");
    std::string joined;
    for (size_t i = 0; i < tokens.size(); i++) { joined += (i ? "\n" : "") + tokens[i]; }
    if (joined.size() + 1 > capacity) { return -1; }
    std::memcpy(out, joined.c_str(), joined.size() + 1);
    return (int)tokens.size();
}

int pathed_host_ltrim(const char *text, char *out, size_t capacity)
{
    const std::string trimmed = pathed::leftTrim(text ? text : "");
    if (trimmed.size() + 1 > capacity) { return -1; }
    std::memcpy(out, trimmed.c_str(), trimmed.size() + 1);
    return (int)trimmed.size();
}

// the materials of an MTL file, one per line in name order: "name\tKd r g b\tKe r g b" (%.9g); returns the count
int pathed_host_parse_mtl(const char *path, char *out, size_t capacity)
{
    try {
        const std::vector<pathed::MtlMaterial> materials = pathed::parseMtlFile(path ? path : "");
        std::string text;
        for (const pathed::MtlMaterial &m : materials) {
            char line[512];
            snprintf(line, sizeof line, "%s\tKd %.9g %.9g %.9g\tKe %.9g %.9g %.9g\n", m.name.c_str(),
                     m.diffuse[0], m.diffuse[1], m.diffuse[2], m.emit[0], m.emit[1], m.emit[2]);
            text += line;
        }
        if (text.size() + 1 > capacity) { g_hostError = "buffer too small"; return -1; }
        std::memcpy(out, text.c_str(), text.size() + 1);
        return (int)materials.size();
    } catch (const std::exception &error) {
        g_hostError = error.what();
        return -2;
    }
}

}  // extern "C"

// ---- job runner (shared by the `pathed` executable and the Python harness) ----------

#include "exr.h"
#include "image.h"
#include "image_decode.h"
#include "integrator.h"
#include "job.h"

#include <algorithm>
#include <chrono>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <iterator>
#include <thread>

// FNV-1a over everything of a loaded scene that a radiance sum depends on (the camera's resolution is in the state header)
static std::string flatSceneDigest(const PathedSceneDesc &desc)
{
    unsigned long long hash = 1469598103934665603ull;
    auto add = [&](const void *data, size_t bytes) {
        const unsigned char *p = static_cast<const unsigned char *>(data);
        for (size_t i = 0; i < bytes; i++) { hash = (hash ^ p[i]) * 1099511628211ull; }
    };
    add(desc.camera.origin, sizeof desc.camera.origin); add(desc.camera.target, sizeof desc.camera.target); add(desc.camera.up, sizeof desc.camera.up);
    add(&desc.camera.vertical_fov, sizeof(float)); add(&desc.camera.flip_handedness, sizeof(int32_t));
    add(desc.positions, sizeof(float) * 3 * desc.n_vertices);
    add(desc.normals, sizeof(float) * 3 * desc.n_vertices);
    add(desc.uvs, sizeof(float) * 2 * desc.n_vertices);
    add(desc.indices, sizeof(uint32_t) * 3 * desc.n_triangles);
    add(desc.tri_material, sizeof(int32_t) * desc.n_triangles);
    add(desc.spheres, sizeof(PathedSphere) * desc.n_spheres);
    add(desc.geoms, sizeof(PathedGeom) * desc.n_geoms);
    for (uint32_t i = 0; i < desc.n_materials; i++) {   // field by field: a digest must not read padding
        const PathedMaterial &m = desc.materials[i];
        add(&m.type, sizeof m.type); add(&m.albedo_type, sizeof m.albedo_type); add(m.diffuse, sizeof m.diffuse); add(m.emit, sizeof m.emit);
        add(m.checker_on, sizeof m.checker_on); add(m.checker_off, sizeof m.checker_off); add(m.checker_res, sizeof m.checker_res);
        add(&m.sigma, sizeof m.sigma); add(&m.alpha, sizeof m.alpha); add(&m.ior, sizeof m.ior); add(&m.distribution, sizeof m.distribution); add(&m.texture, sizeof m.texture);
    }
    if (desc.env) {
        add(&desc.env->width, sizeof(int32_t)); add(&desc.env->height, sizeof(int32_t)); add(&desc.env->scale, sizeof(float));
        add(desc.env->map_to_world, sizeof desc.env->map_to_world);
        add(desc.env->rgba, sizeof(float) * 4 * (size_t)desc.env->width * (size_t)desc.env->height);
    }
    for (uint32_t i = 0; i < desc.n_textures; i++) {
        add(&desc.textures[i].width, sizeof(int32_t)); add(&desc.textures[i].height, sizeof(int32_t));
        add(desc.textures[i].rgb, (size_t)3 * desc.textures[i].width * desc.textures[i].height);
    }
    add(desc.media, sizeof(PathedMedium) * desc.n_media);
    char text[32];
    snprintf(text, sizeof text, "%016llx", hash);
    return text;
}

int runJob(const std::string &jobPath, const std::string &assetRootOverride)
{
    using namespace pathed;
    try {
        Job job(jobPath);
        job.init();

        const int width = job.width();
        const int height = job.height();
        Image image(width, height, job.outputDirectory());

        const std::string assetRoot = !assetRootOverride.empty() ? assetRootOverride : job.assetRoot();
        std::string builder = job.bvhBuilder();
        if (builder != "auto" && builder != "sah" && builder != "lbvh" && builder != "ploc") { throw std::runtime_error("job: unknown bvh_builder: " + builder); }
        const std::vector<int> devices = job.devices();
        const std::string reduce = job.reduceMethod();
        if (reduce != "rccl" && reduce != "peer-copy") { throw std::runtime_error("job: \"reduce\" must be \"rccl\" or \"peer-copy\""); }
        // Several DISTINCT devices and "reduce": "rccl" (the default): the communicator is made first, before the scene is
        // loaded and N trees are built -- a node whose RCCL / xGMI is broken stops the job here, in its first second.
        // (Replicas that share a device, how the fan-out is rehearsed on a one-GPU box, cannot form one: peer copies, decided
        // in Integrator::run.)
        PathedComm *earlyComm = nullptr;
        {
            std::vector<int> ids = devices;
            std::sort(ids.begin(), ids.end());
            const bool distinct = std::adjacent_find(ids.begin(), ids.end()) == ids.end();
            if (devices.size() > 1 && distinct && reduce == "rccl") {
                if (pathed_hip_comm_init((int)devices.size(), devices.data(), &earlyComm) != PATHED_OK) {
                    throw std::runtime_error("job asks for the RCCL reduce over " + std::to_string(devices.size()) + " distinct GPUs and RCCL is unavailable: "
                                             + std::string(pathed_hip_last_error()) + " (set \"reduce\": \"peer-copy\" to use peer copies)");
                }
            }
        }
        struct CommGuard { PathedComm *&comm; ~CommGuard() { if (comm) { pathed_hip_comm_destroy(comm); } } } commGuard{ earlyComm };
        const auto loadBegin = std::chrono::steady_clock::now();
        FlatScene flat = loadScene(job.scene(), width, height, assetRoot);
        const std::string sceneDigest = flatSceneDigest(flat.desc());
        if (builder == "auto") {
            // several replicas of a large mesh: each GPU builds its own tree in milliseconds (PLOC) instead of N host SAH
            // builds competing for the cores before the first sample (include/pathed_hip.h: PATHED_BVH_*)
            builder = (devices.size() > 1 && flat.desc().n_triangles > 1000000u) ? "ploc" : "sah";
        }
        const int builderCode = builder == "ploc" ? PATHED_BVH_PLOC_DEVICE : builder == "lbvh" ? PATHED_BVH_LBVH_DEVICE : PATHED_BVH_SAH_HOST;
        const auto uploadBegin = std::chrono::steady_clock::now();
        // one replica of the scene per device; the device travels with the scene handle, so the
        // render thread below (and its per-device workers) need no device selection of their own
        Scene scene(std::move(flat), devices, builderCode);
        const auto uploadEnd = std::chrono::steady_clock::now();

        std::shared_ptr<Integrator> integrator = job.integrator();
        integrator->configure(job.spp(), job.seed(), job.sppPerLaunch(), job.outputDirectory());
        integrator->setStateFile(job.outputDirectory() + "auto.state", job.resume());
        {
            // what the sums depend on besides resolution, seed and bounce window (those are in the state header): the scene as
            // LOADED -- every vertex, index, material, sphere, environment texel and texture byte, so a mesh or a map edited in
            // place is another render -- the integrator, and the BVH builder (hits do not depend on the tree, but "the same job"
            // should mean the same tree: with "auto" the builder follows the number of GPUs)
            integrator->setStateIdentity(job.scene() + "|" + job.integratorName() + "|" + assetRoot + "|" + builder + "|" + sceneDigest);
        }
        integrator->setUseRccl(reduce == "rccl");
        if (earlyComm) { integrator->adoptComm(earlyComm); earlyComm = nullptr; }
        const std::string metricsLevel = job.metricsLevel();
        if (metricsLevel != "full" && metricsLevel != "basic") { throw std::runtime_error("job: \"metrics\" must be \"full\" or \"basic\""); }

        // the reference renders on a dedicated thread while the UI owns the main thread
        // (app/main.cpp:98); kept so `quit` and the Image lock behave the same
        bool quit = false;
        std::string failure;
        std::thread renderThread([&]() {
            try {
                integrator->run(image, scene, [](RenderStatus) {}, &quit);
            } catch (const std::exception &error) {
                failure = error.what();
                if (failure.empty()) { failure = "render failed"; }
            }
        });
        renderThread.join();
        if (!failure.empty()) { throw std::runtime_error(failure); }

        // <outdir>/metrics.json: what this run did and how fast
        {
            const RenderMetrics &m = integrator->metrics();
            PathedStats stats;
            std::memset(&stats, 0, sizeof stats);
            pathed_hip_get_stats(scene.handle(), &stats);
            const double samples = (double)m.width * m.height * (double)(m.lastSample - m.firstSample);
            // what the reference prints per render (RTCManager::printStats, src/rtc_manager.cpp:94-115) and per wave
            // (src/integrator.cpp:97-102), as rates: a one-sample counting pass over sample index `spp` into a scratch
            // buffer (the COUNT kernel variants never run inside the timed loop)
            PathedStats counted;
            std::memset(&counted, 0, sizeof counted);
            bool haveCounts = false;
            if (metricsLevel == "full" && samples > 0.0) {
                float *scratch = nullptr;
                const size_t floats = (size_t)3 * m.width * m.height;
                if (pathed_hip_accum_alloc(scene.handle(), floats, &scratch) == PATHED_OK) {
                    pathed_hip_set_stats_mode(scene.handle(), 1);
                    pathed_hip_reset_stats(scene.handle());
                    haveCounts = pathed_hip_render_device(scene.handle(), job.seed(), (uint32_t)m.lastSample, 1, job.startBounce(), job.lastBounce(),
                                                          scratch, nullptr, 1) == PATHED_OK
                        && pathed_hip_get_stats(scene.handle(), &counted) == PATHED_OK;
                    pathed_hip_set_stats_mode(scene.handle(), 0);
                    pathed_hip_accum_free(scene.handle(), scratch);
                }
            }
            std::ofstream out(job.outputDirectory() + "metrics.json");
            out << std::setprecision(9);
            out << "{\n";
            out << "  \"width\": " << m.width << ", \"height\": " << m.height << ",\n";
            out << "  \"first_sample\": " << m.firstSample << ", \"last_sample\": " << m.lastSample << ",\n";
            out << "  \"devices\": [";
            for (size_t i = 0; i < devices.size(); i++) { out << (i ? ", " : "") << devices[i]; }
            out << "],\n";
            out << "  \"scene_load_seconds\": " << std::chrono::duration<double>(uploadBegin - loadBegin).count() << ",\n";
            out << "  \"scene_upload_seconds\": " << std::chrono::duration<double>(uploadEnd - uploadBegin).count() << ",\n";
            out << "  \"bvh_builder\": \"" << builder << "\", \"bvh_build_ms\": " << stats.bvh_build_ms
                << ", \"bvh_bytes\": " << stats.bvh_bytes << ", \"bvh_nodes\": " << stats.bvh_nodes << ",\n";
            out << "  \"render_seconds\": " << m.loopSeconds << ",\n";
            out << "  \"msamples_per_second\": " << (m.loopSeconds > 0.0 ? samples / m.loopSeconds / 1e6 : 0.0) << ",\n";
            out << "  \"reduce_seconds\": " << m.reduceSeconds << ", \"reduces\": " << m.reduces << ", \"reduce_method\": \"" << m.reduceMethod << "\",\n";
            if (haveCounts && counted.camera_samples > 0) {
                // per ray 32 B + S_hit (16 closest / 4 any-hit) + 32 B per child box + 48 B per triangle tested
                const double perSample = 1.0 / (double)counted.camera_samples;
                const double rays = (double)(counted.closest_rays + counted.shadow_rays) * perSample;
                const double bytes = (48.0 * counted.closest_rays + 36.0 * counted.shadow_rays + 32.0 * counted.nodes_visited + 48.0 * counted.tris_tested) * perSample;
                const double rate = m.loopSeconds > 0.0 ? samples / m.loopSeconds : 0.0;
                out << "  \"rays_per_sample\": " << rays << ", \"mrays_per_second\": " << rays * rate / 1e6 << ",\n";
                out << "  \"boxes_per_ray\": " << (rays > 0.0 ? counted.nodes_visited * perSample / rays : 0.0)
                    << ", \"triangles_per_ray\": " << (rays > 0.0 ? counted.tris_tested * perSample / rays : 0.0) << ",\n";
                out << "  \"trace_algorithmic_bytes_per_sample\": " << bytes << ", \"trace_algorithmic_gbs\": " << bytes * rate / 1e9
                    << ", \"hbm_roofline_fraction\": " << bytes * rate / 8.0e12 << ",\n";
                out << "  \"counter_bytes_per_sample\": null,\n";   // measured HBM bytes need a rocprofv3 --pmc pass (tools/pmc_per_sample.sh)
            }
            out << "  \"replica_seconds\": [";
            for (size_t i = 0; i < m.replicaSeconds.size(); i++) { out << (i ? ", " : "") << m.replicaSeconds[i]; }
            out << "],\n";
            out << "  \"dropped_samples\": " << stats.dropped_samples << "\n";
            out << "}\n";
        }
        return 0;
    } catch (const std::exception &error) {
        g_hostError = error.what();
        std::cerr << "pathed: " << error.what() << std::endl;
        return 1;
    }
}

extern "C" {

int pathed_host_run_job(const char *jobPath, const char *assetRoot)
{
    return runJob(jobPath ? jobPath : "job.json", assetRoot ? assetRoot : "");
}

int pathed_host_write_exr_float_rgba(const char *path, int width, int height, const float *rgba)
{
    std::string error;
    if (!pathed::writeExrFloatRGBA(path, width, height, rgba, &error)) {
        g_hostError = error;
        return 1;
    }
    return 0;
}

// capacity = number of floats `rgba` can hold; pass NULL to query the size
int pathed_host_read_exr_rgba(const char *path, int *width, int *height, float *rgba, size_t capacity)
{
    std::vector<float> data;
    std::string error;
    if (!pathed::readExrRGBA(path, width, height, &data, &error)) {
        g_hostError = error;
        return 1;
    }
    if (rgba) {
        if (capacity < data.size()) { g_hostError = "buffer too small"; return 2; }
        std::memcpy(rgba, data.data(), data.size() * sizeof(float));
    }
    return 0;
}

// the reference's BMP preview writer (Image::write -> stbi_write_bmp); rgb is top-down interleaved
int pathed_host_write_bmp_rgb8(const char *path, int width, int height, const unsigned char *rgb)
{
    if (!pathed::writeBmpRgb8(path, width, height, rgb)) { g_hostError = std::string("cannot write ") + path; return 1; }
    return 0;
}

// 8-bit RGB decode of a texture file (image_decode.h); pass rgb = NULL to query the size
int pathed_host_load_image_rgb8(const char *path, int *width, int *height, unsigned char *rgb, size_t capacity)
{
    std::vector<uint8_t> data;
    std::string error;
    if (!pathed::loadImageRgb8(path, width, height, &data, &error)) {
        g_hostError = error;
        return 1;
    }
    if (rgb) {
        if (capacity < data.size()) { g_hostError = "buffer too small"; return 2; }
        std::memcpy(rgb, data.data(), data.size());
    }
    return 0;
}

}  // extern "C"

// planar-free helper for the Python multi-GPU job runner: mean image, interleaved RGB, row 0 =
// BOTTOM scanline (the integrator's layout); written like Image::set + Image::save would
// (vertical flip, B/G/R HALF, reference src/image.cpp:21-35, 80-154)
extern "C" int pathed_host_write_exr_half_bgr(const char *path, int width, int height, const float *rgbBottomUp)
{
    const size_t pixels = (size_t)width * height;
    std::vector<float> planes[3];
    for (int c = 0; c < 3; c++) { planes[c].resize(pixels); }
    for (int row = 0; row < height; row++) {
        for (int col = 0; col < width; col++) {
            const size_t source = (size_t)3 * ((size_t)row * width + col);
            const size_t target = (size_t)(height - row - 1) * width + col;
            planes[0][target] = rgbBottomUp[source + 0];
            planes[1][target] = rgbBottomUp[source + 1];
            planes[2][target] = rgbBottomUp[source + 2];
        }
    }
    std::string error;
    if (!pathed::writeExrHalfBGR(path, width, height, planes[0].data(), planes[1].data(), planes[2].data(), &error)) {
        g_hostError = error;
        return 1;
    }
    return 0;
}
