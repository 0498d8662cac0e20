#include "integrator.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <thread>

namespace pathed {

namespace {

std::string hipError(const char *what)
{
    return std::string(what) + ": " + pathed_hip_last_error();
}

// One host thread per replica (= per device), alive for the whole render: `run(work)` hands work(replica) to every
// worker and returns when all are through; the first failure is rethrown on the caller's thread.  (Round 2 created and
// joined the threads for every batch: at the 1-, 1-, 2-, 4-sample batches of the first checkpoints that was most of
// the batch.)
class ReplicaWorkers {
public:
    explicit ReplicaWorkers(size_t replicas) : m_failures(replicas)
    {
        for (size_t r = 1; r < replicas; r++) { m_threads.emplace_back([this, r]() { serve(r); }); }
    }
    ~ReplicaWorkers()
    {
        {
            std::lock_guard<std::mutex> guard(m_mutex);
            m_stop = true;
            m_generation++;
        }
        m_wake.notify_all();
        for (std::thread &thread : m_threads) { thread.join(); }
    }
    void run(const std::function<void(size_t)> &work)
    {
        const size_t replicas = m_failures.size();
        for (std::string &failure : m_failures) { failure.clear(); }
        if (replicas > 1) {
            std::lock_guard<std::mutex> guard(m_mutex);
            m_work = &work;
            m_pending = replicas - 1;
            m_generation++;
        }
        m_wake.notify_all();
        attempt(work, 0);   // replica 0 on the caller's thread
        if (replicas > 1) {
            std::unique_lock<std::mutex> lock(m_mutex);
            m_done.wait(lock, [this]() { return m_pending == 0; });
            m_work = nullptr;
        }
        for (const std::string &failure : m_failures) {
            if (!failure.empty()) { throw std::runtime_error(failure); }
        }
    }

private:
    void attempt(const std::function<void(size_t)> &work, size_t replica)
    {
        try { work(replica); }
        catch (const std::exception &error) { m_failures[replica] = error.what(); if (m_failures[replica].empty()) { m_failures[replica] = "unknown error"; } }
        catch (...) { m_failures[replica] = "unknown error"; }
    }
    void serve(size_t replica)
    {
        unsigned long long seen = 0;
        while (true) {
            const std::function<void(size_t)> *work = nullptr;
            {
                std::unique_lock<std::mutex> lock(m_mutex);
                m_wake.wait(lock, [&]() { return m_generation != seen; });
                seen = m_generation;
                if (m_stop) { return; }
                work = m_work;
            }
            if (work) { attempt(*work, replica); }
            {
                std::lock_guard<std::mutex> guard(m_mutex);
                if (m_pending > 0) { m_pending--; }
            }
            m_done.notify_one();
        }
    }

    std::vector<std::thread> m_threads;
    std::vector<std::string> m_failures;
    std::mutex m_mutex;
    std::condition_variable m_wake, m_done;
    const std::function<void(size_t)> *m_work = nullptr;
    size_t m_pending = 0;
    unsigned long long m_generation = 0;
    bool m_stop = false;
};

// The sidecar a resumed job continues from: the fp32 radiance SUMS and how many samples they hold.
// Because the random stream is a pure function of (seed, pixel, sample, dimension), "the next
// sample" needs no generator state: a resumed run is the straight run, bit for bit, as long as the
// interruption fell on a batch boundary (it always does: the file is written after a batch).
struct StateHeader {
    char magic[8];            // "PATHEDS3"
    int32_t width, height;
    int32_t done;             // samples per pixel already in the sums
    int32_t startBounce, lastBounce;
    uint64_t seed;
    uint64_t jobDigest;       // what else the sums depend on: scene file, integrator, samples per unit (Integrator::setStateIdentity)
};

}  // namespace

Scene::Scene(FlatScene flat, int device)
    : m_flat(std::move(flat)), m_devices(1, device)
{
    upload(0);
}

Scene::Scene(FlatScene flat, const std::vector<int> &devices, int bvhBuilder)
    : m_flat(std::move(flat)), m_devices(devices)
{
    if (m_devices.empty()) { throw std::runtime_error("Scene: no device"); }
    upload(bvhBuilder + 1);
}

void Scene::upload(int bvhBuilderPlusOne)
{
    m_handles.assign(m_devices.size(), nullptr);
    const PathedSceneDesc desc = m_flat.desc();
    try {
        // every device builds / uploads its own replica, in parallel
        ReplicaWorkers workers(m_devices.size());
        workers.run([&](size_t r) {
            PathedSceneOptions options;
            std::memset(&options, 0, sizeof options);
            options.struct_size = sizeof options;
            options.device = m_devices[r];
            options.bvh_builder = bvhBuilderPlusOne;
            if (pathed_hip_scene_create_ex(&desc, &options, &m_handles[r]) != PATHED_OK) {
                throw std::runtime_error(hipError("pathed_hip_scene_create_ex"));
            }
        });
    } catch (...) {
        for (PathedScene *handle : m_handles) { pathed_hip_scene_destroy(handle); }
        throw;
    }
}

Scene::~Scene()
{
    for (PathedScene *handle : m_handles) { pathed_hip_scene_destroy(handle); }
}

void Integrator::configure(int spp, unsigned long long seed, int sppPerLaunch, const std::string &logPrefix)
{
    if (sppPerLaunch < 1) { throw std::runtime_error("spp_per_launch must be >= 1"); }
    if (spp < 1) { throw std::runtime_error("spp must be >= 1"); }
    m_spp = spp;
    m_seed = seed;
    m_sppPerLaunch = sppPerLaunch;
    m_logPrefix = logPrefix;
}

// reference src/integrator.cpp:19-106.  The reference adds one sample per pixel per wave to a host
// vector; here the sums stay on the devices (one buffer per replica) and come to the host only when
// an image is due: at the power-of-two checkpoints (src/integrator.cpp:87-92), at the end, and when
// `quit` is raised.  With G replicas the samples of a batch [done, done + count) are split
// G ways (contiguous shares, strongRange), every replica adds its share into its own buffer on its
// own host thread, and an image is  sum over replicas  / done: waves are additive
// (src/integrator.cpp:42-51), so the union over replicas of what they have rendered is always
// exactly the samples [0, done) -- the single-GPU image up to fp32 summation order.
void Integrator::run(
    Image &image,
    Scene &scene,
    std::function<void(RenderStatus)> callback,
    bool *quit
) {
    const int width = scene.width();
    const int height = scene.height();
    const int primarySamples = m_spp;
    const size_t replicas = scene.replicas();
    const size_t floats = (size_t)3 * width * height;

    printf("Beginning pre-process...\n");
    preprocess(scene);
    printf("Pre-process complete (0.0s elapsed)\n");

    std::vector<float *> deviceSums(replicas, nullptr);
    float *staging = nullptr;   // on replica 0's device: a peer's sums on their way into the total (peer-copy fan-in only)
    float *total = nullptr;     // on replica 0's device: sum over replicas (G > 1 only)
    PathedComm *comm = nullptr; // RCCL communicator over the replicas' devices (distinct devices only)
    struct Cleanup {
        Scene &scene;
        std::vector<float *> &sums;
        float *&staging;
        float *&total;
        PathedComm *&comm;
        ~Cleanup()
        {
            pathed_hip_comm_destroy(comm);
            for (size_t r = 0; r < sums.size(); r++) { if (sums[r]) { pathed_hip_accum_free(scene.handle(r), sums[r]); } }
            if (staging) { pathed_hip_accum_free(scene.handle(0), staging); }
            if (total) { pathed_hip_accum_free(scene.handle(0), total); }
        }
    } cleanup{ scene, deviceSums, staging, total, comm };
    for (size_t r = 0; r < replicas; r++) {
        if (pathed_hip_accum_alloc(scene.handle(r), floats, &deviceSums[r]) != PATHED_OK) { throw std::runtime_error(hipError("pathed_hip_accum_alloc")); }
    }
    m_metrics = RenderMetrics();
    if (replicas > 1) {
        if (pathed_hip_accum_alloc(scene.handle(0), floats, &total) != PATHED_OK) { throw std::runtime_error(hipError("pathed_hip_accum_alloc")); }
        // the path's one exchange step as ONE collective (RCCL reduce over xGMI); replicas that share a
        // device, or a machine without librccl, fall back to peer copies + adds on replica 0
        if (m_adoptedComm) { comm = m_adoptedComm; m_adoptedComm = nullptr; }
        else if (m_useRccl && pathed_hip_comm_init((int)replicas, scene.devices().data(), &comm) != PATHED_OK) {
            comm = nullptr;
            m_metrics.reduceFallback = pathed_hip_last_error();
            // replicas that SHARE a device cannot form a communicator (rehearsals on one GPU): peer copies.  On distinct
            // devices -- a real node -- "reduce": "rccl" means RCCL: a silent fall-back would hide a broken xGMI / RCCL setup
            // behind a slower path, so the job stops here ("reduce": "peer-copy" asks for the copies explicitly).
            std::vector<int> ids = scene.devices();
            std::sort(ids.begin(), ids.end());
            const bool distinct = std::adjacent_find(ids.begin(), ids.end()) == ids.end();
            if (distinct) {
                throw std::runtime_error("job asks for the RCCL reduce over " + std::to_string(replicas) + " distinct GPUs and RCCL is unavailable: "
                                         + m_metrics.reduceFallback + " (set \"reduce\": \"peer-copy\" to use peer copies)");
            }
            std::cout << "[" << m_logPrefix << "] RCCL reduce unavailable (" << m_metrics.reduceFallback << "): replicas share a device, using peer copies" << std::endl;
        }
        if (!comm) {
            if (pathed_hip_accum_alloc(scene.handle(0), floats, &staging) != PATHED_OK) { throw std::runtime_error(hipError("pathed_hip_accum_alloc")); }
        }
    }
    m_metrics.reduceMethod = replicas == 1 ? "none" : comm ? "rccl" : "peer-copy";
    ReplicaWorkers workers(replicas);

    std::vector<float> radianceLookup(floats, 0.f);
    int done = 0;
    if (m_resume) {
        done = loadState(radianceLookup, width, height);
        if (done > 0) {
            // the sums so far continue on replica 0
            if (pathed_hip_accum_upload(scene.handle(0), deviceSums[0], floats, radianceLookup.data()) != PATHED_OK) {
                throw std::runtime_error(hipError("pathed_hip_accum_upload"));
            }
            std::cout << "[" << m_logPrefix << "] resuming at sample " << done << "/" << primarySamples << std::endl;
        }
        if (done >= primarySamples) {
            // nothing left to render: the image is the one the state file holds
            std::lock_guard<std::mutex> guard(image.getLock());
            image.setSpp(done);
            for (int row = 0; row < height; row++) {
                for (int col = 0; col < width; col++) {
                    const size_t index = (size_t)3 * ((size_t)row * width + col);
                    image.set(row, col, radianceLookup[index + 0] / done, radianceLookup[index + 1] / done, radianceLookup[index + 2] / done);
                }
            }
            std::cout << "[" << m_logPrefix << "] the state file already holds " << done << " samples: nothing to render" << std::endl;
        }
    }

    m_metrics.replicas = (int)replicas;
    m_metrics.firstSample = done;
    m_metrics.replicaSeconds.assign(replicas, 0.0);
    const auto loopBegin = std::chrono::steady_clock::now();

    while (done < primarySamples) {
        // batches end on powers of two so checkpoints appear exactly where the reference
        // writes them (src/integrator.cpp:87-92)
        int nextPower = 1;
        while (nextPower <= done) { nextPower *= 2; }
        long long limit = (long long)m_sppPerLaunch * (long long)replicas;
        int count = (int)std::min<long long>(limit, primarySamples - done);
        count = std::min(count, nextPower - done);

        const auto begin = std::chrono::steady_clock::now();
        workers.run([&](size_t r) {
            unsigned first = 0, mine = 0;
            strongRange((unsigned)r, (unsigned)replicas, (unsigned)done, (unsigned)count, &first, &mine);
            if (mine == 0) { return; }
            const auto replicaBegin = std::chrono::steady_clock::now();
            sampleImage(deviceSums[r], scene, r, first, mine);
            m_metrics.replicaSeconds[r] += std::chrono::duration<double>(std::chrono::steady_clock::now() - replicaBegin).count();
        });
        const auto end = std::chrono::steady_clock::now();
        done += count;

        postwave(scene, done);

        RenderStatus status;
        status.sample = done;
        status.elapsedSeconds = std::chrono::duration<double>(end - begin).count();
        if (callback) { callback(status); }

        const bool checkpoint = (done & (done - 1)) == 0;
        const bool stopping = quit && *quit;
        if (checkpoint || done == primarySamples || stopping) {
            const auto reduceBegin = std::chrono::steady_clock::now();
            const float *source = deviceSums[0];
            if (replicas > 1 && comm) {
                std::vector<const float *> send(deviceSums.begin(), deviceSums.end());
                if (pathed_hip_comm_reduce(comm, send.data(), total, floats) != PATHED_OK) { throw std::runtime_error(hipError("pathed_hip_comm_reduce")); }
                source = total;
            } else if (replicas > 1) {
                // per-device sums -> replica 0 one by one (hipMemcpyPeer over xGMI + add)
                if (pathed_hip_accum_copy_peer(scene.handle(0), total, scene.handle(0), deviceSums[0], floats) != PATHED_OK) { throw std::runtime_error(hipError("pathed_hip_accum_copy_peer")); }
                for (size_t r = 1; r < replicas; r++) {
                    if (pathed_hip_accum_copy_peer(scene.handle(0), staging, scene.handle(r), deviceSums[r], floats) != PATHED_OK) { throw std::runtime_error(hipError("pathed_hip_accum_copy_peer")); }
                    if (pathed_hip_accum_add(scene.handle(0), total, staging, floats) != PATHED_OK) { throw std::runtime_error(hipError("pathed_hip_accum_add")); }
                }
                source = total;
            }
            if (pathed_hip_accum_download(scene.handle(0), source, floats, radianceLookup.data()) != PATHED_OK) { throw std::runtime_error(hipError("pathed_hip_accum_download")); }
            m_metrics.reduceSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - reduceBegin).count();
            m_metrics.reduces++;

            std::lock_guard<std::mutex> guard(image.getLock());
            image.setSpp(done);
            for (int row = 0; row < height; row++) {
                for (int col = 0; col < width; col++) {
                    const size_t index = (size_t)3 * ((size_t)row * width + col);
                    image.set(
                        row, col,
                        radianceLookup[index + 0] / done,
                        radianceLookup[index + 1] / done,
                        radianceLookup[index + 2] / done);
                }
            }
            // files: auto.exr + auto-%05dspp.exr at the powers of two, as the reference (src/integrator.cpp:87-92).  The
            // reference renders ALL primarySamples waves (:42) and its last samples reach no file when the count is not a
            // power of two; here the end of the run also refreshes auto.exr -- "the latest image" -- so that every sample
            // rendered is in a file (no numbered checkpoint for it).  python -m pathed_amd.run_job does exactly the same.
            if (checkpoint) { image.saveCheckpoint("auto"); }
            else if (done == primarySamples) { image.save("auto"); }
            saveState(radianceLookup, width, height, done);
        }

        std::ostringstream line;
        line << "[" << m_logPrefix << "] sample: " << done << "/" << primarySamples
             << std::fixed << std::setprecision(1)
             << " (" << status.elapsedSeconds << "s elapsed)";
        std::cout << line.str() << std::endl;

        if (stopping) { break; }
    }
    m_metrics.lastSample = done;
    m_metrics.loopSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - loopBegin).count();
    m_metrics.width = width;
    m_metrics.height = height;
}

void Integrator::setStateIdentity(const std::string &identity)
{
    unsigned long long hash = 1469598103934665603ull;   // FNV-1a
    for (unsigned char c : identity) { hash = (hash ^ c) * 1099511628211ull; }
    m_stateDigest = hash;
}

void Integrator::saveState(const std::vector<float> &sums, int width, int height, int done) const
{
    if (m_statePath.empty()) { return; }
    StateHeader header;
    std::memset(&header, 0, sizeof header);
    std::memcpy(header.magic, "PATHEDS3", 8);
    header.width = width;
    header.height = height;
    header.done = done;
    header.startBounce = stateStartBounce();
    header.lastBounce = stateLastBounce();
    header.seed = m_seed;
    header.jobDigest = m_stateDigest;
    // write beside, then rename: an interrupted write never leaves a truncated state behind
    const std::string scratch = m_statePath + ".tmp";
    {
        std::ofstream out(scratch, std::ios::binary | std::ios::trunc);
        out.write(reinterpret_cast<const char *>(&header), sizeof header);
        out.write(reinterpret_cast<const char *>(sums.data()), (std::streamsize)(sums.size() * sizeof(float)));
        if (!out) { fprintf(stderr, "pathed: cannot write %s\n", scratch.c_str()); return; }
    }
    if (std::rename(scratch.c_str(), m_statePath.c_str()) != 0) { fprintf(stderr, "pathed: cannot rename %s\n", scratch.c_str()); }
}

int Integrator::loadState(std::vector<float> &sums, int width, int height) const
{
    if (m_statePath.empty()) { return 0; }
    std::ifstream in(m_statePath, std::ios::binary);
    if (!in) { return 0; }   // nothing to resume from: start at sample 0
    StateHeader header;
    in.read(reinterpret_cast<char *>(&header), sizeof header);
    if (!in || std::memcmp(header.magic, "PATHEDS3", 8) != 0) { throw std::runtime_error("resume: " + m_statePath + " is not a pathed state file (of this version)"); }
    if (header.width != width || header.height != height) { throw std::runtime_error("resume: state file has another resolution"); }
    if (header.seed != m_seed || header.startBounce != stateStartBounce() || header.lastBounce != stateLastBounce()) {
        throw std::runtime_error("resume: state file was rendered with another seed or bounce window");
    }
    if (header.jobDigest != m_stateDigest) { throw std::runtime_error("resume: state file was rendered from another scene, integrator or job configuration"); }
    if (header.done < 0 || header.done > m_spp) { throw std::runtime_error("resume: state file holds more samples than the job asks for"); }
    in.read(reinterpret_cast<char *>(sums.data()), (std::streamsize)(sums.size() * sizeof(float)));
    if (!in) { throw std::runtime_error("resume: state file is truncated"); }
    return header.done;
}

void HipPathTracer::sampleImage(float *deviceSums, Scene &scene, size_t replica, unsigned begin, unsigned count)
{
    if (pathed_hip_set_integrator(scene.handle(replica), m_integrator) != PATHED_OK) {
        throw std::runtime_error(std::string("pathed_hip_set_integrator: ") + pathed_hip_last_error());
    }
    const int code = pathed_hip_render_device(
        scene.handle(replica), m_seed, begin, count,
        m_bounceController.startBounce(), m_bounceController.lastBounce(),
        deviceSums, nullptr, 1);
    if (code != PATHED_OK) {
        throw std::runtime_error(std::string("pathed_hip_render_device: ") + pathed_hip_last_error());
    }
}

}  // namespace pathed
