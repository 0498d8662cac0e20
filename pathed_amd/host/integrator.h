// The reference's Integrator plug-in surface, re-targeted at the C ABI.
//   reference include/integrator.h:16-55   class Integrator { virtual run(...); hooks }
//   reference src/integrator.cpp:19-106    wave loop, sum -> mean, power-of-two checkpoints
//   reference include/path_tracer.h:6-46   PathTracer(BounceController)
// HipPathTracer overrides run() (as the reference's PDFIntegrator does) so it can hand many
// samples per pixel to the GPU per call instead of one wave.
#pragma once

#include "bounce_controller.h"
#include "image.h"
#include "scene_loader.h"

#include "pathed_hip.h"

#include <functional>
#include <string>
#include <vector>

namespace pathed {

// what the progress callback receives (reference include/render_status.h, UI fields dropped)
struct RenderStatus {
    int sample = 0;
    double elapsedSeconds = 0.0;
};

// what a render leaves behind for <outdir>/metrics.json
struct RenderMetrics {
    int width = 0, height = 0;
    int replicas = 1;
    int firstSample = 0, lastSample = 0;   // samples [first, last) were rendered by this run
    double loopSeconds = 0.0;              // the render loop, checkpoints included
    double reduceSeconds = 0.0;            // fan-in of the per-device sums + download
    int reduces = 0;
    std::string reduceMethod = "none";     // "rccl" (pathed_hip_comm_reduce), "peer-copy" (hipMemcpyPeer + add) or "none" (one replica)
    std::string reduceFallback;            // why RCCL was not used, when it was asked for and is not
    std::vector<double> replicaSeconds;    // time each replica spent inside sampleImage
};

// Host-side Scene: the parsed description plus its upload (reference include/scene.h:83-130
// owns the Embree scene the same way).
// With several devices the description is uploaded to each of them (scene + BVH replicated,
// SURVEY.md §8e); handle(r) is the replica on devices()[r].  The same id may appear twice: two
// replicas on one GPU, which is how the fan-out is tested on a one-GPU box.
class Scene {
public:
    Scene(FlatScene flat, int device);
    Scene(FlatScene flat, const std::vector<int> &devices, int bvhBuilder);
    ~Scene();
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;

    PathedScene *handle() const { return m_handles[0]; }
    PathedScene *handle(size_t replica) const { return m_handles[replica]; }
    size_t replicas() const { return m_handles.size(); }
    const std::vector<int> &devices() const { return m_devices; }
    const FlatScene &flat() const { return m_flat; }
    int width() const { return m_flat.camera.width; }
    int height() const { return m_flat.camera.height; }

private:
    void upload(int bvhBuilder);

    FlatScene m_flat;
    std::vector<int> m_devices;
    std::vector<PathedScene *> m_handles;
};

// Samples [first, first + count) split over `parts` (total work fixed): part r gets a contiguous
// share, the first count % parts parts one sample more.  Same rule as pathed_amd/parallel.py.
inline void strongRange(unsigned part, unsigned parts, unsigned first, unsigned count, unsigned *begin, unsigned *mine)
{
    const unsigned base = count / parts, extra = count % parts;
    *begin = first + part * base + (part < extra ? part : extra);
    *mine = base + (part < extra ? 1u : 0u);
}

class Integrator {
public:
    virtual ~Integrator() {}

    virtual void run(
        Image &image,
        Scene &scene,
        std::function<void(RenderStatus)> callback,
        bool *quit
    );

    virtual void preprocess(const Scene &) {}
    virtual void postwave(const Scene &, int /*waveCount*/) {}

    void configure(int spp, unsigned long long seed, int sppPerLaunch, const std::string &logPrefix);

    // <outdir>/auto.state: fp32 radiance sums + sample count, rewritten whenever an image is published.
    // With resume the run continues from it (a missing file means "start at 0").
    void setStateFile(const std::string &path, bool resume) { m_statePath = path; m_resume = resume; }
    // what the sums depend on besides resolution, seed and bounce window (which the state header carries itself): the
    // caller passes scene path + integrator name + asset root + BVH builder + a digest of the LOADED scene (geometry, materials,
    // environment map, textures: an asset edited in place is another render); a state file of another identity is refused
    void setStateIdentity(const std::string &identity);
    // job key "reduce": "rccl" (default: ONE ncclReduce; an error when RCCL is unavailable on distinct devices, peer copies only
    // for replicas that share a device) or "peer-copy"
    void setUseRccl(bool use) { m_useRccl = use; }
    // A communicator over the replicas' devices made BEFORE the scene was loaded and its trees were built (runJob does that, so
    // that a machine without a usable RCCL stops the job in its first second, not after N BVH builds): run() uses it instead
    // of making its own, and destroys it when it is done.
    void adoptComm(PathedComm *comm) { m_adoptedComm = comm; }
    const RenderMetrics &metrics() const { return m_metrics; }

protected:
    // adds `count` samples of every pixel, starting at sample index `begin`, to the radiance sums
    // `deviceSums` that live on replica `replica`'s device (reference sampleImage adds exactly one
    // sample to radianceLookup)
    virtual void sampleImage(float *deviceSums, Scene &scene, size_t replica, unsigned begin, unsigned count) = 0;

    int m_spp = 1;
    unsigned long long m_seed = 1;
    virtual int stateStartBounce() const { return 0; }
    virtual int stateLastBounce() const { return -1; }
    void saveState(const std::vector<float> &sums, int width, int height, int done) const;
    int loadState(std::vector<float> &sums, int width, int height) const;

    int m_sppPerLaunch = 1024;
    bool m_resume = false;
    bool m_useRccl = true;
    PathedComm *m_adoptedComm = nullptr;
    unsigned long long m_stateDigest = 0;
    std::string m_statePath;
    std::string m_logPrefix;
    RenderMetrics m_metrics;
};

class HipPathTracer : public Integrator {
public:
    // integrator: PATHED_INTEGRATOR_PATH_TRACER (reference PathTracer) or PATHED_INTEGRATOR_VOLUME_PATH_TRACER
    // (reference VolumePathTracer, src/volume_path_tracer.cpp: participating media)
    explicit HipPathTracer(BounceController bounceController, int integrator = PATHED_INTEGRATOR_PATH_TRACER)
        : m_bounceController(bounceController), m_integrator(integrator)
    {}

protected:
    void sampleImage(float *deviceSums, Scene &scene, size_t replica, unsigned begin, unsigned count) override;
    int stateStartBounce() const override { return m_bounceController.startBounce(); }
    int stateLastBounce() const override { return m_bounceController.lastBounce(); }

private:
    BounceController m_bounceController;
    int m_integrator;
};

}  // namespace pathed
