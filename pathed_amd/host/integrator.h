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

// Host-side Scene: the parsed description plus its upload (reference include/scene.h:83-130
// owns the Embree scene the same way).
class Scene {
public:
    Scene(FlatScene flat, int device);
    ~Scene();
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;

    PathedScene *handle() const { return m_handle; }
    const FlatScene &flat() const { return m_flat; }
    int width() const { return m_flat.camera.width; }
    int height() const { return m_flat.camera.height; }

private:
    FlatScene m_flat;
    PathedScene *m_handle;
};

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

    void configure(int spp, unsigned long long seed, int sppPerLaunch, const std::string &logPrefix)
    {
        m_spp = spp;
        m_seed = seed;
        m_sppPerLaunch = sppPerLaunch;
        m_logPrefix = logPrefix;
    }

protected:
    // adds `count` samples of every pixel, starting at sample index `begin`, to radianceLookup
    // (reference sampleImage adds exactly one)
    virtual void sampleImage(std::vector<float> &radianceLookup, Scene &scene, unsigned begin, unsigned count) = 0;

    int m_spp = 1;
    unsigned long long m_seed = 1;
    int m_sppPerLaunch = 64;
    std::string m_logPrefix;
};

class HipPathTracer : public Integrator {
public:
    explicit HipPathTracer(BounceController bounceController)
        : m_bounceController(bounceController)
    {}

protected:
    void sampleImage(std::vector<float> &radianceLookup, Scene &scene, unsigned begin, unsigned count) override;

private:
    BounceController m_bounceController;
};

}  // namespace pathed
