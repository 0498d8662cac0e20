#include "integrator.h"

#include <chrono>
#include <cstdio>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <stdexcept>

namespace pathed {

Scene::Scene(FlatScene flat, int device)
    : m_flat(std::move(flat)), m_handle(nullptr)
{
    if (pathed_hip_init(device) != PATHED_OK) {
        throw std::runtime_error(std::string("pathed_hip_init: ") + pathed_hip_last_error());
    }
    const PathedSceneDesc desc = m_flat.desc();
    if (pathed_hip_scene_create(&desc, &m_handle) != PATHED_OK) {
        throw std::runtime_error(std::string("pathed_hip_scene_create: ") + pathed_hip_last_error());
    }
}

Scene::~Scene()
{
    pathed_hip_scene_destroy(m_handle);
}

// reference src/integrator.cpp:19-106
void Integrator::run(
    Image &image,
    Scene &scene,
    std::function<void(RenderStatus)> callback,
    bool *quit
) {
    const int width = scene.width();
    const int height = scene.height();
    const int primarySamples = m_spp;

    printf("Beginning pre-process...\n");
    preprocess(scene);
    printf("Pre-process complete (0.0s elapsed)\n");

    std::vector<float> radianceLookup((size_t)3 * width * height, 0.f);

    int done = 0;
    while (done < primarySamples) {
        // batches end on powers of two so checkpoints appear exactly where the reference
        // writes them (src/integrator.cpp:87-92)
        int nextPower = 1;
        while (nextPower <= done) { nextPower *= 2; }
        int count = std::min(m_sppPerLaunch, primarySamples - done);
        count = std::min(count, nextPower - done);

        const auto begin = std::chrono::steady_clock::now();
        sampleImage(radianceLookup, scene, (unsigned)done, (unsigned)count);
        const auto end = std::chrono::steady_clock::now();
        done += count;

        postwave(scene, done);

        RenderStatus status;
        status.sample = done;
        status.elapsedSeconds = std::chrono::duration<double>(end - begin).count();
        if (callback) { callback(status); }

        {
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
            if ((done & (done - 1)) == 0) { image.saveCheckpoint("auto"); }
        }

        std::ostringstream line;
        line << "[" << m_logPrefix << "] sample: " << done << "/" << primarySamples
             << std::fixed << std::setprecision(1)
             << " (" << status.elapsedSeconds << "s elapsed)";
        std::cout << line.str() << std::endl;

        if (quit && *quit) { return; }
    }
}

void HipPathTracer::sampleImage(std::vector<float> &radianceLookup, Scene &scene, unsigned begin, unsigned count)
{
    const int code = pathed_hip_render(
        scene.handle(), m_seed, begin, count,
        m_bounceController.startBounce(), m_bounceController.lastBounce(),
        radianceLookup.data());
    if (code != PATHED_OK) {
        throw std::runtime_error(std::string("pathed_hip_render: ") + pathed_hip_last_error());
    }
}

}  // namespace pathed
