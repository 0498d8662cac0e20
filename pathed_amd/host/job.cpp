#include "job.h"

#include "integrator.h"

#include <cerrno>
#include <fstream>
#include <iostream>
#include <sys/stat.h>

namespace pathed {

Job::Job(const std::string &jobPath)
    : Job(Json::parseFile(jobPath))
{}

Job::Job(const Json &json)
    : m_json(json),
      m_bounceController(json["startBounce"].asInt(), json["lastBounce"].asInt())
{}

void Job::init()
{
    const std::string directory = outputDirectory();

    const int result = mkdir(directory.c_str(), S_IRWXU | S_IRWXG | S_IROTH | S_IXOTH);
    if (result == -1) {
        if (errno == EEXIST) {
            std::cout << "Output directory already exists: " << directory << std::endl;
        } else {
            std::cout << "Failed to create: " << directory << std::endl;
        }
        if (!force()) { throw std::runtime_error("output directory exists and \"force\" is not set"); }
    }

    std::ofstream report(directory + "/report.json");
    report << m_json.dump(4) << std::endl;
}

std::vector<int> Job::devices() const
{
    const Json &gpus = m_json["gpus"];
    std::vector<int> ids;
    if (gpus.isArray()) {
        for (size_t i = 0; i < gpus.size(); i++) { ids.push_back(gpus[i].asInt()); }
    } else {
        const int count = gpus.isNumber() ? gpus.asInt() : 1;
        for (int i = 0; i < count; i++) { ids.push_back(gpu() + i); }
    }
    if (ids.empty() || ids.size() > 64) { throw std::runtime_error("job: \"gpus\" must name 1..64 devices"); }
    for (int id : ids) { if (id < 0) { throw std::runtime_error("job: negative device id in \"gpus\""); } }
    return ids;
}

std::shared_ptr<Integrator> Job::integrator() const
{
    const std::string name = integratorName();
    if (name == "PathTracer") {
        return std::make_shared<HipPathTracer>(m_bounceController);
    } else if (name == "VolumePathTracer") {
        // src/job.cpp:71-72: participating media behind passthrough containers
        return std::make_shared<HipPathTracer>(m_bounceController, PATHED_INTEGRATOR_VOLUME_PATH_TRACER);
    } else if (name == "DataParallelIntegrator") {
        // the reference's stage-wise integrator needs its external sampler server; its
        // wavefront STRUCTURE is what HipPathTracer implements
        return std::make_shared<HipPathTracer>(m_bounceController);
    }
    throw std::runtime_error("Unimplemented");  // the reference throws "Unimplemented" (src/job.cpp:96)
}

}  // namespace pathed
