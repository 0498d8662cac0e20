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

std::shared_ptr<Integrator> Job::integrator() const
{
    const std::string name = integratorName();
    if (name == "PathTracer") {
        return std::make_shared<HipPathTracer>(m_bounceController);
    } else if (name == "DataParallelIntegrator") {
        // the reference's stage-wise integrator needs its external sampler server; its
        // wavefront STRUCTURE is what HipPathTracer implements
        return std::make_shared<HipPathTracer>(m_bounceController);
    }
    throw std::runtime_error("Unimplemented");  // the reference throws "Unimplemented" (src/job.cpp:96)
}

}  // namespace pathed
