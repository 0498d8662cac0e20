// job.json reader, reference include/job.h:13-75 + src/job.cpp:25-97.
// Same keys as the reference (spp, integrator, scene, startBounce, lastBounce,
// output_directory, output_name, showUI, force, width, height); job-file numbers are real
// JSON numbers.  Optional extra keys with defaults, so reference job files run unchanged:
//   "seed" (1), "gpu" (0), "asset_root" (directory paths resolve in),
//   "spp_per_launch" (1024): samples per render call.  A call drains its last paths before it returns (a few ms on a
//               BVH scene: 64 spp per call cost 4 %, 256 and more < 1 %); checkpoints still bound a call,
//   "bvh_builder" ("auto" | "sah" | "lbvh" | "ploc": include/pathed_hip.h PATHED_BVH_*; "auto" = the host SAH build on one
//               GPU, the device PLOC build when several GPUs each need the tree of a mesh of more than a million triangles:
//               N replicas would otherwise run N host builds side by side, seconds of serial time before the first sample),
//   "gpus" (1): a count N -> devices gpu .. gpu+N-1, or an explicit list of device ids (an id may
//               repeat: several replicas on one GPU); the samples of every batch are split over them,
//   "resume" (false): continue from <output_directory>/auto.state if it exists.
#pragma once

#include "bounce_controller.h"
#include "json.h"

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace pathed {

class Integrator;

class Job {
public:
    explicit Job(const std::string &jobPath);
    explicit Job(const Json &json);

    // creates <output_directory>/ and writes report.json (src/job.cpp:33-63)
    void init();

    bool showUI() const { return m_json["showUI"].isBool() && m_json["showUI"].asBool(); }
    bool force() const { return m_json["force"].isBool() && m_json["force"].asBool(); }

    int width() const { return m_json["width"].asInt(); }
    int height() const { return m_json["height"].asInt(); }

    int spp() const
    {
        const int spp = m_json["spp"].asInt();
        return spp > 0 ? spp : 9999999;
    }

    std::string outputDirectory() const { return m_json["output_directory"].asString() + "/"; }
    std::string outputName() const { return m_json["output_name"].isString() ? m_json["output_name"].asString() : ""; }
    std::string scene() const { return m_json["scene"].asString(); }
    std::string integratorName() const { return m_json["integrator"].asString(); }

    int startBounce() const { return m_bounceController.startBounce(); }
    int lastBounce() const { return m_bounceController.lastBounce(); }
    BounceController bounceController() const { return m_bounceController; }

    unsigned long long seed() const { return m_json["seed"].isNumber() ? (unsigned long long)m_json["seed"].asNumber() : 1ull; }
    int sppPerLaunch() const
    {
        if (!m_json["spp_per_launch"].isNumber()) { return 1024; }
        const double value = m_json["spp_per_launch"].asNumber();
        if (!(value >= 1.0) || value > 1e6) { throw std::runtime_error("job: spp_per_launch must be in [1, 1000000]"); }
        return (int)value;
    }
    bool resume() const { return m_json["resume"].isBool() && m_json["resume"].asBool(); }
    std::vector<int> devices() const;
    int gpu() const { return m_json["gpu"].isNumber() ? m_json["gpu"].asInt() : 0; }
    std::string assetRoot() const { return m_json["asset_root"].isString() ? m_json["asset_root"].asString() : ""; }
    std::string bvhBuilder() const { return m_json["bvh_builder"].isString() ? m_json["bvh_builder"].asString() : "auto"; }
    // fan-in of several devices' sums: "rccl" (ONE ncclReduce; on distinct devices a missing RCCL is an ERROR, raised before the
    // scene is loaded; only replicas that share a device use peer copies instead) | "peer-copy" (hipMemcpyPeer + add, on request)
    std::string reduceMethod() const { return m_json["reduce"].isString() ? m_json["reduce"].asString() : "rccl"; }
    // metrics.json: "full" (default) adds rays per sample, Mrays/s, algorithmic bytes and the roofline fraction from a
    // one-sample counting pass after the render; "basic" leaves them out
    std::string metricsLevel() const { return m_json["metrics"].isString() ? m_json["metrics"].asString() : "full"; }

    // string -> class factory, src/job.cpp:65-97; only the hot-path integrators exist here
    std::shared_ptr<Integrator> integrator() const;

    const Json &json() const { return m_json; }

private:
    Json m_json;
    BounceController m_bounceController;
};

}  // namespace pathed
