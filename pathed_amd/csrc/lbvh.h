// On-GPU BVH build.  Two binary builders over 63-bit Morton codes -- a linear
// BVH (Karras 2012: fastest) and PLOC (Meister & Bittner 2018: bottom-up clustering along the Morton
// order with an SAH-built top, within 0-2 % of the host SAH tree's render rate) -- then leaves of up to four triangles and a collapse on the device to
// the SAME 4-wide 128-byte node format the host builder (bvh_build.h) emits, so k_trace walks any
// of the three trees unchanged.
// Stands in for rtcCommitScene (reference src/scene.cpp:39), which is a serial host phase in the
// reference; here 5 M triangles take milliseconds instead of seconds.
//
// Hits do not depend on the tree: the intersector's acceptance rule is order-independent
// (DESIGN.md "Intersector specification"), so an image rendered over the LBVH is bit-identical to
// one rendered over the SAH tree (GPU test); only the traversal cost differs.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

namespace pathed {

struct DeviceBvh {
    float4 *nodes = nullptr;      // nodeCapacity x 8 float4 (hipMalloc'ed; the caller owns it)
    float4 *leafTris = nullptr;   // 3 float4 per triangle, leaf (= Morton) order
    size_t nodeCapacity = 0;
    int nodeCount = 0;
    int maxDepth = 0;             // 4-wide levels
    float buildMs = 0.f;          // device time, HIP events around the whole build
    int rounds = 0;               // PLOC: clustering rounds
};

static const int kDeviceBuilderLbvh = 1;   // PATHED_BVH_LBVH_DEVICE
static const int kDeviceBuilderPloc = 2;   // PATHED_BVH_PLOC_DEVICE

// positions / indices are DEVICE pointers (3 floats per vertex, 3 indices per triangle).
// triangleCount must exceed 4 (smaller meshes go through the host builder).
// spheres: DEVICE pointer to (centre.xyz, radius) per sphere, or null with sphereCount 0; every sphere becomes a leaf of its
// own (bvh_build.h does the same on the host; reference src/sphere.cpp:16-48 gives each one to Embree as a geometry).
hipError_t buildBvhOnDevice(int builder, const float *positions, const uint32_t *indices, uint32_t triangleCount,
                            const float4 *spheres, uint32_t sphereCount, hipStream_t stream, DeviceBvh *out, std::string *error);

}  // namespace pathed
