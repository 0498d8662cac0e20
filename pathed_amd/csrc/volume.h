// Participating media: the reference's VolumePathTracer (src/volume_path_tracer.cpp:14-131) with
// DirectLightingHelper::Ld (src/direct_lighting_helper.cpp:37-187), VolumeHelper (src/volume_helper.cpp),
// HomogeneousMedium (src/homogeneous_medium.cpp) and Scene's two "volumetric" ray queries (src/scene.cpp:225-353,
// 383-424), as ONE persistent kernel: a lane carries a path from camera ray to termination in registers
// (like k_path_small) and walks the 4-wide tree itself for every query, with its stack rows in LDS.
//
// This is the row-f4 integrator, built for correctness first: every query is a per-lane traversal
// (no ray queues, no lane refill), a vertex makes up to four of them.  The wavefront kernels do not know media.
//
// Volume events.  The reference makes Embree skip "containers" (passthrough material + internal medium) through
// an intersection filter that records (t, medium) for every container hit Embree REPORTS (src/scene.cpp:42-83);
// which hits those are depends on Embree's traversal order.  Here, as in the oracle, the events of a closest-hit
// query are the container hits with tnear < t < t(final hit), those of an occlusion query the ones inside the query
// interval; equal t count once, the two nearest are used (the reference asserts there are one or two).
#pragma once

#include "shading.h"
#include "trace.h"

namespace pathed {

struct DMedium {
    float sigmaT[3];
    float sigmaS[3];
    float pad[2];
};

// the two nearest distinct-t events met so far
struct VolumeEvents {
    float t0, t1;
    int m0, m1;
    int count;   // 0, 1 or 2
    // the nearest CONTAINER surface a volumetric closest-hit query skipped: with it the same traversal also answers the
    // regular query on that ray (Scene::testIntersect sees containers), which the path's next segment asks for
    float containerT, containerU, containerV;
    int containerPrim;   // -1: none
};

__device__ inline void eventsClear(VolumeEvents &e)
{
    e.t0 = 0.f; e.t1 = 0.f; e.m0 = -1; e.m1 = -1; e.count = 0;
    e.containerT = 0.f; e.containerU = 0.f; e.containerV = 0.f; e.containerPrim = -1;
}

__device__ inline void eventsAdd(VolumeEvents &e, float t, int medium)
{
    if (e.count >= 1 && t == e.t0) { return; }
    if (e.count >= 2 && t == e.t1) { return; }
    if (e.count == 0) { e.t0 = t; e.m0 = medium; e.count = 1; return; }
    if (t < e.t0) { e.t1 = e.t0; e.m1 = e.m0; e.t0 = t; e.m0 = medium; e.count = 2; return; }
    if (e.count == 1 || t < e.t1) { e.t1 = t; e.m1 = medium; e.count = 2; }
}

// a closest-hit query keeps the events in front of its hit
__device__ inline void eventsClip(VolumeEvents &e, float limit)
{
    if (e.count >= 2 && !(e.t1 < limit)) { e.count = 1; }
    if (e.count >= 1 && !(e.t0 < limit)) { e.count = 0; }
}

static const int kQueryRegular = 0;         // Scene::testIntersect: containers are surfaces like any other
static const int kQueryVolumeClosest = 1;   // Scene::testVolumetricIntersect
static const int kQueryVolumeOccluded = 2;  // Scene::testVolumetricOcclusion

template <typename MaterialTable>
struct VolumeContext {
    TraceGeometry geometry;
    LaneStack stack;
    int maxStack;
    const DScene *scene;
    MaterialTable materials;
    const int *primMedium;    // per primitive (triangles, then spheres): internal medium, -1 none
    const DMedium *media;
};

template <typename MaterialTable>
__device__ inline int primMaterial(const VolumeContext<MaterialTable> &c, int prim)
{
    if (prim < c.scene->nTris) { return reinterpret_cast<const int *>(c.scene->triShade + (size_t)kTriShadeQuads * prim)[3]; }
    return c.scene->spheres[prim - c.scene->nTris].material;
}

// Material::isContainer() && Surface::getInternalMedium() != nullptr (src/scene.cpp:61-64)
template <typename MaterialTable>
__device__ inline bool containerPrim(const VolumeContext<MaterialTable> &c, int prim)
{
    return c.materials[primMaterial(c, prim)].type == PATHED_MAT_PASSTHROUGH && c.primMedium[prim] >= 0;
}

// one primitive hit at distance t against the query: returns true when an occlusion query is decided
template <typename MaterialTable>
__device__ inline bool volumeAccept(const VolumeContext<MaterialTable> &c, int mode, LaneRay &ray, VolumeEvents &events,
                                    float t, float u, float v, int prim)
{
    if (!(t > ray.tnear)) { return false; }
    if (mode != kQueryRegular && containerPrim(c, prim)) {
        if (mode == kQueryVolumeClosest || t <= ray.tfar) { eventsAdd(events, t, c.primMedium[prim]); }
        if (mode == kQueryVolumeClosest) {
            // the regular query's acceptance rule, over the containers only
            const bool closer = (events.containerPrim < 0)
                ? (t <= ray.tfar)
                : (t < events.containerT || (t == events.containerT && prim < events.containerPrim));
            if (closer) { events.containerT = t; events.containerU = u; events.containerV = v; events.containerPrim = prim; }
        }
        return false;
    }
    if (ray.anyHit) {
        if (t <= ray.tfar) { ray.occluded = true; return true; }
    } else {
        const bool closer = (ray.bestPrim < 0)
            ? (t <= ray.best)
            : (t < ray.best || (t == ray.best && prim < ray.bestPrim));
        if (closer) { ray.best = t; ray.bestU = u; ray.bestV = v; ray.bestPrim = prim; }
    }
    return false;
}

template <typename MaterialTable>
__device__ inline bool volumeSphere(const VolumeContext<MaterialTable> &c, int mode, LaneRay &ray, VolumeEvents &events, int index)
{
    const DSphere s = c.geometry.spheres[index];
    float t;
    if (!intersectSphere(ray.o, ray.d, v3(s.centerWorld[0], s.centerWorld[1], s.centerWorld[2]), s.radius, ray.tnear, &t)) { return false; }
    return volumeAccept(c, mode, ray, events, t, 0.f, 0.f, c.geometry.nTris + index);
}

// One whole query on one lane.  Returns "hit" (closest) or "occluded" (occlusion).
static const int kVolumeBlock = 256;   // threads per block of the volume kernel = the LDS stack's row stride (kernels.h: kBlock)

template <int ROWS, typename MaterialTable>
__device__ __forceinline__ bool volumeQuery(const VolumeContext<MaterialTable> &c, int mode, V3 o, V3 d, float tfar,
                                         RayHit *hit, VolumeEvents *eventsOut)
{
    LaneRay ray;
    const bool anyHit = mode == kQueryVolumeOccluded;
    laneRayInit(ray, o, d, PATHED_TNEAR, tfar, anyHit);
    VolumeEvents events;
    eventsClear(events);
    bool done = c.geometry.nNodes == 0;
    while (!done) {
        if (ray.pendingLeaf == 0) {
            done = innerStep<false, ROWS, kVolumeBlock>(c.geometry, c.stack, c.maxStack, ray, nullptr);
        } else {
            const int first = ray.pendingLeaf >> 3;
            const int count = ray.pendingLeaf & 7;
            ray.pendingLeaf = 0;
            bool decided = false;
            if (count == 0) {
                decided = volumeSphere(c, mode, ray, events, first - 1);
            } else {
                for (int k = 0; k < count && !decided; k++) {
                    const float4 t0 = c.geometry.tris[3 * (first + k) + 0];
                    const float4 t1 = c.geometry.tris[3 * (first + k) + 1];
                    const float4 t2 = c.geometry.tris[3 * (first + k) + 2];
                    float t, u, v;
                    if (!intersectTriangle(ray.o, ray.d, v3(t0.x, t0.y, t0.z), v3(t1.x, t1.y, t1.z), v3(t2.x, t2.y, t2.z), &t, &u, &v)) { continue; }
                    decided = volumeAccept(c, mode, ray, events, t, u, v, floatAsInt(t0.w));
                }
            }
            done = decided || popWork<ROWS, kVolumeBlock>(c.stack, ray);
        }
    }
    if (!(ray.anyHit && ray.occluded)) {
        for (int i = 0; i < c.geometry.nSpheres; i++) {   // the spheres that are not in the tree
            if (volumeSphere(c, mode, ray, events, i)) { break; }
        }
    }
    if (mode == kQueryVolumeClosest) { eventsClip(events, ray.best); }
    if (eventsOut) { *eventsOut = events; }
    if (anyHit) { return ray.occluded; }
    hit->t = ray.best;
    hit->u = ray.bestU;
    hit->v = ray.bestV;
    hit->prim = ray.bestPrim;
    return ray.bestPrim >= 0;
}

// HomogeneousMedium::transmittance, src/homogeneous_medium.cpp:14-18
__device__ inline Rgb mediumTransmittance(const DMedium &medium, V3 pointA, V3 pointB)
{
    const float distance = length(pointB - pointA);
    // the three channels of sigma_t are equal (the reference asserts it, scene creation refuses anything else): the three
    // expf of the reference are one value
    const float channel = expf(-medium.sigmaT[0] * distance);
    return rgb(channel, channel, channel);
}

// VolumeHelper::rayTransmission, src/volume_helper.cpp:71-123
__device__ inline Rgb rayTransmission(const DMedium *media, V3 o, V3 d, const VolumeEvents &events, int medium)
{
    Rgb transmittance = rgb(1.f);
    if (events.count == 0) { return transmittance; }
    if (medium >= 0) {
        if (events.count == 1) { transmittance = transmittance * mediumTransmittance(media[medium], o, o + d * events.t0); }
        else { transmittance = transmittance * mediumTransmittance(media[medium], o + d * events.t0, o + d * events.t1); }
    } else {
        if (events.count >= 2) { transmittance = transmittance * mediumTransmittance(media[events.m0], o + d * events.t0, o + d * events.t1); }
        else { transmittance = transmittance * mediumTransmittance(media[events.m0], o, o + d * events.t0); }
    }
    return transmittance;
}

// src/passthrough.cpp:29-43
__device__ inline BSDFSample passthroughSample(const Isect &isect)
{
    const float cosTheta = fabsf(dot(-isect.shadingNormal, -isect.wo));
    BSDFSample sample;
    sample.wiWorld = -isect.wo;
    sample.pdf = 1.f;
    sample.throughput = rgb(1.f) / cosTheta;
    return sample;
}

__device__ inline bool volumeIsDelta(const DMaterial &m) { return m.type == PATHED_MAT_PASSTHROUGH || isDelta(m); }

template <typename TRAITS = TraitsAll>
__device__ inline BSDFSample volumeMaterialSample(const DMaterial &m, const Isect &isect, Rng &random)
{
    if (m.type == PATHED_MAT_PASSTHROUGH) { return passthroughSample(isect); }
    return materialSample<TRAITS>(m, isect, random);
}

// dimensions of the medium's distance sample (+0) and light sample (+1..3) on the segment that ends at `vertex`
__device__ inline uint32_t mediumBase(int vertex) { return 0x4000u + 4u * (uint32_t)(vertex - 1); }

}  // namespace pathed
