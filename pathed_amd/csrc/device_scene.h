// Device-resident scene: flat arrays in HBM, uploaded once by pathed_hip_scene_create.
#pragma once

#include "vecmath.h"

#include <cstdint>

namespace pathed {

// material table entry (96 B), staged in LDS by the shade kernel
struct DMaterial {
    int type;
    int albedoType;
    float orenA, orenB;      // OrenNayar A, B (reference src/oren_nayar.cpp:11-18)
    float diffuse[3];
    float alpha;
    float emit[3];
    float ior;
    float checkerOn[3];
    float checkerResU;
    float checkerOff[3];
    float checkerResV;
    int distribution;        // PATHED_DIST_*: Beckmann or GGX (Microfacet / Plastic)
    int texSize;             // image texture: width | height << 16
    const float4 *texels;    // ... its texels, already through powf(x / 255, 2.2) (reference src/texture.cpp:44-48)
};
static_assert(sizeof(DMaterial) == 96, "DMaterial is staged in LDS as 24 words");

// per-triangle shading record, indexed by ORIGINAL primitive id (128 B = 8 x float4):
//   q0 = (p0, material)  q1 = (p1, uv0.u)  q2 = (p2, uv0.v)
//   q3 = (n0, uv1.u)     q4 = (n1, uv1.v)  q5 = (n2, uv2.u)  q6 = (uv2.v, -, -, -)  q7 = pad
static const int kTriShadeQuads = 8;
static const int kMaxPlainRanges = 8;

struct DSphere {
    float centerWorld[3];
    float radius;
    float centerSample[3];
    int material;
};

struct DLight {
    int kind;   // 0 triangle, 1 sphere, 2 environment
    int index;  // triangle prim id / sphere index
};

struct DCamera {
    float origin[3];
    float m[9];        // cameraToWorld rotation rows (reference lookAt, src/transform.cpp:138-164)
    float filmHeight;  // 2 * tanf(fov / 2) * zNear          (src/camera.cpp:34)
    float filmWidth;   // filmHeight * resX / resY            (src/camera.cpp:35)
    int resX, resY;
};

struct DEnv {
    int width, height;
    const float4 *rgba;        // width*height texels
    const float *thetaCdf;     // height
    const float *phiCdf;       // height*width, row-major
    const int *phiEmpty;       // height flags: 1 = the row has zero weight
    // guide tables (one entry per CDF entry + 1): guide[j] = first i with cdf[i] >= j / size, so a
    // sample xi is bracketed by two table reads instead of a log2(size)-deep chain of dependent loads
    const int *thetaGuide;     // height + 1
    const int *phiGuide;       // height * (width + 1), row-major
    // sampling records, one per guide cell (2 x float4): (first candidate lo, last candidate hi, cdf[lo-1], cdf[lo]) (cdf[lo+1],
    // cdf[lo+2], -, -): ONE 32-byte load resolves a sample that lands on lo .. lo+2 (the usual case) with its pdf;
    // lo = -1 marks an empty distribution.  Same index and pdf as the searches over cdf / guide, which stay as the fallback.
    const float4 *thetaRecords;  // 2 * (height + 1)
    const float4 *phiRecords;    // 2 * height * (width + 1), row-major
    int thetaEmpty;
    float scale;
    float mapToWorld[9];       // 3x3 part, row-major (the reference applies it to directions only)
    float worldToMap[9];
};

struct DScene {
    DCamera camera;

    // intersector
    const float4 *nodes;       // 8 x float4 per inner node (trace.h: layout)
    const float4 *nodesQ;      // 4 x float4 per inner node, same indices: the compressed form (trace.h), or null
    const float4 *leafTris;    // 3 x float4 per leaf-ordered triangle: (v0, prim) (e1, -) (e2, -)
    int nNodes;
    int nTris;
    const DSphere *spheres;
    int nSpheres;
    int nLinearSpheres;        // spheres the traversal does not reach through leaves: tested one by one after it

    // shading
    const float4 *triShade;    // kTriShadeQuads x float4 per original primitive
    const float4 *triCompact;  // one float4 per original primitive: (geometric normal, material); valid for the primitives of the plain ranges
    // primitive-id ranges [begin, end) made of triangles without vertex normals and uvs ("plain": shaded from triCompact
    // alone).  Whole meshes, merged when adjacent; a scene with more ranges than fit keeps the first kMaxPlainRanges.
    int plainBegin[8], plainEnd[8];
    int nPlainRanges;
    const DMaterial *materials;
    int nMaterials;
    const DLight *lights;
    int nLights;
    int hasEnv;
    DEnv env;

    // participating media (the volume integrator, volume.h)
    const struct DMedium *media;
    const int *primMedium;     // per primitive (triangles, then spheres): its surface's internal medium, -1 none
    int nMedia;
};

// path state word (rayD.w)
static const int kStBounceMask = 0xFFFF;
static const int kStEligible = 1 << 16;   // the vertex wants the BSDF-sampling MIS term
static const int kStDelta = 1 << 17;      // the vertex BSDF is a delta lobe
static const int kStContinue = 1 << 18;   // the path may continue past this ray
static const int kStAwait = 1 << 26;      // [r5] the slot waits for the trace kernel (a ray of its own to walk the tree, or an occlusion ray on the list):
                                          // a shade launch that does not follow a trace launch leaves it alone
static const int kStLocal = 1 << 27;      // [r5] the slot's ray was resolved by k_shade itself (it cannot meet the mesh part): hit[] is valid, nothing to trace
static const int kStHold = 1 << 28;       // a ray of the slot is parked in a trace wave: do not re-trace, do not shade
static const int kStDone = 1 << 30;       // slot has no ray in flight

// ray interval of Scene::testIntersect (reference src/scene.cpp:102-103)
#define PATHED_TNEAR 1e-3f
#define PATHED_TFAR 1e5f

}  // namespace pathed
