// Device shading library: frames, samplers, BSDFs, light sampling, environment map.
//
// Each function states the reference code it follows (chellmuth/pathed, file:line).
// Arithmetic keeps the reference's operation order in fp32; the few expressions the
// reference evaluates in double because M_PI is a double constant are evaluated in
// fp32 (DESIGN.md "Arithmetic contract").  Branch structure is written for a SIMT
// wave: short bodies are predicated, material dispatch is a switch the caller can
// make wave-uniform by sorting.
#pragma once

#include "device_scene.h"

namespace pathed {

// What a scene can contain, as a compile-time set.  The kernels are written against every material, light and albedo
// kind the path supports; a kernel instantiated with a NARROWER set compiles the rest out, which is not about the
// instructions skipped (a uniform branch skips them anyway) but about the registers the compiler reserves for code that
// never runs: the fused path kernel lives at 128 VGPRs with spills.  The launch picks the narrowest instantiation whose set
// contains the scene's (pathed_hip.hip: sceneTraits).  Results cannot differ: absent kinds are absent.
template <unsigned MATERIALS, bool ENV, bool TRIANGLE_LIGHTS, bool SPHERES, bool VARYING_ALBEDO, bool PAIRED_TRIG = false, unsigned DISTRIBUTIONS = 3u>
struct SceneTraits {
    static constexpr unsigned materials = MATERIALS;       // bit t: material type t (PATHED_MAT_*) may occur
    static constexpr bool beckmann = (DISTRIBUTIONS & 1u) != 0u;   // microfacet / plastic materials with a Beckmann distribution
    static constexpr bool ggx = (DISTRIBUTIONS & 2u) != 0u;        // ... with a GGX distribution
    static constexpr bool env = ENV;                        // an environment light
    static constexpr bool triangleLights = TRIANGLE_LIGHTS; // emissive triangles
    static constexpr bool spheres = SPHERES;                // sphere primitives (as geometry or as lights)
    static constexpr bool varyingAlbedo = VARYING_ALBEDO;   // checkerboard / image-texture albedo
    // cosf and sinf of one angle through ONE sincosf (cosSin below): the same bits, fewer instructions -- for the fused path
    // kernel's instantiations only: in k_shade the smaller kernel upsets the balance of the two pools
    static constexpr bool pairedTrig = PAIRED_TRIG;
    static constexpr bool has(int type) { return ((MATERIALS >> type) & 1u) != 0u; }
};
typedef SceneTraits<0x7Fu, true, true, true, true> TraitsAll;
// Cornell-box-like scenes: constant-albedo Lambertian surfaces, triangle lights, nothing else
typedef SceneTraits<1u << 0, false, true, false, false, true> TraitsLambertianTriangles;
// the Veach MIS scene: Lambertian + plastic surfaces (constant albedo) lit by emissive spheres
typedef SceneTraits<(1u << 0) | (1u << 3), false, false, true, false, true> TraitsLambertianPlasticSpheres;
// the reference's VolumePathTracer scene (scenes/cornell-medium.json): Lambertian walls, a glass sphere, a passthrough container, triangle lights
typedef SceneTraits<(1u << 0) | (1u << 4) | (1u << 6), false, true, true, false> TraitsLambertianGlassContainer;
// Cornell-box variants with any BSDF (Oren-Nayar, microfacet, plastic, glass, mirror): triangle lights only, no spheres, no
// environment, constant albedo -- what the generic instantiation carries beyond that (environment sampling and lookup, sphere
// lights, checkerboard / texture albedo) is registers the fused kernel does not have (103 spilled dwords)
typedef SceneTraits<0x3Fu, false, true, false, false, true> TraitsTriangleLit;
// mesh scenes lit by the environment alone (no emissive material, no sphere): any material, any albedo
typedef SceneTraits<0x3Fu, true, false, false, true> TraitsEnvironmentOnly;
// [r5] the ladder between "Lambertian only" and "any BSDF" for triangle-lit, constant-albedo scenes (the launch takes the first
// set that contains the scene's): rough surfaces -- Lambertian, Oren-Nayar, microfacet, plastic -- over ONE microfacet
// distribution, and smooth ones -- Lambertian, glass, mirror.  What each leaves out is the other's sampling code and, in the
// rough sets, the other distribution's D / G / sampling: registers the fused kernel spills (103 dwords in TraitsTriangleLit).
typedef SceneTraits<0x0Fu, false, true, false, false, true, 1u> TraitsRoughBeckmann;
typedef SceneTraits<0x0Fu, false, true, false, false, true, 2u> TraitsRoughGgx;
typedef SceneTraits<(1u << 0) | (1u << 4) | (1u << 5), false, true, false, false, true, 3u> TraitsSmooth;


#define PATHED_INV_PI 0.3183098861837907f   /* include/util.h:10 */
#define PATHED_PI 3.14159265358979323846f   /* M_PI narrowed to fp32 */
#define PATHED_TWO_PI 6.283185307179586f    /* include/util.h:11 */

// `2 * M_PI * u` as the reference evaluates it (src/monte_carlo.cpp:28, src/beckmann.cpp:33, src/sphere.cpp:59,97):
// M_PI is a double, so the product is formed in double and narrowed once.  An fp32 product is an ulp off for
// a quarter of the inputs, which cosf / sinf turn into relative errors of 1e-5 near their zeros.
__host__ __device__ inline float twoPiTimes(float u) { return (float)(6.283185307179586476925286766559 * (double)u); }

// ---------------------------------------------------------------------------- rng
// Counter-based stream u(seed, pixel, sample, dimension); replaces the reference's
// shared, unseedable mt19937 (src/random_generator.cpp:4-6).  Two bijective 32-bit
// mixers keyed by pixel and sample, so no two (pixel, sample) pairs share a stream.

__host__ __device__ inline uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

struct Rng {
    uint32_t k0, k1;
    uint32_t dimension;

    __device__ inline float next()
    {
        const uint32_t bits = mix32(k0 + mix32(k1 + dimension * 0x9e3779b9u));
        dimension++;
        // 24 bits scaled into [0, 1 - 2^-23): the range of the reference generator
        return (float)(bits >> 8) * 5.9604638e-08f;
    }
};

__host__ __device__ inline void makeKey(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t *k0, uint32_t *k1)
{
    *k0 = mix32(pixel ^ mix32((uint32_t)seed));
    *k1 = mix32(sample ^ mix32((uint32_t)(seed >> 32) ^ 0x9e3779b9u));
}

// dimension layout: [0,1] pixel jitter (X first, src/camera.cpp:51-52);
// vertex k >= 1 owns 8 dimensions from 2 + 8(k-1): +0..2 BSDF, +3 light choice, +4,+5 light
__device__ inline uint32_t vertexBase(int vertex) { return 2u + 8u * (uint32_t)(vertex - 1); }

// ------------------------------------------------------------------------- frames

struct Frame {
    V3 xAxis, normal, zAxis;  // columns of tangentToWorld
};

// src/transform.cpp:201-219
__device__ inline Frame normalToWorldSpace1(V3 normal)
{
    V3 xAxis;
    if (fabsf(normal.x) > fabsf(normal.y)) {
        xAxis = normalized(v3(-normal.z, 0.f, normal.x));
    } else {
        xAxis = normalized(v3(0.f, -normal.z, normal.y));
    }
    Frame frame;
    frame.xAxis = xAxis;
    frame.normal = normal;
    frame.zAxis = cross(normal, xAxis);
    return frame;
}

// src/transform.cpp:182-199
__device__ inline Frame normalToWorldSpace(V3 normal, V3 rayDirection)
{
    if (normal == rayDirection) { return normalToWorldSpace1(normal); }
    Frame frame;
    frame.xAxis = normalized(cross(normal, rayDirection));
    frame.normal = normal;
    frame.zAxis = normalized(cross(normal, frame.xAxis));
    return frame;
}

// Transform::apply(Vector3), src/transform.cpp:90-102, rows (x.i, n.i, z.i)
__device__ inline V3 toWorld(const Frame &f, V3 l)
{
    return v3(
        f.xAxis.x * l.x + f.normal.x * l.y + f.zAxis.x * l.z,
        f.xAxis.y * l.x + f.normal.y * l.y + f.zAxis.y * l.z,
        f.xAxis.z * l.x + f.normal.z * l.y + f.zAxis.z * l.z);
}

__device__ inline V3 toLocal(const Frame &f, V3 w)
{
    return v3(
        f.xAxis.x * w.x + f.xAxis.y * w.y + f.xAxis.z * w.z,
        f.normal.x * w.x + f.normal.y * w.y + f.normal.z * w.z,
        f.zAxis.x * w.x + f.zAxis.y * w.y + f.zAxis.z * w.z);
}

// include/intersection.h:13-56 (hit only; the kernels never build a miss record)
struct Isect {
    V3 point;
    V3 wo;
    V3 normal;
    V3 shadingNormal;
    float u, v;
    int material;
    int prim;
    Frame frame;
    // normalized(toLocal(frame, wo)): what the Oren-Nayar and microfacet lobes see of wo, formed once per vertex by prepareLobes
    // (not by every evaluation at the vertex); NaN until then, so a vertex that skipped it cannot pass a test
    V3 woLocal;
};

// ---------------------------------------------------- tangent-frame trigonometry
// include/tangent_frame.h:12-107, include/trig.h

__device__ inline float tfCos2Theta(V3 v) { return v.y * v.y; }
__device__ inline float tfSinTheta(V3 v) { return sqrtf(smax(0.f, 1.f - tfCos2Theta(v))); }
__device__ inline float tfSin2Theta(V3 v) { return 1.f - tfCos2Theta(v); }
__device__ inline float tfTanTheta(V3 v) { return tfSinTheta(v) / v.y; }
__device__ inline float tfTan2Theta(V3 v) { return tfSin2Theta(v) / tfCos2Theta(v); }

__device__ inline V3 tfClamp(V3 v)
{
    const float max = 0.9999f;
    if (v.x >= max) { return v3(1.f, 0.f, 0.f); }
    if (v.y >= max) { return v3(0.f, 1.f, 0.f); }
    if (v.z >= max) { return v3(0.f, 0.f, 1.f); }
    if (v.x <= -max) { return v3(-1.f, 0.f, 0.f); }
    if (v.y <= -max) { return v3(0.f, -1.f, 0.f); }
    if (v.z <= -max) { return v3(0.f, 0.f, -1.f); }
    return v;
}

__device__ inline float tfCosPhi(V3 v)
{
    const float sinTheta = tfSinTheta(v);
    if (sinTheta == 0.f) { return 1.f; }
    return clampf(v.x / sinTheta, -1.f, 1.f);
}

__device__ inline float tfSinPhi(V3 v)
{
    const V3 clamped = tfClamp(v);
    const float sinTheta = tfSinTheta(clamped);
    if (sinTheta == 0.f) { return 0.f; }
    return clampf(clamped.z / sinTheta, -1.f, 1.f);
}

__device__ inline float tfCos2Phi(V3 v) { const float c = tfCosPhi(v); return c * c; }
__device__ inline float tfSin2Phi(V3 v) { const float s = tfSinPhi(v); return s * s; }

__device__ inline float sinFromCos(float cosTheta)
{
    const float sin2Theta = 1.f - (cosTheta * cosTheta);
    return sqrtf(smax(0.f, sin2Theta));
}

// ----------------------------------------------------------------------- samplers

// cosf and sinf of ONE angle.  PAIRED: from one argument reduction -- ocml's sincosf returns the bits of its sinf and cosf
// (tools/sincos_identity.hip compares them on the GPU: every float in (-8, 8), every 97th beyond), so it is the
// reference's pair of calls at half the instructions (static: 124 against 239).  Which kernels take it: SceneTraits.
template <bool PAIRED>
__device__ inline void cosSin(float angle, float *cosine, float *sine)
{
    if (PAIRED) { sincosf(angle, sine, cosine); }
    else { *cosine = cosf(angle); *sine = sinf(angle); }
}

// src/monte_carlo.cpp:24-41
template <bool PAIRED = false>
__device__ inline V3 cosineSampleHemisphere(Rng &random)
{
    const float xi1 = random.next();
    const float r = sqrtf(xi1);
    const float phi = twoPiTimes(random.next());
    float cosPhi, sinPhi;
    cosSin<PAIRED>(phi, &cosPhi, &sinPhi);
    const float x = r * cosPhi;
    const float z = r * sinPhi;
    const float y = sqrtf(1.f - xi1);
    return v3(x, y, z);
}

__device__ inline float cosineHemispherePdf(V3 v) { return v.y * PATHED_INV_PI; }

// src/coordinate.cpp:7-18
__device__ inline void cartesianToSpherical(V3 cartesian, float *phi, float *theta)
{
    float p = atan2f(cartesian.z, cartesian.x);
    if (p < 0.f) { p += PATHED_TWO_PI; }
    if (p == PATHED_TWO_PI) { p = 0.f; }
    *phi = p;
    *theta = acosf(clampf(cartesian.y, -1.f, 1.f));
}

// src/coordinate.cpp:25-32
template <bool PAIRED = false>
__device__ inline V3 sphericalToCartesian(float phi, float cosTheta, float sinTheta)
{
    const float y = cosTheta;
    float cosPhi, sinPhi;
    cosSin<PAIRED>(phi, &cosPhi, &sinPhi);
    const float x = sinTheta * cosPhi;
    const float z = sinTheta * sinPhi;
    return v3(x, y, z);
}

// ---------------------------------------------------------------- fresnel / snell

// src/fresnel.cpp:30-64, src/snell.cpp:51-57
__device__ inline float dielectricReflectance(float cosThetaIncident, float etaIncident, float etaTransmitted)
{
    const float sinThetaTransmitted =
        (etaIncident / etaTransmitted) * sqrtf(smax(0.f, 1.f - cosThetaIncident * cosThetaIncident));
    if (sinThetaTransmitted > 1.f) { return 1.f; }

    const float cosThetaTransmitted = sqrtf(smax(0.f, 1.f - sinThetaTransmitted * sinThetaTransmitted));

    const float rParallel =
        (etaTransmitted * cosThetaIncident - etaIncident * cosThetaTransmitted)
        / (etaTransmitted * cosThetaIncident + etaIncident * cosThetaTransmitted);
    const float rPerpendicular =
        (etaIncident * cosThetaIncident - etaTransmitted * cosThetaTransmitted)
        / (etaIncident * cosThetaIncident + etaTransmitted * cosThetaTransmitted);

    return 0.5f * (rParallel * rParallel + rPerpendicular * rPerpendicular);
}

// src/snell.cpp:9-37
__device__ inline bool snellRefract(V3 incidentLocal, V3 *transmittedLocal, float etaIncident, float etaTransmitted)
{
    V3 normal = v3(0.f, 1.f, 0.f);
    if (incidentLocal.y < 0.f) { normal = normal * -1.f; }

    const V3 wIncidentPerpendicular = incidentLocal - (normal * dot(incidentLocal, normal));
    const V3 wTransmittedPerpendicular = -wIncidentPerpendicular * (etaIncident / etaTransmitted);

    const float perpendicularLength = length(wTransmittedPerpendicular);
    const float transmittedPerpendicularLength2 = perpendicularLength * perpendicularLength;
    const float wTransmittedParallelLength = sqrtf(smax(0.f, 1.f - transmittedPerpendicularLength2));
    const V3 wTransmittedParallel = normal * -wTransmittedParallelLength;

    const float cosThetaIncident = incidentLocal.y;
    const float sin2ThetaIncident = smax(0.f, 1.f - (cosThetaIncident * cosThetaIncident));
    const float eta2 = (etaIncident / etaTransmitted) * (etaIncident / etaTransmitted);
    const float sin2ThetaTransmitted = eta2 * sin2ThetaIncident;

    *transmittedLocal = normalized(wTransmittedParallel + wTransmittedPerpendicular);

    return !(sin2ThetaTransmitted >= 1.f);
}

// ---------------------------------------------------------------------- materials

struct BSDFSample {
    V3 wiWorld;
    float pdf;
    Rgb throughput;
};

__device__ inline Rgb matDiffuse(const DMaterial &m) { return rgb(m.diffuse[0], m.diffuse[1], m.diffuse[2]); }
__device__ inline Rgb matEmit(const DMaterial &m) { return rgb(m.emit[0], m.emit[1], m.emit[2]); }

__device__ inline bool isDelta(const DMaterial &m)
{
    return m.type == PATHED_MAT_GLASS || m.type == PATHED_MAT_MIRROR;
}

// src/checkerboard.cpp:9-20
__device__ inline Rgb checkerboardLookup(const DMaterial &m, const Isect &isect)
{
    const int uIndex = (int)floorf(isect.u * m.checkerResU);
    const int vIndex = (int)floorf(isect.v * m.checkerResV);
    if (uIndex % 2 == vIndex % 2) { return rgb(m.checkerOn[0], m.checkerOn[1], m.checkerOn[2]); }
    return rgb(m.checkerOff[0], m.checkerOff[1], m.checkerOff[2]);
}

// Texture::lookup, src/texture.cpp:33-49.  The 8-bit texel -> powf(x / 255, 2.2) step has 256
// possible results per channel; the host applies it once (glibc powf, the function the reference
// calls), so the device reads finished float texels.  The clamp only matters for non-finite uv,
// where the reference reads out of bounds.
__device__ inline Rgb textureLookup(const DMaterial &m, const Isect &isect)
{
    const int width = m.texSize & 0xFFFF, height = (m.texSize >> 16) & 0xFFFF;
    const float u = isect.u - (float)(int)floorf(isect.u);
    const float v = 1.f - (isect.v - (float)(int)floorf(isect.v));
    int x = (int)roundf(u * (float)(width - 1));
    int y = (int)roundf(v * (float)(height - 1));
    x = imin(imax(x, 0), width - 1);
    y = imin(imax(y, 0), height - 1);
    const float4 texel = m.texels[(size_t)y * width + x];
    return rgb(texel.x, texel.y, texel.z);
}

// src/lambertian.cpp:16-40
// (the *Local forms take  wi = normalized(toLocal(frame, wiWorld))  and  wo = normalized(toLocal(frame, isect.wo))  from the
//  caller: materialF / materialSample form them once for all lobes, see MaterialLobes)
template <bool VARYING_ALBEDO = true>
__device__ inline Rgb lambertianFLocal(const DMaterial &m, const Isect &isect, V3 wiWorld, V3 wi, float *pdf)
{
    if (dot(isect.wo, isect.shadingNormal) < 0.f) { *pdf = 0.f; return rgb(0.f); }
    if (dot(wiWorld, isect.shadingNormal) < 0.f) { *pdf = 0.f; return rgb(0.f); }

    *pdf = cosineHemispherePdf(wi);

    if (VARYING_ALBEDO && m.albedoType == PATHED_ALBEDO_CHECKERBOARD) { return checkerboardLookup(m, isect) / PATHED_PI; }
    if (VARYING_ALBEDO && m.albedoType == PATHED_ALBEDO_TEXTURE) { return textureLookup(m, isect) / PATHED_PI; }
    return matDiffuse(m) / PATHED_PI;
}
template <bool VARYING_ALBEDO = true>
__device__ inline Rgb lambertianF(const DMaterial &m, const Isect &isect, V3 wiWorld, float *pdf)
{
    return lambertianFLocal<VARYING_ALBEDO>(m, isect, wiWorld, normalized(toLocal(isect.frame, wiWorld)), pdf);
}

// src/lambertian.cpp:42-58
template <bool VARYING_ALBEDO = true, bool PAIRED = false>
__device__ inline BSDFSample lambertianSample(const DMaterial &m, const Isect &isect, Rng &random)
{
    const V3 localSample = cosineSampleHemisphere<PAIRED>(random);
    const V3 worldSample = toWorld(isect.frame, localSample);
    BSDFSample sample;
    sample.wiWorld = worldSample;
    sample.pdf = cosineHemispherePdf(localSample);
    float ignored;
    sample.throughput = lambertianF<VARYING_ALBEDO>(m, isect, worldSample, &ignored);
    return sample;
}

// src/oren_nayar.cpp:20-67 (pdf = 1, not 0, on the rejected configurations).
// The reference forms  cos(phi_i - phi_o) sin(alpha) tan(beta)  through cartesianToSpherical (src/coordinate.cpp:7-18): two
// atan2f, two acosf, then cosf, sinf, tanf -- seven libm calls per evaluation, two evaluations per vertex, 2.7x the whole
// Lambertian vertex on this chip.  For unit vectors in the upper hemisphere (both y >= 0 here) with sin(theta) = |(x, z)|:
//     cos(phi_i - phi_o) = (x_i x_o + z_i z_o) / (sin(theta_i) sin(theta_o)),
//     alpha = max(theta_i, theta_o) belongs to the smaller y, beta to the larger: sin(alpha) tan(beta) = sin(theta_i) sin(theta_o) / max(y_i, y_o),
// so the product is  (x_i x_o + z_i z_o) / max(y_i, y_o)  -- no transcendental, no square root, and better conditioned than
// sin(acos(y)) near the pole.  It is NOT the reference's bits (its own libm calls are each an ulp or two off the same real
// number); the oracle keeps the reference's statements, and the image-level contract (relL2 <= 2e-3 against the oracle at the
// configuration's resolution) is what this form is held to: measured 1e-6 .. 1e-5 (tests/test_gpu_parity.py).
// PATHED_OREN_NAYAR_TRIG=1 builds the reference's form for that comparison.
#ifndef PATHED_OREN_NAYAR_TRIG
#define PATHED_OREN_NAYAR_TRIG 0
#endif
__device__ inline Rgb orenNayarFLocal(const DMaterial &m, const Isect &isect, V3 localWo, V3 localWi, float *pdf)
{
    if (dot(isect.normal, isect.wo) < 0.f) { *pdf = 1.f; return rgb(0.f); }
    if (dot(isect.shadingNormal, isect.wo) < 0.f) { *pdf = 1.f; return rgb(0.f); }

    if (localWo.y < 0.f) { *pdf = 1.f; return rgb(0.f); }
    if (localWi.y < 0.f) { *pdf = 1.f; return rgb(0.f); }

    *pdf = cosineHemispherePdf(localWi);

#if PATHED_OREN_NAYAR_TRIG
    float phiI, thetaI, phiO, thetaO;
    cartesianToSpherical(localWi, &phiI, &thetaI);
    cartesianToSpherical(localWo, &phiO, &thetaO);
    const float alpha = smax(thetaI, thetaO);
    const float beta = smin(thetaI, thetaO);
    const float rough = smax(0.f, cosf(phiI - phiO)) * sinf(alpha) * tanf(beta);
#else
    const float azimuthal = localWi.x * localWo.x + localWi.z * localWo.z;
    const float highest = smax(localWi.y, localWo.y);
    const float rough = (azimuthal > 0.f && highest > 0.f) ? azimuthal / highest : 0.f;
#endif
    const float throughput = PATHED_INV_PI * (m.orenA + m.orenB * rough);

    return matDiffuse(m) * throughput;
}
__device__ inline Rgb orenNayarF(const DMaterial &m, const Isect &isect, V3 wiWorld, float *pdf)
{
    return orenNayarFLocal(m, isect, normalized(toLocal(isect.frame, isect.wo)), normalized(toLocal(isect.frame, wiWorld)), pdf);
}

// src/oren_nayar.cpp:69-85
template <bool PAIRED = false>
__device__ inline BSDFSample orenNayarSample(const DMaterial &m, const Isect &isect, Rng &random)
{
    const V3 localSample = cosineSampleHemisphere<PAIRED>(random);
    const V3 worldSample = toWorld(isect.frame, localSample);
    BSDFSample sample;
    sample.wiWorld = worldSample;
    sample.pdf = cosineHemispherePdf(localSample);
    float ignored;
    sample.throughput = orenNayarF(m, isect, worldSample, &ignored);
    return sample;
}

// src/beckmann.cpp:50-69, including the TangentFrame::clamp asymmetry between cosPhi
// and sinPhi (include/tangent_frame.h:79-102), which changes D near the pole
__device__ inline float beckmannD(float alpha, V3 wh)
{
    const float tan2Theta = tfTan2Theta(wh);
    if (isinf(tan2Theta)) { return 0.f; }

    const float cos2Theta = tfCos2Theta(wh);
    const float cos4Theta = cos2Theta * cos2Theta;
    const float alpha2 = alpha * alpha;

    const float numerator = expf(
        -tan2Theta * (
            (tfCos2Phi(wh) / alpha2)
            + (tfSin2Phi(wh) / alpha2)));
    const float denominator = PATHED_PI * alpha2 * cos4Theta;

    return numerator / denominator;
}

// src/beckmann.cpp:45-48
__device__ inline float beckmannPdf(float alpha, V3 wh) { return beckmannD(alpha, wh) * fabsf(wh.y); }

// src/beckmann.cpp:71-86
__device__ inline float beckmannLambda(float alphaX, float alphaY, V3 w)
{
    const float absTanTheta = fabsf(tfTanTheta(w));
    if (isinf(absTanTheta)) { return 0.f; }

    const float alpha = sqrtf(tfCos2Phi(w) * alphaX * alphaX + tfSin2Phi(w) * alphaY * alphaY);
    const float a = 1.f / (alpha * absTanTheta);
    if (a >= 1.6f) { return 0.f; }

    return (1 - 1.259f * a + 0.396f * a * a) / (3.535f * a + 2.181f * a * a);
}

// src/beckmann.cpp:88-94
__device__ inline float beckmannG(float alpha, V3 wo, V3 wi)
{
    return 1.f / (1.f + beckmannLambda(alpha, alpha, wo) + beckmannLambda(alpha, alpha, wi));
}

// src/beckmann.cpp:13-43: phi is drawn first, then the tan^2 variate
template <bool PAIRED = false>
__device__ inline V3 beckmannSampleWh(float alpha, Rng &random)
{
    const float phi = twoPiTimes(random.next());

    const float xi = random.next();
    float logXi = logf(xi);
    if (isinf(logXi)) { logXi = 0.f; }
    const float tan2Theta = -alpha * alpha * logXi;

    const float cosTheta = 1.f / sqrtf(1.f + tan2Theta);
    const float sinTheta = sinFromCos(cosTheta);
    return sphericalToCartesian<PAIRED>(phi, cosTheta, sinTheta);
}

// src/ggx.cpp:27-46
__device__ inline float ggxD(float alpha, V3 wh)
{
    const float alpha2 = alpha * alpha;
    const float cos2Theta = tfCos2Theta(wh);
    const float cos4Theta = cos2Theta * cos2Theta;
    const float tan2Theta = tfTan2Theta(wh);
    if (isinf(tan2Theta)) { return 0.f; }
    const float sum = alpha2 + tan2Theta;
    const float denominator = PATHED_PI * cos4Theta * sum * sum;
    return alpha2 / denominator;
}

// src/ggx.cpp:48-58
__device__ inline float ggxG1(float alpha, V3 v)
{
    const float tan2Theta = tfTan2Theta(v);
    if (isinf(tan2Theta)) { return 0.f; }
    const float alpha2 = alpha * alpha;
    const float sqrtTerm = (1 + alpha2 * tan2Theta);
    return 2.f / (1 + sqrtf(sqrtTerm));
}

// src/ggx.cpp:13-25: theta variate first, then phi (the opposite of Beckmann)
template <bool PAIRED = false>
__device__ inline V3 ggxSampleWh(float alpha, Rng &random)
{
    const float xi1 = random.next();
    const float xi2 = random.next();
    const float numerator = alpha * sqrtf(xi1);
    const float denominator = sqrtf(1.f - xi1);
    const float theta = atanf(numerator / denominator);
    const float phi = PATHED_TWO_PI * xi2;
    float cosTheta, sinTheta;
    cosSin<PAIRED>(theta, &cosTheta, &sinTheta);
    return sphericalToCartesian<PAIRED>(phi, cosTheta, sinTheta);
}

// MicrofacetDistribution dispatch (include/microfacet_distribution.h); TRAITS: which distributions the scene set contains
template <typename TRAITS>
__device__ inline bool usesGgx(const DMaterial &m)
{
    return TRAITS::ggx && (!TRAITS::beckmann || m.distribution == PATHED_DIST_GGX);
}
template <typename TRAITS>
__device__ inline float distributionD(const DMaterial &m, V3 wh)
{
    return usesGgx<TRAITS>(m) ? ggxD(m.alpha, wh) : beckmannD(m.alpha, wh);
}
template <typename TRAITS>
__device__ inline float distributionPdf(const DMaterial &m, V3 wh) { return distributionD<TRAITS>(m, wh) * fabsf(wh.y); }
template <typename TRAITS>
__device__ inline float distributionG(const DMaterial &m, V3 wo, V3 wi)
{
    return usesGgx<TRAITS>(m) ? ggxG1(m.alpha, wo) * ggxG1(m.alpha, wi) : beckmannG(m.alpha, wo, wi);
}
template <typename TRAITS>
__device__ inline V3 distributionSampleWh(const DMaterial &m, Rng &random)
{
    return usesGgx<TRAITS>(m) ? ggxSampleWh<TRAITS::pairedTrig>(m.alpha, random) : beckmannSampleWh<TRAITS::pairedTrig>(m.alpha, random);
}

// src/microfacet.cpp:12-57 (Fresnel eta hard-coded to 1.5 at :41)
template <typename TRAITS = TraitsAll>
__device__ inline Rgb microfacetFLocal(const DMaterial &m, const Isect &isect, V3 wiWorld, V3 wo, V3 wi, float *pdf)
{
    if (dot(isect.wo, isect.shadingNormal) < 0.f) { *pdf = 0.f; return rgb(0.f); }
    if (dot(wiWorld, isect.shadingNormal) < 0.f) { *pdf = 0.f; return rgb(0.f); }

    const float cosThetaO = fabsf(wo.y);
    const float cosThetaI = fabsf(wi.y);
    const V3 wh = normalized(wo + wi);

    *pdf = distributionPdf<TRAITS>(m, wh) / (4.f * dot(wo, wh));

    if (cosThetaO == 0.f || cosThetaI == 0.f) { return rgb(0.f); }
    if (wh.x == 0.f && wh.y == 0.f && wh.z == 0.f) { return rgb(0.f); }

    const float cosThetaIncident = clampf(dot(wi, wh), 0.f, 1.f);
    const float fresnel = dielectricReflectance(cosThetaIncident, 1.f, 1.5f);
    const float distribution = distributionD<TRAITS>(m, wh);
    const float masking = distributionG<TRAITS>(m, wo, wi);
    const Rgb albedo = rgb(1.f);

    return albedo * distribution * masking * fresnel / (4 * cosThetaI * cosThetaO);
}
template <typename TRAITS = TraitsAll>
__device__ inline Rgb microfacetF(const DMaterial &m, const Isect &isect, V3 wiWorld, float *pdf)
{
    return microfacetFLocal<TRAITS>(m, isect, wiWorld, normalized(toLocal(isect.frame, isect.wo)), normalized(toLocal(isect.frame, wiWorld)), pdf);
}

// src/microfacet.cpp:59-78
template <typename TRAITS = TraitsAll>
__device__ inline BSDFSample microfacetSample(const DMaterial &m, const Isect &isect, Rng &random)
{
    const V3 wo = toLocal(isect.frame, isect.wo);
    const V3 wh = distributionSampleWh<TRAITS>(m, random);
    const V3 wi = reflect(wo, wh);
    const V3 wiWorld = toWorld(isect.frame, wi);

    BSDFSample sample;
    sample.wiWorld = wiWorld;
    sample.pdf = distributionPdf<TRAITS>(m, wh) / (4.f * dot(wo, wh));
    float ignored;
    sample.throughput = microfacetF<TRAITS>(m, isect, wiWorld, &ignored);
    return sample;
}

// src/plastic.cpp:19-33
template <typename TRAITS = TraitsAll>
__device__ inline Rgb plasticF(const DMaterial &m, const Isect &isect, V3 wiWorld, float *pdf)
{
    float lambertianPDF, microfacetPDF;
    const Rgb f = lambertianF(m, isect, wiWorld, &lambertianPDF) + microfacetF<TRAITS>(m, isect, wiWorld, &microfacetPDF);
    *pdf = (lambertianPDF + microfacetPDF) / 2.f;
    return f;
}

// src/plastic.cpp:35-66
template <typename TRAITS = TraitsAll>
__device__ inline BSDFSample plasticSample(const DMaterial &m, const Isect &isect, Rng &random)
{
    const float xi = random.next();
    BSDFSample sample;
    float otherPDF;
    Rgb otherThroughput;
    if (xi > 0.5f) {
        sample = lambertianSample<true, TRAITS::pairedTrig>(m, isect, random);
        otherThroughput = microfacetF<TRAITS>(m, isect, sample.wiWorld, &otherPDF);
    } else {
        sample = microfacetSample<TRAITS>(m, isect, random);
        otherThroughput = lambertianF(m, isect, sample.wiWorld, &otherPDF);
    }
    BSDFSample out;
    out.wiWorld = sample.wiWorld;
    out.pdf = (sample.pdf + otherPDF) / 2.f;
    out.throughput = sample.throughput + otherThroughput;
    return out;
}

// src/glass.cpp:30-85.  Where the reference exit(1)s (refraction branch taken although
// Snell::refract reported total internal reflection — a rounding corner) the direction
// refract() produced is used.
__device__ inline BSDFSample glassSample(const DMaterial &m, const Isect &isect, Rng &random)
{
    const V3 localWo = toLocal(isect.frame, isect.wo);
    V3 localWi = v3(0.f, 0.f, 0.f);

    float etaIncident = 1.f;
    float etaTransmitted = m.ior;
    if (localWo.y < 0.f) {
        const float swap = etaIncident;
        etaIncident = etaTransmitted;
        etaTransmitted = swap;
    }

    snellRefract(localWo, &localWi, etaIncident, etaTransmitted);

    const float fresnelReflectance = dielectricReflectance(fabsf(localWo.y), etaIncident, etaTransmitted);

    BSDFSample sample;
    if (random.next() < fresnelReflectance) {
        localWi = reflect(localWo, v3(0.f, 1.f, 0.f));
        sample.wiWorld = toWorld(isect.frame, localWi);
        sample.pdf = fresnelReflectance;
        sample.throughput = rgb(fresnelReflectance / fabsf(localWi.y));
    } else {
        const float fresnelTransmittance = 1.f - fresnelReflectance;
        sample.wiWorld = toWorld(isect.frame, localWi);
        sample.pdf = fresnelTransmittance;
        sample.throughput = rgb(fresnelTransmittance / fabsf(localWi.y));
    }
    return sample;
}

// src/mirror.cpp:21-37
__device__ inline BSDFSample mirrorSample(const Isect &isect)
{
    const V3 localWo = toLocal(isect.frame, isect.wo);
    const V3 localWi = reflect(localWo, v3(0.f, 1.f, 0.f));
    BSDFSample sample;
    sample.wiWorld = toWorld(isect.frame, localWi);
    sample.pdf = 1.f;
    sample.throughput = rgb(smax(0.f, 1.f / localWi.y));
    return sample;
}

// BSDF::f / BSDF::sample dispatch (include/bsdf.h: virtual calls in the reference).  [r5] BY LOBE, not by material type: a
// wave whose lanes stand on different materials -- any scene with more than one BSDF -- runs a `switch (m.type)` one case after
// the other, and the cases repeat each other: plastic IS a Lambertian lobe plus a microfacet lobe (src/plastic.cpp:19-66), so
// a wave with Lambertian, microfacet and plastic lanes went through lambertianF and microfacetF twice in an evaluation and
// three times in a sample (and through the cosine-hemisphere and half-vector sampling two or three times).  Here every lobe's
// code occurs ONCE and the lanes of every material that has that lobe run it together:
//     Lambertian lobe: Lambertian, plastic | Oren-Nayar lobe | microfacet lobe: microfacet, plastic.
// Per lane the operations and their order are plasticF's / plasticSample's / the single-lobe functions' own (plastic's random
// number that picks the lobe is drawn first, as there; float addition commutes, so "sampled lobe + other lobe" is one sum):
// the same bits as the per-type switch, which -DPATHED_MATERIAL_SWITCH=1 still builds for that comparison.
#ifndef PATHED_MATERIAL_SWITCH
#define PATHED_MATERIAL_SWITCH 0
#endif
template <typename TRAITS>
struct MaterialLobes {
    bool lambertian, orenNayar, facets, plastic;
    __device__ inline MaterialLobes(const DMaterial &m)
    {
        const int type = m.type;
        plastic = TRAITS::has(PATHED_MAT_PLASTIC) && type == PATHED_MAT_PLASTIC;
        lambertian = (TRAITS::has(PATHED_MAT_LAMBERTIAN) && type == PATHED_MAT_LAMBERTIAN) || plastic;
        orenNayar = TRAITS::has(PATHED_MAT_OREN_NAYAR) && type == PATHED_MAT_OREN_NAYAR;
        facets = (TRAITS::has(PATHED_MAT_MICROFACET) && type == PATHED_MAT_MICROFACET) || plastic;
    }
};

// once per vertex, before materialF / materialSample: wo in the shading frame for the lobes that look at it (the Lambertian lobe
// does not: a wave of Lambertian lanes skips the 3 dot products, square root and 3 divisions)
template <typename TRAITS>
__device__ inline void prepareLobes(const DMaterial &m, Isect &isect)
{
    const MaterialLobes<TRAITS> lobes(m);
    if (lobes.orenNayar || lobes.facets) { isect.woLocal = normalized(toLocal(isect.frame, isect.wo)); }
}

// the lobes of one material at one direction: f and pdf of the material (plastic: src/plastic.cpp:19-33)
template <typename TRAITS>
__device__ inline Rgb lobesF(const MaterialLobes<TRAITS> &lobes, const DMaterial &m, const Isect &isect, V3 wiWorld, float *diffusePdf, float *facetPdf,
                             Rgb *facetsOut)
{
    // every lobe works on the two directions in the shading frame: formed once (3 dot products, a square root and 3 divisions
    // each -- more than the Lambertian and Oren-Nayar lobes' own arithmetic), not once per lobe a divergent wave runs
    const V3 wo = isect.woLocal;   // (prepareLobes: once per vertex)
    const V3 wi = normalized(toLocal(isect.frame, wiWorld));
    Rgb diffuse = rgb(0.f);
    *diffusePdf = 0.f;
    if (lobes.lambertian) { diffuse = lambertianFLocal<TRAITS::varyingAlbedo>(m, isect, wiWorld, wi, diffusePdf); }
    if (lobes.orenNayar) { diffuse = orenNayarFLocal(m, isect, wo, wi, diffusePdf); }
    Rgb facets = rgb(0.f);
    *facetPdf = 0.f;
    if (lobes.facets) { facets = microfacetFLocal<TRAITS>(m, isect, wiWorld, wo, wi, facetPdf); }
    *facetsOut = facets;
    return diffuse;
}

template <typename TRAITS = TraitsAll>
__device__ inline Rgb materialF(const DMaterial &m, const Isect &isect, V3 wiWorld, float *pdf)
{
#if PATHED_MATERIAL_SWITCH
    switch (m.type) {
    case PATHED_MAT_LAMBERTIAN: if (TRAITS::has(PATHED_MAT_LAMBERTIAN)) { return lambertianF<TRAITS::varyingAlbedo>(m, isect, wiWorld, pdf); } break;
    case PATHED_MAT_OREN_NAYAR: if (TRAITS::has(PATHED_MAT_OREN_NAYAR)) { return orenNayarF(m, isect, wiWorld, pdf); } break;
    case PATHED_MAT_MICROFACET: if (TRAITS::has(PATHED_MAT_MICROFACET)) { return microfacetF<TRAITS>(m, isect, wiWorld, pdf); } break;
    case PATHED_MAT_PLASTIC: if (TRAITS::has(PATHED_MAT_PLASTIC)) { return plasticF<TRAITS>(m, isect, wiWorld, pdf); } break;
    default: break;
    }
    *pdf = 0.f;
    return rgb(0.f);  // src/glass.cpp:20-28, src/mirror.cpp:11-19
#else
    const MaterialLobes<TRAITS> lobes(m);
    float diffusePdf, facetPdf;
    Rgb facets;
    const Rgb diffuse = lobesF<TRAITS>(lobes, m, isect, wiWorld, &diffusePdf, &facetPdf, &facets);
    if (lobes.plastic) { *pdf = (diffusePdf + facetPdf) / 2.f; return diffuse + facets; }
    if (lobes.facets) { *pdf = facetPdf; return facets; }
    *pdf = diffusePdf;   // (glass, mirror: 0 and black, src/glass.cpp:20-28, src/mirror.cpp:11-19)
    return diffuse;
#endif
}

template <typename TRAITS = TraitsAll>
__device__ inline BSDFSample materialSample(const DMaterial &m, const Isect &isect, Rng &random)
{
#if PATHED_MATERIAL_SWITCH
    switch (m.type) {
    case PATHED_MAT_LAMBERTIAN: if (TRAITS::has(PATHED_MAT_LAMBERTIAN)) { return lambertianSample<TRAITS::varyingAlbedo, TRAITS::pairedTrig>(m, isect, random); } break;
    case PATHED_MAT_OREN_NAYAR: if (TRAITS::has(PATHED_MAT_OREN_NAYAR)) { return orenNayarSample<TRAITS::pairedTrig>(m, isect, random); } break;
    case PATHED_MAT_MICROFACET: if (TRAITS::has(PATHED_MAT_MICROFACET)) { return microfacetSample<TRAITS>(m, isect, random); } break;
    case PATHED_MAT_PLASTIC: if (TRAITS::has(PATHED_MAT_PLASTIC)) { return plasticSample<TRAITS>(m, isect, random); } break;
    case PATHED_MAT_GLASS: if (TRAITS::has(PATHED_MAT_GLASS)) { return glassSample(m, isect, random); } break;
    default: break;
    }
#else
    const MaterialLobes<TRAITS> lobes(m);
    if (lobes.lambertian || lobes.orenNayar || lobes.facets) {
        // which lobe draws the direction (src/plastic.cpp:35-43: xi > 0.5 the Lambertian one)
        bool fromDiffuse = !lobes.facets, fromFacets = !fromDiffuse;
        if (lobes.plastic) {
            const float xi = random.next();
            fromDiffuse = xi > 0.5f;
            fromFacets = !fromDiffuse;
        }
        V3 wiWorld = isect.wo;
        float sampledPdf = 1.f;
        if (fromDiffuse) {
            // src/lambertian.cpp:42-58, src/oren_nayar.cpp:69-85
            const V3 localSample = cosineSampleHemisphere<TRAITS::pairedTrig>(random);
            wiWorld = toWorld(isect.frame, localSample);
            sampledPdf = cosineHemispherePdf(localSample);
        }
        if (fromFacets) {
            // src/microfacet.cpp:59-78
            const V3 wo = toLocal(isect.frame, isect.wo);
            const V3 wh = distributionSampleWh<TRAITS>(m, random);
            const V3 wi = reflect(wo, wh);
            wiWorld = toWorld(isect.frame, wi);
            sampledPdf = distributionPdf<TRAITS>(m, wh) / (4.f * dot(wo, wh));
        }
        float diffusePdf, facetPdf;
        Rgb facets;
        const Rgb diffuse = lobesF<TRAITS>(lobes, m, isect, wiWorld, &diffusePdf, &facetPdf, &facets);
        BSDFSample sample;
        sample.wiWorld = wiWorld;
        if (lobes.plastic) {
            // src/plastic.cpp:45-66: the pdfs of the two lobes averaged, their values added
            const float otherPdf = fromDiffuse ? facetPdf : diffusePdf;
            sample.pdf = (sampledPdf + otherPdf) / 2.f;
            sample.throughput = diffuse + facets;
        } else {
            sample.pdf = sampledPdf;
            sample.throughput = lobes.facets ? facets : diffuse;
        }
        return sample;
    }
    if (TRAITS::has(PATHED_MAT_GLASS) && m.type == PATHED_MAT_GLASS) { return glassSample(m, isect, random); }
#endif
    // a mirror -- or, in a narrowed instantiation, a material type the scene does not contain (never reached)
    if (TRAITS::has(PATHED_MAT_MIRROR)) { return mirrorSample(isect); }
    BSDFSample none;
    none.wiWorld = isect.wo;
    none.pdf = 1.f;
    none.throughput = rgb(0.f);
    return none;
}

template <typename TRAITS = TraitsAll>
__device__ inline bool isDeltaT(const DMaterial &m)
{
    return (TRAITS::has(PATHED_MAT_GLASS) && m.type == PATHED_MAT_GLASS) || (TRAITS::has(PATHED_MAT_MIRROR) && m.type == PATHED_MAT_MIRROR);
}

// ------------------------------------------------------------------------- shapes

struct SurfaceSample {
    V3 point;
    V3 normal;
    float invPDF;
    int solidAngle;  // Measure::SolidAngle (1) or Measure::Area (0)
};

// src/triangle.cpp:64-71
__device__ inline float triangleArea(V3 p0, V3 p1, V3 p2)
{
    const V3 e1 = p1 - p0;
    const V3 e2 = p2 - p0;
    return fabsf(length(cross(e1, e2)) / 2.f);
}

// src/triangle.cpp:16-38
__device__ inline SurfaceSample triangleSample(V3 p0, V3 p1, V3 p2, Rng &random)
{
    const float r1 = random.next();
    const float r2 = random.next();

    const float a = 1.f - sqrtf(r1);
    const float b = sqrtf(r1) * (1.f - r2);
    const float c = 1.f - a - b;

    SurfaceSample sample;
    sample.point = p0 * a + p1 * b + p2 * c;
    sample.normal = normalized(cross(p1 - p0, p2 - p0));
    sample.invPDF = triangleArea(p0, p1, p2);
    sample.solidAngle = 0;
    return sample;
}

// include/measure.h:13-28
__device__ inline float areaToSolidAngle(float areaPDF, V3 referencePoint, V3 surfacePoint, V3 surfaceNormal)
{
    const V3 surfaceDirection = referencePoint - surfacePoint;
    const V3 surfaceWo = normalized(surfaceDirection);
    const float distance = length(surfaceDirection);
    const float distance2 = distance * distance;
    const float projectedArea = smax(0.f, dot(surfaceNormal, surfaceWo));
    return areaPDF * distance2 / projectedArea;
}

// src/triangle.cpp:48-62
__device__ inline float trianglePdfSolidAngle(V3 p0, V3 p1, V3 p2, V3 point, V3 referencePoint)
{
    const float areaPDF = 1.f / triangleArea(p0, p1, p2);
    const V3 normal = normalized(cross(p1 - p0, p2 - p0));
    return areaToSolidAngle(areaPDF, referencePoint, point, normal);
}

__device__ inline float uniformConePdf(float cosThetaMax)
{
    return 1.f / (2.f * PATHED_PI * (1.f - cosThetaMax));
}

// src/sphere.cpp:54-70
template <bool PAIRED = false>
__device__ inline SurfaceSample sphereSampleArea(V3 center, float radius, Rng &random)
{
    const float z = 1 - 2 * random.next();
    const float r = sqrtf(fmaxf(0, 1 - z * z));
    const float phi = twoPiTimes(random.next());
    float cosPhi, sinPhi;
    cosSin<PAIRED>(phi, &cosPhi, &sinPhi);
    const V3 v = v3(r * cosPhi, r * sinPhi, z);

    SurfaceSample sample;
    sample.point = center + v * radius;
    sample.normal = normalized(v);
    sample.invPDF = 4 * PATHED_PI * radius * radius;
    sample.solidAngle = 0;
    return sample;
}

// src/sphere.cpp:77-128 (samples about the UNTRANSFORMED centre, as the reference does)
template <bool PAIRED = false>
__device__ inline SurfaceSample sphereSample(V3 center, float radius, V3 referencePoint, Rng &random)
{
    const float centerDistance = length(center - referencePoint);
    const float centerDistance2 = centerDistance * centerDistance;
    if (centerDistance <= radius) { return sphereSampleArea<PAIRED>(center, radius, random); }

    const float radius2 = radius * radius;
    const float sin2ThetaMax = radius * radius / centerDistance2;
    const float cosThetaMax = sqrtf(smax(0.f, 1.f - sin2ThetaMax));

    const float xi1 = random.next();
    const float cosTheta = (1.f - xi1) + xi1 * cosThetaMax;
    const float phi = twoPiTimes(random.next());

    const float sinTheta = sinFromCos(cosTheta);
    const float sideOppositeTheta = centerDistance * sinTheta;
    const float sideHelper = sqrtf(smax(0.f, radius * radius - sideOppositeTheta * sideOppositeTheta));
    const float sampleDistance = centerDistance * cosTheta - sideHelper;
    const float sampleDistance2 = sampleDistance * sampleDistance;

    const float cosAlpha = clampf(
        (centerDistance2 + radius2 - sampleDistance2) / (2.f * radius * centerDistance),
        0.f, 1.f);
    const float sinAlpha = sinFromCos(cosAlpha);

    const V3 localSample = sphericalToCartesian<PAIRED>(phi, cosAlpha, sinAlpha);
    const Frame localToWorld = normalToWorldSpace1(normalized(referencePoint - center));
    const V3 worldSample = normalized(toWorld(localToWorld, localSample));

    SurfaceSample sample;
    sample.point = center + worldSample * radius;
    sample.normal = normalized(worldSample);
    sample.invPDF = 1.f / uniformConePdf(cosThetaMax);
    sample.solidAngle = 1;
    return sample;
}

// src/sphere.cpp:130-147 (inside the sphere the reference returns the AREA pdf)
__device__ inline float spherePdfSolidAngle(V3 center, float radius, V3 referencePoint)
{
    const float centerDistance = length(center - referencePoint);
    const float centerDistance2 = centerDistance * centerDistance;
    if (centerDistance <= radius) { return 1.f / (4 * PATHED_PI * radius * radius); }

    const float sin2ThetaMax = radius * radius / centerDistance2;
    const float cosThetaMax = sqrtf(smax(0.f, 1.f - sin2ThetaMax));
    return uniformConePdf(cosThetaMax);
}

// -------------------------------------------------------------- environment light

__device__ inline V3 apply3x3(const float *m, V3 v)
{
    return v3(
        m[0] * v.x + m[1] * v.y + m[2] * v.z,
        m[3] * v.x + m[4] * v.y + m[5] * v.z,
        m[6] * v.x + m[7] * v.y + m[8] * v.z);
}

// Distribution::sample, src/distribution.cpp:35-53: the reference scans linearly for the
// first i with xi <= cdf[i] (Distribution::sample, src/distribution.cpp:35-53, scans linearly); the
// CDF is non-decreasing, so a lower-bound binary search returns the same i.  The guide table
// brackets the answer first: guide[j] = first i with cdf[i] >= j / size, and (j - 1) / size < xi <=
// (j + 2) / size for j = (int)(xi * size) whatever the rounding of the product, so the answer lies
// in [guide[j - 1], guide[j + 2]] -- usually one or two entries instead of log2(size) probes.
__device__ inline int cdfSample(const float *cdf, const int *guide, int size, int empty, float xi, float *pdf)
{
    if (empty) { *pdf = 0.f; return 0; }
    const int bucket = (int)(xi * (float)size);
    const int below = bucket - 1 < 0 ? 0 : (bucket - 1 > size ? size : bucket - 1);
    const int above = bucket + 2 > size ? size : bucket + 2;
    int lo = guide[below], hi = guide[above];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (xi <= cdf[mid]) { hi = mid; } else { lo = mid + 1; }
    }
    if (!(xi <= cdf[lo])) { *pdf = 0.f; return size - 1; }
    *pdf = (lo > 0) ? cdf[lo] - cdf[lo - 1] : cdf[lo];
    return lo;
}

// The same sample through the per-cell record (device_scene.h): one dependent load instead of guide -> search -> pdf.
__device__ inline int cdfSampleRecord(const float4 *records, const float *cdf, int size, float xi, float *pdf)
{
    int bucket = (int)(xi * (float)size);
    bucket = bucket < 0 ? 0 : (bucket > size ? size : bucket);
    const float4 a = records[2 * bucket + 0];
    const float4 b = records[2 * bucket + 1];
    const int lo = floatAsInt(a.x);
    if (lo < 0) { *pdf = 0.f; return 0; }   // empty distribution
    if (xi <= a.w) { *pdf = (lo > 0) ? a.w - a.z : a.w; return lo; }
    if (xi <= b.x) { *pdf = b.x - a.w; return lo + 1; }
    if (xi <= b.y) { *pdf = b.y - b.x; return lo + 2; }
    // rare: many entries inside the cell's window (a flat stretch of the CDF)
    int first = lo + 3, last = floatAsInt(a.y);
    if (first > last) { first = last; }
    while (first < last) {
        const int mid = (first + last) >> 1;
        if (xi <= cdf[mid]) { last = mid; } else { first = mid + 1; }
    }
    if (!(xi <= cdf[first])) { *pdf = 0.f; return size - 1; }
    *pdf = (first > 0) ? cdf[first] - cdf[first - 1] : cdf[first];
    return first;
}

// Distribution::pdf, src/distribution.cpp:55-64
__device__ inline float cdfPdf(const float *cdf, int empty, int index)
{
    if (empty) { return 0.f; }
    if (index == 0) { return cdf[0]; }
    return cdf[index] - cdf[index - 1];
}

// EnvironmentLight::emit(lightWo), src/environment_light.cpp:60-80
__device__ inline Rgb envEmit(const DEnv &env, V3 lightWo)
{
    const V3 direction = -lightWo;
    float phi, theta;
    cartesianToSpherical(normalized(apply3x3(env.worldToMap, direction)), &phi, &theta);

    const float phiCanonical = clampf(phi / PATHED_TWO_PI, 0.f, 1.f);
    const float thetaCanonical = clampf(theta / PATHED_PI, 0.f, 1.f);

    const int phiStep = imin((int)floorf(env.width * phiCanonical), env.width - 1);
    const int thetaStep = imin((int)floorf(env.height * thetaCanonical), env.height - 1);

    const float4 texel = env.rgba[(size_t)thetaStep * env.width + phiStep];
    return rgb(texel.x, texel.y, texel.z) * env.scale;
}

// EnvironmentLight::sample, src/environment_light.cpp:82-105
template <bool PAIRED = false>
__device__ inline SurfaceSample envSample(const DEnv &env, V3 point, Rng &random)
{
    float thetaPDF, phiPDF;
    const int thetaStep = cdfSampleRecord(env.thetaRecords, env.thetaCdf, env.height, random.next(), &thetaPDF);
    const int phiStep = cdfSampleRecord(
        env.phiRecords + (size_t)2 * thetaStep * (env.width + 1), env.phiCdf + (size_t)thetaStep * env.width, env.width,
        random.next(), &phiPDF);

    const float phiCanonical = (phiStep + 0.5f) / env.width;
    const float thetaCanonical = (thetaStep + 0.5f) / env.height;

    const float phi = phiCanonical * PATHED_TWO_PI;
    const float theta = (float)((double)thetaCanonical * 3.14159265358979323846);   // M_PI is a double: src/environment_light.cpp:92

    float cosTheta, sinTheta;
    cosSin<PAIRED>(theta, &cosTheta, &sinTheta);
    const float pdf = thetaPDF * phiPDF * env.width * env.height / (sinTheta * PATHED_TWO_PI * PATHED_PI);

    const V3 direction = apply3x3(env.mapToWorld, sphericalToCartesian<PAIRED>(phi, cosTheta, sinTheta));

    SurfaceSample out;
    out.point = point + direction * 10000.f;
    out.normal = direction * -1.f;
    out.invPDF = 1.f / pdf;
    out.solidAngle = 1;
    return out;
}

// EnvironmentLight::emitPDF, src/environment_light.cpp:117-138
__device__ inline float envEmitPDF(const DEnv &env, V3 direction)
{
    float phi, theta;
    cartesianToSpherical(apply3x3(env.worldToMap, direction), &phi, &theta);

    const float phiCanonical = phi / PATHED_TWO_PI;
    const float thetaCanonical = theta / PATHED_PI;

    const int phiStep = imin((int)floorf(phiCanonical * env.width), env.width - 1);
    const int thetaStep = imin((int)floorf(thetaCanonical * env.height), env.height - 1);

    const float thetaPDF = cdfPdf(env.thetaCdf, env.thetaEmpty, thetaStep);
    const float phiPDF = cdfPdf(env.phiCdf + (size_t)thetaStep * env.width, env.phiEmpty[thetaStep], phiStep);

    return thetaPDF * phiPDF * env.width * env.height / (sinf(theta) * PATHED_TWO_PI * PATHED_PI);
}

// ------------------------------------------------------------------------- camera

// Camera::generateRay(float,float), src/camera.cpp:32-47 (film size precomputed on the host)
__device__ inline void cameraRay(const DCamera &camera, float row, float col, V3 *origin, V3 *direction)
{
    const float zNear = 0.01f;
    const float width = camera.filmWidth;
    const float height = camera.filmHeight;

    const V3 local = normalized(v3(
        width * (col + 0.5f) / camera.resX - width / 2.f,
        height * (row + 0.5f) / camera.resY - height / 2.f,
        -zNear));

    const float *m = camera.m;
    *origin = v3(
        m[0] * 0.f + m[1] * 0.f + m[2] * 0.f + camera.origin[0],
        m[3] * 0.f + m[4] * 0.f + m[5] * 0.f + camera.origin[1],
        m[6] * 0.f + m[7] * 0.f + m[8] * 0.f + camera.origin[2]);
    *direction = v3(
        m[0] * local.x + m[1] * local.y + m[2] * local.z,
        m[3] * local.x + m[4] * local.y + m[5] * local.z,
        m[6] * local.x + m[7] * local.y + m[8] * local.z);
}

}  // namespace pathed
