/*
 * oracle.cpp — CPU restatement of the reference radiance loop.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT: see oracle.h.  Nothing under pathed_amd/ may
 * include, link or call this file.
 *
 * What it restates (reference = chellmuth/pathed, paths relative to its root):
 *   estimator      src/sample_integrator.cpp:10-78, src/path_tracer.cpp:19-216,
 *                  src/bounce_controller.cpp:14-25, src/integrator.cpp:37-51
 *   scene queries  src/scene.cpp:91-223 (testIntersect), 355-381 (testOcclusion),
 *                  446-502 (sampleDirectLights, lightsPDF, environmentL/PDF),
 *                  include/scene.h:46-81 (LightSample::solidAnglePDF)
 *   frames/camera  include/intersection.h:27-51, src/transform.cpp:138-231,
 *                  src/camera.cpp:32-55
 *   BSDFs          src/lambertian.cpp, src/oren_nayar.cpp, src/microfacet.cpp,
 *                  src/beckmann.cpp, src/plastic.cpp, src/glass.cpp, src/mirror.cpp,
 *                  src/fresnel.cpp, src/snell.cpp, src/checkerboard.cpp,
 *                  include/tangent_frame.h, include/trig.h
 *   lights/shapes  src/triangle.cpp:16-71, src/sphere.cpp:54-147,
 *                  src/environment_light.cpp:14-138, src/distribution.cpp:6-64,
 *                  include/measure.h:13-28, include/mis.h:4-7
 *   samplers       src/monte_carlo.cpp:24-41, src/coordinate.cpp:7-32
 *
 * Parity status
 *   PINNED   every function above that lives in a reference translation unit which
 *            compiles here without Embree: checked value-for-value against golden
 *            vectors produced by the reference's own object code
 *            (oracle/ref_driver.cpp -> oracle/_ref/refdump -> tests/golden/).
 *   UNPINNED ("parity unpinned" for these rows): the control flow of
 *            PathTracer::L/direct and Scene::* (path_tracer.cpp / scene.cpp include
 *            embree3/rtcore.h, which this image lacks, so they cannot be compiled),
 *            Sphere::sample/pdf (sphere.cpp, same reason) and the ray/primitive
 *            intersector itself (Embree, un-vendored).  Those are restated from
 *            source text and checked by analytic invariants and a brute-force
 *            double-precision intersector (oracle_trace_bruteforce).
 *
 * Deliberate, documented deviations from the reference text
 *   - random numbers: the reference's mt19937 is seeded from random_device and shared
 *     racily between threads (src/random_generator.cpp:4-6), so it cannot be
 *     reproduced.  A counter-based stream u(seed, pixel, sample, dimension) replaces
 *     it; the dimension layout follows the reference's consumption order
 *     (SURVEY.md App. A.9).  The HIP kernels implement the same stream.
 *   - expressions the reference evaluates in double because M_PI is a double
 *     constant are evaluated in fp32 here (<= 1 ulp differences in intermediate
 *     angles / pdfs); everything else keeps the reference's operation order.
 *   - the duplicate closest-hit query of direct()/L() (path_tracer.cpp:42-44 vs
 *     :174-175, identical ray) is traced once.
 *   - non-finite samples are dropped (and counted) instead of poisoning the pixel.
 *   - closest-hit ties (equal t) resolve to the lower primitive index so the result
 *     does not depend on BVH shape.
 *   - two shortcuts shared with the HIP kernels, both value-preserving unless an emission or
 *     a light pdf is itself non-finite: a light sample whose BSDF value is exactly black (finite,
 *     positive pdfs) contributes black without its emission being evaluated; a vertex with nothing
 *     pending whose BSDF sample has exactly black throughput ends the sample without the
 *     continuation ray.  (The occlusion query of the first case is still made here, as in the
 *     reference; the kernels skip it.)
 *
 * Build: g++ -std=c++17 -O2 -ffp-contract=off -fopenmp (see oracle/Makefile); the
 * intersection helpers use explicit fmaf so they round like the GPU's v_fma_f32.
 */
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

thread_local std::string g_error;

const float kInvPi = 0.3183098861837907f;      /* include/util.h:10 INV_PI      */
const float kPi = 3.14159265358979323846f;     /* M_PI narrowed to fp32         */
const float kTwoPi = 6.283185307179586f;       /* include/util.h:11 M_TWO_PI    */

/* `2 * M_PI * u` exactly as the reference forms it (src/monte_carlo.cpp:28, src/beckmann.cpp:33, src/sphere.cpp:59,97):
 * in double, narrowed once.  The kernels do the same (pathed_amd/csrc/shading.h: twoPiTimes). */
inline float twoPiTimes(float u) { return (float)(6.283185307179586476925286766559 * (double)u); }

/* ------------------------------------------------------------------ vectors */

struct Vec3 {
    float x, y, z;
};

inline Vec3 v3(float x, float y, float z) { Vec3 v = { x, y, z }; return v; }
inline Vec3 operator+(Vec3 a, Vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(Vec3 a, Vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator-(Vec3 a) { return v3(-a.x, -a.y, -a.z); }
inline Vec3 operator*(Vec3 a, float t) { return v3(a.x * t, a.y * t, a.z * t); }
inline bool operator==(Vec3 a, Vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

/* src/vector.cpp:18-21 */
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

/* src/vector.cpp:37-44 */
inline Vec3 cross(Vec3 a, Vec3 b)
{
    return v3((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x));
}

/* src/vector.cpp:28-35 */
inline float length(Vec3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }

/* src/vector.cpp:46-62 */
inline Vec3 normalized(Vec3 a)
{
    const float norm = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    return v3(a.x / norm, a.y / norm, a.z / norm);
}

/* src/vector.cpp:64-67: (normal * dot(normal) * 2) - *this */
inline Vec3 reflect(Vec3 v, Vec3 normal) { return (normal * dot(v, normal) * 2.f) - v; }

struct Color {
    float r, g, b;
};

inline Color col(float r, float g, float b) { Color c = { r, g, b }; return c; }
inline Color col(float v) { return col(v, v, v); }
inline Color operator+(Color a, Color b) { return col(a.r + b.r, a.g + b.g, a.b + b.b); }
inline Color operator*(Color a, Color b) { return col(a.r * b.r, a.g * b.g, a.b * b.b); }
inline Color operator*(Color a, float t) { return col(a.r * t, a.g * t, a.b * t); }
/* src/color.cpp:128-135: division multiplies by the reciprocal */
inline Color operator/(Color a, float t) { const float inv = 1.f / t; return a * inv; }
inline bool isBlack(Color c) { return c.r == 0.f && c.g == 0.f && c.b == 0.f; }

inline float clampf(float value, float lowest, float highest)
{
    return std::min(highest, std::max(value, lowest)); /* include/util.h:49-51 */
}

/* ---------------------------------------------------------------------- rng */

inline uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

struct PathKey {
    uint32_t k0, k1;
};

inline PathKey makeKey(uint64_t seed, uint32_t pixel, uint32_t sample)
{
    PathKey key;
    key.k0 = mix32(pixel ^ mix32((uint32_t)seed));
    key.k1 = mix32(sample ^ mix32((uint32_t)(seed >> 32) ^ 0x9e3779b9u));
    return key;
}

inline float keyedUniform(PathKey key, uint32_t dimension)
{
    const uint32_t bits = mix32(key.k0 + mix32(key.k1 + dimension * 0x9e3779b9u));
    /* 24 random bits scaled into [0, 1 - 2^-23), the reference generator's range
     * (uniform_real_distribution<float>(0, 1 - epsilon), src/random_generator.cpp:4-6) */
    return (float)(bits >> 8) * 5.9604638e-08f;
}

/* Stand-in for `RandomGenerator &random`: either the keyed stream with an explicit
 * dimension cursor, or a scripted list of numbers (function-level tests). */
struct Rng {
    PathKey key;
    uint32_t dimension;
    const float *script;
    int scriptLength;
    int scriptCursor;

    float next()
    {
        if (script) {
            const float u = (scriptCursor < scriptLength) ? script[scriptCursor] : 0.5f;
            scriptCursor++;
            return u;
        }
        return keyedUniform(key, dimension++);
    }
};

inline Rng keyedRng(uint64_t seed, uint32_t pixel, uint32_t sample)
{
    Rng rng;
    rng.key = makeKey(seed, pixel, sample);
    rng.dimension = 0;
    rng.script = nullptr;
    rng.scriptLength = 0;
    rng.scriptCursor = 0;
    return rng;
}

inline Rng scriptedRng(const float *script, int length)
{
    Rng rng;
    rng.key.k0 = rng.key.k1 = 0;
    rng.dimension = 0;
    rng.script = script;
    rng.scriptLength = length;
    rng.scriptCursor = 0;
    return rng;
}

/* dimension layout, SURVEY.md App. A.9: [0,1] pixel jitter; vertex k >= 1 owns
 * 8 dimensions starting at 2 + 8(k-1): +0..2 BSDF, +3 light choice, +4,+5 light */
inline uint32_t vertexBase(int vertex) { return 2u + 8u * (uint32_t)(vertex - 1); }

/* ------------------------------------------------------------------- frames */

struct Frame {
    Vec3 xAxis, normal, zAxis; /* columns of tangentToWorld */
};

/* src/transform.cpp:201-219 */
Frame normalToWorldSpace1(Vec3 normal)
{
    Vec3 xAxis;
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

/* src/transform.cpp:182-199 */
Frame normalToWorldSpace(Vec3 normal, Vec3 rayDirection)
{
    if (normal == rayDirection) { return normalToWorldSpace1(normal); }
    Frame frame;
    frame.xAxis = normalized(cross(normal, rayDirection));
    frame.normal = normal;
    frame.zAxis = normalized(cross(normal, frame.xAxis));
    return frame;
}

/* Transform::apply(Vector3) with rows (x.i, n.i, z.i): src/transform.cpp:90-102 */
inline Vec3 toWorld(const Frame &f, Vec3 l)
{
    return v3(
        f.xAxis.x * l.x + f.normal.x * l.y + f.zAxis.x * l.z,
        f.xAxis.y * l.x + f.normal.y * l.y + f.zAxis.y * l.z,
        f.xAxis.z * l.x + f.normal.z * l.y + f.zAxis.z * l.z);
}

/* the transposed matrix */
inline Vec3 toLocal(const Frame &f, Vec3 w)
{
    return v3(
        f.xAxis.x * w.x + f.xAxis.y * w.y + f.xAxis.z * w.z,
        f.normal.x * w.x + f.normal.y * w.y + f.normal.z * w.z,
        f.zAxis.x * w.x + f.zAxis.y * w.y + f.zAxis.z * w.z);
}

/* include/intersection.h:13-56 */
struct Intersection {
    bool hit;
    float t;
    Vec3 point;
    Vec3 wo;
    Vec3 normal;
    Vec3 shadingNormal;
    float u, v; /* uv */
    int material;
    int prim; /* surface: triangle index, or n_triangles + sphere index */
    Frame frame;
};

/* ------------------------------------------------------- tangent-frame trig */
/* include/tangent_frame.h:12-107, include/trig.h */

inline float tfCos2Theta(Vec3 v) { return v.y * v.y; }
inline float tfSinTheta(Vec3 v) { return sqrtf(std::max(0.f, 1.f - tfCos2Theta(v))); }
inline float tfSin2Theta(Vec3 v) { return 1.f - tfCos2Theta(v); }
inline float tfTanTheta(Vec3 v) { return tfSinTheta(v) / v.y; }
inline float tfTan2Theta(Vec3 v) { return tfSin2Theta(v) / tfCos2Theta(v); }

inline Vec3 tfClamp(Vec3 v)
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

inline float tfCosPhi(Vec3 v)
{
    const float sinTheta = tfSinTheta(v);
    if (sinTheta == 0.f) { return 1.f; }
    return clampf(v.x / sinTheta, -1.f, 1.f);
}

inline float tfSinPhi(Vec3 v)
{
    const Vec3 clamped = tfClamp(v);
    const float sinTheta = tfSinTheta(clamped);
    if (sinTheta == 0.f) { return 0.f; }
    return clampf(clamped.z / sinTheta, -1.f, 1.f);
}

inline float tfCos2Phi(Vec3 v) { return tfCosPhi(v) * tfCosPhi(v); }
inline float tfSin2Phi(Vec3 v) { return tfSinPhi(v) * tfSinPhi(v); }

inline float sinFromCos(float cosTheta)
{
    const float sin2Theta = 1.f - (cosTheta * cosTheta);
    return sqrtf(std::max(0.f, sin2Theta));
}

/* ----------------------------------------------------------------- samplers */

/* src/monte_carlo.cpp:24-41 */
Vec3 cosineSampleHemisphere(Rng &random)
{
    const float xi1 = random.next();
    const float r = sqrtf(xi1);
    const float phi = twoPiTimes(random.next());
    const float x = r * cosf(phi);
    const float z = r * sinf(phi);
    const float y = sqrtf(1.f - xi1);
    return v3(x, y, z);
}

inline float cosineHemispherePdf(Vec3 v) { return v.y * kInvPi; }

/* src/coordinate.cpp:7-18 */
void cartesianToSpherical(Vec3 cartesian, float *phi, float *theta)
{
    *phi = atan2f(cartesian.z, cartesian.x);
    if (*phi < 0.f) { *phi += kTwoPi; }
    if (*phi == kTwoPi) { *phi = 0.f; }
    *theta = acosf(clampf(cartesian.y, -1.f, 1.f));
}

/* src/coordinate.cpp:25-32 */
inline Vec3 sphericalToCartesian(float phi, float cosTheta, float sinTheta)
{
    const float y = cosTheta;
    const float x = sinTheta * cosf(phi);
    const float z = sinTheta * sinf(phi);
    return v3(x, y, z);
}

/* --------------------------------------------------------- fresnel / snell */

/* src/fresnel.cpp:30-64, src/snell.cpp:51-57 */
float dielectricReflectance(float cosThetaIncident, float etaIncident, float etaTransmitted)
{
    const float sinThetaTransmitted =
        (etaIncident / etaTransmitted) * sqrtf(std::max(0.f, 1.f - cosThetaIncident * cosThetaIncident));
    if (sinThetaTransmitted > 1.f) { return 1.f; }

    const float cosThetaTransmitted = sqrtf(std::max(0.f, 1.f - sinThetaTransmitted * sinThetaTransmitted));

    const float rParallel =
        (etaTransmitted * cosThetaIncident - etaIncident * cosThetaTransmitted)
        / (etaTransmitted * cosThetaIncident + etaIncident * cosThetaTransmitted);
    const float rPerpendicular =
        (etaIncident * cosThetaIncident - etaTransmitted * cosThetaTransmitted)
        / (etaIncident * cosThetaIncident + etaTransmitted * cosThetaTransmitted);

    return 0.5f * (rParallel * rParallel + rPerpendicular * rPerpendicular);
}

/* src/snell.cpp:9-37 */
bool snellRefract(Vec3 incidentLocal, Vec3 *transmittedLocal, float etaIncident, float etaTransmitted)
{
    Vec3 normal = v3(0.f, 1.f, 0.f);
    if (incidentLocal.y < 0.f) { normal = normal * -1.f; }

    const Vec3 wIncidentPerpendicular = incidentLocal - (normal * dot(incidentLocal, normal));
    const Vec3 wTransmittedPerpendicular = -wIncidentPerpendicular * (etaIncident / etaTransmitted);

    const float transmittedPerpendicularLength2 =
        length(wTransmittedPerpendicular) * length(wTransmittedPerpendicular);
    const float wTransmittedParallelLength = sqrtf(std::max(0.f, 1.f - transmittedPerpendicularLength2));
    const Vec3 wTransmittedParallel = normal * -wTransmittedParallelLength;

    const float cosThetaIncident = incidentLocal.y;
    const float sin2ThetaIncident = std::max(0.f, 1.f - (cosThetaIncident * cosThetaIncident));
    const float eta2 = (etaIncident / etaTransmitted) * (etaIncident / etaTransmitted);
    const float sin2ThetaTransmitted = eta2 * sin2ThetaIncident;

    *transmittedLocal = normalized(wTransmittedParallel + wTransmittedPerpendicular);

    return !(sin2ThetaTransmitted >= 1.f);
}

/* ---------------------------------------------------------------- materials */

struct BSDFSample {
    Vec3 wiWorld;
    float pdf;
    Color throughput;
};

struct Material {
    int type;
    int albedoType;
    Color diffuse;
    Color emit;
    Color checkerOn, checkerOff;
    float checkerResU, checkerResV;
    float orenA, orenB;
    float alpha;
    float ior;
    int distribution; /* PATHED_DIST_* */
    /* image texture (PATHED_ALBEDO_TEXTURE): 8-bit RGB as stbi_load returns it; not owned */
    const unsigned char *texData;
    int texWidth, texHeight;
};

Material materialFromDesc(const PathedMaterial &m)
{
    Material out;
    out.type = m.type;
    out.albedoType = m.albedo_type;
    out.texData = nullptr;
    out.texWidth = 0;
    out.texHeight = 0;
    out.diffuse = col(m.diffuse[0], m.diffuse[1], m.diffuse[2]);
    out.emit = col(m.emit[0], m.emit[1], m.emit[2]);
    out.checkerOn = col(m.checker_on[0], m.checker_on[1], m.checker_on[2]);
    out.checkerOff = col(m.checker_off[0], m.checker_off[1], m.checker_off[2]);
    out.checkerResU = m.checker_res[0];
    out.checkerResV = m.checker_res[1];
    /* src/oren_nayar.cpp:11-18 */
    const float sigma2 = m.sigma * m.sigma;
    out.orenA = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
    out.orenB = (0.45f * sigma2) / (sigma2 + 0.09f);
    out.alpha = m.alpha;
    out.distribution = m.distribution;
    out.ior = m.ior;
    /* only Lambertian carries emission in the reference's parser
     * (src/scene_parser.cpp:574-667: every other ctor passes Color(0)) */
    if (m.type != PATHED_MAT_LAMBERTIAN) { out.emit = col(0.f); }
    return out;
}

inline bool isDelta(const Material &m)
{
    /* glass.h:23, mirror.h, passthrough.h:23 */
    return m.type == PATHED_MAT_GLASS || m.type == PATHED_MAT_MIRROR || m.type == PATHED_MAT_PASSTHROUGH;
}

/* Material::isContainer, include/passthrough.h:25 */
inline bool isContainer(const Material &m) { return m.type == PATHED_MAT_PASSTHROUGH; }

/* src/checkerboard.cpp:9-20 */
Color checkerboardLookup(const Material &m, const Intersection &isect)
{
    const int uIndex = (int)floorf(isect.u * m.checkerResU);
    const int vIndex = (int)floorf(isect.v * m.checkerResV);
    if (uIndex % 2 == vIndex % 2) { return m.checkerOn; }
    return m.checkerOff;
}

/* Texture::lookup, src/texture.cpp:33-49.  The clamp only matters for non-finite uv, where
 * the reference reads out of bounds. */
Color textureLookup(const Material &m, const Intersection &isect)
{
    /* Handle wrapping (:37-39) */
    const float u = isect.u - (int)floorf(isect.u);
    const float v = 1.f - (isect.v - (int)floorf(isect.v));

    int x = (int)roundf(u * (m.texWidth - 1));
    int y = (int)roundf(v * (m.texHeight - 1));
    x = std::min(std::max(x, 0), m.texWidth - 1);
    y = std::min(std::max(y, 0), m.texHeight - 1);

    const unsigned char *texel = m.texData + 3 * ((size_t)y * m.texWidth + x);
    return col(powf(texel[0] / 255.f, 2.2f), powf(texel[1] / 255.f, 2.2f), powf(texel[2] / 255.f, 2.2f));
}

/* src/lambertian.cpp:16-40 */
Color lambertianF(const Material &m, const Intersection &isect, Vec3 wiWorld, float *pdf)
{
    if (dot(isect.wo, isect.shadingNormal) < 0.f) { *pdf = 0.f; return col(0.f); }
    if (dot(wiWorld, isect.shadingNormal) < 0.f) { *pdf = 0.f; return col(0.f); }

    const Vec3 wi = normalized(toLocal(isect.frame, wiWorld));
    *pdf = cosineHemispherePdf(wi);

    if (m.albedoType == PATHED_ALBEDO_CHECKERBOARD) { return checkerboardLookup(m, isect) / kPi; }
    if (m.albedoType == PATHED_ALBEDO_TEXTURE) { return textureLookup(m, isect) / kPi; }
    return m.diffuse / kPi;
}

/* src/lambertian.cpp:42-58 */
BSDFSample lambertianSample(const Material &m, const Intersection &isect, Rng &random)
{
    const Vec3 localSample = cosineSampleHemisphere(random);
    const Vec3 worldSample = toWorld(isect.frame, localSample);
    BSDFSample sample;
    sample.wiWorld = worldSample;
    sample.pdf = cosineHemispherePdf(localSample);
    float ignored;
    sample.throughput = lambertianF(m, isect, worldSample, &ignored);
    return sample;
}

/* src/oren_nayar.cpp:20-67 */
Color orenNayarF(const Material &m, const Intersection &isect, Vec3 wiWorld, float *pdf)
{
    if (dot(isect.normal, isect.wo) < 0.f) { *pdf = 1.f; return col(0.f); }
    if (dot(isect.shadingNormal, isect.wo) < 0.f) { *pdf = 1.f; return col(0.f); }

    const Vec3 localWo = normalized(toLocal(isect.frame, isect.wo));
    const Vec3 localWi = normalized(toLocal(isect.frame, wiWorld));

    if (localWo.y < 0.f) { *pdf = 1.f; return col(0.f); }
    if (localWi.y < 0.f) { *pdf = 1.f; return col(0.f); }

    float phiI, thetaI, phiO, thetaO;
    cartesianToSpherical(localWi, &phiI, &thetaI);
    cartesianToSpherical(localWo, &phiO, &thetaO);

    const float alpha = std::max(thetaI, thetaO);
    const float beta = std::min(thetaI, thetaO);

    *pdf = cosineHemispherePdf(localWi);

    const float throughput = kInvPi * (
        m.orenA
        + m.orenB * std::max(0.f, cosf(phiI - phiO))
            * sinf(alpha)
            * tanf(beta));

    return m.diffuse * throughput;
}

/* src/oren_nayar.cpp:69-85 */
BSDFSample orenNayarSample(const Material &m, const Intersection &isect, Rng &random)
{
    const Vec3 localSample = cosineSampleHemisphere(random);
    const Vec3 worldSample = toWorld(isect.frame, localSample);
    BSDFSample sample;
    sample.wiWorld = worldSample;
    sample.pdf = cosineHemispherePdf(localSample);
    float ignored;
    sample.throughput = orenNayarF(m, isect, worldSample, &ignored);
    return sample;
}

/* src/beckmann.cpp:50-69 */
float beckmannD(float alpha, Vec3 wh)
{
    const float tan2Theta = tfTan2Theta(wh);
    if (std::isinf(tan2Theta)) { return 0.f; }

    const float cos2Theta = tfCos2Theta(wh);
    const float cos4Theta = cos2Theta * cos2Theta;
    const float alpha2 = alpha * alpha;

    const float numerator = expf(
        -tan2Theta * (
            (tfCos2Phi(wh) / alpha2)
            + (tfSin2Phi(wh) / alpha2)));
    const float denominator = kPi * alpha2 * cos4Theta;

    return numerator / denominator;
}

/* src/beckmann.cpp:45-48 */
inline float beckmannPdf(float alpha, Vec3 wh) { return beckmannD(alpha, wh) * fabsf(wh.y); }

/* src/beckmann.cpp:71-86 */
float beckmannLambda(float alphaX, float alphaY, Vec3 w)
{
    const float absTanTheta = fabsf(tfTanTheta(w));
    if (std::isinf(absTanTheta)) { return 0.f; }

    const float alpha = sqrtf(tfCos2Phi(w) * alphaX * alphaX + tfSin2Phi(w) * alphaY * alphaY);
    const float a = 1.f / (alpha * absTanTheta);
    if (a >= 1.6f) { return 0.f; }

    return (1 - 1.259f * a + 0.396f * a * a) / (3.535f * a + 2.181f * a * a);
}

/* src/beckmann.cpp:88-94 */
inline float beckmannG(float alpha, Vec3 wo, Vec3 wi)
{
    return 1.f / (1.f + beckmannLambda(alpha, alpha, wo) + beckmannLambda(alpha, alpha, wi));
}

/* src/beckmann.cpp:13-43: phi is drawn first, then the tan^2 variate */
Vec3 beckmannSampleWh(float alpha, Rng &random)
{
    const float phi = twoPiTimes(random.next());

    const float xi = random.next();
    float logXi = logf(xi);
    if (std::isinf(logXi)) { logXi = 0.f; }
    const float tan2Theta = -alpha * alpha * logXi;

    const float cosTheta = 1.f / sqrtf(1.f + tan2Theta);
    const float sinTheta = sinFromCos(cosTheta);
    return sphericalToCartesian(phi, cosTheta, sinTheta);
}

/* src/ggx.cpp:27-46 */
float ggxD(float alpha, Vec3 wh)
{
    const float alpha2 = alpha * alpha;
    const float cos2Theta = tfCos2Theta(wh);
    const float cos4Theta = cos2Theta * cos2Theta;
    const float tan2Theta = tfTan2Theta(wh);
    if (std::isinf(tan2Theta)) { return 0.f; }
    const float sum = alpha2 + tan2Theta;
    const float denominator = kPi * cos4Theta * sum * sum;
    return alpha2 / denominator;
}

/* src/ggx.cpp:48-58 */
float ggxG1(float alpha, Vec3 v)
{
    const float tan2Theta = tfTan2Theta(v);
    if (std::isinf(tan2Theta)) { return 0.f; }
    const float alpha2 = alpha * alpha;
    const float sqrtTerm = (1 + alpha2 * tan2Theta);
    return 2.f / (1 + sqrtf(sqrtTerm));
}

/* src/ggx.cpp:13-25: theta variate first, then phi (the opposite of Beckmann) */
Vec3 ggxSampleWh(float alpha, Rng &random)
{
    const float xi1 = random.next();
    const float xi2 = random.next();
    const float numerator = alpha * sqrtf(xi1);
    const float denominator = sqrtf(1.f - xi1);
    const float theta = atanf(numerator / denominator);
    const float phi = kTwoPi * xi2;
    return sphericalToCartesian(phi, cosf(theta), sinf(theta));
}

/* MicrofacetDistribution dispatch (include/microfacet_distribution.h) */
inline float distributionD(const Material &m, Vec3 wh)
{
    return m.distribution == PATHED_DIST_GGX ? ggxD(m.alpha, wh) : beckmannD(m.alpha, wh);
}
inline float distributionPdf(const Material &m, Vec3 wh) { return distributionD(m, wh) * fabsf(wh.y); }
inline float distributionG(const Material &m, Vec3 wo, Vec3 wi)
{
    return m.distribution == PATHED_DIST_GGX ? ggxG1(m.alpha, wo) * ggxG1(m.alpha, wi) : beckmannG(m.alpha, wo, wi);
}
inline Vec3 distributionSampleWh(const Material &m, Rng &random)
{
    return m.distribution == PATHED_DIST_GGX ? ggxSampleWh(m.alpha, random) : beckmannSampleWh(m.alpha, random);
}

/* src/microfacet.cpp:12-57 */
Color microfacetF(const Material &m, const Intersection &isect, Vec3 wiWorld, float *pdf)
{
    const Vec3 wo = normalized(toLocal(isect.frame, isect.wo));
    const Vec3 wi = normalized(toLocal(isect.frame, wiWorld));

    if (dot(isect.wo, isect.shadingNormal) < 0.f) { *pdf = 0.f; return col(0.f); }
    if (dot(wiWorld, isect.shadingNormal) < 0.f) { *pdf = 0.f; return col(0.f); }

    const float cosThetaO = fabsf(wo.y);
    const float cosThetaI = fabsf(wi.y);
    const Vec3 wh = normalized(wo + wi);

    *pdf = distributionPdf(m, wh) / (4.f * dot(wo, wh));

    if (cosThetaO == 0.f || cosThetaI == 0.f) { return col(0.f); }
    if (wh.x == 0.f && wh.y == 0.f && wh.z == 0.f) { return col(0.f); }

    const float cosThetaIncident = clampf(dot(wi, wh), 0.f, 1.f);
    const float fresnel = dielectricReflectance(cosThetaIncident, 1.f, 1.5f);
    const float distribution = distributionD(m, wh);
    const float masking = distributionG(m, wo, wi);
    const Color albedo = col(1.f);

    return albedo * distribution * masking * fresnel / (4 * cosThetaI * cosThetaO);
}

/* src/microfacet.cpp:59-78 */
BSDFSample microfacetSample(const Material &m, const Intersection &isect, Rng &random)
{
    const Vec3 wo = toLocal(isect.frame, isect.wo);
    const Vec3 wh = distributionSampleWh(m, random);
    const Vec3 wi = reflect(wo, wh);
    const Vec3 wiWorld = toWorld(isect.frame, wi);

    BSDFSample sample;
    sample.wiWorld = wiWorld;
    sample.pdf = distributionPdf(m, wh) / (4.f * dot(wo, wh));
    float ignored;
    sample.throughput = microfacetF(m, isect, wiWorld, &ignored);
    return sample;
}

/* src/plastic.cpp:19-33 */
Color plasticF(const Material &m, const Intersection &isect, Vec3 wiWorld, float *pdf)
{
    float lambertianPDF, microfacetPDF;
    const Color f = lambertianF(m, isect, wiWorld, &lambertianPDF) + microfacetF(m, isect, wiWorld, &microfacetPDF);
    *pdf = (lambertianPDF + microfacetPDF) / 2.f;
    return f;
}

/* src/plastic.cpp:35-66 */
BSDFSample plasticSample(const Material &m, const Intersection &isect, Rng &random)
{
    const float xi = random.next();
    if (xi > 0.5f) {
        BSDFSample sample = lambertianSample(m, isect, random);
        float microfacetPDF;
        const Color microfacetThroughput = microfacetF(m, isect, sample.wiWorld, &microfacetPDF);
        BSDFSample out;
        out.wiWorld = sample.wiWorld;
        out.pdf = (sample.pdf + microfacetPDF) / 2.f;
        out.throughput = sample.throughput + microfacetThroughput;
        return out;
    }
    BSDFSample sample = microfacetSample(m, isect, random);
    float lambertianPDF;
    const Color lambertianThroughput = lambertianF(m, isect, sample.wiWorld, &lambertianPDF);
    BSDFSample out;
    out.wiWorld = sample.wiWorld;
    out.pdf = (sample.pdf + lambertianPDF) / 2.f;
    out.throughput = sample.throughput + lambertianThroughput;
    return out;
}

/* src/glass.cpp:30-85.  The reference exit(1)s when the refraction branch is taken
 * although Snell::refract reported total internal reflection (a rounding corner);
 * here that branch simply uses the direction refract() produced. */
BSDFSample glassSample(const Material &m, const Intersection &isect, Rng &random)
{
    const Vec3 localWo = toLocal(isect.frame, isect.wo);
    Vec3 localWi = v3(0.f, 0.f, 0.f);

    float etaIncident = 1.f;
    float etaTransmitted = m.ior;
    if (localWo.y < 0.f) { std::swap(etaIncident, etaTransmitted); }

    snellRefract(localWo, &localWi, etaIncident, etaTransmitted);

    const float fresnelReflectance = dielectricReflectance(fabsf(localWo.y), etaIncident, etaTransmitted);

    BSDFSample sample;
    if (random.next() < fresnelReflectance) {
        localWi = reflect(localWo, v3(0.f, 1.f, 0.f));
        sample.wiWorld = toWorld(isect.frame, localWi);
        sample.pdf = fresnelReflectance;
        sample.throughput = col(fresnelReflectance / fabsf(localWi.y));
    } else {
        const float fresnelTransmittance = 1.f - fresnelReflectance;
        sample.wiWorld = toWorld(isect.frame, localWi);
        sample.pdf = fresnelTransmittance;
        sample.throughput = col(fresnelTransmittance / fabsf(localWi.y));
    }
    return sample;
}

/* src/mirror.cpp:21-37 */
BSDFSample mirrorSample(const Intersection &isect)
{
    const Vec3 localWo = toLocal(isect.frame, isect.wo);
    const Vec3 localWi = reflect(localWo, v3(0.f, 1.f, 0.f));
    BSDFSample sample;
    sample.wiWorld = toWorld(isect.frame, localWi);
    sample.pdf = 1.f;
    sample.throughput = col(std::max(0.f, 1.f / localWi.y));
    return sample;
}

/* src/passthrough.cpp:29-43: straight on, pdf 1, throughput 1 / |cos| */
BSDFSample passthroughSample(const Intersection &isect)
{
    const float cosTheta = fabsf(dot(-isect.shadingNormal, -isect.wo));   /* WorldFrame::absCosTheta */
    BSDFSample sample;
    sample.wiWorld = -isect.wo;
    sample.pdf = 1.f;
    sample.throughput = col(1.f) / cosTheta;
    return sample;
}

Color materialF(const Material &m, const Intersection &isect, Vec3 wiWorld, float *pdf)
{
    switch (m.type) {
    case PATHED_MAT_LAMBERTIAN: return lambertianF(m, isect, wiWorld, pdf);
    case PATHED_MAT_OREN_NAYAR: return orenNayarF(m, isect, wiWorld, pdf);
    case PATHED_MAT_MICROFACET: return microfacetF(m, isect, wiWorld, pdf);
    case PATHED_MAT_PLASTIC: return plasticF(m, isect, wiWorld, pdf);
    default: *pdf = 0.f; return col(0.f); /* glass.cpp:20-28, mirror.cpp:11-19 */
    }
}

BSDFSample materialSample(const Material &m, const Intersection &isect, Rng &random)
{
    switch (m.type) {
    case PATHED_MAT_LAMBERTIAN: return lambertianSample(m, isect, random);
    case PATHED_MAT_OREN_NAYAR: return orenNayarSample(m, isect, random);
    case PATHED_MAT_MICROFACET: return microfacetSample(m, isect, random);
    case PATHED_MAT_PLASTIC: return plasticSample(m, isect, random);
    case PATHED_MAT_GLASS: return glassSample(m, isect, random);
    case PATHED_MAT_PASSTHROUGH: return passthroughSample(isect);
    default: return mirrorSample(isect);
    }
}

/* ------------------------------------------------------------------- shapes */

enum Measure { SolidAngle, Area };

struct SurfaceSample {
    Vec3 point;
    Vec3 normal;
    float invPDF;
    Measure measure;
};

struct Triangle {
    Vec3 p0, p1, p2;
    Vec3 n0, n1, n2;
    float uv0[2], uv1[2], uv2[2];
    int material;
};

/* src/triangle.cpp:64-71 */
float triangleArea(const Triangle &tri)
{
    const Vec3 e1 = tri.p1 - tri.p0;
    const Vec3 e2 = tri.p2 - tri.p0;
    return fabsf(length(cross(e1, e2)) / 2.f);
}

/* src/triangle.cpp:16-38 */
SurfaceSample triangleSample(const Triangle &tri, Rng &random)
{
    const float r1 = random.next();
    const float r2 = random.next();

    const float a = 1.f - sqrtf(r1);
    const float b = sqrtf(r1) * (1.f - r2);
    const float c = 1.f - a - b;

    SurfaceSample sample;
    sample.point = tri.p0 * a + tri.p1 * b + tri.p2 * c;
    sample.normal = normalized(cross(tri.p1 - tri.p0, tri.p2 - tri.p0));
    sample.invPDF = triangleArea(tri);
    sample.measure = Area;
    return sample;
}

/* include/measure.h:13-28 */
float areaToSolidAngle(float areaPDF, Vec3 referencePoint, Vec3 surfacePoint, Vec3 surfaceNormal)
{
    const Vec3 surfaceDirection = referencePoint - surfacePoint;
    const Vec3 surfaceWo = normalized(surfaceDirection);
    const float distance = length(surfaceDirection);
    const float distance2 = distance * distance;
    const float projectedArea = std::max(0.f, dot(surfaceNormal, surfaceWo));
    return areaPDF * distance2 / projectedArea;
}

/* src/triangle.cpp:48-62 */
float trianglePdfSolidAngle(const Triangle &tri, Vec3 point, Vec3 referencePoint)
{
    const float areaPDF = 1.f / triangleArea(tri);
    const Vec3 normal = normalized(cross(tri.p1 - tri.p0, tri.p2 - tri.p0));
    return areaToSolidAngle(areaPDF, referencePoint, point, normal);
}

struct Sphere {
    Vec3 centerWorld;   /* what Embree intersects (sphere.cpp:30-35)          */
    Vec3 centerSample;  /* untransformed m_center used by sample/pdf          */
    float radius;
    int material;
};

inline float uniformConePdf(float cosThetaMax) { return 1.f / (2.f * kPi * (1.f - cosThetaMax)); }

/* src/sphere.cpp:54-70 */
SurfaceSample sphereSampleArea(const Sphere &s, Rng &random)
{
    const float z = 1 - 2 * random.next();
    const float r = sqrtf(fmaxf(0, 1 - z * z));
    const float phi = twoPiTimes(random.next());
    const Vec3 v = v3(r * cosf(phi), r * sinf(phi), z);

    SurfaceSample sample;
    sample.point = s.centerSample + v * s.radius;
    sample.normal = normalized(v);
    sample.invPDF = 4 * kPi * s.radius * s.radius;
    sample.measure = Area;
    return sample;
}

/* src/sphere.cpp:77-128 */
SurfaceSample sphereSample(const Sphere &s, Vec3 referencePoint, Rng &random)
{
    const float centerDistance = length(s.centerSample - referencePoint);
    const float centerDistance2 = centerDistance * centerDistance;
    if (centerDistance <= s.radius) { return sphereSampleArea(s, random); }

    const float radius2 = s.radius * s.radius;
    const float sin2ThetaMax = s.radius * s.radius / centerDistance2;
    const float cosThetaMax = sqrtf(std::max(0.f, 1.f - sin2ThetaMax));

    const float xi1 = random.next();
    const float cosTheta = (1.f - xi1) + xi1 * cosThetaMax;
    const float phi = twoPiTimes(random.next());

    const float sinTheta = sinFromCos(cosTheta);
    const float sideOppositeTheta = centerDistance * sinTheta;
    const float sideHelper = sqrtf(std::max(0.f, s.radius * s.radius - sideOppositeTheta * sideOppositeTheta));
    const float sampleDistance = centerDistance * cosTheta - sideHelper;
    const float sampleDistance2 = sampleDistance * sampleDistance;

    const float cosAlpha = clampf(
        (centerDistance2 + radius2 - sampleDistance2) / (2.f * s.radius * centerDistance),
        0.f, 1.f);
    const float sinAlpha = sinFromCos(cosAlpha);

    const Vec3 localSample = sphericalToCartesian(phi, cosAlpha, sinAlpha);
    const Frame localToWorld = normalToWorldSpace1(normalized(referencePoint - s.centerSample));
    const Vec3 worldSample = normalized(toWorld(localToWorld, localSample));

    SurfaceSample sample;
    sample.point = s.centerSample + worldSample * s.radius;
    sample.normal = normalized(worldSample);
    sample.invPDF = 1.f / uniformConePdf(cosThetaMax);
    sample.measure = SolidAngle;
    return sample;
}

/* src/sphere.cpp:130-147 (the inside-the-sphere branch returns the AREA pdf, as the
 * reference's TODO notes) */
float spherePdfSolidAngle(const Sphere &s, Vec3 referencePoint)
{
    const float centerDistance = length(s.centerSample - referencePoint);
    const float centerDistance2 = centerDistance * centerDistance;
    if (centerDistance <= s.radius) { return 1.f / (4 * kPi * s.radius * s.radius); }

    const float sin2ThetaMax = s.radius * s.radius / centerDistance2;
    const float cosThetaMax = sqrtf(std::max(0.f, 1.f - sin2ThetaMax));
    return uniformConePdf(cosThetaMax);
}

/* ------------------------------------------------------ environment light */

/* src/distribution.cpp:6-64 */
struct Distribution {
    bool empty;
    std::vector<float> cdf;

    void build(const float *values, size_t size)
    {
        empty = false;
        cdf.assign(size, 0.f);
        float sum = 0.f;
        for (size_t i = 0; i < size; i++) { sum += values[i]; }
        if (sum == 0.f) { empty = true; return; }
        for (size_t i = 0; i < size; i++) {
            cdf[i] = values[i] / sum;
            if (i > 0) { cdf[i] += cdf[i - 1]; }
        }
        cdf[size - 1] = 1.f;
    }

    /* first i with xi <= cdf[i] (linear scan in the reference) */
    int sample(float *pdf, Rng &random) const
    {
        const float xi = random.next();
        if (empty) { *pdf = 0.f; return 0; } /* reference asserts */
        for (size_t i = 0; i < cdf.size(); i++) {
            if (xi <= cdf[i]) {
                *pdf = (i > 0) ? cdf[i] - cdf[i - 1] : cdf[i];
                return (int)i;
            }
        }
        *pdf = 0.f;
        return (int)cdf.size() - 1;
    }

    float pdf(int index) const
    {
        if (empty) { return 0.f; }
        if (index == 0) { return cdf[0]; }
        return cdf[(size_t)index] - cdf[(size_t)index - 1];
    }
};

struct Mat3x4 {
    float m[3][4];
    Vec3 applyVector(Vec3 v) const /* src/transform.cpp:90-102 */
    {
        return v3(
            m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z,
            m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
            m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z);
    }
};

struct EnvLight {
    int width, height;
    std::vector<float> data; /* RGBA */
    float scale;
    Mat3x4 mapToWorld, worldToMap;
    Distribution thetaDistribution;
    std::vector<Distribution> phiDistributions;

    /* src/environment_light.cpp:14-53 */
    void build(const PathedEnvLight &desc)
    {
        width = desc.width;
        height = desc.height;
        scale = desc.scale;
        data.assign(desc.rgba, desc.rgba + (size_t)4 * width * height);
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 4; j++) {
                mapToWorld.m[i][j] = desc.map_to_world[4 * i + j];
                worldToMap.m[i][j] = desc.world_to_map[4 * i + j];
            }
        }
        std::vector<float> luminance((size_t)width * height, 0.f);
        for (size_t i = 0; i < (size_t)width * height; i++) {
            luminance[i] += data[4 * i + 0];
            luminance[i] += data[4 * i + 1];
            luminance[i] += data[4 * i + 2];
        }
        std::vector<float> thetaData((size_t)height, 0.f);
        phiDistributions.resize((size_t)height);
        for (int thetaStep = 0; thetaStep < height; thetaStep++) {
            float thetaSum = 0.f;
            for (int phiStep = 0; phiStep < width; phiStep++) {
                thetaSum += luminance[(size_t)thetaStep * width + phiStep];
            }
            phiDistributions[(size_t)thetaStep].build(&luminance[(size_t)thetaStep * width], (size_t)width);
            thetaData[(size_t)thetaStep] = thetaSum;
        }
        thetaDistribution.build(thetaData.data(), (size_t)height);
    }

    /* src/environment_light.cpp:60-80 */
    Color emit(Vec3 lightWo) const
    {
        const Vec3 direction = -lightWo;
        float phi, theta;
        cartesianToSpherical(normalized(worldToMap.applyVector(direction)), &phi, &theta);

        const float phiCanonical = clampf(phi / kTwoPi, 0.f, 1.f);
        const float thetaCanonical = clampf(theta / kPi, 0.f, 1.f);

        const int phiStep = std::min((int)floorf(width * phiCanonical), width - 1);
        const int thetaStep = std::min((int)floorf(height * thetaCanonical), height - 1);

        const size_t index = (size_t)thetaStep * width + phiStep;
        return col(data[4 * index + 0], data[4 * index + 1], data[4 * index + 2]) * scale;
    }

    /* src/environment_light.cpp:82-105 */
    SurfaceSample sample(Vec3 point, Rng &random) const
    {
        float thetaPDF, phiPDF;
        const int thetaStep = thetaDistribution.sample(&thetaPDF, random);
        const int phiStep = phiDistributions[(size_t)thetaStep].sample(&phiPDF, random);

        const float phiCanonical = (phiStep + 0.5f) / width;
        const float thetaCanonical = (thetaStep + 0.5f) / height;

        const float phi = phiCanonical * kTwoPi;
        const float theta = (float)((double)thetaCanonical * 3.14159265358979323846);   /* M_PI is a double: src/environment_light.cpp:92 */

        const float pdf = thetaPDF * phiPDF * width * height / (sinf(theta) * kTwoPi * kPi);

        const Vec3 direction = mapToWorld.applyVector(sphericalToCartesian(phi, cosf(theta), sinf(theta)));

        SurfaceSample out;
        out.point = point + direction * 10000.f;
        out.normal = direction * -1.f;
        out.invPDF = 1.f / pdf;
        out.measure = SolidAngle;
        return out;
    }

    /* src/environment_light.cpp:117-138 */
    float emitPDF(Vec3 direction) const
    {
        float phi, theta;
        cartesianToSpherical(worldToMap.applyVector(direction), &phi, &theta);

        const float phiCanonical = phi / kTwoPi;
        const float thetaCanonical = theta / kPi;

        const int phiStep = std::min((int)floorf(phiCanonical * width), width - 1);
        const int thetaStep = std::min((int)floorf(thetaCanonical * height), height - 1);

        const float thetaPDF = thetaDistribution.pdf(thetaStep);
        const float phiPDF = phiDistributions[(size_t)thetaStep].pdf(phiStep);

        return thetaPDF * phiPDF * width * height / (sinf(theta) * kTwoPi * kPi);
    }
};

/* ------------------------------------------------------------------- camera */

struct Camera {
    Vec3 origin;
    float m[3][3]; /* cameraToWorld rotation, columns (s*x, y, dir) */
    float verticalFOV;
    int resolutionX, resolutionY;

    /* src/transform.cpp:138-164 (lookAt) */
    void build(const PathedCamera &desc)
    {
        const Vec3 source = v3(desc.origin[0], desc.origin[1], desc.origin[2]);
        const Vec3 target = v3(desc.target[0], desc.target[1], desc.target[2]);
        const Vec3 up = v3(desc.up[0], desc.up[1], desc.up[2]);

        const Vec3 direction = normalized(source - target);
        const Vec3 xAxis = normalized(cross(normalized(up), direction));
        const Vec3 yAxis = cross(direction, xAxis);
        const float sign = desc.flip_handedness ? -1.f : 1.f;

        m[0][0] = sign * xAxis.x; m[0][1] = yAxis.x; m[0][2] = direction.x;
        m[1][0] = sign * xAxis.y; m[1][1] = yAxis.y; m[1][2] = direction.y;
        m[2][0] = sign * xAxis.z; m[2][1] = yAxis.z; m[2][2] = direction.z;
        origin = source;
        verticalFOV = desc.vertical_fov;
        resolutionX = desc.width;
        resolutionY = desc.height;
    }

    /* src/camera.cpp:32-47 */
    void generateRay(float row, float col, Vec3 *rayOrigin, Vec3 *rayDirection) const
    {
        const float zNear = 0.01f;
        const float height = 2 * tanf(verticalFOV / 2) * zNear;
        const float width = height * resolutionX / resolutionY;

        const Vec3 direction = normalized(v3(
            width * (col + 0.5f) / resolutionX - width / 2.f,
            height * (row + 0.5f) / resolutionY - height / 2.f,
            -zNear));

        /* Transform::apply(Point3(0,0,0)) and apply(Vector3) */
        *rayOrigin = v3(
            m[0][0] * 0.f + m[0][1] * 0.f + m[0][2] * 0.f + origin.x,
            m[1][0] * 0.f + m[1][1] * 0.f + m[1][2] * 0.f + origin.y,
            m[2][0] * 0.f + m[2][1] * 0.f + m[2][2] * 0.f + origin.z);
        *rayDirection = v3(
            m[0][0] * direction.x + m[0][1] * direction.y + m[0][2] * direction.z,
            m[1][0] * direction.x + m[1][1] * direction.y + m[1][2] * direction.z,
            m[2][0] * direction.x + m[2][1] * direction.y + m[2][2] * direction.z);
    }
};

/* --------------------------------------------------------------- intersector
 * Stands in for Embree.  The per-primitive arithmetic below is a SPECIFICATION
 * shared with the HIP kernels (same operations, same fmaf placement), so that both
 * produce bit-identical t/u/v for the same ray and primitive:
 *   cross(a,b).x = fmaf(a.y, b.z, -(a.z*b.y))          (likewise y, z)
 *   dot(a,b)     = fmaf(a.x, b.x, fmaf(a.y, b.y, a.z*b.z))
 */

inline Vec3 xcross(Vec3 a, Vec3 b)
{
    return v3(
        fmaf(a.y, b.z, -(a.z * b.y)),
        fmaf(a.z, b.x, -(a.x * b.z)),
        fmaf(a.x, b.y, -(a.y * b.x)));
}

inline float xdot(Vec3 a, Vec3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }

struct HitRecord {
    float t, u, v;
    int prim;
};

/* Moeller-Trumbore on (v0, e1 = v1 - v0, e2 = v2 - v0); u,v follow Embree's
 * convention P = (1-u-v) v0 + u v1 + v v2.  The inside test is done in
 * "det units" (no division for the triangles a ray misses); one division for a hit. */
inline bool intersectTriangle(Vec3 o, Vec3 d, Vec3 v0, Vec3 e1, Vec3 e2, float *t, float *u, float *v)
{
    const Vec3 pvec = xcross(d, e2);
    const float det = xdot(e1, pvec);
    const Vec3 tvec = o - v0;
    const float uScaled = xdot(tvec, pvec);
    const Vec3 qvec = xcross(tvec, e1);
    const float vScaled = xdot(d, qvec);
    if (det > 0.f) {
        if (!(uScaled >= 0.f && vScaled >= 0.f && uScaled + vScaled <= det)) { return false; }
    } else if (det < 0.f) {
        if (!(uScaled <= 0.f && vScaled <= 0.f && uScaled + vScaled >= det)) { return false; }
    } else {
        return false; /* parallel, degenerate or NaN */
    }
    const float inv = 1.f / det;
    *t = xdot(e2, qvec) * inv;
    *u = uScaled * inv;
    *v = vScaled * inv;
    return true;
}

/* ray / sphere by projection onto the ray (robust for small far spheres):
 * returns the front hit if it is beyond tnear, else the back hit */
inline bool intersectSphere(Vec3 o, Vec3 d, Vec3 center, float radius, float tnear, float *t)
{
    const Vec3 c0 = center - o;
    const float dd = xdot(d, d);
    const float projection = xdot(c0, d) / dd;
    const Vec3 perpendicular = c0 - d * projection;
    const float l2 = xdot(perpendicular, perpendicular);
    const float r2 = radius * radius;
    if (!(l2 <= r2)) { return false; }
    const float td = sqrtf((r2 - l2) / dd);
    const float tFront = projection - td;
    const float tBack = projection + td;
    *t = (tFront > tnear) ? tFront : tBack;
    return true;
}

struct BvhNode {
    float bmin[3], bmax[3];
    int left, right; /* children, or -1 */
    int first, count; /* leaf range into primOrder */
};

struct Counters {
    uint64_t cameraSamples, closestRays, shadowRays, boxTests, triTests, dropped, vertices;
    uint64_t shadowRaysNeeded; /* occlusion queries whose light sample has a non-black unoccluded contribution */
};

struct Bvh {
    std::vector<BvhNode> nodes;
    std::vector<int> primOrder;

    struct BuildPrim {
        float bmin[3], bmax[3], centroid[3];
        int index;
    };

    void build(const std::vector<Triangle> &tris)
    {
        nodes.clear();
        primOrder.clear();
        if (tris.empty()) { return; }
        std::vector<BuildPrim> prims(tris.size());
        for (size_t i = 0; i < tris.size(); i++) {
            const Vec3 p[3] = { tris[i].p0, tris[i].p1, tris[i].p2 };
            BuildPrim &bp = prims[i];
            bp.index = (int)i;
            for (int a = 0; a < 3; a++) {
                const float c[3] = { (&p[0].x)[a], (&p[1].x)[a], (&p[2].x)[a] };
                bp.bmin[a] = std::min(c[0], std::min(c[1], c[2]));
                bp.bmax[a] = std::max(c[0], std::max(c[1], c[2]));
                bp.centroid[a] = 0.5f * (bp.bmin[a] + bp.bmax[a]);
            }
        }
        nodes.reserve(2 * tris.size());
        buildRange(prims, 0, prims.size());
        primOrder.resize(prims.size());
        for (size_t i = 0; i < prims.size(); i++) { primOrder[i] = prims[i].index; }
    }

    /* object-median split on the widest centroid axis; leaves hold <= 2 triangles.
     * Deliberately different from the product's binned-SAH builder: results must not
     * depend on the tree. */
    int buildRange(std::vector<BuildPrim> &prims, size_t begin, size_t end)
    {
        BvhNode node;
        float cmin[3], cmax[3];
        for (int a = 0; a < 3; a++) {
            node.bmin[a] = cmin[a] = std::numeric_limits<float>::infinity();
            node.bmax[a] = cmax[a] = -std::numeric_limits<float>::infinity();
        }
        for (size_t i = begin; i < end; i++) {
            for (int a = 0; a < 3; a++) {
                node.bmin[a] = std::min(node.bmin[a], prims[i].bmin[a]);
                node.bmax[a] = std::max(node.bmax[a], prims[i].bmax[a]);
                cmin[a] = std::min(cmin[a], prims[i].centroid[a]);
                cmax[a] = std::max(cmax[a], prims[i].centroid[a]);
            }
        }
        /* pad so that a hit computed in fp32 on a face lying in a box plane survives */
        for (int a = 0; a < 3; a++) {
            const float pad = 1e-5f * std::max(1.f, std::max(fabsf(node.bmin[a]), fabsf(node.bmax[a])));
            node.bmin[a] -= pad;
            node.bmax[a] += pad;
        }
        node.left = node.right = -1;
        node.first = (int)begin;
        node.count = (int)(end - begin);

        const int index = (int)nodes.size();
        nodes.push_back(node);
        if (end - begin <= 2) { return index; }

        int axis = 0;
        if (cmax[1] - cmin[1] > cmax[axis] - cmin[axis]) { axis = 1; }
        if (cmax[2] - cmin[2] > cmax[axis] - cmin[axis]) { axis = 2; }
        const size_t mid = (begin + end) / 2;
        std::nth_element(
            prims.begin() + (long)begin, prims.begin() + (long)mid, prims.begin() + (long)end,
            [axis](const BuildPrim &a, const BuildPrim &b) {
                if (a.centroid[axis] != b.centroid[axis]) { return a.centroid[axis] < b.centroid[axis]; }
                return a.index < b.index;
            });
        const int left = buildRange(prims, begin, mid);
        const int right = buildRange(prims, mid, end);
        nodes[(size_t)index].left = left;
        nodes[(size_t)index].right = right;
        nodes[(size_t)index].count = 0;
        return index;
    }
};

inline bool slabTest(const float bmin[3], const float bmax[3], Vec3 o, Vec3 invD, float tnear, float tfar)
{
    float t0 = tnear, t1 = tfar;
    const float oo[3] = { o.x, o.y, o.z };
    const float ii[3] = { invD.x, invD.y, invD.z };
    for (int a = 0; a < 3; a++) {
        float ta = (bmin[a] - oo[a]) * ii[a];
        float tb = (bmax[a] - oo[a]) * ii[a];
        if (ta > tb) { std::swap(ta, tb); }
        /* NaN (0 * inf) must not cull: comparisons with NaN are false */
        if (ta > t0) { t0 = ta; }
        if (tb < t1) { t1 = tb; }
    }
    return t0 <= t1 * 1.0000004f;
}

/* the same test, also returning the entry distance (used to order the children of a wide node) */
inline bool slabTestEntry(const float bmin[3], const float bmax[3], Vec3 o, Vec3 invD, float tnear, float tfar, float *entry)
{
    float t0 = tnear, t1 = tfar;
    const float oo[3] = { o.x, o.y, o.z };
    const float ii[3] = { invD.x, invD.y, invD.z };
    for (int a = 0; a < 3; a++) {
        float ta = (bmin[a] - oo[a]) * ii[a];
        float tb = (bmax[a] - oo[a]) * ii[a];
        if (ta > tb) { std::swap(ta, tb); }
        if (ta > t0) { t0 = ta; }
        if (tb < t1) { t1 = tb; }
    }
    *entry = t0;
    return t0 <= t1 * 1.0000004f;
}

struct OracleSceneImpl {
    Camera camera;
    std::vector<Triangle> triangles;
    std::vector<Sphere> spheres;
    std::vector<Material> materials;
    std::vector<std::vector<unsigned char>> textures; /* copies: the caller owns the description */
    Bvh bvh;

    /* participating media (src/homogeneous_medium.cpp) and the internal medium of every primitive's surface
     * (Surface::getInternalMedium; -1 = none), triangles first, then spheres */
    struct Medium { Color sigmaT, sigmaS; };
    std::vector<Medium> media;
    std::vector<int> primMedium;

    struct Light {
        int kind; /* 0 triangle, 1 sphere, 2 environment */
        int index;
    };
    std::vector<Light> lights;
    bool hasEnv;
    EnvLight env;

    int width, height;

    /* -- intersector ------------------------------------------------------ */

    bool closestHit(Vec3 o, Vec3 d, float tnear, float tfar, HitRecord *out, Counters *counters) const
    {
        HitRecord best;
        best.t = tfar;
        best.u = best.v = 0.f;
        best.prim = -1;

        const Vec3 invD = v3(1.f / d.x, 1.f / d.y, 1.f / d.z);

        if (!bvh.nodes.empty()) {
            int stack[128];
            int sp = 0;
            stack[sp++] = 0;
            while (sp > 0) {
                const BvhNode &node = bvh.nodes[(size_t)stack[--sp]];
                if (counters) { counters->boxTests++; }
                if (!slabTest(node.bmin, node.bmax, o, invD, tnear, best.t)) { continue; }
                if (node.left < 0) {
                    for (int i = 0; i < node.count; i++) {
                        const int prim = bvh.primOrder[(size_t)(node.first + i)];
                        const Triangle &tri = triangles[(size_t)prim];
                        float t, u, v;
                        if (counters) { counters->triTests++; }
                        if (!intersectTriangle(o, d, tri.p0, tri.p1 - tri.p0, tri.p2 - tri.p0, &t, &u, &v)) { continue; }
                        if (!(t > tnear)) { continue; }
                        const bool closer = (best.prim < 0) ? (t <= best.t) : (t < best.t || (t == best.t && prim < best.prim));
                        if (closer) { best.t = t; best.u = u; best.v = v; best.prim = prim; }
                    }
                } else {
                    stack[sp++] = node.left;
                    stack[sp++] = node.right;
                }
            }
        }

        for (size_t i = 0; i < spheres.size(); i++) {
            float t;
            if (!intersectSphere(o, d, spheres[i].centerWorld, spheres[i].radius, tnear, &t)) { continue; }
            if (!(t > tnear)) { continue; }
            const int prim = (int)(triangles.size() + i);
            const bool closer = (best.prim < 0) ? (t <= best.t) : (t < best.t || (t == best.t && prim < best.prim));
            if (closer) { best.t = t; best.u = 0.f; best.v = 0.f; best.prim = prim; }
        }

        *out = best;
        return best.prim >= 0;
    }

    bool anyHit(Vec3 o, Vec3 d, float tnear, float tfar, Counters *counters) const
    {
        const Vec3 invD = v3(1.f / d.x, 1.f / d.y, 1.f / d.z);
        if (!bvh.nodes.empty()) {
            int stack[128];
            int sp = 0;
            stack[sp++] = 0;
            while (sp > 0) {
                const BvhNode &node = bvh.nodes[(size_t)stack[--sp]];
                if (counters) { counters->boxTests++; }
                if (!slabTest(node.bmin, node.bmax, o, invD, tnear, tfar)) { continue; }
                if (node.left < 0) {
                    for (int i = 0; i < node.count; i++) {
                        const int prim = bvh.primOrder[(size_t)(node.first + i)];
                        const Triangle &tri = triangles[(size_t)prim];
                        float t, u, v;
                        if (counters) { counters->triTests++; }
                        if (!intersectTriangle(o, d, tri.p0, tri.p1 - tri.p0, tri.p2 - tri.p0, &t, &u, &v)) { continue; }
                        if (t > tnear && t <= tfar) { return true; }
                    }
                } else {
                    stack[sp++] = node.left;
                    stack[sp++] = node.right;
                }
            }
        }
        for (size_t i = 0; i < spheres.size(); i++) {
            float t;
            if (!intersectSphere(o, d, spheres[i].centerWorld, spheres[i].radius, tnear, &t)) { continue; }
            if (t > tnear && t <= tfar) { return true; }
        }
        return false;
    }

    /* -- Scene::testIntersect, src/scene.cpp:91-223 ------------------------ */

    Intersection testIntersect(Vec3 o, Vec3 d, Counters *counters) const
    {
        Intersection isect;
        std::memset(&isect, 0, sizeof isect);
        isect.hit = false;
        isect.material = -1;
        isect.prim = -1;

        if (counters) { counters->closestRays++; }
        HitRecord hit;
        if (!closestHit(o, d, 1e-3f, 1e5f, &hit, counters)) { return isect; }
        return makeIntersection(o, d, hit);
    }

    /* Scene::testOcclusion, src/scene.cpp:355-381 */
    bool testOcclusion(Vec3 o, Vec3 d, float maxT, Counters *counters) const
    {
        if (counters) { counters->shadowRays++; }
        return anyHit(o, d, 1e-3f, maxT - 1e-3f, counters);
    }

    /* -- lights ------------------------------------------------------------ */

    struct LightSample {
        int light;
        Vec3 point;
        Vec3 normal;
        float invPDF;
        Measure measure;
    };

    /* Scene::sampleDirectLights, src/scene.cpp:446-467 */
    LightSample sampleDirectLights(Vec3 point, Rng &random) const
    {
        const int lightCount = (int)lights.size();
        int lightIndex = (int)floorf(random.next() * lightCount);
        lightIndex = std::min(lightIndex, lightCount - 1);

        const Light &light = lights[(size_t)lightIndex];
        SurfaceSample surfaceSample;
        if (light.kind == 0) { surfaceSample = triangleSample(triangles[(size_t)light.index], random); }
        else if (light.kind == 1) { surfaceSample = sphereSample(spheres[(size_t)light.index], point, random); }
        else { surfaceSample = env.sample(point, random); }

        const float lightChoicePDF = 1.f / lightCount;

        LightSample sample;
        sample.light = lightIndex;
        sample.point = surfaceSample.point;
        sample.normal = surfaceSample.normal;
        sample.invPDF = surfaceSample.invPDF * (1.f / lightChoicePDF);
        sample.measure = surfaceSample.measure;
        return sample;
    }

    /* LightSample::solidAnglePDF, include/scene.h:66-80 */
    static float solidAnglePDF(const LightSample &sample, Vec3 referencePoint)
    {
        if (sample.measure == SolidAngle) { return 1.f / sample.invPDF; }
        const Vec3 lightDirection = sample.point - referencePoint;
        const Vec3 lightWo = -normalized(lightDirection);
        const float distance = length(lightDirection);
        const float distance2 = distance * distance;
        const float projectedArea = std::max(0.f, dot(sample.normal, lightWo));
        return (1.f / sample.invPDF) * distance2 / projectedArea;
    }

    Color lightEmit(int lightIndex, Vec3 lightWo) const
    {
        const Light &light = lights[(size_t)lightIndex];
        if (light.kind == 0) { return materials[(size_t)triangles[(size_t)light.index].material].emit; }
        if (light.kind == 1) { return materials[(size_t)spheres[(size_t)light.index].material].emit; }
        return env.emit(lightWo);
    }

    /* Scene::lightsPDF, src/scene.cpp:469-484 */
    float lightsPDF(Vec3 referencePoint, const Intersection &lightIntersection) const
    {
        const int lightCount = (int)lights.size();
        float measurePDF;
        if (lightIntersection.prim < (int)triangles.size()) {
            measurePDF = trianglePdfSolidAngle(triangles[(size_t)lightIntersection.prim], lightIntersection.point, referencePoint);
        } else {
            measurePDF = spherePdfSolidAngle(spheres[(size_t)lightIntersection.prim - triangles.size()], referencePoint);
        }
        return measurePDF / lightCount;
    }

    /* Scene::environmentL / environmentPDF, src/scene.cpp:486-502 */
    Color environmentL(Vec3 direction) const
    {
        if (hasEnv) { return env.emit(-direction); }
        return col(0.f);
    }

    float environmentPDF(Vec3 direction) const { return env.emitPDF(direction) / lights.size(); }

    /* -- BounceController, src/bounce_controller.cpp:14-25 ------------------- */

    static bool checkDone(int lastBounce, int bounce)
    {
        if (lastBounce == -1) { return false; }
        return bounce > lastBounce;
    }

    static bool checkCounts(int startBounce, int lastBounce, int bounce)
    {
        if (startBounce > bounce) { return false; }
        return !checkDone(lastBounce, bounce);
    }

    /* -- PathTracer::directSampleLights, src/path_tracer.cpp:113-165 ---------- */

    Color directSampleLights(const Intersection &isect, const Material &material, Rng &random, Counters *counters,
                             bool *queryMatters = nullptr) const
    {
        if (queryMatters) { *queryMatters = false; }
        if (isDelta(material)) { return col(0.f); }
        if (lights.empty()) { return col(0.f); } /* the reference would index an empty vector */

        const LightSample lightSample = sampleDirectLights(isect.point, random);

        const Vec3 lightDirection = lightSample.point - isect.point;
        const Vec3 wiWorld = normalized(lightDirection);

        if (dot(lightSample.normal, wiWorld) >= 0.f) { return col(0.f); }

        const float lightDistance = length(lightDirection);
        /* The reference queries the occlusion first (:140-141) and evaluates the contribution after; both are
         * pure, so evaluating first changes nothing -- it only lets the oracle count the queries whose answer
         * matters. */
        const float pdf = solidAnglePDF(lightSample, isect.point);
        float brdfPDF;
        const Color f = materialF(material, isect, wiWorld, &brdfPDF);
        const float lightWeight = (1 * pdf) / (1 * pdf + 1 * brdfPDF); /* include/mis.h:4-7 */

        /* Shortcut shared with the HIP kernels (deliberate deviation, header): with f exactly black and finite,
         * positive pdfs the product below is black whatever the occlusion query says; it is not evaluated (only a
         * non-finite emission could tell the difference).  The query is still made here, as in the reference, but
         * it does not count as needed. */
        const bool blackLobe = isBlack(f) && pdf > 0.f && pdf < 3e38f && brdfPDF >= 0.f && brdfPDF < 3e38f;
        if (blackLobe) {
            (void)testOcclusion(isect.point, wiWorld, lightDistance, counters);
            return col(0.f);
        }

        const Vec3 lightWo = -normalized(lightDirection);

        const Color contribution = lightEmit(lightSample.light, lightWo)
            * lightWeight
            * f
            * fabsf(dot(isect.shadingNormal, wiWorld))
            / pdf;
        if (counters) { counters->shadowRaysNeeded++; }
        if (queryMatters) { *queryMatters = true; }

        if (testOcclusion(isect.point, wiWorld, lightDistance, counters)) { return col(0.f); }
        return contribution;
    }

    /* -- PathTracer::directSampleBSDF, src/path_tracer.cpp:167-216, given the
     *    already-traced intersection along bsdfSample.wiWorld ------------------ */

    Color directSampleBSDF(
        const Intersection &isect, const Material &material,
        const BSDFSample &bsdfSample, const Intersection &bounceIntersection
    ) const {
        if (bounceIntersection.hit
            && !isBlack(materials[(size_t)bounceIntersection.material].emit)
            && dot(bounceIntersection.wo, bounceIntersection.shadingNormal) >= 0.f
        ) {
            const float lightPDF = lightsPDF(isect.point, bounceIntersection);
            const float brdfWeight = isDelta(material)
                ? 1.f
                : (1 * bsdfSample.pdf) / (1 * bsdfSample.pdf + 1 * lightPDF);

            return materials[(size_t)bounceIntersection.material].emit
                * brdfWeight
                * bsdfSample.throughput
                * fabsf(dot(isect.shadingNormal, bsdfSample.wiWorld))
                / bsdfSample.pdf;
        } else if (!bounceIntersection.hit) {
            const Color environmentLight = environmentL(bsdfSample.wiWorld);
            if (!isBlack(environmentLight)) {
                const float lightPDF = environmentPDF(bsdfSample.wiWorld);
                const float brdfWeight = isDelta(material)
                    ? 1.f
                    : (1 * bsdfSample.pdf) / (1 * bsdfSample.pdf + 1 * lightPDF);

                return environmentLight
                    * brdfWeight
                    * bsdfSample.throughput
                    * fabsf(dot(isect.shadingNormal, bsdfSample.wiWorld))
                    / bsdfSample.pdf;
            }
        }
        return col(0.f);
    }

    /* ======================================================================================================
     * Participating media: VolumePathTracer (src/volume_path_tracer.cpp), DirectLightingHelper::Ld
     * (src/direct_lighting_helper.cpp:37-187), VolumeHelper (src/volume_helper.cpp), HomogeneousMedium
     * (src/homogeneous_medium.cpp) and the two "volumetric" ray queries of Scene (src/scene.cpp:225-353, 383-424).
     *
     * PARITY UNPINNED, and in one place DEFINED rather than restated.  The reference makes Embree skip container
     * surfaces (passthrough material + internal medium) through an intersection filter that records a VolumeEvent
     * (t, medium) for every container hit Embree reports during its traversal (src/scene.cpp:42-83).  Which hits
     * those are depends on Embree's traversal order: a container beyond the final opaque hit is reported if it is
     * met first and culled if it is met later.  Here the events of a closest-hit query are the container hits
     * with tnear < t < t(final hit), those of an occlusion query the container hits inside the query interval;
     * equal t are recorded once (scene.cpp:72-77), and the first two in order of t are used (the reference asserts
     * there are one or two, volume_helper.cpp:43, 83, 101).
     * ====================================================================================================== */

    struct VolumeEvent { float t; int medium; };

    bool containerPrim(int prim) const
    {
        const int material = prim < (int)triangles.size() ? triangles[(size_t)prim].material : spheres[(size_t)prim - triangles.size()].material;
        return isContainer(materials[(size_t)material]) && primMedium[(size_t)prim] >= 0;
    }

    /* every primitive (container or not) a ray meets in (tnear, tmax], brute force: the media scenes are small */
    template <typename Visit>
    void forEachHit(Vec3 o, Vec3 d, float tnear, Visit visit) const
    {
        for (size_t i = 0; i < triangles.size(); i++) {
            const Triangle &tri = triangles[i];
            float t, u, v;
            if (!intersectTriangle(o, d, tri.p0, tri.p1 - tri.p0, tri.p2 - tri.p0, &t, &u, &v)) { continue; }
            if (!(t > tnear)) { continue; }
            visit((int)i, t, u, v);
        }
        for (size_t i = 0; i < spheres.size(); i++) {
            float t;
            if (!intersectSphere(o, d, spheres[i].centerWorld, spheres[i].radius, tnear, &t)) { continue; }
            if (!(t > tnear)) { continue; }
            visit((int)(triangles.size() + i), t, 0.f, 0.f);
        }
    }

    static void addEvent(std::vector<VolumeEvent> &events, float t, int medium)
    {
        for (const VolumeEvent &existing : events) { if (existing.t == t) { return; } }
        events.push_back({ t, medium });
    }

    static void sortEvents(std::vector<VolumeEvent> &events)
    {
        std::sort(events.begin(), events.end(), [](const VolumeEvent &a, const VolumeEvent &b) { return a.t < b.t; });
    }

    Intersection makeIntersection(Vec3 o, Vec3 d, const HitRecord &hit) const
    {
        Intersection isect;
        std::memset(&isect, 0, sizeof isect);

        Vec3 geometricNormal;
        Vec3 shadingNormal = v3(0.f, 0.f, 0.f);
        float uvU = 0.f, uvV = 0.f;
        int material;

        if (hit.prim < (int)triangles.size()) {
            const Triangle &tri = triangles[(size_t)hit.prim];
            const float w = 1.f - hit.u - hit.v;
            /* rtcInterpolate0 with weights (1-u-v, u, v) */
            uvU = fmaf(w, tri.uv0[0], fmaf(hit.u, tri.uv1[0], hit.v * tri.uv2[0]));
            uvV = fmaf(w, tri.uv0[1], fmaf(hit.u, tri.uv1[1], hit.v * tri.uv2[1]));
            shadingNormal = v3(
                fmaf(w, tri.n0.x, fmaf(hit.u, tri.n1.x, hit.v * tri.n2.x)),
                fmaf(w, tri.n0.y, fmaf(hit.u, tri.n1.y, hit.v * tri.n2.y)),
                fmaf(w, tri.n0.z, fmaf(hit.u, tri.n1.z, hit.v * tri.n2.z)));
            /* Ng = (v1 - v0) x (v2 - v0) */
            geometricNormal = normalized(xcross(tri.p1 - tri.p0, tri.p2 - tri.p0));
            material = tri.material;
        } else {
            const Sphere &sphere = spheres[(size_t)hit.prim - triangles.size()];
            const Vec3 p = o + d * hit.t;
            geometricNormal = normalized(p - sphere.centerWorld);
            material = sphere.material;
        }

        if (length(shadingNormal) == 0.f) { shadingNormal = geometricNormal; }

        isect.hit = true;
        isect.t = hit.t;
        isect.point = o + d * hit.t; /* Ray::at, src/ray.cpp:9-12 */
        isect.wo = -d;
        isect.normal = geometricNormal;
        isect.shadingNormal = normalized(shadingNormal);
        isect.u = uvU;
        isect.v = uvV;
        isect.material = material;
        isect.prim = hit.prim;
        isect.frame = normalToWorldSpace(isect.shadingNormal, isect.wo);
        return isect;
    }


    /* Scene::testVolumetricIntersect, src/scene.cpp:225-353 */
    Intersection testVolumetricIntersect(Vec3 o, Vec3 d, std::vector<VolumeEvent> *events, Counters *counters) const
    {
        if (counters) { counters->closestRays++; }
        HitRecord best;
        best.t = 1e5f; best.u = best.v = 0.f; best.prim = -1;
        forEachHit(o, d, 1e-3f, [&](int prim, float t, float u, float v) {
            if (containerPrim(prim)) { return; }
            const bool closer = (best.prim < 0) ? (t <= best.t) : (t < best.t || (t == best.t && prim < best.prim));
            if (closer) { best.t = t; best.u = u; best.v = v; best.prim = prim; }
        });
        events->clear();
        forEachHit(o, d, 1e-3f, [&](int prim, float t, float, float) {
            if (containerPrim(prim) && t < best.t) { addEvent(*events, t, primMedium[(size_t)prim]); }
        });
        sortEvents(*events);
        Intersection isect;
        std::memset(&isect, 0, sizeof isect);
        isect.hit = false;
        isect.material = -1;
        isect.prim = -1;
        if (best.prim < 0) { return isect; }
        return makeIntersection(o, d, best);
    }

    /* Scene::testVolumetricOcclusion, src/scene.cpp:383-424 */
    bool testVolumetricOcclusion(Vec3 o, Vec3 d, float maxT, std::vector<VolumeEvent> *events, Counters *counters) const
    {
        if (counters) { counters->shadowRays++; }
        const float tfar = maxT - 1e-3f;
        bool occluded = false;
        events->clear();
        forEachHit(o, d, 1e-3f, [&](int prim, float t, float, float) {
            if (!(t <= tfar)) { return; }
            if (containerPrim(prim)) { addEvent(*events, t, primMedium[(size_t)prim]); }
            else { occluded = true; }
        });
        sortEvents(*events);
        return occluded;
    }

    /* HomogeneousMedium::transmittance, src/homogeneous_medium.cpp:14-18 */
    Color mediumTransmittance(int medium, Vec3 pointA, Vec3 pointB) const
    {
        const Vec3 path = pointB - pointA;
        const Color sigmaT = media[(size_t)medium].sigmaT;
        const float distance = length(path);
        return col(expf(-sigmaT.r * distance), expf(-sigmaT.g * distance), expf(-sigmaT.b * distance));
    }

    /* VolumeHelper::rayTransmission, src/volume_helper.cpp:71-123 */
    Color rayTransmission(Vec3 o, Vec3 d, const std::vector<VolumeEvent> &events, int medium) const
    {
        Color transmittance = col(1.f);
        const size_t eventCount = events.size();
        if (eventCount == 0) { return transmittance; }
        if (medium >= 0) {
            if (eventCount == 1) { transmittance = transmittance * mediumTransmittance(medium, o, o + d * events[0].t); }
            else { transmittance = transmittance * mediumTransmittance(medium, o + d * events[0].t, o + d * events[1].t); }
        } else {
            const int eventMedium = events[0].medium;
            if (eventCount >= 2) { transmittance = transmittance * mediumTransmittance(eventMedium, o + d * events[0].t, o + d * events[1].t); }
            else { transmittance = transmittance * mediumTransmittance(eventMedium, o, o + d * events[0].t); }
        }
        return transmittance;
    }

    /* VolumeHelper::directSampleLights, src/volume_helper.cpp:12-69 (isotropic phase function 1 / 4 pi) */
    Color volumeDirectSampleLights(int medium, Vec3 samplePoint, Rng &random, Counters *counters) const
    {
        if (lights.empty()) { return col(0.f); }
        const LightSample lightSample = sampleDirectLights(samplePoint, random);
        const Vec3 sampleDirection = lightSample.point - samplePoint;
        const Vec3 wiWorld = normalized(sampleDirection);
        if (dot(lightSample.normal, wiWorld) >= 0.f) { return col(0.f); }
        const float lightDistance = length(sampleDirection);
        std::vector<VolumeEvent> events;
        if (testVolumetricOcclusion(samplePoint, wiWorld, lightDistance, &events, counters)) { return col(0.f); }
        const float pdf = solidAnglePDF(lightSample, samplePoint);
        const Vec3 lightWo = -normalized(sampleDirection);
        Color shadowTransmittance = col(0.f);
        if (events.size() == 1) { shadowTransmittance = mediumTransmittance(medium, samplePoint, samplePoint + wiWorld * events[0].t); }
        else if (events.size() >= 2) { shadowTransmittance = mediumTransmittance(medium, samplePoint + wiWorld * events[0].t, samplePoint + wiWorld * events[1].t); }
        /* emit * T * 1.f / (4.f * M_PI) / pdf: the product 4 pi is a double, Color::operator/ takes it as a float */
        const float fourPi = (float)(4.f * 3.14159265358979323846);
        return lightEmit(lightSample.light, lightWo) * shadowTransmittance * 1.f / fourPi / pdf;
    }

    /* VolumePathTracer::scatter -> HomogeneousMedium::integrate, src/volume_path_tracer.cpp:114-131,
     * src/homogeneous_medium.cpp:36-66: one distance sample on the segment; if it lands inside, the in-scattered
     * light of one light sample there (the path itself goes on undeflected) */
    Color scatter(int medium, Vec3 entry, Vec3 exit, Rng &random, Counters *counters) const
    {
        if (medium < 0) { return col(0.f); }
        const float sigmaT = media[(size_t)medium].sigmaT.r;
        const Vec3 travel = exit - entry;
        const float distance = length(travel);
        const float xi = random.next();
        const float sampleT = -logf(1 - xi) / sigmaT;
        if (sampleT >= distance) { return col(0.f); }
        const Vec3 samplePoint = entry + normalized(travel) * sampleT;
        return volumeDirectSampleLights(medium, samplePoint, random, counters);
    }

    /* DirectLightingHelper::Ld, src/direct_lighting_helper.cpp:37-187 */
    Color volumeLd(const Intersection &isect, int medium, const Material &material, const BSDFSample &bsdfSample, Rng &random, Counters *counters) const
    {
        if (isContainer(material)) { return col(0.f); }
        if (!isBlack(material.emit)) { return col(0.f); }
        Color result = col(0.f);
        /* directSampleLights, :74-134 */
        Color lightContribution = col(0.f);
        if (!isDelta(material) && !lights.empty()) {
            const LightSample lightSample = sampleDirectLights(isect.point, random);
            const Vec3 lightDirection = lightSample.point - isect.point;
            const Vec3 wiWorld = normalized(lightDirection);
            if (!(dot(lightSample.normal, wiWorld) >= 0.f)) {
                const float lightDistance = length(lightDirection);
                std::vector<VolumeEvent> events;
                if (!testVolumetricOcclusion(isect.point, wiWorld, lightDistance, &events, counters)) {
                    const Color transmittance = rayTransmission(isect.point, wiWorld, events, medium);
                    const float pdf = solidAnglePDF(lightSample, isect.point);
                    float brdfPDF;
                    const Color f = materialF(material, isect, wiWorld, &brdfPDF);
                    const float lightWeight = (1 * pdf) / (1 * pdf + 1 * brdfPDF);
                    const Vec3 lightWo = -normalized(lightDirection);
                    lightContribution = lightEmit(lightSample.light, lightWo)
                        * transmittance
                        * lightWeight
                        * f
                        * fabsf(dot(isect.shadingNormal, wiWorld))
                        / pdf;
                }
            }
        }
        result = result + lightContribution;
        /* directSampleBSDF, :136-187: the query skips containers; no transmittance is applied (as in the reference) */
        std::vector<VolumeEvent> events;
        const Intersection bounce = testVolumetricIntersect(isect.point, bsdfSample.wiWorld, &events, counters);
        result = result + directSampleBSDF(isect, material, bsdfSample, bounce);
        return result;
    }

    /* SampleIntegrator::samplePixel + VolumePathTracer::L, src/sample_integrator.cpp:10-78, src/volume_path_tracer.cpp:14-99.
     * Random dimensions: the BSDF and light samples of vertex k where PathTracer has them; the distance sample and the
     * light sample of the medium on the segment that ENDS at vertex k at kMediumDimensions + 4 (k - 1) + {0; 1, 2, 3}. */
    static uint32_t mediumBase(int vertex) { return 0x4000u + 4u * (uint32_t)(vertex - 1); }

    Color samplePixelVolume(uint64_t seed, int row, int col_, uint32_t sampleIndex, int startBounce, int lastBounce, Counters *counters) const
    {
        const uint32_t pixelIndex = (uint32_t)(row * width + col_);
        Rng random = keyedRng(seed, pixelIndex, sampleIndex);
        if (counters) { counters->cameraSamples++; }
        random.dimension = 0;
        const float jitterX = random.next() - 0.5f;
        const float jitterY = random.next() - 0.5f;
        Vec3 rayOrigin, rayDirection;
        camera.generateRay(row + jitterY, col_ + jitterX, &rayOrigin, &rayDirection);

        Color color = col(0.f);
        const Intersection intersection = testIntersect(rayOrigin, rayDirection, counters);
        if (!intersection.hit) { return color + environmentL(rayDirection); }

        if (checkCounts(startBounce, lastBounce, 0)) {
            const Material &first = materials[(size_t)intersection.material];
            const bool backside = dot(intersection.normal, intersection.wo) < 0.f;
            if (!isBlack(first.emit) && !backside) { color = color + first.emit; }
            if (isContainer(first)) {
                /* what is seen through the container, src/sample_integrator.cpp:35-51 */
                std::vector<VolumeEvent> events;
                const Intersection through = testVolumetricIntersect(rayOrigin, rayDirection, &events, counters);
                const Color transmittance = rayTransmission(rayOrigin, rayDirection, events, -1);
                if (through.hit) { color = color + materials[(size_t)through.material].emit * transmittance; }
                else { color = color + environmentL(rayDirection) * transmittance; }
            }
        }

        /* ---- VolumePathTracer::L ---- */
        int medium = -1;
        Intersection last = intersection;
        random.dimension = vertexBase(1);
        BSDFSample bsdfSample = materialSample(materials[(size_t)last.material], last, random);
        if (counters) { counters->vertices++; }
        Color result = col(0.f);
        if (checkCounts(startBounce, lastBounce, 1)) {
            random.dimension = vertexBase(1) + 3;
            result = volumeLd(last, medium, materials[(size_t)last.material], bsdfSample, random, counters);
        }
        Color modulation = col(1.f);
        for (int bounce = 2; !checkDone(lastBounce, bounce); bounce++) {
            /* refraction: the medium changes (:43-51) */
            if (dot(last.wo, bsdfSample.wiWorld) < 0.f) {
                if (dot(last.normal, bsdfSample.wiWorld) < 0.f) { medium = primMedium[(size_t)last.prim]; }
                else { medium = -1; }
            }
            const Intersection next = testIntersect(last.point, bsdfSample.wiWorld, counters);
            if (!next.hit) { break; }
            if (counters) { counters->vertices++; }
            const float invPDF = 1.f / bsdfSample.pdf;
            const float cosTheta = fabsf(dot(last.shadingNormal, bsdfSample.wiWorld));
            modulation = modulation * (bsdfSample.throughput * cosTheta * invPDF);

            random.dimension = mediumBase(bounce);
            const Color Ls = scatter(medium, last.point, next.point, random, counters);
            result = result + Ls * modulation;
            if (medium >= 0) { modulation = modulation * mediumTransmittance(medium, last.point, next.point); }
            else { modulation = modulation * col(1.f); }
            if (isBlack(modulation)) { break; }

            random.dimension = vertexBase(bounce);
            bsdfSample = materialSample(materials[(size_t)next.material], next, random);
            last = next;
            if (checkCounts(startBounce, lastBounce, bounce)) {
                random.dimension = vertexBase(bounce) + 3;
                const Color Ld = volumeLd(next, medium, materials[(size_t)next.material], bsdfSample, random, counters);
                result = result + Ld * modulation;
            }
        }
        return color + result;
    }

    /* -- SampleIntegrator::samplePixel + PathTracer::L --------------------------
     * src/sample_integrator.cpp:10-78, src/path_tracer.cpp:19-77.  The ray along
     * bsdfSample.wiWorld is traced once and serves both direct()'s BSDF-sampling
     * term and the continuation. */

    Color samplePixel(uint64_t seed, int row, int col_, uint32_t sampleIndex, int startBounce, int lastBounce, Counters *counters) const
    {
        const uint32_t pixelIndex = (uint32_t)(row * width + col_);
        Rng random = keyedRng(seed, pixelIndex, sampleIndex);
        if (counters) { counters->cameraSamples++; }

        /* Camera::generateRay(int,int), src/camera.cpp:49-55: X jitter drawn first */
        random.dimension = 0;
        const float jitterX = random.next() - 0.5f;
        const float jitterY = random.next() - 0.5f;
        Vec3 rayOrigin, rayDirection;
        camera.generateRay(row + jitterY, col_ + jitterX, &rayOrigin, &rayDirection);

        Color color = col(0.f);

        Intersection intersection = testIntersect(rayOrigin, rayDirection, counters);
        if (!intersection.hit) {
            color = color + environmentL(rayDirection);
            return color;
        }

        if (checkCounts(startBounce, lastBounce, 0)) {
            const Color emit = materials[(size_t)intersection.material].emit;
            const bool backside = dot(intersection.normal, intersection.wo) < 0.f;
            if (!isBlack(emit) && !backside) { color = color + emit; }
        }

        /* ---- PathTracer::L ---- */
        Color result = col(0.f);
        Color modulation = col(1.f);
        Intersection last = intersection;
        int bounce = 1; /* index of the vertex `last` */

        while (true) {
            if (counters) { counters->vertices++; }
            const Material &material = materials[(size_t)last.material];

            random.dimension = vertexBase(bounce);
            const BSDFSample bsdfSample = materialSample(material, last, random);

            const bool counts = checkCounts(startBounce, lastBounce, bounce);
            const bool emissive = !isBlack(material.emit);
            const bool wantDirect = counts && !emissive; /* direct() returns 0 on emitters */
            const bool wantContinue = !checkDone(lastBounce, bounce + 1);

            Color lightTerm = col(0.f);
            bool queryMatters = false;
            if (wantDirect) {
                random.dimension = vertexBase(bounce) + 3;
                lightTerm = directSampleLights(last, material, random, counters, &queryMatters);
            }

            if (!wantDirect && !wantContinue) { break; }

            /* Shortcut shared with the HIP kernels (deliberate deviation, header): nothing pending at this vertex
             * and a BSDF sample of exactly black throughput.  The reference traces the continuation ray, finds its
             * MIS term and the new modulation black, and stops with the same result (only a non-finite emission or
             * light pdf on that ray could tell the difference). */
            if (isBlack(bsdfSample.throughput) && bsdfSample.pdf > 0.f && bsdfSample.pdf < 3e38f
                && !queryMatters && isBlack(lightTerm)) { break; }

            const Intersection next = testIntersect(last.point, bsdfSample.wiWorld, counters);

            if (wantDirect) {
                const Color bsdfTerm = directSampleBSDF(last, material, bsdfSample, next);
                const Color Ld = lightTerm + bsdfTerm;
                if (bounce == 1) { result = Ld; }
                else { result = result + Ld * modulation; }
            }

            if (!wantContinue) { break; }
            if (!next.hit) { break; }

            const float invPDF = 1.f / bsdfSample.pdf;
            const float cosTheta = fabsf(dot(last.shadingNormal, bsdfSample.wiWorld));
            modulation = modulation * (bsdfSample.throughput * cosTheta * invPDF);
            if (isBlack(modulation)) { break; }

            last = next;
            bounce++;
        }

        color = color + result;
        return color;
    }
};

inline bool finiteColor(Color c) { return std::isfinite(c.r) && std::isfinite(c.g) && std::isfinite(c.b); }

}  // namespace

struct OracleScene {
    OracleSceneImpl impl;
    int integrator = 0;   /* PATHED_INTEGRATOR_*: 0 PathTracer, 1 VolumePathTracer */
};

extern "C" {

const char *oracle_last_error(void) { return g_error.c_str(); }

OracleScene *oracle_scene_create(const PathedSceneDesc *desc)
{
    if (!desc || desc->abi_version != PATHED_ABI_VERSION) {
        g_error = "oracle: bad scene description";
        return nullptr;
    }
    OracleScene *scene = new OracleScene();
    OracleSceneImpl &impl = scene->impl;

    impl.camera.build(desc->camera);
    impl.width = desc->camera.width;
    impl.height = desc->camera.height;

    impl.materials.resize(desc->n_materials);
    impl.textures.resize(desc->n_textures);
    for (uint32_t t = 0; t < desc->n_textures; t++) {
        const PathedTexture &texture = desc->textures[t];
        impl.textures[t].assign(texture.rgb, texture.rgb + (size_t)3 * texture.width * texture.height);
    }
    for (uint32_t i = 0; i < desc->n_materials; i++) {
        impl.materials[i] = materialFromDesc(desc->materials[i]);
        if (desc->materials[i].albedo_type == PATHED_ALBEDO_TEXTURE) {
            const int t = desc->materials[i].texture;
            if (t < 0 || (uint32_t)t >= desc->n_textures) {
                g_error = "oracle: material texture index out of range";
                delete scene;
                return nullptr;
            }
            impl.materials[i].texData = impl.textures[(size_t)t].data();
            impl.materials[i].texWidth = desc->textures[t].width;
            impl.materials[i].texHeight = desc->textures[t].height;
        }
    }

    auto vertex = [&](uint32_t index) {
        return v3(desc->positions[3 * index + 0], desc->positions[3 * index + 1], desc->positions[3 * index + 2]);
    };
    auto vertexNormal = [&](uint32_t index) {
        return v3(desc->normals[3 * index + 0], desc->normals[3 * index + 1], desc->normals[3 * index + 2]);
    };

    impl.triangles.resize(desc->n_triangles);
    for (uint32_t i = 0; i < desc->n_triangles; i++) {
        Triangle &tri = impl.triangles[i];
        const uint32_t i0 = desc->indices[3 * i + 0];
        const uint32_t i1 = desc->indices[3 * i + 1];
        const uint32_t i2 = desc->indices[3 * i + 2];
        if (i0 >= desc->n_vertices || i1 >= desc->n_vertices || i2 >= desc->n_vertices) {
            g_error = "oracle: vertex index out of range";
            delete scene;
            return nullptr;
        }
        tri.p0 = vertex(i0); tri.p1 = vertex(i1); tri.p2 = vertex(i2);
        tri.n0 = vertexNormal(i0); tri.n1 = vertexNormal(i1); tri.n2 = vertexNormal(i2);
        tri.uv0[0] = desc->uvs[2 * i0]; tri.uv0[1] = desc->uvs[2 * i0 + 1];
        tri.uv1[0] = desc->uvs[2 * i1]; tri.uv1[1] = desc->uvs[2 * i1 + 1];
        tri.uv2[0] = desc->uvs[2 * i2]; tri.uv2[1] = desc->uvs[2 * i2 + 1];
        tri.material = desc->tri_material[i];
    }

    impl.spheres.resize(desc->n_spheres);
    for (uint32_t i = 0; i < desc->n_spheres; i++) {
        const PathedSphere &s = desc->spheres[i];
        impl.spheres[i].centerWorld = v3(s.center_world[0], s.center_world[1], s.center_world[2]);
        impl.spheres[i].centerSample = v3(s.center_sample[0], s.center_sample[1], s.center_sample[2]);
        impl.spheres[i].radius = s.radius;
        impl.spheres[i].material = s.material;
    }

    /* lights: every emissive surface in model order, environment light last
     * (src/scene_parser.cpp:173-190) */
    for (uint32_t g = 0; g < desc->n_geoms; g++) {
        const PathedGeom &geom = desc->geoms[g];
        if (geom.type == PATHED_GEOM_MESH) {
            for (int i = 0; i < geom.count; i++) {
                const int tri = geom.first + i;
                if (!isBlack(impl.materials[(size_t)impl.triangles[(size_t)tri].material].emit)) {
                    impl.lights.push_back({ 0, tri });
                }
            }
        } else {
            if (!isBlack(impl.materials[(size_t)impl.spheres[(size_t)geom.first].material].emit)) {
                impl.lights.push_back({ 1, geom.first });
            }
        }
    }
    impl.hasEnv = desc->env != nullptr;
    if (impl.hasEnv) {
        impl.env.build(*desc->env);
        impl.lights.push_back({ 2, 0 });
    }

    impl.media.resize(desc->n_media);
    for (uint32_t i = 0; i < desc->n_media; i++) {
        impl.media[i].sigmaT = col(desc->media[i].sigma_t[0], desc->media[i].sigma_t[1], desc->media[i].sigma_t[2]);
        impl.media[i].sigmaS = col(desc->media[i].sigma_s[0], desc->media[i].sigma_s[1], desc->media[i].sigma_s[2]);
    }
    impl.primMedium.assign((size_t)desc->n_triangles + desc->n_spheres, -1);
    for (uint32_t g = 0; g < desc->n_geoms; g++) {
        const PathedGeom &geom = desc->geoms[g];
        const int medium = (geom.medium >= 0 && (uint32_t)geom.medium < desc->n_media) ? geom.medium : -1;
        if (geom.type == PATHED_GEOM_MESH) {
            for (int i = 0; i < geom.count; i++) { impl.primMedium[(size_t)(geom.first + i)] = medium; }
        } else {
            impl.primMedium[(size_t)desc->n_triangles + (size_t)geom.first] = medium;
        }
    }

    impl.bvh.build(impl.triangles);
    return scene;
}

void oracle_scene_destroy(OracleScene *scene) { delete scene; }

int oracle_set_integrator(OracleScene *scene, int integrator)
{
    if (!scene || (integrator != PATHED_INTEGRATOR_PATH_TRACER && integrator != PATHED_INTEGRATOR_VOLUME_PATH_TRACER)) { return -1; }
    scene->integrator = integrator;
    return 0;
}

int oracle_light_count(OracleScene *scene) { return scene ? (int)scene->impl.lights.size() : -1; }

float oracle_rng(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t dimension)
{
    return keyedUniform(makeKey(seed, pixel, sample), dimension);
}

int oracle_sample_pixel(OracleScene *scene, uint64_t seed, int row, int col_, uint32_t sample,
                        int start_bounce, int last_bounce, float *rgb)
{
    if (!scene || !rgb) { return -1; }
    const Color c = scene->impl.samplePixel(seed, row, col_, sample, start_bounce, last_bounce, nullptr);
    rgb[0] = c.r; rgb[1] = c.g; rgb[2] = c.b;
    return 0;
}

int oracle_render_chunked(OracleScene *scene, uint64_t seed, uint32_t spp_begin, uint32_t spp_count,
                          int start_bounce, int last_bounce, float *accum, int threads, uint64_t *stats, int chunk);

int oracle_render(OracleScene *scene, uint64_t seed, uint32_t spp_begin, uint32_t spp_count,
                  int start_bounce, int last_bounce, float *accum, int threads, uint64_t *stats)
{
    return oracle_render_chunked(scene, seed, spp_begin, spp_count, start_bounce, last_bounce, accum, threads, stats, 1);
}

/* chunk > 1 mirrors the product's summation granularity (pathed_hip_set_samples_per_unit):
 * samples are summed in order inside groups of `chunk`, group sums are added in order. */
int oracle_render_chunked(OracleScene *scene, uint64_t seed, uint32_t spp_begin, uint32_t spp_count,
                          int start_bounce, int last_bounce, float *accum, int threads, uint64_t *stats, int chunk)
{
    if (!scene || !accum) { g_error = "oracle: null argument"; return -1; }
    const OracleSceneImpl &impl = scene->impl;
    if (impl.lights.empty() && start_bounce <= 1) {
        /* the reference would divide by a zero light count; nothing to sample */
    }
    const int width = impl.width, height = impl.height;
    const bool volume = scene->integrator == PATHED_INTEGRATOR_VOLUME_PATH_TRACER;
    Counters total;
    std::memset(&total, 0, sizeof total);

#ifdef _OPENMP
    if (threads > 1) { omp_set_num_threads(threads); }
#endif
    /* rows in parallel, columns inside — src/sample_integrator.cpp:99-110; samples of a
     * pixel are added in index order like the reference's wave loop (integrator.cpp:42).
     * The reference's loop is schedule(static); rows differ in cost by several times (walls vs
     * the boxes), so the timed baseline hands rows out dynamically — a pixel's sum does not
     * depend on which thread computes it, counters are per thread. */
#pragma omp parallel if (threads > 1)
    {
        Counters local;
        std::memset(&local, 0, sizeof local);
#pragma omp for schedule(dynamic, 1)
        for (int row = 0; row < height; row++) {
            for (int col_ = 0; col_ < width; col_++) {
                float *pixel = accum + 3 * ((size_t)row * width + col_);
                if (chunk <= 1) {
                    for (uint32_t s = 0; s < spp_count; s++) {
                        const Color c = volume
                            ? impl.samplePixelVolume(seed, row, col_, spp_begin + s, start_bounce, last_bounce, &local)
                            : impl.samplePixel(seed, row, col_, spp_begin + s, start_bounce, last_bounce, &local);
                        if (!finiteColor(c)) { local.dropped++; continue; }
                        pixel[0] += c.r;
                        pixel[1] += c.g;
                        pixel[2] += c.b;
                    }
                } else {
                    for (uint32_t first = 0; first < spp_count; first += (uint32_t)chunk) {
                        float partial[3] = { 0.f, 0.f, 0.f };
                        for (uint32_t s = first; s < spp_count && s < first + (uint32_t)chunk; s++) {
                            const Color c = volume
                                ? impl.samplePixelVolume(seed, row, col_, spp_begin + s, start_bounce, last_bounce, &local)
                                : impl.samplePixel(seed, row, col_, spp_begin + s, start_bounce, last_bounce, &local);
                            if (!finiteColor(c)) { local.dropped++; continue; }
                            partial[0] += c.r;
                            partial[1] += c.g;
                            partial[2] += c.b;
                        }
                        pixel[0] += partial[0];
                        pixel[1] += partial[1];
                        pixel[2] += partial[2];
                    }
                }
            }
        }
#pragma omp critical
        {
            total.cameraSamples += local.cameraSamples;
            total.closestRays += local.closestRays;
            total.shadowRays += local.shadowRays;
            total.boxTests += local.boxTests;
            total.triTests += local.triTests;
            total.dropped += local.dropped;
            total.vertices += local.vertices;
            total.shadowRaysNeeded += local.shadowRaysNeeded;
        }
    }
    if (stats) {
        stats[0] = total.cameraSamples; stats[1] = total.closestRays; stats[2] = total.shadowRays;
        stats[3] = total.boxTests; stats[4] = total.triTests; stats[5] = total.dropped;
        stats[6] = total.vertices; stats[7] = total.shadowRaysNeeded;
    }
    return 0;
}

int oracle_trace(OracleScene *scene, const float *rays, size_t n, int any_hit, void *hits)
{
    if (!scene || (!rays && n) || (!hits && n)) { return -1; }
    const OracleSceneImpl &impl = scene->impl;
    for (size_t i = 0; i < n; i++) {
        const float *r = rays + 8 * i;
        const Vec3 o = v3(r[0], r[1], r[2]);
        const Vec3 d = v3(r[4], r[5], r[6]);
        if (any_hit) {
            ((int32_t *)hits)[i] = impl.anyHit(o, d, r[3], r[7], nullptr) ? 1 : 0;
        } else {
            HitRecord hit;
            impl.closestHit(o, d, r[3], r[7], &hit, nullptr);
            float *out = (float *)hits + 4 * i;
            if (hit.prim < 0) { hit.t = 0.f; hit.u = hit.v = 0.f; }
            out[0] = hit.t; out[1] = hit.u; out[2] = hit.v;
            std::memcpy(out + 3, &hit.prim, 4);
        }
    }
    return 0;
}

int oracle_trace_bruteforce(OracleScene *scene, const float *rays, size_t n, double *t_out, int32_t *prim_out)
{
    if (!scene) { return -1; }
    const OracleSceneImpl &impl = scene->impl;
    for (size_t i = 0; i < n; i++) {
        const float *r = rays + 8 * i;
        const double o[3] = { r[0], r[1], r[2] };
        const double d[3] = { r[4], r[5], r[6] };
        const double tnear = r[3], tfar = r[7];
        double best = tfar;
        int bestPrim = -1;
        for (size_t p = 0; p < impl.triangles.size(); p++) {
            const Triangle &tri = impl.triangles[p];
            const double v0[3] = { tri.p0.x, tri.p0.y, tri.p0.z };
            const double e1[3] = { (double)tri.p1.x - tri.p0.x, (double)tri.p1.y - tri.p0.y, (double)tri.p1.z - tri.p0.z };
            const double e2[3] = { (double)tri.p2.x - tri.p0.x, (double)tri.p2.y - tri.p0.y, (double)tri.p2.z - tri.p0.z };
            const double pv[3] = { d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0] };
            const double det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
            if (det == 0.0) { continue; }
            const double inv = 1.0 / det;
            const double tv[3] = { o[0] - v0[0], o[1] - v0[1], o[2] - v0[2] };
            const double u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * inv;
            if (u < 0.0 || u > 1.0) { continue; }
            const double qv[3] = { tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0] };
            const double v = (d[0] * qv[0] + d[1] * qv[1] + d[2] * qv[2]) * inv;
            if (v < 0.0 || u + v > 1.0) { continue; }
            const double t = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * inv;
            if (t > tnear && t < best) { best = t; bestPrim = (int)p; }
        }
        for (size_t s = 0; s < impl.spheres.size(); s++) {
            const Sphere &sphere = impl.spheres[s];
            const double oc[3] = { o[0] - sphere.centerWorld.x, o[1] - sphere.centerWorld.y, o[2] - sphere.centerWorld.z };
            const double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            const double b = oc[0] * d[0] + oc[1] * d[1] + oc[2] * d[2];
            const double c = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - (double)sphere.radius * sphere.radius;
            const double disc = b * b - a * c;
            if (disc < 0.0) { continue; }
            const double root = std::sqrt(disc);
            double t = (-b - root) / a;
            if (!(t > tnear)) { t = (-b + root) / a; }
            if (t > tnear && t < best) { best = t; bestPrim = (int)(impl.triangles.size() + s); }
        }
        t_out[i] = bestPrim >= 0 ? best : 0.0;
        prim_out[i] = bestPrim;
    }
    return 0;
}

int oracle_count_exported_bvh(const float *nodes, size_t n_nodes, const float *tris, size_t n_tris,
                              const float *rays, size_t n, int any_hit, uint64_t *counts)
{
    /* Layout exported by pathed_hip_scene_export_bvh (see include/pathed_hip.h):
     * node = 8 x float4 with the four children in the components:
     *   (lo.x[4]) (lo.y[4]) (lo.z[4]) (hi.x[4]) (hi.y[4]) (hi.z[4]) (ref[4]) (unused);
     *   ref >= 0 inner node, ref <= -2 leaf with -ref - 1 = (first triangle << 3) | count,
     *   ref == INT_MIN empty slot;
     * triangles are 3 x float4: (v0.xyz, prim) (e1.xyz, -) (e2.xyz, -).
     * The walk visits the hit children leaves first, then near to far, like the kernel. */
    if (!nodes || !tris || !counts) { return -1; }
    uint64_t boxes = 0, triangles = 0;
    for (size_t i = 0; i < n; i++) {
        const float *r = rays + 8 * i;
        const Vec3 o = v3(r[0], r[1], r[2]);
        const Vec3 d = v3(r[4], r[5], r[6]);
        const Vec3 invD = v3(1.f / d.x, 1.f / d.y, 1.f / d.z);
        const float tnear = r[3];
        float best = r[7];
        int bestPrim = -1;
        bool done = false;

        auto testLeaf = [&](int first, int count) {
            for (int k = 0; k < count && !done; k++) {
                if ((size_t)(first + k) >= n_tris) { return; }
                const float *tri = tris + 12 * (size_t)(first + k);
                int prim;
                std::memcpy(&prim, tri + 3, 4);
                float t, u, v;
                triangles++;
                if (!intersectTriangle(o, d, v3(tri[0], tri[1], tri[2]), v3(tri[4], tri[5], tri[6]), v3(tri[8], tri[9], tri[10]), &t, &u, &v)) { continue; }
                if (!(t > tnear)) { continue; }
                if (any_hit) {
                    if (t <= best) { done = true; }
                } else {
                    const bool closer = (bestPrim < 0) ? (t <= best) : (t < best || (t == best && prim < bestPrim));
                    if (closer) { best = t; bestPrim = prim; }
                }
            }
        };

        if (n_nodes == 0) {
            testLeaf(0, (int)n_tris);
            continue;
        }
        int stack[256];
        int sp = 0;
        stack[sp++] = 0;
        while (sp > 0 && !done) {
            const int entry = stack[--sp];
            if (entry < 0) {
                const int leaf = -entry - 1;
                testLeaf(leaf >> 3, leaf & 7);
                continue;
            }
            if ((size_t)entry >= n_nodes) { return -2; }
            const float *node = nodes + 32 * (size_t)entry;
            uint32_t keys[4];
            int refs[4];
            int hits = 0;
            for (int child = 0; child < 4; child++) {
                int ref;
                std::memcpy(&ref, node + 24 + child, 4);
                if (ref == INT32_MIN) { continue; } /* empty child slot */
                boxes++;
                const float lo[3] = { node[child], node[4 + child], node[8 + child] };
                const float hi[3] = { node[12 + child], node[16 + child], node[20 + child] };
                float tEntry;
                if (!slabTestEntry(lo, hi, o, invD, tnear, best, &tEntry)) { continue; }
                uint32_t bits;
                std::memcpy(&bits, &tEntry, 4);
                keys[hits] = (ref >= 0 ? 0x80000000u : 0u) | ((bits >> 1) & 0x7FFFFFFCu) | (uint32_t)child;
                refs[hits] = ref;
                hits++;
            }
            for (int a = 1; a < hits; a++) { /* insertion sort, ascending keys */
                const uint32_t key = keys[a];
                const int ref = refs[a];
                int b = a - 1;
                while (b >= 0 && keys[b] > key) { keys[b + 1] = keys[b]; refs[b + 1] = refs[b]; b--; }
                keys[b + 1] = key;
                refs[b + 1] = ref;
            }
            for (int k = hits - 1; k >= 0; k--) { /* far first: the nearest is popped next */
                if (sp < 255) { stack[sp++] = refs[k]; }
            }
        }
    }
    counts[0] = boxes;
    counts[1] = triangles;
    return 0;
}

/* -------------------------------------------------------------- oracle_eval */

static Material materialFromFloats(const float *p)
{
    PathedMaterial m;
    std::memset(&m, 0, sizeof m);
    m.type = (int)p[0];
    m.albedo_type = (int)p[1];
    for (int i = 0; i < 3; i++) {
        m.diffuse[i] = p[2 + i];
        m.emit[i] = p[5 + i];
        m.checker_on[i] = p[8 + i];
        m.checker_off[i] = p[11 + i];
    }
    m.checker_res[0] = p[14];
    m.checker_res[1] = p[15];
    m.sigma = p[16];
    m.alpha = p[17];
    m.ior = p[18];
    m.distribution = (int)p[19];
    return materialFromDesc(m);
}

static Intersection intersectionFromFloats(const float *p)
{
    /* normal(3) shadingNormal(3) wo(3) uv(2) */
    Intersection isect;
    std::memset(&isect, 0, sizeof isect);
    isect.hit = true;
    isect.t = 1.f;
    isect.point = v3(0.f, 0.f, 0.f);
    isect.normal = v3(p[0], p[1], p[2]);
    isect.shadingNormal = v3(p[3], p[4], p[5]);
    isect.wo = v3(p[6], p[7], p[8]);
    isect.u = p[9];
    isect.v = p[10];
    isect.frame = normalToWorldSpace(isect.shadingNormal, isect.wo);
    return isect;
}

int oracle_eval(const char *fn, const float *in, int n_in, float *out, int n_out)
{
    const std::string name(fn ? fn : "");
    auto need = [&](int inputs, int outputs) { return n_in >= inputs && n_out >= outputs; };

    if (name == "reflect") {
        if (!need(6, 3)) { return -2; }
        const Vec3 r = reflect(v3(in[0], in[1], in[2]), v3(in[3], in[4], in[5]));
        out[0] = r.x; out[1] = r.y; out[2] = r.z;
        return 3;
    }
    if (name == "texture_lookup") {
        /* in: width height u v, then 3*width*height texel bytes as floats; out: rgb */
        if (!need(4, 3)) { return -2; }
        const int width = (int)in[0], height = (int)in[1];
        if (width < 1 || height < 1 || n_in < 4 + 3 * width * height) { return -2; }
        std::vector<unsigned char> bytes((size_t)3 * width * height);
        for (size_t k = 0; k < bytes.size(); k++) { bytes[k] = (unsigned char)in[4 + k]; }
        Material m = materialFromFloats(std::vector<float>(20, 0.f).data());
        m.albedoType = PATHED_ALBEDO_TEXTURE;
        m.texData = bytes.data();
        m.texWidth = width;
        m.texHeight = height;
        Intersection isect;
        std::memset(&isect, 0, sizeof isect);
        isect.u = in[2];
        isect.v = in[3];
        const Color c = textureLookup(m, isect);
        out[0] = c.r; out[1] = c.g; out[2] = c.b;
        return 3;
    }
    if (name == "frame") {
        if (!need(6, 9)) { return -2; }
        const Frame f = normalToWorldSpace(v3(in[0], in[1], in[2]), v3(in[3], in[4], in[5]));
        out[0] = f.xAxis.x; out[1] = f.normal.x; out[2] = f.zAxis.x;
        out[3] = f.xAxis.y; out[4] = f.normal.y; out[5] = f.zAxis.y;
        out[6] = f.xAxis.z; out[7] = f.normal.z; out[8] = f.zAxis.z;
        return 9;
    }
    if (name == "frame1") {
        if (!need(3, 9)) { return -2; }
        const Frame f = normalToWorldSpace1(v3(in[0], in[1], in[2]));
        out[0] = f.xAxis.x; out[1] = f.normal.x; out[2] = f.zAxis.x;
        out[3] = f.xAxis.y; out[4] = f.normal.y; out[5] = f.zAxis.y;
        out[6] = f.xAxis.z; out[7] = f.normal.z; out[8] = f.zAxis.z;
        return 9;
    }
    if (name == "camera_ray") {
        /* origin3 target3 up3 fov W H flip row col */
        if (!need(15, 6)) { return -2; }
        PathedCamera desc;
        for (int i = 0; i < 3; i++) { desc.origin[i] = in[i]; desc.target[i] = in[3 + i]; desc.up[i] = in[6 + i]; }
        desc.vertical_fov = in[9];
        desc.width = (int)in[10];
        desc.height = (int)in[11];
        desc.flip_handedness = (int)in[12];
        Camera camera;
        camera.build(desc);
        Vec3 o, d;
        camera.generateRay(in[13], in[14], &o, &d);
        out[0] = o.x; out[1] = o.y; out[2] = o.z; out[3] = d.x; out[4] = d.y; out[5] = d.z;
        return 6;
    }
    if (name == "cosine_hemisphere") {
        if (!need(2, 4)) { return -2; }
        Rng random = scriptedRng(in, 2);
        const Vec3 v = cosineSampleHemisphere(random);
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = cosineHemispherePdf(v);
        return 4;
    }
    if (name == "spherical") {
        if (!need(3, 2)) { return -2; }
        cartesianToSpherical(v3(in[0], in[1], in[2]), &out[0], &out[1]);
        return 2;
    }
    if (name == "fresnel") {
        if (!need(3, 1)) { return -2; }
        out[0] = dielectricReflectance(in[0], in[1], in[2]);
        return 1;
    }
    if (name == "refract") {
        if (!need(5, 4)) { return -2; }
        Vec3 wt;
        const bool ok = snellRefract(v3(in[0], in[1], in[2]), &wt, in[3], in[4]);
        out[0] = ok ? 1.f : 0.f; out[1] = wt.x; out[2] = wt.y; out[3] = wt.z;
        return 4;
    }
    if (name == "material_f") {
        /* material(20) isect(11) wi(3) -> f(3) pdf */
        if (!need(34, 4)) { return -2; }
        const Material m = materialFromFloats(in);
        const Intersection isect = intersectionFromFloats(in + 20);
        float pdf = 0.f;
        const Color f = materialF(m, isect, v3(in[31], in[32], in[33]), &pdf);
        out[0] = f.r; out[1] = f.g; out[2] = f.b; out[3] = pdf;
        return 4;
    }
    if (name == "material_sample") {
        /* material(20) isect(11) u(3) -> wi(3) pdf thr(3) */
        if (!need(34, 7)) { return -2; }
        const Material m = materialFromFloats(in);
        const Intersection isect = intersectionFromFloats(in + 20);
        Rng random = scriptedRng(in + 31, 3);
        const BSDFSample s = materialSample(m, isect, random);
        out[0] = s.wiWorld.x; out[1] = s.wiWorld.y; out[2] = s.wiWorld.z; out[3] = s.pdf;
        out[4] = s.throughput.r; out[5] = s.throughput.g; out[6] = s.throughput.b;
        return 7;
    }
    if (name == "beckmann") {
        /* alpha wh(3) wo(3) wi(3) -> D pdf G */
        if (!need(10, 3)) { return -2; }
        out[0] = beckmannD(in[0], v3(in[1], in[2], in[3]));
        out[1] = beckmannPdf(in[0], v3(in[1], in[2], in[3]));
        out[2] = beckmannG(in[0], v3(in[4], in[5], in[6]), v3(in[7], in[8], in[9]));
        return 3;
    }
    if (name == "ggx") {
        /* alpha wh(3) wo(3) wi(3) -> D pdf G */
        if (!need(10, 3)) { return -2; }
        out[0] = ggxD(in[0], v3(in[1], in[2], in[3]));
        out[1] = ggxD(in[0], v3(in[1], in[2], in[3])) * fabsf(in[2]);
        out[2] = ggxG1(in[0], v3(in[4], in[5], in[6])) * ggxG1(in[0], v3(in[7], in[8], in[9]));
        return 3;
    }
    if (name == "ggx_sample") {
        if (!need(3, 3)) { return -2; }
        Rng random = scriptedRng(in + 1, 2);
        const Vec3 wh = ggxSampleWh(in[0], random);
        out[0] = wh.x; out[1] = wh.y; out[2] = wh.z;
        return 3;
    }
    if (name == "beckmann_sample") {
        if (!need(3, 3)) { return -2; }
        Rng random = scriptedRng(in + 1, 2);
        const Vec3 wh = beckmannSampleWh(in[0], random);
        out[0] = wh.x; out[1] = wh.y; out[2] = wh.z;
        return 3;
    }
    if (name == "triangle_sample") {
        /* p0 p1 p2 u1 u2 -> point(3) normal(3) invPDF */
        if (!need(11, 7)) { return -2; }
        Triangle tri;
        std::memset(&tri, 0, sizeof tri);
        tri.p0 = v3(in[0], in[1], in[2]); tri.p1 = v3(in[3], in[4], in[5]); tri.p2 = v3(in[6], in[7], in[8]);
        Rng random = scriptedRng(in + 9, 2);
        const SurfaceSample s = triangleSample(tri, random);
        out[0] = s.point.x; out[1] = s.point.y; out[2] = s.point.z;
        out[3] = s.normal.x; out[4] = s.normal.y; out[5] = s.normal.z; out[6] = s.invPDF;
        return 7;
    }
    if (name == "triangle_pdf") {
        /* p0 p1 p2 point(3) ref(3) -> solid-angle pdf, area */
        if (!need(15, 2)) { return -2; }
        Triangle tri;
        std::memset(&tri, 0, sizeof tri);
        tri.p0 = v3(in[0], in[1], in[2]); tri.p1 = v3(in[3], in[4], in[5]); tri.p2 = v3(in[6], in[7], in[8]);
        out[0] = trianglePdfSolidAngle(tri, v3(in[9], in[10], in[11]), v3(in[12], in[13], in[14]));
        out[1] = triangleArea(tri);
        return 2;
    }
    if (name == "sphere_sample") {
        /* center(3) radius ref(3) u1 u2 -> point(3) normal(3) invPDF measure */
        if (!need(9, 8)) { return -2; }
        Sphere s;
        s.centerWorld = s.centerSample = v3(in[0], in[1], in[2]);
        s.radius = in[3];
        s.material = 0;
        Rng random = scriptedRng(in + 7, 2);
        const SurfaceSample r = sphereSample(s, v3(in[4], in[5], in[6]), random);
        out[0] = r.point.x; out[1] = r.point.y; out[2] = r.point.z;
        out[3] = r.normal.x; out[4] = r.normal.y; out[5] = r.normal.z;
        out[6] = r.invPDF; out[7] = (r.measure == SolidAngle) ? 0.f : 1.f;
        return 8;
    }
    if (name == "sphere_pdf") {
        if (!need(7, 1)) { return -2; }
        Sphere s;
        s.centerWorld = s.centerSample = v3(in[0], in[1], in[2]);
        s.radius = in[3];
        s.material = 0;
        out[0] = spherePdfSolidAngle(s, v3(in[4], in[5], in[6]));
        return 1;
    }
    if (name == "area_to_solid_angle") {
        if (!need(10, 1)) { return -2; }
        out[0] = areaToSolidAngle(in[0], v3(in[1], in[2], in[3]), v3(in[4], in[5], in[6]), v3(in[7], in[8], in[9]));
        return 1;
    }
    if (name == "mis_balance") {
        if (!need(2, 1)) { return -2; }
        out[0] = (1 * in[0]) / (1 * in[0] + 1 * in[1]);
        return 1;
    }
    if (name == "bounce_controller") {
        /* start last bounce -> counts done */
        if (!need(3, 2)) { return -2; }
        out[0] = OracleSceneImpl::checkCounts((int)in[0], (int)in[1], (int)in[2]) ? 1.f : 0.f;
        out[1] = OracleSceneImpl::checkDone((int)in[1], (int)in[2]) ? 1.f : 0.f;
        return 2;
    }
    if (name == "distribution") {
        /* n values... u -> index pdf ; then pdf(index) for checking */
        if (n_in < 2) { return -2; }
        const int n = (int)in[0];
        if (!need(2 + n, 3)) { return -2; }
        Distribution d;
        d.build(in + 1, (size_t)n);
        Rng random = scriptedRng(in + 1 + n, 1);
        float pdf = 0.f;
        const int index = d.sample(&pdf, random);
        out[0] = (float)index; out[1] = pdf; out[2] = d.pdf(index);
        return 3;
    }
    return -1;
}

/* Environment-light function-level checks need an image; they go through a scene. */
int oracle_env_eval(OracleScene *scene, const char *fn, const float *in, int n_in, float *out, int n_out)
{
    if (!scene || !scene->impl.hasEnv) { return -3; }
    const EnvLight &env = scene->impl.env;
    const std::string name(fn ? fn : "");
    if (name == "env_emit") {
        if (n_in < 3 || n_out < 3) { return -2; }
        const Color c = env.emit(v3(in[0], in[1], in[2]));
        out[0] = c.r; out[1] = c.g; out[2] = c.b;
        return 3;
    }
    if (name == "env_pdf") {
        if (n_in < 3 || n_out < 1) { return -2; }
        out[0] = env.emitPDF(v3(in[0], in[1], in[2]));
        return 1;
    }
    if (name == "env_sample") {
        /* point(3) u1 u2 -> point(3) normal(3) invPDF */
        if (n_in < 5 || n_out < 7) { return -2; }
        Rng random = scriptedRng(in + 3, 2);
        const SurfaceSample s = env.sample(v3(in[0], in[1], in[2]), random);
        out[0] = s.point.x; out[1] = s.point.y; out[2] = s.point.z;
        out[3] = s.normal.x; out[4] = s.normal.y; out[5] = s.normal.z; out[6] = s.invPDF;
        return 7;
    }
    return -1;
}

}  // extern "C"
