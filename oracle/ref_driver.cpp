/*
 * ref_driver.cpp — dumps golden vectors from the REFERENCE's own object code.
 *
 * TEST INFRASTRUCTURE.  Built only where /root/reference exists (Makefile.ref), from
 * the reference translation units where they lie; nothing of the reference is copied
 * into this repository — only the numbers this tool prints (tests/golden/*.jsonl).
 *
 * It links the reference TUs that compile without Embree (vector, transform,
 * monte_carlo, lambertian, oren_nayar, microfacet, beckmann, plastic, glass, mirror,
 * fresnel, snell, triangle, distribution, camera, environment_light, ...) and calls
 * their functions on seeded inputs.  The reference RandomGenerator cannot be seeded
 * through its interface (src/random_generator.cpp:4-6), so this file reaches its
 * mt19937 member directly (`#define private public` around the reference headers,
 * which does not change layout) to (a) seed it and (b) replay a copy of its state to
 * learn which numbers the reference function consumed.  No reference header, library
 * or tool is replaced by a stand-in.
 *
 * Output: one JSON object per line: {"fn": name, "in": [...], "out": [...]} with the
 * argument layouts of oracle_eval (oracle/oracle.cpp).
 */
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <random>
#include <sstream>
#include <string>
#include <vector>
#include <map>
#include <queue>
#include <algorithm>
#include <functional>
#include <mutex>
#include <assert.h>

#define TINYEXR_IMPLEMENTATION
#include "tinyexr.h"
/* the reference defines the stb_image implementation in its executables (app/main.cpp:16) */
#define STB_IMAGE_IMPLEMENTATION
#include "stb_image.h"
#define STB_IMAGE_WRITE_IMPLEMENTATION
#include "stb_image_write.h"

#define private public
#define protected public
#include "mtl_parser.h"
#include "random_generator.h"
#include "transform.h"
#include "environment_light.h"
#include "texture.h"
#undef private
#undef protected

#include "beckmann.h"
#include "bounce_controller.h"
#include "camera.h"
#include "checkerboard.h"
#include "coordinate.h"
#include "distribution.h"
#include "fresnel.h"
#include "ggx.h"
#include "glass.h"
#include "intersection.h"
#include "lambertian.h"
#include "measure.h"
#include "microfacet.h"
#include "mirror.h"
#include "mis.h"
#include "monte_carlo.h"
#include "oren_nayar.h"
#include "plastic.h"
#include "ray.h"
#include "snell.h"
#include "string_util.h"
#include "triangle.h"
#include "util.h"
#include "vector.h"

static FILE *g_out = stdout;

/* text records: {"fn": name, "text": "...", "tokens": [...]} -- JSON string escaping for the few characters the inputs hold */
static std::string quoted(const std::string &text)
{
    std::string out = "\"";
    for (char c : text) {
        if (c == '\\' || c == '"') { out += '\\'; out += c; }
        else if (c == '\t') { out += "\\t"; }
        else if (c == '\n') { out += "\\n"; }
        else if (c == '\r') { out += "\\r"; }
        else { out += c; }
    }
    return out + "\"";
}

static void emit(const char *fn, const std::vector<float> &in, const std::vector<float> &out)
{
    fprintf(g_out, "{\"fn\": \"%s\", \"in\": [", fn);
    for (size_t i = 0; i < in.size(); i++) { fprintf(g_out, "%s%.9g", i ? ", " : "", in[i]); }
    fprintf(g_out, "], \"out\": [");
    for (size_t i = 0; i < out.size(); i++) {
        const float v = out[i];
        if (std::isnan(v)) { fprintf(g_out, "%s\"nan\"", i ? ", " : ""); }
        else if (std::isinf(v)) { fprintf(g_out, "%s\"%sinf\"", i ? ", " : "", v < 0 ? "-" : ""); }
        else { fprintf(g_out, "%s%.9g", i ? ", " : "", v); }
    }
    fprintf(g_out, "]}\n");
}

/* test-input generator (independent of the reference's RandomGenerator) */
static std::mt19937 g_inputs(20240607u);

static float uniform01()
{
    return (float)(g_inputs() >> 8) * (1.f / 16777216.f);
}

static Vector3 randomUnit()
{
    while (true) {
        const float x = 2.f * uniform01() - 1.f;
        const float y = 2.f * uniform01() - 1.f;
        const float z = 2.f * uniform01() - 1.f;
        const float n = x * x + y * y + z * z;
        if (n > 1e-3f && n <= 1.f) { return Vector3(x, y, z).normalized(); }
    }
}

/* direction in the hemisphere of `n` (cosine >= minCos) */
static Vector3 randomAbout(const Vector3 &n, float minCos)
{
    while (true) {
        Vector3 v = randomUnit();
        if (v.dot(n) >= minCos) { return v; }
    }
}

/* the numbers a reference call is about to draw: replay a copy of the generator */
static std::vector<float> peek(RandomGenerator &random, int count)
{
    std::mt19937 generator = random.m_generator;
    std::uniform_real_distribution<float> distribution = random.m_distribution;
    std::vector<float> values;
    for (int i = 0; i < count; i++) { values.push_back(distribution(generator)); }
    return values;
}

static void push3(std::vector<float> &v, const Vector3 &a) { v.push_back(a.x()); v.push_back(a.y()); v.push_back(a.z()); }
static void push3(std::vector<float> &v, const Point3 &a) { v.push_back(a.x()); v.push_back(a.y()); v.push_back(a.z()); }
static void push3(std::vector<float> &v, const Color &a) { v.push_back(a.r()); v.push_back(a.g()); v.push_back(a.b()); }

/* material parameter block of oracle_eval: 20 floats */
struct MaterialSpec {
    int type = 0;       /* PATHED_MAT_* numbering */
    int albedoType = 0;
    Color diffuse = Color(0.f);
    Color emit = Color(0.f);
    Color on = Color(0.f), off = Color(0.f);
    float resU = 0.f, resV = 0.f;
    float sigma = 0.f, alpha = 0.f, ior = 1.4f;
    int distribution = 0; /* 0 Beckmann, 1 GGX */

    std::unique_ptr<MicrofacetDistribution> makeDistribution() const
    {
        if (distribution == 1) { return std::make_unique<GGX>(alpha); }
        return std::make_unique<Beckmann>(alpha);
    }

    void push(std::vector<float> &v) const
    {
        v.push_back((float)type); v.push_back((float)albedoType);
        push3(v, diffuse); push3(v, emit); push3(v, on); push3(v, off);
        v.push_back(resU); v.push_back(resV);
        v.push_back(sigma); v.push_back(alpha); v.push_back(ior);
        v.push_back((float)distribution);
    }

    std::shared_ptr<Material> build() const
    {
        switch (type) {
        case 0:
            if (albedoType == 1) {
                auto checker = std::make_shared<Checkerboard>(on, off, UV{ resU, resV });
                return std::make_shared<Lambertian>(checker, emit);
            }
            return std::make_shared<Lambertian>(diffuse, emit);
        case 1: return std::make_shared<OrenNayar>(diffuse, sigma);
        case 2: return std::make_shared<Microfacet>(makeDistribution());
        case 3: return std::make_shared<Plastic>(diffuse, makeDistribution());
        case 4: return std::make_shared<Glass>(ior);
        default: return std::make_shared<Mirror>();
        }
    }
};

struct IsectSpec {
    Vector3 normal = Vector3(0.f, 1.f, 0.f);
    Vector3 shadingNormal = Vector3(0.f, 1.f, 0.f);
    Vector3 wo = Vector3(0.f, 1.f, 0.f);
    UV uv = { 0.f, 0.f };

    void push(std::vector<float> &v) const
    {
        push3(v, normal); push3(v, shadingNormal); push3(v, wo);
        v.push_back(uv.u); v.push_back(uv.v);
    }

    Intersection build(Material *material) const
    {
        return Intersection(true, 1.f, Point3(0.f, 0.f, 0.f), wo, normal, shadingNormal, uv, material, nullptr);
    }
};

static IsectSpec randomIsect(bool allowBackside, bool smoothNormal)
{
    IsectSpec spec;
    spec.normal = randomUnit();
    spec.shadingNormal = smoothNormal ? randomAbout(spec.normal, 0.8f) : spec.normal;
    if (allowBackside && uniform01() < 0.25f) {
        spec.wo = randomAbout(-spec.shadingNormal, 0.05f);
    } else {
        spec.wo = randomAbout(spec.shadingNormal, 0.05f);
    }
    spec.uv = { uniform01(), uniform01() };
    return spec;
}

static void dumpMaterial(const MaterialSpec &spec, int cases, bool allowBackside, unsigned seedBase)
{
    std::shared_ptr<Material> material = spec.build();
    for (int i = 0; i < cases; i++) {
        const IsectSpec isectSpec = randomIsect(allowBackside, (i % 2) == 1);
        const Intersection isect = isectSpec.build(material.get());

        /* f / pdf for an arbitrary direction */
        {
            const Vector3 wi = (i % 5 == 4) ? randomUnit() : randomAbout(isectSpec.shadingNormal, 0.02f);
            float pdf = 0.f;
            const Color f = material->f(isect, wi, &pdf);
            std::vector<float> in, out;
            spec.push(in); isectSpec.push(in); push3(in, wi);
            push3(out, f); out.push_back(pdf);
            emit("material_f", in, out);
        }
        /* sample */
        {
            RandomGenerator random;
            random.m_generator.seed(seedBase + (unsigned)i);
            const std::vector<float> u = peek(random, 3);
            const BSDFSample sample = material->sample(isect, random);
            std::vector<float> in, out;
            spec.push(in); isectSpec.push(in);
            in.insert(in.end(), u.begin(), u.end());
            push3(out, sample.wiWorld); out.push_back(sample.pdf); push3(out, sample.throughput);
            emit("material_sample", in, out);
        }
    }
}

int main(int argc, char **argv)
{
    /* refdump --image-hash <file>...: size and FNV-1a hash of stbi_load(file, .., 3), for spot checks of the
     * host decoder on image files too large to keep as fixtures */
    if (argc > 2 && std::string(argv[1]) == "--image-hash") {
        for (int a = 2; a < argc; a++) {
            int width = 0, height = 0, channels = 0;
            unsigned char *data = stbi_load(argv[a], &width, &height, &channels, 3);
            if (!data) { printf("%s: cannot load\n", argv[a]); continue; }
            unsigned long long hash = 1469598103934665603ull;
            for (size_t i = 0; i < (size_t)3 * width * height; i++) { hash = (hash ^ data[i]) * 1099511628211ull; }
            printf("%s %d %d %016llx\n", argv[a], width, height, hash);
            stbi_image_free(data);
        }
        return 0;
    }
    if (argc > 1) {
        g_out = fopen(argv[1], "w");
        if (!g_out) { fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
    }
    const char *envPath = (argc > 2) ? argv[2] : "/root/reference/test_scenes/1_pixel_test.exr";

    /* ---- known-answer tests the reference's own test suite holds ------------- */
    {
        /* test/vector_test.cpp:6-14 */
        const Vector3 v(-0.5f, -0.5f, 0.f), n(0.f, 1.f, 0.f);
        std::vector<float> in, out;
        push3(in, v); push3(in, n); push3(out, v.reflect(n));
        emit("reflect", in, out);
        for (int i = 0; i < 8; i++) {
            const Vector3 a = randomUnit(), b = randomUnit();
            in.clear(); out.clear();
            push3(in, a); push3(in, b); push3(out, a.reflect(b));
            emit("reflect", in, out);
        }
    }
    {
        /* test/transform_test.cpp:6-14 and more frames */
        auto dumpFrame = [](const Vector3 &normal, const Vector3 &wo) {
            const Transform t = normalToWorldSpace(normal, wo);
            std::vector<float> in, out;
            push3(in, normal); push3(in, wo);
            for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) { out.push_back(t.m_matrix[r][c]); } }
            emit("frame", in, out);
        };
        dumpFrame(Vector3(1, 2, 3).normalized(), Vector3(1, 0, 0));
        for (int i = 0; i < 24; i++) {
            const Vector3 n = randomUnit();
            dumpFrame(n, randomAbout(n, -0.5f));
        }
        for (int i = 0; i < 6; i++) {
            const Vector3 n = randomUnit();
            dumpFrame(n, n); /* the `normal == rayDirection` branch */
        }
        for (int i = 0; i < 12; i++) {
            const Vector3 n = (i == 0) ? Vector3(0.f, 1.f, 0.f) : (i == 1) ? Vector3(1.f, 0.f, 0.f) : randomUnit();
            const Transform t = normalToWorldSpace(n);
            std::vector<float> in, out;
            push3(in, n);
            for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) { out.push_back(t.m_matrix[r][c]); } }
            emit("frame1", in, out);
        }
    }

    /* ---- camera rays: src/camera.cpp:32-47 ------------------------------------ */
    {
        struct Cam { Point3 o, t; Vector3 up; float fovDeg; int w, h; bool flip; };
        const Cam cams[] = {
            { Point3(0.f, 1.f, 6.8f), Point3(0.f, 1.f, 0.f), Vector3(0.f, 1.f, 0.f), 19.5f, 256, 256, false },   /* scenes/cornell.json */
            { Point3(0.f, 2.f, 15.f), Point3(0.f, 1.69521f, 14.0476f), Vector3(0.f, 0.952421f, -0.304787f), 28.0000262073138f, 1024, 1024, true }, /* mis-pbrt */
            { Point3(23.895f, 11.2207f, 0.0400773f), Point3(-0.953633f, 2.17253f, -0.0972613f), Vector3(0.f, 1.f, 0.f), 35.f, 640, 360, false }, /* teapot */
            { Point3(277.f, -240.f, 250.f), Point3(0.f, 60.f, -30.f), Vector3(0.f, 0.f, 1.f), 33.f, 1920, 1080, true }, /* dragon */
        };
        for (const Cam &cam : cams) {
            const float fov = cam.fovDeg / 180.f * M_PI;
            Camera camera(cam.o, cam.t, cam.up, fov, { cam.w, cam.h }, cam.flip);
            for (int i = 0; i < 12; i++) {
                const float row = uniform01() * cam.h - 0.5f;
                const float col = uniform01() * cam.w - 0.5f;
                const Ray ray = camera.generateRay(row, col);
                std::vector<float> in, out;
                push3(in, cam.o); push3(in, cam.t); push3(in, cam.up);
                in.push_back(fov); in.push_back((float)cam.w); in.push_back((float)cam.h); in.push_back(cam.flip ? 1.f : 0.f);
                in.push_back(row); in.push_back(col);
                push3(out, ray.origin()); push3(out, ray.direction());
                emit("camera_ray", in, out);
            }
        }
    }

    /* ---- samplers ---------------------------------------------------------------- */
    for (int i = 0; i < 32; i++) {
        RandomGenerator random;
        random.m_generator.seed(1000u + (unsigned)i);
        const std::vector<float> u = peek(random, 2);
        const Vector3 v = CosineSampleHemisphere(random);
        std::vector<float> out;
        push3(out, v); out.push_back(CosineHemispherePdf(v));
        emit("cosine_hemisphere", u, out);
    }
    for (int i = 0; i < 32; i++) {
        const Vector3 v = randomUnit();
        float phi, theta;
        cartesianToSpherical(v, &phi, &theta);
        std::vector<float> in;
        push3(in, v);
        emit("spherical", in, { phi, theta });
    }

    /* ---- Fresnel / Snell (cf. app/testbed.cpp:27-57) ------------------------------ */
    {
        const float etas[] = { 1.1f, 1.4f, 1.5f, 2.0f };
        for (float eta : etas) {
            for (int i = 0; i <= 24; i++) {
                const float cosTheta = (float)i / 24.f;
                emit("fresnel", { cosTheta, 1.f, eta }, { Fresnel::dielectricReflectance(cosTheta, 1.f, eta) });
                emit("fresnel", { cosTheta, eta, 1.f }, { Fresnel::dielectricReflectance(cosTheta, eta, 1.f) });
            }
        }
        for (int i = 0; i < 48; i++) {
            Vector3 wo = randomUnit();
            const float etaI = (i % 2) ? 1.f : 1.4f;
            const float etaT = (i % 2) ? 1.4f : 1.f;
            Vector3 wt(0.f);
            const bool ok = Snell::refract(wo, &wt, etaI, etaT);
            std::vector<float> in, out;
            push3(in, wo); in.push_back(etaI); in.push_back(etaT);
            out.push_back(ok ? 1.f : 0.f); push3(out, wt);
            emit("refract", in, out);
        }
    }

    /* ---- Beckmann pieces ----------------------------------------------------------- */
    {
        const float alphas[] = { 0.005f, 0.02f, 0.05f, 0.1f, 0.4f };
        for (float alpha : alphas) {
            Beckmann beckmann(alpha);
            for (int i = 0; i < 16; i++) {
                /* half vectors concentrated near the pole, where the reference's
                 * TangentFrame::clamp quirk (include/tangent_frame.h:13-37) matters */
                const float spread = (i < 8) ? alpha * 3.f : 0.9f;
                const float theta = uniform01() * spread;
                const float phi = uniform01() * 6.2831853f;
                const Vector3 wh(sinf(theta) * cosf(phi), cosf(theta), sinf(theta) * sinf(phi));
                const Vector3 up(0.f, 1.f, 0.f);
                const Vector3 wo = randomAbout(up, 0.02f);
                const Vector3 wi = randomAbout(up, 0.02f);
                std::vector<float> in;
                in.push_back(alpha); push3(in, wh); push3(in, wo); push3(in, wi);
                emit("beckmann", in, { beckmann.D(wh), beckmann.pdf(wh), beckmann.G(wo, wi) });
            }
            for (int i = 0; i < 8; i++) {
                RandomGenerator random;
                random.m_generator.seed(2000u + (unsigned)i);
                const std::vector<float> u = peek(random, 2);
                const Vector3 wh = beckmann.sampleWh(Vector3(0.f, 1.f, 0.f), random);
                std::vector<float> in = { alpha, u[0], u[1] }, out;
                push3(out, wh);
                emit("beckmann_sample", in, out);
            }
        }
    }

    /* ---- GGX pieces (reference src/ggx.cpp) -------------------------------------------- */
    {
        const float alphas[] = { 0.02f, 0.1f, 0.4f };
        for (float alpha : alphas) {
            GGX ggx(alpha);
            for (int i = 0; i < 16; i++) {
                const float spread = (i < 8) ? alpha * 3.f : 0.9f;
                const float theta = uniform01() * spread;
                const float phi = uniform01() * 6.2831853f;
                const Vector3 wh(sinf(theta) * cosf(phi), cosf(theta), sinf(theta) * sinf(phi));
                const Vector3 up(0.f, 1.f, 0.f);
                const Vector3 wo = randomAbout(up, 0.02f);
                const Vector3 wi = randomAbout(up, 0.02f);
                std::vector<float> in;
                in.push_back(alpha); push3(in, wh); push3(in, wo); push3(in, wi);
                emit("ggx", in, { ggx.D(wh), ggx.pdf(wh), ggx.G(wo, wi) });
            }
            for (int i = 0; i < 8; i++) {
                RandomGenerator random;
                random.m_generator.seed(2500u + (unsigned)i);
                const std::vector<float> u = peek(random, 2);
                const Vector3 wh = ggx.sampleWh(Vector3(0.f, 1.f, 0.f), random);
                std::vector<float> in = { alpha, u[0], u[1] }, out;
                push3(out, wh);
                emit("ggx_sample", in, out);
            }
        }
    }

    /* ---- materials: f / pdf / sample ------------------------------------------------ */
    {
        MaterialSpec lambert;
        lambert.type = 0; lambert.diffuse = Color(0.725f, 0.71f, 0.68f);
        dumpMaterial(lambert, 24, true, 3000u);

        MaterialSpec emitter = lambert;
        emitter.diffuse = Color(0.78f, 0.78f, 0.78f); emitter.emit = Color(17.f, 12.f, 4.f);
        dumpMaterial(emitter, 4, false, 3100u);

        MaterialSpec checker;
        checker.type = 0; checker.albedoType = 1;
        checker.on = Color(0.725f, 0.71f, 0.68f); checker.off = Color(0.325f, 0.31f, 0.25f);
        checker.resU = 20.f; checker.resV = 20.f;
        dumpMaterial(checker, 16, false, 3200u);

        const float sigmas[] = { 0.f, 0.3f, 0.5f, 1.f };
        for (float sigma : sigmas) {
            MaterialSpec oren;
            oren.type = 1; oren.diffuse = Color(0.6f, 0.5f, 0.4f); oren.sigma = sigma;
            dumpMaterial(oren, 16, true, 3300u + (unsigned)(sigma * 100));
        }

        const float alphas[] = { 0.005f, 0.02f, 0.05f, 0.1f, 0.3f };
        for (float alpha : alphas) {
            MaterialSpec micro;
            micro.type = 2; micro.alpha = alpha;
            dumpMaterial(micro, 16, true, 3500u + (unsigned)(alpha * 1000));

            MaterialSpec plastic;
            plastic.type = 3; plastic.alpha = alpha; plastic.diffuse = Color(0.07f, 0.09f, 0.13f);
            dumpMaterial(plastic, 24, true, 3700u + (unsigned)(alpha * 1000));
        }

        const float ggxAlphas[] = { 0.05f, 0.3f };
        for (float alpha : ggxAlphas) {
            MaterialSpec micro;
            micro.type = 2; micro.alpha = alpha; micro.distribution = 1;
            dumpMaterial(micro, 16, true, 3800u + (unsigned)(alpha * 1000));

            MaterialSpec plastic;
            plastic.type = 3; plastic.alpha = alpha; plastic.distribution = 1; plastic.diffuse = Color(0.1f, 0.1f, 0.4f);
            dumpMaterial(plastic, 16, true, 3850u + (unsigned)(alpha * 1000));
        }

        const float iors[] = { 1.4f, 1.5f, 1.1f };
        for (float ior : iors) {
            MaterialSpec glass;
            glass.type = 4; glass.ior = ior;
            dumpMaterial(glass, 32, true, 3900u + (unsigned)(ior * 100));
        }

        MaterialSpec mirror;
        mirror.type = 5;
        dumpMaterial(mirror, 16, true, 4000u);
    }

    /* ---- triangles ------------------------------------------------------------------- */
    for (int i = 0; i < 32; i++) {
        Point3 p0(uniform01() * 4.f - 2.f, uniform01() * 4.f - 2.f, uniform01() * 4.f - 2.f);
        Point3 p1(uniform01() * 4.f - 2.f, uniform01() * 4.f - 2.f, uniform01() * 4.f - 2.f);
        Point3 p2(uniform01() * 4.f - 2.f, uniform01() * 4.f - 2.f, uniform01() * 4.f - 2.f);
        if (i == 0) {
            /* first light triangle of scenes/CornellBox-Original.obj */
            p0 = Point3(-0.24f, 1.98f, 0.16f); p1 = Point3(-0.24f, 1.98f, -0.22f); p2 = Point3(0.23f, 1.98f, -0.22f);
        }
        Triangle triangle(p0, p1, p2);
        RandomGenerator random;
        random.m_generator.seed(5000u + (unsigned)i);
        const std::vector<float> u = peek(random, 2);
        const SurfaceSample sample = triangle.sample(random);
        {
            std::vector<float> in, out;
            push3(in, p0); push3(in, p1); push3(in, p2); in.push_back(u[0]); in.push_back(u[1]);
            push3(out, sample.point); push3(out, sample.normal); out.push_back(sample.invPDF);
            emit("triangle_sample", in, out);
        }
        {
            const Point3 reference(uniform01() * 6.f - 3.f, uniform01() * 6.f - 3.f, uniform01() * 6.f - 3.f);
            std::vector<float> in;
            push3(in, p0); push3(in, p1); push3(in, p2); push3(in, sample.point); push3(in, reference);
            emit("triangle_pdf", in, { triangle.pdf(sample.point, reference, Measure::SolidAngle), triangle.area() });
        }
    }

    /* ---- measure conversion / MIS / bounce window -------------------------------------- */
    for (int i = 0; i < 16; i++) {
        const float areaPDF = 0.1f + uniform01() * 10.f;
        const Point3 reference(uniform01() * 4.f - 2.f, uniform01() * 4.f - 2.f, uniform01() * 4.f - 2.f);
        const Point3 surface(uniform01() * 4.f - 2.f, uniform01() * 4.f - 2.f, uniform01() * 4.f - 2.f);
        const Vector3 normal = randomUnit();
        std::vector<float> in = { areaPDF };
        push3(in, reference); push3(in, surface); push3(in, normal);
        emit("area_to_solid_angle", in, { MeasureConversion::areaToSolidAngle(areaPDF, reference, surface, normal) });
    }
    for (int i = 0; i < 16; i++) {
        const float a = uniform01() * 20.f, b = uniform01() * 20.f;
        emit("mis_balance", { a, b }, { MIS::balanceWeight(1, 1, a, b) });
    }
    {
        const int windows[][2] = { { 0, 10 }, { 0, -1 }, { 2, 2 }, { 1, 3 }, { 0, 0 } };
        for (const auto &window : windows) {
            BounceController controller(window[0], window[1]);
            for (int bounce = 0; bounce <= 12; bounce++) {
                emit("bounce_controller", { (float)window[0], (float)window[1], (float)bounce },
                     { controller.checkCounts(bounce) ? 1.f : 0.f, controller.checkDone(bounce) ? 1.f : 0.f });
            }
        }
    }

    /* ---- Distribution -------------------------------------------------------------------- */
    for (int i = 0; i < 24; i++) {
        const int n = 2 + (int)(uniform01() * 14.f);
        std::vector<float> values;
        for (int k = 0; k < n; k++) { values.push_back(uniform01() < 0.3f ? 0.f : uniform01() * 5.f); }
        values[(size_t)(uniform01() * n) % (size_t)n] += 1.f;
        Distribution distribution(values);
        RandomGenerator random;
        random.m_generator.seed(6000u + (unsigned)i);
        const std::vector<float> u = peek(random, 1);
        float pdf = 0.f;
        const int index = distribution.sample(&pdf, random);
        std::vector<float> in = { (float)n };
        in.insert(in.end(), values.begin(), values.end());
        in.push_back(u[0]);
        emit("distribution", in, { (float)index, pdf, distribution.pdf(index) });
    }

    /* ---- EnvironmentLight on the reference's own test map ------------------------------- */
    {
        std::ifstream probe(envPath);
        if (probe.good()) {
            EnvironmentLight light(envPath, 1.f, Transform());
            std::vector<float> texels;
            for (int i = 0; i < light.m_width * light.m_height; i++) {
                const float r = light.m_data[4 * i + 0], g = light.m_data[4 * i + 1], b = light.m_data[4 * i + 2];
                if (r != 0.f || g != 0.f || b != 0.f) {
                    texels.push_back((float)(i % light.m_width));
                    texels.push_back((float)(i / light.m_width));
                    texels.push_back(r); texels.push_back(g); texels.push_back(b);
                }
            }
            emit("env_image", { (float)light.m_width, (float)light.m_height }, texels);

            /* directions through and around the lit texel plus random ones */
            for (int i = 0; i < 48; i++) {
                Vector3 direction = randomUnit();
                if (i < 24 && texels.size() >= 5) {
                    const float phi = (texels[0] + uniform01()) / light.m_width * M_TWO_PI;
                    const float theta = (texels[1] + uniform01() * (i < 12 ? 1.f : 3.f) - (i < 12 ? 0.f : 1.f)) / light.m_height * M_PI;
                    direction = sphericalToCartesian(phi, theta);
                }
                const Color le = light.emit(direction * -1.f);
                std::vector<float> in, out;
                push3(in, direction * -1.f); push3(out, le);
                emit("env_emit", in, out);
                in.clear();
                push3(in, direction);
                emit("env_pdf", in, { light.emitPDF(direction, Measure::SolidAngle) });
            }
            for (int i = 0; i < 8; i++) {
                RandomGenerator random;
                random.m_generator.seed(7000u + (unsigned)i);
                const std::vector<float> u = peek(random, 2);
                const Point3 point(uniform01(), uniform01(), uniform01());
                const SurfaceSample sample = light.sample(point, random);
                std::vector<float> in, out;
                push3(in, point); in.push_back(u[0]); in.push_back(u[1]);
                push3(out, sample.point); push3(out, sample.normal); out.push_back(sample.invPDF);
                emit("env_sample", in, out);
            }
        }
    }

    /* ---- Texture (src/texture.cpp) on the committed fixture files --------------------- */
    /* texture_image:  in = file index            out = width height, then the 8-bit RGB texels
     *                                                  stbi_load(path, .., 3) returned
     * texture_lookup: in = file index, u, v       out = Texture::lookup rgb */
    {
        const std::string directory = (argc > 3) ? argv[3] : "tests/golden/textures";
        const char *files[] = { "rgb8_7x5.png", "rgba16_4x3.png", "greyalpha8_3x4.png", "grey2_5x6.png",
                                "palette4_5x4.png", "rgb_2x3.ppm", "wood_48x32.png" };
        for (int f = 0; f < 7; f++) {
            const std::string path = directory + "/" + files[f];
            std::ifstream probe(path);
            if (!probe.good()) { fprintf(stderr, "texture fixture missing: %s\n", path.c_str()); continue; }
            Texture texture(path);
            texture.load();
            std::vector<float> image = { (float)texture.m_width, (float)texture.m_height };
            for (int i = 0; i < 3 * texture.m_width * texture.m_height; i++) { image.push_back((float)texture.m_data[i]); }
            emit("texture_image", { (float)f }, image);

            Intersection isect(true, 1.f, Point3(0.f, 0.f, 0.f), Vector3(0.f, 1.f, 0.f), Vector3(0.f, 1.f, 0.f),
                               Vector3(0.f, 1.f, 0.f), UV({ 0.f, 0.f }), nullptr, nullptr);
            const int lookups = f == 6 ? 24 : 40;
            for (int i = 0; i < lookups; i++) {
                float u = uniform01(), v = uniform01();
                if (i % 4 == 1) { u = u * 6.f - 3.f; v = v * 6.f - 3.f; }          /* wrapping, negative uv */
                if (i % 8 == 2) { u = (float)(i % 3); v = (float)(i % 5) - 2.f; }   /* integers: the flipped v lands on 1 */
                if (i % 8 == 6) { u = (0.5f + (float)(i % texture.m_width)) / (float)(texture.m_width - 1 > 0 ? texture.m_width - 1 : 1); }  /* rounding ties */
                isect.uv = { u, v };
                const Color c = texture.lookup(isect);
                std::vector<float> out;
                push3(out, c);
                emit("texture_lookup", { (float)f, u, v }, out);
            }
        }
    }

    /* ---- PIZ-compressed EXR files, written AND read back by the reference's vendored tinyexr ---- */
    /* The reference reads environment maps through tinyexr LoadEXR (src/environment_light.cpp:14-28),
     * which accepts PIZ.  Two fixture files are written next to the texture fixtures (data, generated
     * by the reference's own code) and what LoadEXR returns for them is dumped:
     * exr_piz_image: in = file index   out = width height, then the RGBA floats of LoadEXR */
    {
        const std::string directory = (argc > 3) ? argv[3] : "tests/golden/textures";
        struct Spec { const char *name; int width, height, pixelType; };
        const Spec specs[2] = {
            { "piz_float_45x70.exr", 45, 70, TINYEXR_PIXELTYPE_FLOAT },   /* > 2 blocks of 32 lines, odd sizes, 16-bit wavelet */
            { "piz_half_33x40.exr", 33, 40, TINYEXR_PIXELTYPE_HALF },     /* few distinct values: 14-bit wavelet */
        };
        for (int f = 0; f < 2; f++) {
            const Spec &spec = specs[f];
            const size_t pixels = (size_t)spec.width * spec.height;
            std::vector<float> planes[4];   /* A B G R: the channel order tinyexr wants */
            for (int c = 0; c < 4; c++) { planes[c].resize(pixels); }
            for (int y = 0; y < spec.height; y++) {
                for (int x = 0; x < spec.width; x++) {
                    const size_t i = (size_t)y * spec.width + x;
                    if (f == 0) {
                        const float sun = (x > 30 && x < 36 && y > 10 && y < 15) ? 4000.f : 0.f;
                        planes[3][i] = 0.2f + 0.01f * x + 0.3f * uniform01() + sun;        /* R */
                        planes[2][i] = 0.1f + 0.005f * y + 0.1f * uniform01() + sun;      /* G */
                        planes[1][i] = (y < 35) ? 0.7f + 0.2f * uniform01() : 0.f;          /* B: exact zeros */
                        planes[0][i] = 1.f;                                                   /* A */
                    } else {
                        planes[3][i] = 0.25f * (float)((x / 4 + y / 4) % 4);
                        planes[2][i] = 0.5f * (float)((x / 8) % 2);
                        planes[1][i] = (float)(y % 3);
                        planes[0][i] = 1.f;
                    }
                }
            }
            EXRHeader header;
            InitEXRHeader(&header);
            EXRImage image;
            InitEXRImage(&image);
            image.num_channels = 4;
            unsigned char *pointers[4] = {
                (unsigned char *)planes[0].data(), (unsigned char *)planes[1].data(),
                (unsigned char *)planes[2].data(), (unsigned char *)planes[3].data() };
            image.images = pointers;
            image.width = spec.width;
            image.height = spec.height;
            header.num_channels = 4;
            EXRChannelInfo channelInfo[4];
            memset(channelInfo, 0, sizeof channelInfo);
            const char *names[4] = { "A", "B", "G", "R" };
            int inputTypes[4], outputTypes[4];
            for (int c = 0; c < 4; c++) {
                strncpy(channelInfo[c].name, names[c], 255);
                inputTypes[c] = TINYEXR_PIXELTYPE_FLOAT;
                outputTypes[c] = spec.pixelType;
            }
            header.channels = channelInfo;
            header.pixel_types = inputTypes;
            header.requested_pixel_types = outputTypes;
            header.compression_type = TINYEXR_COMPRESSIONTYPE_PIZ;
            const std::string path = directory + "/" + spec.name;
            const char *message = nullptr;
            if (SaveEXRImageToFile(&image, &header, path.c_str(), &message) != TINYEXR_SUCCESS) {
                fprintf(stderr, "cannot write %s: %s\n", path.c_str(), message ? message : "?");
                continue;
            }
            float *loaded = nullptr;
            int width = 0, height = 0;
            if (LoadEXR(&loaded, &width, &height, path.c_str(), &message) != TINYEXR_SUCCESS) {
                fprintf(stderr, "cannot read back %s: %s\n", path.c_str(), message ? message : "?");
                continue;
            }
            std::vector<float> out = { (float)width, (float)height };
            out.insert(out.end(), loaded, loaded + (size_t)4 * width * height);
            free(loaded);
            emit("exr_piz_image", { (float)f }, out);
        }
    }

    /* ---- Image::write -> stbi_write_bmp (src/image.cpp:156-161) ------------------------- */
    /* bmp_bytes: in = width height, then 3*width*height RGB bytes   out = the file stb wrote, byte by byte */
    {
        const int sizes[3][2] = { { 5, 3 }, { 4, 2 }, { 7, 1 } };   /* row padding 1, 0, 3 */
        for (int k = 0; k < 3; k++) {
            const int width = sizes[k][0], height = sizes[k][1];
            std::vector<unsigned char> pixels((size_t)3 * width * height);
            std::vector<float> in = { (float)width, (float)height };
            for (size_t i = 0; i < pixels.size(); i++) { pixels[i] = (unsigned char)(uniform01() * 255.99f); in.push_back((float)pixels[i]); }
            const char *path = "/tmp/pathed_refdump.bmp";
            stbi_write_bmp(path, width, height, 3, pixels.data());
            std::ifstream file(path, std::ios::binary);
            std::vector<float> out;
            char byte;
            while (file.get(byte)) { out.push_back((float)(unsigned char)byte); }
            emit("bmp_bytes", in, out);
        }
    }

    /* ---- JPEG texture files through the reference's stb_image (texture_image records, index 100+) ---- */
    {
        const std::string directory = (argc > 3) ? argv[3] : "tests/golden/textures";
        const char *files[] = { "jpeg_444_37x29.jpg", "jpeg_420_37x29.jpg", "jpeg_422_37x29.jpg", "jpeg_progressive_420_37x29.jpg",
                                "jpeg_progressive_444_37x29.jpg", "jpeg_noise_420_26x21.jpg", "jpeg_grey_37x29.jpg",
                                "jpeg_420_1x9.jpg", "jpeg_restart_420_100x70.jpg",
                                "adam7_rgb8_13x11.png", "adam7_grey4_10x9.png", "adam7_rgba16_3x2.png" };
        for (int f = 0; f < 12; f++) {
            const std::string path = directory + "/" + files[f];
            std::ifstream probe(path);
            if (!probe.good()) { fprintf(stderr, "texture fixture missing: %s\n", path.c_str()); continue; }
            Texture texture(path);
            texture.load();
            std::vector<float> image = { (float)texture.m_width, (float)texture.m_height };
            for (int i = 0; i < 3 * texture.m_width * texture.m_height; i++) { image.push_back((float)texture.m_data[i]); }
            emit("texture_image", { (float)(100 + f) }, image);
        }
    }

    /* ---- the text layer under the OBJ / MTL readers: src/string_util.cpp, src/mtl_parser.cpp ----
     * first the reference's own known answers (test/string_util_test.cpp:9-37), then harder lines */
    {
        const char *trims[] = { "token", "token  ", "t o  ken", "  token", "  token  ", "  t o  ken", " ", "     ", "",
                                "\ttoken", " \t token\t", "\t\t" };
        for (const char *text : trims) {
            fprintf(g_out, "{\"fn\": \"ltrim\", \"text\": %s, \"result\": %s}\n", quoted(text).c_str(), quoted(lTrim(text)).c_str());
        }
        const char *lines[] = { "a b c", " a b c", "  a b c", "a b c ", "  a  b      c    ", "", " ", " \t ",
                                "newmtl\tname", "Kd 0.1\t0.2  0.3", "f 1/2/3 4//5 6", "\tusemtl  light ", "v  -1.5e-3\t2 3 # no comment syntax",
                                "single", "trailing\t", "a\t\tb" };
        for (const char *line : lines) {
            std::queue<std::string> tokens = tokenize(line);
            fprintf(g_out, "{\"fn\": \"tokenize\", \"text\": %s, \"tokens\": [", quoted(line).c_str());
            bool first = true;
            while (!tokens.empty()) { fprintf(g_out, "%s%s", first ? "" : ", ", quoted(tokens.front()).c_str()); tokens.pop(); first = false; }
            fprintf(g_out, "]}\n");
        }
        /* MtlParser over the two libraries the reference ships and two fixture files; paths relative to the repository root */
        const std::string root = (argc > 4) ? argv[4] : ".";
        const char *files[] = { "scenes/CornellBox-Original.mtl", "scenes/cornell-glossy/CornellBox-Glossy.mtl",
                                "tests/golden/mtl/edge_cases.mtl", "tests/golden/mtl/values_before_newmtl.mtl", "tests/golden/mtl/absent.mtl" };
        for (const char *file : files) {
            MtlParser parser(root + "/" + file);
            parser.parse();
            fprintf(g_out, "{\"fn\": \"mtl_parse\", \"file\": %s, \"materials\": [", quoted(file).c_str());
            bool first = true;
            for (const auto &item : parser.m_mtlLookup) {   /* std::map: name order */
                const Color kd = item.second.diffuse, ke = item.second.emit;
                fprintf(g_out, "%s{\"name\": %s, \"Kd\": [%.9g, %.9g, %.9g], \"Ke\": [%.9g, %.9g, %.9g]}", first ? "" : ", ",
                        quoted(item.first).c_str(), kd.r(), kd.g(), kd.b(), ke.r(), ke.g(), ke.b());
                first = false;
            }
            fprintf(g_out, "], \"baked\": %d}\n", (int)parser.materialLookup().size());
        }
    }

    if (g_out != stdout) { fclose(g_out); }
    return 0;
}
