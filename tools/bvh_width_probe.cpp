// What a wider tree would buy the traversal kernel, in COUNTS (CPU, no GPU): the product's host builder makes the 4-wide
// tree (bvh_build.h); this tool collapses it further to 8-wide nodes with the builder's own rule (open the inner child
// with the largest box while the node has room) and walks both trees with the same rays and the kernel's policy --
// closest hit, children hit visited leaves first, then near to far -- counting per ray: inner-node visits (= dependent
// node fetches, the kernel's "steps"), child boxes tested, leaf visits, triangles tested, stack pushes.
//   usage: bvh_width_probe <asset root> <scene.json> [rays per axis = 160]
// Rays: the scene's camera through a grid of pixels, and from every hit one bounce in a cosine-ish direction plus one ray
// towards +z (the stand-in scenes' sky): the mix of coherent and incoherent rays a render traces.
#include "../pathed_amd/csrc/bvh_build.h"
#include "../pathed_amd/host/scene_loader.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

struct Ray { float o[3], d[3]; };
struct Counts { unsigned long long steps = 0, boxes = 0, leaves = 0, tris = 0, pushes = 0, rays = 0, maxStack = 0; };

struct WideNode { std::vector<float> lo, hi; std::vector<int> ref; };   // per child: box (3 + 3), ref as in the flat tree

static std::vector<WideNode> fromFlat(const pathed::FlatBvh &bvh)
{
    std::vector<WideNode> nodes((size_t)bvh.nodeCount);
    for (int n = 0; n < bvh.nodeCount; n++) {
        const float *node = bvh.nodes.data() + (size_t)pathed::kNodeFloats * n;
        for (int c = 0; c < 4; c++) {
            int ref;
            std::memcpy(&ref, node + 24 + c, 4);
            if (ref == pathed::kEmptyChildRef) { continue; }
            for (int a = 0; a < 3; a++) { nodes[(size_t)n].lo.push_back(node[4 * a + c]); nodes[(size_t)n].hi.push_back(node[12 + 4 * a + c]); }
            nodes[(size_t)n].ref.push_back(ref);
        }
    }
    return nodes;
}

// collapse to `width` children per node: open the inner child with the largest half-area while everything fits
static std::vector<WideNode> widen(const std::vector<WideNode> &four, int width)
{
    std::vector<WideNode> out;
    std::vector<int> newIndex(four.size(), -1);
    std::vector<int> order = { 0 };
    newIndex[0] = 0;
    out.emplace_back();
    for (size_t k = 0; k < order.size(); k++) {
        WideNode node = four[(size_t)order[k]];
        while (true) {
            int best = -1;
            float bestArea = -1.f;
            for (size_t c = 0; c < node.ref.size(); c++) {
                if (node.ref[c] < 0) { continue; }
                const WideNode &child = four[(size_t)node.ref[c]];
                if ((int)(node.ref.size() - 1 + child.ref.size()) > width) { continue; }
                const float dx = node.hi[3 * c] - node.lo[3 * c], dy = node.hi[3 * c + 1] - node.lo[3 * c + 1], dz = node.hi[3 * c + 2] - node.lo[3 * c + 2];
                const float area = dx * dy + dy * dz + dz * dx;
                if (area > bestArea) { bestArea = area; best = (int)c; }
            }
            if (best < 0) { break; }
            const WideNode &child = four[(size_t)node.ref[(size_t)best]];
            node.lo.erase(node.lo.begin() + 3 * best, node.lo.begin() + 3 * best + 3);
            node.hi.erase(node.hi.begin() + 3 * best, node.hi.begin() + 3 * best + 3);
            node.ref.erase(node.ref.begin() + best);
            node.lo.insert(node.lo.end(), child.lo.begin(), child.lo.end());
            node.hi.insert(node.hi.end(), child.hi.begin(), child.hi.end());
            node.ref.insert(node.ref.end(), child.ref.begin(), child.ref.end());
        }
        for (int &ref : node.ref) {
            if (ref >= 0) {
                if (newIndex[(size_t)ref] < 0) { newIndex[(size_t)ref] = (int)order.size(); order.push_back(ref); out.emplace_back(); }
                ref = newIndex[(size_t)ref];
            }
        }
        out[k] = node;
    }
    return out;
}

static bool hitTriangle(const float *tri, const Ray &r, float tnear, float *t)
{
    const float *v0 = tri, *e1 = tri + 4, *e2 = tri + 8;
    const float px = r.d[1] * e2[2] - r.d[2] * e2[1], py = r.d[2] * e2[0] - r.d[0] * e2[2], pz = r.d[0] * e2[1] - r.d[1] * e2[0];
    const float det = e1[0] * px + e1[1] * py + e1[2] * pz;
    if (det == 0.f) { return false; }
    const float tx = r.o[0] - v0[0], ty = r.o[1] - v0[1], tz = r.o[2] - v0[2];
    const float u = (tx * px + ty * py + tz * pz) / det;
    const float qx = ty * e1[2] - tz * e1[1], qy = tz * e1[0] - tx * e1[2], qz = tx * e1[1] - ty * e1[0];
    const float v = (r.d[0] * qx + r.d[1] * qy + r.d[2] * qz) / det;
    if (u < 0.f || v < 0.f || u + v > 1.f) { return false; }
    *t = (e2[0] * qx + e2[1] * qy + e2[2] * qz) / det;
    return *t > tnear;
}

static float trace(const std::vector<WideNode> &nodes, const pathed::FlatBvh &bvh, const Ray &ray, Counts &counts)
{
    float best = 1e5f;
    const float inv[3] = { 1.f / ray.d[0], 1.f / ray.d[1], 1.f / ray.d[2] };
    std::vector<int> stack;
    int current = 0;
    counts.rays++;
    while (true) {
        if (current >= 0) {
            const WideNode &node = nodes[(size_t)current];
            counts.steps++;
            struct Hit { float t; int ref; };
            Hit hits[8];
            int n = 0;
            for (size_t c = 0; c < node.ref.size(); c++) {
                counts.boxes++;
                float tmin = 1e-3f, tmax = best;
                for (int a = 0; a < 3; a++) {
                    const float t0 = (node.lo[3 * c + a] - ray.o[a]) * inv[a], t1 = (node.hi[3 * c + a] - ray.o[a]) * inv[a];
                    tmin = std::max(tmin, std::min(t0, t1));
                    tmax = std::min(tmax, std::max(t0, t1));
                }
                if (tmin <= tmax * 1.0000004f) { hits[n++] = { tmin, node.ref[c] }; }
            }
            // leaves first, then near to far (the kernel's order)
            std::sort(hits, hits + n, [](const Hit &a, const Hit &b) { return (a.ref >= 0) != (b.ref >= 0) ? a.ref < 0 : a.t < b.t; });
            for (int k = n - 1; k >= 1; k--) { stack.push_back(hits[k].ref); counts.pushes++; }
            counts.maxStack = std::max<unsigned long long>(counts.maxStack, stack.size());
            if (n > 0) { current = hits[0].ref; continue; }
        } else {
            const int leaf = -current - 1, first = leaf >> 3, count = leaf & 7;
            counts.leaves++;
            for (int k = 0; k < count; k++) {
                counts.tris++;
                float t;
                if (hitTriangle(bvh.leafTris.data() + (size_t)12 * (first + k), ray, 1e-3f, &t) && t < best) { best = t; }
            }
        }
        if (stack.empty()) { break; }
        current = stack.back();
        stack.pop_back();
    }
    return best;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s <asset root> <scene.json> [grid]\n", argv[0]); return 2; }
    const int grid = argc > 3 ? atoi(argv[3]) : 160;
    pathed::FlatScene flat = pathed::loadScene(argv[2], 16 * grid, 9 * grid, argv[1]);
    const PathedSceneDesc desc = flat.desc();
    const pathed::FlatBvh bvh = pathed::buildBvh(desc.positions, desc.indices, desc.n_triangles, nullptr, 0, 0);
    const std::vector<WideNode> four = fromFlat(bvh);
    const std::vector<WideNode> eight = widen(four, 8);
    printf("%u triangles; 4-wide: %zu nodes; 8-wide: %zu nodes (%.2f children per node)\n", desc.n_triangles, four.size(), eight.size(),
           [&]() { double c = 0; for (const WideNode &n : eight) { c += (double)n.ref.size(); } return c / (double)eight.size(); }());
    // camera rays (reference src/camera.cpp:32-47 without jitter), then one bounce + one sky ray from every hit
    const PathedCamera &cam = desc.camera;
    auto norm = [](float *v) { const float l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); for (int a = 0; a < 3; a++) { v[a] /= l; } };
    float dir[3] = { cam.origin[0] - cam.target[0], cam.origin[1] - cam.target[1], cam.origin[2] - cam.target[2] };
    norm(dir);
    float up[3] = { cam.up[0], cam.up[1], cam.up[2] };
    norm(up);
    float x[3] = { up[1] * dir[2] - up[2] * dir[1], up[2] * dir[0] - up[0] * dir[2], up[0] * dir[1] - up[1] * dir[0] };
    norm(x);
    const float y[3] = { dir[1] * x[2] - dir[2] * x[1], dir[2] * x[0] - dir[0] * x[2], dir[0] * x[1] - dir[1] * x[0] };
    const float sign = cam.flip_handedness ? -1.f : 1.f;
    const float h = 2.f * std::tan(cam.vertical_fov / 2.f), w = h * 16.f / 9.f;
    std::vector<Ray> primary, secondary;
    unsigned int seed = 12345u;
    auto uniform = [&]() { seed = seed * 1664525u + 1013904223u; return (float)(seed >> 8) * (1.f / 16777216.f); };
    for (int row = 0; row < 9 * grid / 16; row++) {
        for (int col = 0; col < grid; col++) {
            const float cx = w * ((float)col + 0.5f) / (float)grid - w / 2.f, cy = h * ((float)row + 0.5f) / (float)(9 * grid / 16) - h / 2.f;
            Ray r;
            for (int a = 0; a < 3; a++) { r.o[a] = cam.origin[a]; r.d[a] = sign * x[a] * cx + y[a] * cy - dir[a]; }
            norm(r.d);
            primary.push_back(r);
        }
    }
    Counts c4p, c8p, c4s, c8s;
    for (const Ray &r : primary) {
        const float t = trace(four, bvh, r, c4p);
        trace(eight, bvh, r, c8p);
        if (t < 1e5f) {
            Ray b;
            for (int a = 0; a < 3; a++) { b.o[a] = r.o[a] + r.d[a] * (t - 1e-2f); }
            b.d[0] = uniform() * 2.f - 1.f; b.d[1] = uniform() * 2.f - 1.f; b.d[2] = uniform() * 2.f - 1.f;
            norm(b.d);
            secondary.push_back(b);
            Ray sky = b;
            sky.d[0] = 0.3f * (uniform() - 0.5f); sky.d[1] = 0.3f * (uniform() - 0.5f); sky.d[2] = 1.f;
            norm(sky.d);
            secondary.push_back(sky);
        }
    }
    for (const Ray &r : secondary) { trace(four, bvh, r, c4s); trace(eight, bvh, r, c8s); }
    auto report = [](const char *name, const Counts &c) {
        const double n = (double)c.rays;
        printf("%-28s rays %8llu  node visits %6.2f  boxes %6.2f  leaf visits %5.2f  triangles %5.2f  pushes %5.2f  deepest stack %llu\n",
               name, c.rays, c.steps / n, c.boxes / n, c.leaves / n, c.tris / n, c.pushes / n, c.maxStack);
    };
    report("4-wide, camera rays", c4p);
    report("8-wide, camera rays", c8p);
    report("4-wide, bounce + sky rays", c4s);
    report("8-wide, bounce + sky rays", c8s);
    return 0;
}
