// Host-side BVH builder: binary binned-SAH tree, collapsed to 4-wide nodes (128 B, trace.h).
// Stands in for rtcCommitScene (reference src/scene.cpp:39).  Runs once per scene on the
// host; an on-GPU LBVH builder is the "next" row f3 of SURVEY.md §8.
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace pathed {

struct FlatBvh {
    std::vector<float> nodes;     // 32 floats per 4-wide node
    std::vector<float> leafTris;  // 12 floats per triangle, leaf order
    int maxDepth = 0;             // 4-wide node levels on the longest root-to-leaf chain
    int nodeCount = 0;
};
static const int kNodeFloats = 32;
static const int kEmptyChildRef = (int)0x80000000u;

namespace bvh_detail {

struct Prim {
    float bmin[3], bmax[3], centroid[3];
    uint32_t index;   // triangle id, or triangleCount + sphere index
    bool sphere;      // a sphere gets a leaf of its own (reference src/sphere.cpp:16-48 gives each one to Embree as a geometry)
};

struct Box {
    float lo[3], hi[3];
    void reset()
    {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::numeric_limits<float>::infinity();
            hi[a] = -std::numeric_limits<float>::infinity();
        }
    }
    void grow(const float *bmin, const float *bmax)
    {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], bmin[a]);
            hi[a] = std::max(hi[a], bmax[a]);
        }
    }
    float halfArea() const
    {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.f)) { return 0.f; }
        return dx * dy + dy * dz + dz * dx;
    }
};

struct TempNode {
    Box box;
    int left = -1, right = -1;  // temp node indices
    uint32_t first = 0, count = 0;  // leaf range
};

#ifndef PATHED_SAH_BINS
#define PATHED_SAH_BINS 16
#endif
static const int kBins = PATHED_SAH_BINS;
// leaf size: <= 7 (3-bit count in the traversal's leaf references).  [r5] 3 for a scene's tree (was 4): the wavefront's
// traversal kernel runs the teapot 1.6 % and the 5.2 M-triangle mesh 1.2 - 1.9 % faster on leaves of at most 3 (2: +2.5 % / the
// same, 1.6x the nodes; profiles/r5_ab_leaf_size.log); the hybrid kernel's tree part, walked in short bursts by few lanes,
// keeps 4 (buildBvh's maxLeaf).  PATHED_MAX_LEAF overrides for tuning in the experiments build (the product library reads no
// PATHED_* variable)
// ... and 2 for trees of fewer than 256 Ki primitives (the teapot: another 0.9 %; 1.6x the nodes of a tree that is small anyway)
static const uint32_t kSmallTreePrims = 262144u;
inline uint32_t maxLeafSize()
{
    static uint32_t value = 0;
    if (value == 0) {
        value = 3;
#if defined(PATHED_EXPERIMENTS) && PATHED_EXPERIMENTS
        if (const char *text = getenv("PATHED_MAX_LEAF")) {
            const int parsed = atoi(text);
            if (parsed >= 1 && parsed <= 7) { value = (uint32_t)parsed; }
        }
#endif
    }
    return value;
}

class Builder {
public:
    Builder(std::vector<Prim> &prims) : m_prims(prims) {}

    std::vector<TempNode> nodes;

    // Subtrees of at most deferThreshold primitives (0 = never) are not built but recorded, so
    // that the caller can build them on other threads: they work on disjoint ranges of the
    // primitive array and make the same decisions as a sequential build, so the tree is the same.
    struct Deferred { int nodeIndex; uint32_t begin, end; int depth; };
    std::vector<Deferred> deferred;
    uint32_t deferThreshold = 0;
    uint32_t maxLeaf = maxLeafSize();   // triangles per leaf, at most

    int build(uint32_t begin, uint32_t end, int depth, int *maxDepth)
    {
        TempNode node;
        node.box.reset();
        Box centroidBox;
        centroidBox.reset();
        for (uint32_t i = begin; i < end; i++) {
            node.box.grow(m_prims[i].bmin, m_prims[i].bmax);
            centroidBox.grow(m_prims[i].centroid, m_prims[i].centroid);
        }
        const int index = (int)nodes.size();
        nodes.push_back(node);

        const uint32_t count = end - begin;
        bool spheresInRange = false;
        if (count <= maxLeaf) {
            for (uint32_t i = begin; i < end; i++) { spheresInRange = spheresInRange || m_prims[i].sphere; }
        }
        // leaves are homogeneous: up to maxLeaf triangles, or exactly one sphere
        if (count <= maxLeaf && (!spheresInRange || count == 1)) {
            nodes[(size_t)index].first = begin;
            nodes[(size_t)index].count = count;
            *maxDepth = std::max(*maxDepth, depth);
            return index;
        }
        if (deferThreshold != 0 && count <= deferThreshold) {
            deferred.push_back({ index, begin, end, depth });
            return index;
        }

        // binned SAH over the three axes
        int bestAxis = -1, bestSplit = -1;
        float bestCost = std::numeric_limits<float>::infinity();
        for (int axis = 0; axis < 3; axis++) {
            const float lo = centroidBox.lo[axis], hi = centroidBox.hi[axis];
            if (!(hi > lo)) { continue; }
            const float scale = (float)kBins / (hi - lo);
            Box binBox[kBins];
            uint32_t binCount[kBins];
            for (int b = 0; b < kBins; b++) { binBox[b].reset(); binCount[b] = 0; }
            for (uint32_t i = begin; i < end; i++) {
                int b = (int)((m_prims[i].centroid[axis] - lo) * scale);
                b = std::min(std::max(b, 0), kBins - 1);
                binBox[b].grow(m_prims[i].bmin, m_prims[i].bmax);
                binCount[b]++;
            }
            float rightArea[kBins];
            uint32_t rightCount[kBins];
            Box accumulated;
            accumulated.reset();
            uint32_t running = 0;
            for (int b = kBins - 1; b > 0; b--) {
                accumulated.grow(binBox[b].lo, binBox[b].hi);
                running += binCount[b];
                rightArea[b] = accumulated.halfArea();
                rightCount[b] = running;
            }
            accumulated.reset();
            running = 0;
            for (int b = 0; b < kBins - 1; b++) {
                accumulated.grow(binBox[b].lo, binBox[b].hi);
                running += binCount[b];
                if (running == 0 || rightCount[b + 1] == 0) { continue; }
                const float cost = accumulated.halfArea() * (float)running + rightArea[b + 1] * (float)rightCount[b + 1];
                if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestSplit = b; }
            }
        }

        uint32_t mid;
        // depth guard: past 48 levels fall back to object-median so the traversal stack bound holds
        if (bestAxis >= 0 && depth < 48) {
            const float lo = centroidBox.lo[bestAxis], hi = centroidBox.hi[bestAxis];
            const float scale = (float)kBins / (hi - lo);
            auto middle = std::partition(
                m_prims.begin() + begin, m_prims.begin() + end,
                [=](const Prim &p) {
                    int b = (int)((p.centroid[bestAxis] - lo) * scale);
                    b = std::min(std::max(b, 0), kBins - 1);
                    return b <= bestSplit;
                });
            mid = (uint32_t)(middle - m_prims.begin());
        } else {
            mid = begin;  // force the median fallback below
        }
        if (mid == begin || mid == end) {
            int axis = 0;
            for (int a = 1; a < 3; a++) {
                if (centroidBox.hi[a] - centroidBox.lo[a] > centroidBox.hi[axis] - centroidBox.lo[axis]) { axis = a; }
            }
            mid = begin + count / 2;
            std::nth_element(
                m_prims.begin() + begin, m_prims.begin() + mid, m_prims.begin() + end,
                [axis](const Prim &a, const Prim &b) {
                    if (a.centroid[axis] != b.centroid[axis]) { return a.centroid[axis] < b.centroid[axis]; }
                    return a.index < b.index;
                });
        }

        const int left = build(begin, mid, depth + 1, maxDepth);
        const int right = build(mid, end, depth + 1, maxDepth);
        nodes[(size_t)index].left = left;
        nodes[(size_t)index].right = right;
        return index;
    }

private:
    std::vector<Prim> &m_prims;
};

inline void padBox(const Box &in, float *lo, float *hi)
{
    // a hit computed in fp32 on a face lying in a box plane must survive the slab test
    for (int a = 0; a < 3; a++) {
        const float pad = 1e-5f * std::max(1.f, std::max(std::fabs(in.lo[a]), std::fabs(in.hi[a])));
        lo[a] = in.lo[a] - pad;
        hi[a] = in.hi[a] + pad;
    }
}

inline void putInt(float *slot, int value) { std::memcpy(slot, &value, 4); }

}  // namespace bvh_detail

// positions: 3 floats per vertex; indices: 3 per triangle; spheres: (centre.xyz, radius) each, may be null.
// Sphere s becomes the leaf reference -(((s + 1) << 3) | 0) - 1: count 0 marks it, trace.h tests it.
// `buildThreads` > 0 fixes the number of host threads (PathedSceneOptions.build_threads); the tree does not depend on it
inline FlatBvh buildBvh(const float *positions, const uint32_t *indices, uint32_t triangleCount,
                        const float *spheres = nullptr, uint32_t sphereCount = 0, int buildThreads = 0, uint32_t maxLeaf = 0)
{
    using namespace bvh_detail;
    FlatBvh out;
    const uint32_t primCount = triangleCount + sphereCount;
    if (primCount == 0) { return out; }

    std::vector<Prim> prims(primCount);
    for (uint32_t s = 0; s < sphereCount; s++) {
        Prim &p = prims[triangleCount + s];
        p.index = triangleCount + s;
        p.sphere = true;
        const float radius = std::fabs(spheres[4 * s + 3]);
        for (int a = 0; a < 3; a++) {
            // a hit point computed in fp32 may sit an ulp outside centre +- radius: pad by a relative 1e-5 as well
            const float centre = spheres[4 * s + a];
            const float reach = radius * 1.00001f + 1e-5f * std::fabs(centre);
            p.bmin[a] = centre - reach;
            p.bmax[a] = centre + reach;
            p.centroid[a] = centre;
        }
    }
    for (uint32_t i = 0; i < triangleCount; i++) {
        Prim &p = prims[i];
        p.index = i;
        p.sphere = false;
        for (int a = 0; a < 3; a++) {
            const float c0 = positions[3 * indices[3 * i + 0] + a];
            const float c1 = positions[3 * indices[3 * i + 1] + a];
            const float c2 = positions[3 * indices[3 * i + 2] + a];
            p.bmin[a] = std::min(c0, std::min(c1, c2));
            p.bmax[a] = std::max(c0, std::max(c1, c2));
            p.centroid[a] = 0.5f * (p.bmin[a] + p.bmax[a]);
        }
    }

    Builder builder(prims);
    if (maxLeaf != 0) { builder.maxLeaf = maxLeaf; }
    else if (primCount < kSmallTreePrims && builder.maxLeaf > 2u) { builder.maxLeaf = 2u; }   // (see maxLeafSize)
    builder.nodes.reserve((size_t)primCount);
    int maxDepth = 0;
    // large meshes: the top of the tree here, its subtrees on the host's other cores
    unsigned int hostThreads = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    if (buildThreads >= 1) { hostThreads = (unsigned int)buildThreads; }
    if (primCount >= 200000 && hostThreads > 1) { builder.deferThreshold = primCount / 256; }
    const int root = builder.build(0, primCount, 0, &maxDepth);
    if (!builder.deferred.empty()) {
        const size_t jobs = builder.deferred.size();
        std::vector<std::vector<TempNode>> built(jobs);
        std::atomic<size_t> next(0);
        auto worker = [&]() {
            for (size_t job = next.fetch_add(1); job < jobs; job = next.fetch_add(1)) {
                const Builder::Deferred &item = builder.deferred[job];
                Builder sub(prims);
                sub.maxLeaf = builder.maxLeaf;
                sub.nodes.reserve((size_t)(item.end - item.begin));
                int subDepth = 0;
                sub.build(item.begin, item.end, item.depth, &subDepth);
                built[job].swap(sub.nodes);
            }
        };
        std::vector<std::thread> pool;
        for (unsigned int t = 1; t < hostThreads; t++) { pool.emplace_back(worker); }
        worker();
        for (std::thread &thread : pool) { thread.join(); }
        // splice: a subtree's root replaces its placeholder, the rest is appended
        for (size_t job = 0; job < jobs; job++) {
            const int offset = (int)builder.nodes.size() - 1;   // sub node k (k >= 1) lands at offset + k
            const std::vector<TempNode> &sub = built[job];
            auto moved = [&](TempNode node) {
                if (node.left >= 0) { node.left += offset; node.right += offset; }
                return node;
            };
            builder.nodes[(size_t)builder.deferred[job].nodeIndex] = moved(sub[0]);
            for (size_t k = 1; k < sub.size(); k++) { builder.nodes.push_back(moved(sub[k])); }
        }
    }

    // leaf-ordered triangles: (v0, prim) (e1, 0) (e2, 0); spheres take no slot, so a triangle's place in leaf order
    // is its place among the triangles of the sorted primitive array
    out.leafTris.resize((size_t)12 * triangleCount);
    std::vector<uint32_t> trianglePlace(primCount, 0u);
    uint32_t placed = 0;
    for (uint32_t i = 0; i < primCount; i++) {
        trianglePlace[i] = placed;
        if (prims[i].sphere) { continue; }
        const uint32_t prim = prims[i].index;
        const float *v0 = positions + 3 * indices[3 * prim + 0];
        const float *v1 = positions + 3 * indices[3 * prim + 1];
        const float *v2 = positions + 3 * indices[3 * prim + 2];
        float *tri = out.leafTris.data() + (size_t)12 * placed;
        placed++;
        for (int a = 0; a < 3; a++) {
            tri[a] = v0[a];
            tri[4 + a] = v1[a] - v0[a];
            tri[8 + a] = v2[a] - v0[a];
        }
        putInt(tri + 3, (int)prim);
        tri[7] = 0.f;
        tri[11] = 0.f;
    }

    // Collapse to 4-wide nodes: a node adopts its binary children, then repeatedly replaces the
    // inner child with the largest box by that child's two children until it has four (or only
    // leaves are left).  Wide nodes get consecutive ids in DFS order.
    const std::vector<TempNode> &temp = builder.nodes;
    struct WideNode {
        int child[4];   // temp node indices
        int id[4];      // wide node id of an inner child
        int count;
    };
    std::vector<WideNode> wide;
    wide.reserve(temp.size() / 3 + 1);
    int wideDepth = 0;
    if (temp[(size_t)root].left < 0) {
        WideNode only;
        only.child[0] = root;
        only.id[0] = -1;
        only.count = 1;
        wide.push_back(only);
        wideDepth = 1;
    } else {
        struct Pending { int tempNode, parent, slot, depth; };
        std::vector<Pending> work;
        work.push_back({ root, -1, 0, 1 });
        while (!work.empty()) {
            const Pending item = work.back();
            work.pop_back();
            WideNode node;
            node.count = 0;
            node.child[node.count++] = temp[(size_t)item.tempNode].left;
            node.child[node.count++] = temp[(size_t)item.tempNode].right;
            while (node.count < 4) {
                int pick = -1;
                float pickArea = -1.f;
                for (int k = 0; k < node.count; k++) {
                    const TempNode &candidate = temp[(size_t)node.child[k]];
                    if (candidate.left < 0) { continue; }
                    const float area = candidate.box.halfArea();
                    if (area > pickArea) { pickArea = area; pick = k; }
                }
                if (pick < 0) { break; }
                const int opened = node.child[pick];
                node.child[pick] = temp[(size_t)opened].left;
                node.child[node.count++] = temp[(size_t)opened].right;
            }
            for (int k = 0; k < 4; k++) { node.id[k] = -1; }
            const int id = (int)wide.size();
            wide.push_back(node);
            if (item.parent >= 0) { wide[(size_t)item.parent].id[item.slot] = id; }
            wideDepth = std::max(wideDepth, item.depth);
            // push in reverse so that child 0's subtree follows its parent in memory
            for (int k = node.count - 1; k >= 0; k--) {
                if (temp[(size_t)node.child[k]].left >= 0) { work.push_back({ node.child[k], id, k, item.depth + 1 }); }
            }
        }
    }

    out.nodes.assign((size_t)kNodeFloats * wide.size(), 0.f);
    for (size_t n = 0; n < wide.size(); n++) {
        float *node = out.nodes.data() + (size_t)kNodeFloats * n;
        for (int k = 0; k < 4; k++) {
            float lo[3] = { 0.f, 0.f, 0.f }, hi[3] = { 0.f, 0.f, 0.f };
            int ref = kEmptyChildRef;
            if (k < wide[n].count) {
                const TempNode &child = temp[(size_t)wide[n].child[k]];
                padBox(child.box, lo, hi);
                // leaf: -((first << 3) | count) - 1 (trace.h encodeLeaf); a sphere: count 0, first = sphere + 1; inner: the wide node id
                if (child.left >= 0) { ref = wide[n].id[k]; }
                else if (prims[child.first].sphere) { ref = -(int)(((prims[child.first].index - triangleCount + 1u) << 3) | 0u) - 1; }
                else { ref = -(int)((trianglePlace[child.first] << 3) | child.count) - 1; }
            }
            for (int a = 0; a < 3; a++) {
                node[4 * a + k] = lo[a];
                node[12 + 4 * a + k] = hi[a];
            }
            putInt(node + 24 + k, ref);
        }
    }
    out.nodeCount = (int)wide.size();
    out.maxDepth = wideDepth;
    return out;
}

}  // namespace pathed
