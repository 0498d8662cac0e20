// Host-side BVH2 builder (binned SAH) + flattening to the 64-B two-child node layout.
// Stands in for rtcCommitScene (reference src/scene.cpp:39).  Runs once per scene on the
// host; an on-GPU LBVH builder is the "next" row f3 of SURVEY.md §8.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

namespace pathed {

struct FlatBvh {
    std::vector<float> nodes;     // 16 floats per inner node
    std::vector<float> leafTris;  // 12 floats per triangle, leaf order
    int maxDepth = 0;             // inner-node levels on the longest root-to-leaf chain
    int nodeCount = 0;
};

namespace bvh_detail {

struct Prim {
    float bmin[3], bmax[3], centroid[3];
    uint32_t index;
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

static const int kBins = 16;
// leaf size: <= 7 (3-bit count in the traversal's leaf references); PATHED_MAX_LEAF overrides for tuning
inline uint32_t maxLeafSize()
{
    static uint32_t value = 0;
    if (value == 0) {
        value = 4;
        if (const char *text = getenv("PATHED_MAX_LEAF")) {
            const int parsed = atoi(text);
            if (parsed >= 1 && parsed <= 7) { value = (uint32_t)parsed; }
        }
    }
    return value;
}

class Builder {
public:
    Builder(std::vector<Prim> &prims) : m_prims(prims) {}

    std::vector<TempNode> nodes;

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
        if (count <= maxLeafSize()) {
            nodes[(size_t)index].first = begin;
            nodes[(size_t)index].count = count;
            *maxDepth = std::max(*maxDepth, depth);
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

// positions: 3 floats per vertex; indices: 3 per triangle.
inline FlatBvh buildBvh(const float *positions, const uint32_t *indices, uint32_t triangleCount)
{
    using namespace bvh_detail;
    FlatBvh out;
    if (triangleCount == 0) { return out; }

    std::vector<Prim> prims(triangleCount);
    for (uint32_t i = 0; i < triangleCount; i++) {
        Prim &p = prims[i];
        p.index = i;
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
    builder.nodes.reserve((size_t)triangleCount);
    int maxDepth = 0;
    const int root = builder.build(0, triangleCount, 0, &maxDepth);

    // leaf-ordered triangles: (v0, prim) (e1, 0) (e2, 0)
    out.leafTris.resize((size_t)12 * triangleCount);
    for (uint32_t i = 0; i < triangleCount; i++) {
        const uint32_t prim = prims[i].index;
        const float *v0 = positions + 3 * indices[3 * prim + 0];
        const float *v1 = positions + 3 * indices[3 * prim + 1];
        const float *v2 = positions + 3 * indices[3 * prim + 2];
        float *tri = out.leafTris.data() + (size_t)12 * i;
        for (int a = 0; a < 3; a++) {
            tri[a] = v0[a];
            tri[4 + a] = v1[a] - v0[a];
            tri[8 + a] = v2[a] - v0[a];
        }
        putInt(tri + 3, (int)prim);
        tri[7] = 0.f;
        tri[11] = 0.f;
    }

    // flatten: inner temp nodes get consecutive ids in DFS order
    const std::vector<TempNode> &temp = builder.nodes;
    std::vector<int> innerId(temp.size(), -1);
    int innerCount = 0;
    {
        std::vector<int> work;
        work.push_back(root);
        while (!work.empty()) {
            const int t = work.back();
            work.pop_back();
            if (temp[(size_t)t].left < 0) { continue; }
            innerId[(size_t)t] = innerCount++;
            work.push_back(temp[(size_t)t].right);
            work.push_back(temp[(size_t)t].left);
        }
    }

    auto writeChild = [&](float *slot, int t) {
        const TempNode &child = temp[(size_t)t];
        padBox(child.box, slot, slot + 4);
        if (child.left < 0) {
            putInt(slot + 3, (int)child.first);
            putInt(slot + 7, (int)child.count);
        } else {
            putInt(slot + 3, innerId[(size_t)t]);
            putInt(slot + 7, 0);
        }
    };
    auto writeEmpty = [&](float *slot) {
        for (int a = 0; a < 3; a++) {
            slot[a] = std::numeric_limits<float>::infinity();
            slot[4 + a] = -std::numeric_limits<float>::infinity();
        }
        putInt(slot + 3, 0);
        putInt(slot + 7, -1);
    };

    if (innerCount == 0) {
        // the whole scene is one leaf: a root with one real child
        out.nodes.assign(16, 0.f);
        writeChild(out.nodes.data(), root);
        writeEmpty(out.nodes.data() + 8);
        out.nodeCount = 1;
        out.maxDepth = 1;
        return out;
    }

    out.nodes.assign((size_t)16 * innerCount, 0.f);
    for (size_t t = 0; t < temp.size(); t++) {
        if (innerId[t] < 0) { continue; }
        float *node = out.nodes.data() + (size_t)16 * innerId[t];
        writeChild(node, temp[t].left);
        writeChild(node + 8, temp[t].right);
    }
    out.nodeCount = innerCount;
    out.maxDepth = maxDepth;
    return out;
}

}  // namespace pathed
