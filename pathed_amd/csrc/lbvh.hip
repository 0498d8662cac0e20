// On-GPU BVH build for gfx950 (see lbvh.h), all on one stream:
//   1 k_lbvh_prims      triangle boxes + centroid bounds (wave min/max, one atomic per wave)
//   2 k_lbvh_morton     63-bit Morton code of the box centre (21 bits per axis)
//   3 rocprim radix sort of (code, triangle) pairs
//   4-5 a binary tree over the sorted triangles, with boxes and triangle counts:
//       LBVH  k_lbvh_hierarchy  Karras' binary radix tree over the codes (ties broken by position)
//             k_lbvh_fit_pass   boxes bottom-up, one launch per level (kernel boundaries instead of agent-scope fences)
//       PLOC  k_ploc_nearest / _flags / _merge, once per round: mutual nearest neighbours (by the
//             surface area of their union, within 16 places of the Morton order) merge; compaction
//             by exclusive scans; ~60 rounds for millions of triangles, until 1 024 clusters are left,
//             which a binned-SAH build on the host joins (buildTopSah: the top of the tree is where a bad
//             split costs every ray)
//   6 k_lbvh_wide_*     breadth-first collapse to 4-wide nodes, one level per launch pair; ids are
//                       handed out by an exclusive scan, so the tree is the same every run.  A
//                       frontier entry carries its subtree's first position in leaf order; subtrees
//                       of at most four triangles become leaves and write their (v0, prim) (e1) (e2)
//                       records there.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "lbvh.h"

namespace pathed {

namespace {

constexpr int kLbvhBlock = 256;
constexpr unsigned int kLeafFlag = 0x80000000u;   // child reference: a single sorted triangle
constexpr unsigned int kNoChild = 0xFFFFFFFFu;
constexpr int kLbvhMaxLeaf = 4;                   // as the host builder's default leaf size
constexpr int kEmptyRef = (int)0x80000000u;       // trace.h kEmptyChild
// "count" of a subtree: the triangles below it, and in the top bit whether a SPHERE is below it.  A sphere gets a leaf of
// its own (bvh_build.h; reference src/sphere.cpp:16-48 hands each one to Embree as a geometry): a subtree with the bit set
// never becomes a triangle leaf, and as an unsigned number it is "more than kLbvhMaxLeaf" wherever that is the question.
constexpr unsigned int kHasSphere = 0x80000000u;
__host__ __device__ inline unsigned int mergeCounts(unsigned int a, unsigned int b) { return ((a & ~kHasSphere) + (b & ~kHasSphere)) | ((a | b) & kHasSphere); }

// float <-> unsigned with the same order
__device__ inline unsigned int orderedBits(float v)
{
    const unsigned int u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__host__ __device__ inline float fromOrderedBits(unsigned int u)
{
    const unsigned int bits = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float v;
    memcpy(&v, &bits, 4);
    return v;
}

// Grid-stride over the triangles with a bounded grid, so that the centroid bounds cost a few
// thousand atomics in all: atomics to ONE cache line serialise in L2 at about 88 per microsecond on
// MI355X, and one set per wave made this kernel 5.6 ms for 5.2 M triangles (now ~0.3 ms).
constexpr unsigned int kReduceBlocks = 1024;

__global__ __launch_bounds__(kLbvhBlock) void k_lbvh_prims(
    const float *positions, const uint32_t *indices, uint32_t n,
    float4 *boxLo, float4 *boxHi, unsigned int *centroidBounds)
{
    __shared__ float partial[kLbvhBlock / 64][6];
    const float inf = __builtin_huge_valf();
    float cmin[3] = { inf, inf, inf }, cmax[3] = { -inf, -inf, -inf };
    for (size_t i = (size_t)blockIdx.x * kLbvhBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kLbvhBlock) {
        const uint32_t i0 = indices[3 * i + 0], i1 = indices[3 * i + 1], i2 = indices[3 * i + 2];
        float lo[3], hi[3];
        #pragma unroll
        for (int a = 0; a < 3; a++) {
            const float c0 = positions[3 * (size_t)i0 + a];
            const float c1 = positions[3 * (size_t)i1 + a];
            const float c2 = positions[3 * (size_t)i2 + a];
            lo[a] = fminf(c0, fminf(c1, c2));
            hi[a] = fmaxf(c0, fmaxf(c1, c2));
            const float centre = 0.5f * (lo[a] + hi[a]);   // bvh_build.h: the box centre is the "centroid"
            cmin[a] = fminf(cmin[a], centre);
            cmax[a] = fmaxf(cmax[a], centre);
        }
        boxLo[i] = make_float4(lo[0], lo[1], lo[2], 0.f);
        boxHi[i] = make_float4(hi[0], hi[1], hi[2], 0.f);
    }
    #pragma unroll
    for (int a = 0; a < 3; a++) {
        float low = cmin[a], high = cmax[a];
        #pragma unroll
        for (int offset = 32; offset > 0; offset >>= 1) {
            low = fminf(low, __shfl_xor(low, offset));
            high = fmaxf(high, __shfl_xor(high, offset));
        }
        if ((threadIdx.x & 63) == 0) {
            partial[threadIdx.x >> 6][a] = low;
            partial[threadIdx.x >> 6][3 + a] = high;
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float low = partial[0][threadIdx.x], high = partial[0][3 + threadIdx.x];
        #pragma unroll
        for (int w = 1; w < kLbvhBlock / 64; w++) {
            low = fminf(low, partial[w][threadIdx.x]);
            high = fmaxf(high, partial[w][3 + threadIdx.x]);
        }
        if (low <= high) {
            atomicMin(&centroidBounds[threadIdx.x], orderedBits(low));
            atomicMax(&centroidBounds[3 + threadIdx.x], orderedBits(high));
        }
    }
}

// spheres join the primitives behind the triangles: primitive triangleCount + s, bounds as bvh_build.h gives them
// (centre +- (radius * 1.00001 + 1e-5 |centre|): a hit point computed in fp32 may sit an ulp outside the exact bounds)
__global__ __launch_bounds__(kLbvhBlock) void k_lbvh_sphere_prims(
    const float4 *spheres, uint32_t sphereCount, uint32_t triangleCount, float4 *boxLo, float4 *boxHi, unsigned int *centroidBounds)
{
    const uint32_t s = blockIdx.x * kLbvhBlock + threadIdx.x;
    if (s >= sphereCount) { return; }
    const float4 sphere = spheres[s];
    const float centre[3] = { sphere.x, sphere.y, sphere.z };
    const float radius = fabsf(sphere.w);
    float lo[3], hi[3];
    #pragma unroll
    for (int a = 0; a < 3; a++) {
        const float reach = radius * 1.00001f + 1e-5f * fabsf(centre[a]);
        lo[a] = centre[a] - reach;
        hi[a] = centre[a] + reach;
        const float middle = 0.5f * (lo[a] + hi[a]);
        if (middle == middle) {   // few spheres: one atomic pair each
            atomicMin(&centroidBounds[a], orderedBits(middle));
            atomicMax(&centroidBounds[3 + a], orderedBits(middle));
        }
    }
    boxLo[triangleCount + s] = make_float4(lo[0], lo[1], lo[2], 0.f);
    boxHi[triangleCount + s] = make_float4(hi[0], hi[1], hi[2], 0.f);
}

// triangles (1) and spheres (kHasSphere) of the sorted order, for the prefix sums the Karras hierarchy takes its counts from
__global__ __launch_bounds__(kLbvhBlock) void k_lbvh_kinds(const unsigned int *sorted, uint32_t n, uint32_t triangleCount, unsigned int *isTriangle, unsigned int *isSphere)
{
    const uint32_t i = blockIdx.x * kLbvhBlock + threadIdx.x;
    if (i >= n) { return; }
    const bool sphere = sorted[i] >= triangleCount;
    isTriangle[i] = sphere ? 0u : 1u;
    isSphere[i] = sphere ? 1u : 0u;
}

__device__ inline unsigned long long spread21(unsigned long long v)
{
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x001F00000000FFFFull;
    v = (v | (v << 16)) & 0x001F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

__global__ __launch_bounds__(kLbvhBlock) void k_lbvh_morton(
    const float4 *boxLo, const float4 *boxHi, uint32_t n, const unsigned int *centroidBounds,
    unsigned long long *keys, unsigned int *values)
{
    const uint32_t i = blockIdx.x * kLbvhBlock + threadIdx.x;
    if (i >= n) { return; }
    const float4 lo = boxLo[i], hi = boxHi[i];
    const float centre[3] = { 0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z) };
    unsigned long long code = 0ull;
    #pragma unroll
    for (int a = 0; a < 3; a++) {
        const float low = fromOrderedBits(centroidBounds[a]);
        const float high = fromOrderedBits(centroidBounds[3 + a]);
        const float extent = high - low;
        float q = extent > 0.f ? (centre[a] - low) / extent * 2097152.f : 0.f;
        q = fminf(fmaxf(q, 0.f), 2097151.f);   // NaN -> 0
        code |= spread21((unsigned long long)(unsigned int)q) << (2 - a);
    }
    keys[i] = code;
    values[i] = i;
}

// length of the common prefix of sorted entries i and j; equal codes fall back to the positions
__device__ inline int commonPrefix(const unsigned long long *keys, int n, unsigned long long ki, int i, int j)
{
    if (j < 0 || j >= n) { return -1; }
    const unsigned long long kj = keys[j];
    if (ki == kj) { return 64 + __clz((unsigned int)(i ^ j)); }
    return __clzll((long long)(ki ^ kj));
}

// Karras 2012, "Maximizing parallelism in the construction of BVHs, octrees, and k-d trees", §3-4
__global__ __launch_bounds__(kLbvhBlock) void k_lbvh_hierarchy(
    const unsigned long long *keys, int n, const unsigned int *trianglesBefore, const unsigned int *spheresBefore,
    const unsigned int *sorted, uint32_t triangleCount,
    uint2 *children, unsigned int *count, int *parentOfInternal, int *parentOfLeaf)
{
    const int i = blockIdx.x * kLbvhBlock + threadIdx.x;
    if (i >= n - 1) { return; }
    const unsigned long long ki = keys[i];
    const int direction = commonPrefix(keys, n, ki, i, i + 1) - commonPrefix(keys, n, ki, i, i - 1) >= 0 ? 1 : -1;
    const int floorPrefix = commonPrefix(keys, n, ki, i, i - direction);
    int reach = 2;
    while (commonPrefix(keys, n, ki, i, i + reach * direction) > floorPrefix) { reach *= 2; }
    int length = 0;
    for (int step = reach / 2; step >= 1; step /= 2) {
        if (commonPrefix(keys, n, ki, i, i + (length + step) * direction) > floorPrefix) { length += step; }
    }
    const int j = i + length * direction;
    const int nodePrefix = commonPrefix(keys, n, ki, i, j);
    int split = 0;
    for (int step = (length + 1) / 2; ; step = (step + 1) / 2) {
        if (commonPrefix(keys, n, ki, i, i + (split + step) * direction) > nodePrefix) { split += step; }
        if (step <= 1) { break; }
    }
    const int gamma = i + split * direction + (direction < 0 ? -1 : 0);
    const int first = i < j ? i : j, last = i < j ? j : i;

    unsigned int left, right;
    if (first == gamma) { left = (unsigned int)gamma | kLeafFlag; parentOfLeaf[gamma] = i; }
    else { left = (unsigned int)gamma; parentOfInternal[gamma] = i; }
    if (last == gamma + 1) { right = (unsigned int)(gamma + 1) | kLeafFlag; parentOfLeaf[gamma + 1] = i; }
    else { right = (unsigned int)(gamma + 1); parentOfInternal[gamma + 1] = i; }
    children[i] = make_uint2(left, right);
    // the subtree's primitives are sorted entries first..last: triangles among them, and whether a sphere is
    // (exclusive prefix sums over the sorted order; entry `last` itself is added by its kind)
    const bool lastIsSphere = sorted[last] >= triangleCount;
    const unsigned int triangles = trianglesBefore[last] - trianglesBefore[first] + (lastIsSphere ? 0u : 1u);
    const unsigned int spheres = spheresBefore[last] - spheresBefore[first] + (lastIsSphere ? 1u : 0u);
    count[i] = triangles | (spheres ? kHasSphere : 0u);
    if (i == 0) { parentOfInternal[0] = -1; }
}

// Boxes bottom-up, LEVEL-SYNCHRONOUS: one launch per level of dependency.  A pass gives every internal node whose
// two children had their boxes BEFORE the pass its own box and marks it in the other copy of the flags; kernel
// boundaries make a pass's boxes visible to the next (the eight XCDs have separate L2s: inside one launch that
// visibility cost agent-scope fences and coherent loads, 36 of the build's 45 ms), and the double-buffered flags keep
// a node from being read in the pass that writes it.  As many passes as the radix tree is high (~45 for 5 M triangles).
__global__ __launch_bounds__(kLbvhBlock) void k_lbvh_fit_pass(
    int nInternal, const unsigned int *sorted, const float4 *boxLo, const float4 *boxHi, const uint2 *children,
    float4 *nodeLo, float4 *nodeHi, const unsigned char *readyIn, unsigned char *readyOut, unsigned int *pending)
{
    const int node = blockIdx.x * kLbvhBlock + threadIdx.x;
    bool waiting = false;
    if (node < nInternal) {
        unsigned char ready = readyIn[node];
        if (!ready) {
            const uint2 pair = children[node];
            const bool leftLeaf = (pair.x & kLeafFlag) != 0u, rightLeaf = (pair.y & kLeafFlag) != 0u;
            const bool leftReady = leftLeaf || readyIn[pair.x] != 0, rightReady = rightLeaf || readyIn[pair.y] != 0;
            if (leftReady && rightReady) {
                float4 lo, hi, otherLo, otherHi;
                if (leftLeaf) { const unsigned int tri = sorted[pair.x & ~kLeafFlag]; lo = boxLo[tri]; hi = boxHi[tri]; }
                else { lo = nodeLo[pair.x]; hi = nodeHi[pair.x]; }
                if (rightLeaf) { const unsigned int tri = sorted[pair.y & ~kLeafFlag]; otherLo = boxLo[tri]; otherHi = boxHi[tri]; }
                else { otherLo = nodeLo[pair.y]; otherHi = nodeHi[pair.y]; }
                nodeLo[node] = make_float4(fminf(lo.x, otherLo.x), fminf(lo.y, otherLo.y), fminf(lo.z, otherLo.z), 0.f);
                nodeHi[node] = make_float4(fmaxf(hi.x, otherHi.x), fmaxf(hi.y, otherHi.y), fmaxf(hi.z, otherHi.z), 0.f);
                ready = 1;
            } else {
                waiting = true;
            }
        }
        readyOut[node] = ready;
    }
    const unsigned long long mask = __ballot(waiting);
    if ((threadIdx.x & 63) == 0 && mask != 0ull) { atomicAdd(pending, (unsigned int)__popcll(mask)); }
}

// ------------------------------------------------------------------------- PLOC
// Parallel locally-ordered clustering (Meister & Bittner 2018): bottom-up agglomeration along the
// Morton order.  Every round, each cluster looks kPlocRadius places left and right for the
// neighbour whose union with it has the smallest surface area; mutual nearest neighbours merge
// into a new binary node; the cluster array is compacted (exclusive scans, so node ids and order
// are the same every run) and the round repeats until one cluster is left.  Quality is close to a
// top-down SAH build, at a few tens of launches.
constexpr int kPlocRadius = 16;      // default search radius
constexpr int kPlocMaxRadius = 64;   // PATHED_PLOC_RADIUS may raise it up to here (LDS tile size)

struct PlocClusters {
    float4 *lo, *hi;
    unsigned int *node;    // child reference: triangle | kLeafFlag, or binary node id
    unsigned int *count;   // triangles below
};

__global__ __launch_bounds__(kLbvhBlock) void k_ploc_init(
    uint32_t n, uint32_t triangleCount, const unsigned int *sorted, const float4 *boxLo, const float4 *boxHi, PlocClusters clusters)
{
    const uint32_t i = blockIdx.x * kLbvhBlock + threadIdx.x;
    if (i >= n) { return; }
    const unsigned int tri = sorted[i];
    clusters.lo[i] = boxLo[tri];
    clusters.hi[i] = boxHi[tri];
    clusters.node[i] = i | kLeafFlag;
    clusters.count[i] = tri >= triangleCount ? kHasSphere : 1u;
}

__global__ __launch_bounds__(kLbvhBlock) void k_ploc_nearest(PlocClusters clusters, unsigned int c, int radius, unsigned int *nearest)
{
    __shared__ float4 tileLo[kLbvhBlock + 2 * kPlocMaxRadius];
    __shared__ float4 tileHi[kLbvhBlock + 2 * kPlocMaxRadius];
    const long long blockStart = (long long)blockIdx.x * kLbvhBlock;
    for (int k = threadIdx.x; k < kLbvhBlock + 2 * radius; k += kLbvhBlock) {
        const long long j = blockStart - radius + k;
        if (j >= 0 && j < (long long)c) {
            tileLo[k] = clusters.lo[j];
            tileHi[k] = clusters.hi[j];
        }
    }
    __syncthreads();
    const long long i = blockStart + threadIdx.x;
    if (i >= (long long)c) { return; }
    const float4 lo = tileLo[threadIdx.x + radius], hi = tileHi[threadIdx.x + radius];
    float bestArea = __builtin_huge_valf();
    unsigned int best = (unsigned int)i;
    for (int d = -radius; d <= radius; d++) {
        const long long j = i + d;
        if (d == 0 || j < 0 || j >= (long long)c) { continue; }
        const float4 olo = tileLo[threadIdx.x + radius + d], ohi = tileHi[threadIdx.x + radius + d];
        const float dx = fmaxf(hi.x, ohi.x) - fminf(lo.x, olo.x);
        const float dy = fmaxf(hi.y, ohi.y) - fminf(lo.y, olo.y);
        const float dz = fmaxf(hi.z, ohi.z) - fminf(lo.z, olo.z);
        const float area = dx * dy + dy * dz + dz * dx;
        if (area < bestArea) { bestArea = area; best = (unsigned int)j; }   // ascending j: ties go to the lower index
    }
    nearest[i] = best;
}

// flags: x = this cluster survives the round, y = it absorbs its partner into a new node
__global__ __launch_bounds__(kLbvhBlock) void k_ploc_flags(unsigned int c, const unsigned int *nearest, unsigned int *survives, unsigned int *merges)
{
    const unsigned int i = blockIdx.x * kLbvhBlock + threadIdx.x;
    if (i >= c) { return; }
    const unsigned int j = nearest[i];
    const bool mutual = j != i && nearest[j] == i;
    survives[i] = (mutual && i > j) ? 0u : 1u;
    merges[i] = (mutual && i < j) ? 1u : 0u;
}

__global__ __launch_bounds__(kLbvhBlock) void k_ploc_merge(
    PlocClusters from, PlocClusters to, unsigned int c, const unsigned int *nearest,
    const unsigned int *survives, const unsigned int *survivorIndex, const unsigned int *merges, const unsigned int *mergeIndex,
    unsigned int nodeBase, uint2 *children, unsigned int *count, float4 *nodeLo, float4 *nodeHi, unsigned int *totals)
{
    const unsigned int i = blockIdx.x * kLbvhBlock + threadIdx.x;
    if (i >= c) { return; }
    if (i == c - 1) {
        totals[0] = survivorIndex[i] + survives[i];
        totals[1] = mergeIndex[i] + merges[i];
    }
    if (!survives[i]) { return; }
    const unsigned int position = survivorIndex[i];
    float4 lo = from.lo[i], hi = from.hi[i];
    unsigned int node = from.node[i], triangles = from.count[i];
    if (merges[i]) {
        const unsigned int j = nearest[i];
        const float4 olo = from.lo[j], ohi = from.hi[j];
        lo = make_float4(fminf(lo.x, olo.x), fminf(lo.y, olo.y), fminf(lo.z, olo.z), 0.f);
        hi = make_float4(fmaxf(hi.x, ohi.x), fmaxf(hi.y, ohi.y), fmaxf(hi.z, ohi.z), 0.f);
        const unsigned int id = nodeBase + mergeIndex[i];
        children[id] = make_uint2(node, from.node[j]);   // i < j: Morton order is kept left to right
        triangles = mergeCounts(triangles, from.count[j]);
        count[id] = triangles;
        nodeLo[id] = lo;
        nodeHi[id] = hi;
        node = id;
    }
    to.lo[position] = lo;
    to.hi[position] = hi;
    to.node[position] = node;
    to.count[position] = triangles;
}

// ------------------------------------------------------------------------- collapse to 4-wide
// binary nodes that stay inner nodes of the wide tree (more than kLbvhMaxLeaf triangles below)
__global__ __launch_bounds__(kLbvhBlock) void k_lbvh_count_inner(const unsigned int *count, int nInternal, unsigned int *total)
{
    __shared__ unsigned int partial[kLbvhBlock / 64];
    unsigned int mine = 0;
    for (size_t i = (size_t)blockIdx.x * kLbvhBlock + threadIdx.x; i < (size_t)nInternal; i += (size_t)gridDim.x * kLbvhBlock) {
        mine += count[i] > (unsigned int)kLbvhMaxLeaf ? 1u : 0u;
    }
    #pragma unroll
    for (int offset = 32; offset > 0; offset >>= 1) { mine += __shfl_xor(mine, offset); }
    if ((threadIdx.x & 63) == 0) { partial[threadIdx.x >> 6] = mine; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int sum = 0;
        #pragma unroll
        for (int w = 0; w < kLbvhBlock / 64; w++) { sum += partial[w]; }
        if (sum) { atomicAdd(total, sum); }   // one atomic per block of a bounded grid
    }
}

struct WideInputs {
    const uint2 *children;
    const unsigned int *count;       // triangles below a binary node
    const float4 *nodeLo, *nodeHi;
    const float4 *boxLo, *boxHi;
    const unsigned int *sorted;
    const float *positions;
    const uint32_t *indices;
    uint32_t triangleCount;          // primitives at or beyond it are spheres
};

// the packed count of a child reference (triangles below | kHasSphere)
__device__ inline unsigned int countBelow(const WideInputs &in, unsigned int ref)
{
    if (ref == kNoChild) { return 0u; }
    if (ref & kLeafFlag) { return in.sorted[ref & ~kLeafFlag] >= in.triangleCount ? kHasSphere : 1u; }
    return in.count[ref];
}

__device__ inline bool isInnerRef(const WideInputs &in, unsigned int ref)
{
    return ref != kNoChild && !(ref & kLeafFlag) && in.count[ref] > (unsigned int)kLbvhMaxLeaf;   // a sphere below counts as "more"
}

// A wide node adopts the two children of its binary root, then keeps replacing the inner child
// with the largest box by that child's two children until it has four (bvh_build.h, same rule).
// The children stay in left-to-right order of the binary tree, so that a subtree's triangles are
// consecutive in leaf order: child k starts at the node's first triangle + the triangles of 0..k-1.
__global__ __launch_bounds__(kLbvhBlock) void k_lbvh_wide_children(
    WideInputs in, const uint2 *frontier, unsigned int m, uint4 *adopted, unsigned int *innerCount)
{
    const unsigned int e = blockIdx.x * kLbvhBlock + threadIdx.x;
    if (e >= m) { return; }
    const uint2 pair = in.children[frontier[e].x];
    unsigned int c[4] = { pair.x, pair.y, kNoChild, kNoChild };
    int held = 2;
    while (held < 4) {
        int pick = -1;
        float pickArea = -1.f;
        for (int k = 0; k < held; k++) {
            if (!isInnerRef(in, c[k])) { continue; }
            const float4 lo = in.nodeLo[c[k]], hi = in.nodeHi[c[k]];
            const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
            const float area = dx * dy + dy * dz + dz * dx;
            if (area > pickArea) { pickArea = area; pick = k; }
        }
        if (pick < 0) { break; }
        const uint2 inside = in.children[c[pick]];
        for (int k = held; k > pick + 1; k--) { c[k] = c[k - 1]; }   // keep the order: both halves take the opened slot's place
        c[pick] = inside.x;
        c[pick + 1] = inside.y;
        held++;
    }
    adopted[e] = make_uint4(c[0], c[1], c[2], c[3]);
    innerCount[e] = (isInnerRef(in, c[0]) ? 1u : 0u) + (isInnerRef(in, c[1]) ? 1u : 0u)
        + (isInnerRef(in, c[2]) ? 1u : 0u) + (isInnerRef(in, c[3]) ? 1u : 0u);
}

// the (v0, prim) (e1) (e2) records of a leaf's triangles, left to right, starting at `first`
__device__ inline void writeLeafTriangles(const WideInputs &in, unsigned int ref, unsigned int first, float4 *leafTris)
{
    unsigned int stack[kLbvhMaxLeaf];
    int sp = 0;
    stack[sp++] = ref;
    unsigned int at = first;
    while (sp > 0) {
        const unsigned int r = stack[--sp];
        if (r & kLeafFlag) {
            const unsigned int prim = in.sorted[r & ~kLeafFlag];
            const float *v0 = in.positions + 3 * (size_t)in.indices[3 * (size_t)prim + 0];
            const float *v1 = in.positions + 3 * (size_t)in.indices[3 * (size_t)prim + 1];
            const float *v2 = in.positions + 3 * (size_t)in.indices[3 * (size_t)prim + 2];
            const float ax = v0[0], ay = v0[1], az = v0[2];
            leafTris[3 * (size_t)at + 0] = make_float4(ax, ay, az, __int_as_float((int)prim));
            leafTris[3 * (size_t)at + 1] = make_float4(v1[0] - ax, v1[1] - ay, v1[2] - az, 0.f);
            leafTris[3 * (size_t)at + 2] = make_float4(v2[0] - ax, v2[1] - ay, v2[2] - az, 0.f);
            at++;
        } else {
            const uint2 pair = in.children[r];
            if (sp + 2 <= kLbvhMaxLeaf) {   // always true below a node of <= kLbvhMaxLeaf triangles
                stack[sp++] = pair.y;
                stack[sp++] = pair.x;
            }
        }
    }
}

__global__ __launch_bounds__(kLbvhBlock) void k_lbvh_wide_emit(
    WideInputs in, const uint2 *frontier, unsigned int m, unsigned int levelBase,
    const uint4 *adopted, const unsigned int *innerCount, const unsigned int *childBase,
    uint2 *nextFrontier, unsigned int *nextCount, float4 *nodesOut, float4 *leafTris)
{
    const unsigned int e = blockIdx.x * kLbvhBlock + threadIdx.x;
    if (e >= m) { return; }
    const uint4 four = adopted[e];
    const unsigned int base = childBase[e];
    unsigned int first = frontier[e].y;   // leaf-order position of this subtree's first triangle
    unsigned int rank = 0;
    float lo[3][4], hi[3][4];
    int refs[4];
    #pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned int ref = k == 0 ? four.x : k == 1 ? four.y : k == 2 ? four.z : four.w;
        float4 boxLow = make_float4(0.f, 0.f, 0.f, 0.f), boxHigh = boxLow;
        refs[k] = kEmptyRef;
        const bool present = ref != kNoChild;
        if (present) {
            const unsigned int packed = countBelow(in, ref);
            const unsigned int triangles = packed & ~kHasSphere;
            if (ref & kLeafFlag) {
                const unsigned int tri = in.sorted[ref & ~kLeafFlag];
                boxLow = in.boxLo[tri];
                boxHigh = in.boxHi[tri];
            } else {
                boxLow = in.nodeLo[ref];
                boxHigh = in.nodeHi[ref];
            }
            if ((ref & kLeafFlag) && (packed & kHasSphere)) {
                // a sphere: a leaf of its own, count 0, first = sphere + 1 (bvh_build.h; trace.h tests it); no place in leaf order
                const unsigned int sphere = in.sorted[ref & ~kLeafFlag] - in.triangleCount;
                refs[k] = -(int)(((sphere + 1u) << 3) | 0u) - 1;
            } else if (packed <= (unsigned int)kLbvhMaxLeaf) {
                refs[k] = -(int)((first << 3) | triangles) - 1;   // trace.h encodeLeaf
                writeLeafTriangles(in, ref, first, leafTris);
            } else {
                refs[k] = (int)(levelBase + m + base + rank);     // next level's ids follow this level's
                nextFrontier[base + rank] = make_uint2(ref, first);
                rank++;
            }
            first += triangles;
        }
        const float low[3] = { boxLow.x, boxLow.y, boxLow.z }, high[3] = { boxHigh.x, boxHigh.y, boxHigh.z };
        #pragma unroll
        for (int a = 0; a < 3; a++) {
            // bvh_build.h padBox: a hit computed in fp32 on a face lying in a box plane must survive the slab test
            const float pad = 1e-5f * fmaxf(1.f, fmaxf(fabsf(low[a]), fabsf(high[a])));
            lo[a][k] = present ? low[a] - pad : 0.f;
            hi[a][k] = present ? high[a] + pad : 0.f;
        }
    }
    float4 *node = nodesOut + (size_t)8 * (levelBase + e);
    #pragma unroll
    for (int a = 0; a < 3; a++) {
        node[a] = make_float4(lo[a][0], lo[a][1], lo[a][2], lo[a][3]);
        node[3 + a] = make_float4(hi[a][0], hi[a][1], hi[a][2], hi[a][3]);
    }
    node[6] = make_float4(__int_as_float(refs[0]), __int_as_float(refs[1]), __int_as_float(refs[2]), __int_as_float(refs[3]));
    node[7] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e == m - 1) { *nextCount = base + innerCount[e]; }
}

// ------------------------------------------------------------------------- SAH top of the tree
// The last few thousand clusters are joined on the host by a top-down binned-SAH build (weighted by the
// triangles below each cluster) instead of by more clustering rounds: near the root every ray pays for a
// bad split, the Morton-order window sees too little there, and a thousand boxes take well under a millisecond.
// (The same idea as HLBVH's "SAH over treelet roots", Garanzha et al. 2011.)
struct TopItem {
    float lo[3], hi[3];
    unsigned int ref, triangles;
};

struct TopNodes {
    std::vector<uint2> children;
    std::vector<unsigned int> count;
    std::vector<float4> lo, hi;
};

// builds the subtree over items[begin, end) (reordered in place); returns its child reference
unsigned int buildTopSah(std::vector<TopItem> &items, size_t begin, size_t end, unsigned int nodeBase, TopNodes &out)
{
    if (end - begin == 1) { return items[begin].ref; }
    float lo[3] = { 3e38f, 3e38f, 3e38f }, hi[3] = { -3e38f, -3e38f, -3e38f }, clo[3] = { 3e38f, 3e38f, 3e38f }, chi[3] = { -3e38f, -3e38f, -3e38f };
    unsigned int triangles = 0;
    for (size_t i = begin; i < end; i++) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], items[i].lo[a]);
            hi[a] = std::max(hi[a], items[i].hi[a]);
            const float centre = 0.5f * (items[i].lo[a] + items[i].hi[a]);
            clo[a] = std::min(clo[a], centre);
            chi[a] = std::max(chi[a], centre);
        }
        triangles = mergeCounts(triangles, items[i].triangles);
    }
    auto halfArea = [](const float *l, const float *h) {
        const float dx = h[0] - l[0], dy = h[1] - l[1], dz = h[2] - l[2];
        return dx >= 0.f ? dx * dy + dy * dz + dz * dx : 0.f;
    };
    const int kBins = 16;
    int bestAxis = -1, bestSplit = -1;
    float bestCost = 3e38f;
    for (int axis = 0; axis < 3; axis++) {
        if (!(chi[axis] > clo[axis])) { continue; }
        const float scale = (float)kBins / (chi[axis] - clo[axis]);
        float binLo[kBins][3], binHi[kBins][3];
        unsigned int binTriangles[kBins];
        size_t binItems[kBins];
        for (int b = 0; b < kBins; b++) {
            for (int a = 0; a < 3; a++) { binLo[b][a] = 3e38f; binHi[b][a] = -3e38f; }
            binTriangles[b] = 0;
            binItems[b] = 0;
        }
        for (size_t i = begin; i < end; i++) {
            int b = (int)((0.5f * (items[i].lo[axis] + items[i].hi[axis]) - clo[axis]) * scale);
            b = b < 0 ? 0 : b > kBins - 1 ? kBins - 1 : b;
            for (int a = 0; a < 3; a++) { binLo[b][a] = std::min(binLo[b][a], items[i].lo[a]); binHi[b][a] = std::max(binHi[b][a], items[i].hi[a]); }
            binTriangles[b] += (items[i].triangles & ~kHasSphere) + ((items[i].triangles & kHasSphere) ? 1u : 0u);   // SAH weight
            binItems[b]++;
        }
        float rightArea[kBins];
        unsigned int rightTriangles[kBins];
        size_t rightItems[kBins];
        float accLo[3] = { 3e38f, 3e38f, 3e38f }, accHi[3] = { -3e38f, -3e38f, -3e38f };
        unsigned int runningTriangles = 0;
        size_t runningItems = 0;
        for (int b = kBins - 1; b > 0; b--) {
            for (int a = 0; a < 3; a++) { accLo[a] = std::min(accLo[a], binLo[b][a]); accHi[a] = std::max(accHi[a], binHi[b][a]); }
            runningTriangles += binTriangles[b];
            runningItems += binItems[b];
            rightArea[b] = halfArea(accLo, accHi);
            rightTriangles[b] = runningTriangles;
            rightItems[b] = runningItems;
        }
        for (int a = 0; a < 3; a++) { accLo[a] = 3e38f; accHi[a] = -3e38f; }
        runningTriangles = 0;
        runningItems = 0;
        for (int b = 0; b < kBins - 1; b++) {
            for (int a = 0; a < 3; a++) { accLo[a] = std::min(accLo[a], binLo[b][a]); accHi[a] = std::max(accHi[a], binHi[b][a]); }
            runningTriangles += binTriangles[b];
            runningItems += binItems[b];
            if (runningItems == 0 || rightItems[b + 1] == 0) { continue; }
            const float cost = halfArea(accLo, accHi) * (float)runningTriangles + rightArea[b + 1] * (float)rightTriangles[b + 1];
            if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestSplit = b; }
        }
    }
    size_t mid = begin;
    if (bestAxis >= 0) {
        const float scale = (float)kBins / (chi[bestAxis] - clo[bestAxis]);
        const float origin = clo[bestAxis];
        const int axis = bestAxis, split = bestSplit;
        auto middle = std::partition(items.begin() + (long)begin, items.begin() + (long)end, [=](const TopItem &item) {
            int b = (int)((0.5f * (item.lo[axis] + item.hi[axis]) - origin) * scale);
            b = b < 0 ? 0 : b > kBins - 1 ? kBins - 1 : b;
            return b <= split;
        });
        mid = (size_t)(middle - items.begin());
    }
    if (mid == begin || mid == end) { mid = begin + (end - begin) / 2; }   // coincident centres: halve in the given order
    const unsigned int left = buildTopSah(items, begin, mid, nodeBase, out);
    const unsigned int right = buildTopSah(items, mid, end, nodeBase, out);
    out.children.push_back(make_uint2(left, right));
    out.count.push_back(triangles);
    out.lo.push_back(make_float4(lo[0], lo[1], lo[2], 0.f));
    out.hi.push_back(make_float4(hi[0], hi[1], hi[2], 0.f));
    return nodeBase + (unsigned int)(out.children.size() - 1);
}

struct Scratch {
    void *pointers[48];
    int used = 0;
    hipError_t status = hipSuccess;

    template <typename T>
    T *get(size_t count)
    {
        if (status != hipSuccess) { return nullptr; }
        if (used >= 48) { status = hipErrorOutOfMemory; return nullptr; }
        void *p = nullptr;
        status = hipMalloc(&p, (count ? count : 1) * sizeof(T));
        if (status != hipSuccess) { return nullptr; }
        pointers[used++] = p;
        return static_cast<T *>(p);
    }

    ~Scratch()
    {
        for (int i = 0; i < used; i++) { (void)hipFree(pointers[i]); }
    }
};

inline unsigned int blocksFor(size_t n) { return (unsigned int)((n + kLbvhBlock - 1) / kLbvhBlock); }

}  // namespace

hipError_t buildBvhOnDevice(int builder, const float *positions, const uint32_t *indices, uint32_t triangleCount,
                            const float4 *spheres, uint32_t sphereCount, hipStream_t stream, DeviceBvh *out, std::string *error)
{
    auto failed = [&](hipError_t status, const char *what) {
        if (error) { *error = std::string("device bvh: ") + what + ": " + hipGetErrorString(status); }
        if (out->nodes) { (void)hipFree(out->nodes); out->nodes = nullptr; }
        if (out->leafTris) { (void)hipFree(out->leafTris); out->leafTris = nullptr; }
        return status == hipSuccess ? hipErrorInvalidValue : status;
    };
    *out = DeviceBvh();
    const uint32_t n = triangleCount + sphereCount;   // primitives: triangles, then spheres
    if (builder != kDeviceBuilderLbvh && builder != kDeviceBuilderPloc) { return failed(hipErrorInvalidValue, "unknown builder"); }
    if (triangleCount <= (uint32_t)kLbvhMaxLeaf) { return failed(hipErrorInvalidValue, "fewer than five triangles"); }
    if (n >= (1u << 28) || n < triangleCount) { return failed(hipErrorInvalidValue, "more than 2^28 primitives"); }
    if (sphereCount > 0 && !spheres) { return failed(hipErrorInvalidValue, "null sphere array"); }

    hipEvent_t started = nullptr, finished = nullptr;
    hipError_t status;
    if ((status = hipEventCreate(&started)) != hipSuccess) { return failed(status, "event"); }
    if ((status = hipEventCreate(&finished)) != hipSuccess) { (void)hipEventDestroy(started); return failed(status, "event"); }
    struct EventGuard {
        hipEvent_t a, b;
        ~EventGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
    } guard { started, finished };
    (void)hipEventRecord(started, stream);

    Scratch scratch;
    float4 *boxLo = scratch.get<float4>(n);
    float4 *boxHi = scratch.get<float4>(n);
    unsigned int *words = scratch.get<unsigned int>(16);   // 0..5 centroid bounds, 6 inner count, 7 next frontier size, 8..9 PLOC totals
    unsigned long long *keysIn = scratch.get<unsigned long long>(n);
    unsigned long long *keysOut = scratch.get<unsigned long long>(n);
    unsigned int *valuesIn = scratch.get<unsigned int>(n);
    unsigned int *sorted = scratch.get<unsigned int>(n);
    uint2 *children = scratch.get<uint2>(n - 1);
    unsigned int *count = scratch.get<unsigned int>(n - 1);
    float4 *nodeLo = scratch.get<float4>(n - 1);
    float4 *nodeHi = scratch.get<float4>(n - 1);
    if (scratch.status != hipSuccess) { return failed(scratch.status, "scratch allocation"); }

    // 1-2: boxes, centroid bounds, Morton codes
    {
        const unsigned int init[10] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u, 0u, 0u };
        if ((status = hipMemcpyAsync(words, init, sizeof init, hipMemcpyHostToDevice, stream)) != hipSuccess) { return failed(status, "init"); }
        if ((status = hipStreamSynchronize(stream)) != hipSuccess) { return failed(status, "init"); }   // `init` is on this stack frame
    }
    const unsigned int reduceGrid = blocksFor(n) < kReduceBlocks ? blocksFor(n) : kReduceBlocks;
    const unsigned int triangleGrid = blocksFor(triangleCount) < kReduceBlocks ? blocksFor(triangleCount) : kReduceBlocks;
    hipLaunchKernelGGL(k_lbvh_prims, dim3(triangleGrid), dim3(kLbvhBlock), 0, stream, positions, indices, triangleCount, boxLo, boxHi, words);
    if (sphereCount > 0) {
        hipLaunchKernelGGL(k_lbvh_sphere_prims, dim3(blocksFor(sphereCount)), dim3(kLbvhBlock), 0, stream, spheres, sphereCount, triangleCount, boxLo, boxHi, words);
    }
    hipLaunchKernelGGL(k_lbvh_morton, dim3(blocksFor(n)), dim3(kLbvhBlock), 0, stream, boxLo, boxHi, n, words, keysIn, valuesIn);

    // 3: sort
    {
        size_t bytes = 0;
        if ((status = rocprim::radix_sort_pairs(nullptr, bytes, keysIn, keysOut, valuesIn, sorted, (size_t)n, 0u, 63u, stream)) != hipSuccess) {
            return failed(status, "radix sort (query)");
        }
        void *temporary = scratch.get<unsigned char>(bytes);
        if (scratch.status != hipSuccess) { return failed(scratch.status, "radix sort scratch"); }
        if ((status = rocprim::radix_sort_pairs(temporary, bytes, keysIn, keysOut, valuesIn, sorted, (size_t)n, 0u, 63u, stream)) != hipSuccess) {
            return failed(status, "radix sort");
        }
    }

    unsigned int root = 0;
    int rounds = 0;
    if (builder == kDeviceBuilderLbvh) {
        // 4-5: Karras hierarchy (root = node 0), boxes bottom-up
        int *parentOfInternal = scratch.get<int>(n - 1);
        int *parentOfLeaf = scratch.get<int>(n);
        unsigned char *ready[2] = { scratch.get<unsigned char>(n), scratch.get<unsigned char>(n) };
        unsigned int *pending = scratch.get<unsigned int>(64);
        if (scratch.status != hipSuccess) { return failed(scratch.status, "scratch allocation"); }
        if ((status = hipMemsetAsync(ready[0], 0, (size_t)n, stream)) != hipSuccess) { return failed(status, "memset"); }
        // triangles / spheres before every sorted entry (exclusive scans): a node's count is read off its sorted range
        unsigned int *isTriangle = scratch.get<unsigned int>(n), *isSphere = scratch.get<unsigned int>(n);
        unsigned int *trianglesBefore = scratch.get<unsigned int>(n), *spheresBefore = scratch.get<unsigned int>(n);
        if (scratch.status != hipSuccess) { return failed(scratch.status, "scratch allocation"); }
        hipLaunchKernelGGL(k_lbvh_kinds, dim3(blocksFor(n)), dim3(kLbvhBlock), 0, stream, sorted, n, triangleCount, isTriangle, isSphere);
        {
            size_t kindBytes = 0;
            if ((status = rocprim::exclusive_scan(nullptr, kindBytes, isTriangle, trianglesBefore, 0u, (size_t)n, rocprim::plus<unsigned int>(), stream)) != hipSuccess) { return failed(status, "scan (query)"); }
            void *kindTemporary = scratch.get<unsigned char>(kindBytes);
            if (scratch.status != hipSuccess) { return failed(scratch.status, "scan scratch"); }
            size_t bytes = kindBytes;
            if ((status = rocprim::exclusive_scan(kindTemporary, bytes, isTriangle, trianglesBefore, 0u, (size_t)n, rocprim::plus<unsigned int>(), stream)) != hipSuccess) { return failed(status, "scan"); }
            bytes = kindBytes;
            if ((status = rocprim::exclusive_scan(kindTemporary, bytes, isSphere, spheresBefore, 0u, (size_t)n, rocprim::plus<unsigned int>(), stream)) != hipSuccess) { return failed(status, "scan"); }
        }
        hipLaunchKernelGGL(k_lbvh_hierarchy, dim3(blocksFor(n - 1)), dim3(kLbvhBlock), 0, stream,
                           keysOut, (int)n, trianglesBefore, spheresBefore, sorted, triangleCount, children, count, parentOfInternal, parentOfLeaf);
        // boxes: one pass per level; the count of nodes still waiting is read back every eighth pass
        const int passesPerCheck = 8;
        bool fitted = false;
        for (int pass = 0; pass < 512 && !fitted; pass += passesPerCheck) {
            if ((status = hipMemsetAsync(pending, 0, passesPerCheck * sizeof(unsigned int), stream)) != hipSuccess) { return failed(status, "memset"); }
            for (int k = 0; k < passesPerCheck; k++) {
                hipLaunchKernelGGL(k_lbvh_fit_pass, dim3(blocksFor(n - 1)), dim3(kLbvhBlock), 0, stream,
                                   (int)(n - 1), sorted, boxLo, boxHi, children, nodeLo, nodeHi, ready[(pass + k) & 1], ready[(pass + k + 1) & 1], pending + k);
            }
            unsigned int waiting[passesPerCheck];
            if ((status = hipMemcpyAsync(waiting, pending, sizeof waiting, hipMemcpyDeviceToHost, stream)) != hipSuccess) { return failed(status, "read back"); }
            if ((status = hipStreamSynchronize(stream)) != hipSuccess) { return failed(status, "LBVH box pass"); }
            for (int k = 0; k < passesPerCheck; k++) { fitted = fitted || waiting[k] == 0u; }
        }
        if (!fitted) { return failed(hipErrorUnknown, "LBVH boxes did not converge"); }
        root = 0;
    } else {
        // 4-5 (PLOC): agglomerate along the Morton order until one cluster is left
        PlocClusters a, b;
        a.lo = scratch.get<float4>(n); a.hi = scratch.get<float4>(n); a.node = scratch.get<unsigned int>(n); a.count = scratch.get<unsigned int>(n);
        b.lo = scratch.get<float4>(n); b.hi = scratch.get<float4>(n); b.node = scratch.get<unsigned int>(n); b.count = scratch.get<unsigned int>(n);
        unsigned int *nearest = scratch.get<unsigned int>(n);
        unsigned int *survives = scratch.get<unsigned int>(n);
        unsigned int *survivorIndex = scratch.get<unsigned int>(n);
        unsigned int *merges = scratch.get<unsigned int>(n);
        unsigned int *mergeIndex = scratch.get<unsigned int>(n);
        if (scratch.status != hipSuccess) { return failed(scratch.status, "scratch allocation"); }
        size_t scanBytes = 0;
        if ((status = rocprim::exclusive_scan(nullptr, scanBytes, survives, survivorIndex, 0u, (size_t)n, rocprim::plus<unsigned int>(), stream)) != hipSuccess) {
            return failed(status, "scan (query)");
        }
        void *scanTemporary = scratch.get<unsigned char>(scanBytes);
        if (scratch.status != hipSuccess) { return failed(scratch.status, "scan scratch"); }

        int radius = kPlocRadius;
#if PATHED_EXPERIMENTS
        if (const char *text = getenv("PATHED_PLOC_RADIUS")) {   // tuning (experiments build only: the product library reads no PATHED_* variable)
            const int value = atoi(text);
            if (value >= 1 && value <= kPlocMaxRadius) { radius = value; }
        }
#endif
        hipLaunchKernelGGL(k_ploc_init, dim3(blocksFor(n)), dim3(kLbvhBlock), 0, stream, n, triangleCount, sorted, boxLo, boxHi, a);
        unsigned int topLimit = 1024;   // clusters left to the host's SAH build (0: cluster all the way); 1 024: fewer rounds AND a better top
#if PATHED_EXPERIMENTS
        if (const char *text = getenv("PATHED_PLOC_TOP")) {
            const int value = atoi(text);
            if (value >= 0 && value <= (1 << 20)) { topLimit = (unsigned int)value; }
        }
#endif
        unsigned int c = n, nodeBase = 0;
        while (c > 1 && c > topLimit) {
            if (++rounds > 4096) { return failed(hipErrorInvalidValue, "clustering does not converge"); }
            hipLaunchKernelGGL(k_ploc_nearest, dim3(blocksFor(c)), dim3(kLbvhBlock), 0, stream, a, c, radius, nearest);
            hipLaunchKernelGGL(k_ploc_flags, dim3(blocksFor(c)), dim3(kLbvhBlock), 0, stream, c, nearest, survives, merges);
            size_t bytes = scanBytes;
            if ((status = rocprim::exclusive_scan(scanTemporary, bytes, survives, survivorIndex, 0u, (size_t)c, rocprim::plus<unsigned int>(), stream)) != hipSuccess) { return failed(status, "scan"); }
            bytes = scanBytes;
            if ((status = rocprim::exclusive_scan(scanTemporary, bytes, merges, mergeIndex, 0u, (size_t)c, rocprim::plus<unsigned int>(), stream)) != hipSuccess) { return failed(status, "scan"); }
            hipLaunchKernelGGL(k_ploc_merge, dim3(blocksFor(c)), dim3(kLbvhBlock), 0, stream,
                               a, b, c, nearest, survives, survivorIndex, merges, mergeIndex, nodeBase, children, count, nodeLo, nodeHi, words + 8);
            unsigned int totals[2] = { 0u, 0u };
            if ((status = hipMemcpyAsync(totals, words + 8, sizeof totals, hipMemcpyDeviceToHost, stream)) != hipSuccess) { return failed(status, "round totals"); }
            if ((status = hipStreamSynchronize(stream)) != hipSuccess) { return failed(status, "clustering kernels"); }
            // the closest pair of a round is always mutual, so every round merges at least once
            if (totals[1] == 0u || totals[0] + totals[1] != c || (size_t)nodeBase + totals[1] > (size_t)n - 1) {
                return failed(hipErrorInvalidValue, "clustering round made no progress");
            }
            nodeBase += totals[1];
            c = totals[0];
            const PlocClusters swap = a;
            a = b;
            b = swap;
        }
        if (c > 1) {
            // the top of the tree: SAH over the remaining clusters, on the host
            std::vector<float4> hostLo(c), hostHi(c);
            std::vector<unsigned int> hostNode(c), hostCount(c);
            if ((status = hipMemcpy(hostLo.data(), a.lo, c * sizeof(float4), hipMemcpyDeviceToHost)) != hipSuccess) { return failed(status, "top clusters"); }
            if ((status = hipMemcpy(hostHi.data(), a.hi, c * sizeof(float4), hipMemcpyDeviceToHost)) != hipSuccess) { return failed(status, "top clusters"); }
            if ((status = hipMemcpy(hostNode.data(), a.node, c * sizeof(unsigned int), hipMemcpyDeviceToHost)) != hipSuccess) { return failed(status, "top clusters"); }
            if ((status = hipMemcpy(hostCount.data(), a.count, c * sizeof(unsigned int), hipMemcpyDeviceToHost)) != hipSuccess) { return failed(status, "top clusters"); }
            std::vector<TopItem> items(c);
            for (unsigned int i = 0; i < c; i++) {
                items[i].lo[0] = hostLo[i].x; items[i].lo[1] = hostLo[i].y; items[i].lo[2] = hostLo[i].z;
                items[i].hi[0] = hostHi[i].x; items[i].hi[1] = hostHi[i].y; items[i].hi[2] = hostHi[i].z;
                items[i].ref = hostNode[i];
                items[i].triangles = hostCount[i];
            }
            TopNodes top;
            top.children.reserve(c); top.count.reserve(c); top.lo.reserve(c); top.hi.reserve(c);
            buildTopSah(items, 0, c, nodeBase, top);
            const size_t made = top.children.size();
            if (made != (size_t)c - 1 || (size_t)nodeBase + made != (size_t)n - 1) { return failed(hipErrorInvalidValue, "top build made a wrong node count"); }
            if ((status = hipMemcpy(children + nodeBase, top.children.data(), made * sizeof(uint2), hipMemcpyHostToDevice)) != hipSuccess) { return failed(status, "top nodes"); }
            if ((status = hipMemcpy(count + nodeBase, top.count.data(), made * sizeof(unsigned int), hipMemcpyHostToDevice)) != hipSuccess) { return failed(status, "top nodes"); }
            if ((status = hipMemcpy(nodeLo + nodeBase, top.lo.data(), made * sizeof(float4), hipMemcpyHostToDevice)) != hipSuccess) { return failed(status, "top nodes"); }
            if ((status = hipMemcpy(nodeHi + nodeBase, top.hi.data(), made * sizeof(float4), hipMemcpyHostToDevice)) != hipSuccess) { return failed(status, "top nodes"); }
            nodeBase += (unsigned int)made;
        }
        if (nodeBase != n - 1) { return failed(hipErrorInvalidValue, "clustering ended with a wrong node count"); }
        root = n - 2;   // the last node created
    }

    // 6: wide nodes.  Every wide node is rooted at a binary node with more than four triangles
    // below it, so their number bounds the allocation.
    hipLaunchKernelGGL(k_lbvh_count_inner, dim3(reduceGrid), dim3(kLbvhBlock), 0, stream, count, (int)(n - 1), words + 6);
    unsigned int capacity = 0;
    if ((status = hipMemcpyAsync(&capacity, words + 6, sizeof capacity, hipMemcpyDeviceToHost, stream)) != hipSuccess) { return failed(status, "count"); }
    if ((status = hipStreamSynchronize(stream)) != hipSuccess) { return failed(status, "hierarchy kernels"); }
    if (capacity == 0) { return failed(hipErrorInvalidValue, "empty hierarchy"); }

    if ((status = hipMalloc((void **)&out->nodes, (size_t)capacity * 8 * sizeof(float4))) != hipSuccess) { return failed(status, "node allocation"); }
    if ((status = hipMalloc((void **)&out->leafTris, (size_t)triangleCount * 3 * sizeof(float4))) != hipSuccess) { return failed(status, "triangle allocation"); }
    out->nodeCapacity = capacity;

    uint2 *frontierA = scratch.get<uint2>(capacity);
    uint2 *frontierB = scratch.get<uint2>(capacity);
    uint4 *adopted = scratch.get<uint4>(capacity);
    unsigned int *innerCount = scratch.get<unsigned int>(capacity);
    unsigned int *childBase = scratch.get<unsigned int>(capacity);
    if (scratch.status != hipSuccess) { return failed(scratch.status, "frontier allocation"); }
    size_t scanBytes = 0;
    if ((status = rocprim::exclusive_scan(nullptr, scanBytes, innerCount, childBase, 0u, (size_t)capacity, rocprim::plus<unsigned int>(), stream)) != hipSuccess) {
        return failed(status, "scan (query)");
    }
    void *scanTemporary = scratch.get<unsigned char>(scanBytes);
    if (scratch.status != hipSuccess) { return failed(scratch.status, "scan scratch"); }

    WideInputs inputs;
    inputs.children = children;
    inputs.count = count;
    inputs.nodeLo = nodeLo;
    inputs.nodeHi = nodeHi;
    inputs.boxLo = boxLo;
    inputs.boxHi = boxHi;
    inputs.sorted = sorted;
    inputs.positions = positions;
    inputs.indices = indices;
    inputs.triangleCount = triangleCount;

    {
        const uint2 rootEntry = make_uint2(root, 0u);   // (binary node, first triangle in leaf order)
        if ((status = hipMemcpyAsync(frontierA, &rootEntry, sizeof rootEntry, hipMemcpyHostToDevice, stream)) != hipSuccess) { return failed(status, "root"); }
        if ((status = hipStreamSynchronize(stream)) != hipSuccess) { return failed(status, "root"); }
    }
    unsigned int levelBase = 0, m = 1;
    int depth = 0;
    uint2 *frontier = frontierA, *nextFrontier = frontierB;
    while (m > 0) {
        if ((size_t)levelBase + m > capacity) { return failed(hipErrorInvalidValue, "wide node count exceeds its bound"); }
        if (++depth > 4096) { return failed(hipErrorInvalidValue, "runaway hierarchy depth"); }
        hipLaunchKernelGGL(k_lbvh_wide_children, dim3(blocksFor(m)), dim3(kLbvhBlock), 0, stream, inputs, frontier, m, adopted, innerCount);
        size_t bytes = scanBytes;
        if ((status = rocprim::exclusive_scan(scanTemporary, bytes, innerCount, childBase, 0u, (size_t)m, rocprim::plus<unsigned int>(), stream)) != hipSuccess) {
            return failed(status, "scan");
        }
        hipLaunchKernelGGL(k_lbvh_wide_emit, dim3(blocksFor(m)), dim3(kLbvhBlock), 0, stream,
                           inputs, frontier, m, levelBase, adopted, innerCount, childBase, nextFrontier, words + 7, out->nodes, out->leafTris);
        unsigned int next = 0;
        if ((status = hipMemcpyAsync(&next, words + 7, sizeof next, hipMemcpyDeviceToHost, stream)) != hipSuccess) { return failed(status, "frontier size"); }
        if ((status = hipStreamSynchronize(stream)) != hipSuccess) { return failed(status, "collapse kernels"); }
        levelBase += m;
        m = next;
        uint2 *swap = frontier;
        frontier = nextFrontier;
        nextFrontier = swap;
    }
    out->nodeCount = (int)levelBase;
    out->maxDepth = depth;
    out->rounds = rounds;

    (void)hipEventRecord(finished, stream);
    if ((status = hipStreamSynchronize(stream)) != hipSuccess) { return failed(status, "build"); }
    if ((status = hipGetLastError()) != hipSuccess) { return failed(status, "kernel launch"); }
    (void)hipEventElapsedTime(&out->buildMs, started, finished);
    return hipSuccess;
}

}  // namespace pathed
