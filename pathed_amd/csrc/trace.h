// BVH4 traversal + primitive intersection for CDNA4 (the stand-in for Embree's
// rtcIntersect1 / rtcOccluded1, reference src/scene.cpp:113, :374).
//
// Layout (DESIGN.md "Data layout"):
//   node   8 x float4 = 128 B, children in the four components:
//          (lo.x[4]) (lo.y[4]) (lo.z[4]) (hi.x[4]) (hi.y[4]) (hi.z[4]) (ref[4]) (-)
//          ref >= 0: inner node index; ref <= -2: leaf, -ref - 1 = (first triangle << 3) | count;
//          ref == kEmptyChild: empty slot
//   nodeQ  4 x float4 = 64 B, the COMPRESSED form of the same node (same index, same refs; k_compress_nodes): the
//          children's boxes on an 8-bit grid over their union,
//          (origin.xyz, scale.x) (scale.y, scale.z, qlo.x[4], qlo.y[4]) (qlo.z[4], qhi.x[4], qhi.y[4], qhi.z[4]) (ref[4])
//          plane = origin + q * scale, scale a power of two; q rounded OUTWARD plus 1/256 of a step, so the grid box
//          contains the float box and hits stay bit-exact (the triangle test decides, box tests only cull).  Half the
//          bytes per visit and four 16-byte loads per lane instead of seven, for one v_cvt_f32_ubyte per plane.
//   node8  8 x float4 = 128 B, the 8-WIDE compressed form (k_widen_nodes): node i keeps its index and takes the place of
//          its 4-wide node; children of its children are pulled up while they fit (largest box first), the nodes they came
//          from are simply never referenced again.  Same grid as nodeQ, two q words per plane (children 0-3, 4-7):
//          (origin.xyz, scale.x) (scale.y, scale.z, qlo.x[8]) (qlo.y[8], qlo.z[8]) (qhi.x[8], qhi.y[8]) (qhi.z[8], -, -)
//          (ref[0..3]) (ref[4..7]) (-)
//   tri    3 x float4 = 48 B: (v0.xyz, prim) (e1.xyz, -) (e2.xyz, -), leaf order
// Per-lane traversal stack: the first ROWS entries live in LDS, laid out [row][thread] so a
// wave's 64 lanes hit 64 consecutive dwords (bank-conflict-free ds_read_b32 / ds_write_b32);
// deeper entries (rare: a 4-wide tree of depth d can stack 3d) spill to a per-thread column in HBM.
//
// The traversal is written as a per-lane STATE MACHINE (LaneRay + innerStep / leafStep): one call
// visits one inner node or tests one leaf.  The persistent kernel interleaves steps of 64 independent rays and
// refills lanes whose ray has finished from the ray pool, so a wave is not held hostage by its
// slowest ray (ray costs are heavy-tailed: mean ~10 node visits, worst several hundred).
//
// Intersector specification (shared with the CPU checker so hits agree bit for bit):
//   Moeller-Trumbore with xcross/xdot (explicit fmaf), u,v in Embree's convention,
//   hit accepted iff t > tnear and (t < best, or t == best and prim < bestPrim);
//   spheres by projection onto the ray.  Box tests only need to be conservative.
#pragma once

#include "device_scene.h"

namespace pathed {

struct TraceCounters {
    unsigned int boxes;
    unsigned int tris;
};

struct RayHit {
    float t, u, v;
    int prim;
};

// Inside test in "det units": no division for the triangles a ray misses (an IEEE fp32
// division is ~10 VALU instructions), one division for a hit.
__device__ inline bool intersectTriangle(V3 o, V3 d, V3 v0, V3 e1, V3 e2, float *t, float *u, float *v)
{
    const V3 pvec = xcross(d, e2);
    const float det = xdot(e1, pvec);
    const V3 tvec = o - v0;
    const float uScaled = xdot(tvec, pvec);
    const V3 qvec = xcross(tvec, e1);
    const float vScaled = xdot(d, qvec);
    if (det > 0.f) {
        if (!(uScaled >= 0.f && vScaled >= 0.f && uScaled + vScaled <= det)) { return false; }
    } else if (det < 0.f) {
        if (!(uScaled <= 0.f && vScaled <= 0.f && uScaled + vScaled >= det)) { return false; }
    } else {
        return false;  // parallel, degenerate or NaN
    }
    const float inv = 1.f / det;
    *t = xdot(e2, qvec) * inv;
    *u = uScaled * inv;
    *v = vScaled * inv;
    return true;
}

__device__ inline bool intersectSphere(V3 o, V3 d, V3 center, float radius, float tnear, float *t)
{
    const V3 c0 = center - o;
    const float dd = xdot(d, d);
    const float projection = xdot(c0, d) / dd;
    const V3 perpendicular = c0 - d * projection;
    const float l2 = xdot(perpendicular, perpendicular);
    const float r2 = radius * radius;
    if (!(l2 <= r2)) { return false; }
    const float td = sqrtf((r2 - l2) / dd);
    const float tFront = projection - td;
    const float tBack = projection + td;
    *t = (tFront > tnear) ? tFront : tBack;
    return true;
}

// conservative slab test: one v_fma per plane, NaN (0 * inf) never culls
__device__ inline bool slabTest(float4 lo, float4 hi, V3 invD, V3 oInvD, float tnear, float tfar, float *tEntry)
{
    const float tx0 = fmaf(lo.x, invD.x, -oInvD.x);
    const float tx1 = fmaf(hi.x, invD.x, -oInvD.x);
    const float ty0 = fmaf(lo.y, invD.y, -oInvD.y);
    const float ty1 = fmaf(hi.y, invD.y, -oInvD.y);
    const float tz0 = fmaf(lo.z, invD.z, -oInvD.z);
    const float tz1 = fmaf(hi.z, invD.z, -oInvD.z);
    // fminf/fmaxf return the non-NaN operand, which is what a conservative test wants
    const float tmin = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), tnear));
    const float tmax = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), tfar));
    *tEntry = tmin;
    return tmin <= tmax * 1.0000004f;
}

// Scene geometry as the traversal sees it: pointers are either HBM or LDS copies.
struct TraceGeometry {
    const float4 *nodes;
    const float4 *tris;
    int nNodes;
    int nTris;
    const DSphere *spheres;
    int nSpheres;          // spheres to test AFTER the traversal (those that are not in the tree's leaves)
};

// One ray in flight on one lane.
struct LaneRay {
    V3 o, d, invD, oInvD;
    float tnear;
    float tfar;      // the query's far bound (any-hit accepts t <= tfar)
    float best;      // closest accepted t so far (starts at tfar)
    float bestU, bestV;
    int bestPrim;
    int current;     // inner node to visit next
    int pendingLeaf; // > 0: leaf waiting to be tested, (first triangle << 3) | count
    int sp;          // entries on this lane's LDS stack
    bool anyHit;
    bool occluded;
};

__device__ inline void laneRayInit(LaneRay &ray, V3 o, V3 d, float tnear, float tfar, bool anyHit)
{
    ray.o = o;
    ray.d = d;
    // 1/d clamped to a large FINITE value: with +-inf the one-fma slab form computes
    // inf - inf = NaN for axis-parallel rays.  3e30 x coordinate stays below FLT_MAX for
    // |coordinate| < 1e8 and still orders every slab plane correctly.
    const float kHuge = 3e30f;
    ray.invD = v3(
        fminf(fmaxf(1.f / d.x, -kHuge), kHuge),
        fminf(fmaxf(1.f / d.y, -kHuge), kHuge),
        fminf(fmaxf(1.f / d.z, -kHuge), kHuge));
    ray.oInvD = v3(o.x * ray.invD.x, o.y * ray.invD.y, o.z * ray.invD.z);
    ray.tnear = tnear;
    ray.tfar = tfar;
    ray.best = tfar;
    ray.bestU = 0.f;
    ray.bestV = 0.f;
    ray.bestPrim = -1;
    ray.current = 0;
    ray.pendingLeaf = 0;
    ray.sp = 0;
    ray.anyHit = anyHit;
    ray.occluded = false;
}

// Keeps the compiler from sinking a 16-byte load below a branch that may skip its use: the four
// components must be in VGPRs here, so the load is issued (and waited for) before this point.
// Without it hipcc splits node / triangle loads into "index first, box later" pairs and every
// node visit pays two dependent memory latencies instead of one.
__device__ inline void pinLoaded(const float4 &a)
{
    asm volatile("" :: "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w));
}

__device__ inline void testLeafTriangle(
    LaneRay &ray, float4 t0, float4 t1, float4 t2, bool *terminate
) {
    float t, u, v;
    if (!intersectTriangle(ray.o, ray.d, v3(t0.x, t0.y, t0.z), v3(t1.x, t1.y, t1.z), v3(t2.x, t2.y, t2.z), &t, &u, &v)) { return; }
    if (!(t > ray.tnear)) { return; }
    const int prim = floatAsInt(t0.w);
    if (ray.anyHit) {
        if (t <= ray.tfar) { ray.occluded = true; *terminate = true; }
    } else {
        const bool closer = (ray.bestPrim < 0)
            ? (t <= ray.best)
            : (t < ray.best || (t == ray.best && prim < ray.bestPrim));
        if (closer) { ray.best = t; ray.bestU = u; ray.bestV = v; ray.bestPrim = prim; }
    }
}

// Leaves are not tested where they are found: the lane records the leaf (pendingLeaf) or
// pushes it on its stack as a negative entry, and the kernel runs a TRIANGLE PHASE for the whole
// wave once enough lanes have a leaf pending.  Testing leaves inline left ~1 lane in 6 busy in
// the triangle code and made it two thirds of the instructions issued; hits are unaffected by
// the order (the acceptance rule is order-independent), only the culling bound shrinks later.
__host__ __device__ inline int encodeLeaf(int first, int count) { return (first << 3) | count; }
static const int kEmptyChild = (int)0x80000000u;

// One lane's traversal stack: ROWS entries in LDS (+ one scratch row that absorbs the writes of
// children that are not pushed), the rest in a global column.
struct LaneStack {
    int *lds;             // entry k at lds[k * STRIDE]
    int *overflow;        // entry ROWS + k at overflow[k * overflowStride]
    size_t overflowStride;
};

template <int ROWS, int STRIDE>
__device__ inline int stackRead(const LaneStack &s, int index)
{
    if (index < ROWS) { return s.lds[index * STRIDE]; }
    return s.overflow[(size_t)(index - ROWS) * s.overflowStride];
}

template <int ROWS, int STRIDE>
__device__ inline void stackWrite(const LaneStack &s, int index, int value)
{
    if (index < ROWS) { s.lds[index * STRIDE] = value; }
    else { s.overflow[(size_t)(index - ROWS) * s.overflowStride] = value; }
}

// Pops the lane's next piece of work.  Returns true when the stack is exhausted.
template <int ROWS, int STRIDE>
__device__ inline bool popWork(const LaneStack &stack, LaneRay &ray)
{
    if (ray.sp == 0) { return true; }
    ray.sp--;
    const int entry = stackRead<ROWS, STRIDE>(stack, ray.sp);
    if (entry >= 0) { ray.current = entry; ray.pendingLeaf = 0; }
    else { ray.pendingLeaf = -entry - 1; }
    return false;
}

// Cache warming for the dependent fetch chain: a 4-byte load whose destination is the lane's dword of the stack's
// scratch row in LDS (global_load_lds_dword: no VGPR result, nothing ever reads it), so the 128-byte line of a node
// or the first line of a leaf is on its way to L2 / L1 while the lane still works on something else.
template <int ROWS, int STRIDE>
__device__ inline void warmLine(const LaneStack &stack, const float4 *address)
{
    int *waveRow = stack.lds + ROWS * STRIDE - (int)(threadIdx.x & 63u);   // wave-uniform: lane 0's scratch dword
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)address,
                                     (__attribute__((address_space(3))) void *)waveRow, 4, 0, 0);
}

// Visit ONE inner node (lane must not have a leaf pending): four slab tests, the children that
// are hit sorted leaves-first then near-to-far, the first becomes the lane's next piece of work,
// the others are stacked far-to-near.  Returns true when the BVH part of the query is complete.
// `maxStack` is the tree's bound on stack entries (3 per level).
template <bool COUNT, int ROWS, int STRIDE, bool WARM = false, bool QUANT = false>
__device__ inline bool innerStep(
    const TraceGeometry &g, const LaneStack &stack, int maxStack, LaneRay &ray, TraceCounters *counters
) {
    const float4 *node = g.nodes + (QUANT ? 4 : 8) * ray.current;
    int ref[4];
    // sort key: misses last, leaves before inner nodes (they shrink `best`), then entry distance,
    // then the child slot (keys are distinct, so the order is total and the same everywhere)
    unsigned int key[4];
    int hits = 0;
    auto classify = [&](int c, float tx0, float tx1, float ty0, float ty1, float tz0, float tz1) {
        // fminf/fmaxf return the non-NaN operand, which is what a conservative test wants
        const float tmin = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), ray.tnear));
        const float tmax = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), ray.best));
        const bool valid = ref[c] != kEmptyChild;
        const bool hit = valid && (tmin <= tmax * 1.0000004f);
        if (COUNT) { counters->boxes += valid ? 1u : 0u; }
        const unsigned int inner = ref[c] >= 0 ? 0x80000000u : 0u;
        key[c] = hit ? (inner | (((unsigned int)floatAsInt(tmin) >> 1) & 0x7FFFFFFCu) | (unsigned int)c) : 0xFFFFFFFFu;
        hits += hit ? 1 : 0;
    };
    if (QUANT) {
        // all 64 bytes of the node in one round trip
        const float4 head = node[0], mid = node[1], tail = node[2];
        const float4 refBits = node[3];
        pinLoaded(head); pinLoaded(mid); pinLoaded(tail); pinLoaded(refBits);
        ref[0] = floatAsInt(refBits.x); ref[1] = floatAsInt(refBits.y); ref[2] = floatAsInt(refBits.z); ref[3] = floatAsInt(refBits.w);
        // t(plane q) = (origin + q * scale - o) / d = q * (scale / d) + (origin / d - o / d); scale / d is exact
        const float ax = head.w * ray.invD.x, ay = mid.x * ray.invD.y, az = mid.y * ray.invD.z;
        const float bx = fmaf(head.x, ray.invD.x, -ray.oInvD.x);
        const float by = fmaf(head.y, ray.invD.y, -ray.oInvD.y);
        const float bz = fmaf(head.z, ray.invD.z, -ray.oInvD.z);
        const unsigned int qlx = (unsigned int)floatAsInt(mid.z), qly = (unsigned int)floatAsInt(mid.w), qlz = (unsigned int)floatAsInt(tail.x);
        const unsigned int qhx = (unsigned int)floatAsInt(tail.y), qhy = (unsigned int)floatAsInt(tail.z), qhz = (unsigned int)floatAsInt(tail.w);
        #pragma unroll
        for (int c = 0; c < 4; c++) {
            // (x >> 8c) & 255 -> float is one v_cvt_f32_ubyte<c>
            classify(c,
                fmaf((float)((qlx >> (8 * c)) & 255u), ax, bx), fmaf((float)((qhx >> (8 * c)) & 255u), ax, bx),
                fmaf((float)((qly >> (8 * c)) & 255u), ay, by), fmaf((float)((qhy >> (8 * c)) & 255u), ay, by),
                fmaf((float)((qlz >> (8 * c)) & 255u), az, bz), fmaf((float)((qhz >> (8 * c)) & 255u), az, bz));
        }
    } else {
        // all 112 used bytes of the node in one round trip
        const float4 lox = node[0], loy = node[1], loz = node[2];
        const float4 hix = node[3], hiy = node[4], hiz = node[5];
        const float4 refBits = node[6];
        pinLoaded(lox); pinLoaded(loy); pinLoaded(loz);
        pinLoaded(hix); pinLoaded(hiy); pinLoaded(hiz);
        pinLoaded(refBits);
        const float lx[4] = { lox.x, lox.y, lox.z, lox.w }, ly[4] = { loy.x, loy.y, loy.z, loy.w }, lz[4] = { loz.x, loz.y, loz.z, loz.w };
        const float hx[4] = { hix.x, hix.y, hix.z, hix.w }, hy[4] = { hiy.x, hiy.y, hiy.z, hiy.w }, hz[4] = { hiz.x, hiz.y, hiz.z, hiz.w };
        ref[0] = floatAsInt(refBits.x); ref[1] = floatAsInt(refBits.y); ref[2] = floatAsInt(refBits.z); ref[3] = floatAsInt(refBits.w);
        #pragma unroll
        for (int c = 0; c < 4; c++) {
            classify(c,
                fmaf(lx[c], ray.invD.x, -ray.oInvD.x), fmaf(hx[c], ray.invD.x, -ray.oInvD.x),
                fmaf(ly[c], ray.invD.y, -ray.oInvD.y), fmaf(hy[c], ray.invD.y, -ray.oInvD.y),
                fmaf(lz[c], ray.invD.z, -ray.oInvD.z), fmaf(hz[c], ray.invD.z, -ray.oInvD.z));
        }
    }

    // 5-comparator sorting network on (key, ref)
    #define PATHED_SORT2(a, b) { \
        const bool swap = key[b] < key[a]; \
        const unsigned int lowKey = swap ? key[b] : key[a], highKey = swap ? key[a] : key[b]; \
        const int lowRef = swap ? ref[b] : ref[a], highRef = swap ? ref[a] : ref[b]; \
        key[a] = lowKey; key[b] = highKey; ref[a] = lowRef; ref[b] = highRef; }
    PATHED_SORT2(0, 1) PATHED_SORT2(2, 3) PATHED_SORT2(0, 2) PATHED_SORT2(1, 3) PATHED_SORT2(1, 2)
    #undef PATHED_SORT2

    if (hits == 0) { return popWork<ROWS, STRIDE>(stack, ray); }

    // ref[1 .. hits-1] go on the stack far-to-near, so the nearest of them is popped first
    const int top = ray.sp + hits - 1;   // stack size after the pushes
    if (top <= ROWS) {
        // common case, branch-free: a child that is not pushed writes the scratch row (ROWS)
        #pragma unroll
        for (int k = 1; k < 4; k++) {
            const int row = k < hits ? top - k : ROWS;
            stack.lds[row * STRIDE] = ref[k];
        }
        ray.sp = top;
    } else {
        #pragma unroll
        for (int k = 3; k >= 1; k--) {
            if (k < hits && ray.sp < maxStack) { stackWrite<ROWS, STRIDE>(stack, ray.sp, ref[k]); ray.sp++; }
        }
    }
    if (ref[0] >= 0) { ray.current = ref[0]; }
    else { ray.pendingLeaf = -ref[0] - 1; }
    if (WARM) {
        // the leaf this lane will test in the wave's next triangle phase, and the entry it will pop first
        // (a sphere leaf, count 0, has no triangle record: those lanes touch the node they already hold)
        const int leaf0 = -ref[0] - 1, leaf1 = -ref[1] - 1;
        const float4 *nearLeaf = (ref[0] < 0 && (leaf0 & 7) != 0) ? g.tris + 3 * (leaf0 >> 3) : node;
        const float4 *nextEntry = hits < 2 ? node
            : (ref[1] >= 0 ? g.nodes + (QUANT ? 4 : 8) * ref[1] : ((leaf1 & 7) != 0 ? g.tris + 3 * (leaf1 >> 3) : node));
        warmLine<ROWS, STRIDE>(stack, nearLeaf);
        warmLine<ROWS, STRIDE>(stack, nextEntry);
    }
    return false;
}

// Visit ONE 8-wide compressed node.  Same contract as innerStep; the up-to-eight hits are not sorted in registers: every
// child's RANK among the keys (28 compares) is its distance from the top of the stack, each child writes its own stack
// row (misses write the scratch row), and the nearest is popped back -- an LDS round trip instead of a 19-comparator network.
template <bool COUNT, int ROWS, int STRIDE>
__device__ inline bool innerStep8(
    const TraceGeometry &g, const LaneStack &stack, int maxStack, LaneRay &ray, TraceCounters *counters
) {
    const float4 *node = g.nodes + 8 * ray.current;
    const float4 w0 = node[0], w1 = node[1], w2 = node[2], w3 = node[3], w4 = node[4];
    const float4 refLow = node[5], refHigh = node[6];
    pinLoaded(w0); pinLoaded(w1); pinLoaded(w2); pinLoaded(w3); pinLoaded(w4);
    pinLoaded(refLow); pinLoaded(refHigh);
    const int ref[8] = { floatAsInt(refLow.x), floatAsInt(refLow.y), floatAsInt(refLow.z), floatAsInt(refLow.w),
                         floatAsInt(refHigh.x), floatAsInt(refHigh.y), floatAsInt(refHigh.z), floatAsInt(refHigh.w) };
    const float ax = w0.w * ray.invD.x, ay = w1.x * ray.invD.y, az = w1.y * ray.invD.z;
    const float bx = fmaf(w0.x, ray.invD.x, -ray.oInvD.x);
    const float by = fmaf(w0.y, ray.invD.y, -ray.oInvD.y);
    const float bz = fmaf(w0.z, ray.invD.z, -ray.oInvD.z);
    // [plane][half]: lo.x, lo.y, lo.z, hi.x, hi.y, hi.z
    const unsigned int q[6][2] = {
        { (unsigned int)floatAsInt(w1.z), (unsigned int)floatAsInt(w1.w) }, { (unsigned int)floatAsInt(w2.x), (unsigned int)floatAsInt(w2.y) },
        { (unsigned int)floatAsInt(w2.z), (unsigned int)floatAsInt(w2.w) }, { (unsigned int)floatAsInt(w3.x), (unsigned int)floatAsInt(w3.y) },
        { (unsigned int)floatAsInt(w3.z), (unsigned int)floatAsInt(w3.w) }, { (unsigned int)floatAsInt(w4.x), (unsigned int)floatAsInt(w4.y) } };
    unsigned int key[8];
    int hits = 0;
    #pragma unroll
    for (int c = 0; c < 8; c++) {
        const int half = c >> 2, shift = 8 * (c & 3);
        const float tx0 = fmaf((float)((q[0][half] >> shift) & 255u), ax, bx), tx1 = fmaf((float)((q[3][half] >> shift) & 255u), ax, bx);
        const float ty0 = fmaf((float)((q[1][half] >> shift) & 255u), ay, by), ty1 = fmaf((float)((q[4][half] >> shift) & 255u), ay, by);
        const float tz0 = fmaf((float)((q[2][half] >> shift) & 255u), az, bz), tz1 = fmaf((float)((q[5][half] >> shift) & 255u), az, bz);
        const float tmin = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), ray.tnear));
        const float tmax = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), ray.best));
        const bool valid = ref[c] != kEmptyChild;
        const bool hit = valid && (tmin <= tmax * 1.0000004f);
        if (COUNT) { counters->boxes += valid ? 1u : 0u; }
        const unsigned int inner = ref[c] >= 0 ? 0x80000000u : 0u;
        // leaves first, then entry distance, then the slot: distinct among the hits
        key[c] = hit ? (inner | (((unsigned int)floatAsInt(tmin) >> 1) & 0x7FFFFFF8u) | (unsigned int)c) : 0xFFFFFFFFu;
        hits += hit ? 1 : 0;
    }
    if (hits == 0) { return popWork<ROWS, STRIDE>(stack, ray); }

    // rank[c] = number of children whose key is below key[c]: for i < j the pair adds one to exactly one of the two
    int rank[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    #pragma unroll
    for (int i = 0; i < 8; i++) {
        #pragma unroll
        for (int j = i + 1; j < 8; j++) {
            const int below = key[i] < key[j] ? 1 : 0;   // equal only for two misses, whose rows are the scratch row anyway
            rank[j] += below;
            rank[i] += 1 - below;
        }
    }
    // all hits go on the stack, the nearest on top, and the nearest is popped back
    const int top = ray.sp + hits;
    if (top <= ROWS) {
        #pragma unroll
        for (int c = 0; c < 8; c++) {
            const int row = rank[c] < hits ? top - 1 - rank[c] : ROWS;
            stack.lds[row * STRIDE] = ref[c];
        }
        ray.sp = top;
    } else {
        #pragma unroll
        for (int c = 0; c < 8; c++) {
            const int index = top - 1 - rank[c];
            if (rank[c] < hits && index < maxStack) { stackWrite<ROWS, STRIDE>(stack, index, ref[c]); }
        }
        ray.sp = top < maxStack ? top : maxStack;
    }
    return popWork<ROWS, STRIDE>(stack, ray);
}

// One sphere against the lane's ray, same acceptance rule as the triangles (prim id = nTris + sphere index).
// Returns true when an any-hit query is decided.
__device__ inline bool testSphere(const TraceGeometry &g, LaneRay &ray, int index)
{
    const DSphere s = g.spheres[index];
    float t;
    if (!intersectSphere(ray.o, ray.d, v3(s.centerWorld[0], s.centerWorld[1], s.centerWorld[2]), s.radius, ray.tnear, &t)) { return false; }
    if (!(t > ray.tnear)) { return false; }
    const int prim = g.nTris + index;
    if (ray.anyHit) {
        if (t <= ray.tfar) { ray.occluded = true; return true; }
    } else {
        const bool closer = (ray.bestPrim < 0)
            ? (t <= ray.best)
            : (t < ray.best || (t == ray.best && prim < ray.bestPrim));
        if (closer) { ray.best = t; ray.bestU = 0.f; ray.bestV = 0.f; ray.bestPrim = prim; }
    }
    return false;
}

// Test the lane's pending leaf (<= 7 triangles, or ONE sphere: count 0, first = sphere index + 1 -- reference
// src/sphere.cpp:16-48 hands its spheres to Embree's tree the same way), then pop the next piece of work.
// Returns true when the query is complete.
// SPHERES = false: the scene has no sphere primitives, their code (and the registers it wants) is compiled out
template <bool COUNT, int ROWS, int STRIDE, bool SPHERES = true>
__device__ inline bool leafStep(const TraceGeometry &g, const LaneStack &stack, LaneRay &ray, TraceCounters *counters)
{
    const int first = ray.pendingLeaf >> 3;
    const int count = ray.pendingLeaf & 7;
    ray.pendingLeaf = 0;
    if (SPHERES && count == 0) {
        if (COUNT) { counters->tris++; }   // one primitive test
        if (testSphere(g, ray, first - 1)) { return true; }
        return popWork<ROWS, STRIDE>(stack, ray);
    }
    // the next triangle's 48 bytes are in flight while the current one is tested
    float4 t0 = g.tris[3 * first + 0];
    float4 t1 = g.tris[3 * first + 1];
    float4 t2 = g.tris[3 * first + 2];
    bool terminate = false;
    for (int k = 0; k < count; k++) {
        pinLoaded(t0);
        pinLoaded(t1);
        pinLoaded(t2);
        const float4 c0 = t0, c1 = t1, c2 = t2;
        if (k + 1 < count) {
            t0 = g.tris[3 * (first + k + 1) + 0];
            t1 = g.tris[3 * (first + k + 1) + 1];
            t2 = g.tris[3 * (first + k + 1) + 2];
        }
        if (COUNT) { counters->tris++; }
        testLeafTriangle(ray, c0, c1, c2, &terminate);
        if (terminate) { return true; }
    }
    return popWork<ROWS, STRIDE>(stack, ray);
}

// After the BVH: the spheres that are not in the tree (g.nSpheres of them: all of a tiny scene's, none when the host
// builder put them into leaves) are tested one by one, then the result is final.
template <bool SPHERES = true>
__device__ inline void finishRay(const TraceGeometry &g, LaneRay &ray)
{
    if (!SPHERES) { return; }
    if (ray.anyHit && ray.occluded) { return; }
    for (int i = 0; i < g.nSpheres; i++) {
        if (testSphere(g, ray, i)) { return; }
    }
}

// ---- proxies: "can this ray meet a set of primitives at all", from the set's bounds alone (k_path_hybrid's tree part,
// k_shade's local rays: the bounds of everything but a scene's few large triangles)
// conservative "the segment (tnear, tfar] of the ray meets the box": reciprocal by v_rcp_f32 (1 ulp), the box padded by the
// host by 1e-4 of its size, tmax by 1e-5 relative: misses only what innerStep's slab test of the root's children would miss too
__device__ __forceinline__ bool hybridProxy(const float *lo, const float *hi, V3 o, V3 d, float tfar)
{
    const float kHuge = 3e30f;
    const float ix = fminf(fmaxf(__builtin_amdgcn_rcpf(d.x), -kHuge), kHuge);
    const float iy = fminf(fmaxf(__builtin_amdgcn_rcpf(d.y), -kHuge), kHuge);
    const float iz = fminf(fmaxf(__builtin_amdgcn_rcpf(d.z), -kHuge), kHuge);
    const float tx0 = (lo[0] - o.x) * ix, tx1 = (hi[0] - o.x) * ix;
    const float ty0 = (lo[1] - o.y) * iy, ty1 = (hi[1] - o.y) * iy;
    const float tz0 = (lo[2] - o.z) * iz, tz1 = (hi[2] - o.z) * iz;
    // fminf / fmaxf return the non-NaN operand (0 x inf): what a conservative test wants
    const float tmin = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), 0.f));
    const float tmax = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), tfar));
    return tmin <= tmax * 1.00001f + 1e-6f;
}

// ... and "the ray's LINE comes within the part's bounding sphere, ahead of the origin unless that lies inside": the box's
// corners are empty space around a round cluster (the reference's ball: half the box).  sphere = centre, radius^2 (padded)
__device__ __forceinline__ bool hybridProxySphere(const float *sphere, V3 o, V3 d)
{
    const V3 c0 = v3(sphere[0] - o.x, sphere[1] - o.y, sphere[2] - o.z);
    const float cc = dot(c0, c0), cd = dot(c0, d), dd = dot(d, d);
    const float r2 = sphere[3];
    if (cc <= r2) { return true; }                       // the origin is inside
    if (cd <= 0.f) { return false; }                     // outside and heading away (a NaN falls through and keeps the ray)
    // |perpendicular|^2 |d|^2 = cc dd - cd^2 <= r^2 dd, with 1e-5 of slack for the rounding of the three products
    return !(cc * dd - cd * cd > r2 * dd + 1e-5f * cc * dd);
}

// One whole ray on one lane (test hook / simple callers).
// FORMAT: 0 the 128-byte float nodes, 1 nodeQ, 2 node8
template <bool COUNT, int ROWS, int STRIDE, int FORMAT = 0>
__device__ inline bool traverse(
    const TraceGeometry &g, const LaneStack &stack, int maxStack,
    V3 o, V3 d, float tnear, float tfar, bool anyHit,
    RayHit *hit, TraceCounters *counters
) {
    LaneRay ray;
    laneRayInit(ray, o, d, tnear, tfar, anyHit);
    if (g.nNodes > 0) {
        bool done = false;
        while (!done) {
            done = ray.pendingLeaf
                ? leafStep<COUNT, ROWS, STRIDE>(g, stack, ray, counters)
                : (FORMAT == 2 ? innerStep8<COUNT, ROWS, STRIDE>(g, stack, maxStack, ray, counters)
                               : innerStep<COUNT, ROWS, STRIDE, false, FORMAT == 1>(g, stack, maxStack, ray, counters));
        }
    }
    finishRay(g, ray);
    if (anyHit) { return ray.occluded; }
    hit->t = ray.best;
    hit->u = ray.bestU;
    hit->v = ray.bestV;
    hit->prim = ray.bestPrim;
    return ray.bestPrim >= 0;
}

}  // namespace pathed
