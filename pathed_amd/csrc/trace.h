// BVH2 traversal + primitive intersection for CDNA4 (the stand-in for Embree's
// rtcIntersect1 / rtcOccluded1, reference src/scene.cpp:113, :374).
//
// Layout (DESIGN.md "Data layout"):
//   node   4 x float4 = 64 B: (lmin.xyz, left) (lmax.xyz, lcount) (rmin.xyz, right) (rmax.xyz, rcount)
//          count > 0: leaf, index = first leaf triangle; count == 0: inner node index;
//          count < 0: empty slot
//   tri    3 x float4 = 48 B: (v0.xyz, prim) (e1.xyz, -) (e2.xyz, -), leaf order
// Per-lane traversal stack lives in LDS, laid out [depth][thread] so a wave's 64 lanes
// hit 64 consecutive dwords (bank-conflict-free ds_read_b32 / ds_write_b32).
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

__device__ inline bool intersectTriangle(V3 o, V3 d, V3 v0, V3 e1, V3 e2, float *t, float *u, float *v)
{
    const V3 pvec = xcross(d, e2);
    const float det = xdot(e1, pvec);
    if (det == 0.f) { return false; }
    const float inv = 1.f / det;
    const V3 tvec = o - v0;
    const float uu = xdot(tvec, pvec) * inv;
    if (!(uu >= 0.f && uu <= 1.f)) { return false; }
    const V3 qvec = xcross(tvec, e1);
    const float vv = xdot(d, qvec) * inv;
    if (!(vv >= 0.f && uu + vv <= 1.f)) { return false; }
    *t = xdot(e2, qvec) * inv;
    *u = uu;
    *v = vv;
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
    int nSpheres;
};

// One ray.  ANY_HIT: stop at the first accepted hit in (tnear, tfar].
// `stack` points at this lane's column: entry k is stack[k * STRIDE].
template <bool ANY_HIT, bool COUNT, int STRIDE>
__device__ inline bool traverse(
    const TraceGeometry &g, int *stack, int stackDepth,
    V3 o, V3 d, float tnear, float tfar,
    RayHit *hit, TraceCounters *counters
) {
    float best = tfar;
    int bestPrim = -1;
    float bestU = 0.f, bestV = 0.f;

    // 1/d clamped to a large FINITE value: with +-inf the one-fma slab form computes
    // inf - inf = NaN for axis-parallel rays.  3e30 x coordinate stays below FLT_MAX for
    // |coordinate| < 1e8 and still orders every slab plane correctly.
    const float kHuge = 3e30f;
    const V3 invD = v3(
        fminf(fmaxf(1.f / d.x, -kHuge), kHuge),
        fminf(fmaxf(1.f / d.y, -kHuge), kHuge),
        fminf(fmaxf(1.f / d.z, -kHuge), kHuge));
    const V3 oInvD = v3(o.x * invD.x, o.y * invD.y, o.z * invD.z);

    if (g.nNodes > 0) {
        int sp = 0;
        int current = 0;
        while (true) {
            const float4 n0 = g.nodes[4 * current + 0];
            const float4 n1 = g.nodes[4 * current + 1];
            const float4 n2 = g.nodes[4 * current + 2];
            const float4 n3 = g.nodes[4 * current + 3];
            const int leftIndex = floatAsInt(n0.w), leftCount = floatAsInt(n1.w);
            const int rightIndex = floatAsInt(n2.w), rightCount = floatAsInt(n3.w);

            float tLeft, tRight;
            const bool hitLeft = (leftCount >= 0) && slabTest(n0, n1, invD, oInvD, tnear, best, &tLeft);
            const bool hitRight = (rightCount >= 0) && slabTest(n2, n3, invD, oInvD, tnear, best, &tRight);
            if (COUNT) { counters->boxes += (leftCount >= 0) + (rightCount >= 0); }

            // leaves of this node first: they can only shrink `best`
            #pragma unroll
            for (int side = 0; side < 2; side++) {
                const bool isLeaf = side == 0 ? (hitLeft && leftCount > 0) : (hitRight && rightCount > 0);
                if (!isLeaf) { continue; }
                const int first = side == 0 ? leftIndex : rightIndex;
                const int count = side == 0 ? leftCount : rightCount;
                for (int k = 0; k < count; k++) {
                    const float4 t0 = g.tris[3 * (first + k) + 0];
                    const float4 t1 = g.tris[3 * (first + k) + 1];
                    const float4 t2 = g.tris[3 * (first + k) + 2];
                    if (COUNT) { counters->tris++; }
                    float t, u, v;
                    if (!intersectTriangle(o, d, v3(t0.x, t0.y, t0.z), v3(t1.x, t1.y, t1.z), v3(t2.x, t2.y, t2.z), &t, &u, &v)) { continue; }
                    if (!(t > tnear)) { continue; }
                    const int prim = floatAsInt(t0.w);
                    if (ANY_HIT) {
                        if (t <= tfar) { hit->t = t; hit->prim = prim; return true; }
                    } else {
                        const bool closer = (bestPrim < 0) ? (t <= best) : (t < best || (t == best && prim < bestPrim));
                        if (closer) { best = t; bestU = u; bestV = v; bestPrim = prim; }
                    }
                }
            }

            const bool goLeft = hitLeft && leftCount == 0;
            const bool goRight = hitRight && rightCount == 0;
            if (goLeft && goRight) {
                const bool leftFirst = tLeft <= tRight;
                const int nearNode = leftFirst ? leftIndex : rightIndex;
                const int farNode = leftFirst ? rightIndex : leftIndex;
                if (sp < stackDepth) { stack[sp * STRIDE] = farNode; sp++; }
                current = nearNode;
            } else if (goLeft) {
                current = leftIndex;
            } else if (goRight) {
                current = rightIndex;
            } else {
                if (sp == 0) { break; }
                sp--;
                current = stack[sp * STRIDE];
            }
        }
    }

    for (int i = 0; i < g.nSpheres; i++) {
        const DSphere s = g.spheres[i];
        float t;
        if (!intersectSphere(o, d, v3(s.centerWorld[0], s.centerWorld[1], s.centerWorld[2]), s.radius, tnear, &t)) { continue; }
        if (!(t > tnear)) { continue; }
        const int prim = g.nTris + i;
        if (ANY_HIT) {
            if (t <= tfar) { hit->t = t; hit->prim = prim; return true; }
        } else {
            const bool closer = (bestPrim < 0) ? (t <= best) : (t < best || (t == best && prim < bestPrim));
            if (closer) { best = t; bestU = 0.f; bestV = 0.f; bestPrim = prim; }
        }
    }

    if (ANY_HIT) { return false; }
    hit->t = best;
    hit->u = bestU;
    hit->v = bestV;
    hit->prim = bestPrim;
    return bestPrim >= 0;
}

}  // namespace pathed
