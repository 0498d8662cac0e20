// The records phase 1 of the all-triangles intersector reads in the fused path kernel (kernels.h: smallCandidatesItems).
//
// The scenes that take this intersector are built of QUADS (Cornell: 17 of its 18 -- the left wall is not planar --, of which 4
// are trapezoids 0.6 % off a parallelogram): CornellBox-Original.obj is 18 of them, the Veach scene's plates
// and floor are quads, `"type": "quad"` models are quads (reference src/quad.cpp:27-151, src/obj_parser.cpp's (0,1,2),(0,2,3)
// split).  Two triangles that share a diagonal and form a parallelogram are ONE Moeller-Trumbore evaluation in phase 1: the
// parallelogram c0 + alpha a1 + beta a2, alpha, beta in [0, 1], holds both (triangle A: beta <= alpha, triangle B: alpha <=
// beta), so u det, v det, det and t det are computed once for the two of them -- half the arithmetic of the pass, which is a
// third of k_path_small's instructions.  Phase 1 only has to be CONSERVATIVE (phase 2, trace.h: intersectTriangle +
// testLeafTriangle, decides): but here its quantities are no longer the bits phase 2 computes for the triangle (another
// origin, other edges), so every bound carries a tolerance that covers both evaluations' rounding and the distance delta
// between the triangles' corners (as the intersector sees them: v0, v0 + e1, v0 + e2) and the parallelogram's.
//
// With the three edge functions of a triangle E(P, Q) = d . ((P - o) x (Q - o)) (which is what u det, v det and
// (det - u - v) det are, whatever vertex order), |d| <= 1.0001, r = |o - c0| + amax the distance from the ray's origin to any
// corner (amax = the longest of a1, a2, a1 + a2), u = 2^-24:
//     |rounding of either evaluation| <= 8 u r amax per edge function,   |moving a corner by delta| <= 2 delta r
//  => E_uv = (32 u amax + 4 delta) r,   E_t = (32 u |a1||a2| + 8 delta amax) r   (t det = (c0 - o) . (a1 x a2)),
//     E_det = 16 u |a1||a2| + 4 delta amax.
// The errors scale with the distance of the ORIGIN, not with the scene: a ray leaving a quad sees that quad (t = 0, to be
// told from t >= tnear / 2) with the error of a nearby origin, however far away the camera is.  A bound "X sign(det) >= -E" is
// tested as X det >= -E |det| with E |det| <= kappa det^2 + E^2 / (4 kappa) and r^2 <= 2 (|o - c0|^2 + amax^2), so the kernel
// compares against   tol = kappa det^2 + K2 (|o - c0|^2 + amax^2),   K2 = (E / r)^2 / (2 kappa),
// kappa = 1e-5 for the barycentric bounds (a relative slack of 1e-5), kappa_t = tnear / 16 for the t bounds (an absolute
// slack in t, well inside the tnear / 2 by which the interval is widened anyway).  Where |det| <= E_det the computed sign of
// det cannot be trusted; every bound is then at most Xmax E_det in size, which the K terms include: nothing is rejected there.  The far bound's error grows with the ray's own tfar:
// tolFar = tolT + tfarHigh (kappa_far det^2 + E_det^2 / (4 kappa_far)), kappa_far = 1e-6 (a light 20 units away is told from
// an occluder 1e-3 in front of it with 2e-5 of slack).  Triangles that find no partner keep the exact test.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace pathed {

static const int kSmallQuadWords = 16;        // float2 per packed PAIR of parallelograms: c0.xyz, a1.xyz, a2.xyz, K2uv, amax^2, K2t, -, cD, Edet, kappaUV
static const int kSmallLoneWords = 9;         // float2 per packed pair of lone triangles: v0.xyz, e1.xyz, e2.xyz (kSmallPairWords)
static const int kSmallItemFloats = 2 * 9 * 32;   // the kernarg array of SmallTris, in floats
static const float kSmallKappa = 1e-5f;       // relative slack of the barycentric bounds
static const float kSmallKappaFar = 1e-6f;    // relative slack of a shadow ray's far bound (its tolerance is tfarHigh x (kappaFar det^2 + cD))

struct SmallItemsLayout {
    int nQuads = 0;          // parallelograms: item-order triangles 2 q (A: beta <= alpha) and 2 q + 1 (B)
    int nLone = 0;           // triangles without a partner: item-order 2 nQuads + k
    float kappaT = 0.f;      // absolute slack of the t bounds per det^2 (a length): a sixteenth of tnear
};

// leafTris: 12 floats per triangle, (v0, prim) (e1, -) (e2, -).  extraPoints: origins rays may have besides the triangles
// themselves (the camera; sphere bounds).  records: kSmallItemFloats floats, [pair of parallelograms][component][half], then
// the lone pairs as SmallTris stores them.  itemTris: the 12-float records again, in ITEM order (what phase 2 indexes).
inline SmallItemsLayout buildSmallItems(const float *leafTris, int nTris, const float *extraPoints, int nExtra, bool pairQuads, float tnear,
                                        float *records, std::vector<float> *itemTris)
{
    SmallItemsLayout layout;
    std::memset(records, 0, sizeof(float) * kSmallItemFloats);
    itemTris->assign((size_t)12 * std::max(nTris, 1), 0.f);
    struct Corners { double p[3][3]; };
    std::vector<Corners> corners((size_t)nTris);
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    auto cover = [&](const double *point) {
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], point[a]); hi[a] = std::max(hi[a], point[a]); }
    };
    for (int k = 0; k < nTris; k++) {
        const float *tri = leafTris + (size_t)12 * k;
        for (int a = 0; a < 3; a++) {
            corners[k].p[0][a] = tri[a];
            corners[k].p[1][a] = (double)tri[a] + (double)tri[4 + a];   // the triangle the intersector sees: v0, v0 + e1, v0 + e2
            corners[k].p[2][a] = (double)tri[a] + (double)tri[8 + a];
        }
        for (int c = 0; c < 3; c++) { cover(corners[k].p[c]); }
    }
    for (int i = 0; i < nExtra; i++) {
        const double point[3] = { extraPoints[3 * i], extraPoints[3 * i + 1], extraPoints[3 * i + 2] };
        cover(point);
    }
    double diameter = 0.0;
    for (int a = 0; a < 3; a++) { if (hi[a] > lo[a]) { diameter += (hi[a] - lo[a]) * (hi[a] - lo[a]); } }
    diameter = std::sqrt(diameter) * 1.001;
    layout.kappaT = tnear / 16.f;

    auto distance = [](const double *a, const double *b) {
        return std::sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
    };
    struct Quad { int a, b; double c0[3], a1[3], a2[3], delta, excess, amax; };
    std::vector<Quad> quads;
    std::vector<int> partner((size_t)nTris, -1);
    const double unit = 1.0 / 16777216.0;
    for (int a = 0; pairQuads && a < nTris; a++) {
        if (partner[a] >= 0) { continue; }
        for (int b = a + 1; b < nTris; b++) {
            if (partner[b] >= 0) { continue; }
            // two shared corners (the diagonal), within a few ulps of the scene's size
            const double same = 8.0 * unit * diameter;
            int sharedA[2], sharedB[2], nShared = 0;
            bool usedB[3] = { false, false, false };
            for (int i = 0; i < 3 && nShared <= 2; i++) {
                for (int j = 0; j < 3; j++) {
                    if (!usedB[j] && distance(corners[a].p[i], corners[b].p[j]) <= same) {
                        if (nShared < 2) { sharedA[nShared] = i; sharedB[nShared] = j; }
                        nShared++;
                        usedB[j] = true;
                        break;
                    }
                }
            }
            if (nShared != 2) { continue; }
            const int restA = 3 - sharedA[0] - sharedA[1], restB = 3 - sharedB[0] - sharedB[1];
            const double *rA = corners[a].p[restA], *rB = corners[b].p[restB];
            // Either end of the diagonal may be the origin c0.  With the diagonal's other end at c0 + alpha2 (rA - c0) + beta2
            // (rB - c0), the edges a1 = alpha2 (rA - c0), a2 = beta2 (rB - c0) put it at (1, 1): triangle A = (c0, rA, far end) is
            // the part beta <= alpha of the quad, B the part alpha <= beta, and the quad lies in [0, max(1, 1 / alpha2)] x
            // [0, max(1, 1 / beta2)].  A parallelogram has alpha2 = beta2 = 1; a planar quad that is nearly one (the walls of the
            // Cornell box are trapezoids, 0.6 % off) sticks out of the unit square by `excess`, which joins the relative slack of
            // the box test.  Take the origin with the smaller excess; quads further off than 5 % are left alone.
            Quad quad;
            quad.a = a; quad.b = b;
            double bestExcess = 1e300, delta = 0.0, amax = 0.0;
            for (int end = 0; end < 2; end++) {
                const double *d0 = corners[a].p[sharedA[end]], *d1 = corners[a].p[sharedA[1 - end]];
                const double *d0b = corners[b].p[sharedB[end]], *d1b = corners[b].p[sharedB[1 - end]];
                double e1[3], e2[3], g[3];
                for (int x = 0; x < 3; x++) { e1[x] = rA[x] - d0[x]; e2[x] = rB[x] - d0[x]; g[x] = d1[x] - d0[x]; }
                // least squares g = alpha2 e1 + beta2 e2
                const double m11 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2], m22 = e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2];
                const double m12 = e1[0] * e2[0] + e1[1] * e2[1] + e1[2] * e2[2];
                const double r1 = e1[0] * g[0] + e1[1] * g[1] + e1[2] * g[2], r2 = e2[0] * g[0] + e2[1] * g[1] + e2[2] * g[2];
                const double determinant = m11 * m22 - m12 * m12;
                if (!(determinant > 1e-12 * m11 * m22)) { continue; }   // rA, c0, rB on a line
                const double alpha2 = (r1 * m22 - r2 * m12) / determinant, beta2 = (r2 * m11 - r1 * m12) / determinant;
                if (!(alpha2 > 0.5 && beta2 > 0.5)) { continue; }
                const double excess = std::max(std::max(1.0 / alpha2, 1.0 / beta2), 1.0) - 1.0;
                if (!(excess < bestExcess)) { continue; }
                Quad candidate = quad;
                for (int x = 0; x < 3; x++) {
                    candidate.c0[x] = (float)d0[x];
                    candidate.a1[x] = (float)(alpha2 * e1[x]);
                    candidate.a2[x] = (float)(beta2 * e2[x]);
                }
                // where the stored (float) parallelogram puts the six corners of the two triangles, against where they are
                double worst = 0.0, longest = 0.0;
                auto place = [&](double alpha, double beta, double *out) {
                    for (int x = 0; x < 3; x++) { out[x] = candidate.c0[x] + alpha * candidate.a1[x] + beta * candidate.a2[x]; }
                };
                double q[3];
                place(0.0, 0.0, q); worst = std::max(worst, std::max(distance(d0, q), distance(d0b, q)));
                place(1.0, 1.0, q); worst = std::max(worst, std::max(distance(d1, q), distance(d1b, q)));
                place(1.0 / alpha2, 0.0, q); worst = std::max(worst, distance(rA, q));
                place(0.0, 1.0 / beta2, q); worst = std::max(worst, distance(rB, q));
                const double zero[3] = { 0.0, 0.0, 0.0 };
                longest = std::max(std::max(distance(candidate.a1, zero), distance(candidate.a2, zero)), std::max(distance(d1, d0), distance(rA, rB)));
                candidate.excess = excess;
                quad = candidate;
                bestExcess = excess;
                delta = worst;
                amax = longest;
            }
            if (!(bestExcess <= 0.05)) { continue; }
            // planar and where the parallelogram says, up to rounding
            if (!(delta <= 64.0 * unit * std::max(amax, 1e-30))) { continue; }
            // ... and small enough for the interval test to mean something: a ray that leaves the quad itself sees it at t = 0
            // +- the t tolerance, which for an origin on the quad (|o - c0| <= amax) at cos(theta) = 1/4 is
            // 1.5 (32 u)^2 4 amax^2 / (2 kappa_t) x 16; beyond tnear / 4 the quad would be a candidate of most rays that leave it
            {
                const double kappaT = tnear / 16.0;
                const double selfTolerance = 1.5 * (32.0 * unit) * (32.0 * unit) * 4.0 * amax * amax / (2.0 * kappaT) * 16.0;
                if (!(selfTolerance <= tnear / 4.0)) { continue; }
            }
            quad.amax = amax;
            quad.delta = delta;
            partner[a] = b;
            partner[b] = a;
            quads.push_back(quad);
            break;
        }
    }
    // the kernarg array holds 32 lone pairs; a pair of parallelograms takes 12 of its float2 where two lone pairs take 18, but an
    // odd count leaves half a packed pair unused: drop parallelograms until everything fits (only scenes of 62+ triangles can need it)
    auto floatsNeeded = [&](size_t nQuads) {
        return (size_t)2 * kSmallQuadWords * ((nQuads + 1) / 2) + (size_t)2 * kSmallLoneWords * (((size_t)nTris - 2 * nQuads + 1) / 2);
    };
    while (!quads.empty() && floatsNeeded(quads.size()) > (size_t)kSmallItemFloats) {
        partner[quads.back().a] = -1;
        partner[quads.back().b] = -1;
        quads.pop_back();
    }
    layout.nQuads = (int)quads.size();
    layout.nLone = nTris - 2 * layout.nQuads;

    // ---- records
    const double kappa = kSmallKappa, kappaT = layout.kappaT;
    for (int q = 0; q < layout.nQuads; q++) {
        const Quad &quad = quads[q];
        float *record = records + (size_t)2 * kSmallQuadWords * (q / 2);   // kSmallQuadWords float2 per packed pair
        const int half = q & 1;
        const double zero[3] = { 0.0, 0.0, 0.0 };
        const double amax = quad.amax;
        // errors per unit of r = |o - c0| + amax (header)
        const double eUV = 1.01 * (32.0 * unit * amax + 4.0 * quad.delta);
        // (t det and det are products of BOTH edges: |a1||a2| bounds them, which for a long thin plate is far below amax^2)
        const double area = distance(quad.a1, zero) * distance(quad.a2, zero);
        const double eT = 1.01 * (32.0 * unit * area + 8.0 * quad.delta * amax);
        const double eDet = 1.01 * (16.0 * unit * area + 4.0 * quad.delta * amax);
        // ... plus what covers a det too small to trust its sign (|det| <= E_det): there every bound X det is at most
        // Xmax E_det in size, Xmax = |o - c0| amax (u, v) or |o - c0| |a1||a2| (t), and |o - c0| <= (|o - c0|^2 / amax + amax) / 2:
        // a second pair of (K2, K0) terms, folded into the first -- no test of |det| in the kernel
        const double k2UV = 1.5 * (eUV * eUV / (2.0 * kappa) + eDet / 2.0);
        const double k2T = 1.5 * (eT * eT / (2.0 * std::max(kappaT, 1e-300)) + eDet * area / (2.0 * std::max(amax, 1e-300)));
        const double cD = 1.5 * eDet * eDet / (4.0 * kSmallKappaFar);
        for (int x = 0; x < 3; x++) {
            record[2 * (0 + x) + half] = (float)quad.c0[x];
            record[2 * (3 + x) + half] = (float)quad.a1[x];
            record[2 * (6 + x) + half] = (float)quad.a2[x];
        }
        record[2 * 9 + half] = (float)std::max(k2UV, 1e-37);
        record[2 * 10 + half] = (float)std::max(amax * amax * 1.000001, 1e-37);   // K0 = K2 amax^2: the kernel multiplies K2 by (|o - c0|^2 + amax^2)
        record[2 * 11 + half] = (float)std::max(k2T, 1e-37);
        record[2 * 13 + half] = (float)std::max(cD, 1e-37);
        record[2 * 14 + half] = (float)std::max(eDet, 1e-37);
        record[2 * 15 + half] = (float)(kappa + 1.0001 * quad.excess);   // relative slack of the box test: kappa + how far the quad sticks out
        std::memcpy(itemTris->data() + (size_t)12 * (2 * q), leafTris + (size_t)12 * quad.a, 12 * sizeof(float));
        std::memcpy(itemTris->data() + (size_t)12 * (2 * q + 1), leafTris + (size_t)12 * quad.b, 12 * sizeof(float));
    }
    float *lone = records + (size_t)2 * kSmallQuadWords * ((layout.nQuads + 1) / 2);
    int k = 0;
    for (int t = 0; t < nTris; t++) {
        if (partner[t] >= 0) { continue; }
        const float *tri = leafTris + (size_t)12 * t;
        float *record = lone + (size_t)2 * kSmallLoneWords * (k / 2);
        for (int row = 0; row < 3; row++) {
            for (int axis = 0; axis < 3; axis++) { record[2 * (3 * row + axis) + (k & 1)] = tri[4 * row + axis]; }
        }
        std::memcpy(itemTris->data() + (size_t)12 * (2 * layout.nQuads + k), tri, 12 * sizeof(float));
        k++;
    }
    return layout;
}

}  // namespace pathed
