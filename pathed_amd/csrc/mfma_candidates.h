// Phase 1 of the all-triangles intersector on the MATRIX pipe (v_mfma_f32_32x32x2_f32), for the fused path kernel.
//
// kernels.h: smallCandidates evaluates Moeller-Trumbore's u det, v det, (det - u - v) det, t det for every (ray, triangle)
// on the VALU -- a third of k_path_small's instructions -- only to learn which one or two triangles of the scene phase 2
// (trace.h: intersectTriangle + testLeafTriangle) has to look at.  Those quantities are LINEAR in the ray's Pluecker
// coordinates: with p = o - centre, q = p x d, a = v0 - centre,
//     U  = (o - v0) . (d x e2)  =  q . e2        + d . (a x e2)
//     V  = d . ((o - v0) x e1)  =  q . (-e1)     + d . (e1 x a)
//     det = e1 . (d x e2)       =                  d . (e2 x e1)
//     W  = det - U - V          =  q . (e1 - e2) + d . (e2 x e1 - a x e2 - e1 x a)
//     T  = e2 . ((o - v0) x e1) =  p . (e1 x e2) - a . (e1 x e2)             (t = T / det)
// so "all rays of the wave against all triangles" is a dense (4 rows per triangle) x (6 per ray) contraction plus a
// (1 row) x (4 per origin) one: rows on the A side (a table built in double on the host, staged in LDS), rays on the B
// side (one v_permlane32_swap per K-step turns per-lane ray components into both 32-ray operands), results in the
// accumulator layout where the four rows of a triangle land in ONE lane.  The matrix pipe runs beside the other waves'
// VALU work; what is left on the VALU is the decision per (ray, triangle): 9-12 plain instructions instead of ~27 packed.
//
// The decision only has to be CONSERVATIVE (phase 2 decides, hits stay those of the BVH path bit for bit).  Every row is
// scaled on the host so that ONE tolerance covers both evaluations' rounding (derivation: DESIGN.md, "matrix-pipe phase 1";
// tried first in tools/mfma_candidates_proto.py):
//     U' = U / ((R + |a|) |e2|)          V' = V / ((R + |a|) |e1|)
//     W' = W / ((R + |a|)(|e1| + |e2|) + |e1||e2|)
//     det' = det / s_t, T' = T / s_t,    s_t = (R + |a| + 1) |e1||e2|
// with R >= |o - centre| for every ray origin (vertices and the camera) and |d| <= 1.00005.  The MFMA is a k-ordered fmaf
// chain (cdna_hip_programming.md, "FP32-input MFMA"), so per row |computed - exact| <= 11 u (u = 2^-24: p, q, table entry,
// six chained fmas) and phase 2's own u det, v det, u det + v det <= det differ from the exact forms by <= 8 u in the same
// units:  phase 2 accepts  =>  min(U', V', W') >= -19 u  or  max(U', V', W') <= 19 u;  kGamma = 24 u.
// Interval: phase 2 accepts only t > tnear and (any-hit) t <= tfar; in det units with sigma = sign(det')
//     sigma (T' - tnear/2 det') >= -17.2 u,        sigma (tfarHigh det' - T') >= -(10 tfar + 15.1) u
// hold for every accepted hit whose det' sign is the computed one; below |det'| <= 10 u the sign is not trusted and the
// interval tests do not reject.  A ray outside the stated bounds (|d|, |p|, NaN) keeps every triangle.
#pragma once

#include "trace.h"

#include <cmath>
#include <vector>

namespace pathed {

typedef float f16v __attribute__((ext_vector_type(16)));

static const int kMfmaMaxTris = 64;
static const int kMfmaTileTris = 8;                                    // 4 rows per triangle, 32 rows per tile
static const int kMfmaMaxTiles = kMfmaMaxTris / kMfmaTileTris;         // 8
static const int kMfmaEdgeTileFloats = 3 * 64;                         // K = 6: three K-steps of one float per lane
static const int kMfmaEdgeFloats = kMfmaMaxTiles * kMfmaEdgeTileFloats;
static const int kMfmaTimeGroupFloats = 2 * 64;                        // K = 4 (p, 1): two K-steps; a group = 32 triangles
static const int kMfmaTableFloats = kMfmaEdgeFloats + 2 * kMfmaTimeGroupFloats;   // 1 792 floats = 7 KiB

static const float kMfmaUnit = 1.f / 16777216.f;         // 2^-24
static const float kMfmaGamma = 24.f * kMfmaUnit;        // edge rows
static const float kMfmaGammaNear = 20.f * kMfmaUnit;    // near test
static const float kMfmaTinyDet = 10.f * kMfmaUnit;      // |det'| below which its sign is not trusted

struct MfmaFrame {
    float centre[3];
    float radiusSquaredLimit;   // origins with |o - centre|^2 above this keep every triangle
};

// The A-side table.  leafTris: 12 floats per triangle, (v0, prim) (e1, -) (e2, -) as the kernels read them.
// originsToCover: points ray origins may lie at besides the triangles themselves (the camera).
inline void buildMfmaTable(const float *leafTris, int nTris, const float *originsToCover, int nOrigins, std::vector<float> *table, MfmaFrame *frame)
{
    table->assign(kMfmaTableFloats, 0.f);
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    auto cover = [&](const double *point) {
        for (int axis = 0; axis < 3; axis++) { lo[axis] = std::fmin(lo[axis], point[axis]); hi[axis] = std::fmax(hi[axis], point[axis]); }
    };
    auto corners = [&](int k, double out[3][3]) {
        const float *tri = leafTris + (size_t)12 * k;
        for (int axis = 0; axis < 3; axis++) {
            out[0][axis] = tri[axis];
            out[1][axis] = (double)tri[axis] + (double)tri[4 + axis];
            out[2][axis] = (double)tri[axis] + (double)tri[8 + axis];
        }
    };
    for (int k = 0; k < nTris; k++) {
        double c[3][3];
        corners(k, c);
        for (int i = 0; i < 3; i++) { cover(c[i]); }
    }
    for (int i = 0; i < nOrigins; i++) {
        const double point[3] = { originsToCover[3 * i], originsToCover[3 * i + 1], originsToCover[3 * i + 2] };
        cover(point);
    }
    double centre[3];
    for (int axis = 0; axis < 3; axis++) {
        frame->centre[axis] = nTris > 0 || nOrigins > 0 ? (float)(0.5 * (lo[axis] + hi[axis])) : 0.f;
        centre[axis] = frame->centre[axis];   // the kernel subtracts the float
    }
    double radius = 0.0;
    auto reach = [&](const double *point) {
        const double dx = point[0] - centre[0], dy = point[1] - centre[1], dz = point[2] - centre[2];
        radius = std::fmax(radius, std::sqrt(dx * dx + dy * dy + dz * dz));
    };
    for (int k = 0; k < nTris; k++) {
        double c[3][3];
        corners(k, c);
        for (int i = 0; i < 3; i++) { reach(c[i]); }
    }
    for (int i = 0; i < nOrigins; i++) {
        const double point[3] = { originsToCover[3 * i], originsToCover[3 * i + 1], originsToCover[3 * i + 2] };
        reach(point);
    }
    radius *= 1.001;   // hit points are rounded, not exact points of their triangles
    frame->radiusSquaredLimit = (float)(radius * radius * 0.9995);   // the lane's check is in float: stay inside R

    auto cross = [](const double *a, const double *b, double *out) {
        out[0] = a[1] * b[2] - a[2] * b[1];
        out[1] = a[2] * b[0] - a[0] * b[2];
        out[2] = a[0] * b[1] - a[1] * b[0];
    };
    auto norm = [](const double *a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); };
    const double tiny = 1e-300;
    for (int k = 0; k < nTris && k < kMfmaMaxTris; k++) {
        const float *tri = leafTris + (size_t)12 * k;
        double a[3], e1[3], e2[3];
        for (int axis = 0; axis < 3; axis++) { a[axis] = (double)tri[axis] - centre[axis]; e1[axis] = tri[4 + axis]; e2[axis] = tri[8 + axis]; }
        const double na = norm(a), n1 = norm(e1), n2 = norm(e2);
        const double sU = std::fmax((radius + na) * n2, tiny);
        const double sV = std::fmax((radius + na) * n1, tiny);
        const double sW = std::fmax((radius + na) * (n1 + n2) + n1 * n2, tiny);
        const double sT = std::fmax((radius + na + 1.0) * n1 * n2, tiny);
        double mU[3], mV[3], mD[3], n[3];
        cross(a, e2, mU);
        cross(e1, a, mV);
        cross(e2, e1, mD);
        cross(e1, e2, n);
        // rows on (d.xyz, q.xyz)
        double rows[4][6];
        for (int axis = 0; axis < 3; axis++) {
            rows[0][axis] = mU[axis] / sU;                               rows[0][3 + axis] = e2[axis] / sU;
            rows[1][axis] = mV[axis] / sV;                               rows[1][3 + axis] = -e1[axis] / sV;
            rows[2][axis] = (mD[axis] - mU[axis] - mV[axis]) / sW;       rows[2][3 + axis] = (e1[axis] - e2[axis]) / sW;
            rows[3][axis] = mD[axis] / sT;                               rows[3][3 + axis] = 0.0;
        }
        // A operand of v_mfma_f32_32x32x2_f32: lane l holds A[row l & 31][k = l >> 5] of the K-step
        const int tile = k / kMfmaTileTris, local = k % kMfmaTileTris;
        for (int function = 0; function < 4; function++) {
            const int row = 4 * local + function;
            for (int component = 0; component < 6; component++) {
                const int step = component >> 1, half = component & 1;
                (*table)[(size_t)tile * kMfmaEdgeTileFloats + step * 64 + half * 32 + row] = (float)rows[function][component];
            }
        }
        // the T row, on (p.xyz, 1): in tile row 8 j + 4 h + i for triangle 32 g + 8 j + 2 i + h, so that accumulator
        // register 4 j + i of lane (column, h) belongs to the triangle whose edge rows that lane holds in edge tile 4 g + j
        const int group = k / 32, within = k % 32;
        const int j = within >> 3, i = (within >> 1) & 3, h = within & 1;
        const int row = 8 * j + 4 * h + i;
        const double tRow[4] = { n[0] / sT, n[1] / sT, n[2] / sT, -(a[0] * n[0] + a[1] * n[1] + a[2] * n[2]) / sT };
        for (int component = 0; component < 4; component++) {
            const int step = component >> 1, half = component & 1;
            (*table)[(size_t)kMfmaEdgeFloats + (size_t)group * kMfmaTimeGroupFloats + step * 64 + half * 32 + row] = (float)tRow[component];
        }
    }
}

#ifdef __HIPCC__

// lanes 32-63 of `low` swap with lanes 0-31 of `high`: from (component k0, component k1) of the 64 rays to the B operands
// of ray block 0 (rays 0-31) and ray block 1 (rays 32-63) of one K-step -- B[k = lane >> 5][column = lane & 31]
__device__ __forceinline__ void mfmaSplitBlocks(float k0, float k1, float *block0, float *block1)
{
    const auto swapped = __builtin_amdgcn_permlane32_swap(__float_as_uint(k0), __float_as_uint(k1), false, false);
    *block0 = __uint_as_float(swapped[0]);
    *block1 = __uint_as_float(swapped[1]);
}

// The decision for the four triangles a lane holds of one edge tile: bit j = keep triangle j of the lane.
template <bool FAR>
__device__ __forceinline__ unsigned int mfmaKeepBits(const f16v &edge, float t0, float t1, float t2, float t3, float tnearLow, float tfarHigh, float negGammaFar)
{
    unsigned int bits = 0u;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float U = edge[4 * j + 0], V = edge[4 * j + 1], W = edge[4 * j + 2], D = edge[4 * j + 3];
        const float T = j == 0 ? t0 : j == 1 ? t1 : j == 2 ? t2 : t3;
        const float lowest = fminf(fminf(U, V), W);
        const float highest = fmaxf(fmaxf(U, V), W);
        const bool inside = (lowest >= -kMfmaGamma) || (highest <= kMfmaGamma);
        const unsigned int sign = __float_as_uint(D) & 0x80000000u;
        const float x = fmaf(-tnearLow, D, T);
        bool interval = __uint_as_float(__float_as_uint(x) ^ sign) >= -kMfmaGammaNear;
        if (FAR) {
            const float w = fmaf(tfarHigh, D, -T);
            interval = interval && (__uint_as_float(__float_as_uint(w) ^ sign) >= negGammaFar);
        }
        interval = interval || (fabsf(D) <= kMfmaTinyDet);
        bits |= (inside && interval) ? (1u << j) : 0u;
    }
    return bits;
}

// Phase 1 for the wave's 64 continuation rays (A) and, when `anyShadow`, its 64 shadow rays (B), which leave the same
// origins.  Wave-uniform control flow; every lane takes part whether or not it carries a ray (its masks are the
// caller's to ignore).  Results: bit b of evenX = keep triangle 2 b, bit b of oddX = keep triangle 2 b + 1 (leaf order).
__device__ __forceinline__ void mfmaCandidatesPair(const float *table, int nTris, const MfmaFrame &frame, bool anyShadow,
                                                   V3 origin, V3 directionA, V3 directionB, float tnearLow, float tfarHighB,
                                                   unsigned int *evenA, unsigned int *oddA, unsigned int *evenB, unsigned int *oddB)
{
    const int lane = threadIdx.x & 63;
    const V3 p = v3(origin.x - frame.centre[0], origin.y - frame.centre[1], origin.z - frame.centre[2]);
    const V3 qA = xcross(p, directionA);
    const V3 qB = xcross(p, directionB);

    const float tfarClamped = fminf(tfarHighB, 1e30f);
    const int nTiles = (nTris + kMfmaTileTris - 1) / kMfmaTileTris;
    const int nGroups = (nTiles + 3) >> 2;
    unsigned int maskA[2] = { 0u, 0u }, maskB[2] = { 0u, 0u };
    f16v zero;
#pragma unroll
    for (int r = 0; r < 16; r++) { zero[r] = 0.f; }

    // One ray block (32 rays x both ray sets) at a time, and within it one accumulator tile in flight besides the T tile:
    // the kernel lives at 128 registers per lane with a whole path's state in them (k_path_small), so the pass may hold
    // 9 operands + 2 x 16 accumulators, not the 18 + 48 a "both blocks, both sets" schedule would (measured: 81 spilled
    // dwords).  The half exchanges are issued once per block; the other half of each result is simply not used.
#pragma unroll
    for (int block = 0; block < 2; block++) {
        float rayA[3], rayB[3], rayO[2], farB, unused;
        if (block == 0) {
            mfmaSplitBlocks(directionA.x, directionA.y, &rayA[0], &unused);
            mfmaSplitBlocks(directionA.z, qA.x, &rayA[1], &unused);
            mfmaSplitBlocks(qA.y, qA.z, &rayA[2], &unused);
            mfmaSplitBlocks(p.x, p.y, &rayO[0], &unused);
            mfmaSplitBlocks(p.z, 1.f, &rayO[1], &unused);
        } else {
            mfmaSplitBlocks(directionA.x, directionA.y, &unused, &rayA[0]);
            mfmaSplitBlocks(directionA.z, qA.x, &unused, &rayA[1]);
            mfmaSplitBlocks(qA.y, qA.z, &unused, &rayA[2]);
            mfmaSplitBlocks(p.x, p.y, &unused, &rayO[0]);
            mfmaSplitBlocks(p.z, 1.f, &unused, &rayO[1]);
        }
        if (anyShadow) {
            if (block == 0) {
                mfmaSplitBlocks(directionB.x, directionB.y, &rayB[0], &unused);
                mfmaSplitBlocks(directionB.z, qB.x, &rayB[1], &unused);
                mfmaSplitBlocks(qB.y, qB.z, &rayB[2], &unused);
                mfmaSplitBlocks(tfarClamped, tfarClamped, &farB, &unused);
            } else {
                mfmaSplitBlocks(directionB.x, directionB.y, &unused, &rayB[0]);
                mfmaSplitBlocks(directionB.z, qB.x, &unused, &rayB[1]);
                mfmaSplitBlocks(qB.y, qB.z, &unused, &rayB[2]);
                mfmaSplitBlocks(tfarClamped, tfarClamped, &unused, &farB);
            }
        } else {
            rayB[0] = 0.f; rayB[1] = 0.f; rayB[2] = 0.f; farB = 0.f;
        }
        // any-hit tolerance of this block's rays: (10 tfar + 16) u  (header)
        const float negGammaFar = -fmaf(farB, 10.f * kMfmaUnit, 16.f * kMfmaUnit);
        for (int group = 0; group < nGroups; group++) {
            const float *timeTable = table + kMfmaEdgeFloats + group * kMfmaTimeGroupFloats;
            f16v time = __builtin_amdgcn_mfma_f32_32x32x2f32(timeTable[lane], rayO[0], zero, 0, 0, 0);
            time = __builtin_amdgcn_mfma_f32_32x32x2f32(timeTable[64 + lane], rayO[1], time, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int tile = 4 * group + t;
                if (tile < nTiles) {
                    const float *edgeTable = table + tile * kMfmaEdgeTileFloats;
                    const float a0 = edgeTable[lane], a1 = edgeTable[64 + lane], a2 = edgeTable[128 + lane];
                    {
                        f16v edge = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, rayA[0], zero, 0, 0, 0);
                        edge = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, rayA[1], edge, 0, 0, 0);
                        edge = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, rayA[2], edge, 0, 0, 0);
                        maskA[block] |= mfmaKeepBits<false>(edge, time[4 * t], time[4 * t + 1], time[4 * t + 2], time[4 * t + 3], tnearLow, 0.f, 0.f) << (4 * tile);
                    }
                    if (anyShadow) {
                        __builtin_amdgcn_sched_barrier(0);   // one accumulator tile at a time (above)
                        f16v edge = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, rayB[0], zero, 0, 0, 0);
                        edge = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, rayB[1], edge, 0, 0, 0);
                        edge = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, rayB[2], edge, 0, 0, 0);
                        maskB[block] |= mfmaKeepBits<true>(edge, time[4 * t], time[4 * t + 1], time[4 * t + 2], time[4 * t + 3], tnearLow, farB, negGammaFar) << (4 * tile);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

    // lane (column, h) of block b holds the bits of ray column + 32 b for the triangles 2 (bit) + h: one half exchange
    // per ray set hands every lane its own ray's even word and odd word
    const int evenCount = (nTris + 1) >> 1, oddCount = nTris >> 1;
    const unsigned int evenValid = evenCount >= 32 ? 0xFFFFFFFFu : (1u << evenCount) - 1u;
    const unsigned int oddValid = oddCount >= 32 ? 0xFFFFFFFFu : (1u << oddCount) - 1u;
    const float pp = fmaf(p.x, p.x, fmaf(p.y, p.y, p.z * p.z));
    const bool originCovered = pp <= frame.radiusSquaredLimit;
    {
        const auto swapped = __builtin_amdgcn_permlane32_swap(maskA[0], maskA[1], false, false);
        const float dd = fmaf(directionA.x, directionA.x, fmaf(directionA.y, directionA.y, directionA.z * directionA.z));
        const bool covered = originCovered && dd <= 1.0001f;   // false for NaN: such a ray keeps every triangle
        *evenA = covered ? (swapped[0] & evenValid) : evenValid;
        *oddA = covered ? (swapped[1] & oddValid) : oddValid;
    }
    if (anyShadow) {
        const auto swapped = __builtin_amdgcn_permlane32_swap(maskB[0], maskB[1], false, false);
        const float dd = fmaf(directionB.x, directionB.x, fmaf(directionB.y, directionB.y, directionB.z * directionB.z));
        const bool covered = originCovered && dd <= 1.0001f;
        *evenB = covered ? (swapped[0] & evenValid) : evenValid;
        *oddB = covered ? (swapped[1] & oddValid) : oddValid;
    } else {
        *evenB = 0u;
        *oddB = 0u;
    }
}

// Phase 2 over the even / odd words (kernels.h: smallResolve): a wave-level loop, lanes without a ray pass empty words.
__device__ __forceinline__ void mfmaResolve(const TraceGeometry &geometry, LaneRay &ray, unsigned int even, unsigned int odd)
{
    while (__ballot((even | odd) != 0u) != 0ull) {
        if ((even | odd) != 0u) {
            int k;
            if (even != 0u) { const int b = __ffs((int)even) - 1; even &= even - 1u; k = 2 * b; }
            else { const int b = __ffs((int)odd) - 1; odd &= odd - 1u; k = 2 * b + 1; }
            const float4 t0 = geometry.tris[3 * k + 0];
            const float4 t1 = geometry.tris[3 * k + 1];
            const float4 t2 = geometry.tris[3 * k + 2];
            bool terminate = false;
            testLeafTriangle(ray, t0, t1, t2, &terminate);
            if (terminate) { even = 0u; odd = 0u; }  // shadow ray occluded
        }
    }
}

#endif  // __HIPCC__

}  // namespace pathed
