// Kernels that were built, measured and did NOT become the default (DESIGN.md section 4 has the A/B tables, profiles/ the logs):
//   k_shade_staged            staged, state-sorted shade stage per block                      [r2]  equal or slower beside the other pool's trace kernel
//   k_vertex + k_regen        split shade stage over the trace kernel's hit / miss lists      [r3]  17-30 % slower
//   k_compress_nodes / k_widen_nodes   64-byte and 8-wide compressed BVH nodes                [r3]  equal / slower
//   k_debug_small_candidates  test hook of the matrix-pipe phase 1 (mfma_candidates.h)        [r4]  25 % slower
//   k_valu_clock_probe        the VALU probe with s_memtime clocks
// They are compiled only into libpathed_hip_experiments.so (`make experiments`, -DPATHED_EXPERIMENTS=1); the product library
// carries one shade organisation per scene class, one node format and no probes beyond the two bench.py reports.
#pragma once

#include "kernels.h"

namespace pathed {

// ------------------------------------------------------------------------- staged shade
// k_shade_staged: the same per-slot arithmetic as k_shade, organised as a STAGED, COMPACTED wavefront
// inside every block (the reference's stage-wise arrays, src/data_parallel_integrator.cpp:307-371:
// intersections -> direct light -> bounce, each over all pixels; here each stage runs over a dense,
// state-sorted list of the block's slots).
//
// A block owns ROUNDS x 256 consecutive slots and walks them in three phases:
//   A  classify   one lane per slot, two 16-byte loads (state word, hit).  A slot with a finished ray is
//                 a MISS (sample ends: regenerate), a TERMINAL hit (the path was not going to continue and
//                 the hit is no emitter: the sample ends without an intersection record) or a VERTEX.
//                 Vertices get a key -- 0: an emitter was hit and the BSDF-sampling MIS term has to be
//                 finished (light pdf, triangle corners), 1 + BSDF type otherwise -- and are counting-
//                 sorted by it into an LDS list (wave ballots + one wave-level scan, no atomics).
//   B  vertex     dense over the sorted list: intersection record, previous vertex's MIS term, throughput,
//                 BSDF sample, light sample, next ray + shadow ray.  Waves are full and (up to one boundary
//                 wave per key) uniform in what they execute.  A sample that ends here leaves its colour in
//                 the slot's `res` and joins the regeneration list.
//   C  regenerate dense over the misses, terminal hits and the samples phase B ended: environment term of a
//                 miss, colour -> the unit's partial sum, next sample or next unit, camera ray.
// Only 2-byte slot indices move through LDS; state is read where it is used (the block's lines are in L2
// from phase A on).  The result cannot depend on the order of the lists: every slot is shaded from its own
// state with k_shade's operations in k_shade's order (GPU test: images bit-identical to k_shade's).
static const int kStageKeys = 8;                 // 0 emitter hit with a pending MIS term, 1 + PATHED_MAT_* (0..5), 7 spare
static const unsigned int kEntrySlotMask = 0x0FFFu;
static const int kEntryKindShift = 12;
static const unsigned int kRegenMiss = 0u;       // the ray missed: environment term, then the sample ends
static const unsigned int kRegenTerminal = 1u;   // last ray of the path hit a non-emitter: pending light term, then the sample ends
static const unsigned int kRegenColor = 2u;      // phase B ended the sample: its colour is in res.rgb

#ifndef PATHED_STAGED_WAVES
#define PATHED_STAGED_WAVES 4
#endif
template <bool LDS_MATERIALS, int ROUNDS>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PATHED_STAGED_WAVES, PATHED_STAGED_WAVES))) void k_shade_staged(RenderParams p)
{
    constexpr int kSlots = ROUNDS * kBlock;
    constexpr int kCountEntries = kStageKeys * ROUNDS * kWavesPerBlock;   // 64 (ROUNDS 2) or 128 (ROUNDS 4)
    static_assert(kSlots <= (int)kEntrySlotMask + 1, "slot index must fit the list entry");
    static_assert(kCountEntries % 64 == 0, "one wave scans the key counts");
    extern __shared__ float4 ldsDynamic[];           // LDS_MATERIALS: the material table, nMaterials x 96 B
    __shared__ unsigned short vertexList[kSlots];
    __shared__ unsigned short regenList[kSlots];
    __shared__ unsigned int keyOffsets[kCountEntries];   // [key][round][wave]: counts, then exclusive offsets
    __shared__ unsigned int scratch[kWavesPerBlock + 1];
    __shared__ unsigned int listCounts[2];            // [0] vertices, [1] regenerations

    MaterialAccess<LDS_MATERIALS> materials;
    if (LDS_MATERIALS) {
        const int words = p.scene.nMaterials * (int)(sizeof(DMaterial) / 4);
        const int *source = reinterpret_cast<const int *>(p.scene.materials);
        int *target = reinterpret_cast<int *>(ldsDynamic);
        for (int i = threadIdx.x; i < words; i += kBlock) { target[i] = source[i]; }
        materials.table = reinterpret_cast<const DMaterial *>(ldsDynamic);
    } else {
        materials.table = p.scene.materials;
    }
    if (threadIdx.x < 2) { listCounts[threadIdx.x] = 0u; }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int blockBase = blockIdx.x * kSlots;
    const DScene &scene = p.scene;

    // rewind the card cursors for the pool's next trace launch (same stream: it starts after us)
    if (blockIdx.x == 0 && threadIdx.x < kTraceShards) { p.counters[kCtrTraceCursor + threadIdx.x * kCursorStride] = 0u; }
    // ... and empty the shadow list the NEXT shade launch will fill (the trace launch that read it is over)
    if (blockIdx.x == 0 && threadIdx.x == kTraceShards) { p.counters[kCtrShadowCount + (p.parity ^ 1) * kCursorStride] = 0u; }

    // ---------------------------------------------------------------- phase A: classify + sort
    int myKey[ROUNDS];            // -1: not a vertex
    unsigned int myRank[ROUNDS];  // position among the wave's lanes with the same key
    #pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const int local = r * kBlock + threadIdx.x;
        const int slot = blockBase + local;
        const float4 rd = p.state.rayD[slot];
        const float4 h = p.state.hit[slot];
        const int st = floatAsInt(rd.w);
        const int prim = floatAsInt(h.w);
        int key = -1;
        int regenKind = -1;
        if (!(st & kStDone)) {
            bool parked = false;
            if (p.suspendLanes > 0) {
                // a slot with a parked ray (see kSuspendLanes) sits this iteration out, untouched
                parked = prim == kPrimSuspended;
                if (!parked && (st & kStEligible)) {
                    parked = reinterpret_cast<const int *>(p.state.pend + slot)[3] == kShadowSuspended;
                }
                if (parked && !(st & kStHold)) { reinterpret_cast<int *>(p.state.rayD + slot)[3] = st | kStHold; }
            }
            if (!parked) {
                if (prim < 0) {
                    regenKind = (int)kRegenMiss;
                } else {
                    int material;
                    if (prim < scene.nTris) { material = reinterpret_cast<const int *>(scene.triShade + (size_t)kTriShadeQuads * prim)[3]; }
                    else { material = scene.spheres[prim - scene.nTris].material; }
                    const DMaterial &hitMaterial = materials[material];
                    const bool emitter = !(hitMaterial.emit[0] == 0.f && hitMaterial.emit[1] == 0.f && hitMaterial.emit[2] == 0.f);
                    const int rayBounce = st & kStBounceMask;
                    const bool pendingTerm = rayBounce != 0 && (st & kStEligible) != 0 && emitter;
                    if (rayBounce != 0 && !(st & kStContinue) && !pendingTerm) { regenKind = (int)kRegenTerminal; }
                    else { key = pendingTerm ? 0 : 1 + hitMaterial.type; }
                }
            }
        }
        // regeneration list: order is irrelevant, one LDS atomic per wave
        {
            const unsigned long long mask = __ballot(regenKind >= 0);
            unsigned int base = 0;
            if (lane == 0 && mask != 0ull) { base = atomicAdd(&listCounts[1], (unsigned int)__popcll(mask)); }
            base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
            if (regenKind >= 0) {
                const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
                regenList[base + rank] = (unsigned short)((unsigned int)local | ((unsigned int)regenKind << kEntryKindShift));
            }
        }
        // vertex list: per (key, round, wave) counts now, positions after the scan
        myKey[r] = key;
        myRank[r] = 0u;
        #pragma unroll
        for (int k = 0; k < kStageKeys; k++) {
            const unsigned long long mask = __ballot(key == k);
            if (key == k) { myRank[r] = __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u)); }
            if (lane == 0) { keyOffsets[(k * ROUNDS + r) * kWavesPerBlock + wave] = (unsigned int)__popcll(mask); }
        }
    }
    __syncthreads();
    if (wave == 0) {
        // exclusive scan of the counts in (key, round, wave) order: every lane owns kItems consecutive entries
        constexpr int kItems = kCountEntries / 64;
        unsigned int values[kItems];
        unsigned int sum = 0u;
        #pragma unroll
        for (int k = 0; k < kItems; k++) { values[k] = keyOffsets[lane * kItems + k]; sum += values[k]; }
        unsigned int inclusive = sum;
        #pragma unroll
        for (int delta = 1; delta < 64; delta <<= 1) {
            const unsigned int other = (unsigned int)__shfl_up((int)inclusive, delta, 64);
            if (lane >= delta) { inclusive += other; }
        }
        unsigned int running = inclusive - sum;
        #pragma unroll
        for (int k = 0; k < kItems; k++) { keyOffsets[lane * kItems + k] = running; running += values[k]; }
        if (lane == 63) { listCounts[0] = inclusive; }
    }
    __syncthreads();
    #pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        if (myKey[r] >= 0) {
            const unsigned int position = keyOffsets[(myKey[r] * ROUNDS + r) * kWavesPerBlock + wave] + myRank[r];
            vertexList[position] = (unsigned short)(r * kBlock + threadIdx.x);
        }
    }
    __syncthreads();
    const unsigned int nVertex = listCounts[0];
    SHADE_REGION(0, true);

    // ---------------------------------------------------------------- phase B: vertices, dense and key-sorted
    for (unsigned int listBase = 0; listBase < nVertex; listBase += kBlock) {
        const unsigned int index = listBase + threadIdx.x;
        const bool have = index < nVertex;
        ShadowRequest shadow;
        shadow.push = false;
        shadow.origin = v3(0.f, 0.f, 0.f);
        shadow.direction = v3(0.f, 0.f, 0.f);
        shadow.tfar = 0.f;
        bool ended = false;       // the sample ended at this vertex
        int slot = 0;
        SHADE_REGION(1, have);
        if (have) {
            slot = blockBase + (int)vertexList[index];
            float4 rd = p.state.rayD[slot];
            const float4 h = p.state.hit[slot];
            const float4 ro = p.state.rayO[slot];
            const float4 resIn = p.state.res[slot];
            pinLoaded(rd);
            pinLoaded(h);
            pinLoaded(ro);
            pinLoaded(resIn);
            const int st = floatAsInt(rd.w) & ~kStHold;
            float4 pendIn = make_float4(0.f, 0.f, 0.f, 0.f);
            if (st & kStEligible) { pendIn = p.state.pend[slot]; }

            const V3 o = v3(ro.x, ro.y, ro.z);
            const V3 d = v3(rd.x, rd.y, rd.z);
            const int rayBounce = st & kStBounceMask;  // vertex that spawned this ray, 0 = camera
            const int sampleInUnit = (st >> kStSampleShift) & kStSampleMask;
            const unsigned int unit = (unsigned int)floatAsInt(resIn.w);
            int firstEmitMaterial = floatAsInt(ro.w);

            uint32_t pixel, firstSample, endSample;
            unitSamples(p, unit, &pixel, &firstSample, &endSample);
            const uint32_t sample = firstSample + (uint32_t)sampleInUnit;

            Rgb result = rgb(resIn.x, resIn.y, resIn.z);
            Rgb modulation = rgb(1.f);
            Rgb color = rgb(0.f);
            bool haveVertex = false;
            const int vertex = rayBounce + 1;
            Isect isect = makeIsect(scene, o, d, h);

            if (rayBounce == 0) {
                // SampleIntegrator::samplePixel, src/sample_integrator.cpp:18-59
                firstEmitMaterial = -1;
                if (checkCounts(p.startBounce, p.lastBounce, 0)) {
                    const Rgb emit = matEmit(materials[isect.material]);
                    const bool backside = dot(isect.normal, isect.wo) < 0.f;
                    if (!isBlack(emit) && !backside) { firstEmitMaterial = isect.material; }
                }
                result = rgb(0.f);
                haveVertex = true;
            } else {
                // the ray left vertex `rayBounce` along its BSDF sample
                const float4 modIn = p.state.mod[slot];
                const float4 thrIn = p.state.thr[slot];
                modulation = rgb(modIn.x, modIn.y, modIn.z);
                const float bsdfPdf = modIn.w;
                const Rgb throughput = rgb(thrIn.x, thrIn.y, thrIn.z);
                const float cosTheta = thrIn.w;

                if (st & kStEligible) {
                    // PathTracer::directSampleBSDF, src/path_tracer.cpp:167-216 (the hit branch)
                    Rgb bsdfTerm = rgb(0.f);
                    const Rgb emit = matEmit(materials[isect.material]);
                    if (!isBlack(emit) && dot(isect.wo, isect.shadingNormal) >= 0.f) {
                        const float lightPDF = lightsPDF(scene, o, isect);
                        const float brdfWeight = (st & kStDelta)
                            ? 1.f
                            : (1 * bsdfPdf) / (1 * bsdfPdf + 1 * lightPDF);
                        bsdfTerm = emit * brdfWeight * throughput * cosTheta / bsdfPdf;
                    }
                    const Rgb Ld = rgb(pendIn.x, pendIn.y, pendIn.z) + bsdfTerm;
                    if (rayBounce == 1) { result = Ld; }
                    else { result = result + Ld * modulation; }
                }

                // PathTracer::L loop body, src/path_tracer.cpp:41-58
                if (!(st & kStContinue)) {
                    ended = true;
                } else {
                    const float invPDF = 1.f / bsdfPdf;
                    modulation = modulation * (throughput * cosTheta * invPDF);
                    if (isBlack(modulation)) { ended = true; }
                    else { haveVertex = true; }
                }
                if (ended) {
                    Rgb first = rgb(0.f);
                    if (firstEmitMaterial >= 0) { first = first + matEmit(materials[firstEmitMaterial]); }
                    color = first + result;
                }
            }

            float4 outMod = make_float4(modulation.r, modulation.g, modulation.b, 1.f);
            float4 outRayO = ro, outRayD = rd;
            float4 outThr = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 outPend = make_float4(0.f, 0.f, 0.f, 0.f);

            SHADE_REGION(2, haveVertex);
            if (haveVertex) {
                // PathTracer::L: sample the BSDF, then direct(), src/path_tracer.cpp:30-36, 60-73
                const DMaterial &material = materials[isect.material];

                Rng random;
                makeKey(((uint64_t)p.seedHi << 32) | p.seedLo, pixel, sample, &random.k0, &random.k1);
                random.dimension = vertexBase(vertex);
                prepareLobes<TraitsAll>(material, isect);
                const BSDFSample bsdfSample = materialSample(material, isect, random);

                const bool counts = checkCounts(p.startBounce, p.lastBounce, vertex);
                const bool emissive = !isBlack(matEmit(material));
                const bool wantDirect = counts && !emissive;  // direct() returns 0 on emitters (:86-90)
                const bool wantContinue = !checkDone(p.lastBounce, vertex + 1);

                Rgb lightTerm = rgb(0.f);
                SHADE_REGION(3, wantDirect);
                if (wantDirect) {
                    random.dimension = vertexBase(vertex) + 3;
                    lightTerm = sampleLightsTerm(scene, materials, isect, material, random, &shadow);
                }

                // see k_shade: a vertex with nothing pending whose BSDF sample has exactly black throughput ends the sample
                const bool deadEnd = isBlack(bsdfSample.throughput) && bsdfSample.pdf > 0.f && bsdfSample.pdf < 3e38f
                    && !shadow.push && isBlack(lightTerm);
                if ((!wantDirect && !wantContinue) || deadEnd) {
                    ended = true;
                    Rgb first = rgb(0.f);
                    if (firstEmitMaterial >= 0) { first = first + matEmit(materials[firstEmitMaterial]); }
                    color = first + result;
                    shadow.push = false;
                } else {
                    int nextState = vertex | (sampleInUnit << kStSampleShift);
                    if (wantDirect) { nextState |= kStEligible; }
                    if (isDelta(material)) { nextState |= kStDelta; }
                    if (wantContinue) { nextState |= kStContinue; }
                    outRayO = make_float4(isect.point.x, isect.point.y, isect.point.z, intAsFloat(firstEmitMaterial));
                    outRayD = make_float4(bsdfSample.wiWorld.x, bsdfSample.wiWorld.y, bsdfSample.wiWorld.z, intAsFloat(nextState));
                    outMod.w = bsdfSample.pdf;
                    outThr = make_float4(
                        bsdfSample.throughput.r, bsdfSample.throughput.g, bsdfSample.throughput.b,
                        fabsf(dot(isect.shadingNormal, bsdfSample.wiWorld)));
                    outPend = make_float4(lightTerm.r, lightTerm.g, lightTerm.b, 0.f);
                }
            }

            if (ended) {
                // phase C adds the colour to the unit's partial sum and restarts the slot
                p.state.res[slot] = make_float4(color.r, color.g, color.b, intAsFloat((int)unit));
            } else {
                p.state.rayO[slot] = outRayO;
                p.state.rayD[slot] = outRayD;
                p.state.mod[slot] = outMod;
                p.state.thr[slot] = outThr;
                p.state.res[slot] = make_float4(result.r, result.g, result.b, intAsFloat((int)unit));
                p.state.pend[slot] = outPend;
            }
        }

        // samples that ended here join the regeneration list
        {
            const unsigned long long mask = __ballot(ended);
            unsigned int base = 0;
            if (lane == 0 && mask != 0ull) { base = atomicAdd(&listCounts[1], (unsigned int)__popcll(mask)); }
            base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
            if (ended) {
                const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
                regenList[base + rank] = (unsigned short)((unsigned int)(slot - blockBase) | (kRegenColor << kEntryKindShift));
            }
        }

        // shadow-ray list: wave ballot + prefix popcount + LDS block scan, then ONE atomic per block and round
        {
            const unsigned long long mask = __ballot(shadow.push);
            const unsigned int before = (unsigned int)__popcll(mask & ((1ull << lane) - 1ull));
            if (lane == 0) { scratch[wave] = (unsigned int)__popcll(mask); }
            __syncthreads();
            unsigned int offset = 0, total = 0;
            #pragma unroll
            for (int w = 0; w < kWavesPerBlock; w++) {
                const unsigned int count = scratch[w];
                if (w < wave) { offset += count; }
                total += count;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                scratch[kWavesPerBlock] = total ? atomicAdd(&p.counters[kCtrShadowCount + p.parity * kCursorStride], total) : 0u;
            }
            __syncthreads();
            SHADE_REGION(4, shadow.push);
            if (shadow.push) {
                const unsigned int at = scratch[kWavesPerBlock] + offset + before;
                p.state.shO[at] = make_float4(shadow.origin.x, shadow.origin.y, shadow.origin.z, shadow.tfar);
                p.state.shD[at] = make_float4(shadow.direction.x, shadow.direction.y, shadow.direction.z, intAsFloat(slot));
            }
            __syncthreads();   // scratch is reused by the next round
        }
    }
    __syncthreads();
    const unsigned int nRegen = listCounts[1];

    // ---------------------------------------------------------------- phase C: end of sample, regeneration
    unsigned int retiredTotal = 0;
    for (unsigned int listBase = 0; listBase < nRegen; listBase += kBlock) {
        const unsigned int index = listBase + threadIdx.x;
        const bool have = index < nRegen;
        bool needUnit = false, startNext = false;
        uint32_t nextPixel = 0, nextSample = 0;
        int sampleInUnit = 0;
        unsigned int unit = 0xFFFFFFFFu;
        int slot = 0;
        float4 outRayO = make_float4(0.f, 0.f, 0.f, 0.f), outRayD = make_float4(0.f, 0.f, 0.f, 0.f);
        SHADE_REGION(5, have);
        if (have) {
            const unsigned int entry = regenList[index];
            const unsigned int kind = entry >> kEntryKindShift;
            slot = blockBase + (int)(entry & kEntrySlotMask);
            const float4 rd = p.state.rayD[slot];
            const float4 resIn = p.state.res[slot];
            const float4 partialIn = p.state.acc[slot];
            const int st = floatAsInt(rd.w) & ~kStHold;
            outRayD = rd;
            unit = (unsigned int)floatAsInt(resIn.w);
            sampleInUnit = (st >> kStSampleShift) & kStSampleMask;
            Rgb color = rgb(resIn.x, resIn.y, resIn.z);
            SHADE_REGION(6, kind != kRegenColor);
            if (kind != kRegenColor) {
                const float4 ro = p.state.rayO[slot];
                const V3 d = v3(rd.x, rd.y, rd.z);
                const int rayBounce = st & kStBounceMask;
                const int firstEmitMaterial = floatAsInt(ro.w);
                if (rayBounce == 0) {
                    // SampleIntegrator::samplePixel: the camera ray missed, src/sample_integrator.cpp:18-31
                    color = rgb(0.f) + environmentL(scene, d);
                } else {
                    Rgb result = rgb(resIn.x, resIn.y, resIn.z);
                    if (st & kStEligible) {
                        const float4 pendIn = p.state.pend[slot];
                        const float4 modIn = p.state.mod[slot];
                        const float4 thrIn = p.state.thr[slot];
                        const Rgb modulation = rgb(modIn.x, modIn.y, modIn.z);
                        const float bsdfPdf = modIn.w;
                        const Rgb throughput = rgb(thrIn.x, thrIn.y, thrIn.z);
                        const float cosTheta = thrIn.w;
                        // PathTracer::directSampleBSDF, src/path_tracer.cpp:167-216: a terminal hit of a non-emitter
                        // adds nothing, a miss adds the environment's term
                        Rgb bsdfTerm = rgb(0.f);
                        if (kind == kRegenMiss) {
                            const Rgb environmentLight = environmentL(scene, d);
                            if (!isBlack(environmentLight)) {
                                // Scene::environmentPDF, src/scene.cpp:494-502
                                const float lightPDF = envEmitPDF(scene.env, d) / scene.nLights;
                                const float brdfWeight = (st & kStDelta)
                                    ? 1.f
                                    : (1 * bsdfPdf) / (1 * bsdfPdf + 1 * lightPDF);
                                bsdfTerm = environmentLight * brdfWeight * throughput * cosTheta / bsdfPdf;
                            }
                        }
                        const Rgb Ld = rgb(pendIn.x, pendIn.y, pendIn.z) + bsdfTerm;
                        if (rayBounce == 1) { result = Ld; }
                        else { result = result + Ld * modulation; }
                    }
                    Rgb first = rgb(0.f);
                    if (firstEmitMaterial >= 0) { first = first + matEmit(materials[firstEmitMaterial]); }
                    color = first + result;
                }
            }

            // radianceLookup += color, src/sample_integrator.cpp:61-63; non-finite samples dropped
            float4 partial = partialIn;
            const bool finite = isfinite(color.r) && isfinite(color.g) && isfinite(color.b);
            if (finite) {
                partial.x += color.r;
                partial.y += color.g;
                partial.z += color.b;
            } else {
                atomicAdd(&p.stats[kStatDropped], 1ull);
            }
            uint32_t pixel, firstSample, endSample;
            unitSamples(p, unit, &pixel, &firstSample, &endSample);
            sampleInUnit++;
            if (firstSample + (uint32_t)sampleInUnit < endSample) {
                startNext = true;
                nextPixel = pixel;
                nextSample = firstSample + (uint32_t)sampleInUnit;
                p.state.acc[slot] = partial;
            } else {
                p.state.chunkBuf[partialIndex(p, unit)] = partial;
                p.state.acc[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
                needUnit = true;
            }
        }

        // block-aggregated grab of the next units (wave ballot + LDS scan, one atomic per block and round)
        const unsigned int newUnit = grabUnits(p, needUnit, scratch);
        bool retired = false;
        if (needUnit) {
            unit = newUnit;
            if (newUnit != 0xFFFFFFFFu) {
                uint32_t endSample;
                unitSamples(p, newUnit, &nextPixel, &nextSample, &endSample);
                sampleInUnit = 0;
                startNext = true;
            } else {
                outRayD.w = intAsFloat(kStDone);
                retired = true;
            }
        }
        SHADE_REGION(7, startNext);
        if (startNext) { startSample(p, nextPixel, nextSample, sampleInUnit, &outRayO, &outRayD); }
        if (have) {
            p.state.rayO[slot] = outRayO;
            p.state.rayD[slot] = outRayD;
            p.state.mod[slot] = make_float4(1.f, 1.f, 1.f, 1.f);
            p.state.thr[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
            p.state.res[slot] = make_float4(0.f, 0.f, 0.f, intAsFloat((int)unit));
            p.state.pend[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        retiredTotal += (unsigned int)__popcll(__ballot(retired));
    }
    // slots that ran out of units (only at the tail of a render call)
    if (lane == 0 && retiredTotal != 0u) { atomicSub(&p.counters[kCtrRemaining], retiredTotal); }
}

// ------------------------------------------------------------------------- split shade stage
// k_vertex + k_regen: k_shade's per-slot arithmetic as two DENSE kernels over the lists the trace kernel wrote
// (the reference's stage-wise arrays, src/data_parallel_integrator.cpp:307-371: intersections -> direct light -> bounce,
// each over everything that needs it).  Nothing is classified or sorted here: k_trace knows hit from miss when a ray
// finishes and appends the slot to the list of the kernel that has to see it.
//   k_vertex  one lane per HIT: intersection record, the previous vertex's BSDF-sampling MIS term (emitter hit),
//             throughput, BSDF sample, light sample, next ray + shadow ray.  No regeneration code, full waves.  A sample
//             that ends at the vertex leaves its colour in `res` and joins the miss list (kEntryColorReady).
//   k_regen   one lane per MISS or ended sample: environment term of the miss, colour -> the unit's partial sum, next
//             sample or next unit, camera ray.  Few registers, no material table in LDS.
// Both are persistent (a block walks groups of 256 list items), so the prologue is paid once per block.  A slot is shaded
// from its own state with k_shade's operations in k_shade's order: images are bit-identical (GPU test).
// A slot whose closest-hit ray is in but whose shadow ray is parked in a trace wave cannot be shaded yet: it is put on
// hold and handed to the next iteration through the deferred list of its kernel.
#ifndef PATHED_VERTEX_WAVES
#define PATHED_VERTEX_WAVES 4
#endif
#ifndef PATHED_REGEN_WAVES
#define PATHED_REGEN_WAVES 6
#endif

template <bool LDS_MATERIALS>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PATHED_VERTEX_WAVES, PATHED_VERTEX_WAVES))) void k_vertex(RenderParams p)
{
    extern __shared__ float4 ldsDynamic[];           // LDS_MATERIALS: the material table, nMaterials x 96 B
    __shared__ unsigned int scratch[kWavesPerBlock + 1];
    __shared__ unsigned int shardCounts[kListShards + 1];
    __shared__ unsigned int endedBuffers[kWavesPerBlock * 128];

    MaterialAccess<LDS_MATERIALS> materials;
    if (LDS_MATERIALS) {
        const int words = p.scene.nMaterials * (int)(sizeof(DMaterial) / 4);
        const int *source = reinterpret_cast<const int *>(p.scene.materials);
        int *target = reinterpret_cast<int *>(ldsDynamic);
        for (int i = threadIdx.x; i < words; i += kBlock) { target[i] = source[i]; }
        materials.table = reinterpret_cast<const DMaterial *>(ldsDynamic);
    } else {
        materials.table = p.scene.materials;
    }
    ListReader hits;
    listReaderInit(hits, shardCounts, p, kListHit, p.parity);   // its barriers also publish the material table

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const DScene &scene = p.scene;
    ListWriter endedWriter;   // samples that end at a vertex: on to k_regen through the miss list
    listWriterInit(endedWriter, endedBuffers + wave * 128);

    const unsigned int groups = (hits.items + kBlock - 1u) / kBlock;
    for (unsigned int group = blockIdx.x; group < groups; group += gridDim.x) {
        const unsigned int entry = listEntry(hits, group * kBlock + threadIdx.x);
        bool have = entry != kEntryInvalid;
        const int slot = (int)(entry & kEntrySlotBits);
        ShadowRequest shadow;
        shadow.push = false;
        shadow.origin = v3(0.f, 0.f, 0.f);
        shadow.direction = v3(0.f, 0.f, 0.f);
        shadow.tfar = 0.f;
        bool ended = false;       // the sample ended at this vertex
        bool defer = false;       // the slot's shadow ray is parked
        if (have) {
            float4 rd = p.state.rayD[slot];
            const float4 h = p.state.hit[slot];
            const float4 ro = p.state.rayO[slot];
            const float4 resIn = p.state.res[slot];
            pinLoaded(rd);
            pinLoaded(h);
            pinLoaded(ro);
            pinLoaded(resIn);
            const int stIn = floatAsInt(rd.w);
            const int st = stIn & ~kStHold;
            float4 pendIn = make_float4(0.f, 0.f, 0.f, 0.f);
            if (st & kStEligible) {
                pendIn = p.state.pend[slot];
                if (p.suspendLanes > 0 && floatAsInt(pendIn.w) == kShadowSuspended) {
                    if (!(stIn & kStHold)) { reinterpret_cast<int *>(p.state.rayD + slot)[3] = stIn | kStHold; }
                    defer = true;
                    have = false;
                }
            }
            if (have) {
                rd.w = intAsFloat(st);
                const V3 o = v3(ro.x, ro.y, ro.z);
                const V3 d = v3(rd.x, rd.y, rd.z);
                const int rayBounce = st & kStBounceMask;  // vertex that spawned this ray, 0 = camera
                const int sampleInUnit = (st >> kStSampleShift) & kStSampleMask;
                const unsigned int unit = (unsigned int)floatAsInt(resIn.w);
                int firstEmitMaterial = floatAsInt(ro.w);

                uint32_t pixel, firstSample, endSample;
                unitSamples(p, unit, &pixel, &firstSample, &endSample);
                const uint32_t sample = firstSample + (uint32_t)sampleInUnit;

                Rgb result = rgb(resIn.x, resIn.y, resIn.z);
                Rgb modulation = rgb(1.f);
                Rgb color = rgb(0.f);
                bool haveVertex = false;
                const int vertex = rayBounce + 1;
                Isect isect = makeIsect(scene, o, d, h);

                if (rayBounce == 0) {
                    // SampleIntegrator::samplePixel, src/sample_integrator.cpp:18-59
                    firstEmitMaterial = -1;
                    if (checkCounts(p.startBounce, p.lastBounce, 0)) {
                        const Rgb emit = matEmit(materials[isect.material]);
                        const bool backside = dot(isect.normal, isect.wo) < 0.f;
                        if (!isBlack(emit) && !backside) { firstEmitMaterial = isect.material; }
                    }
                    result = rgb(0.f);
                    haveVertex = true;
                } else {
                    // the ray left vertex `rayBounce` along its BSDF sample
                    const float4 modIn = p.state.mod[slot];
                    const float4 thrIn = p.state.thr[slot];
                    modulation = rgb(modIn.x, modIn.y, modIn.z);
                    const float bsdfPdf = modIn.w;
                    const Rgb throughput = rgb(thrIn.x, thrIn.y, thrIn.z);
                    const float cosTheta = thrIn.w;

                    if (st & kStEligible) {
                        // PathTracer::directSampleBSDF, src/path_tracer.cpp:167-216 (the hit branch)
                        Rgb bsdfTerm = rgb(0.f);
                        const Rgb emit = matEmit(materials[isect.material]);
                        if (!isBlack(emit) && dot(isect.wo, isect.shadingNormal) >= 0.f) {
                            const float lightPDF = lightsPDF(scene, o, isect);
                            const float brdfWeight = (st & kStDelta)
                                ? 1.f
                                : (1 * bsdfPdf) / (1 * bsdfPdf + 1 * lightPDF);
                            bsdfTerm = emit * brdfWeight * throughput * cosTheta / bsdfPdf;
                        }
                        const Rgb Ld = rgb(pendIn.x, pendIn.y, pendIn.z) + bsdfTerm;
                        if (rayBounce == 1) { result = Ld; }
                        else { result = result + Ld * modulation; }
                    }

                    // PathTracer::L loop body, src/path_tracer.cpp:41-58
                    if (!(st & kStContinue)) {
                        ended = true;
                    } else {
                        const float invPDF = 1.f / bsdfPdf;
                        modulation = modulation * (throughput * cosTheta * invPDF);
                        if (isBlack(modulation)) { ended = true; }
                        else { haveVertex = true; }
                    }
                    if (ended) {
                        Rgb first = rgb(0.f);
                        if (firstEmitMaterial >= 0) { first = first + matEmit(materials[firstEmitMaterial]); }
                        color = first + result;
                    }
                }

                float4 outMod = make_float4(modulation.r, modulation.g, modulation.b, 1.f);
                float4 outRayO = ro, outRayD = rd;
                float4 outThr = make_float4(0.f, 0.f, 0.f, 0.f);
                float4 outPend = make_float4(0.f, 0.f, 0.f, 0.f);

                if (haveVertex) {
                    // PathTracer::L: sample the BSDF, then direct(), src/path_tracer.cpp:30-36, 60-73
                    const DMaterial &material = materials[isect.material];

                    Rng random;
                    makeKey(((uint64_t)p.seedHi << 32) | p.seedLo, pixel, sample, &random.k0, &random.k1);
                    random.dimension = vertexBase(vertex);
                    prepareLobes<TraitsAll>(material, isect);
                    const BSDFSample bsdfSample = materialSample(material, isect, random);

                    const bool counts = checkCounts(p.startBounce, p.lastBounce, vertex);
                    const bool emissive = !isBlack(matEmit(material));
                    const bool wantDirect = counts && !emissive;  // direct() returns 0 on emitters (:86-90)
                    const bool wantContinue = !checkDone(p.lastBounce, vertex + 1);

                    Rgb lightTerm = rgb(0.f);
                    if (wantDirect) {
                        random.dimension = vertexBase(vertex) + 3;
                        lightTerm = sampleLightsTerm(scene, materials, isect, material, random, &shadow);
                    }

                    // see k_shade: a vertex with nothing pending whose BSDF sample has exactly black throughput ends the sample
                    const bool deadEnd = isBlack(bsdfSample.throughput) && bsdfSample.pdf > 0.f && bsdfSample.pdf < 3e38f
                        && !shadow.push && isBlack(lightTerm);
                    if ((!wantDirect && !wantContinue) || deadEnd) {
                        ended = true;
                        Rgb first = rgb(0.f);
                        if (firstEmitMaterial >= 0) { first = first + matEmit(materials[firstEmitMaterial]); }
                        color = first + result;
                        shadow.push = false;
                    } else {
                        int nextState = vertex | (sampleInUnit << kStSampleShift);
                        if (wantDirect) { nextState |= kStEligible; }
                        if (isDelta(material)) { nextState |= kStDelta; }
                        if (wantContinue) { nextState |= kStContinue; }
                        outRayO = make_float4(isect.point.x, isect.point.y, isect.point.z, intAsFloat(firstEmitMaterial));
                        outRayD = make_float4(bsdfSample.wiWorld.x, bsdfSample.wiWorld.y, bsdfSample.wiWorld.z, intAsFloat(nextState));
                        outMod.w = bsdfSample.pdf;
                        outThr = make_float4(
                            bsdfSample.throughput.r, bsdfSample.throughput.g, bsdfSample.throughput.b,
                            fabsf(dot(isect.shadingNormal, bsdfSample.wiWorld)));
                        outPend = make_float4(lightTerm.r, lightTerm.g, lightTerm.b, 0.f);
                    }
                }

                if (ended) {
                    // k_regen adds the colour to the unit's partial sum and restarts the slot
                    p.state.res[slot] = make_float4(color.r, color.g, color.b, intAsFloat((int)unit));
                } else {
                    p.state.rayO[slot] = outRayO;
                    p.state.rayD[slot] = outRayD;
                    p.state.mod[slot] = outMod;
                    p.state.thr[slot] = outThr;
                    p.state.res[slot] = make_float4(result.r, result.g, result.b, intAsFloat((int)unit));
                    p.state.pend[slot] = outPend;
                }
            }
        }

        deferSlot(p, kListHit, defer, (unsigned int)slot);
        listAppend(endedWriter, p, kListMiss, ended, (unsigned int)slot | kEntryColorReady);

        // shadow-ray list: wave ballot + prefix popcount + LDS block scan, then ONE atomic per block and group
        {
            const unsigned long long mask = __ballot(shadow.push);
            const unsigned int before = (unsigned int)__popcll(mask & ((1ull << lane) - 1ull));
            if (lane == 0) { scratch[wave] = (unsigned int)__popcll(mask); }
            __syncthreads();
            unsigned int offset = 0, total = 0;
            #pragma unroll
            for (int w = 0; w < kWavesPerBlock; w++) {
                const unsigned int count = scratch[w];
                if (w < wave) { offset += count; }
                total += count;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                scratch[kWavesPerBlock] = total ? atomicAdd(&p.counters[kCtrShadowCount + p.parity * kCursorStride], total) : 0u;
            }
            __syncthreads();
            if (shadow.push) {
                const unsigned int at = scratch[kWavesPerBlock] + offset + before;
                p.state.shO[at] = make_float4(shadow.origin.x, shadow.origin.y, shadow.origin.z, shadow.tfar);
                p.state.shD[at] = make_float4(shadow.direction.x, shadow.direction.y, shadow.direction.z, intAsFloat(slot));
            }
            __syncthreads();   // scratch is reused by the next group
        }
    }
    if (endedWriter.count != 0u) { listFlush(endedWriter, p, kListMiss); }
}

__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PATHED_REGEN_WAVES, PATHED_REGEN_WAVES))) void k_regen(RenderParams p)
{
    __shared__ unsigned int scratch[kWavesPerBlock + 1];
    __shared__ unsigned int shardCounts[kListShards + 1];

    // rewind the card cursors for the pool's next trace launch (same stream: it starts after us)
    if (blockIdx.x == 0 && threadIdx.x < kTraceShards) { p.counters[kCtrTraceCursor + threadIdx.x * kCursorStride] = 0u; }
    // ... and empty the shadow list the NEXT k_vertex will fill (the trace launch that read it is over)
    if (blockIdx.x == 0 && threadIdx.x == kTraceShards) { p.counters[kCtrShadowCount + (p.parity ^ 1) * kCursorStride] = 0u; }

    ListReader misses;
    listReaderInit(misses, shardCounts, p, kListMiss, p.parity);

    const int lane = threadIdx.x & 63;
    const DScene &scene = p.scene;
    const DMaterial *materials = p.scene.materials;   // only the emission of a directly visible emitter is read here

    unsigned int retiredTotal = 0;
    const unsigned int groups = (misses.items + kBlock - 1u) / kBlock;
    for (unsigned int group = blockIdx.x; group < groups; group += gridDim.x) {
        const unsigned int entry = listEntry(misses, group * kBlock + threadIdx.x);
        bool have = entry != kEntryInvalid;
        const int slot = (int)(entry & kEntrySlotBits);
        const bool colorReady = (entry & kEntryColorReady) != 0u;
        bool needUnit = false, startNext = false, defer = false;
        uint32_t nextPixel = 0, nextSample = 0;
        int sampleInUnit = 0;
        unsigned int unit = 0xFFFFFFFFu;
        float4 outRayO = make_float4(0.f, 0.f, 0.f, 0.f), outRayD = make_float4(0.f, 0.f, 0.f, 0.f);
        if (have) {
            const float4 rd = p.state.rayD[slot];
            const float4 resIn = p.state.res[slot];
            const int stIn = floatAsInt(rd.w);
            const int st = stIn & ~kStHold;
            outRayD = rd;
            unit = (unsigned int)floatAsInt(resIn.w);
            sampleInUnit = (st >> kStSampleShift) & kStSampleMask;
            Rgb color = rgb(resIn.x, resIn.y, resIn.z);
            if (!colorReady) {
                const V3 d = v3(rd.x, rd.y, rd.z);
                const int rayBounce = st & kStBounceMask;
                if (rayBounce == 0) {
                    // SampleIntegrator::samplePixel: the camera ray missed, src/sample_integrator.cpp:18-31
                    color = rgb(0.f) + environmentL(scene, d);
                } else {
                    Rgb result = rgb(resIn.x, resIn.y, resIn.z);
                    if (st & kStEligible) {
                        const float4 pendIn = p.state.pend[slot];
                        if (p.suspendLanes > 0 && floatAsInt(pendIn.w) == kShadowSuspended) {
                            if (!(stIn & kStHold)) { reinterpret_cast<int *>(p.state.rayD + slot)[3] = stIn | kStHold; }
                            defer = true;
                            have = false;
                        } else {
                            const float4 modIn = p.state.mod[slot];
                            const float4 thrIn = p.state.thr[slot];
                            const Rgb modulation = rgb(modIn.x, modIn.y, modIn.z);
                            const float bsdfPdf = modIn.w;
                            const Rgb throughput = rgb(thrIn.x, thrIn.y, thrIn.z);
                            const float cosTheta = thrIn.w;
                            // PathTracer::directSampleBSDF, src/path_tracer.cpp:167-216 (the miss branch)
                            Rgb bsdfTerm = rgb(0.f);
                            const Rgb environmentLight = environmentL(scene, d);
                            if (!isBlack(environmentLight)) {
                                // Scene::environmentPDF, src/scene.cpp:494-502
                                const float lightPDF = envEmitPDF(scene.env, d) / scene.nLights;
                                const float brdfWeight = (st & kStDelta)
                                    ? 1.f
                                    : (1 * bsdfPdf) / (1 * bsdfPdf + 1 * lightPDF);
                                bsdfTerm = environmentLight * brdfWeight * throughput * cosTheta / bsdfPdf;
                            }
                            const Rgb Ld = rgb(pendIn.x, pendIn.y, pendIn.z) + bsdfTerm;
                            if (rayBounce == 1) { result = Ld; }
                            else { result = result + Ld * modulation; }
                        }
                    }
                    if (have) {
                        const int firstEmitMaterial = floatAsInt(p.state.rayO[slot].w);
                        Rgb first = rgb(0.f);
                        if (firstEmitMaterial >= 0) { first = first + matEmit(materials[firstEmitMaterial]); }
                        color = first + result;
                    }
                }
            }

            if (have) {
                // radianceLookup += color, src/sample_integrator.cpp:61-63; non-finite samples dropped.  With one sample
                // per unit (the default) the unit's partial sum is 0 + colour: the `acc` stream is not touched at all.
                const bool singleSample = p.chunk == 1;
                float4 partial = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!singleSample) { partial = p.state.acc[slot]; }
                const bool finite = isfinite(color.r) && isfinite(color.g) && isfinite(color.b);
                if (finite) {
                    partial.x += color.r;
                    partial.y += color.g;
                    partial.z += color.b;
                } else {
                    atomicAdd(&p.stats[kStatDropped], 1ull);
                }
                uint32_t pixel, firstSample, endSample;
                unitSamples(p, unit, &pixel, &firstSample, &endSample);
                sampleInUnit++;
                if (firstSample + (uint32_t)sampleInUnit < endSample) {
                    startNext = true;
                    nextPixel = pixel;
                    nextSample = firstSample + (uint32_t)sampleInUnit;
                    p.state.acc[slot] = partial;
                } else {
                    p.state.chunkBuf[partialIndex(p, unit)] = partial;
                    if (!singleSample) { p.state.acc[slot] = make_float4(0.f, 0.f, 0.f, 0.f); }
                    needUnit = true;
                }
            }
        }
        deferSlot(p, kListMiss, defer, (unsigned int)slot);

        // block-aggregated grab of the next units (wave ballot + LDS scan, one atomic per block and group)
        const unsigned int newUnit = grabUnits(p, needUnit, scratch);
        bool retired = false;
        if (needUnit) {
            unit = newUnit;
            if (newUnit != 0xFFFFFFFFu) {
                uint32_t endSample;
                unitSamples(p, newUnit, &nextPixel, &nextSample, &endSample);
                sampleInUnit = 0;
                startNext = true;
            } else {
                outRayD.w = intAsFloat(kStDone);
                retired = true;
            }
        }
        if (startNext) { startSample(p, nextPixel, nextSample, sampleInUnit, &outRayO, &outRayD); }
        if (have) {
            // mod / thr / pend are not reset: a camera ray's vertex reads none of them (k_vertex: rayBounce == 0, no
            // eligible bit) and writes all three for the rays that follow
            p.state.rayO[slot] = outRayO;
            p.state.rayD[slot] = outRayD;
            p.state.res[slot] = make_float4(0.f, 0.f, 0.f, intAsFloat((int)unit));
        }
        retiredTotal += (unsigned int)__popcll(__ballot(retired));
    }
    // slots that ran out of units (only at the tail of a render call)
    if (lane == 0 && retiredTotal != 0u) { atomicSub(&p.counters[kCtrRemaining], retiredTotal); }
}

// The compressed form of the tree (trace.h: nodeQ), one thread per node, after any of the three builders.  Per axis:
// origin = the smallest lower bound of the node's children, scale = the smallest power of two with 253 * scale >= extent,
// q = the plane's grid index (computed exactly, in double) rounded outward past an extra 1/256 of a step (that covers the
// rounding of the traversal's q * (scale / d) + (origin / d - o / d), <= 2^-16 of a step within the node; further away
// the traversal's 1.0000004 factor covers both forms alike).  The refs are copied.
__global__ __launch_bounds__(kBlock) void k_compress_nodes(const float4 *nodes, int nNodes, float4 *nodesQ)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nNodes) { return; }
    const float4 *node = nodes + (size_t)8 * i;
    const float4 refBits = node[6];
    const int ref[4] = { floatAsInt(refBits.x), floatAsInt(refBits.y), floatAsInt(refBits.z), floatAsInt(refBits.w) };
    float origin[3], scale[3];
    unsigned int qlo[3], qhi[3];
    #pragma unroll
    for (int a = 0; a < 3; a++) {
        const float4 lo4 = node[a], hi4 = node[3 + a];
        const float lo[4] = { lo4.x, lo4.y, lo4.z, lo4.w }, hi[4] = { hi4.x, hi4.y, hi4.z, hi4.w };
        float low = INFINITY, high = -INFINITY;
        for (int c = 0; c < 4; c++) {
            if (ref[c] != kEmptyChild) { low = fminf(low, lo[c]); high = fmaxf(high, hi[c]); }
        }
        if (!(low <= high)) { low = 0.f; high = 0.f; }   // no children
        const double extent = (double)high - (double)low;    // exact
        int exponent = 0;
        (void)frexp(extent / 253.0, &exponent);            // extent / 253 = m * 2^exponent with m < 1
        const float step = extent > 0.0 ? ldexpf(1.f, exponent) : 0.f;
        const double perStep = step > 0.f ? 1.0 / (double)step : 0.0;   // exact: a power of two
        origin[a] = low;
        scale[a] = step;
        qlo[a] = 0u;
        qhi[a] = 0u;
        for (int c = 0; c < 4; c++) {
            if (ref[c] == kEmptyChild) { continue; }
            const double below = floor(((double)lo[c] - (double)low) * perStep - 1.0 / 256.0);
            const double above = ceil(((double)hi[c] - (double)low) * perStep + 1.0 / 256.0);
            const unsigned int qBelow = (unsigned int)fmin(fmax(below, 0.0), 255.0);
            const unsigned int qAbove = step > 0.f ? (unsigned int)fmin(fmax(above, 0.0), 255.0) : 0u;
            qlo[a] |= qBelow << (8 * c);
            qhi[a] |= qAbove << (8 * c);
        }
    }
    float4 *out = nodesQ + (size_t)4 * i;
    out[0] = make_float4(origin[0], origin[1], origin[2], scale[0]);
    out[1] = make_float4(scale[1], scale[2], intAsFloat((int)qlo[0]), intAsFloat((int)qlo[1]));
    out[2] = make_float4(intAsFloat((int)qlo[2]), intAsFloat((int)qhi[0]), intAsFloat((int)qhi[1]), intAsFloat((int)qhi[2]));
    out[3] = refBits;
}

// The 8-wide compressed form (trace.h: node8), one thread per node of the 4-wide tree, each on its own: node i gathers its
// children, then, largest box first, replaces an inner child by that child's children while the total stays within eight.
// Whatever the other threads decide, the refs it ends up with are nodes of the 4-wide tree, and those become 8-wide nodes
// the same way under their own index; a node that was pulled into its parent is computed too and never visited.  The tree
// gets no deeper.  Grid and rounding as in k_compress_nodes.
struct WidenEntry {
    float lo[3], hi[3];
    int ref;
    int expandable;
};

__device__ inline int widenLoadChildren(const float4 *nodes, int index, WidenEntry *out)
{
    const float4 *node = nodes + (size_t)8 * index;
    const float4 refBits = node[6];
    const int ref[4] = { floatAsInt(refBits.x), floatAsInt(refBits.y), floatAsInt(refBits.z), floatAsInt(refBits.w) };
    float lo[3][4], hi[3][4];
    for (int a = 0; a < 3; a++) {
        const float4 l = node[a], h = node[3 + a];
        lo[a][0] = l.x; lo[a][1] = l.y; lo[a][2] = l.z; lo[a][3] = l.w;
        hi[a][0] = h.x; hi[a][1] = h.y; hi[a][2] = h.z; hi[a][3] = h.w;
    }
    int count = 0;
    for (int c = 0; c < 4; c++) {
        if (ref[c] == kEmptyChild) { continue; }
        for (int a = 0; a < 3; a++) { out[count].lo[a] = lo[a][c]; out[count].hi[a] = hi[a][c]; }
        out[count].ref = ref[c];
        out[count].expandable = ref[c] >= 0 ? 1 : 0;
        count++;
    }
    return count;
}

__global__ __launch_bounds__(kBlock) void k_widen_nodes(const float4 *nodes, int nNodes, float4 *nodes8)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nNodes) { return; }
    WidenEntry entry[8];
    int n = widenLoadChildren(nodes, i, entry);
    while (true) {
        int best = -1;
        float bestArea = -1.f;
        for (int k = 0; k < n; k++) {
            if (!entry[k].expandable) { continue; }
            const float dx = entry[k].hi[0] - entry[k].lo[0], dy = entry[k].hi[1] - entry[k].lo[1], dz = entry[k].hi[2] - entry[k].lo[2];
            const float area = dx * dy + dy * dz + dz * dx;
            if (best < 0 || area > bestArea) { best = k; bestArea = area; }
        }
        if (best < 0) { break; }
        WidenEntry kids[4];
        const int m = widenLoadChildren(nodes, entry[best].ref, kids);
        if (m == 0 || n - 1 + m > 8) { entry[best].expandable = 0; continue; }
        entry[best] = kids[0];
        for (int k = 1; k < m; k++) { entry[n++] = kids[k]; }
    }

    float origin[3], scale[3];
    unsigned int qlo[3][2] = { { 0u, 0u }, { 0u, 0u }, { 0u, 0u } }, qhi[3][2] = { { 0u, 0u }, { 0u, 0u }, { 0u, 0u } };
    for (int a = 0; a < 3; a++) {
        float low = INFINITY, high = -INFINITY;
        for (int k = 0; k < n; k++) { low = fminf(low, entry[k].lo[a]); high = fmaxf(high, entry[k].hi[a]); }
        if (!(low <= high)) { low = 0.f; high = 0.f; }
        const double extent = (double)high - (double)low;
        int exponent = 0;
        (void)frexp(extent / 253.0, &exponent);
        const float step = extent > 0.0 ? ldexpf(1.f, exponent) : 0.f;
        const double perStep = step > 0.f ? 1.0 / (double)step : 0.0;
        origin[a] = low;
        scale[a] = step;
        for (int k = 0; k < n; k++) {
            const double below = floor(((double)entry[k].lo[a] - (double)low) * perStep - 1.0 / 256.0);
            const double above = ceil(((double)entry[k].hi[a] - (double)low) * perStep + 1.0 / 256.0);
            const unsigned int qBelow = (unsigned int)fmin(fmax(below, 0.0), 255.0);
            const unsigned int qAbove = step > 0.f ? (unsigned int)fmin(fmax(above, 0.0), 255.0) : 0u;
            qlo[a][k >> 2] |= qBelow << (8 * (k & 3));
            qhi[a][k >> 2] |= qAbove << (8 * (k & 3));
        }
    }
    int ref[8];
    for (int k = 0; k < 8; k++) { ref[k] = k < n ? entry[k].ref : kEmptyChild; }
    float4 *out = nodes8 + (size_t)8 * i;
    #define PATHED_BITS(x) intAsFloat((int)(x))
    out[0] = make_float4(origin[0], origin[1], origin[2], scale[0]);
    out[1] = make_float4(scale[1], scale[2], PATHED_BITS(qlo[0][0]), PATHED_BITS(qlo[0][1]));
    out[2] = make_float4(PATHED_BITS(qlo[1][0]), PATHED_BITS(qlo[1][1]), PATHED_BITS(qlo[2][0]), PATHED_BITS(qlo[2][1]));
    out[3] = make_float4(PATHED_BITS(qhi[0][0]), PATHED_BITS(qhi[0][1]), PATHED_BITS(qhi[1][0]), PATHED_BITS(qhi[1][1]));
    out[4] = make_float4(PATHED_BITS(qhi[2][0]), PATHED_BITS(qhi[2][1]), 0.f, 0.f);
    out[5] = make_float4(PATHED_BITS(ref[0]), PATHED_BITS(ref[1]), PATHED_BITS(ref[2]), PATHED_BITS(ref[3]));
    out[6] = make_float4(PATHED_BITS(ref[4]), PATHED_BITS(ref[5]), PATHED_BITS(ref[6]), PATHED_BITS(ref[7]));
    out[7] = make_float4(0.f, 0.f, 0.f, 0.f);
    #undef PATHED_BITS
}

// The same probe with its own clocks (pathed_hip_measure_valu_clocks): CHAINS independent v_fma_f32 chains per lane on
// three VGPR operands, and every wave reads the shader clock (s_memtime) and the constant-rate wall clock (s_memrealtime)
// around its loop.  cycles per instruction = shader-clock ticks / instructions issued, per wave, whatever frequency the
// chip ran at under this load; ticks / wall ticks x the wall-clock rate is that frequency.
template <int CHAINS>
__global__ __launch_bounds__(kBlock) void k_valu_clock_probe(int iterations, float seed, float *sink, unsigned long long *clocks)
{
    static_assert(CHAINS == 8 || CHAINS == 16, "8 or 16 chains");
    float a[16];
    #pragma unroll
    for (int k = 0; k < 16; k++) { a[k] = seed + threadIdx.x + k; }
    const float m = 0.999f, c = 0.001f;
    const unsigned long long shaderStart = __builtin_amdgcn_s_memtime();
    const unsigned long long wallStart = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iterations; i++) {
        #pragma unroll
        for (int k = 0; k < kValuProbeUnroll / 16; k++) {
            if (CHAINS == 16) {
                asm volatile(
                    "v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"
                    "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n"
                    "v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"
                    "v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                      "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
                    : "v"(m), "v"(c));
            } else {
                asm volatile(
                    "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                    "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                    "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                    "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m), "v"(c));
            }
        }
    }
    const unsigned long long shaderTicks = __builtin_amdgcn_s_memtime() - shaderStart;
    const unsigned long long wallTicks = __builtin_amdgcn_s_memrealtime() - wallStart;
    if ((threadIdx.x & 63) == 0) {
        const size_t wave = (size_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
        clocks[2 * wave + 0] = shaderTicks;
        clocks[2 * wave + 1] = wallTicks;
    }
    float total = 0.f;
    #pragma unroll
    for (int k = 0; k < 16; k++) { total += a[k]; }
    if (total == 12345.678f) { *sink = total; }   // keeps the chains alive
}


}  // namespace pathed
