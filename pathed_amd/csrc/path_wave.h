// k_path_wave: the path tracer over a BVH with NO path state in HBM (design (B) of the round-3 review: "path state resident
// on chip, a wave alternates a traversal burst and a shade burst over its own paths").
//
// A persistent wave owns 64 paths, one per lane, in registers -- k_path_small's organisation -- but the RAYS of the wave are
// common property: after a shade burst every lane posts its continuation ray and its shadow ray to the wave's list in LDS,
// and in the traversal burst the lanes draw rays from that list the way k_trace's lanes draw them from a staged card
// (ray cost is heavy-tailed: one lane per path with no refill runs at 0.46 lane utilisation).  A
// finished ray's result goes to its owner through LDS.  Once the list is empty and fewer than p.suspendLanes rays are
// still in flight the burst ends: those rays stay on their lanes (registers + the lane's stack rows) across the shade
// burst, the paths that wait for them sit it out, and the next burst carries them on together with the new rays.
//
// Every operation on a path's values is k_shade's, in k_shade's order (pathVertex below is k_path_small's vertex code), the
// traversal is k_trace's (trace.h), and the unit decomposition fixes the summation order: images are the wavefront
// kernels' bit for bit.
//
// Included by kernels.h inside namespace pathed.

struct PathRegisters {
    V3 o, d;                         // the ray in flight (the one whose hit the next vertex shades)
    int st;                          // device_scene.h state word: vertex that spawned the ray + eligible / delta / continue
    int firstEmitMaterial;
    Rgb result, modulation, throughput, pend;
    float bsdfPdf, cosTheta;
    Rng random;
};

// One vertex of one path: SampleIntegrator::samplePixel / PathTracer::L on register state -- the code of k_path_small's loop
// body between "the vertex" and "end of a sample" (kernels.h), as a function.  `h` is the hit of the ray (path.o, path.d).
// Returns true when the sample is finished (*color is its value); otherwise path.o / path.d hold the next ray and *shadow
// the vertex's occlusion query, if any.
template <typename TRAITS, typename MATERIALS>
__device__ __forceinline__ bool pathVertex(const RenderParams &p, const DScene &scene, const MATERIALS &materials, PathRegisters &path,
                                           float4 h, ShadowRequest *shadowOut, Rgb *color)
{
    ShadowRequest shadow;
    shadow.push = false;
    shadow.origin = v3(0.f, 0.f, 0.f);
    shadow.direction = v3(0.f, 0.f, 1.f);
    shadow.tfar = 0.f;
    bool finished = false;
    *color = rgb(0.f);
    const bool miss = floatAsInt(h.w) < 0;
    const int st = path.st;
    const int rayBounce = st & kStBounceMask;  // vertex that spawned this ray, 0 = camera
    bool haveVertex = false;
    Isect isect;
    const int vertex = rayBounce + 1;
    if (!miss) { isect = makeIsect<TRAITS>(scene, path.o, path.d, h); }

    if (rayBounce == 0) {
        // SampleIntegrator::samplePixel, src/sample_integrator.cpp:18-59
        if (miss) {
            *color = rgb(0.f) + environmentL<TRAITS>(scene, path.d);
            finished = true;
        } else {
            path.firstEmitMaterial = -1;
            if (checkCounts(p.startBounce, p.lastBounce, 0)) {
                const Rgb emit = matEmit(materials[isect.material]);
                const bool backside = dot(isect.normal, isect.wo) < 0.f;
                if (!isBlack(emit) && !backside) { path.firstEmitMaterial = isect.material; }
            }
            path.result = rgb(0.f);
            haveVertex = true;
        }
    } else {
        // the ray left vertex `rayBounce` along its BSDF sample
        if (st & kStEligible) {
            // PathTracer::directSampleBSDF, src/path_tracer.cpp:167-216
            Rgb bsdfTerm = rgb(0.f);
            if (!miss) {
                const Rgb emit = matEmit(materials[isect.material]);
                if (!isBlack(emit) && dot(isect.wo, isect.shadingNormal) >= 0.f) {
                    const float lightPDF = lightsPDF<TRAITS>(scene, path.o, isect);
                    const float brdfWeight = (st & kStDelta)
                        ? 1.f
                        : (1 * path.bsdfPdf) / (1 * path.bsdfPdf + 1 * lightPDF);
                    bsdfTerm = emit * brdfWeight * path.throughput * path.cosTheta / path.bsdfPdf;
                }
            } else {
                const Rgb environmentLight = environmentL<TRAITS>(scene, path.d);
                if (TRAITS::env && !isBlack(environmentLight)) {
                    // Scene::environmentPDF, src/scene.cpp:494-502
                    const float lightPDF = envEmitPDF(scene.env, path.d) / scene.nLights;
                    const float brdfWeight = (st & kStDelta)
                        ? 1.f
                        : (1 * path.bsdfPdf) / (1 * path.bsdfPdf + 1 * lightPDF);
                    bsdfTerm = environmentLight * brdfWeight * path.throughput * path.cosTheta / path.bsdfPdf;
                }
            }
            const Rgb Ld = path.pend + bsdfTerm;
            if (rayBounce == 1) { path.result = Ld; }
            else { path.result = path.result + Ld * path.modulation; }
        }

        // PathTracer::L loop body, src/path_tracer.cpp:41-58
        if (!(st & kStContinue) || miss) {
            finished = true;
        } else {
            const float invPDF = 1.f / path.bsdfPdf;
            path.modulation = path.modulation * (path.throughput * path.cosTheta * invPDF);
            if (isBlack(path.modulation)) { finished = true; }
            else { haveVertex = true; }
        }
        if (finished) {
            Rgb first = rgb(0.f);
            if (path.firstEmitMaterial >= 0) { first = first + matEmit(materials[path.firstEmitMaterial]); }
            *color = first + path.result;
        }
    }

    if (haveVertex) {
        // PathTracer::L: sample the BSDF, then direct(), src/path_tracer.cpp:30-36, 60-73
        const DMaterial &material = materials[isect.material];
        prepareLobes<TRAITS>(material, isect);

        path.random.dimension = vertexBase(vertex);
        const BSDFSample bsdfSample = materialSample<TRAITS>(material, isect, path.random);

        const bool counts = checkCounts(p.startBounce, p.lastBounce, vertex);
        const bool emissive = !isBlack(matEmit(material));
        const bool wantDirect = counts && !emissive;  // direct() returns 0 on emitters (:86-90)
        const bool wantContinue = !checkDone(p.lastBounce, vertex + 1);

        Rgb lightTerm = rgb(0.f);
        if (wantDirect) {
            path.random.dimension = vertexBase(vertex) + 3;
            lightTerm = sampleLightsTerm<false, TRAITS>(scene, materials, isect, material, path.random, &shadow);
        }

        // see k_shade: a vertex with nothing pending whose BSDF sample has exactly black throughput ends the sample
        const bool deadEnd = isBlack(bsdfSample.throughput) && bsdfSample.pdf > 0.f && bsdfSample.pdf < 3e38f
            && !shadow.push && isBlack(lightTerm);
        if ((!wantDirect && !wantContinue) || deadEnd) {
            finished = true;
            Rgb first = rgb(0.f);
            if (path.firstEmitMaterial >= 0) { first = first + matEmit(materials[path.firstEmitMaterial]); }
            *color = first + path.result;
            shadow.push = false;
        } else {
            int nextState = vertex;
            if (wantDirect) { nextState |= kStEligible; }
            if (isDeltaT<TRAITS>(material)) { nextState |= kStDelta; }
            if (wantContinue) { nextState |= kStContinue; }
            path.st = nextState;
            path.o = isect.point;
            path.d = bsdfSample.wiWorld;
            path.bsdfPdf = bsdfSample.pdf;
            path.throughput = bsdfSample.throughput;
            path.cosTheta = fabsf(dot(isect.shadingNormal, bsdfSample.wiWorld));
            path.pend = lightTerm;
        }
    }
    *shadowOut = shadow;
    return finished;
}

#ifndef PATHED_WAVE_WAVES
#define PATHED_WAVE_WAVES 3   // blocks per CU = waves per SIMD: what 160 KB of LDS holds of stacks + ray lists (168 VGPRs each)
#endif
static const int kWaveListRays = 128;                 // a wave's ray list: at most one continuation + one shadow ray per lane
static const int kWaveRing = kWavesPerBlock * kWaveListRays;   // BLOCK: the block's ray ring (every path has at most two rays in it)
static const unsigned int kWaveSpinBound = 1u << 22;  // BLOCK: polls (of ~0.2 us) before a wave gives up on the others: a bug, reported as dropped samples
static const unsigned int kWaveShadowTag = 0x100u;    // list entry / lane target: owner lane | this bit for a shadow ray
static const unsigned int kWaveOccludedFlag = 0x100u; // ownerFlags: the low byte counts the owner's finished rays

// dynamic LDS of a block: [STACK + 1][kBlock] stack rows | per wave 128 x 2 float4 of ray list | kBlock float4 hits |
// kBlock flag words | (BLOCK) the ring's kWaveRing + 4 words | (LDS_MATERIALS) the material table
// (22 stack rows, no ring: 44 KiB + the material table, 9 KiB at most: three blocks per CU)
__host__ __device__ inline size_t pathWaveLdsBytes(int stackRows, int nLdsMaterials, bool blockRing)
{
    return (size_t)(stackRows + 1) * kBlock * sizeof(int) + (size_t)kWavesPerBlock * kWaveListRays * 2 * sizeof(float4)
        + (size_t)kBlock * sizeof(float4) + (size_t)kBlock * sizeof(unsigned int) + (blockRing ? (size_t)(kWaveRing + 4) * sizeof(unsigned int) : 0)
        + (size_t)nLdsMaterials * sizeof(DMaterial);
}

// BLOCK (instantiated in the experiments build only: 13-20 % slower, profiles/r4_ab_wave.log -- with one path per lane there
// is about one ray per lane in the whole block, shared or not, so the idle lanes find the ring empty and the polling costs):
// the four waves of a block share ONE ray queue (a ring in LDS, every path has at most two rays in it) and work
// asynchronously: a wave posts its new rays there, any wave's idle lanes draw from it, and a wave shades when enough of its
// paths have all their rays back (p.parkMinCardsPerWave of them) or it has nothing to traverse.  A wave's own list holds
// fewer rays than it has lanes (one continuation ray per path, a shadow ray for a quarter of them): its bursts are all tail.
template <bool LDS_MATERIALS, int STACK, typename TRAITS, bool SPHERES, bool BLOCK = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PATHED_WAVE_WAVES, PATHED_WAVE_WAVES))) void k_path_wave(RenderParams p)
{
    extern __shared__ float4 ldsRaw[];
    LaneStack stack;
    stack.lds = reinterpret_cast<int *>(ldsRaw) + threadIdx.x;
    stack.overflowStride = (size_t)gridDim.x * kBlock;
    stack.overflow = p.stackOverflow + ((size_t)blockIdx.x * kBlock + threadIdx.x);
    const int lane = threadIdx.x & 63;
    const int wave = (int)(threadIdx.x >> 6);
    // BLOCK: one ring of kWaveRing records for the block, results and flags indexed by the owner's thread
    float4 *listO = ldsRaw + ((STACK + 1) * kBlock) / 4 + (BLOCK ? 0 : wave * (2 * kWaveListRays));   // .w = tfar
    float4 *listD = listO + (BLOCK ? kWaveRing : kWaveListRays);                                        // .w = owner | kWaveShadowTag
    float4 *hitRows = ldsRaw + ((STACK + 1) * kBlock) / 4 + kWavesPerBlock * 2 * kWaveListRays + (BLOCK ? 0 : wave * 64);
    unsigned int *ownerFlags = reinterpret_cast<unsigned int *>(ldsRaw + ((STACK + 1) * kBlock) / 4 + kWavesPerBlock * 2 * kWaveListRays + kBlock) + (BLOCK ? 0 : wave * 64);
    unsigned int *ringFlags = reinterpret_cast<unsigned int *>(ldsRaw + ((STACK + 1) * kBlock) / 4 + kWavesPerBlock * 2 * kWaveListRays + kBlock + kBlock / 4);   // [kWaveRing]: ticket + 1 once the record is written
    unsigned int *ringHead = ringFlags + kWaveRing, *ringTail = ringHead + 1;
    const int self = BLOCK ? (int)threadIdx.x : lane;   // this lane's row of hitRows / ownerFlags
    const unsigned int ownerMask = BLOCK ? 0xFFu : 0x3Fu;
    if (BLOCK) {
        for (int i = threadIdx.x; i < kWaveRing + 4; i += kBlock) { ringFlags[i] = 0u; }
        __syncthreads();
    }

    MaterialAccess<LDS_MATERIALS> materials;
    if (LDS_MATERIALS) {
        const int words = p.scene.nMaterials * (int)(sizeof(DMaterial) / 4);
        const int *source = reinterpret_cast<const int *>(p.scene.materials);
        int *target = reinterpret_cast<int *>(ldsRaw + ((STACK + 1) * kBlock) / 4 + kWavesPerBlock * 2 * kWaveListRays + kBlock + kBlock / 4 + (BLOCK ? (kWaveRing + 4) / 4 : 0));
        for (int i = threadIdx.x; i < words; i += kBlock) { target[i] = source[i]; }
        __syncthreads();
        materials.table = reinterpret_cast<const DMaterial *>(target);
    } else {
        materials.table = p.scene.materials;
    }

    TraceGeometry geometry;
    geometry.nodes = p.scene.nodes;
    geometry.tris = p.scene.leafTris;
    geometry.nNodes = p.scene.nNodes;
    geometry.nTris = p.scene.nTris;
    geometry.spheres = p.scene.spheres;
    geometry.nSpheres = p.scene.nLinearSpheres;

    const DScene &scene = p.scene;
    const unsigned int waveId = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const uint64_t seed = ((uint64_t)p.seedHi << 32) | p.seedLo;

    // ---- work units: as k_path_small
    unsigned int queue = waveId % (unsigned int)p.nQueues, queuesTried = 0;
    unsigned int reservedNext = 0, reservedEnd = 0;
    auto takeUnits = [&](bool want) -> unsigned int {
        unsigned int mine = 0xFFFFFFFFu;
        unsigned long long wanting = __ballot(want);
        while (wanting != 0ull) {
            if (reservedNext == reservedEnd) {
                if (queuesTried >= (unsigned int)p.nQueues) { break; }   // every queue is dealt out
                unsigned int ticket = 0;
                if (lane == 0) { ticket = atomicAdd(&p.counters[kCtrUnitCursor + queue * kCursorStride], (unsigned int)p.unitGrab); }
                ticket = (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
                const unsigned int limit = p.queueUnits[queue];
                if (ticket >= limit) {
                    queue = (queue + 1u) % (unsigned int)p.nQueues;
                    queuesTried++;
                    continue;
                }
                reservedNext = ticket;
                reservedEnd = ticket + (unsigned int)p.unitGrab < limit ? ticket + (unsigned int)p.unitGrab : limit;
            }
            const unsigned int available = reservedEnd - reservedNext;
            const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(wanting >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)wanting, 0u));
            const bool served = ((wanting >> lane) & 1ull) != 0ull && rank < available;
            if (served) { mine = queue * p.unitsPerQueue + reservedNext + rank; }
            const unsigned int count = (unsigned int)__popcll(wanting);
            reservedNext += count < available ? count : available;
            wanting &= ~__ballot(served);
        }
        return mine;
    };

    // ---- the path a lane owns
    bool alive = false;
    unsigned int unit = 0xFFFFFFFFu;
    uint32_t pixel = 0, sample = 0, endSample = 0;
    PathRegisters path;
    path.random.k0 = 0u; path.random.k1 = 0u; path.random.dimension = 0u;
    path.o = v3(0.f, 0.f, 0.f); path.d = v3(0.f, 0.f, 1.f);
    path.st = 0;
    path.firstEmitMaterial = -1;
    path.result = rgb(0.f); path.modulation = rgb(1.f); path.throughput = rgb(0.f); path.pend = rgb(0.f);
    path.bsdfPdf = 1.f; path.cosTheta = 0.f;
    float4 partial = make_float4(0.f, 0.f, 0.f, 0.f);
    bool fresh = false;              // path.o / path.d (and the shadow request) are new: post them
    bool pendingShadow = false;
    V3 shadowDirection = v3(0.f, 0.f, 1.f);
    float shadowTfar = 0.f;
    unsigned int waiting = 0;        // rays of this lane's path that are posted and not yet collected (0, 1 or 2)

    // ---- the ray a lane traverses (anybody's)
    LaneRay ray;
    laneRayInit(ray, path.o, path.d, PATHED_TNEAR, PATHED_TFAR, false);
    bool active = false;
    unsigned int target = 0;         // owner lane | kWaveShadowTag

#ifdef PATHED_SHADE_PROFILE
    // (tuning builds, tools/wave_profile.py) where a wave's time goes and how full its bursts are
    unsigned long long profIterations = 0, profAlive = 0, profSteps = 0, profLaneSteps = 0, profShades = 0, profReady = 0, profPosted = 0, profLeft = 0;
    unsigned long long profBurstCycles = 0, profShadeCycles = 0, profStart = __builtin_amdgcn_s_memtime(), profStamp = 0;
#endif
    bool startNext = false;
    {
        unit = takeUnits(true);
        if (unit != 0xFFFFFFFFu) {
            unitSamples(p, unit, &pixel, &sample, &endSample);
            alive = true;
            startNext = true;
        }
    }

    while (true) {
        if (startNext) {
            makeKey(seed, pixel, sample, &path.random.k0, &path.random.k1);
            path.random.dimension = 0;
            const int width = scene.camera.resX;
            const int row = (int)fastDivide((unsigned int)pixel, p.divWidth);
            const int col = (int)pixel - row * width;
            const float jitterX = path.random.next() - 0.5f;
            const float jitterY = path.random.next() - 0.5f;
            cameraRay(scene.camera, row + jitterY, col + jitterX, &path.o, &path.d);
            path.st = 0;
            path.firstEmitMaterial = -1;
            path.result = rgb(0.f);
            path.modulation = rgb(1.f);
            path.throughput = rgb(0.f);
            path.pend = rgb(0.f);
            path.bsdfPdf = 1.f;
            path.cosTheta = 0.f;
            startNext = false;
            fresh = true;
            pendingShadow = false;
        }
        // (a wave's own rays are all back when its paths are dead; BLOCK: it may still hold other waves' rays)
        if (__ballot(alive || (BLOCK && active)) == 0ull) { break; }

#ifdef PATHED_SHADE_PROFILE
        profIterations++; profAlive += (unsigned long long)__popcll(__ballot(alive));
#endif
        // ---- post the new rays: continuation rays first, then the shadow rays (which leave the same point)
        unsigned int listCount = 0, listPos = 0;   // wave-uniform
        if (BLOCK) {
            const bool postClosest = alive && fresh;
            const bool postShadow = postClosest && pendingShadow;
            const unsigned long long closestMask = __ballot(postClosest), shadowMask = __ballot(postShadow);
            const unsigned int count = (unsigned int)(__popcll(closestMask) + __popcll(shadowMask));
            if (count != 0u) {
                unsigned int base = 0u;
                if (lane == 0) { base = atomicAdd(ringTail, count); }
                base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                const unsigned int closestTicket = base + __builtin_amdgcn_mbcnt_hi((unsigned int)(closestMask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)closestMask, 0u));
                const unsigned int shadowTicket = base + (unsigned int)__popcll(closestMask)
                    + __builtin_amdgcn_mbcnt_hi((unsigned int)(shadowMask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)shadowMask, 0u));
                if (postClosest) {
                    ownerFlags[self] = 0u;
                    waiting = postShadow ? 2u : 1u;
                    listO[closestTicket % kWaveRing] = make_float4(path.o.x, path.o.y, path.o.z, PATHED_TFAR);
                    listD[closestTicket % kWaveRing] = make_float4(path.d.x, path.d.y, path.d.z, intAsFloat(self));
                }
                if (postShadow) {
                    listO[shadowTicket % kWaveRing] = make_float4(path.o.x, path.o.y, path.o.z, shadowTfar);
                    listD[shadowTicket % kWaveRing] = make_float4(shadowDirection.x, shadowDirection.y, shadowDirection.z, intAsFloat(self | (int)kWaveShadowTag));
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (postClosest) { *(volatile unsigned int *)&ringFlags[closestTicket % kWaveRing] = closestTicket + 1u; }
                if (postShadow) { *(volatile unsigned int *)&ringFlags[shadowTicket % kWaveRing] = shadowTicket + 1u; }
            }
            fresh = false;
            pendingShadow = false;
#ifdef PATHED_SHADE_PROFILE
            profPosted += count; profStamp = __builtin_amdgcn_s_memtime();
#endif
        } else {
            const bool postClosest = alive && fresh;
            const bool postShadow = postClosest && pendingShadow;
            const unsigned long long closestMask = __ballot(postClosest), shadowMask = __ballot(postShadow);
            const unsigned int closestRank = __builtin_amdgcn_mbcnt_hi((unsigned int)(closestMask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)closestMask, 0u));
            const unsigned int shadowRank = (unsigned int)__popcll(closestMask)
                + __builtin_amdgcn_mbcnt_hi((unsigned int)(shadowMask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)shadowMask, 0u));
            if (postClosest) {
                listO[closestRank] = make_float4(path.o.x, path.o.y, path.o.z, PATHED_TFAR);
                listD[closestRank] = make_float4(path.d.x, path.d.y, path.d.z, intAsFloat(lane));
                ownerFlags[lane] = 0u;
                waiting = postShadow ? 2u : 1u;
            }
            if (postShadow) {
                listO[shadowRank] = make_float4(path.o.x, path.o.y, path.o.z, shadowTfar);
                listD[shadowRank] = make_float4(shadowDirection.x, shadowDirection.y, shadowDirection.z, intAsFloat(lane | (int)kWaveShadowTag));
            }
            fresh = false;
            pendingShadow = false;
            listCount = (unsigned int)(__popcll(closestMask) + __popcll(shadowMask));
#ifdef PATHED_SHADE_PROFILE
            profPosted += listCount; profStamp = __builtin_amdgcn_s_memtime();
#endif
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }

        // ---- BLOCK: traverse rays of the block's ring until this wave has paths to shade
        if (BLOCK) {
            unsigned int spins = 0u;
            while (true) {
                // idle lanes claim records of the ring
                const unsigned long long idleMask = __ballot(!active);
                bool ringEmpty = false;
                if (idleMask != 0ull) {
                    const unsigned int wanted = (unsigned int)__popcll(idleMask);
                    unsigned int claimedBase = 0u, claimed = 0u;
                    if (lane == 0) {
                        unsigned int head = *(volatile unsigned int *)ringHead;
                        while (true) {
                            const unsigned int available = *(volatile unsigned int *)ringTail - head;
                            const unsigned int take = wanted < available ? wanted : available;
                            if (take == 0u) { break; }
                            const unsigned int seen = atomicCAS(ringHead, head, head + take);
                            if (seen == head) { claimedBase = head; claimed = take; break; }
                            head = seen;
                        }
                    }
                    claimedBase = (unsigned int)__builtin_amdgcn_readfirstlane((int)claimedBase);
                    claimed = (unsigned int)__builtin_amdgcn_readfirstlane((int)claimed);
                    ringEmpty = claimed < wanted;
                    const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(idleMask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)idleMask, 0u));
                    if (!active && rank < claimed) {
                        const unsigned int ticket = claimedBase + rank, slot = ticket % kWaveRing;
                        unsigned int polls = 0u;
                        while (*(volatile unsigned int *)&ringFlags[slot] != ticket + 1u && polls < kWaveSpinBound) { polls++; }   // (written right after the ticket was drawn)
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        const float4 first = listO[slot];
                        const float4 second = listD[slot];
                        // (a record is overwritten 512 tickets later, when its ray has long been claimed and read; a reader that found
                        // another ticket's flag would be a bug: reported, not traced)
                        if (polls >= kWaveSpinBound) {
                            atomicAdd(&p.stats[kStatDropped], 1ull << 40);
                        } else {
                            target = (unsigned int)floatAsInt(second.w);
                            laneRayInit(ray, v3(first.x, first.y, first.z), v3(second.x, second.y, second.z), PATHED_TNEAR, first.w, (target & kWaveShadowTag) != 0u);
                            active = true;
                        }
                    }
                }
                const int inFlight = __popcll(__ballot(active));
                const bool complete = alive && waiting != 0u && (*(volatile unsigned int *)&ownerFlags[self] & 0xFFu) == waiting;
                const int nReady = __popcll(__ballot(complete));
                if (nReady >= p.parkMinCardsPerWave) { break; }
                if (inFlight == 0) {
                    if (nReady > 0 || __ballot(alive) == 0ull) { break; }
                    // this wave's rays are on other waves' lanes
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > kWaveSpinBound) {
                        if (lane == 0) { atomicAdd(&p.stats[kStatDropped], 1ull << 40); }
                        alive = false;
                        break;
                    }
                    continue;
                }
                if (ringEmpty && inFlight < p.suspendLanes && nReady > 0 && nReady >= 2 * inFlight) { break; }
                // a few steps, then look again (a full wave keeps going until it has thinned out)
                for (int step = 0; step < 16; step++) {
                    const unsigned long long leafMask = __ballot(active && ray.pendingLeaf != 0);
                    const unsigned long long innerMask = __ballot(active && ray.pendingLeaf == 0);
                    const bool trianglePhase = __popcll(leafMask) >= kLeafThreshold || innerMask == 0ull;
#ifdef PATHED_SHADE_PROFILE
                    profSteps++; profLaneSteps += (unsigned long long)__popcll(trianglePhase ? leafMask : innerMask);
#endif
                    bool done = false;
                    if (trianglePhase) {
                        if (active && ray.pendingLeaf != 0) { done = leafStep<false, STACK, kBlock, SPHERES>(geometry, stack, ray, nullptr); }
                    } else {
                        if (active && ray.pendingLeaf == 0) {
                            done = (geometry.nNodes == 0) || innerStep<false, STACK, kBlock, false, false>(geometry, stack, p.maxStack, ray, nullptr);
                        }
                    }
                    if (done) {
                        finishRay<SPHERES>(geometry, ray);
                        const unsigned int owner = target & ownerMask;
                        if (ray.anyHit) {
                            atomicAdd(&ownerFlags[owner], ray.occluded ? (1u + kWaveOccludedFlag) : 1u);
                        } else {
                            hitRows[owner] = make_float4(ray.best, ray.bestU, ray.bestV, intAsFloat(ray.bestPrim));
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            atomicAdd(&ownerFlags[owner], 1u);
                        }
                        active = false;
                    }
                    const int busy = __popcll(__ballot(active));
                    if (busy == 0 || (busy < p.suspendPatience && step >= 1)) { break; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        // ---- traversal burst (k_trace's loop over the wave's own list)
        // The list is dealt out and few rays are still in flight: the burst ends if the shade burst has something to do --
        // at least twice as many paths with all their rays back as rays in flight (so every burst ends with progress, and the
        // end of a render, where few paths are left, does not shade one lane at a time).
        auto leaveStragglers = [&](unsigned long long activeMask) -> bool {
            const int inFlight = __popcll(activeMask);
            if (inFlight >= p.suspendLanes) { return false; }
            const bool complete = alive && waiting != 0u && (*(volatile unsigned int *)&ownerFlags[lane] & 0xFFu) == waiting;
            return __popcll(__ballot(complete)) >= 2 * inFlight;
        };
        while (!BLOCK) {
            // hand rays to the idle lanes
            while (listPos < listCount) {
                const unsigned long long idleMask = __ballot(!active);
                if (idleMask == 0ull) { break; }
                const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(idleMask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)idleMask, 0u));
                const unsigned int available = listCount - listPos;
                if (!active && rank < available) {
                    const float4 first = listO[listPos + rank];
                    const float4 second = listD[listPos + rank];
                    target = (unsigned int)floatAsInt(second.w);
                    const bool shadowRay = (target & kWaveShadowTag) != 0u;
                    laneRayInit(ray, v3(first.x, first.y, first.z), v3(second.x, second.y, second.z), PATHED_TNEAR, first.w, shadowRay);
                    active = true;
                }
                const unsigned int wanted = (unsigned int)__popcll(idleMask);
                listPos += wanted < available ? wanted : available;
            }
            const bool dry = listPos == listCount;
            {
                const unsigned long long activeMask = __ballot(active);
                if (activeMask == 0ull) { break; }
                if (dry && leaveStragglers(activeMask)) { break; }
            }
            // steps until the wave has thinned out enough to be worth refilling
            while (true) {
                const unsigned long long leafMask = __ballot(active && ray.pendingLeaf != 0);
                const unsigned long long innerMask = __ballot(active && ray.pendingLeaf == 0);
                const bool trianglePhase = __popcll(leafMask) >= kLeafThreshold || innerMask == 0ull;
#ifdef PATHED_SHADE_PROFILE
                profSteps++; profLaneSteps += (unsigned long long)__popcll(trianglePhase ? leafMask : innerMask);
#endif
                bool done = false;
                if (trianglePhase) {
                    if (active && ray.pendingLeaf != 0) { done = leafStep<false, STACK, kBlock, SPHERES>(geometry, stack, ray, nullptr); }
                } else {
                    if (active && ray.pendingLeaf == 0) {
                        done = (geometry.nNodes == 0) || innerStep<false, STACK, kBlock, false, false>(geometry, stack, p.maxStack, ray, nullptr);
                    }
                }
                if (done) {
                    finishRay<SPHERES>(geometry, ray);
                    const unsigned int owner = target & 63u;
                    if (ray.anyHit) {
                        atomicAdd(&ownerFlags[owner], ray.occluded ? (1u + kWaveOccludedFlag) : 1u);
                    } else {
                        hitRows[owner] = make_float4(ray.best, ray.bestU, ray.bestV, intAsFloat(ray.bestPrim));
                        atomicAdd(&ownerFlags[owner], 1u);
                    }
                    active = false;
                }
                const unsigned long long activeMask = __ballot(active);
                if (activeMask == 0ull) { break; }
                if (!dry && __popcll(activeMask) < p.suspendPatience) { break; }   // (here: the refill threshold)
                if (dry && leaveStragglers(activeMask)) { break; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

#ifdef PATHED_SHADE_PROFILE
        profLeft += (unsigned long long)__popcll(__ballot(active));
        { const unsigned long long now = __builtin_amdgcn_s_memtime(); profBurstCycles += now - profStamp; profStamp = now; }
#endif
        // ---- shade burst: the paths whose rays have all come back
        bool ready = false;
        float4 h = make_float4(0.f, 0.f, 0.f, intAsFloat(-1));
        if (alive && waiting != 0u) {
            const unsigned int flags = *(volatile unsigned int *)&ownerFlags[self];
            if ((flags & 0xFFu) == waiting) {
                ready = true;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // the hit was written before the count moved
                h = hitRows[self];
                if (flags & kWaveOccludedFlag) { path.pend = rgb(0.f); }
                waiting = 0u;
            }
        }
#ifdef PATHED_SHADE_PROFILE
        { const unsigned long long mask = __ballot(ready); if (mask != 0ull) { profShades++; profReady += (unsigned long long)__popcll(mask); } }
#endif
        bool finished = false;
        Rgb color = rgb(0.f);
        if (ready) {
            ShadowRequest shadow;
            finished = pathVertex<TRAITS>(p, scene, materials, path, h, &shadow, &color);
            if (!finished) {
                fresh = true;
                if (shadow.push) {
                    pendingShadow = true;
                    shadowDirection = shadow.direction;
                    shadowTfar = shadow.tfar;
                }
            }
        }

        // ---- end of a sample (k_path_small)
        bool needUnit = false;
        if (ready && finished) {
            const bool finite = isfinite(color.r) && isfinite(color.g) && isfinite(color.b);
            if (finite) {
                partial.x += color.r;
                partial.y += color.g;
                partial.z += color.b;
            } else {
                atomicAdd(&p.stats[kStatDropped], 1ull);
            }
            sample++;
            if (sample < endSample) {
                startNext = true;
            } else {
                p.state.chunkBuf[partialIndex(p, unit)] = partial;
                partial = make_float4(0.f, 0.f, 0.f, 0.f);
                needUnit = true;
            }
        }
        if (__ballot(needUnit) != 0ull) {
            const unsigned int newUnit = takeUnits(needUnit);
            if (needUnit) {
                unit = newUnit;
                if (newUnit != 0xFFFFFFFFu) {
                    unitSamples(p, newUnit, &pixel, &sample, &endSample);
                    startNext = true;
                } else {
                    alive = false;
                }
            }
        }
#ifdef PATHED_SHADE_PROFILE
        profShadeCycles += __builtin_amdgcn_s_memtime() - profStamp;
#endif
    }
#ifdef PATHED_SHADE_PROFILE
    if (lane == 0) {
        const unsigned long long values[12] = { profIterations, profAlive, profSteps, profLaneSteps, profShades, profReady, profPosted, profLeft,
                                                __builtin_amdgcn_s_memtime() - profStart, profBurstCycles, profShadeCycles, 1ull };
        for (int i = 0; i < 12; i++) { atomicAdd(&p.stats[kStatShadeProfile + i], values[i]); }
    }
#endif
}
