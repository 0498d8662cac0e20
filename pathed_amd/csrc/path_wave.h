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
static const unsigned int kWaveShadowTag = 0x100u;    // list entry / lane target: owner lane | this bit for a shadow ray
static const unsigned int kWaveOccludedFlag = 0x100u; // ownerFlags: the low byte counts the owner's finished rays

// dynamic LDS of a block: [STACK + 1][kBlock] stack rows | per wave 128 x 2 float4 of ray list | kBlock float4 hits |
// kBlock flag words | (LDS_MATERIALS) the material table
__host__ __device__ inline size_t pathWaveLdsBytes(int stackRows, int nLdsMaterials)
{
    return (size_t)(stackRows + 1) * kBlock * sizeof(int) + (size_t)kWavesPerBlock * kWaveListRays * 2 * sizeof(float4)
        + (size_t)kBlock * sizeof(float4) + (size_t)kBlock * sizeof(unsigned int) + (size_t)nLdsMaterials * sizeof(DMaterial);
}

template <bool LDS_MATERIALS, int STACK, typename TRAITS, bool SPHERES>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PATHED_WAVE_WAVES, PATHED_WAVE_WAVES))) void k_path_wave(RenderParams p)
{
    extern __shared__ float4 ldsRaw[];
    LaneStack stack;
    stack.lds = reinterpret_cast<int *>(ldsRaw) + threadIdx.x;
    stack.overflowStride = (size_t)gridDim.x * kBlock;
    stack.overflow = p.stackOverflow + ((size_t)blockIdx.x * kBlock + threadIdx.x);
    const int lane = threadIdx.x & 63;
    const int wave = (int)(threadIdx.x >> 6);
    float4 *listO = ldsRaw + ((STACK + 1) * kBlock) / 4 + wave * (2 * kWaveListRays);   // .w = tfar
    float4 *listD = listO + kWaveListRays;                                                // .w = owner lane | kWaveShadowTag
    float4 *hitRows = ldsRaw + ((STACK + 1) * kBlock) / 4 + kWavesPerBlock * 2 * kWaveListRays + wave * 64;
    unsigned int *ownerFlags = reinterpret_cast<unsigned int *>(ldsRaw + ((STACK + 1) * kBlock) / 4 + kWavesPerBlock * 2 * kWaveListRays + kBlock) + wave * 64;

    MaterialAccess<LDS_MATERIALS> materials;
    if (LDS_MATERIALS) {
        const int words = p.scene.nMaterials * (int)(sizeof(DMaterial) / 4);
        const int *source = reinterpret_cast<const int *>(p.scene.materials);
        int *target = reinterpret_cast<int *>(ldsRaw + ((STACK + 1) * kBlock) / 4 + kWavesPerBlock * 2 * kWaveListRays + kBlock + kBlock / 4);
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
        if (__ballot(alive) == 0ull) { break; }   // (a dead lane's path has no ray in flight: nothing is left behind)

        // ---- post the new rays: continuation rays first, then the shadow rays (which leave the same point)
        unsigned int listCount = 0, listPos = 0;   // wave-uniform
        {
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
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
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
        while (true) {
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

        // ---- shade burst: the paths whose rays have all come back
        bool ready = false;
        float4 h = make_float4(0.f, 0.f, 0.f, intAsFloat(-1));
        if (alive && waiting != 0u) {
            const unsigned int flags = *(volatile unsigned int *)&ownerFlags[lane];
            if ((flags & 0xFFu) == waiting) {
                ready = true;
                h = hitRows[lane];
                if (flags & kWaveOccludedFlag) { path.pend = rgb(0.f); }
                waiting = 0u;
            }
        }
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
    }
}
