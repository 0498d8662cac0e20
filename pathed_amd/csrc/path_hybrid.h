// k_path_hybrid: the fused path kernel for scenes of 65 .. kHybridMaxTris triangles (round 5: "the cliff at triangle 65").
//
// k_path_small serves scenes of <= 64 triangles: every ray tests every primitive record out of the kernel arguments, whole
// paths stay in registers, 2 650 Msamples/s on the Cornell box.  One triangle more and the scene fell to the BVH kernels:
// the reference's own cornell-glossy / cornell-glass (1 112 triangles: the box, a cube and a 1 088-triangle ball) ran 894 / 826.
// Those scenes are a FEW LARGE triangles (walls, boxes: what most rays hit) plus a cluster of small ones that few rays come
// near.  This kernel splits the scene accordingly (pathed_hip.hip: buildHybrid):
//   * the DIRECT set, up to 64 triangles, the largest by area: k_path_small's two-phase all-items intersector (parallelogram
//     records in the kernel arguments, candidates resolved across the wave; small_items.h, kernels.h);
//   * the TREE part, everything else, with a 4-wide BVH of its own (same builder, same node format, resident in L2): a ray
//     walks it only if its segment -- already cut to the direct hit -- meets the part's bounding box (a packed slab test
//     on kernarg scalars), and the walk is k_trace's (trace.h: innerStep / leafStep, LDS stack rows, deferred leaves) with
//     the wave's idle lanes taking rays off the lanes that have two (the path's ray and the shadow ray): a lane that
//     finishes pulls the next ray of the wave's list over ds_bpermute from the lane that owns it.  A burst ends when its
//     list is dealt and only a few long rays are left (ray cost is heavy-tailed: the first version ran 19 - 26 steps per
//     burst at 8 of 64 lanes, profiles/r5_hybrid_profile.log): those STRAGGLERS stay parked on their lanes -- eight words:
//     the hit so far, the node to visit, the stack height; the stack rows stay in LDS -- their paths sit the vertex and the
//     next pass out, and the next burst carries them on (k_path_wave's rule, path_wave.h).
// Hits are the full-tree walk's bit for bit: every triangle is tested by intersectTriangle with the ray's own (o, d), the
// acceptance rule is "smallest t, then smallest primitive id" whatever the order candidates arrive in, and the two parts are
// merged by that rule; an occlusion query is the OR of the parts.  The vertex code is pathVertex (path_wave.h), the unit
// decomposition k_path_small's: images are the wavefront kernels' bit for bit (tests/test_gpu_hybrid.py).
//
// Replaces rtcIntersect1 / rtcOccluded1 (reference src/scene.cpp:113, :374) under PathTracer::L (src/path_tracer.cpp:19-216).
// Included by kernels.h inside namespace pathed.

static const int kHybridMaxTris = 4096;        // scenes up to this many triangles (no spheres) take the hybrid kernel
static const int kHybridStackRows = 5;         // LDS rows of a lane's traversal stack; deeper entries spill to p.stackOverflow
static const int kHybridStashRows = 5;
static const int kHybridResolveItems = 64;     // capacity of the direct pass's shared phase 2 (more: the owners finish in place)
#ifndef PATHED_HYBRID_WAVES
#define PATHED_HYBRID_WAVES 4
#endif
static const int kHybridRefill = 40;           // default of p.suspendPatience: idle lanes draw from the burst's list once fewer than this many are busy
static const int kHybridStragglers = 16;       // default of p.suspendLanes: a burst may end once its list is dealt and fewer rays than this are in flight
// Bursts are BATCHED: a scene's rays mostly never come near the tree part (the reference's dragon scene: 11 of 88 rays per
// iteration and wave), and a burst over a dozen rays runs as many steps as one over sixty.  Rays wait in the list, their paths
// with them, while the others carry on; a burst runs once kHybridBatch rays are waiting or in flight, or fewer than
// kHybridReady paths can still proceed without one.
static const int kHybridBatch = 24;            // default of p.hybridBatch (profiles/r5_ab_hybrid.log: 24 / 28 on the scenes that take the kernel by default)
static const int kHybridReady = 28;            // default of p.hybridReady

// per-wave LDS, in 4-byte words: [kHybridStackRows + 1][64] stack rows | 64 x float4 hit rows | 64 owner flags | the list of
// rays waiting for a burst (128 ushort entries) | the direct pass's phase-2 scratch (best 128, uv 128, items 32, occluded 64,
// count 1).  4.4 KiB per wave; with the stash 38 KiB per block: four blocks per CU
static const int kHybridStackWords = (kHybridStackRows + 1) * 64;
static const int kHybridScratchWords = 128 + 2 * kHybridResolveItems + kHybridResolveItems / 2 + 64 + 1;
static const int kHybridWaveWords = kHybridStackWords + 64 * 4 + 64 + 64 + ((kHybridScratchWords + 3) & ~3);
static const unsigned int kHybridOccluded = 0x100u;   // owner flags: the low byte counts the owner's finished rays

// (hybridProxy / hybridProxySphere, the two conservative "can this ray meet the tree part at all" tests: trace.h)

template <typename TRAITS>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PATHED_HYBRID_WAVES, PATHED_HYBRID_WAVES))) void k_path_hybrid(RenderParams p, SmallTris smallTris)
{
    extern __shared__ float4 ldsDynamic[];           // the material table, nMaterials x 96 B
    __shared__ float4 stashRows[kHybridStashRows * kBlock];
    __shared__ unsigned int waveWords[kWavesPerBlock * kHybridWaveWords];

    MaterialAccess<true> materials;
    {
        const int words = p.scene.nMaterials * (int)(sizeof(DMaterial) / 4);
        const int *source = reinterpret_cast<const int *>(p.scene.materials);
        int *target = reinterpret_cast<int *>(ldsDynamic);
        for (int i = threadIdx.x; i < words; i += kBlock) { target[i] = source[i]; }
        __syncthreads();
        materials.table = reinterpret_cast<const DMaterial *>(ldsDynamic);
    }

    const DScene &scene = p.scene;
    const int lane = threadIdx.x & 63;
    const int wave = (int)(threadIdx.x >> 6);
    const unsigned int waveId = blockIdx.x * kWavesPerBlock + (unsigned int)wave;
    const uint64_t seed = ((uint64_t)p.seedHi << 32) | p.seedLo;

    // the direct set: item-ordered triangle records (phase 2 indexes them in the order phase 1's bits come in)
    TraceGeometry direct;
    direct.nodes = nullptr;
    direct.tris = p.scene.leafTris;
    direct.nNodes = 0;
    direct.nTris = p.hybridDirectTris;
    direct.spheres = nullptr;
    direct.nSpheres = 0;
    // the tree part
    TraceGeometry tree;
    tree.nodes = p.hybridNodes;
    tree.tris = p.hybridTris;
    tree.nNodes = p.hybridNodeCount;
    tree.nTris = p.hybridTreeTris;
    tree.spheres = nullptr;
    tree.nSpheres = 0;

    unsigned int *mine = waveWords + wave * kHybridWaveWords;
    LaneStack stack;
    stack.lds = reinterpret_cast<int *>(mine) + lane;                       // entry k at lds[k * 64]
    stack.overflowStride = (size_t)gridDim.x * kBlock;
    stack.overflow = p.stackOverflow + ((size_t)blockIdx.x * kBlock + threadIdx.x);
    float4 *hitRows = reinterpret_cast<float4 *>(mine + kHybridStackWords);          // per owner lane: the path ray's hit so far
    unsigned int *ownerFlags = mine + kHybridStackWords + 64 * 4;                     // per owner lane: finished rays | kHybridOccluded
    unsigned short *entries = reinterpret_cast<unsigned short *>(mine + kHybridStackWords + 64 * 4 + 64);   // [128]: owner lane | 0x100 for a shadow ray
    unsigned int *scratchWords = mine + kHybridStackWords + 64 * 4 + 64 + 64;
    ResolveScratch scratch;
    scratch.best = reinterpret_cast<unsigned long long *>(scratchWords);                                   // 128 words
    scratch.uv = reinterpret_cast<float2 *>(scratchWords + 128);                                           // 2 words per item
    scratch.items = reinterpret_cast<unsigned short *>(scratchWords + 128 + 2 * kHybridResolveItems);      // half a word per item
    scratch.occluded = scratchWords + 128 + 2 * kHybridResolveItems + kHybridResolveItems / 2;             // 64 words
    scratch.count = scratch.occluded + 64;

    // ---- work units: as k_path_small
    unsigned int queue = waveId % (unsigned int)p.nQueues, queuesTried = 0;
    unsigned int reservedNext = 0, reservedEnd = 0;
    auto takeUnits = [&](bool want) -> unsigned int {
        unsigned int taken = 0xFFFFFFFFu;
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
            const unsigned int rank = laneRank(wanting);
            const bool served = ((wanting >> lane) & 1ull) != 0ull && rank < available;
            if (served) { taken = queue * p.unitsPerQueue + reservedNext + rank; }
            const unsigned int count = (unsigned int)__popcll(wanting);
            reservedNext += count < available ? count : available;
            wanting &= ~__ballot(served);
        }
        return taken;
    };

    // ---- the path a lane carries
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
    bool pendingShadow = false;
    V3 shadowDirection = v3(0.f, 0.f, 1.f);
    float shadowTfar = 0.f;
    unsigned int waiting = 0;        // rays of this lane's path that are in the tree part and not yet back (0, 1 or 2)
    unsigned int pendingCount = 0;   // (wave-uniform) rays in the list, waiting for the next burst

    // ---- the ray a lane walks through the tree part (anybody's); between bursts a straggler is PARKED: these eight words
    bool parked = false;
    unsigned int target = 0;         // owner lane | 0x100 for a shadow ray
    float parkedBest = 0.f, parkedU = 0.f, parkedV = 0.f;
    int parkedPrim = -1, parkedCurrent = 0, parkedLeaf = 0, parkedSp = 0;

#ifdef PATHED_SHADE_PROFILE
    // (tuning builds, tools/hybrid_profile.py) where a wave's time goes and how full its bursts are
    unsigned long long profIterations = 0, profAlive = 0, profPosted = 0, profBursts = 0, profSteps = 0, profLaneSteps = 0, profRefills = 0, profLeafSteps = 0;
    unsigned long long profPassCycles = 0, profBurstCycles = 0, profShadeCycles = 0, profStart = __builtin_amdgcn_s_memtime(), profStamp = 0;
    unsigned long long profShaded = 0, profLeft = 0;
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
            pendingShadow = false;
        }
        // (a dead path has no ray in flight: its last vertex waited for them)
        if (__ballot(alive) == 0ull) { break; }
        // paths whose rays are all back carry on; the others sit this pass and this vertex out
        const bool ready = alive && waiting == 0u;
#ifdef PATHED_SHADE_PROFILE
        profIterations++; profAlive += (unsigned long long)__popcll(__ballot(alive)); profStamp = __builtin_amdgcn_s_memtime();
#endif

        // ---- what the ray queries do not need waits in LDS, [row][thread] (the random stream's keys are rebuilt from pixel
        // and sample afterwards: 30 integer instructions against a sixth row); a waiting path's rows stay as they are
        if (ready) {
            float4 *stash = stashRows + threadIdx.x;
            stash[0 * kBlock] = make_float4(path.result.r, path.result.g, path.result.b, path.bsdfPdf);
            stash[1 * kBlock] = make_float4(path.modulation.r, path.modulation.g, path.modulation.b, path.cosTheta);
            stash[2 * kBlock] = make_float4(path.throughput.r, path.throughput.g, path.throughput.b, intAsFloat(path.st));
            stash[3 * kBlock] = make_float4(partial.x, partial.y, partial.z, intAsFloat(path.firstEmitMaterial));
            stash[4 * kBlock] = make_float4(intAsFloat((int)pixel), intAsFloat((int)sample), intAsFloat((int)endSample), intAsFloat((int)unit));
        }
        asm volatile("" ::: "memory");

        const bool traceShadow = ready && pendingShadow;
        LaneRay ray;
        laneRayInit(ray, path.o, path.d, PATHED_TNEAR, PATHED_TFAR, false);
        bool occluded = false;

        // ---- the direct set: k_path_small's all-items intersector
        if (direct.nTris > 0 && __ballot(ready) != 0ull) {
            LaneRay shadowRay;
            laneRayInit(shadowRay, path.o, shadowDirection, PATHED_TNEAR, shadowTfar, true);
            unsigned int candidatesLow = 0, candidatesHigh = 0, shadowLow = 0, shadowHigh = 0;
            if (ready) {
                smallCandidatesItems<true, true, true>(smallTris.data, p.smallQuads, direct.nTris, p.smallKappaT, path.o, path.d, shadowDirection,
                                                        &candidatesLow, &candidatesHigh, &shadowLow, &shadowHigh,
                                                        candidateNear(PATHED_TNEAR), candidateFar(shadowTfar));
                if (!traceShadow) { shadowLow = 0u; shadowHigh = 0u; }
            }
            smallResolveShared<kHybridResolveItems, false, false>(direct, ray, shadowRay, candidatesLow, candidatesHigh, shadowLow, shadowHigh, 0u, 0u, scratch);
            occluded = traceShadow && shadowRay.occluded;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();   // the scratch words become the burst's ray list
        }
        if (traceShadow && occluded) { path.pend = rgb(0.f); }
#ifdef PATHED_SHADE_PROFILE
        { const unsigned long long now = __builtin_amdgcn_s_memtime(); profPassCycles += now - profStamp; profStamp = now; }
#endif

        // ---- the tree part: only rays whose segment meets its box, cut to what the direct set left.  The owner's row holds
        // the path ray's hit so far (the direct set's); whoever walks the ray replaces it if the tree part has a better one
        if (ready) {
            hitRows[lane] = make_float4(ray.best, ray.bestU, ray.bestV, intAsFloat(ray.bestPrim));
            ownerFlags[lane] = 0u;
        }
        const float pathFar = ray.best;                    // (PATHED_TFAR without a direct hit; the owner's row holds it for the walk)
        if (tree.nNodes > 0) {
            const bool postPath = ready && hybridProxy(p.hybridLo, p.hybridHi, path.o, path.d, pathFar) && hybridProxySphere(p.hybridSphere, path.o, path.d);
            const bool postShadow = traceShadow && !occluded && hybridProxy(p.hybridLo, p.hybridHi, path.o, shadowDirection, shadowTfar)
                && hybridProxySphere(p.hybridSphere, path.o, shadowDirection);
            const unsigned long long pathMask = __ballot(postPath), shadowMask = __ballot(postShadow);
            if (postPath) { entries[pendingCount + laneRank(pathMask)] = (unsigned short)lane; }
            if (postShadow) { entries[pendingCount + (unsigned int)__popcll(pathMask) + laneRank(shadowMask)] = (unsigned short)(lane | 0x100); }
            if (ready) { waiting = (postPath ? 1u : 0u) + (postShadow ? 1u : 0u); }
            pendingCount += (unsigned int)(__popcll(pathMask) + __popcll(shadowMask));
        }
        pendingShadow = false;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- a burst, if enough rays have gathered or too few paths can go on without one
        const int nParked = __popcll(__ballot(parked));
        const int canProceed = __popcll(__ballot(alive && waiting == 0u));
        const unsigned int gathered = pendingCount + (unsigned int)nParked;
        if (gathered != 0u && ((int)gathered >= p.hybridBatch || canProceed < p.hybridReady)) {
            const unsigned int listCount = pendingCount;
            pendingCount = 0u;
#ifdef PATHED_SHADE_PROFILE
            profPosted += listCount; profBursts++;
#endif
            unsigned int listPos = 0;
            bool active = false;
            LaneRay walk;
            laneRayInit(walk, path.o, path.d, PATHED_TNEAR, PATHED_TFAR, false);
            // ---- the stragglers of the last burst carry on where they stopped: the ray again from its owner, then the eight words
            if (__ballot(parked) != 0ull) {
                const int owner = (int)(target & 63u);
                const bool forShadow = (target & 0x100u) != 0u;
                const float ox = __shfl(path.o.x, owner), oy = __shfl(path.o.y, owner), oz = __shfl(path.o.z, owner);
                const float ax = __shfl(path.d.x, owner), ay = __shfl(path.d.y, owner), az = __shfl(path.d.z, owner);
                const float bx = __shfl(shadowDirection.x, owner), by = __shfl(shadowDirection.y, owner), bz = __shfl(shadowDirection.z, owner);
                const float farShadow = __shfl(shadowTfar, owner);
                if (parked) {
                    laneRayInit(walk, v3(ox, oy, oz), forShadow ? v3(bx, by, bz) : v3(ax, ay, az), PATHED_TNEAR, forShadow ? farShadow : parkedBest, forShadow);
                    walk.best = forShadow ? farShadow : parkedBest;
                    walk.bestU = parkedU; walk.bestV = parkedV; walk.bestPrim = parkedPrim;
                    walk.current = parkedCurrent; walk.pendingLeaf = parkedLeaf; walk.sp = parkedSp;
                    active = true;
                    parked = false;
                }
            }
            while (true) {
                // idle lanes take the next rays of the list from the lanes that own them
                while (listPos < listCount) {
                    const unsigned long long idleMask = __ballot(!active);
                    if (idleMask == 0ull) { break; }
                    const unsigned int rank = laneRank(idleMask);
                    const unsigned int available = listCount - listPos;
                    const bool take = !active && rank < available;
                    const unsigned int entry = take ? (unsigned int)entries[listPos + rank] : (unsigned int)lane;
                    const int owner = (int)(entry & 63u);
                    const bool forShadow = (entry & 0x100u) != 0u;
                    // (every lane takes part in the exchange)
                    const float ox = __shfl(path.o.x, owner), oy = __shfl(path.o.y, owner), oz = __shfl(path.o.z, owner);
                    const float ax = __shfl(path.d.x, owner), ay = __shfl(path.d.y, owner), az = __shfl(path.d.z, owner);
                    const float bx = __shfl(shadowDirection.x, owner), by = __shfl(shadowDirection.y, owner), bz = __shfl(shadowDirection.z, owner);
                    const float farShadow = __shfl(shadowTfar, owner);
                    const float farPath = hitRows[owner].x;      // the direct set's hit (or PATHED_TFAR): the owner wrote it when it posted
                    if (take) {
                        laneRayInit(walk, v3(ox, oy, oz), forShadow ? v3(bx, by, bz) : v3(ax, ay, az), PATHED_TNEAR, forShadow ? farShadow : farPath, forShadow);
                        target = entry;
                        active = true;
                    }
                    const unsigned int wanted = (unsigned int)__popcll(idleMask);
                    listPos += wanted < available ? wanted : available;
#ifdef PATHED_SHADE_PROFILE
                    profRefills++;
#endif
                }
                const bool dry = listPos == listCount;
                // The list is dealt and few rays are still in flight: the burst ends if the vertex code has something to do -- at
                // least twice as many paths with all their rays back as rays in flight (every burst ends with progress; the end
                // of a render, where few paths are left, does not shade one lane at a time).
                auto leaveStragglers = [&](unsigned long long activeMask) -> bool {
                    const int inFlight = __popcll(activeMask);
                    if (inFlight >= p.suspendLanes) { return false; }
                    const bool complete = alive && (waiting == 0u || (*(volatile unsigned int *)&ownerFlags[lane] & 0xFFu) == waiting);
                    return __popcll(__ballot(complete)) >= 2 * inFlight;
                };
                {
                    const unsigned long long activeMask = __ballot(active);
                    if (activeMask == 0ull) { break; }
                    if (dry && leaveStragglers(activeMask)) { break; }
                }
                while (true) {
                    const unsigned long long leafMask = __ballot(active && walk.pendingLeaf != 0);
                    const unsigned long long innerMask = __ballot(active && walk.pendingLeaf == 0);
                    const bool trianglePhase = __popcll(leafMask) >= kLeafThreshold || innerMask == 0ull;
#ifdef PATHED_SHADE_PROFILE
                    profSteps++; profLaneSteps += (unsigned long long)__popcll(trianglePhase ? leafMask : innerMask); profLeafSteps += trianglePhase ? 1ull : 0ull;
#endif
                    bool done = false;
                    if (trianglePhase) {
                        if (active && walk.pendingLeaf != 0) { done = leafStep<false, kHybridStackRows, 64, false>(tree, stack, walk, nullptr); }
                    } else {
                        if (active && walk.pendingLeaf == 0) { done = innerStep<false, kHybridStackRows, 64, false, false>(tree, stack, p.maxStack, walk, nullptr); }
                    }
                    if (done) {
                        const unsigned int owner = target & 63u;
                        if (walk.anyHit) {
                            atomicAdd(&ownerFlags[owner], walk.occluded ? (1u + kHybridOccluded) : 1u);
                        } else {
                            if (walk.bestPrim >= 0) {
                                // the acceptance rule across the two parts: smaller t, then smaller primitive id (the walk only
                                // accepted t <= the direct set's, so only a tie in t needs the row)
                                const float4 held = hitRows[owner];
                                const int heldPrim = floatAsInt(held.w);
                                if (heldPrim < 0 || walk.best < held.x || (walk.best == held.x && walk.bestPrim < heldPrim)) {
                                    hitRows[owner] = make_float4(walk.best, walk.bestU, walk.bestV, intAsFloat(walk.bestPrim));
                                }
                            }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            atomicAdd(&ownerFlags[owner], 1u);
                        }
                        active = false;
                    }
                    const unsigned long long activeMask = __ballot(active);
                    if (activeMask == 0ull) { break; }
                    if (!dry && __popcll(activeMask) < p.suspendPatience) { break; }
                    if (dry && leaveStragglers(activeMask)) { break; }
                }
            }
            // ---- what is still in flight is parked on its lane until the next burst
            if (active) {
                parked = true;
                parkedBest = walk.best; parkedU = walk.bestU; parkedV = walk.bestV; parkedPrim = walk.bestPrim;
                parkedCurrent = walk.current; parkedLeaf = walk.pendingLeaf; parkedSp = walk.sp;
            }
#ifdef PATHED_SHADE_PROFILE
            profLeft += (unsigned long long)__popcll(__ballot(active));
#endif
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }

        // ---- the paths whose rays are all back
        bool shade = false;
        float4 h = make_float4(0.f, 0.f, 0.f, intAsFloat(-1));
        if (alive) {
            const unsigned int flags = waiting != 0u ? *(volatile unsigned int *)&ownerFlags[lane] : 0u;
            if (waiting == 0u || (flags & 0xFFu) == waiting) {
                shade = true;
                h = hitRows[lane];
                if (flags & kHybridOccluded) { path.pend = rgb(0.f); }
                waiting = 0u;
            }
        }
        __builtin_amdgcn_wave_barrier();   // (the rows are rewritten by the next pass)
#ifdef PATHED_SHADE_PROFILE
        { const unsigned long long now = __builtin_amdgcn_s_memtime(); profBurstCycles += now - profStamp; profStamp = now; }
        profShaded += (unsigned long long)__popcll(__ballot(shade));
#endif

        if (shade) {
            asm volatile("" ::: "memory");
            const float4 *stash = stashRows + threadIdx.x;
            const float4 s0 = stash[0 * kBlock], s1 = stash[1 * kBlock], s2 = stash[2 * kBlock], s3 = stash[3 * kBlock], s4 = stash[4 * kBlock];
            path.result.r = s0.x; path.result.g = s0.y; path.result.b = s0.z; path.bsdfPdf = s0.w;
            path.modulation.r = s1.x; path.modulation.g = s1.y; path.modulation.b = s1.z; path.cosTheta = s1.w;
            path.throughput.r = s2.x; path.throughput.g = s2.y; path.throughput.b = s2.z; path.st = floatAsInt(s2.w);
            partial = make_float4(s3.x, s3.y, s3.z, 0.f); path.firstEmitMaterial = floatAsInt(s3.w);
            pixel = (uint32_t)floatAsInt(s4.x); sample = (uint32_t)floatAsInt(s4.y); endSample = (uint32_t)floatAsInt(s4.z); unit = (unsigned int)floatAsInt(s4.w);
            makeKey(seed, pixel, sample, &path.random.k0, &path.random.k1);
        }

        // ---- the vertex (path_wave.h: pathVertex = k_path_small's vertex code)
        bool finished = false;
        Rgb color = rgb(0.f);
        if (shade) {
            ShadowRequest shadow;
            finished = pathVertex<TRAITS>(p, scene, materials, path, h, &shadow, &color);
            if (!finished && shadow.push) {
                pendingShadow = true;
                shadowDirection = shadow.direction;
                shadowTfar = shadow.tfar;
            }
        }

        // ---- end of a sample (k_path_small)
        bool needUnit = false;
        if (shade && finished) {
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
        const unsigned long long values[15] = { profIterations, profAlive, profPosted, profBursts, profSteps, profLaneSteps, profRefills, profLeafSteps,
                                                __builtin_amdgcn_s_memtime() - profStart, profPassCycles, profBurstCycles, profShadeCycles, 1ull, profShaded, profLeft };
        for (int i = 0; i < 15; i++) { atomicAdd(&p.stats[kStatShadeProfile + i], values[i]); }
    }
#endif
}
