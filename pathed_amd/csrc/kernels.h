// The wavefront path-tracing kernels (gfx950).
//
// Work decomposition.  A UNIT is `chunk` consecutive samples of one pixel (default 1): chunk c of
// pixel x covers the samples sppBegin + c * chunk + [0, chunk); which unit id stands for which (c, x) is
// THE UNIT ORDER below.  A pool of path SLOTS pulls units dynamically (block-aggregated grab from one of
// kUnitQueues cursors, moving on to the other queues when its own is empty), so lanes stay busy whatever
// the per-pixel path length ("path regeneration").  A unit's samples are summed in sample order into a
// partial sum; the partial is written to chunkBuf[c * nPixels + x], and k_resolve adds a pixel's partials
// to the radiance sum in chunk order.  The result is deterministic and does not depend on scheduling;
// with chunk == 1 it is exactly the reference's order
// (radianceLookup += one sample per wave, src/integrator.cpp:42-51, sample_integrator.cpp:61-63).
//
// Per iteration, two launches on one stream:
//   k_trace  persistent waves walk [slot rays | per-block shadow rays] in a static stride
//            closest hit  -> hit[slot]                  (Scene::testIntersect's rtcIntersect1)
//            any hit      -> pend[slot] = 0 if occluded (Scene::testOcclusion's rtcOccluded1)
//   k_shade  one lane per slot: finishes the previous vertex's BSDF-sampling MIS term with the
//            new hit, updates throughput, samples the BSDF and one light at the new vertex
//            (PathTracer::L / direct / directSampleLights / directSampleBSDF,
//            src/path_tracer.cpp:19-216), emits the next ray and a shadow ray, or terminates the
//            sample, adds it to the unit's partial sum and starts the next sample / unit.
//            Shadow rays and unit requests are compacted with wave ballots + an LDS block scan.
// No single-address atomics on the hot path: one counter saturates at ~88 ops/us on MI355X.
// State is SoA-of-float4 in HBM so a wave reads 1 KiB contiguous per stream.
#pragma once

// Phase 2 of k_path_small<QUADS> (A/B builds: tools/ab_resolve.py, profiles/r4_ab_resolve.log): 0 every lane resolves its own
// candidates; 1 the first candidate of each ray in place, the wave shares the left-overs out; 2 (the product) as 1, with all
// of the shadow ray's candidates shared out
#ifndef PATHED_RESOLVE_SHARED
#define PATHED_RESOLVE_SHARED 2
#endif
#ifndef PATHED_EXPERIMENTS
#define PATHED_EXPERIMENTS 0   // 1: `make experiments` -- the measured-and-rejected kernel organisations (kernels_experiments.h)
#endif
#include "mfma_candidates.h"
#include "small_items.h"
#include "shading.h"
#include "trace.h"
#include "volume.h"

namespace pathed {

static const int kBlock = 256;
static_assert(kBlock == kVolumeBlock, "volume.h indexes the LDS stack rows with the block size");
static const int kWavesPerBlock = kBlock / 64;
static const int kMaxLdsMaterials = 96;  // 96 x 96 B = 9 KiB of LDS
#ifndef PATHED_UNIT_QUEUES
#define PATHED_UNIT_QUEUES 32
#endif
static const int kUnitQueues = PATHED_UNIT_QUEUES;       // sharded work-unit cursors
static const unsigned int kUnitGroup = 256u;   // pixels of a group of the unit order: a 32 x 8 tile where the resolution allows

// n / d for a launch-invariant d (Granlund & Montgomery 1994, figure 4.1; exact for every 32-bit n): the unit order
// needs three such divisions per unit and hipcc's general 32-bit division is ~25 instructions.
struct FastDiv {
    unsigned int magic, shift1, shift2;
};
__host__ inline FastDiv makeFastDiv(unsigned int d)
{
    int l = 0;
    while ((1ull << l) < (unsigned long long)d) { l++; }
    FastDiv f;
    f.magic = (unsigned int)(((1ull << 32) * ((1ull << l) - (unsigned long long)d)) / (unsigned long long)d + 1ull);
    f.shift1 = l < 1 ? (unsigned int)l : 1u;
    f.shift2 = l > 1 ? (unsigned int)(l - 1) : 0u;
    return f;
}
__device__ inline unsigned int fastDivide(unsigned int n, FastDiv d)
{
    const unsigned int t = __umulhi(d.magic, n);
    return (t + ((n - t) >> d.shift1)) >> d.shift2;
}
#ifndef PATHED_REFILL
#define PATHED_REFILL 48
#endif
static const int kRefillThreshold = PATHED_REFILL;  // refill a wave's idle lanes once fewer than this many are busy
#ifndef PATHED_LEAF_THRESHOLD
#define PATHED_LEAF_THRESHOLD 24
#endif
#ifndef PATHED_WARM_LINES
#define PATHED_WARM_LINES 0   // 1: k_trace touches the next leaf / stack entry one step early (trace.h: warmLine) -- measured 4-6 % SLOWER
#endif
static const int kLeafThreshold = PATHED_LEAF_THRESHOLD;  // lanes with a leaf pending that trigger a triangle phase

// counters[] layout (unsigned int)
static const int kCtrRemaining = 0;    // slots that still have work
// Every cursor sits on its own 128-byte line: atomics to one line serialise in L2 (about 88 per
// microsecond on MI355X), whatever word of the line they address.
static const int kCursorStride = 32;   // words
static const int kCtrUnitCursor = kCursorStride;                                  // + queue * kCursorStride: next unit of the queue
static const int kTraceShards = 32;
static const int kCtrTraceCursor = kCtrUnitCursor + kUnitQueues * kCursorStride;  // + shard * kCursorStride: next card (rewound by k_shade)
static const int kCtrShadowCount = kCtrTraceCursor + kTraceShards * kCursorStride;  // + parity * kCursorStride: length of the shadow-ray list
// The split shade stage (k_vertex + k_regen) works from lists the PRODUCER of a result writes: k_trace appends the slot of
// every finished closest-hit ray to the HIT list or the MISS list, k_vertex appends the slots whose sample ended at the
// vertex to the miss list as well (flagged kEntryColorReady).  A list is made of 64-entry blocks; a wave collects entries
// in LDS and writes one block at a time, its number drawn from one of kListShards cursors (block = ticket * shards +
// shard, the shard advancing with every block so the cursors stay level): one atomic per 64 entries, no line shared.
static const int kListShards = 32;
static const int kListHit = 0, kListMiss = 1;
static const int kCtrListCount = kCtrShadowCount + 2 * kCursorStride;                   // + ((list * 2 + parity) * kListShards + shard) * kCursorStride
static const int kCtrDeferred = kCtrListCount + 2 * 2 * kListShards * kCursorStride;    // + (list * 2 + parity) * kCursorStride: slots waiting for a parked shadow ray
static const int kCtrCount = kCtrDeferred + 2 * 2 * kCursorStride;
static const unsigned int kEntryInvalid = 0xFFFFFFFFu;      // padding of a partly filled list block
static const unsigned int kEntryColorReady = 0x40000000u;   // miss-list entry written by k_vertex: the sample's colour is in res.rgb
static const unsigned int kEntrySlotBits = 0x3FFFFFFFu;
static const unsigned int kListPending = 0x80000000u;       // k_trace, bit of a lane's `target`: the finished ray's slot has yet to join its list
#ifndef PATHED_CARD_ROUNDS
#define PATHED_CARD_ROUNDS 2
#endif
static const int kCardRounds = PATHED_CARD_ROUNDS;   // every lane fetches this many rays of a card
static const int kCard = 64 * kCardRounds;           // rays per trace card

// Tail suspension.  Once the pool is dealt a wave only thins out: the last few long rays would
// keep it (and the launch) alive at 5-6 busy lanes for a third of its lifetime.  Instead, a wave
// that has been out of cards for kSuspendPatience steps and is down to fewer than kSuspendLanes
// rays parks them -- ray, best hit so far and traversal stack -- in its private save area and
// exits; the same wave of the pool's next trace launch picks them up again, first thing, beside
// a full load of fresh rays.  The slot of a parked ray is marked (hit.w / pend.w) so k_shade
// leaves it alone until the ray's result is in.  Results do not depend on any of this: the hit
// acceptance rule is order-independent and every slot is shaded from its own finished rays.
static const int kSuspendLanes = 32;                     // default; RenderParams::suspendLanes = 0 disables (PATHED_SUSPEND_LANES)
static const int kSuspendPatience = 24;                   // default steps a wave rides out its tail before parking it (PATHED_SUSPEND_PATIENCE)
static const int kSaveWords = 18;                        // per-lane record ahead of the stack entries
static const int kPrimSuspended = (int)0x80000001u;      // hit.w of a slot whose closest-hit ray is parked
static const int kShadowSuspended = 0x7fc0dead;          // pend.w of a slot whose shadow ray is parked (a NaN pattern)

// stats[] layout (unsigned long long)
static const int kStatSamples = 0;
static const int kStatClosest = 1;
static const int kStatShadow = 2;
static const int kStatBoxes = 3;
static const int kStatTris = 4;
static const int kStatDropped = 5;
static const int kStatMaxBoxes = 6;   // most child boxes tested by a single ray (stats mode)
static const int kStatWaveSteps = 7;  // traversal steps executed by waves (stats mode)
static const int kStatLaneSteps = 8;  // ... and by lanes: lane utilisation = lane / (64 * wave)
static const int kStatRefills = 9;
static const int kStatWaveCycles = 10;     // sum of trace-wave lifetimes, shader clocks (stats mode)
static const int kStatWaveCyclesMax = 11;  // longest single trace wave
static const int kStatTailSteps = 12;      // wave steps run after the card pool ran dry
static const int kStatTailLaneSteps = 13;
static const int kStatTailCycles = 14;
static const int kStatParked = 15;         // rays parked by tail suspension
static const int kStatRefillCycles = 16;   // shader clocks spent refilling lanes / in inner phases / in triangle phases (stats mode)
static const int kStatInnerCycles = 17;
static const int kStatLeafCycles = 18;
static const int kStatInnerSteps = 19;
static const int kStatLocalClosest = 20;   // rays k_shade resolved itself (counting mode): closest-hit / occlusion
static const int kStatLocalShadow = 21;
static const int kStatShadeProfile = 24;   // PATHED_SHADE_PROFILE builds: (waves, lanes) per k_shade region, 2 words each
static const int kStatCount = 48;

// state word (rayD.w): bits 0..15 vertex that spawned the ray (0 = camera ray),
// 16 eligible, 17 delta, 18 continue (device_scene.h), 19..25 sample index inside the unit
static const int kStSampleShift = 19;
static const int kStSampleMask = 0x7F;
static const int kMaxChunk = 128;

struct PathState {
    float4 *rayO;   // origin.xyz, bits(material of a directly visible emitter, or -1)
    float4 *rayD;   // direction.xyz, bits(state word)
    float4 *hit;    // t, u, v, bits(prim)
    float4 *mod;    // modulation.rgb, pdf of the pending BSDF sample
    float4 *thr;    // throughput.rgb of the pending BSDF sample, |n_s . wi|
    float4 *res;    // result.rgb of the sample in flight, bits(unit)
    float4 *pend;   // light-sampling term of the pending vertex (zeroed if occluded)
    float4 *acc;    // partial radiance sum of the unit in flight
    float4 *shO;    // shadow rays, one dense list per iteration: origin.xyz, tfar
    float4 *shD;    //                                             direction.xyz, bits(slot)
    float4 *chunkBuf;                // one partial sum per unit, index = chunk * nPixels + pixel (partialIndex)
    // split shade stage: the hit / miss lists (64-entry blocks, kListShards x listCap of them each) and, per list and
    // parity, the dense list of slots whose shading waits for a parked shadow ray
    unsigned int *lists[2];
    unsigned int *deferred[2][2];
};

struct RenderParams {
    DScene scene;
    PathState state;
    unsigned int *counters;
    unsigned long long *stats;
    unsigned long long *suspendMask;  // per trace wave: lanes with a parked ray
    int *suspendData;                 // per trace wave: (kSaveWords + maxStack) x 64 words, word-major
    int *stackOverflow;               // per trace thread: stack entries beyond the LDS rows, [row][thread]
    int maxStack;                     // the tree's bound on stack entries (3 per level)
    int suspendLanes;                 // park the tail once fewer rays than this are left (0 = never)
    int suspendPatience;              // ... and the wave has run this many steps since its last card
    int parkMinCardsPerWave;          // ... and the pool still has this many cards' worth of live slots per wave
    int parity;                       // iteration & 1: k_shade(n) fills shadow list n & 1, k_trace(n) consumes list (n - 1) & 1
    unsigned int listCap;             // split shade stage: list blocks per shard (a list holds kListShards * listCap blocks)
    float *accum;          // 3*W*H radiance sums, index 3*(row*W+col)+c
    int nSlots;            // multiple of kBlock
    int nPixels;
    unsigned int nUnits;   // units of THIS pool
    // unit order (unitPlace): which pixel groups this pool's queues walk
    int unitOrder;                     // kOrderStripes | kOrderStripesTiled | kOrderTiles
    unsigned int pool, pools;          // queue q of pool h owns the chunks / pixel groups (q * pools + h) + j * (nQueues * pools), j = 0, 1, ..
    unsigned int nGroups;              // what the queues share out: chunks per pixel (stripes) or ceil(nPixels / kUnitGroup) (tiles)
    unsigned int lastGroupPixels;      // tiles: pixels of group nGroups - 1 (1 .. kUnitGroup)
    unsigned int groupUnits;           // units of one owned item: nPixels (stripes) or kUnitGroup * chunksPerPixel (tiles)
    FastDiv divStride, divGroupUnits, divBand, divWidth;   // by unitsPerQueue, by groupUnits, by 8 * image width, by the image width
    unsigned int queueUnits[kUnitQueues];        // units each queue of this pool holds
    int nQueues;           // min(kUnitQueues, shade blocks): every queue has a consumer
    unsigned int unitsPerQueue;        // stride of the unit ids: unit = queue * unitsPerQueue + position in the queue
    int unitGrab;          // k_path_small: units a wave reserves per atomic
    int chunk;             // samples per unit
    int chunksPerPixel;
    uint32_t seedLo, seedHi;
    uint32_t sppBegin, sppEnd;
    int startBounce, lastBounce;
    int smallQuads;           // k_path_small: parallelograms among the phase-1 records (small_items.h), item-order triangles 2 q, 2 q + 1
    // k_path_small<QUADS>: the scene's spheres, two per packed record (centre x, y, z, radius^2; the odd one out paired with a
    // sphere of radius -1 that no ray touches): phase 1 of the sphere queries (smallSphereCandidates)
    float spherePairs[8][4][2];   // [pair][component][half], kBruteForceMaxSpheres / 2 pairs
    float smallKappaT;        // ... and the absolute slack of their t bounds per det^2
    const float *mfmaTable;   // k_path_small<.., MFMA>: the A-side rows of the matrix-pipe phase 1 (mfma_candidates.h), kMfmaTableFloats
    MfmaFrame mfmaFrame;
    // k_path_hybrid (path_hybrid.h): scene.leafTris holds the DIRECT set in item order (hybridDirectTris of them), the rest of the
    // scene has a tree of its own whose bounds (padded) are hybridLo / hybridHi
    const float4 *hybridNodes, *hybridTris;
    int hybridNodeCount, hybridTreeTris, hybridDirectTris;
    float hybridLo[3], hybridHi[3];
    float hybridSphere[4];   // ... and a bounding sphere of the same triangles: centre, radius^2 (padded)
    int hybridBatch, hybridReady;   // a burst runs once this many rays wait or are in flight, or fewer paths than hybridReady can proceed
    // [r5] LOCAL RAYS of the wavefront (k_shade): the scene's few large triangles (a floor, a backdrop: at most kMaxLocalTris,
    // records as the tree's leaves hold them) and -- in hybridLo / hybridHi / hybridSphere -- the bounds of everything else.  A
    // ray that cannot meet those bounds can only hit one of the large triangles: k_shade tests them itself, writes the hit
    // and marks the slot kStLocal; an occlusion ray of that kind never joins the shadow list.  localCount = 0: off
    int localCount, localCounting;
    int afterTrace;           // k_shade: 1 = this launch follows a trace launch of the pool (every slot's results are in); 0 = a further
                              // shade launch of the same iteration: only the slots whose rays were all local take part
    float4 localTris[3 * 8];
};
static const int kMaxLocalTris = 8;

// BounceController, reference src/bounce_controller.cpp:14-25
__device__ inline bool checkDone(int lastBounce, int bounce)
{
    if (lastBounce == -1) { return false; }
    return bounce > lastBounce;
}

__device__ inline bool checkCounts(int startBounce, int lastBounce, int bounce)
{
    if (startBounce > bounce) { return false; }
    return !checkDone(lastBounce, bounce);
}

// ------------------------------------------------------------------------- slot lists (split shade stage)

__device__ inline unsigned int laneRank(unsigned long long mask)   // set bits of `mask` below this lane
{
    return __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
}

// One wave's end of a slot list: entries collect in `buffer` (128 words of LDS owned by the wave) and leave 64 at a time
// as one block.  All of a writer's state is ONE wave-uniform count: the trace kernel lives at its register limit, and every
// value the list code keeps alive across the traversal loop is a spill reload inside it (measured: +45 % on the kernel
// with two writers of five words each, against +3 % with a count each).  The cursor a block's number is drawn from is picked
// from the shader clock: nothing to carry along, and the shards stay level to a few per cent, which is all the consumers
// need (they walk max-over-shards blocks per shard).
struct ListWriter {
    unsigned int *buffer;
    unsigned int count;        // entries waiting in the buffer, < 64 between calls
};

__device__ inline void listWriterInit(ListWriter &w, unsigned int *buffer)
{
    w.buffer = buffer;
    w.count = 0u;
}

// Writes the first min(count, 64) buffered entries as one block (padded with kEntryInvalid).  The list has room for every
// slot once plus one partly filled block per writing wave and every shard takes its share of that (listCap has slack), so a
// shard with room is found at once; the loop is bounded anyway.
__device__ inline void listFlush(ListWriter &w, const RenderParams &p, int list)
{
    const unsigned int lane = threadIdx.x & 63u;
    unsigned int *cursors = p.counters + kCtrListCount + (list * 2 + p.parity) * kListShards * kCursorStride;
    __builtin_amdgcn_wave_barrier();
    unsigned int shard = ((unsigned int)__builtin_amdgcn_s_memtime() >> 5) % (unsigned int)kListShards;
    unsigned int ticket = 0u, shardUsed = 0u;
    bool placed = false;
    for (int attempt = 0; attempt < 2 * kListShards && !placed; attempt++) {
        if (lane == 0u) { ticket = atomicAdd(&cursors[shard * kCursorStride], 1u); }
        ticket = (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
        shardUsed = shard;
        shard = (shard + 1u) % (unsigned int)kListShards;
        placed = ticket < p.listCap;
    }
    const unsigned int head = lane < w.count ? w.buffer[lane] : kEntryInvalid;
    const unsigned int rest = w.buffer[64u + lane];
    if (placed) { p.state.lists[list][((size_t)ticket * kListShards + shardUsed) * 64u + lane] = head; }
    __builtin_amdgcn_wave_barrier();
    w.buffer[lane] = rest;
    __builtin_amdgcn_wave_barrier();
    w.count = w.count > 64u ? w.count - 64u : 0u;
}

// Call from wave-uniform control flow; `value` of the lanes with `want` joins the list.
__device__ inline void listAppend(ListWriter &w, const RenderParams &p, int list, bool want, unsigned int value)
{
    const unsigned long long mask = __ballot(want);
    if (mask == 0ull) { return; }
    if (want) { w.buffer[w.count + laneRank(mask)] = value; }
    w.count = (unsigned int)__builtin_amdgcn_readfirstlane((int)(w.count + (unsigned int)__popcll(mask)));
    if (w.count >= 64u) { listFlush(w, p, list); }
}

// A list as its consumer sees it: item i of the first `blocks * 64` is entry i of the block list (invalid beyond a shard's
// count), the items after that are the deferred slots.  Block-uniform; `shardCounts` is kListShards + 1 words of LDS.
struct ListReader {
    const unsigned int *list;
    const unsigned int *deferred;
    unsigned int *shardCounts;
    unsigned int blocks;          // kListShards * (most blocks any shard holds)
    unsigned int deferredCount;
    unsigned int items;           // blocks * 64 + deferredCount
};

__device__ inline void listReaderInit(ListReader &r, unsigned int *shardCounts, const RenderParams &p, int list, int parity)
{
    r.list = p.state.lists[list];
    r.deferred = p.state.deferred[list][parity];
    r.shardCounts = shardCounts;
    if (threadIdx.x == 0) { shardCounts[kListShards] = 0u; }
    __syncthreads();
    if (threadIdx.x < kListShards) {
        const unsigned int written = p.counters[kCtrListCount + ((list * 2 + parity) * kListShards + threadIdx.x) * kCursorStride];
        const unsigned int count = written < p.listCap ? written : p.listCap;
        shardCounts[threadIdx.x] = count;
        atomicMax(&shardCounts[kListShards], count);
    }
    __syncthreads();
    r.blocks = shardCounts[kListShards] * (unsigned int)kListShards;
    r.deferredCount = p.counters[kCtrDeferred + (list * 2 + parity) * kCursorStride];
    r.items = r.blocks * 64u + r.deferredCount;
}

__device__ inline unsigned int listEntry(const ListReader &r, unsigned int item)
{
    if (item < r.blocks * 64u) {
        const unsigned int block = item >> 6;
        const unsigned int shard = block % (unsigned int)kListShards, ticket = block / (unsigned int)kListShards;
        return ticket < r.shardCounts[shard] ? r.list[item] : kEntryInvalid;
    }
    return item < r.items ? r.deferred[item - r.blocks * 64u] : kEntryInvalid;
}

// A slot whose shading has to wait for its parked shadow ray joins the list the next iteration's kernel reads.
__device__ inline void deferSlot(const RenderParams &p, int list, bool defer, unsigned int slot)
{
    const unsigned long long mask = __ballot(defer);
    if (mask == 0ull) { return; }
    unsigned int base = 0u;
    if ((threadIdx.x & 63u) == (unsigned int)(__ffsll((long long)mask) - 1)) {
        base = atomicAdd(&p.counters[kCtrDeferred + (list * 2 + (p.parity ^ 1)) * kCursorStride], (unsigned int)__popcll(mask));
    }
    base = (unsigned int)__shfl((int)base, __ffsll((long long)mask) - 1, 64);
    if (defer) { p.state.deferred[list][p.parity ^ 1][base + laneRank(mask)] = slot; }
}

// ------------------------------------------------------------------------- trace

// amdgpu_waves_per_eu(5, 5) caps the kernel at 96 VGPRs: the trace waves resident on a SIMD
// (3 or 4) must leave registers for a wave of the other pool's k_shade (104), or the two pools
// stop overlapping.
#ifndef PATHED_TRACE_WAVES
#define PATHED_TRACE_WAVES 5
#endif
#ifndef PATHED_EXP_LISTS_ONLY
#define PATHED_EXP_LISTS_ONLY 0   // experiment builds: 1 = the trace kernel writes its lists, the per-slot k_shade shades (and ignores them)
#endif

// LISTS: the split shade stage follows.  The slot of every closest-hit ray that finishes joins the hit list or the miss
// list (see kCtrListCount); a miss writes no hit record; a closest-hit ray that is parked puts its slot on hold itself
// (no shade kernel visits a slot that is on no list).
// SPHERES = false: scenes without sphere primitives (the mesh configurations): no sphere code in the kernel
// FORMAT: the tree is walked as 0 its 128-byte float nodes, 1 nodeQ (64-byte compressed), 2 node8 (8-wide compressed) (trace.h)
template <int STACK, bool LDS_SCENE, bool COUNT, bool LISTS, bool SPHERES = true, int FORMAT = 0>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PATHED_TRACE_WAVES, PATHED_TRACE_WAVES))) void k_trace(RenderParams p)
{
    extern __shared__ float4 ldsRaw[];
    // LDS: [STACK + 1][kBlock] traversal stack rows (the last one is scratch), then 2 x kBlock
    // float4 of ray staging, then (LISTS) 2 x 128 words per wave of list entries, then (LDS_SCENE) the tree
    LaneStack stack;
    stack.lds = reinterpret_cast<int *>(ldsRaw) + threadIdx.x;
    stack.overflowStride = (size_t)gridDim.x * kBlock;
    stack.overflow = p.stackOverflow + ((size_t)blockIdx.x * kBlock + threadIdx.x);

    static_assert(!(FORMAT != 0 && LDS_SCENE), "the LDS copy of a small tree is the uncompressed one");
    TraceGeometry geometry;
    geometry.nodes = FORMAT != 0 ? p.scene.nodesQ : p.scene.nodes;
    geometry.tris = p.scene.leafTris;
    geometry.nNodes = p.scene.nNodes;
    geometry.nTris = p.scene.nTris;
    geometry.spheres = p.scene.spheres;
    geometry.nSpheres = p.scene.nLinearSpheres;

    float4 *stageO = ldsRaw + ((STACK + 1) * kBlock) / 4 + (threadIdx.x >> 6) * kCard;  // this wave's kCard entries
    float4 *stageD = stageO + kBlock * kCardRounds;

    constexpr int kListQuads = LISTS ? kWavesPerBlock * 2 * 128 / 4 : 0;   // float4 units
    if (LDS_SCENE) {
        // small scenes: the whole BVH + leaf triangles are staged in LDS once per block
        float4 *ldsNodes = ldsRaw + ((STACK + 1) * kBlock) / 4 + 2 * kBlock * kCardRounds + kListQuads;
        float4 *ldsTris = ldsNodes + 8 * p.scene.nNodes;
        for (int i = threadIdx.x; i < 8 * p.scene.nNodes; i += kBlock) { ldsNodes[i] = p.scene.nodes[i]; }
        for (int i = threadIdx.x; i < 3 * p.scene.nTris; i += kBlock) { ldsTris[i] = p.scene.leafTris[i]; }
        __syncthreads();
        geometry.nodes = ldsNodes;
        geometry.tris = ldsTris;
    }

    const int lane = threadIdx.x & 63;
    const unsigned int waveId = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);

    ListWriter hitWriter, missWriter;
    listWriterInit(hitWriter, nullptr);
    listWriterInit(missWriter, nullptr);
    if (LISTS) {
        unsigned int *listBuffers = reinterpret_cast<unsigned int *>(ldsRaw + ((STACK + 1) * kBlock) / 4 + 2 * kBlock * kCardRounds)
            + (threadIdx.x >> 6) * 256;
        listWriterInit(hitWriter, listBuffers);
        listWriterInit(missWriter, listBuffers + 128);
        // the cursors of the other parity are the next trace launch's (and the deferred lists of that parity the ones the
        // shade stage after THIS launch appends to): nobody uses them while this launch runs
        if (blockIdx.x == 0) {
            for (int i = threadIdx.x; i < 2 * kListShards; i += kBlock) {
                const int list = i / kListShards, shard = i % kListShards;
                p.counters[kCtrListCount + ((list * 2 + (p.parity ^ 1)) * kListShards + shard) * kCursorStride] = 0u;
            }
            if (threadIdx.x < 2) { p.counters[kCtrDeferred + (threadIdx.x * 2 + (p.parity ^ 1)) * kCursorStride] = 0u; }
        }
    }

    // Ray pool of this launch: item i < nSlots is the closest-hit ray of slot i; item
    // nSlots + j is entry j of the shadow-ray list the previous k_shade compacted.
    // The pool is cut into 64-item cards.  Persistent waves draw cards from kTraceShards
    // sharded cursors (one wave-level atomic per 64 rays; a wave whose home shard has run dry
    // moves on to the next one), so every wave keeps drawing until the whole pool is dealt and
    // no wave is left holding a long private queue.
    // A card is fetched by the whole wave at once -- 64 coalesced ray loads, one HBM round trip --
    // and its live rays are compacted into the wave's LDS staging rows; lanes whose ray has
    // finished are then refilled from LDS (ballot + prefix popcount), never from HBM.  Refilling
    // lane by lane from the state streams cost two dependent HBM latencies per refill round and
    // was a third of the kernel's time.
    const unsigned int shadowCount = p.counters[kCtrShadowCount + (p.parity ^ 1) * kCursorStride];
    const unsigned int totalItems = (unsigned int)p.nSlots + shadowCount;
    const unsigned int totalCards = (totalItems + kCard - 1u) / kCard;   // nSlots is a multiple of kBlock
    // parking pays only in a launch with plenty of rays per wave; at the end of a render, when a
    // launch carries a few stragglers, they are simply run to completion
    const bool mayPark = p.counters[kCtrRemaining] >= (unsigned int)p.parkMinCardsPerWave * kCard * gridDim.x * kWavesPerBlock;
    unsigned int shard = waveId % kTraceShards, shardsTried = 0;
    unsigned int stagedCount = 0, stagedPos = 0;                  // wave-uniform
    bool stagedShadow = false;                                    // the staged card holds shadow rays
    bool exhausted = false;                                       // no card left to draw

    auto stageCard = [&]() {
        unsigned int card = 0;
        bool have = false;
        while (shardsTried < kTraceShards) {
            unsigned int ticket = 0;
            if (lane == 0) { ticket = atomicAdd(&p.counters[kCtrTraceCursor + shard * kCursorStride], 1u); }
            ticket = (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
            card = ticket * kTraceShards + shard;
            if (card < totalCards) { have = true; break; }
            shard = (shard + 1u) % kTraceShards;   // this shard is dealt out: try the next one
            shardsTried++;
        }
        if (!have) { exhausted = true; return; }
        stagedShadow = card * kCard >= (unsigned int)p.nSlots;    // nSlots is a multiple of kCard
        bool valid[kCardRounds];
        float4 first[kCardRounds], second[kCardRounds];
        #pragma unroll
        for (int r = 0; r < kCardRounds; r++) {
            const unsigned int item = card * kCard + (unsigned int)(r * 64 + lane);
            valid[r] = false;
            first[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            second[r] = first[r];
            if (!stagedShadow) {
                const float4 rd = p.state.rayD[item];
                const float4 ro = p.state.rayO[item];
                valid[r] = !(floatAsInt(rd.w) & (kStDone | kStHold | kStLocal));
                first[r] = ro;
                second[r] = make_float4(rd.x, rd.y, rd.z, intAsFloat((int)item));  // .w = the slot
            } else {
                const unsigned int entry = item - (unsigned int)p.nSlots;
                valid[r] = entry < shadowCount;
                if (valid[r]) {
                    first[r] = p.state.shO[entry];    // .w = tfar
                    second[r] = p.state.shD[entry];   // .w = the slot
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        unsigned int staged = 0;
        #pragma unroll
        for (int r = 0; r < kCardRounds; r++) {
            const unsigned long long validMask = __ballot(valid[r]);
            const unsigned int rank = staged + __builtin_amdgcn_mbcnt_hi(
                (unsigned int)(validMask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)validMask, 0u));
            if (valid[r]) { stageO[rank] = first[r]; stageD[rank] = second[r]; }
            staged += (unsigned int)__popcll(validMask);
        }
        __builtin_amdgcn_wave_barrier();
        stagedCount = staged;
        stagedPos = 0;
    };

    TraceCounters counters;
    counters.boxes = 0;
    counters.tris = 0;
    unsigned int closestRays = 0, shadowRays = 0;
    unsigned int maxBoxes = 0, rayBoxesStart = 0;
    unsigned int waveSteps = 0, laneSteps = 0, refills = 0;
    unsigned long long waveStart = 0, tailStart = 0;
    unsigned int tailSteps = 0, tailLaneSteps = 0;
    if (COUNT) { waveStart = __builtin_amdgcn_s_memtime(); }

    LaneRay ray;
    bool active = false;
    unsigned int target = 0;  // slot of the ray in flight on this lane

    // rays this wave parked at the end of the pool's previous launch
    unsigned long long parkedMask = 0ull;
    bool restored = false;
    unsigned int stepsSinceLastCard = 0;
    if (p.suspendLanes > 0) {
        parkedMask = p.suspendMask[waveId];
        if ((parkedMask >> lane) & 1ull) {
            const int *save = p.suspendData + (size_t)waveId * (size_t)((kSaveWords + p.maxStack) * 64) + lane;
            const V3 o = v3(intAsFloat(save[0 * 64]), intAsFloat(save[1 * 64]), intAsFloat(save[2 * 64]));
            const V3 d = v3(intAsFloat(save[3 * 64]), intAsFloat(save[4 * 64]), intAsFloat(save[5 * 64]));
            const int flags = save[14 * 64];
            laneRayInit(ray, o, d, intAsFloat(save[6 * 64]), intAsFloat(save[7 * 64]), (flags & 1) != 0);
            ray.best = intAsFloat(save[8 * 64]);
            ray.bestU = intAsFloat(save[9 * 64]);
            ray.bestV = intAsFloat(save[10 * 64]);
            ray.bestPrim = save[11 * 64];
            ray.current = save[12 * 64];
            ray.pendingLeaf = save[13 * 64];
            ray.sp = save[15 * 64];
            target = (unsigned int)save[16 * 64];
            if (COUNT) { rayBoxesStart = counters.boxes - (unsigned int)save[17 * 64]; }
            for (int k = 0; k < ray.sp; k++) { stackWrite<STACK, kBlock>(stack, k, save[(kSaveWords + k) * 64]); }
            active = true;
            restored = true;
        }
        parkedMask = 0ull;
    }

    unsigned long long refillCycles = 0, innerCycles = 0, leafCycles = 0;
    unsigned int innerSteps = 0;
    while (true) {
        unsigned long long stamp = 0;
        if (COUNT) { stamp = __builtin_amdgcn_s_memtime(); }
        if (LISTS) {
            // the closest-hit rays that finished in the last burst: their lanes are idle and still hold slot and result
            const bool finished = (target & kListPending) != 0u;
            if (__ballot(finished) != 0ull) {
                target &= ~kListPending;
                listAppend(hitWriter, p, kListHit, finished && ray.bestPrim >= 0, target);
                listAppend(missWriter, p, kListMiss, finished && ray.bestPrim < 0, target);
            }
        }
        while (true) {
            const unsigned long long idleMask = __ballot(!active);
            if (idleMask == 0ull) { break; }
            if (stagedPos == stagedCount) {
                if (exhausted) { break; }
                stageCard();
                continue;
            }
            const unsigned int rank = __builtin_amdgcn_mbcnt_hi(
                (unsigned int)(idleMask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)idleMask, 0u));  // idle lanes below this one
            const unsigned int available = stagedCount - stagedPos;
            if (!active && rank < available) {
                const float4 first = stageO[stagedPos + rank];
                const float4 second = stageD[stagedPos + rank];
                laneRayInit(ray, v3(first.x, first.y, first.z), v3(second.x, second.y, second.z), PATHED_TNEAR,
                            stagedShadow ? first.w : PATHED_TFAR, stagedShadow);
                target = (unsigned int)floatAsInt(second.w);
                active = true;
                restored = false;
                if (COUNT) {
                    if (stagedShadow) { shadowRays++; } else { closestRays++; }
                    rayBoxesStart = counters.boxes;
                }
            }
            const unsigned int wanted = (unsigned int)__popcll(idleMask);
            stagedPos += wanted < available ? wanted : available;
        }
        const bool dry = exhausted && stagedPos == stagedCount;   // nothing left to hand out

        if (COUNT) { refillCycles += __builtin_amdgcn_s_memtime() - stamp; }
        if (__ballot(active) == 0ull) { break; }  // only reached with the pool dealt out

        // traversal burst.  Each step the wave runs ONE phase: the triangle phase when enough lanes
        // have a leaf pending (or nothing else can run), else the inner-node phase; lanes in the
        // other mode sit the step out.  The burst ends when the wave has thinned out enough to be
        // worth refilling.
        if (COUNT && lane == 0) { refills++; }
        while (true) {
            const unsigned long long leafMask = __ballot(active && ray.pendingLeaf != 0);
            const unsigned long long innerMask = __ballot(active && ray.pendingLeaf == 0);
            const bool trianglePhase = __popcll(leafMask) >= kLeafThreshold || innerMask == 0ull;
            if (COUNT) {
                waveSteps++;
                laneSteps += (unsigned int)__popcll(trianglePhase ? leafMask : innerMask);
                if (dry) {
                    if (tailSteps == 0) { tailStart = __builtin_amdgcn_s_memtime(); }
                    tailSteps++;
                    tailLaneSteps += (unsigned int)__popcll(trianglePhase ? leafMask : innerMask);
                }
            }
            bool done = false;
            if (COUNT) { stamp = __builtin_amdgcn_s_memtime(); }
            if (trianglePhase) {
                if (active && ray.pendingLeaf != 0) { done = leafStep<COUNT, STACK, kBlock, SPHERES>(geometry, stack, ray, &counters); }
            } else {
                if (active && ray.pendingLeaf == 0) {
                    done = (geometry.nNodes == 0)
                        || (FORMAT == 2 ? innerStep8<COUNT, STACK, kBlock>(geometry, stack, p.maxStack, ray, &counters)
                                        : innerStep<COUNT, STACK, kBlock, PATHED_WARM_LINES && !LDS_SCENE, FORMAT == 1>(geometry, stack, p.maxStack, ray, &counters));
                }
            }
            if (COUNT) {
                const unsigned long long elapsed = __builtin_amdgcn_s_memtime() - stamp;
                if (trianglePhase) { leafCycles += elapsed; } else { innerCycles += elapsed; innerSteps++; }
            }
            if (done) {
                finishRay<SPHERES>(geometry, ray);
                if (ray.anyHit) {
                    if (ray.occluded) { p.state.pend[target] = make_float4(0.f, 0.f, 0.f, 0.f); }
                    else if (restored) { reinterpret_cast<int *>(p.state.pend + target)[3] = 0; }
                } else if (!LISTS || PATHED_EXP_LISTS_ONLY || ray.bestPrim >= 0) {
                    p.state.hit[target] = make_float4(ray.best, ray.bestU, ray.bestV, intAsFloat(ray.bestPrim));
                }
                if (COUNT) {
                    const unsigned int delta = counters.boxes - rayBoxesStart;
                    maxBoxes = delta > maxBoxes ? delta : maxBoxes;
                }
                active = false;
                // the slot joins its list when the burst is over (kListPending): no list code inside this loop
                if (LISTS && !ray.anyHit) { target |= kListPending; }
            }
            const unsigned long long activeMask = __ballot(active);
            if (activeMask == 0ull) { break; }
            if (!dry && __popcll(activeMask) < kRefillThreshold) { break; }
            if (dry) { stepsSinceLastCard++; }
            if (dry && mayPark && __popcll(activeMask) < p.suspendLanes && stepsSinceLastCard >= (unsigned int)p.suspendPatience) {
                if (active) {
                    int *save = p.suspendData + (size_t)waveId * (size_t)((kSaveWords + p.maxStack) * 64) + lane;
                    save[0 * 64] = floatAsInt(ray.o.x); save[1 * 64] = floatAsInt(ray.o.y); save[2 * 64] = floatAsInt(ray.o.z);
                    save[3 * 64] = floatAsInt(ray.d.x); save[4 * 64] = floatAsInt(ray.d.y); save[5 * 64] = floatAsInt(ray.d.z);
                    save[6 * 64] = floatAsInt(ray.tnear);
                    save[7 * 64] = floatAsInt(ray.tfar);
                    save[8 * 64] = floatAsInt(ray.best);
                    save[9 * 64] = floatAsInt(ray.bestU);
                    save[10 * 64] = floatAsInt(ray.bestV);
                    save[11 * 64] = ray.bestPrim;
                    save[12 * 64] = ray.current;
                    save[13 * 64] = ray.pendingLeaf;
                    save[14 * 64] = ray.anyHit ? 1 : 0;
                    save[15 * 64] = ray.sp;
                    save[16 * 64] = (int)target;
                    save[17 * 64] = COUNT ? (int)(counters.boxes - rayBoxesStart) : 0;
                    for (int k = 0; k < ray.sp; k++) { save[(kSaveWords + k) * 64] = stackRead<STACK, kBlock>(stack, k); }
                    if (ray.anyHit) { reinterpret_cast<int *>(p.state.pend + target)[3] = kShadowSuspended; }
                    else if (LISTS) { reinterpret_cast<int *>(p.state.rayD + target)[3] |= kStHold; }
                    else { reinterpret_cast<int *>(p.state.hit + target)[3] = kPrimSuspended; }
                    active = false;
                }
                parkedMask = activeMask;
                if (COUNT && lane == 0) { atomicAdd(&p.stats[kStatParked], (unsigned long long)__popcll(activeMask)); }
                break;
            }
        }
    }
    if (p.suspendLanes > 0 && lane == 0) { p.suspendMask[waveId] = parkedMask; }
    if (LISTS) {
        if (hitWriter.count != 0u) { listFlush(hitWriter, p, kListHit); }
        if (missWriter.count != 0u) { listFlush(missWriter, p, kListMiss); }
    }

    if (COUNT) {
        atomicAdd(&p.stats[kStatBoxes], (unsigned long long)counters.boxes);
        atomicAdd(&p.stats[kStatTris], (unsigned long long)counters.tris);
        atomicAdd(&p.stats[kStatClosest], (unsigned long long)closestRays);
        atomicAdd(&p.stats[kStatShadow], (unsigned long long)shadowRays);
        atomicMax(&p.stats[kStatMaxBoxes], (unsigned long long)maxBoxes);
        if (lane == 0) {
            atomicAdd(&p.stats[kStatWaveSteps], (unsigned long long)waveSteps);
            atomicAdd(&p.stats[kStatLaneSteps], (unsigned long long)laneSteps);
            atomicAdd(&p.stats[kStatRefills], (unsigned long long)refills);
            atomicAdd(&p.stats[kStatWaveCycles], (unsigned long long)(__builtin_amdgcn_s_memtime() - waveStart));
            atomicMax(&p.stats[kStatWaveCyclesMax], (unsigned long long)(__builtin_amdgcn_s_memtime() - waveStart));
            atomicAdd(&p.stats[kStatRefillCycles], refillCycles);
            atomicAdd(&p.stats[kStatInnerCycles], innerCycles);
            atomicAdd(&p.stats[kStatLeafCycles], leafCycles);
            atomicAdd(&p.stats[kStatInnerSteps], (unsigned long long)innerSteps);
            atomicAdd(&p.stats[kStatTailSteps], (unsigned long long)tailSteps);
            atomicAdd(&p.stats[kStatTailLaneSteps], (unsigned long long)tailLaneSteps);
            if (tailSteps) { atomicAdd(&p.stats[kStatTailCycles], (unsigned long long)(__builtin_amdgcn_s_memtime() - tailStart)); }
        }
    }
}

// Tiny scenes (<= kBruteForceMaxTris triangles, e.g. the 36-triangle Cornell box): every ray
// tests every triangle.  Control flow is wave-uniform, the triangle records are fetched with
// SCALAR loads (uniform address) and broadcast to all 64 lanes, there is no traversal stack, no
// LDS and no dependent memory access at all, so the kernel runs at VALU issue rate with full
// lanes — faster than walking a BVH whose every step diverges.  Hits are identical to the BVH
// path by the intersector specification (equal-t ties resolve by primitive id, not test order).
static const int kBruteForceMaxTris = 64;
static const int kBruteForceMaxSpheres = 16;   // ... and at most this many spheres (each ray tests them one by one)

// The triangle records travel as a KERNEL ARGUMENT (2.3 KB of the 4 KB kernarg segment): kernarg
// reads are s_load from the constant address space, so a uniform index gives true scalar loads.
// (Uniform global_load_dwordx4 still return 1 KiB per wave through the 64 B/clk vector path.)
// Triangles are stored two by two, component-interleaved -- (a.v0x, b.v0x), (a.v0y, b.v0y), ... --
// so that one packed-fp32 instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, full rate on
// CDNA3/4) works on triangle a in its low half and triangle b in its high half.
typedef float f2 __attribute__((ext_vector_type(2)));
static const int kSmallPairWords = 9;  // f2 per pair: v0.xyz, e1.xyz, e2.xyz
struct SmallTris {
    f2 data[kSmallPairWords * (kBruteForceMaxTris / 2)];
};

__device__ inline f2 splat2(float v) { f2 r = { v, v }; return r; }
__device__ inline f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// Phase 1 of the all-triangles intersector for ONE lane's ray (called under wave-uniform control flow):
// a CONSERVATIVE inside test of every triangle on scalar-loaded records, two triangles per packed
// instruction, that only RECORDS the candidates in a per-lane bitmask.  A line through the box pierces 4-6
// triangles, so with 64 lanes "some lane is inside" is true for almost every triangle: doing the division
// and the acceptance logic there made them two thirds of the instructions.  The test is
// intersectTriangle()'s predicate with both det signs folded by multiplying through with det:
// u det >= 0, v det >= 0, (det - u - v) det >= 0, t det >= 0, taken as "not (min < 0)" so that -0,
// underflow and NaN all err on the side of keeping the candidate; phase 2 decides.  Bits are shifted in
// (cand = 2 cand + bit), so triangle k of a 32-triangle word ends up at bit 31 - (k & 31).
// INTERVAL: a candidate must also lie inside the ray's interval, loosely: t >= tnearLow and (FAR) t <= tfarHigh in the same
// det units, (t - tnearLow) det^2 >= 0 and (tfarHigh - t) det^2 >= 0, one and two packed operations more.  The bounds are
// the query's own moved OUTWARD by half of tnear (candidateBounds), far more than the two evaluations of t can differ by, so
// phase 2 still sees every triangle it could accept; what goes are the triangle the ray starts on (t = 0) and, for a
// shadow ray, the light's own triangle and everything behind it -- two of the three candidates of a shadow ray in the
// Cornell box, one of the two of a path's ray, and with them half of the turns of the resolve loops.
__device__ inline float candidateNear(float tnear) { return 0.5f * tnear; }
__device__ inline float candidateFar(float tfar) { return tfar + 0.5f * PATHED_TNEAR + 1e-5f * fabsf(tfar); }

template <bool FAR = false>
__device__ __forceinline__ void smallCandidates(const f2 *pairRecords, int nTris, V3 origin, V3 direction, unsigned int *low, unsigned int *high,
                                                float tnearLow = 0.f, float tfarHigh = 0.f)
{
    const int nPairs = (nTris + 1) / 2;
    unsigned int candidatesLow = 0, candidatesHigh = 0;
    const f2 dx = splat2(direction.x), dy = splat2(direction.y), dz = splat2(direction.z);
    const f2 ox = splat2(origin.x), oy = splat2(origin.y), oz = splat2(origin.z);
    const f2 nearLow = splat2(-tnearLow), farHigh = splat2(tfarHigh);
    // one pair of triangles: returns (bit of a) * 2 + bit of b
    auto testPair = [&](int pair) -> unsigned int {
        // uniform index into the kernarg segment -> scalar loads, broadcast to the wave
        const f2 *record = pairRecords + kSmallPairWords * pair;
        const f2 v0x = record[0], v0y = record[1], v0z = record[2];
        const f2 e1x = record[3], e1y = record[4], e1z = record[5];
        const f2 e2x = record[6], e2y = record[7], e2z = record[8];
        // pvec = d x e2, det = e1 . pvec
        const f2 px = fma2(dy, e2z, -(dz * e2y));
        const f2 py = fma2(dz, e2x, -(dx * e2z));
        const f2 pz = fma2(dx, e2y, -(dy * e2x));
        const f2 det = fma2(e1x, px, fma2(e1y, py, e1z * pz));
        const f2 tx = ox - v0x, ty = oy - v0y, tz = oz - v0z;
        const f2 uScaled = fma2(tx, px, fma2(ty, py, tz * pz));
        // qvec = tvec x e1
        const f2 qx = fma2(ty, e1z, -(tz * e1y));
        const f2 qy = fma2(tz, e1x, -(tx * e1z));
        const f2 qz = fma2(tx, e1y, -(ty * e1x));
        const f2 vScaled = fma2(dx, qx, fma2(dy, qy, dz * qz));
        const f2 tScaled = fma2(e2x, qx, fma2(e2y, qy, e2z * qz));
        const f2 a = uScaled * det, b = vScaled * det, c = (det - (uScaled + vScaled)) * det, e = fma2(nearLow, det, tScaled) * det;
        float worstA = fminf(fminf(a.x, b.x), fminf(c.x, e.x));
        float worstB = fminf(fminf(a.y, b.y), fminf(c.y, e.y));
        if (FAR) {
            const f2 g = fma2(farHigh, det, -tScaled) * det;
            worstA = fminf(worstA, g.x);
            worstB = fminf(worstB, g.y);
        }
        return ((worstA < 0.f) ? 0u : 2u) | ((worstB < 0.f) ? 0u : 1u);
    };
    const int lowPairs = nPairs < 16 ? nPairs : 16;
    for (int pair = 0; pair < lowPairs; pair++) { candidatesLow = (candidatesLow << 2) | testPair(pair); }
    for (int pair = 16; pair < nPairs; pair++) { candidatesHigh = (candidatesHigh << 2) | testPair(pair); }
    // left-align: the last pair shifted in sits at bit 0; pad pairs and a padding triangle drop out
    const int lowTris = nTris < 32 ? nTris : 32;
    const int lowShifted = 2 * (nPairs < 16 ? nPairs : 16);
    if (lowShifted > 0 && lowShifted < 32) { candidatesLow <<= 32 - lowShifted; }
    candidatesLow &= lowTris > 0 ? 0xFFFFFFFFu << (32 - lowTris) : 0u;
    const int highTris = nTris - 32;
    if (highTris > 0) {
        const int highShifted = 2 * (nPairs - 16);
        if (highShifted < 32) { candidatesHigh <<= 32 - highShifted; }
        candidatesHigh &= 0xFFFFFFFFu << (32 - highTris);
    } else {
        candidatesHigh = 0u;
    }
    *low = candidatesLow;
    *high = candidatesHigh;
}

// Phase 1 for TWO rays that leave the same point (a vertex's continuation ray and its shadow ray): tvec = o - v0,
// qvec = tvec x e1 and t det = e2 . qvec do not depend on the direction, so the pair costs 12 + 2 x 21 packed
// operations per two triangles instead of 2 x 33.  Same expressions as smallCandidates, ray by ray.
// Ray B is the shadow ray: its candidates are also held to its far bound (smallCandidates: INTERVAL).
__device__ __forceinline__ void smallCandidatesPair(const f2 *pairRecords, int nTris, V3 origin, V3 directionA, V3 directionB,
                                                    unsigned int *lowA, unsigned int *highA, unsigned int *lowB, unsigned int *highB,
                                                    float tnearLow = 0.f, float tfarHighB = 3e38f)
{
    const int nPairs = (nTris + 1) / 2;
    unsigned int aLow = 0, aHigh = 0, bLow = 0, bHigh = 0;
    const f2 ax = splat2(directionA.x), ay = splat2(directionA.y), az = splat2(directionA.z);
    const f2 bx = splat2(directionB.x), by = splat2(directionB.y), bz = splat2(directionB.z);
    const f2 ox = splat2(origin.x), oy = splat2(origin.y), oz = splat2(origin.z);
    const f2 nearLow = splat2(-tnearLow), farHighB = splat2(tfarHighB);
    auto testPair = [&](int pair, unsigned int *bitsA, unsigned int *bitsB) {
        const f2 *record = pairRecords + kSmallPairWords * pair;
        const f2 v0x = record[0], v0y = record[1], v0z = record[2];
        const f2 e1x = record[3], e1y = record[4], e1z = record[5];
        const f2 e2x = record[6], e2y = record[7], e2z = record[8];
        const f2 tx = ox - v0x, ty = oy - v0y, tz = oz - v0z;
        // qvec = tvec x e1
        const f2 qx = fma2(ty, e1z, -(tz * e1y));
        const f2 qy = fma2(tz, e1x, -(tx * e1z));
        const f2 qz = fma2(tx, e1y, -(ty * e1x));
        const f2 tScaled = fma2(e2x, qx, fma2(e2y, qy, e2z * qz));
        auto oneRay = [&](f2 dx, f2 dy, f2 dz, bool far) -> unsigned int {
            // pvec = d x e2, det = e1 . pvec
            const f2 px = fma2(dy, e2z, -(dz * e2y));
            const f2 py = fma2(dz, e2x, -(dx * e2z));
            const f2 pz = fma2(dx, e2y, -(dy * e2x));
            const f2 det = fma2(e1x, px, fma2(e1y, py, e1z * pz));
            const f2 uScaled = fma2(tx, px, fma2(ty, py, tz * pz));
            const f2 vScaled = fma2(dx, qx, fma2(dy, qy, dz * qz));
            const f2 a = uScaled * det, b = vScaled * det, c = (det - (uScaled + vScaled)) * det, e = fma2(nearLow, det, tScaled) * det;
            float worstA = fminf(fminf(a.x, b.x), fminf(c.x, e.x));
            float worstB = fminf(fminf(a.y, b.y), fminf(c.y, e.y));
            if (far) {
                const f2 g = fma2(farHighB, det, -tScaled) * det;
                worstA = fminf(worstA, g.x);
                worstB = fminf(worstB, g.y);
            }
            return ((worstA < 0.f) ? 0u : 2u) | ((worstB < 0.f) ? 0u : 1u);
        };
        *bitsA = oneRay(ax, ay, az, false);
        *bitsB = oneRay(bx, by, bz, true);
    };
    const int lowPairs = nPairs < 16 ? nPairs : 16;
    for (int pair = 0; pair < lowPairs; pair++) {
        unsigned int bitsA, bitsB;
        testPair(pair, &bitsA, &bitsB);
        aLow = (aLow << 2) | bitsA;
        bLow = (bLow << 2) | bitsB;
    }
    for (int pair = 16; pair < nPairs; pair++) {
        unsigned int bitsA, bitsB;
        testPair(pair, &bitsA, &bitsB);
        aHigh = (aHigh << 2) | bitsA;
        bHigh = (bHigh << 2) | bitsB;
    }
    // left-align as smallCandidates does
    const int lowTris = nTris < 32 ? nTris : 32;
    const int lowShifted = 2 * (nPairs < 16 ? nPairs : 16);
    if (lowShifted > 0 && lowShifted < 32) { aLow <<= 32 - lowShifted; bLow <<= 32 - lowShifted; }
    const unsigned int lowMask = lowTris > 0 ? 0xFFFFFFFFu << (32 - lowTris) : 0u;
    aLow &= lowMask;
    bLow &= lowMask;
    const int highTris = nTris - 32;
    if (highTris > 0) {
        const int highShifted = 2 * (nPairs - 16);
        if (highShifted < 32) { aHigh <<= 32 - highShifted; bHigh <<= 32 - highShifted; }
        aHigh &= 0xFFFFFFFFu << (32 - highTris);
        bHigh &= 0xFFFFFFFFu << (32 - highTris);
    } else {
        aHigh = 0u;
        bHigh = 0u;
    }
    *lowA = aLow; *highA = aHigh; *lowB = bLow; *highB = bHigh;
}

// word = 2 word + bit in ONE instruction: v_addc_co_u32 with a lane mask as the carry-in (a select, a shift and an or
// otherwise).  The masks are the compare instructions' own results (ballots of single comparisons fold into the compare;
// combining them is scalar work), the carry-out is dropped.
__device__ __forceinline__ unsigned int shiftInBit(unsigned int word, unsigned long long laneMask)
{
    unsigned int result;
    unsigned long long carryOut;
    asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(result), "=s"(carryOut) : "v"(word), "s"(laneMask));
    return result;
}

// Phase 1 over the ITEM records of the fused kernel (small_items.h): first the parallelograms -- one Moeller-Trumbore
// evaluation for the two triangles of a quad, every bound with the tolerance derived there -- then the triangles that found
// no partner, two per packed instruction with the exact test above.  Bits are shifted in in item order (triangle k of a word
// ends up at bit 31 - (k & 31), as smallCandidates leaves them); phase 2 indexes the item-ordered copy of the triangle records.
// SHADOW: ray B (the vertex's shadow ray, held to its far bound) exists; otherwise only ray A is tested.
// QUADS / LONE: which of the two sections the instantiation contains (a scene made of quads only needs no code, and no
// registers, for the other test).
template <bool SHADOW, bool QUADS = true, bool LONE = true>
__device__ __forceinline__ void smallCandidatesItems(const f2 *records, int nQuads, int nTris, float kappaT, V3 origin, V3 directionA, V3 directionB,
                                                     unsigned int *lowA, unsigned int *highA, unsigned int *lowB, unsigned int *highB,
                                                     float tnearLow, float tfarHighB)
{
    // candidate bits of ray A / ray B, appended at the low end of a 64-bit word (two registers: one v_alignbit_b32 for the
    // upper half, then one v_addc per bit -- shiftInBit -- for the lower): after the last item triangle 0 is bit nTris - 1
    unsigned int loA = 0u, hiA = 0u, loB = 0u, hiB = 0u;
    const f2 ax = splat2(directionA.x), ay = splat2(directionA.y), az = splat2(directionA.z);
    const f2 bx = splat2(directionB.x), by = splat2(directionB.y), bz = splat2(directionB.z);
    const f2 ox = splat2(origin.x), oy = splat2(origin.y), oz = splat2(origin.z);
    const f2 nearLow = splat2(-tnearLow), farHighB = splat2(fminf(tfarHighB, 1e30f));
    const f2 kappaLength = splat2(kappaT);

    // ---- parallelograms, two per packed instruction
    const int nQuadPairs = QUADS ? (nQuads + 1) >> 1 : 0;
    for (int pair = 0; QUADS && pair < nQuadPairs; pair++) {
        const f2 *record = records + kSmallQuadWords * pair;   // uniform index into the kernarg segment: scalar loads
        const f2 c0x = record[0], c0y = record[1], c0z = record[2];
        const f2 a1x = record[3], a1y = record[4], a1z = record[5];
        const f2 a2x = record[6], a2y = record[7], a2z = record[8];
        const f2 k2UV = record[9], amaxSquared = record[10], k2T = record[11], cD = record[13], kappaUV = record[15];
        const f2 tx = ox - c0x, ty = oy - c0y, tz = oz - c0z;
        const f2 qx = fma2(ty, a1z, -(tz * a1y));
        const f2 qy = fma2(tz, a1x, -(tx * a1z));
        const f2 qz = fma2(tx, a1y, -(ty * a1x));
        const f2 tScaled = fma2(a2x, qx, fma2(a2y, qy, a2z * qz));
        // the part of the tolerances that grows with the distance of the origin (small_items.h): K2 |o - c0|^2 + K0
        const f2 tt = fma2(tx, tx, fma2(ty, ty, tz * tz));
        const f2 reachSquared = tt + amaxSquared;   // bounds r^2 / 2 (small_items.h)
        const f2 originUV = k2UV * reachSquared, originT = k2T * reachSquared;
        auto oneRay = [&](f2 dx, f2 dy, f2 dz, bool far, unsigned int &lo, unsigned int &hi) {
            const f2 px = fma2(dy, a2z, -(dz * a2y));
            const f2 py = fma2(dz, a2x, -(dx * a2z));
            const f2 pz = fma2(dx, a2y, -(dy * a2x));
            const f2 det = fma2(a1x, px, fma2(a1y, py, a1z * pz));
            const f2 uScaled = fma2(tx, px, fma2(ty, py, tz * pz));     // alpha det
            const f2 vScaled = fma2(dx, qx, fma2(dy, qy, dz * qz));     // beta det
            const f2 dd = det * det;
            // 0 <= alpha, beta <= 1 in det^2 units, centred: |alpha det^2 - det^2 / 2| <= det^2 / 2 (+ tolerance), the same for beta
            const f2 half = dd * 0.5f;
            const f2 uCentred = fma2(uScaled, det, -half), vCentred = fma2(vScaled, det, -half);
            const f2 diagonal = uCentred - vCentred;                    // (alpha - beta) det^2: >= 0 triangle A, <= 0 triangle B
            const f2 tolUV = fma2(kappaUV, dd, originUV);   // kappaUV: 1e-5 + how far a not-quite-parallelogram sticks out of the unit square
            const f2 reach = half + tolUV;
            const f2 tolT = fma2(kappaLength, dd, originT);
            const f2 nearBound = fma2(nearLow, det, tScaled) * det;     // (t - tnearLow) det^2
            // "x > bound" rejects: -0, underflow and NaN all keep the candidate.  Lane masks (scalar registers) from here on
            #define PATHED_LANES(condition) __builtin_amdgcn_ballot_w64(condition)
            unsigned long long rejectX = PATHED_LANES(fmaxf(fabsf(uCentred.x), fabsf(vCentred.x)) > reach.x) | PATHED_LANES(nearBound.x < -tolT.x);
            unsigned long long rejectY = PATHED_LANES(fmaxf(fabsf(uCentred.y), fabsf(vCentred.y)) > reach.y) | PATHED_LANES(nearBound.y < -tolT.y);
            if (far) {
                const f2 farBound = fma2(farHighB, det, -tScaled) * det;   // (tfarHigh - t) det^2
                const f2 tolFar = fma2(farHighB, fma2(splat2(kSmallKappaFar), dd, cD), tolT);
                rejectX |= PATHED_LANES(farBound.x < -tolFar.x);
                rejectY |= PATHED_LANES(farBound.y < -tolFar.y);
            }
            // (a det too small to trust its sign: the tolerances cover it, small_items.h)
            const f2 twice = tolUV + tolUV;
            const float twiceX = twice.x, twiceY = twice.y;
            hi = __builtin_amdgcn_alignbit(hi, lo, 28);
            lo = shiftInBit(lo, ~(rejectX | PATHED_LANES(diagonal.x < -twiceX)));   // first quad: triangle A, then B
            lo = shiftInBit(lo, ~(rejectX | PATHED_LANES(diagonal.x > twiceX)));
            lo = shiftInBit(lo, ~(rejectY | PATHED_LANES(diagonal.y < -twiceY)));   // second quad
            lo = shiftInBit(lo, ~(rejectY | PATHED_LANES(diagonal.y > twiceY)));
        };
        oneRay(ax, ay, az, false, loA, hiA);
        __builtin_amdgcn_sched_barrier(0);   // one ray's temporaries at a time: interleaved, the two chains spilled 39 dwords of path state
        if (SHADOW) { oneRay(bx, by, bz, true, loB, hiB); }
        __builtin_amdgcn_sched_barrier(0);
    }
    unsigned long long accA, accB;
    if (QUADS && (nQuads & 1)) {   // an odd count: the last pair's second half is padding
        accA = (((unsigned long long)hiA << 32) | loA) >> 2; accB = (((unsigned long long)hiB << 32) | loB) >> 2;
        loA = (unsigned int)accA; hiA = (unsigned int)(accA >> 32); loB = (unsigned int)accB; hiB = (unsigned int)(accB >> 32);
    }

    // ---- triangles without a partner, two per packed instruction: smallCandidatesPair's test, ray by ray
    const f2 *lone = records + kSmallQuadWords * nQuadPairs;
    const int nLone = LONE ? nTris - 2 * nQuads : 0;
    const int nLonePairs = (nLone + 1) >> 1;
    for (int pair = 0; LONE && pair < nLonePairs; pair++) {
        const f2 *record = lone + kSmallPairWords * pair;
        const f2 v0x = record[0], v0y = record[1], v0z = record[2];
        const f2 e1x = record[3], e1y = record[4], e1z = record[5];
        const f2 e2x = record[6], e2y = record[7], e2z = record[8];
        const f2 tx = ox - v0x, ty = oy - v0y, tz = oz - v0z;
        const f2 qx = fma2(ty, e1z, -(tz * e1y));
        const f2 qy = fma2(tz, e1x, -(tx * e1z));
        const f2 qz = fma2(tx, e1y, -(ty * e1x));
        const f2 tScaled = fma2(e2x, qx, fma2(e2y, qy, e2z * qz));
        auto oneRay = [&](f2 dx, f2 dy, f2 dz, bool far, unsigned int &lo, unsigned int &hi) {
            const f2 px = fma2(dy, e2z, -(dz * e2y));
            const f2 py = fma2(dz, e2x, -(dx * e2z));
            const f2 pz = fma2(dx, e2y, -(dy * e2x));
            const f2 det = fma2(e1x, px, fma2(e1y, py, e1z * pz));
            const f2 uScaled = fma2(tx, px, fma2(ty, py, tz * pz));
            const f2 vScaled = fma2(dx, qx, fma2(dy, qy, dz * qz));
            const f2 a = uScaled * det, b = vScaled * det, c = (det - (uScaled + vScaled)) * det, e = fma2(nearLow, det, tScaled) * det;
            float worstA = fminf(fminf(a.x, b.x), fminf(c.x, e.x));
            float worstB = fminf(fminf(a.y, b.y), fminf(c.y, e.y));
            if (far) {
                const f2 g = fma2(splat2(tfarHighB), det, -tScaled) * det;
                worstA = fminf(worstA, g.x);
                worstB = fminf(worstB, g.y);
            }
            hi = __builtin_amdgcn_alignbit(hi, lo, 30);
            lo = shiftInBit(lo, ~PATHED_LANES(worstA < 0.f));
            lo = shiftInBit(lo, ~PATHED_LANES(worstB < 0.f));
            #undef PATHED_LANES
        };
        oneRay(ax, ay, az, false, loA, hiA);
        if (SHADOW) { oneRay(bx, by, bz, true, loB, hiB); }
    }
    accA = ((unsigned long long)hiA << 32) | loA; accB = ((unsigned long long)hiB << 32) | loB;
    if (LONE && (nLone & 1)) { accA >>= 1; accB >>= 1; }

    // left-align: triangle k is bit 31 - (k & 31) of word k >> 5
    const int shift = 64 - nTris;   // 1 <= nTris <= 64
    accA <<= shift;
    accB <<= shift;
    *lowA = (unsigned int)(accA >> 32); *highA = (unsigned int)accA; *lowB = (unsigned int)(accB >> 32); *highB = (unsigned int)accB;
}

// Phase 2 (a wave-level loop: call it from wave-uniform control flow, lanes without a ray pass empty masks):
// the few candidates of each lane go through the ordinary intersector + acceptance rule.
__device__ __forceinline__ void smallResolve(const TraceGeometry &geometry, LaneRay &ray, unsigned int candidatesLow, unsigned int candidatesHigh)
{
    while (__ballot((candidatesLow | candidatesHigh) != 0u) != 0ull) {
        if ((candidatesLow | candidatesHigh) != 0u) {
            int k;  // triangle k of a word is bit 31 - (k & 31): take the highest set bit first
            if (candidatesLow != 0u) { k = __clz((int)candidatesLow); candidatesLow &= ~(0x80000000u >> k); }
            else { k = __clz((int)candidatesHigh); candidatesHigh &= ~(0x80000000u >> k); k += 32; }
            const float4 t0 = geometry.tris[3 * k + 0];
            const float4 t1 = geometry.tris[3 * k + 1];
            const float4 t2 = geometry.tris[3 * k + 2];
            bool terminate = false;
            testLeafTriangle(ray, t0, t1, t2, &terminate);
            if (terminate) { candidatesLow = 0u; candidatesHigh = 0u; }  // shadow ray occluded
        }
    }
}

// Phase 1 of the SPHERE queries of k_path_small<QUADS> (scenes with sphere primitives: the Veach scene's lights), two spheres
// per packed instruction, the path's ray and the shadow ray together (they leave the same point: c0 = centre - o and |c0|^2
// are shared).  A ray's line misses a sphere iff |c0|^2 |d|^2 - (c0 . d)^2 > r^2 |d|^2; intersectSphere (trace.h) decides that
// from the perpendicular's length, in other roundings (both within 16 u |c0|^2 |d|^2 of the truth), so the candidate test keeps
// everything up to 2e-5 |c0|^2 |d|^2 beyond: phase 2 -- intersectSphere + the acceptance rule, on whichever lane is free
// (smallResolveShared) -- decides.  No interval test: the target of a shadow ray ends 1e-3 before the light it aims at, which
// only the exact t can tell.  Bit k of a word = sphere k.
__device__ __forceinline__ void smallSphereCandidates(const RenderParams &p, int nSpheres, V3 origin, V3 directionA, V3 directionB,
                                                      unsigned int *candidatesA, unsigned int *candidatesB)
{
    const f2 ox = splat2(origin.x), oy = splat2(origin.y), oz = splat2(origin.z);
    const f2 ax = splat2(directionA.x), ay = splat2(directionA.y), az = splat2(directionA.z);
    const f2 bx = splat2(directionB.x), by = splat2(directionB.y), bz = splat2(directionB.z);
    const f2 aa = splat2(dot(directionA, directionA)), bb = splat2(dot(directionB, directionB));
    const f2 slack = splat2(2e-5f);
    unsigned int bitsA = 0u, bitsB = 0u;   // sphere 2 pair at bit 2 pair (the words are built from the top pair down)
    const int nPairs = (nSpheres + 1) >> 1;
    for (int pair = nPairs - 1; pair >= 0; pair--) {
        const f2 *record = reinterpret_cast<const f2 *>(&p.spherePairs[pair][0][0]);   // uniform index: scalar loads from the kernarg segment
        const f2 cx = record[0] - ox, cy = record[1] - oy, cz = record[2] - oz, radiusSquared = record[3];
        const f2 cc = fma2(cx, cx, fma2(cy, cy, cz * cz));
        auto oneRay = [&](f2 dx, f2 dy, f2 dz, f2 dd, unsigned int &bits) {
            const f2 cd = fma2(cx, dx, fma2(cy, dy, cz * dz));
            const f2 scale = cc * dd;
            const f2 distance = fma2(-cd, cd, scale);            // |perpendicular|^2 |d|^2
            const f2 bound = fma2(slack, scale, radiusSquared * dd);
            // "x > bound" rejects: NaN keeps the candidate
            bits = shiftInBit(bits, ~__builtin_amdgcn_ballot_w64(distance.y > bound.y));   // sphere 2 pair + 1
            bits = shiftInBit(bits, ~__builtin_amdgcn_ballot_w64(distance.x > bound.x));   // sphere 2 pair
        };
        oneRay(ax, ay, az, aa, bitsA);
        oneRay(bx, by, bz, bb, bitsB);
    }
    const unsigned int present = nSpheres >= 32 ? 0xFFFFFFFFu : (1u << nSpheres) - 1u;   // (an odd count's padding sphere)
    *candidatesA = bitsA & present;
    *candidatesB = bitsB & present;
}

// Phase 2 with the wave's idle lanes lending a hand (k_path_small<QUADS>).  smallResolve runs max-over-lanes turns: 4.6 for the
// path's ray and 2.3 for the shadow ray on Cornell, at a quarter of the lanes, because a few lanes hold 3-5 candidates while
// most hold one (profiles/r3_fused_profile.log).  Here every lane tests the FIRST candidate of its path ray itself (one full
// turn; SHADOW_IN_PLACE: of its shadow ray too), the other candidates of the whole wave are listed in LDS, and lane i tests
// items i, i + 64, ... whoever they belong to: the ray comes over ds_bpermute from the lane that owns it, an accepted hit goes
// back as a 64-bit (t, prim, item) key through an LDS atomic min -- the acceptance rule of testLeafTriangle is "smallest t,
// then smallest primitive id", independent of order -- and an accepted occluder as an atomic or.  u, v of the winner wait in
// the item's slot.  More than CAPACITY items (rare): the owners finish them in place as before.  Same hits bit for bit: every
// candidate still goes through intersectTriangle with its own ray and the same acceptance conditions.
// Measured (profiles/r4_ab_resolve.log): Cornell 1024^2 +5.0 % with the left-overs shared, +6.7 % with the shadow ray's
// candidates shared as well; Veach -0.4 % (few left-overs: the cost is the registers the exchange takes from the shading code).
struct ResolveScratch {
    unsigned long long *best;   // [64]  per lane of the wave: smallest accepted key
    float2 *uv;                 // [CAPACITY]  per item
    unsigned short *items;      // [CAPACITY]  owner lane | candidate << 6 | shadow << 12
    unsigned int *occluded;     // [64]  per lane
    unsigned int *count;        // [1]
};

// SPHERES: sphereCandidates / shadowSphereCandidates (smallSphereCandidates: bit k = sphere k) are resolved on the shared
// list as well -- a shadow ray's own target is always among them, so the exact sphere test would otherwise run for every
// sphere at a handful of lanes each.  Returns false when the list overflowed and the spheres are still to be tested
// (finishRay's loop), true otherwise.
template <int CAPACITY, bool SHADOW_IN_PLACE, bool SPHERES>
__device__ __forceinline__ bool smallResolveShared(const TraceGeometry &geometry, LaneRay &ray, LaneRay &shadowRay,
                                                   unsigned int low, unsigned int high, unsigned int shadowLow, unsigned int shadowHigh,
                                                   unsigned int sphereCandidates, unsigned int shadowSphereCandidates,
                                                   const ResolveScratch &scratch)
{
    if (!SPHERES) { sphereCandidates = 0u; shadowSphereCandidates = 0u; }
    const int lane = threadIdx.x & 63;
    auto takeFirst = [](unsigned int &lowWord, unsigned int &highWord) -> int {
        int k;  // triangle k of a word is bit 31 - (k & 31): take the highest set bit first
        if (lowWord != 0u) { k = __clz((int)lowWord); lowWord &= ~(0x80000000u >> k); }
        else { k = __clz((int)highWord); highWord &= ~(0x80000000u >> k); k += 32; }
        return k;
    };
    // ---- every lane's first candidate of each ray, in place
    if (__ballot((low | high) != 0u) != 0ull) {
        if ((low | high) != 0u) {
            const int k = takeFirst(low, high);
            bool terminate = false;
            testLeafTriangle(ray, geometry.tris[3 * k + 0], geometry.tris[3 * k + 1], geometry.tris[3 * k + 2], &terminate);
        }
    }
    if (SHADOW_IN_PLACE && __ballot((shadowLow | shadowHigh) != 0u) != 0ull) {
        if ((shadowLow | shadowHigh) != 0u) {
            const int k = takeFirst(shadowLow, shadowHigh);
            bool terminate = false;
            testLeafTriangle(shadowRay, geometry.tris[3 * k + 0], geometry.tris[3 * k + 1], geometry.tris[3 * k + 2], &terminate);
            if (terminate) { shadowLow = 0u; shadowHigh = 0u; }  // occluded: the other candidates do not matter
        }
    }
    // ---- the left-overs of the wave.  Sharing them out costs about 1.7 turns of the in-place loop (ten ds_bpermute, the
    // list, the atomics), so it is taken when the loops would run two turns or more: left-overs on both rays, or a lane
    // with two of one kind.
    const unsigned int mineRay = (unsigned int)(__popc(low) + __popc(high) + (SPHERES ? __popc(sphereCandidates) : 0));
    const unsigned int mineShadow = (unsigned int)(__popc(shadowLow) + __popc(shadowHigh) + (SPHERES ? __popc(shadowSphereCandidates) : 0));
    const unsigned int mine = mineRay + mineShadow;
    const bool anyRay = __ballot(mineRay != 0u) != 0ull, anyShadow = __ballot(mineShadow != 0u) != 0ull;
    if (!anyRay && !anyShadow) { return true; }
    const bool anySphere = SPHERES && __ballot((sphereCandidates | shadowSphereCandidates) != 0u) != 0ull;
    bool share = anySphere || (anyRay && anyShadow) || __ballot(mineRay > 1u || mineShadow > 1u) != 0ull;
    unsigned int base = 0u, total = 0u;
    if (share) {
        if (lane == 0) { *scratch.count = 0u; }
        scratch.best[lane] = ~0ull;
        scratch.occluded[lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (mine != 0u) { base = atomicAdd(scratch.count, mine); }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        total = (unsigned int)__builtin_amdgcn_readfirstlane((int)*scratch.count);
        share = total <= (unsigned int)CAPACITY;     // more than the list holds (rare): as before
    }
    if (!share) {
        while (__ballot((low | high) != 0u) != 0ull) {
            if ((low | high) != 0u) {
                const int k = takeFirst(low, high);
                bool terminate = false;
                testLeafTriangle(ray, geometry.tris[3 * k + 0], geometry.tris[3 * k + 1], geometry.tris[3 * k + 2], &terminate);
            }
        }
        while (__ballot((shadowLow | shadowHigh) != 0u) != 0ull) {
            if ((shadowLow | shadowHigh) != 0u) {
                const int k = takeFirst(shadowLow, shadowHigh);
                bool terminate = false;
                testLeafTriangle(shadowRay, geometry.tris[3 * k + 0], geometry.tris[3 * k + 1], geometry.tris[3 * k + 2], &terminate);
                if (terminate) { shadowLow = 0u; shadowHigh = 0u; }
            }
        }
        return !anySphere;
    }
    // the owners list their items (a short loop: a lane holds a handful at most)
    {
        unsigned int at = base;
        while ((low | high) != 0u) { const int k = takeFirst(low, high); scratch.items[at++] = (unsigned short)((unsigned int)lane | ((unsigned int)k << 6)); }
        while ((shadowLow | shadowHigh) != 0u) { const int k = takeFirst(shadowLow, shadowHigh); scratch.items[at++] = (unsigned short)((unsigned int)lane | ((unsigned int)k << 6) | 0x1000u); }
        if (SPHERES) {   // 0x2000: the candidate is sphere k
            while (sphereCandidates != 0u) {
                const int k = __ffs((int)sphereCandidates) - 1; sphereCandidates &= sphereCandidates - 1u;
                scratch.items[at++] = (unsigned short)((unsigned int)lane | ((unsigned int)k << 6) | 0x2000u);
            }
            while (shadowSphereCandidates != 0u) {
                const int k = __ffs((int)shadowSphereCandidates) - 1; shadowSphereCandidates &= shadowSphereCandidates - 1u;
                scratch.items[at++] = (unsigned short)((unsigned int)lane | ((unsigned int)k << 6) | 0x3000u);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // lane i tests item i (+ 64, ...)
    for (unsigned int chunk = 0u; chunk < total; chunk += 64u) {
        const unsigned int index = chunk + (unsigned int)lane;
        const bool helping = index < total;
        const unsigned int item = helping ? scratch.items[index] : 0u;
        const int owner = (int)(item & 63u), k = (int)((item >> 6) & 63u);
        const bool forShadow = (item & 0x1000u) != 0u;
        // the owner's rays (every lane takes part in the exchange)
        const float ox = __shfl(ray.o.x, owner), oy = __shfl(ray.o.y, owner), oz = __shfl(ray.o.z, owner);
        const float ax = __shfl(ray.d.x, owner), ay = __shfl(ray.d.y, owner), az = __shfl(ray.d.z, owner);
        const float bx = __shfl(shadowRay.d.x, owner), by = __shfl(shadowRay.d.y, owner), bz = __shfl(shadowRay.d.z, owner);
        const float shadowFar = __shfl(shadowRay.tfar, owner);
        if (SPHERES && helping && (item & 0x2000u) != 0u) {
            // testSphere (trace.h) for somebody else's ray
            const DSphere sphere = geometry.spheres[k];
            const V3 direction = forShadow ? v3(bx, by, bz) : v3(ax, ay, az);
            float t;
            if (intersectSphere(v3(ox, oy, oz), direction, v3(sphere.centerWorld[0], sphere.centerWorld[1], sphere.centerWorld[2]), sphere.radius, PATHED_TNEAR, &t)
                && t > PATHED_TNEAR) {
                if (forShadow) {
                    if (t <= shadowFar) { atomicOr(&scratch.occluded[owner], 1u); }
                } else if (t <= PATHED_TFAR) {
                    scratch.uv[index] = make_float2(0.f, 0.f);
                    const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32)
                        | ((unsigned long long)(unsigned int)(geometry.nTris + k) << 8) | (unsigned long long)index;
                    atomicMin(&scratch.best[owner], key);
                }
            }
        } else if (helping) {
            const float4 t0 = geometry.tris[3 * k + 0], t1 = geometry.tris[3 * k + 1], t2 = geometry.tris[3 * k + 2];
            const V3 direction = forShadow ? v3(bx, by, bz) : v3(ax, ay, az);
            float t, u, v;
            if (intersectTriangle(v3(ox, oy, oz), direction, v3(t0.x, t0.y, t0.z), v3(t1.x, t1.y, t1.z), v3(t2.x, t2.y, t2.z), &t, &u, &v)) {
                if (t > PATHED_TNEAR) {   // the tnear of both queries (laneRayInit in k_path_small)
                    if (forShadow) {
                        if (t <= shadowFar) { atomicOr(&scratch.occluded[owner], 1u); }
                    } else if (t <= PATHED_TFAR) {
                        scratch.uv[index] = make_float2(u, v);
                        const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32)
                            | ((unsigned long long)(unsigned int)floatAsInt(t0.w) << 8) | (unsigned long long)index;
                        atomicMin(&scratch.best[owner], key);
                    }
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // the owners take what came back
    const unsigned long long key = scratch.best[lane];
    if (key != ~0ull) {
        const float t = __uint_as_float((unsigned int)(key >> 32));
        const int prim = (int)((key >> 8) & 0xFFFFFFull);
        const bool closer = (ray.bestPrim < 0) ? (t <= ray.best) : (t < ray.best || (t == ray.best && prim < ray.bestPrim));
        if (closer) {
            const float2 uv = scratch.uv[(int)(key & 255ull)];
            ray.best = t; ray.bestU = uv.x; ray.bestV = uv.y; ray.bestPrim = prim;
        }
    }
    if (scratch.occluded[lane] != 0u) { shadowRay.occluded = true; }
    return true;
}

template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_trace_small(RenderParams p, SmallTris smallTris)
{
    TraceGeometry geometry;
    geometry.nodes = nullptr;
    geometry.tris = p.scene.leafTris;
    geometry.nNodes = 0;
    geometry.nTris = p.scene.nTris;
    geometry.spheres = p.scene.spheres;
    geometry.nSpheres = p.scene.nLinearSpheres;

    const int lane = threadIdx.x & 63;
    const unsigned int waveId = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const unsigned int waveCount = gridDim.x * kWavesPerBlock;
    const unsigned int slotBatches = (unsigned int)p.nSlots / 64u;
    const unsigned int shadowCount = p.counters[kCtrShadowCount + (p.parity ^ 1) * kCursorStride];
    const unsigned int totalBatches = slotBatches + (shadowCount + 63u) / 64u;
    const int nTris = p.scene.nTris;

    unsigned int closestRays = 0, shadowRays = 0, trisTested = 0;

    for (unsigned int batch = waveId; batch < totalBatches; batch += waveCount) {
        LaneRay ray;
        bool valid = false;
        unsigned int target = 0;
        if (batch < slotBatches) {
            const unsigned int slot = batch * 64u + lane;
            const float4 rd = p.state.rayD[slot];
            if (!(floatAsInt(rd.w) & kStDone)) {
                const float4 ro = p.state.rayO[slot];
                laneRayInit(ray, v3(ro.x, ro.y, ro.z), v3(rd.x, rd.y, rd.z), PATHED_TNEAR, PATHED_TFAR, false);
                target = slot;
                valid = true;
            }
        } else {
            const unsigned int entry = (batch - slotBatches) * 64u + lane;
            if (entry < shadowCount) {
                const float4 so = p.state.shO[entry];
                const float4 sd = p.state.shD[entry];
                laneRayInit(ray, v3(so.x, so.y, so.z), v3(sd.x, sd.y, sd.z), PATHED_TNEAR, so.w, true);
                target = (unsigned int)floatAsInt(sd.w);
                valid = true;
            }
        }
        if (__ballot(valid) == 0ull) { continue; }

        // Phase 1 (wave-uniform, straight-line): every lane runs a CONSERVATIVE inside test of every
        // triangle on scalar-loaded records, two triangles per packed instruction, and only
        // RECORDS the candidates in a per-lane bitmask.  A line through the box pierces 4-6
        // triangles, so with 64 lanes "some lane is inside" is true for almost every triangle:
        // doing the division and the acceptance logic there made them two thirds of the
        // instructions.  The test is intersectTriangle()'s predicate with both det signs folded
        // by multiplying through with det:  u det >= 0, v det >= 0, (det - u - v) det >= 0,
        // t det >= 0, taken as "not (min < 0)" so that -0, underflow and NaN all err on the side of
        // keeping the candidate; phase 2 decides.  Bits are shifted in (cand = 2 cand + bit), so
        // triangle k of a 32-triangle word ends up at bit 31 - (k & 31).
        unsigned int candidatesLow = 0, candidatesHigh = 0;
        if (valid) {
            smallCandidates<true>(smallTris.data, nTris, ray.o, ray.d, &candidatesLow, &candidatesHigh,
                                  candidateNear(ray.tnear), ray.anyHit ? candidateFar(ray.tfar) : 3e38f);
            if (COUNT) { trisTested += (unsigned int)nTris; }
        }

        // Phase 2: the few candidates of each lane (typically 1-3) go through the ordinary
        // intersector + acceptance rule, so hits are those of the BVH path bit for bit.
        smallResolve(geometry, ray, candidatesLow, candidatesHigh);

        if (valid) {
            finishRay(geometry, ray);
            if (ray.anyHit) {
                if (ray.occluded) { p.state.pend[target] = make_float4(0.f, 0.f, 0.f, 0.f); }
                if (COUNT) { shadowRays++; }
            } else {
                p.state.hit[target] = make_float4(ray.best, ray.bestU, ray.bestV, intAsFloat(ray.bestPrim));
                if (COUNT) { closestRays++; }
            }
        }
    }

    if (COUNT) {
        atomicAdd(&p.stats[kStatTris], (unsigned long long)trisTested);
        atomicAdd(&p.stats[kStatClosest], (unsigned long long)closestRays);
        atomicAdd(&p.stats[kStatShadow], (unsigned long long)shadowRays);
    }
}

// test hook kernel behind pathed_hip_trace: plain grid, arbitrary ray intervals (FORMAT: as k_trace's)
template <int STACK, int FORMAT>
__global__ __launch_bounds__(kBlock) void k_trace_rays(
    DScene scene, const float4 *rays, int n, int anyHit, float4 *hitsOut, int *occludedOut,
    int *stackOverflow, int maxStack
) {
    extern __shared__ float4 ldsRaw[];
    LaneStack stack;
    stack.lds = reinterpret_cast<int *>(ldsRaw) + threadIdx.x;
    stack.overflowStride = (size_t)gridDim.x * kBlock;
    stack.overflow = stackOverflow + ((size_t)blockIdx.x * kBlock + threadIdx.x);

    TraceGeometry geometry;
    geometry.nodes = FORMAT != 0 ? scene.nodesQ : scene.nodes;
    geometry.tris = scene.leafTris;
    geometry.nNodes = scene.nNodes;
    geometry.nTris = scene.nTris;
    geometry.spheres = scene.spheres;
    geometry.nSpheres = scene.nLinearSpheres;

    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) { return; }
    const float4 ro = rays[2 * i + 0];
    const float4 rd = rays[2 * i + 1];
    RayHit hit;
    hit.t = 0.f; hit.u = 0.f; hit.v = 0.f; hit.prim = -1;
    TraceCounters counters;
    if (anyHit) {
        const bool occluded = traverse<false, STACK, kBlock, FORMAT>(
            geometry, stack, maxStack, v3(ro.x, ro.y, ro.z), v3(rd.x, rd.y, rd.z), ro.w, rd.w, true, &hit, &counters);
        occludedOut[i] = occluded ? 1 : 0;
    } else {
        const bool found = traverse<false, STACK, kBlock, FORMAT>(
            geometry, stack, maxStack, v3(ro.x, ro.y, ro.z), v3(rd.x, rd.y, rd.z), ro.w, rd.w, false, &hit, &counters);
        if (!found) { hit.t = 0.f; hit.u = 0.f; hit.v = 0.f; hit.prim = -1; }
        hitsOut[i] = make_float4(hit.t, hit.u, hit.v, intAsFloat(hit.prim));
    }
}

// ------------------------------------------------------------------------- shade

struct TriShade {
    V3 p0, p1, p2;
    int material;
};

__device__ inline TriShade loadTriCorners(const DScene &scene, int prim)
{
    const float4 *q = scene.triShade + (size_t)kTriShadeQuads * prim;
    const float4 q0 = q[0], q1 = q[1], q2 = q[2];
    TriShade tri;
    tri.p0 = v3(q0.x, q0.y, q0.z);
    tri.p1 = v3(q1.x, q1.y, q1.z);
    tri.p2 = v3(q2.x, q2.y, q2.z);
    tri.material = floatAsInt(q0.w);
    return tri;
}

// Ng = (v1-v0) x (v2-v0), normalised: ONE definition, used where a hit is shaded and where the 16-byte records are built
__device__ inline V3 triangleNormal(V3 p0, V3 p1, V3 p2) { return normalized(xcross(p1 - p0, p2 - p0)); }

// Scene::testIntersect's record construction, reference src/scene.cpp:121-218
template <typename TRAITS = TraitsAll>
__device__ inline Isect makeIsect(const DScene &scene, V3 o, V3 d, float4 h)
{
    const float t = h.x, u = h.y, v = h.z;
#if defined(PATHED_ABLATE) && PATHED_ABLATE == 1
    const int prim = floatAsInt(h.w) & 1023;   // profiling build: shading records from a cache-resident corner of the table
#else
    const int prim = floatAsInt(h.w);
#endif

    V3 geometricNormal;
    V3 shadingNormal = v3(0.f, 0.f, 0.f);
    float uvU = 0.f, uvV = 0.f;
    int material;

    if (!TRAITS::spheres || prim < scene.nTris) {
        // A triangle without vertex normals and uvs (all exactly zero: the interpolated shading normal has length 0 and
        // the geometric normal takes its place, uv = 0) is shaded from a 16-byte record.  Which primitives are plain is a
        // few id ranges in the kernel arguments (scalar compares), so either load is issued at once.
        bool plain = false;
        for (int r = 0; r < scene.nPlainRanges; r++) { plain = plain || (prim >= scene.plainBegin[r] && prim < scene.plainEnd[r]); }
        if (plain) {
            const float4 compact = scene.triCompact[prim];
            material = floatAsInt(compact.w);
            geometricNormal = v3(compact.x, compact.y, compact.z);
        } else {
            const float4 *q = scene.triShade + (size_t)kTriShadeQuads * prim;
            const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6];
            const V3 p0 = v3(q0.x, q0.y, q0.z), p1 = v3(q1.x, q1.y, q1.z), p2 = v3(q2.x, q2.y, q2.z);
            const float w = 1.f - u - v;
            // rtcInterpolate0 with weights (1-u-v, u, v): uv slot 0, normal slot 1
            uvU = fmaf(w, q1.w, fmaf(u, q3.w, v * q5.w));
            uvV = fmaf(w, q2.w, fmaf(u, q4.w, v * q6.x));
            shadingNormal = v3(
                fmaf(w, q3.x, fmaf(u, q4.x, v * q5.x)),
                fmaf(w, q3.y, fmaf(u, q4.y, v * q5.y)),
                fmaf(w, q3.z, fmaf(u, q4.z, v * q5.z)));
            geometricNormal = triangleNormal(p0, p1, p2);
            material = floatAsInt(q0.w);
        }
    } else {
        const DSphere sphere = scene.spheres[prim - scene.nTris];
        const V3 point = o + d * t;
        geometricNormal = normalized(point - v3(sphere.centerWorld[0], sphere.centerWorld[1], sphere.centerWorld[2]));
        material = sphere.material;
    }

    if (length(shadingNormal) == 0.f) { shadingNormal = geometricNormal; }

    Isect isect;
    isect.point = o + d * t;  // Ray::at, src/ray.cpp:9-12
    isect.wo = -d;
    isect.normal = geometricNormal;
    isect.shadingNormal = normalized(shadingNormal);
    isect.u = uvU;
    isect.v = uvV;
    isect.material = material;
    isect.prim = prim;
    isect.frame = normalToWorldSpace(isect.shadingNormal, isect.wo);
    isect.woLocal = v3(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));   // prepareLobes (shading.h)
    return isect;
}

// Scene::lightsPDF, src/scene.cpp:469-484
template <typename TRAITS = TraitsAll>
__device__ inline float lightsPDF(const DScene &scene, V3 referencePoint, const Isect &lightIsect)
{
    float measurePDF;
    if (!TRAITS::spheres || lightIsect.prim < scene.nTris) {
        const TriShade tri = loadTriCorners(scene, lightIsect.prim);
        measurePDF = trianglePdfSolidAngle(tri.p0, tri.p1, tri.p2, lightIsect.point, referencePoint);
    } else {
        const DSphere sphere = scene.spheres[lightIsect.prim - scene.nTris];
        measurePDF = spherePdfSolidAngle(
            v3(sphere.centerSample[0], sphere.centerSample[1], sphere.centerSample[2]), sphere.radius, referencePoint);
    }
    return measurePDF / scene.nLights;
}

// Scene::environmentL, src/scene.cpp:486-492
template <typename TRAITS = TraitsAll>
__device__ inline Rgb environmentL(const DScene &scene, V3 direction)
{
    if (!TRAITS::env) { return rgb(0.f); }
#if defined(PATHED_ABLATE) && PATHED_ABLATE == 3
    return rgb(0.f);   // profiling build: no environment lookups on misses
#endif
    if (scene.hasEnv) { return envEmit(scene.env, -direction); }
    return rgb(0.f);
}

struct ShadowRequest {
    bool push;
    V3 origin, direction;
    float tfar;
};

// PathTracer::directSampleLights, src/path_tracer.cpp:113-165, up to (not including) the
// occlusion query: returns the contribution assuming visibility and the shadow ray to test.
// ENV_ONLY: the scene's one light is the environment and no material emits (the env-lit mesh scenes): the light choice is
// index 0 (its random number is still consumed), the triangle / sphere sampling code and the emitter look-ups are compiled out.
// With one light every `* lightCount`, `/ nLights` is a multiplication or division by 1: the same floats.
template <bool ENV_ONLY = false, typename TRAITS = TraitsAll, typename MaterialTable>
__device__ inline Rgb sampleLightsTerm(
    const DScene &scene, const MaterialTable &materials,
    const Isect &isect, const DMaterial &material, Rng &random, ShadowRequest *shadow
) {
    shadow->push = false;
#if defined(PATHED_ABLATE) && PATHED_ABLATE == 2
    return rgb(0.f);   // profiling build: no light sampling, no shadow rays
#endif
    if (isDeltaT<TRAITS>(material)) { return rgb(0.f); }
    if (scene.nLights == 0) { return rgb(0.f); }

    // Scene::sampleDirectLights, src/scene.cpp:446-467
    const int lightCount = ENV_ONLY ? 1 : scene.nLights;
    DLight light;
    light.kind = 2;
    light.index = 0;
    if (ENV_ONLY) {
        random.dimension++;   // the light choice: floor(u * 1) = 0
    } else {
        int lightIndex = (int)floorf(random.next() * lightCount);
        lightIndex = imin(lightIndex, lightCount - 1);
        light = scene.lights[lightIndex];
    }

    SurfaceSample surfaceSample;
    int lightMaterial = 0;
    if (ENV_ONLY) {
        surfaceSample = envSample<TRAITS::pairedTrig>(scene.env, isect.point, random);
    } else if (TRAITS::triangleLights && (light.kind == 0 || (!TRAITS::spheres && !TRAITS::env))) {
        const TriShade tri = loadTriCorners(scene, light.index);
        surfaceSample = triangleSample(tri.p0, tri.p1, tri.p2, random);
        lightMaterial = tri.material;
    } else if (TRAITS::spheres && (light.kind == 1 || !TRAITS::env)) {
        const DSphere sphere = scene.spheres[light.index];
        surfaceSample = sphereSample<TRAITS::pairedTrig>(
            v3(sphere.centerSample[0], sphere.centerSample[1], sphere.centerSample[2]), sphere.radius, isect.point, random);
        lightMaterial = sphere.material;
    } else if (TRAITS::env) {
        surfaceSample = envSample<TRAITS::pairedTrig>(scene.env, isect.point, random);
    } else {   // not reached: a light of a kind the instantiation's scene set does not contain
        surfaceSample.point = isect.point;
        surfaceSample.normal = v3(0.f, 0.f, 0.f);
        surfaceSample.invPDF = 1.f;
        surfaceSample.solidAngle = 1;
    }
    const float lightChoicePDF = 1.f / lightCount;
    const float invPDF = surfaceSample.invPDF * (1.f / lightChoicePDF);

    const V3 lightDirection = surfaceSample.point - isect.point;
    const V3 wiWorld = normalized(lightDirection);

    if (dot(surfaceSample.normal, wiWorld) >= 0.f) { return rgb(0.f); }  // back of the light

    const float lightDistance = length(lightDirection);

    // LightSample::solidAnglePDF, include/scene.h:66-80
    float pdf;
    if (surfaceSample.solidAngle) {
        pdf = 1.f / invPDF;
    } else {
        const V3 lightWoForPdf = -normalized(lightDirection);
        const float distance2 = lightDistance * lightDistance;
        const float projectedArea = smax(0.f, dot(surfaceSample.normal, lightWoForPdf));
        pdf = (1.f / invPDF) * distance2 / projectedArea;
    }

    float brdfPDF;
    const Rgb f = materialF<TRAITS>(material, isect, wiWorld, &brdfPDF);
    const float lightWeight = (1 * pdf) / (1 * pdf + 1 * brdfPDF);  // include/mis.h:4-7

    // The reference asks for the occlusion first and evaluates emitted * weight * f * cos / pdf afterwards.
    // With f exactly black (the light is below the surface's horizon) and finite, positive pdfs the product is
    // black whatever the occlusion query says, so neither the emission lookup nor the shadow ray is needed.
    // (Only a non-finite emission could tell the difference; the oracle takes the same shortcut.  Testing the
    // finished product instead costs k_shade one more VGPR and with it the second wave next to the trace waves.)
    if (isBlack(f) && pdf > 0.f && pdf < 3e38f && brdfPDF >= 0.f && brdfPDF < 3e38f) { return rgb(0.f); }

    const V3 lightWo = -normalized(lightDirection);
    Rgb emitted;
    if (ENV_ONLY || (TRAITS::env && light.kind == 2)) { emitted = envEmit(scene.env, lightWo); }
    else { emitted = matEmit(materials[lightMaterial]); }

    // Scene::testOcclusion's interval, src/scene.cpp:366-367
    shadow->push = true;
    shadow->origin = isect.point;
    shadow->direction = wiWorld;
    shadow->tfar = lightDistance - 1e-3f;

    const Rgb contribution = emitted
        * lightWeight
        * f
        * fabsf(dot(isect.shadingNormal, wiWorld))
        / pdf;
    return contribution;
}

// Camera::generateRay(int,int), src/camera.cpp:49-55, for (pixel, sample)
__device__ inline void startSample(const RenderParams &p, uint32_t pixel, uint32_t sample, int sampleInUnit, float4 *rayO, float4 *rayD)
{
    const int width = p.scene.camera.resX;
    const int row = (int)fastDivide((unsigned int)pixel, p.divWidth);   // pixel / width, exactly
    const int col = (int)pixel - row * width;
    Rng random;
    makeKey(((uint64_t)p.seedHi << 32) | p.seedLo, pixel, sample, &random.k0, &random.k1);
    random.dimension = 0;
    const float jitterX = random.next() - 0.5f;
    const float jitterY = random.next() - 0.5f;
    V3 origin, direction;
    cameraRay(p.scene.camera, row + jitterY, col + jitterX, &origin, &direction);
    *rayO = make_float4(origin.x, origin.y, origin.z, intAsFloat(-1));
    *rayD = make_float4(direction.x, direction.y, direction.z, intAsFloat(sampleInUnit << kStSampleShift));
}

// Block-aggregated grab of work units: returns this lane's unit, or 0xFFFFFFFF when every queue is dealt out.
// One atomic per block and queue tried.  A block starts at its home queue (blockIdx mod queues: the slots of a block then
// work on one patch of the image, see THE UNIT ORDER) and moves on through the others once that is empty, so no unit
// is left behind by a block whose own slots have all retired, and no slot idles while any queue still holds work.
__device__ inline unsigned int grabUnits(const RenderParams &p, bool want, unsigned int *ldsScratch)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const unsigned int queues = (unsigned int)p.nQueues;
    const unsigned int home = blockIdx.x % queues;
    unsigned int mine = 0xFFFFFFFFu;
    for (unsigned int attempt = 0; attempt < queues; attempt++) {
        const unsigned long long mask = __ballot(want);
        const unsigned int before = (unsigned int)__popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) { ldsScratch[wave] = (unsigned int)__popcll(mask); }
        __syncthreads();
        unsigned int offset = 0, total = 0;
        #pragma unroll
        for (int w = 0; w < kWavesPerBlock; w++) {
            const unsigned int count = ldsScratch[w];
            if (w < wave) { offset += count; }
            total += count;
        }
        if (total == 0u) { __syncthreads(); break; }   // nobody (left) wants a unit: block-uniform
        // (block-uniform; readfirstlane says so to the compiler, which otherwise may index the kernel arguments' queueUnits with
        // a vector register and, to do that, keep a private copy of all of RenderParams: k_shade_env did, 1.6 KB per lane)
        const unsigned int queue = (unsigned int)__builtin_amdgcn_readfirstlane((int)(home + attempt < queues ? home + attempt : home + attempt - queues));
        const unsigned int limit = p.queueUnits[queue];
        if (threadIdx.x == 0) {
            unsigned int *cursor = &p.counters[kCtrUnitCursor + queue * kCursorStride];
            // a queue that is already dealt out costs a load, not an atomic (the drain of a pass asks every queue)
            const unsigned int seen = __atomic_load_n(cursor, __ATOMIC_RELAXED);
            ldsScratch[kWavesPerBlock] = seen < limit ? atomicAdd(cursor, total) : 0xFFFFFFFFu;
        }
        __syncthreads();
        const unsigned int base = ldsScratch[kWavesPerBlock];
        __syncthreads();  // scratch is reused by the next attempt and by the caller
        if (base >= limit) { continue; }
        if (want) {
            const unsigned int k = base + offset + before;
            if (k < limit) { mine = queue * p.unitsPerQueue + k; want = false; }
        }
        if (base + total <= limit) { break; }   // everybody was served
    }
    return mine;
}

// THE UNIT ORDER.  A unit is (pixel, chunk of the pass's samples); which slot renders it, and when, cannot change a
// result (every unit's partial sum has its own place and k_resolve adds them in chunk order), so the order is free to
// serve the memory system and the balance of the queues.  Unit id = queue * unitsPerQueue + position in the queue.
//   kOrderStripes (default): queue q of pool h owns the CHUNKS (q * pools + h) + j * (queues * pools), j = 0, 1, ..,
//     each a whole image walked in pixel order: every queue starts every chunk at the first pixel, so all queues sweep
//     the image in step, a band of it in flight at any time, whatever the number of chunks (the former contiguous
//     split of a chunk-major numbering did that only when the chunk count was a multiple of the queue count: a pass
//     of 517 chunks per pixel ran 10 % slower than one of 512).  Queues are equal up to one chunk; grabUnits moves a
//     block on when its queue is dealt out.
//   kOrderStripesTiled: the same with the pixels of a chunk walked in 32 x 8 tiles (below) instead of rows.
//   kOrderTiles: pixel GROUPS of 256 (tiles) dealt round-robin to the queues, a queue hands out all chunks of a group
//     before it moves on: the slots of a block all work on one patch of the image.  Measured 6-8 % SLOWER on the mesh
//     scenes than the stripes (tools/chunk_sweep2.py): kept selectable (PATHED_UNIT_ORDER).
// Tile walk: bands of 8 rows, blocks of 32 columns inside a band, row-major inside a block; ragged last band / block.
static const int kOrderStripes = 0, kOrderStripesTiled = 1, kOrderTiles = 2;

__device__ inline unsigned int tileWalkPixel(const RenderParams &p, unsigned int index)
{
    const unsigned int width = (unsigned int)p.scene.camera.resX, height = (unsigned int)p.scene.camera.resY;
    const unsigned int band = fastDivide(index, p.divBand);
    const unsigned int inBand = index - band * 8u * width;
    const unsigned int rowsLeft = height - 8u * band;
    const unsigned int rows = rowsLeft < 8u ? rowsLeft : 8u;
    const unsigned int block = rows == 8u ? inBand >> 8 : inBand / (32u * rows);
    const unsigned int inBlock = inBand - block * 32u * rows;
    const unsigned int columnsLeft = width - 32u * block;
    const unsigned int columns = columnsLeft < 32u ? columnsLeft : 32u;
    const unsigned int y = columns == 32u ? inBlock >> 5 : inBlock / columns;
    const unsigned int x = inBlock - y * columns;
    return (8u * band + y) * width + 32u * block + x;
}

__device__ inline void unitPlace(const RenderParams &p, unsigned int unit, uint32_t *pixel, uint32_t *chunkIndex)
{
    const unsigned int queue = fastDivide(unit, p.divStride);
    const unsigned int k = unit - queue * p.unitsPerQueue;
    const unsigned int j = fastDivide(k, p.divGroupUnits);
    const unsigned int w = k - j * p.groupUnits;
    const unsigned int owned = queue * p.pools + p.pool + j * (unsigned int)p.nQueues * p.pools;   // j-th chunk / group of the queue
    if (p.unitOrder != kOrderTiles) {
        *chunkIndex = owned;                                   // groupUnits = pixels of the image
        *pixel = p.unitOrder == kOrderStripes ? w : tileWalkPixel(p, w);
        return;
    }
    unsigned int chunk, within;
    if (owned + 1u < p.nGroups || p.lastGroupPixels == kUnitGroup) {
        chunk = w >> 8;
        within = w & (kUnitGroup - 1u);
    } else {   // the ragged last group
        chunk = w / p.lastGroupPixels;
        within = w - chunk * p.lastGroupPixels;
    }
    *pixel = tileWalkPixel(p, owned * kUnitGroup + within);
    *chunkIndex = chunk;
}

// first sample index of a unit and one-past-last
__device__ inline void unitSamples(const RenderParams &p, unsigned int unit, uint32_t *pixel, uint32_t *first, uint32_t *end)
{
    uint32_t chunkIndex;
    unitPlace(p, unit, pixel, &chunkIndex);
    *first = p.sppBegin + chunkIndex * (unsigned int)p.chunk;
    const uint32_t last = *first + (unsigned int)p.chunk;
    *end = last < p.sppEnd ? last : p.sppEnd;
}

// where a unit's partial sum goes: chunk-major, so that k_resolve reads a chunk's pixels coalesced
__device__ inline size_t partialIndex(const RenderParams &p, unsigned int unit)
{
    uint32_t pixel, chunkIndex;
    unitPlace(p, unit, &pixel, &chunkIndex);
    return (size_t)chunkIndex * (size_t)p.nPixels + pixel;
}

__global__ __launch_bounds__(kBlock) void k_init(RenderParams p)
{
    __shared__ unsigned int scratch[kWavesPerBlock + 1];
    const int slot = blockIdx.x * kBlock + threadIdx.x;
    const unsigned int unit = grabUnits(p, true, scratch);

    float4 rayO = make_float4(0.f, 0.f, 0.f, intAsFloat(-1));
    float4 rayD = make_float4(0.f, 0.f, 0.f, intAsFloat(kStDone));
    const bool started = unit != 0xFFFFFFFFu;
    if (started) {
        uint32_t pixel, first, end;
        unitSamples(p, unit, &pixel, &first, &end);
        startSample(p, pixel, first, 0, &rayO, &rayD);
    }
    p.state.rayO[slot] = rayO;
    p.state.rayD[slot] = rayD;
    p.state.res[slot] = make_float4(0.f, 0.f, 0.f, intAsFloat((int)unit));
    p.state.mod[slot] = make_float4(1.f, 1.f, 1.f, 1.f);
    p.state.thr[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    p.state.pend[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    p.state.acc[slot] = make_float4(0.f, 0.f, 0.f, 0.f);

    const unsigned long long mask = __ballot(started);
    if ((threadIdx.x & 63) == 0 && mask != 0ull) {
        atomicAdd(&p.counters[kCtrRemaining], (unsigned int)__popcll(mask));
    }
}

// adds a pixel's unit partial sums to the radiance sum, in chunk order
__global__ __launch_bounds__(kBlock) void k_resolve(RenderParams p)
{
    const int pixel = blockIdx.x * kBlock + threadIdx.x;
    if (pixel >= p.nPixels) { return; }
    float *out = p.accum + 3 * (size_t)pixel;
    float r = out[0], g = out[1], b = out[2];
    // the additions are a dependent chain, the loads are not: eight in flight per lane
    int chunk = 0;
    for (; chunk + 8 <= p.chunksPerPixel; chunk += 8) {
        float4 partial[8];
        #pragma unroll
        for (int k = 0; k < 8; k++) { partial[k] = p.state.chunkBuf[(size_t)(chunk + k) * p.nPixels + pixel]; }
        #pragma unroll
        for (int k = 0; k < 8; k++) {
            r += partial[k].x;
            g += partial[k].y;
            b += partial[k].z;
        }
    }
    for (; chunk < p.chunksPerPixel; chunk++) {
        const float4 partial = p.state.chunkBuf[(size_t)chunk * p.nPixels + pixel];
        r += partial.x;
        g += partial.y;
        b += partial.z;
    }
    out[0] = r;
    out[1] = g;
    out[2] = b;
}

// Tuning builds only (-DPATHED_SHADE_PROFILE): how many waves enter a region of k_shade and with how
// many lanes -- lane utilisation per region (tools/shade_profile.py prints the table).
#ifdef PATHED_SHADE_PROFILE
#define SHADE_REGION(index, predicate)                                                                        \
    do {                                                                                                      \
        const unsigned long long mask_ = __ballot(predicate);                                                 \
        if (mask_ != 0ull && (threadIdx.x & 63) == __ffsll((long long)mask_) - 1) {                           \
            atomicAdd(&p.stats[kStatShadeProfile + 2 * (index)], 1ull);                                       \
            atomicAdd(&p.stats[kStatShadeProfile + 2 * (index) + 1], (unsigned long long)__popcll(mask_));    \
        }                                                                                                     \
    } while (0)
// ... and for a resolve loop (smallResolve): its iterations (the wave's largest candidate count) and the candidates it tests
#define RESOLVE_PROBE(index, low, high)                                                                       \
    do {                                                                                                      \
        int most_ = __popc(low) + __popc(high), all_ = most_;                                                 \
        for (int step_ = 32; step_ >= 1; step_ >>= 1) {                                                       \
            const int other_ = __shfl_xor(most_, step_, 64);                                                  \
            most_ = other_ > most_ ? other_ : most_;                                                          \
            all_ += __shfl_xor(all_, step_, 64);                                                              \
        }                                                                                                     \
        if ((threadIdx.x & 63) == 0) {                                                                        \
            atomicAdd(&p.stats[kStatShadeProfile + 2 * (index)], (unsigned long long)most_);                  \
            atomicAdd(&p.stats[kStatShadeProfile + 2 * (index) + 1], (unsigned long long)all_);               \
        }                                                                                                     \
    } while (0)
#else
#define SHADE_REGION(index, predicate) do { } while (0)
#define RESOLVE_PROBE(index, low, high) do { } while (0)
#endif

template <bool LDS_MATERIALS>
struct MaterialAccess {
    const DMaterial *table;
    __device__ inline const DMaterial &operator[](int index) const { return table[index]; }
};

#ifndef PATHED_SHADE_TRIM
// 1: a slot that regenerates skips the reset of mod / thr / pend (its camera ray's vertex reads none of them) and, with one
// sample per unit, the `acc` stream: 80 bytes less per regenerated slot -- and SLOWER (teapot 1 731 against 1 762, the mesh
// filling the frame 1 362 against 1 380 Msamples/s, same box, profiles/r3_ab_shade_trim.log): a store that half the lanes
// of a wave skip writes partial cache lines, which cost more than the full 1 KiB lines they replace.
// 2 [r5]: ONLY the `acc` stream goes (one sample per unit: the unit's partial sum is 0 + colour, nobody needs the running sum):
// a stream that only finished lanes touched at all -- 32 of the 240 state bytes per vertex, a line of it fetched and written
// back whenever one of its eight slots finished.  +1.0 % on the teapot, the 5.2 M-triangle mesh and its close-up, same bits
// (profiles/r5_ab_acc_trim.log): the pipeline is not bound by its state BYTES.
#define PATHED_SHADE_TRIM 2
#endif
#ifdef PATHED_SHADE_WAVES   // experiments: cap k_shade's registers for this many waves per SIMD
#define PATHED_SHADE_ATTRIBUTE __attribute__((amdgpu_waves_per_eu(PATHED_SHADE_WAVES, PATHED_SHADE_WAVES)))
#else
#define PATHED_SHADE_ATTRIBUTE
#endif
template <bool LDS_MATERIALS, bool ENV_ONLY, typename TRAITS = TraitsAll>
__global__ __launch_bounds__(kBlock) PATHED_SHADE_ATTRIBUTE void k_shade(RenderParams p)
{
    __shared__ DMaterial ldsMaterials[LDS_MATERIALS ? kMaxLdsMaterials : 1];
    __shared__ unsigned int scratch[kWavesPerBlock + 1];

    MaterialAccess<LDS_MATERIALS> materials;
    if (LDS_MATERIALS) {
        // material parameters staged in LDS: every lane indexes them by its own hit
        const int words = p.scene.nMaterials * (int)(sizeof(DMaterial) / 4);
        const int *source = reinterpret_cast<const int *>(p.scene.materials);
        int *target = reinterpret_cast<int *>(ldsMaterials);
        for (int i = threadIdx.x; i < words; i += kBlock) { target[i] = source[i]; }
        __syncthreads();
        materials.table = ldsMaterials;
    } else {
        materials.table = p.scene.materials;
    }

    const int slot = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const DScene &scene = p.scene;
    // ENV_ONLY scenes have no emissive material: every emission look-up below is black at compile time
    auto emission = [&](int material) -> Rgb { return ENV_ONLY ? rgb(0.f) : matEmit(materials[material]); };

    // rewind the card cursors for the pool's next trace launch (same stream: it starts after us)
    if (blockIdx.x == 0 && threadIdx.x < kTraceShards) { p.counters[kCtrTraceCursor + threadIdx.x * kCursorStride] = 0u; }
    // ... and empty the shadow list the NEXT k_shade will fill (the trace launch that read it is over)
    if (blockIdx.x == 0 && threadIdx.x == kTraceShards) { p.counters[kCtrShadowCount + (p.parity ^ 1) * kCursorStride] = 0u; }

    // state word and hit record travel together: the shading-record gather that depends on the
    // hit then starts one HBM round trip earlier
    float4 rd = p.state.rayD[slot];
    float4 h = p.state.hit[slot];
    const float4 roIn = p.state.rayO[slot];
    const float4 resIn0 = p.state.res[slot];
    pinLoaded(rd);
    pinLoaded(h);
    pinLoaded(roIn);
    pinLoaded(resIn0);
    int st = floatAsInt(rd.w);
    bool active = !(st & kStDone);
    float4 pendIn = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        // a slot with a parked ray (see kSuspendLanes) sits this iteration out, untouched
        if (st & kStEligible) { pendIn = p.state.pend[slot]; }
        if (p.suspendLanes > 0) {
            const bool parked = floatAsInt(h.w) == kPrimSuspended
                || ((st & kStEligible) && floatAsInt(pendIn.w) == kShadowSuspended);
            if (parked) {
                if (!(st & kStHold)) { reinterpret_cast<int *>(p.state.rayD + slot)[3] = st | kStHold; }
                active = false;
            }
            st &= ~kStHold;
            rd.w = intAsFloat(st);
        }
    }

    // [r5] several shade launches may follow one trace launch (local rays): in the later ones a slot that waits for the
    // trace kernel sits the launch out, untouched
    if (!p.afterTrace && (st & kStAwait)) { active = false; }
    st &= ~(kStLocal | kStAwait);  // (the hit of a local slot is this kernel's own, written by the launch before)
    rd.w = intAsFloat(st);

    ShadowRequest shadow;
    shadow.push = false;
    shadow.origin = v3(0.f, 0.f, 0.f);
    shadow.direction = v3(0.f, 0.f, 0.f);
    shadow.tfar = 0.f;

    // what the convergent tail needs
    bool finished = false;        // the sample in flight ended this iteration
    Rgb color = rgb(0.f);
    unsigned int unit = 0xFFFFFFFFu;
    int sampleInUnit = 0;
    float4 outRayO = make_float4(0.f, 0.f, 0.f, 0.f), outRayD = rd;
    float4 outMod = make_float4(1.f, 1.f, 1.f, 1.f);
    float4 outThr = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 outPend = make_float4(0.f, 0.f, 0.f, 0.f);
    Rgb result = rgb(0.f);

    SHADE_REGION(0, true);        // every wave, 64 lanes
    SHADE_REGION(1, active);      // slots with work
    if (active) {
        const float4 ro = roIn;
        const float4 resIn = resIn0;
        outRayO = ro;

        const V3 o = v3(ro.x, ro.y, ro.z);
        const V3 d = v3(rd.x, rd.y, rd.z);
        const int rayBounce = st & kStBounceMask;  // vertex that spawned this ray, 0 = camera
        sampleInUnit = (st >> kStSampleShift) & kStSampleMask;
        const bool miss = floatAsInt(h.w) < 0;
        unit = (unsigned int)floatAsInt(resIn.w);
        int firstEmitMaterial = floatAsInt(ro.w);

        uint32_t pixel, firstSample, endSample;
        unitSamples(p, unit, &pixel, &firstSample, &endSample);
        const uint32_t sample = firstSample + (uint32_t)sampleInUnit;

        result = rgb(resIn.x, resIn.y, resIn.z);
        Rgb modulation = rgb(1.f);

        bool haveVertex = false;  // a new surface vertex to process this iteration
        Isect isect;
        const int vertex = rayBounce + 1;
        // ONE inlined copy of the intersection record, ahead of the bounce-0 / later-bounce split:
        // the lanes of a wave sit at different path depths, and code inlined in both branches is
        // executed twice by every mixed wave
        SHADE_REGION(2, !miss);   // makeIsect
        if (!miss) { isect = makeIsect<TRAITS>(scene, o, d, h); }
        SHADE_REGION(3, rayBounce == 0);
        SHADE_REGION(4, rayBounce != 0 && (st & kStEligible) != 0);   // finishes the previous vertex's MIS term

        if (rayBounce == 0) {
            // SampleIntegrator::samplePixel, src/sample_integrator.cpp:18-59
            if (miss) {
                color = rgb(0.f) + environmentL<TRAITS>(scene, d);
                finished = true;
            } else {
                firstEmitMaterial = -1;
                if (checkCounts(p.startBounce, p.lastBounce, 0)) {
                    const Rgb emit = emission(isect.material);
                    const bool backside = dot(isect.normal, isect.wo) < 0.f;
                    if (!isBlack(emit) && !backside) { firstEmitMaterial = isect.material; }
                }
                result = rgb(0.f);
                haveVertex = true;
            }
        } else {
            // the ray left vertex `rayBounce` along its BSDF sample
            const float4 modIn = p.state.mod[slot];
            const float4 thrIn = p.state.thr[slot];
            modulation = rgb(modIn.x, modIn.y, modIn.z);
            const float bsdfPdf = modIn.w;
            const Rgb throughput = rgb(thrIn.x, thrIn.y, thrIn.z);
            const float cosTheta = thrIn.w;

            if (st & kStEligible) {
                // PathTracer::directSampleBSDF, src/path_tracer.cpp:167-216
                Rgb bsdfTerm = rgb(0.f);
                if (!miss) {
                    const Rgb emit = emission(isect.material);
                    if (!isBlack(emit) && dot(isect.wo, isect.shadingNormal) >= 0.f) {
                        const float lightPDF = lightsPDF<TRAITS>(scene, o, isect);
                        const float brdfWeight = (st & kStDelta)
                            ? 1.f
                            : (1 * bsdfPdf) / (1 * bsdfPdf + 1 * lightPDF);
                        bsdfTerm = emit * brdfWeight * throughput * cosTheta / bsdfPdf;
                    }
                } else {
                    const Rgb environmentLight = environmentL<TRAITS>(scene, d);
                    if (!isBlack(environmentLight)) {
                        // Scene::environmentPDF, src/scene.cpp:494-502
                        const float lightPDF = envEmitPDF(scene.env, d) / (ENV_ONLY ? 1 : scene.nLights);
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

            // PathTracer::L loop body, src/path_tracer.cpp:41-58
            if (!(st & kStContinue) || miss) {
                finished = true;
            } else {
                const float invPDF = 1.f / bsdfPdf;
                modulation = modulation * (throughput * cosTheta * invPDF);
                if (isBlack(modulation)) { finished = true; }
                else { haveVertex = true; }
            }
            if (finished) {
                Rgb first = rgb(0.f);
                if (firstEmitMaterial >= 0) { first = first + emission(firstEmitMaterial); }
                color = first + result;
            }
        }

        outMod = make_float4(modulation.r, modulation.g, modulation.b, 1.f);

        SHADE_REGION(5, haveVertex);
        if (haveVertex) {
            // PathTracer::L: sample the BSDF, then direct(), src/path_tracer.cpp:30-36, 60-73
            const DMaterial &material = materials[isect.material];
            prepareLobes<TRAITS>(material, isect);

            Rng random;
            makeKey(((uint64_t)p.seedHi << 32) | p.seedLo, pixel, sample, &random.k0, &random.k1);
            random.dimension = vertexBase(vertex);
            const BSDFSample bsdfSample = materialSample<TRAITS>(material, isect, random);

            const bool counts = checkCounts(p.startBounce, p.lastBounce, vertex);
            const bool emissive = !ENV_ONLY && !isBlack(matEmit(material));
            const bool wantDirect = counts && !emissive;  // direct() returns 0 on emitters (:86-90)
            const bool wantContinue = !checkDone(p.lastBounce, vertex + 1);

            Rgb lightTerm = rgb(0.f);
            SHADE_REGION(6, wantDirect);
            SHADE_REGION(10, isBlack(bsdfSample.throughput));   // the continuation ray cannot contribute
            if (wantDirect) {
                random.dimension = vertexBase(vertex) + 3;
                lightTerm = sampleLightsTerm<ENV_ONLY, TRAITS>(scene, materials, isect, material, random, &shadow);
            }

            // A vertex with nothing pending (no light term, no shadow ray) whose BSDF sample has exactly black
            // throughput ends the sample here: the reference would trace the continuation ray, multiply the
            // modulation by that black throughput and stop, with the same result (paths that land on a
            // black-bodied emitter: 10 % of the vertices of the Veach scene).
            const bool deadEnd = isBlack(bsdfSample.throughput) && bsdfSample.pdf > 0.f && bsdfSample.pdf < 3e38f
                && !shadow.push && isBlack(lightTerm);
            if ((!wantDirect && !wantContinue) || deadEnd) {
                finished = true;
                Rgb first = rgb(0.f);
                if (firstEmitMaterial >= 0) { first = first + emission(firstEmitMaterial); }
                color = first + result;
                shadow.push = false;
            } else {
                int nextState = vertex | (sampleInUnit << kStSampleShift);
                if (wantDirect) { nextState |= kStEligible; }
                if (isDeltaT<TRAITS>(material)) { nextState |= kStDelta; }
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
    }

    // ---- sample / unit bookkeeping --------------------------------------------------------
    bool needUnit = false;
    bool startNext = false;       // the slot starts a camera ray: one inlined copy of startSample below
    uint32_t nextPixel = 0, nextSample = 0;
    SHADE_REGION(7, active && finished);
    if (active && finished) {
        // radianceLookup += color, src/sample_integrator.cpp:61-63; non-finite samples dropped.  With one sample per
        // unit (the default) the unit's partial sum is 0 + colour: the `acc` stream is not touched at all.
        const bool singleSample = PATHED_SHADE_TRIM && p.chunk == 1;
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
        result = rgb(0.f);
        outMod = make_float4(1.f, 1.f, 1.f, 1.f);
        outThr = make_float4(0.f, 0.f, 0.f, 0.f);
        outPend = make_float4(0.f, 0.f, 0.f, 0.f);
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

    // block-aggregated grab of the next units (wave ballot + LDS scan, one atomic per block)
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

    SHADE_REGION(8, startNext);
    SHADE_REGION(9, shadow.push);
    if (startNext) { startSample(p, nextPixel, nextSample, sampleInUnit, &outRayO, &outRayD); }

    // ---- [r5] local rays.  Most rays of "an object on a floor under a sky" never come near the object: 87 % of the queries of
    // the dragon configuration miss the mesh's bounding box or its bounding sphere (profiles/r5_ab_hybrid_large.log).  Such a
    // ray can only hit one of the scene's few LARGE triangles (RenderParams::localTris), which this kernel tests itself, with
    // the tree walk's intersector and acceptance rule (same hit, bit for bit): the slot's ray then skips the trace kernel
    // (kStLocal), an occlusion ray is decided here and never listed.  Scene::testIntersect / testOcclusion, src/scene.cpp:91-223, 355-381.
    if (p.localCount > 0) {
        const bool hasRay = active && !retired;
        auto meetsRest = [&](V3 origin, V3 direction, float tfar) -> bool {
            return hybridProxy(p.hybridLo, p.hybridHi, origin, direction, tfar) && hybridProxySphere(p.hybridSphere, origin, direction);
        };
        const V3 nextO = v3(outRayO.x, outRayO.y, outRayO.z), nextD = v3(outRayD.x, outRayD.y, outRayD.z);
        const bool localClosest = hasRay && !meetsRest(nextO, nextD, PATHED_TFAR);
        const bool localShadow = shadow.push && !meetsRest(shadow.origin, shadow.direction, shadow.tfar);
        if (__ballot(localClosest || localShadow) != 0ull) {
            float best = PATHED_TFAR, bestU = 0.f, bestV = 0.f;
            int bestPrim = -1;
            bool occluded = false;
            for (int k = 0; k < p.localCount; k++) {
                const float4 t0 = p.localTris[3 * k + 0], t1 = p.localTris[3 * k + 1], t2 = p.localTris[3 * k + 2];   // kernel arguments: scalar loads
                const V3 v0 = v3(t0.x, t0.y, t0.z), e1 = v3(t1.x, t1.y, t1.z), e2 = v3(t2.x, t2.y, t2.z);
                const int prim = floatAsInt(t0.w);
                float t, u, v;
                if (localClosest && intersectTriangle(nextO, nextD, v0, e1, e2, &t, &u, &v) && t > PATHED_TNEAR) {
                    // testLeafTriangle's rule (trace.h)
                    const bool closer = (bestPrim < 0) ? (t <= best) : (t < best || (t == best && prim < bestPrim));
                    if (closer) { best = t; bestU = u; bestV = v; bestPrim = prim; }
                }
                if (localShadow && !occluded && intersectTriangle(shadow.origin, shadow.direction, v0, e1, e2, &t, &u, &v) && t > PATHED_TNEAR && t <= shadow.tfar) {
                    occluded = true;
                }
            }
            if (localClosest) {
                p.state.hit[slot] = make_float4(best, bestU, bestV, intAsFloat(bestPrim));
                outRayD.w = intAsFloat(floatAsInt(outRayD.w) | kStLocal);
            }
            if (localShadow) {
                if (occluded) { outPend = make_float4(0.f, 0.f, 0.f, 0.f); }
                shadow.push = false;
            }
            if (p.localCounting) {
                const unsigned long long closestMask = __ballot(localClosest), shadowMask = __ballot(localShadow);
                if (lane == 0) {
                    atomicAdd(&p.stats[kStatLocalClosest], (unsigned long long)__popcll(closestMask));
                    atomicAdd(&p.stats[kStatLocalShadow], (unsigned long long)__popcll(shadowMask));
                }
            }
        }
        // whatever is left for the trace kernel makes the slot wait for it (kStAwait: later shade launches of this iteration pass it by)
        if (hasRay && (!(floatAsInt(outRayD.w) & kStLocal) || shadow.push)) { outRayD.w = intAsFloat(floatAsInt(outRayD.w) | kStAwait); }
    }

    if (active) {
        p.state.rayO[slot] = outRayO;
        p.state.rayD[slot] = outRayD;
        p.state.res[slot] = make_float4(result.r, result.g, result.b, intAsFloat((int)unit));
        // a slot that starts a camera ray (or retires) keeps its stale mod / thr / pend: the camera ray's vertex reads
        // none of them (rayBounce == 0, no eligible bit) and writes all three for the rays that follow -- 48 bytes less
        // per regenerated slot on streams that bound this kernel
        if (!finished || PATHED_SHADE_TRIM != 1) {   // (PATHED_SHADE_TRIM 2: only the acc stream is trimmed)
            p.state.mod[slot] = outMod;
            p.state.thr[slot] = outThr;
            p.state.pend[slot] = outPend;
        }
    }

    // ---- shadow-ray list: wave ballot + prefix popcount + LDS block scan, then ONE atomic per
    // block reserves the block's range in the iteration's dense list (its order varies from run
    // to run; results do not: each shadow ray only ever clears its own slot's pending term)
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
            const unsigned int index = scratch[kWavesPerBlock] + offset + before;
            p.state.shO[index] = make_float4(shadow.origin.x, shadow.origin.y, shadow.origin.z, shadow.tfar);
            p.state.shD[index] = make_float4(shadow.direction.x, shadow.direction.y, shadow.direction.z, intAsFloat(slot));
        }
    }

    // slots that ran out of units (only at the tail of a render call)
    const unsigned long long retiredMask = __ballot(retired);
    if (lane == 0 && retiredMask != 0ull) {
        atomicSub(&p.counters[kCtrRemaining], (unsigned int)__popcll(retiredMask));
    }
}

// the wavefront's local rays (RenderParams::localTris): can a ray meet anything but the scene's large triangles, and its closest
// hit among those (the tree walk's intersector and acceptance rule)
__device__ __forceinline__ bool localMeetsRest(const RenderParams &p, V3 origin, V3 direction, float tfar)
{
    return hybridProxy(p.hybridLo, p.hybridHi, origin, direction, tfar) && hybridProxySphere(p.hybridSphere, origin, direction);
}
// (call it from WAVE-UNIFORM control flow, lanes without a ray pass want = false: inside a divergent branch the compiler takes
// the loop counter for a per-lane value, cannot index the kernel arguments with it and keeps a private copy of all 1.6 KB)
__device__ __forceinline__ float4 localClosestHit(const RenderParams &p, bool want, V3 origin, V3 direction)
{
    float best = PATHED_TFAR, bestU = 0.f, bestV = 0.f;
    int bestPrim = -1;
    if (__ballot(want) == 0ull) { return make_float4(best, bestU, bestV, intAsFloat(bestPrim)); }
    for (int k = 0; k < p.localCount; k++) {
        const float4 t0 = p.localTris[3 * k + 0], t1 = p.localTris[3 * k + 1], t2 = p.localTris[3 * k + 2];   // kernel arguments: scalar loads
        const int prim = floatAsInt(t0.w);
        float t, u, v;
        if (want && intersectTriangle(origin, direction, v3(t0.x, t0.y, t0.z), v3(t1.x, t1.y, t1.z), v3(t2.x, t2.y, t2.z), &t, &u, &v) && t > PATHED_TNEAR) {
            const bool closer = (bestPrim < 0) ? (t <= best) : (t < best || (t == best && prim < bestPrim));   // testLeafTriangle's rule (trace.h)
            if (closer) { best = t; bestU = u; bestV = v; bestPrim = prim; }
        }
    }
    return make_float4(best, bestU, bestV, intAsFloat(bestPrim));
}
__device__ __forceinline__ bool localOccluded(const RenderParams &p, bool want, V3 origin, V3 direction, float tfar)
{
    bool occluded = false;
    if (__ballot(want) == 0ull) { return false; }
    for (int k = 0; k < p.localCount; k++) {
        const float4 t0 = p.localTris[3 * k + 0], t1 = p.localTris[3 * k + 1], t2 = p.localTris[3 * k + 2];
        float t, u, v;
        if (want && !occluded && intersectTriangle(origin, direction, v3(t0.x, t0.y, t0.z), v3(t1.x, t1.y, t1.z), v3(t2.x, t2.y, t2.z), &t, &u, &v)
            && t > PATHED_TNEAR && t <= tfar) { occluded = true; }
    }
    return occluded;
}

// k_shade for scenes lit by the ENVIRONMENT alone (no emissive surface: what k_shade<.., ENV_ONLY = true> served), with a CHAIN
// [r5]: a sample that ends when its ray comes back -- a miss, the last bounce, a black throughput: most samples of "an
// object on a floor under a sky" -- starts the next sample's camera ray at once, and if that ray is a LOCAL one (it cannot
// meet the mesh: k_shade's local rays) and hits one of the scene's large triangles, the new sample's first vertex is shaded in
// this very launch, by the lane that would otherwise sit the vertex code of its wave out.  One slot visit -- 112 bytes read,
// 128 written, one trace launch waited for -- less per such sample; the dragon configuration's typical sample (floor, sky) is
// two visits instead of three.  Every operation on a path's values is k_shade's, in k_shade's order: same floats
// (tests/test_gpu_local_rays.py, test_gpu_operating_size.py).  The finish logic needs no intersection record here (no surface
// emits), so makeIsect stays ONE inlined copy, in front of the vertex code.
// Five waves per SIMD (96 VGPRs, 12 bytes of spills) like k_shade: left alone the compiler takes 100 - 104 and four waves, and
// the teapot / dragon configurations run 1 - 2 % slower (profiles/r5_ab_shade_env_waves.log)
#ifndef PATHED_SHADE_ENV_WAVES
#define PATHED_SHADE_ENV_WAVES 5
#endif
#define PATHED_SHADE_ENV_ATTRIBUTE __attribute__((amdgpu_waves_per_eu(PATHED_SHADE_ENV_WAVES, PATHED_SHADE_ENV_WAVES)))
template <bool LDS_MATERIALS, typename TRAITS = TraitsAll>
__global__ __launch_bounds__(kBlock) PATHED_SHADE_ENV_ATTRIBUTE void k_shade_env(RenderParams p)
{
    __shared__ DMaterial ldsMaterials[LDS_MATERIALS ? kMaxLdsMaterials : 1];
    __shared__ unsigned int scratch[kWavesPerBlock + 1];

    MaterialAccess<LDS_MATERIALS> materials;
    if (LDS_MATERIALS) {
        const int words = p.scene.nMaterials * (int)(sizeof(DMaterial) / 4);
        const int *source = reinterpret_cast<const int *>(p.scene.materials);
        int *target = reinterpret_cast<int *>(ldsMaterials);
        for (int i = threadIdx.x; i < words; i += kBlock) { target[i] = source[i]; }
        __syncthreads();
        materials.table = ldsMaterials;
    } else {
        materials.table = p.scene.materials;
    }

    const int slot = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const DScene &scene = p.scene;

    // rewind the card cursors for the pool's next trace launch, empty the shadow list the NEXT shade launch fills (k_shade)
    if (blockIdx.x == 0 && threadIdx.x < kTraceShards) { p.counters[kCtrTraceCursor + threadIdx.x * kCursorStride] = 0u; }
    if (blockIdx.x == 0 && threadIdx.x == kTraceShards) { p.counters[kCtrShadowCount + (p.parity ^ 1) * kCursorStride] = 0u; }

    float4 rd = p.state.rayD[slot];
    float4 h = p.state.hit[slot];
    const float4 roIn = p.state.rayO[slot];
    const float4 resIn0 = p.state.res[slot];
    pinLoaded(rd);
    pinLoaded(h);
    pinLoaded(roIn);
    pinLoaded(resIn0);
    int st = floatAsInt(rd.w);
    bool active = !(st & kStDone);
    float4 pendIn = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        if (st & kStEligible) { pendIn = p.state.pend[slot]; }
        if (p.suspendLanes > 0) {
            const bool parked = floatAsInt(h.w) == kPrimSuspended
                || ((st & kStEligible) && floatAsInt(pendIn.w) == kShadowSuspended);
            if (parked) {
                if (!(st & kStHold)) { reinterpret_cast<int *>(p.state.rayD + slot)[3] = st | kStHold; }
                active = false;
            }
            st &= ~kStHold;
            rd.w = intAsFloat(st);
        }
    }
    if (!p.afterTrace && (st & kStAwait)) { active = false; }
    st &= ~(kStLocal | kStAwait);
    rd.w = intAsFloat(st);

    ShadowRequest shadow;
    shadow.push = false;
    shadow.origin = v3(0.f, 0.f, 0.f);
    shadow.direction = v3(0.f, 0.f, 0.f);
    shadow.tfar = 0.f;

    bool finished = false;        // the sample in flight ended
    Rgb color = rgb(0.f);
    unsigned int unit = 0xFFFFFFFFu;
    int sampleInUnit = 0;
    float4 outRayO = make_float4(0.f, 0.f, 0.f, 0.f), outRayD = rd;
    float4 outMod = make_float4(1.f, 1.f, 1.f, 1.f);
    float4 outThr = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 outPend = make_float4(0.f, 0.f, 0.f, 0.f);
    Rgb result = rgb(0.f);
    // the vertex the second half shades: the one the slot's ray reached, or the first of the sample the slot starts
    V3 o = v3(0.f, 0.f, 0.f), d = v3(0.f, 0.f, 1.f);
    float4 hitNow = h;
    int rayBounce = 0, firstEmitMaterial = -1;
    uint32_t pixel = 0, sample = 0;
    bool haveVertex = false;
    bool hitWritten = false;      // outRayD's ray is a local one and its hit is in localHit
    float4 localHit = make_float4(PATHED_TFAR, 0.f, 0.f, intAsFloat(-1));

    unsigned int countedClosest = 0u, countedShadow = 0u;

    SHADE_REGION(0, true);
    SHADE_REGION(1, active);
    // ---- first half: what the ray that came back means for the sample in flight (k_shade's logic without an intersection
    // record: no surface of such a scene emits, so a hit contributes nothing here)
    if (active) {
        const float4 ro = roIn;
        const float4 resIn = resIn0;
        outRayO = ro;
        o = v3(ro.x, ro.y, ro.z);
        d = v3(rd.x, rd.y, rd.z);
        rayBounce = st & kStBounceMask;
        sampleInUnit = (st >> kStSampleShift) & kStSampleMask;
        const bool miss = floatAsInt(h.w) < 0;
        unit = (unsigned int)floatAsInt(resIn.w);
        firstEmitMaterial = floatAsInt(ro.w);
        uint32_t firstSample, endSample;
        unitSamples(p, unit, &pixel, &firstSample, &endSample);
        sample = firstSample + (uint32_t)sampleInUnit;
        result = rgb(resIn.x, resIn.y, resIn.z);
        Rgb modulation = rgb(1.f);
        // what the environment sends along a ray that left the scene: looked up ONCE, for the camera rays and the BSDF samples of
        // the wave together (the direction-to-texel mapping is an atan2f and an acosf; as two inlined copies a wave with both
        // kinds of miss -- every wave of "an object under a sky" -- ran it twice)
        Rgb missLight = rgb(0.f);
        if (miss && (rayBounce == 0 || (st & kStEligible))) { missLight = environmentL<TRAITS>(scene, d); }

        if (rayBounce == 0) {
            // SampleIntegrator::samplePixel, src/sample_integrator.cpp:18-59
            if (miss) {
                color = rgb(0.f) + missLight;
                finished = true;
            } else {
                firstEmitMaterial = -1;
                result = rgb(0.f);
                haveVertex = true;
            }
        } else {
            const float4 modIn = p.state.mod[slot];
            const float4 thrIn = p.state.thr[slot];
            modulation = rgb(modIn.x, modIn.y, modIn.z);
            const float bsdfPdf = modIn.w;
            const Rgb throughput = rgb(thrIn.x, thrIn.y, thrIn.z);
            const float cosTheta = thrIn.w;
            if (st & kStEligible) {
                // PathTracer::directSampleBSDF, src/path_tracer.cpp:167-216
                Rgb bsdfTerm = rgb(0.f);
                if (miss) {
                    const Rgb environmentLight = missLight;
                    if (!isBlack(environmentLight)) {
                        const float lightPDF = envEmitPDF(scene.env, d) / 1;
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
            // PathTracer::L loop body, src/path_tracer.cpp:41-58
            if (!(st & kStContinue) || miss) {
                finished = true;
            } else {
                const float invPDF = 1.f / bsdfPdf;
                modulation = modulation * (throughput * cosTheta * invPDF);
                if (isBlack(modulation)) { finished = true; }
                else { haveVertex = true; }
            }
            if (finished) {
                const Rgb first = rgb(0.f);
                color = first + result;
            }
        }
        outMod = make_float4(modulation.r, modulation.g, modulation.b, 1.f);
    }

    // ---- end of a sample, next unit, camera ray: used after either half (a macro, not a lambda: a closure that is called twice
    // takes the address of the kernel's arguments, and the compiler then keeps a private copy of all 1.6 KB of them)
    bool retiredAny = false;
#define PATHED_END_OF_SAMPLE(ended, value, startedNext, nextPixel, nextSample)                                                 \
    {                                                                                                                        \
        bool needUnit_ = false;                                                                                              \
        startedNext = false;                                                                                                 \
        SHADE_REGION(7, ended);                                                                                              \
        if (ended) {                                                                                                         \
            const bool singleSample_ = PATHED_SHADE_TRIM && p.chunk == 1;                                                    \
            float4 partial_ = make_float4(0.f, 0.f, 0.f, 0.f);                                                               \
            if (!singleSample_) { partial_ = p.state.acc[slot]; }                                                            \
            const bool finite_ = isfinite((value).r) && isfinite((value).g) && isfinite((value).b);                          \
            if (finite_) {                                                                                                   \
                partial_.x += (value).r;                                                                                     \
                partial_.y += (value).g;                                                                                     \
                partial_.z += (value).b;                                                                                     \
            } else {                                                                                                         \
                atomicAdd(&p.stats[kStatDropped], 1ull);                                                                     \
            }                                                                                                                \
            uint32_t unitPixel_, firstSample_, endSample_;                                                                   \
            unitSamples(p, unit, &unitPixel_, &firstSample_, &endSample_);                                                   \
            sampleInUnit++;                                                                                                  \
            result = rgb(0.f);                                                                                               \
            outMod = make_float4(1.f, 1.f, 1.f, 1.f);                                                                        \
            outThr = make_float4(0.f, 0.f, 0.f, 0.f);                                                                        \
            outPend = make_float4(0.f, 0.f, 0.f, 0.f);                                                                       \
            if (firstSample_ + (uint32_t)sampleInUnit < endSample_) {                                                        \
                startedNext = true;                                                                                          \
                nextPixel = unitPixel_;                                                                                      \
                nextSample = firstSample_ + (uint32_t)sampleInUnit;                                                          \
                p.state.acc[slot] = partial_;                                                                                \
            } else {                                                                                                         \
                p.state.chunkBuf[partialIndex(p, unit)] = partial_;                                                          \
                if (!singleSample_) { p.state.acc[slot] = make_float4(0.f, 0.f, 0.f, 0.f); }                                 \
                needUnit_ = true;                                                                                            \
            }                                                                                                                \
        }                                                                                                                    \
        const unsigned int newUnit_ = grabUnits(p, needUnit_, scratch);                                                      \
        if (needUnit_) {                                                                                                     \
            unit = newUnit_;                                                                                                 \
            if (newUnit_ != 0xFFFFFFFFu) {                                                                                   \
                uint32_t endSample_;                                                                                         \
                unitSamples(p, newUnit_, &nextPixel, &nextSample, &endSample_);                                              \
                sampleInUnit = 0;                                                                                            \
                startedNext = true;                                                                                          \
            } else {                                                                                                         \
                outRayD.w = intAsFloat(kStDone);                                                                             \
                retiredAny = true;                                                                                           \
            }                                                                                                                \
        }                                                                                                                    \
        SHADE_REGION(8, startedNext);                                                                                        \
        if (startedNext) { startSample(p, nextPixel, nextSample, sampleInUnit, &outRayO, &outRayD); }                        \
    }

    // ---- the sample ended with the ray that came back: the next one starts HERE, and if its camera ray is a local one that
    // hits a large triangle, its first vertex is the one the second half shades
    {
        bool startedNext;
        uint32_t nextPixel = 0, nextSample = 0;
        PATHED_END_OF_SAMPLE(active && finished, color, startedNext, nextPixel, nextSample)
        if (p.localCount > 0) {
            const V3 cameraO = v3(outRayO.x, outRayO.y, outRayO.z), cameraD = v3(outRayD.x, outRayD.y, outRayD.z);
            const bool localCamera = startedNext && !localMeetsRest(p, cameraO, cameraD, PATHED_TFAR);
            const float4 found = localClosestHit(p, localCamera, cameraO, cameraD);
            if (localCamera) {
                countedClosest++;
                if (floatAsInt(found.w) >= 0) {
                    o = cameraO; d = cameraD; hitNow = found;
                    rayBounce = 0; firstEmitMaterial = -1;
                    pixel = nextPixel; sample = nextSample;
                    result = rgb(0.f);
                    haveVertex = true;
                } else {
                    // it leaves the scene: the next launch ends that sample (the slot's ray needs no trace)
                    localHit = found;
                    hitWritten = true;
                    outRayD.w = intAsFloat(floatAsInt(outRayD.w) | kStLocal);
                }
            }
        }
    }

    // ---- second half: the vertex (k_shade's, with makeIsect in front of it)
    SHADE_REGION(5, haveVertex);
    if (haveVertex) {
        Isect isect = makeIsect<TRAITS>(scene, o, d, hitNow);
        const int vertex = rayBounce + 1;
        // PathTracer::L: sample the BSDF, then direct(), src/path_tracer.cpp:30-36, 60-73
        const DMaterial &material = materials[isect.material];
        prepareLobes<TRAITS>(material, isect);
        Rng random;
        makeKey(((uint64_t)p.seedHi << 32) | p.seedLo, pixel, sample, &random.k0, &random.k1);
        random.dimension = vertexBase(vertex);
        const BSDFSample bsdfSample = materialSample<TRAITS>(material, isect, random);

        const bool counts = checkCounts(p.startBounce, p.lastBounce, vertex);
        const bool wantDirect = counts;          // (no surface emits)
        const bool wantContinue = !checkDone(p.lastBounce, vertex + 1);

        Rgb lightTerm = rgb(0.f);
        SHADE_REGION(6, wantDirect);
        if (wantDirect) {
            random.dimension = vertexBase(vertex) + 3;
            lightTerm = sampleLightsTerm<true, TRAITS>(scene, materials, isect, material, random, &shadow);
        }
        const bool deadEnd = isBlack(bsdfSample.throughput) && bsdfSample.pdf > 0.f && bsdfSample.pdf < 3e38f
            && !shadow.push && isBlack(lightTerm);
        if ((!wantDirect && !wantContinue) || deadEnd) {
            // The sample ends with this vertex (the last bounce outside the window, a dead end: a few lanes in a hundred).  Ending
            // it here would be a second end-of-sample + regeneration block that nearly every wave runs for one or two lanes:
            // instead the slot leaves with a ray that is already answered -- a local miss, not eligible, not to be continued --
            // and the NEXT launch's first half ends the sample: `color = first + result`, these bits.
            outRayO = make_float4(isect.point.x, isect.point.y, isect.point.z, intAsFloat(firstEmitMaterial));
            outRayD = make_float4(d.x, d.y, d.z, intAsFloat((vertex | (sampleInUnit << kStSampleShift)) | kStLocal));
            outMod.w = 1.f;
            outThr = make_float4(0.f, 0.f, 0.f, 0.f);
            outPend = make_float4(0.f, 0.f, 0.f, 0.f);
            localHit = make_float4(PATHED_TFAR, 0.f, 0.f, intAsFloat(-1));
            hitWritten = true;
            shadow.push = false;
        } else {
            int nextState = vertex | (sampleInUnit << kStSampleShift);
            if (wantDirect) { nextState |= kStEligible; }
            if (isDeltaT<TRAITS>(material)) { nextState |= kStDelta; }
            if (wantContinue) { nextState |= kStContinue; }
            outRayO = make_float4(isect.point.x, isect.point.y, isect.point.z, intAsFloat(firstEmitMaterial));
            outRayD = make_float4(bsdfSample.wiWorld.x, bsdfSample.wiWorld.y, bsdfSample.wiWorld.z, intAsFloat(nextState));
            if (rayBounce == 0) { outMod = make_float4(1.f, 1.f, 1.f, 1.f); }   // (a camera-ray vertex: the modulation is 1)
            outMod.w = bsdfSample.pdf;
            outThr = make_float4(
                bsdfSample.throughput.r, bsdfSample.throughput.g, bsdfSample.throughput.b,
                fabsf(dot(isect.shadingNormal, bsdfSample.wiWorld)));
            outPend = make_float4(lightTerm.r, lightTerm.g, lightTerm.b, 0.f);
            hitWritten = false;
        }
    }
#undef PATHED_END_OF_SAMPLE

    // ---- local rays (k_shade): the ray the slot leaves with, and its vertex's occlusion ray
    if (p.localCount > 0) {
        const bool retiredNow = (floatAsInt(outRayD.w) & kStDone) != 0;
        const bool hasRay = active && !retiredNow;
        const V3 nextO = v3(outRayO.x, outRayO.y, outRayO.z), nextD = v3(outRayD.x, outRayD.y, outRayD.z);
        const bool localClosest = hasRay && !hitWritten && !localMeetsRest(p, nextO, nextD, PATHED_TFAR);
        const bool localShadow = shadow.push && !localMeetsRest(p, shadow.origin, shadow.direction, shadow.tfar);
        {
            const float4 found = localClosestHit(p, localClosest, nextO, nextD);
            if (localClosest) {
                localHit = found;
                hitWritten = true;
                outRayD.w = intAsFloat(floatAsInt(outRayD.w) | kStLocal);
                countedClosest++;
            }
            const bool occluded = localOccluded(p, localShadow, shadow.origin, shadow.direction, shadow.tfar);
            if (localShadow) {
                if (occluded) { outPend = make_float4(0.f, 0.f, 0.f, 0.f); }
                shadow.push = false;
                countedShadow++;
            }
        }
        if (hasRay && (!(floatAsInt(outRayD.w) & kStLocal) || shadow.push)) { outRayD.w = intAsFloat(floatAsInt(outRayD.w) | kStAwait); }
        if (p.localCounting) {
            for (int offset = 32; offset >= 1; offset >>= 1) {
                countedClosest += (unsigned int)__shfl_xor((int)countedClosest, offset, 64);
                countedShadow += (unsigned int)__shfl_xor((int)countedShadow, offset, 64);
            }
            if (lane == 0 && (countedClosest | countedShadow) != 0u) {
                atomicAdd(&p.stats[kStatLocalClosest], (unsigned long long)countedClosest);
                atomicAdd(&p.stats[kStatLocalShadow], (unsigned long long)countedShadow);
            }
        }
    }

    if (active) {
        p.state.rayO[slot] = outRayO;
        p.state.rayD[slot] = outRayD;
        p.state.res[slot] = make_float4(result.r, result.g, result.b, intAsFloat((int)unit));
        if (hitWritten) { p.state.hit[slot] = localHit; }
        p.state.mod[slot] = outMod;
        p.state.thr[slot] = outThr;
        p.state.pend[slot] = outPend;
    }

    // ---- shadow-ray list (k_shade)
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
            const unsigned int index = scratch[kWavesPerBlock] + offset + before;
            p.state.shO[index] = make_float4(shadow.origin.x, shadow.origin.y, shadow.origin.z, shadow.tfar);
            p.state.shD[index] = make_float4(shadow.direction.x, shadow.direction.y, shadow.direction.z, intAsFloat(slot));
        }
    }

    const unsigned long long retiredMask = __ballot(retiredAny);
    if (lane == 0 && retiredMask != 0ull) {
        atomicSub(&p.counters[kCtrRemaining], (unsigned int)__popcll(retiredMask));
    }
}

// ------------------------------------------------------------------------- fused path kernel (tiny scenes)
// k_path_small: scenes of <= kBruteForceMaxTris triangles, whose ray queries are two straight-line passes
// over kernarg-resident triangle records, need no ray or hit buffers at all.  One lane carries one PATH from
// camera ray to termination with its whole state in registers -- the wavefront iteration
// "trace, shade, trace shadow, regenerate" becomes the body of one loop -- and waves are persistent: a lane
// whose sample ended takes the next sample of its work unit, or the next unit from the wave's reserved
// range (one wave-level atomic per kUnitGrab units on a sharded cursor; a wave whose shard is dealt out
// moves on to the next).  No path state touches HBM: per unit one 16-byte partial sum is written.
// Every operation on a path's values is k_shade's, in k_shade's order, and the unit decomposition fixes
// the summation order, so the radiance sums are the wavefront kernels' bit for bit (GPU test).
// What the wavefront keeps and this gives up is the DENSE shadow-ray list: a shadow ray is traced by the lane
// whose vertex asked for it (~70-75 % of the lanes on Cornell).  It leaves the vertex like the continuation ray
// does, so the two share one pass over the triangles and a third of its arithmetic (smallCandidatesPair).
#ifndef PATHED_FUSED_WAVES
#define PATHED_FUSED_WAVES 4
#endif
// TRAITS: the compile-time set of material / light / albedo kinds the scene may contain (shading.h: SceneTraits)
// MFMA: phase 1 of both ray queries on the matrix pipe (mfma_candidates.h) instead of the VALU pass (experiments build)
// QUADS: some triangles of the scene are halves of parallelograms (small_items.h): phase 1 runs over the item records
// (parallelograms, then the triangles without a partner); otherwise the exact pair-of-triangles test over all of them
template <bool LDS_MATERIALS, bool COUNT, typename TRAITS, bool MFMA = false, bool QUADS = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PATHED_FUSED_WAVES, PATHED_FUSED_WAVES))) void k_path_small(RenderParams p, SmallTris smallTris)
{
    extern __shared__ float4 ldsDynamic[];           // LDS_MATERIALS: the material table, nMaterials x 96 B
    __shared__ float mfmaRows[MFMA ? kMfmaTableFloats : 1];
    // QUADS: what a lane's path carries across the pass over the triangles and does not use in it waits in LDS, 96 bytes per
    // lane ([quad][thread]: a wave stores 1 KiB contiguous): the parallelogram test needs two dozen registers more than the
    // pair-of-triangles test, and what the compiler spilled for them went to scratch memory (48 dwords, profiles/r4_ab_quads.log)
    __shared__ float4 stashRows[QUADS ? 6 * kBlock : 1];
    // ... and phase 2 shares the wave's candidates out (smallResolveShared): 2.25 KiB per wave.  With the stash 33.5 KiB per
    // block: four blocks per CU as long as the material table stays below 6.5 KiB (pathed_hip.hip pairs triangles up to 64 materials)
    constexpr int kResolveCapacity = PATHED_RESOLVE_SHARED == 2 ? 128 : 64;   // items per wave
    __shared__ unsigned long long resolveBest[QUADS ? kBlock : 1];
    __shared__ float2 resolveUv[QUADS ? kWavesPerBlock * kResolveCapacity : 1];
    __shared__ unsigned short resolveItems[QUADS ? kWavesPerBlock * kResolveCapacity : 1];
    __shared__ unsigned int resolveOccluded[QUADS ? kBlock : 1], resolveCount[QUADS ? kWavesPerBlock : 1];
    if (MFMA) {
        for (int i = threadIdx.x; i < kMfmaTableFloats; i += kBlock) { mfmaRows[i] = p.mfmaTable[i]; }
        if (!LDS_MATERIALS) { __syncthreads(); }
    }
    MaterialAccess<LDS_MATERIALS> materials;
    if (LDS_MATERIALS) {
        const int words = p.scene.nMaterials * (int)(sizeof(DMaterial) / 4);
        const int *source = reinterpret_cast<const int *>(p.scene.materials);
        int *target = reinterpret_cast<int *>(ldsDynamic);
        for (int i = threadIdx.x; i < words; i += kBlock) { target[i] = source[i]; }
        __syncthreads();
        materials.table = reinterpret_cast<const DMaterial *>(ldsDynamic);
    } else {
        materials.table = p.scene.materials;
    }

    TraceGeometry geometry;
    geometry.nodes = nullptr;
    geometry.tris = p.scene.leafTris;
    geometry.nNodes = 0;
    geometry.nTris = p.scene.nTris;
    geometry.spheres = p.scene.spheres;
    geometry.nSpheres = p.scene.nLinearSpheres;

    const DScene &scene = p.scene;
    const int lane = threadIdx.x & 63;
    const unsigned int waveId = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nTris = p.scene.nTris;
    const uint64_t seed = ((uint64_t)p.seedHi << 32) | p.seedLo;

    // ---- work units: the wave reserves p.unitGrab consecutive units of a queue at a time (wave-uniform state)
    unsigned int queue = waveId % (unsigned int)p.nQueues, queuesTried = 0;
    unsigned int reservedNext = 0, reservedEnd = 0;
    // hands a unit to every lane that wants one, in lane order; 0xFFFFFFFF once the pass is dealt out
    auto takeUnits = [&](bool want) -> unsigned int {
        unsigned int mine = 0xFFFFFFFFu;
        unsigned long long wanting = __ballot(want);
        while (wanting != 0ull) {
            if (reservedNext == reservedEnd) {
                if (queuesTried >= (unsigned int)p.nQueues) { break; }   // every queue is dealt out
                unsigned int ticket = 0;
                if (lane == 0) { ticket = atomicAdd(&p.counters[kCtrUnitCursor + queue * kCursorStride], (unsigned int)p.unitGrab); }
                ticket = (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
                const unsigned int limit = p.queueUnits[queue];   // unit ids of queue q: q * unitsPerQueue + [0, limit)
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

    // ---- the path a lane carries
    bool alive = false;
    unsigned int unit = 0xFFFFFFFFu;
    uint32_t pixel = 0, sample = 0, endSample = 0;   // sample: absolute index of the sample in flight
    Rng random;
    random.k0 = 0u; random.k1 = 0u; random.dimension = 0u;
    V3 o = v3(0.f, 0.f, 0.f), d = v3(0.f, 0.f, 1.f);
    int st = 0;                      // device_scene.h state word: vertex that spawned the ray + eligible / delta / continue
    int firstEmitMaterial = -1;
    Rgb result = rgb(0.f), modulation = rgb(1.f), throughput = rgb(0.f), pend = rgb(0.f);
    float bsdfPdf = 1.f, cosTheta = 0.f;
    float4 partial = make_float4(0.f, 0.f, 0.f, 0.f);
    bool pendingShadow = false;      // the vertex the ray left asked for an occlusion query along shadowDirection
    V3 shadowDirection = v3(0.f, 0.f, 1.f);
    float shadowTfar = 0.f;

    unsigned int closestRays = 0, shadowRays = 0, trisTested = 0;

    // Camera::generateRay(int,int) for (pixel, sample), src/camera.cpp:49-55: one inlined copy, reached from the
    // initial fill and from the end of every sample
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
        SHADE_REGION(0, alive);        // (profile builds, tools/fused_profile.py) iterations / live lanes
        SHADE_REGION(1, startNext);    // camera ray
        if (startNext) {
            makeKey(seed, pixel, sample, &random.k0, &random.k1);
            random.dimension = 0;
            const int width = scene.camera.resX;
            const int row = (int)fastDivide((unsigned int)pixel, p.divWidth);   // pixel / width, exactly
            const int col = (int)pixel - row * width;
            const float jitterX = random.next() - 0.5f;
            const float jitterY = random.next() - 0.5f;
            cameraRay(scene.camera, row + jitterY, col + jitterX, &o, &d);
            st = 0;
            firstEmitMaterial = -1;
            result = rgb(0.f);
            modulation = rgb(1.f);
            throughput = rgb(0.f);
            pend = rgb(0.f);
            bsdfPdf = 1.f;
            cosTheta = 0.f;
            startNext = false;
        }
        if (__ballot(alive) == 0ull) { break; }

        // ---- the path's ray (Scene::testIntersect's rtcIntersect1) and, from the same point, the shadow ray of the
        // vertex it leaves (Scene::testOcclusion's rtcOccluded1): an occluded light sample contributes nothing
        if (QUADS) {
            float4 *stash = stashRows + threadIdx.x;
            stash[0 * kBlock] = make_float4(result.r, result.g, result.b, bsdfPdf);
            stash[1 * kBlock] = make_float4(modulation.r, modulation.g, modulation.b, cosTheta);
            stash[2 * kBlock] = make_float4(throughput.r, throughput.g, throughput.b, intAsFloat(st));
            stash[3 * kBlock] = partial;
            stash[4 * kBlock] = make_float4(intAsFloat((int)pixel), intAsFloat((int)sample), intAsFloat((int)endSample), intAsFloat((int)unit));
            stash[5 * kBlock] = make_float4(intAsFloat((int)random.k0), intAsFloat((int)random.k1), intAsFloat(firstEmitMaterial), intAsFloat((int)random.dimension));
            asm volatile("" ::: "memory");   // the values below are re-read from LDS: the registers are free for the pass
        }
        LaneRay ray;
        laneRayInit(ray, o, d, PATHED_TNEAR, PATHED_TFAR, false);
        {
            const bool traceShadow = alive && pendingShadow;
            LaneRay shadowRay;
            laneRayInit(shadowRay, o, shadowDirection, PATHED_TNEAR, shadowTfar, true);
            unsigned int candidatesLow = 0, candidatesHigh = 0, shadowLow = 0, shadowHigh = 0;
            SHADE_REGION(2, traceShadow);   // passes that carry shadow rays / lanes with one
            if (MFMA) {
                // every lane takes part (the matrix instructions are the wave's); words: even / odd triangles
                mfmaCandidatesPair(mfmaRows, nTris, p.mfmaFrame, __ballot(traceShadow) != 0ull, o, d, shadowDirection,
                                   candidateNear(PATHED_TNEAR), candidateFar(shadowTfar), &candidatesLow, &candidatesHigh, &shadowLow, &shadowHigh);
                if (!alive) { candidatesLow = 0u; candidatesHigh = 0u; }
                if (!traceShadow) { shadowLow = 0u; shadowHigh = 0u; }
            } else if (QUADS) {
                // one instantiation of the pass: a wave without a single shadow ray is rare (3 % of the passes) and pays for
                // the second ray's arithmetic rather than for a second copy of the loop's registers
                if (alive) {
                    smallCandidatesItems<true, true, true>(smallTris.data, p.smallQuads, nTris, p.smallKappaT, o, d, shadowDirection,
                                                            &candidatesLow, &candidatesHigh, &shadowLow, &shadowHigh,
                                                            candidateNear(PATHED_TNEAR), candidateFar(shadowTfar));
                    if (!traceShadow) { shadowLow = 0u; shadowHigh = 0u; }
                }
            } else if (__ballot(traceShadow) != 0ull) {
                if (alive) {
                    smallCandidatesPair(smallTris.data, nTris, o, d, shadowDirection, &candidatesLow, &candidatesHigh, &shadowLow, &shadowHigh,
                                        candidateNear(PATHED_TNEAR), candidateFar(shadowTfar));
                    if (!traceShadow) { shadowLow = 0u; shadowHigh = 0u; }
                }
            } else if (alive) {
                smallCandidates(smallTris.data, nTris, o, d, &candidatesLow, &candidatesHigh, candidateNear(PATHED_TNEAR));
            }
            if (COUNT && alive) {
                trisTested += (unsigned int)nTris * (traceShadow ? 2u : 1u);
                closestRays++;
                if (traceShadow) { shadowRays++; }
            }
            bool spheresDone = false;   // (wave-uniform) the shared phase 2 tested the sphere candidates: no loop over the spheres
            RESOLVE_PROBE(9, candidatesLow, candidatesHigh);   // (profile builds) the resolve loops: iterations, candidates
            RESOLVE_PROBE(10, shadowLow, shadowHigh);
            if (MFMA) {
                mfmaResolve(geometry, ray, candidatesLow, candidatesHigh);
                mfmaResolve(geometry, shadowRay, shadowLow, shadowHigh);
            } else if (QUADS && PATHED_RESOLVE_SHARED) {
                ResolveScratch scratch;
                const int waveBase = (int)(threadIdx.x & ~63u), wave = (int)(threadIdx.x >> 6);
                scratch.best = resolveBest + waveBase; scratch.uv = resolveUv + wave * kResolveCapacity; scratch.items = resolveItems + wave * kResolveCapacity;
                scratch.occluded = resolveOccluded + waveBase; scratch.count = resolveCount + wave;
                // spheres (the Veach scene's lights): a packed line-misses-sphere test for both rays, the exact tests on the shared list
                unsigned int sphereCandidates = 0u, shadowSphereCandidates = 0u;
                if (TRAITS::spheres && geometry.nSpheres > 0 && alive) {
                    smallSphereCandidates(p, geometry.nSpheres, o, d, shadowDirection, &sphereCandidates, &shadowSphereCandidates);
                    if (!traceShadow) { shadowSphereCandidates = 0u; }
                }
                spheresDone = smallResolveShared<kResolveCapacity, PATHED_RESOLVE_SHARED != 2, TRAITS::spheres>(
                    geometry, ray, shadowRay, candidatesLow, candidatesHigh, shadowLow, shadowHigh, sphereCandidates, shadowSphereCandidates, scratch);
            } else {
                smallResolve(geometry, ray, candidatesLow, candidatesHigh);
                smallResolve(geometry, shadowRay, shadowLow, shadowHigh);
            }
            if (alive && !spheresDone) { finishRay(geometry, ray); }
            if (traceShadow) {
                if (!spheresDone) { finishRay(geometry, shadowRay); }
                if (shadowRay.occluded) { pend = rgb(0.f); }
            }
            pendingShadow = false;
        }
        if (QUADS) {
            asm volatile("" ::: "memory");
            const float4 *stash = stashRows + threadIdx.x;
            const float4 s0 = stash[0 * kBlock], s1 = stash[1 * kBlock], s2 = stash[2 * kBlock], s4 = stash[4 * kBlock], s5 = stash[5 * kBlock];
            result = rgb(0.f); result.r = s0.x; result.g = s0.y; result.b = s0.z; bsdfPdf = s0.w;
            modulation.r = s1.x; modulation.g = s1.y; modulation.b = s1.z; cosTheta = s1.w;
            throughput.r = s2.x; throughput.g = s2.y; throughput.b = s2.z; st = floatAsInt(s2.w);
            partial = stash[3 * kBlock];
            pixel = (uint32_t)floatAsInt(s4.x); sample = (uint32_t)floatAsInt(s4.y); endSample = (uint32_t)floatAsInt(s4.z); unit = (unsigned int)floatAsInt(s4.w);
            random.k0 = (uint32_t)floatAsInt(s5.x); random.k1 = (uint32_t)floatAsInt(s5.y); firstEmitMaterial = floatAsInt(s5.z); random.dimension = (uint32_t)floatAsInt(s5.w);
        }
        const bool miss = ray.bestPrim < 0;
        const float4 h = make_float4(ray.best, ray.bestU, ray.bestV, intAsFloat(ray.bestPrim));

        // ---- the vertex: k_shade's body on register state
        ShadowRequest shadow;
        shadow.push = false;
        shadow.origin = v3(0.f, 0.f, 0.f);
        shadow.direction = v3(0.f, 0.f, 1.f);
        shadow.tfar = 0.f;
        bool finished = false;
        Rgb color = rgb(0.f);
        if (alive) {
            const int rayBounce = st & kStBounceMask;  // vertex that spawned this ray, 0 = camera
            bool haveVertex = false;
            Isect isect;
            const int vertex = rayBounce + 1;
            SHADE_REGION(3, !miss);        // makeIsect
            if (!miss) { isect = makeIsect<TRAITS>(scene, o, d, h); }

            SHADE_REGION(4, rayBounce == 0);   // camera-ray vertex
            if (rayBounce == 0) {
                // SampleIntegrator::samplePixel, src/sample_integrator.cpp:18-59
                if (miss) {
                    color = rgb(0.f) + environmentL<TRAITS>(scene, d);
                    finished = true;
                } else {
                    firstEmitMaterial = -1;
                    if (checkCounts(p.startBounce, p.lastBounce, 0)) {
                        const Rgb emit = matEmit(materials[isect.material]);
                        const bool backside = dot(isect.normal, isect.wo) < 0.f;
                        if (!isBlack(emit) && !backside) { firstEmitMaterial = isect.material; }
                    }
                    result = rgb(0.f);
                    haveVertex = true;
                }
            } else {
                // the ray left vertex `rayBounce` along its BSDF sample
                if (st & kStEligible) {
                    // PathTracer::directSampleBSDF, src/path_tracer.cpp:167-216
                    Rgb bsdfTerm = rgb(0.f);
                    if (!miss) {
                        const Rgb emit = matEmit(materials[isect.material]);
                        SHADE_REGION(5, !isBlack(emit) && dot(isect.wo, isect.shadingNormal) >= 0.f);   // BSDF sample met an emitter: lightsPDF
                        if (!isBlack(emit) && dot(isect.wo, isect.shadingNormal) >= 0.f) {
                            const float lightPDF = lightsPDF<TRAITS>(scene, o, isect);
                            const float brdfWeight = (st & kStDelta)
                                ? 1.f
                                : (1 * bsdfPdf) / (1 * bsdfPdf + 1 * lightPDF);
                            bsdfTerm = emit * brdfWeight * throughput * cosTheta / bsdfPdf;
                        }
                    } else {
                        const Rgb environmentLight = environmentL<TRAITS>(scene, d);
                        if (TRAITS::env && !isBlack(environmentLight)) {
                            // Scene::environmentPDF, src/scene.cpp:494-502
                            const float lightPDF = envEmitPDF(scene.env, d) / scene.nLights;
                            const float brdfWeight = (st & kStDelta)
                                ? 1.f
                                : (1 * bsdfPdf) / (1 * bsdfPdf + 1 * lightPDF);
                            bsdfTerm = environmentLight * brdfWeight * throughput * cosTheta / bsdfPdf;
                        }
                    }
                    const Rgb Ld = pend + bsdfTerm;
                    if (rayBounce == 1) { result = Ld; }
                    else { result = result + Ld * modulation; }
                }

                // PathTracer::L loop body, src/path_tracer.cpp:41-58
                if (!(st & kStContinue) || miss) {
                    finished = true;
                } else {
                    const float invPDF = 1.f / bsdfPdf;
                    modulation = modulation * (throughput * cosTheta * invPDF);
                    if (isBlack(modulation)) { finished = true; }
                    else { haveVertex = true; }
                }
                if (finished) {
                    Rgb first = rgb(0.f);
                    if (firstEmitMaterial >= 0) { first = first + matEmit(materials[firstEmitMaterial]); }
                    color = first + result;
                }
            }

            SHADE_REGION(6, haveVertex);   // new vertex: BSDF sample
            if (haveVertex) {
                // PathTracer::L: sample the BSDF, then direct(), src/path_tracer.cpp:30-36, 60-73
                const DMaterial &material = materials[isect.material];
                prepareLobes<TRAITS>(material, isect);

                random.dimension = vertexBase(vertex);
                const BSDFSample bsdfSample = materialSample<TRAITS>(material, isect, random);

                const bool counts = checkCounts(p.startBounce, p.lastBounce, vertex);
                const bool emissive = !isBlack(matEmit(material));
                const bool wantDirect = counts && !emissive;  // direct() returns 0 on emitters (:86-90)
                const bool wantContinue = !checkDone(p.lastBounce, vertex + 1);

                Rgb lightTerm = rgb(0.f);
                SHADE_REGION(7, wantDirect);   // light sampling
                if (wantDirect) {
                    random.dimension = vertexBase(vertex) + 3;
                    lightTerm = sampleLightsTerm<false, TRAITS>(scene, materials, isect, material, random, &shadow);
                }

                // see k_shade: a vertex with nothing pending whose BSDF sample has exactly black throughput ends the sample
                const bool deadEnd = isBlack(bsdfSample.throughput) && bsdfSample.pdf > 0.f && bsdfSample.pdf < 3e38f
                    && !shadow.push && isBlack(lightTerm);
                if ((!wantDirect && !wantContinue) || deadEnd) {
                    finished = true;
                    Rgb first = rgb(0.f);
                    if (firstEmitMaterial >= 0) { first = first + matEmit(materials[firstEmitMaterial]); }
                    color = first + result;
                    shadow.push = false;
                } else {
                    int nextState = vertex;
                    if (wantDirect) { nextState |= kStEligible; }
                    if (isDeltaT<TRAITS>(material)) { nextState |= kStDelta; }
                    if (wantContinue) { nextState |= kStContinue; }
                    st = nextState;
                    o = isect.point;
                    d = bsdfSample.wiWorld;
                    bsdfPdf = bsdfSample.pdf;
                    throughput = bsdfSample.throughput;
                    cosTheta = fabsf(dot(isect.shadingNormal, bsdfSample.wiWorld));
                    pend = lightTerm;
                }
            }
        }

        // the vertex's shadow ray leaves isect.point like the continuation ray: both are traced in the next pass
        if (alive && shadow.push) {
            pendingShadow = true;
            shadowDirection = shadow.direction;
            shadowTfar = shadow.tfar;
        }

        // ---- end of a sample: radianceLookup += color (src/sample_integrator.cpp:61-63; non-finite samples
        // dropped), then the unit's next sample or the next unit
        bool needUnit = false;
        SHADE_REGION(8, alive && finished);   // sample finished
        if (alive && finished) {
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

    if (COUNT) {
        atomicAdd(&p.stats[kStatTris], (unsigned long long)trisTested);
        atomicAdd(&p.stats[kStatClosest], (unsigned long long)closestRays);
        atomicAdd(&p.stats[kStatShadow], (unsigned long long)shadowRays);
    }
}

#include "path_wave.h"
#include "path_hybrid.h"
static_assert(sizeof(RenderParams) + sizeof(SmallTris) <= 4096, "the fused kernels' arguments must fit the 4 KB kernarg segment");

// Test hook behind pathed_hip_debug_small_candidates: for every ray pair (origin, continuation direction, shadow direction,
// shadow far bound) the candidate sets of the phase-1 forms and the set phase 2 accepts, one bit per ORIGINAL primitive id
// (the forms index the triangles in different orders).  n is padded to whole waves by the host (the matrix instructions of
// the experiments build need every lane).  Words per ray: pair-of-triangles VALU form (A, B), matrix-pipe form (A, B; zero in
// the product library), accepted by phase 2 (A, B), item form with parallelograms -- what k_path_small runs -- (A, B).
__global__ __launch_bounds__(kBlock) void k_debug_small_candidates(DScene scene, SmallTris smallTris, SmallTris smallItems, int nQuads, float kappaT,
                                                                   const float4 *itemTris, const float *mfmaTable, MfmaFrame frame,
                                                                   const float *rays, int n, unsigned long long *out)
{
#if PATHED_EXPERIMENTS
    __shared__ float mfmaRows[kMfmaTableFloats];
    for (int i = threadIdx.x; i < kMfmaTableFloats; i += kBlock) { mfmaRows[i] = mfmaTable[i]; }
    __syncthreads();
#endif
    const int index = blockIdx.x * kBlock + threadIdx.x;
    const float *r = rays + (size_t)10 * index;
    const V3 o = v3(r[0], r[1], r[2]), dA = v3(r[3], r[4], r[5]), dB = v3(r[6], r[7], r[8]);
    const float tfarB = r[9];
    const int nTris = scene.nTris;

    unsigned int aLow, aHigh, bLow, bHigh;
    smallCandidatesPair(smallTris.data, nTris, o, dA, dB, &aLow, &aHigh, &bLow, &bHigh, candidateNear(PATHED_TNEAR), candidateFar(tfarB));
    unsigned int iaLow, iaHigh, ibLow, ibHigh;
    smallCandidatesItems<true>(smallItems.data, nQuads, nTris, kappaT, o, dA, dB, &iaLow, &iaHigh, &ibLow, &ibHigh, candidateNear(PATHED_TNEAR), candidateFar(tfarB));
    unsigned int evenA = 0u, oddA = 0u, evenB = 0u, oddB = 0u;
#if PATHED_EXPERIMENTS
    mfmaCandidatesPair(mfmaRows, nTris, frame, true, o, dA, dB, candidateNear(PATHED_TNEAR), candidateFar(tfarB), &evenA, &oddA, &evenB, &oddB);
#endif

    unsigned long long valuA = 0, valuB = 0, mfmaA = 0, mfmaB = 0, acceptA = 0, acceptB = 0, itemsA = 0, itemsB = 0;
    for (int k = 0; k < nTris; k++) {
        const float4 t0 = scene.leafTris[3 * k + 0], t1 = scene.leafTris[3 * k + 1], t2 = scene.leafTris[3 * k + 2];
        const unsigned long long bit = 1ull << (floatAsInt(t0.w) & 63);
        if (((k < 32 ? aLow : aHigh) >> (31 - (k & 31))) & 1u) { valuA |= bit; }
        if (((k < 32 ? bLow : bHigh) >> (31 - (k & 31))) & 1u) { valuB |= bit; }
        if ((((k & 1) ? oddA : evenA) >> (k >> 1)) & 1u) { mfmaA |= bit; }
        if ((((k & 1) ? oddB : evenB) >> (k >> 1)) & 1u) { mfmaB |= bit; }
        float t, u, v;
        if (intersectTriangle(o, dA, v3(t0.x, t0.y, t0.z), v3(t1.x, t1.y, t1.z), v3(t2.x, t2.y, t2.z), &t, &u, &v) && t > PATHED_TNEAR && t <= PATHED_TFAR) { acceptA |= bit; }
        if (intersectTriangle(o, dB, v3(t0.x, t0.y, t0.z), v3(t1.x, t1.y, t1.z), v3(t2.x, t2.y, t2.z), &t, &u, &v) && t > PATHED_TNEAR && t <= tfarB) { acceptB |= bit; }
        const unsigned long long itemBit = 1ull << (floatAsInt(itemTris[3 * k].w) & 63);
        if (((k < 32 ? iaLow : iaHigh) >> (31 - (k & 31))) & 1u) { itemsA |= itemBit; }
        if (((k < 32 ? ibLow : ibHigh) >> (31 - (k & 31))) & 1u) { itemsB |= itemBit; }
    }
    if (index < n) {
        unsigned long long *row = out + (size_t)8 * index;
        row[0] = valuA; row[1] = valuB; row[2] = mfmaA; row[3] = mfmaB; row[4] = acceptA; row[5] = acceptB; row[6] = itemsA; row[7] = itemsB;
    }
}

// ------------------------------------------------------------------------- volume path kernel
// volumeQuery for scenes of <= kBruteForceMaxTris triangles: the all-triangles intersector (smallCandidates on the
// kernarg pair records + a resolve loop) with volumeAccept's rule instead of a per-lane walk of a tiny tree.  Inlined at
// its six call sites: as a real function it cost 40-60 % (call frames in scratch, records through LDS instead of scalar
// loads).  Same hits and events as volumeQuery: acceptance and the two nearest events do not depend on the order
// primitives are met in; an occlusion query that is decided returns no events anybody reads.
// QUADS: the records are the item records of small_items.h (parallelograms first) and c.geometry.tris their item-ordered copy
template <bool QUADS = false, typename MaterialTable>
__device__ __forceinline__ bool volumeQuerySmall(const VolumeContext<MaterialTable> &c, const f2 *pairRecords, int smallQuads, float smallKappaT,
                                                 int mode, V3 o, V3 d, float tfar, RayHit *hit, VolumeEvents *eventsOut)
{
    LaneRay ray;
    const bool anyHit = mode == kQueryVolumeOccluded;
    laneRayInit(ray, o, d, PATHED_TNEAR, tfar, anyHit);
    VolumeEvents events;
    eventsClear(events);
    unsigned int low = 0u, high = 0u;
    if (QUADS) {
        // the single query as the pair form's second ray (the one that is held to a far bound); the first ray's words are dropped
        unsigned int unusedLow = 0u, unusedHigh = 0u;
        smallCandidatesItems<true>(pairRecords, smallQuads, c.geometry.nTris, smallKappaT, o, d, d, &unusedLow, &unusedHigh, &low, &high,
                                   candidateNear(PATHED_TNEAR), anyHit ? candidateFar(tfar) : 3e38f);
    } else {
        smallCandidates<true>(pairRecords, c.geometry.nTris, o, d, &low, &high, candidateNear(PATHED_TNEAR), anyHit ? candidateFar(tfar) : 3e38f);
    }
    bool decided = false;
    while (__ballot((low | high) != 0u) != 0ull) {
        if ((low | high) != 0u) {
            int k;  // triangle k of a word is bit 31 - (k & 31): take the highest set bit first
            if (low != 0u) { k = __clz((int)low); low &= ~(0x80000000u >> k); }
            else { k = __clz((int)high); high &= ~(0x80000000u >> k); k += 32; }
            const float4 t0 = c.geometry.tris[3 * k + 0];
            const float4 t1 = c.geometry.tris[3 * k + 1];
            const float4 t2 = c.geometry.tris[3 * k + 2];
            float t, u, v;
            if (intersectTriangle(ray.o, ray.d, v3(t0.x, t0.y, t0.z), v3(t1.x, t1.y, t1.z), v3(t2.x, t2.y, t2.z), &t, &u, &v)) {
                if (volumeAccept(c, mode, ray, events, t, u, v, floatAsInt(t0.w))) { decided = true; low = 0u; high = 0u; }
            }
        }
    }
    if (!decided) {
        for (int i = 0; i < c.geometry.nSpheres; i++) {
            if (volumeSphere(c, mode, ray, events, i)) { break; }
        }
    }
    if (mode == kQueryVolumeClosest) { eventsClip(events, ray.best); }
    if (eventsOut) { *eventsOut = events; }
    if (anyHit) { return ray.occluded; }
    hit->t = ray.best;
    hit->u = ray.bestU;
    hit->v = ray.bestV;
    hit->prim = ray.bestPrim;
    return ray.bestPrim >= 0;
}


// The two queries a vertex makes along rays that leave the SAME point, in one pass over the pair records
// (smallCandidatesPair: 12 + 2 x 21 packed operations per two triangles instead of 2 x 33): the closest-hit query of the
// vertex's BSDF sample -- `modeA` per lane: kQueryVolumeClosest where direct lighting wants it, kQueryRegular where only the
// path's next segment does -- and the occlusion query of its light sample.  Lanes pass wantA / wantB = false for a ray
// they do not have.  Candidates are resolved as in volumeQuerySmall, ray by ray: same hits, same events.
template <bool QUADS = false, typename MaterialTable>
__device__ __forceinline__ void volumeQueryPairSmall(const VolumeContext<MaterialTable> &c, const f2 *pairRecords, int smallQuads, float smallKappaT, V3 o,
                                                     bool wantA, int modeA, V3 dA, RayHit *hitA, bool *foundA, VolumeEvents *eventsA,
                                                     bool wantB, V3 dB, float tfarB, bool *occludedB, VolumeEvents *eventsB)
{
    unsigned int lowA = 0u, highA = 0u, lowB = 0u, highB = 0u;
    if (QUADS) {
        smallCandidatesItems<true>(pairRecords, smallQuads, c.geometry.nTris, smallKappaT, o, dA, dB, &lowA, &highA, &lowB, &highB,
                                   candidateNear(PATHED_TNEAR), candidateFar(tfarB));
    } else {
        smallCandidatesPair(pairRecords, c.geometry.nTris, o, dA, dB, &lowA, &highA, &lowB, &highB, candidateNear(PATHED_TNEAR), candidateFar(tfarB));
    }
    if (!wantA) { lowA = 0u; highA = 0u; }
    if (!wantB) { lowB = 0u; highB = 0u; }
    auto resolve = [&](int mode, LaneRay &ray, VolumeEvents &events, unsigned int low, unsigned int high, bool want) {
        bool decided = false;
        while (__ballot((low | high) != 0u) != 0ull) {
            if ((low | high) != 0u) {
                int k;  // triangle k of a word is bit 31 - (k & 31): take the highest set bit first
                if (low != 0u) { k = __clz((int)low); low &= ~(0x80000000u >> k); }
                else { k = __clz((int)high); high &= ~(0x80000000u >> k); k += 32; }
                const float4 t0 = c.geometry.tris[3 * k + 0];
                const float4 t1 = c.geometry.tris[3 * k + 1];
                const float4 t2 = c.geometry.tris[3 * k + 2];
                float t, u, v;
                if (intersectTriangle(ray.o, ray.d, v3(t0.x, t0.y, t0.z), v3(t1.x, t1.y, t1.z), v3(t2.x, t2.y, t2.z), &t, &u, &v)) {
                    if (volumeAccept(c, mode, ray, events, t, u, v, floatAsInt(t0.w))) { decided = true; low = 0u; high = 0u; }
                }
            }
        }
        if (want && !decided) {
            for (int i = 0; i < c.geometry.nSpheres; i++) {
                if (volumeSphere(c, mode, ray, events, i)) { break; }
            }
        }
    };
    {
        LaneRay ray;
        laneRayInit(ray, o, dA, PATHED_TNEAR, PATHED_TFAR, false);
        VolumeEvents events;
        eventsClear(events);
        resolve(modeA, ray, events, lowA, highA, wantA);
        if (modeA == kQueryVolumeClosest) { eventsClip(events, ray.best); }
        *eventsA = events;
        hitA->t = ray.best; hitA->u = ray.bestU; hitA->v = ray.bestV; hitA->prim = ray.bestPrim;
        *foundA = ray.bestPrim >= 0;
    }
    {
        LaneRay ray;
        laneRayInit(ray, o, dB, PATHED_TNEAR, tfarB, true);
        VolumeEvents events;
        eventsClear(events);
        resolve(kQueryVolumeOccluded, ray, events, lowB, highB, wantB);
        *eventsB = events;
        *occludedB = ray.occluded;
    }
}

// k_path_volume: SampleIntegrator::samplePixel + VolumePathTracer::L (see volume.h), one path per lane, persistent waves,
// work units as in k_path_small.  Arithmetic on a path's values follows the reference statement by statement; on a
// scene without media the result is PathTracer's, bit for bit (the two share their direct-lighting arithmetic; GPU test).
#ifndef PATHED_VOLUME_WAVES
#define PATHED_VOLUME_WAVES 4   // 128 registers per lane + scratch; 3 / 4 / 5 waves: 787 / 845 / 799 (Cornell), 732 / 787 / 707 (cornell-medium), 368 / 388 / 382 (teapot) Msamples/s; uncapped the kernel takes 220-260 registers and runs one or two waves; with one pass per vertex (round 3) 3 / 4 waves: 1 139 / 1 225 (Cornell), 901 / 985 (cornell-medium)
#endif
template <bool LDS_MATERIALS, int STACK, bool SMALL, typename TRAITS = TraitsAll, bool QUADS = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PATHED_VOLUME_WAVES, PATHED_VOLUME_WAVES))) void k_path_volume(RenderParams p, SmallTris smallTris)
{
    extern __shared__ float4 ldsRaw[];
    // LDS: [STACK + 1][kBlock] traversal stack rows, then (LDS_MATERIALS) the material table
    MaterialAccess<LDS_MATERIALS> materials;
    if (LDS_MATERIALS) {
        float4 *table = ldsRaw + ((STACK + 1) * kBlock) / 4;
        const int words = p.scene.nMaterials * (int)(sizeof(DMaterial) / 4);
        const int *source = reinterpret_cast<const int *>(p.scene.materials);
        int *target = reinterpret_cast<int *>(table);
        for (int i = threadIdx.x; i < words; i += kBlock) { target[i] = source[i]; }
        materials.table = reinterpret_cast<const DMaterial *>(table);
    } else {
        materials.table = p.scene.materials;
    }
    if (LDS_MATERIALS) { __syncthreads(); }

    const DScene &scene = p.scene;
    VolumeContext<MaterialAccess<LDS_MATERIALS>> context;
    context.geometry.nodes = scene.nodes;
    context.geometry.tris = scene.leafTris;
    context.geometry.nNodes = scene.nNodes;
    context.geometry.nTris = scene.nTris;
    context.geometry.spheres = scene.spheres;
    context.geometry.nSpheres = scene.nLinearSpheres;
    context.stack.lds = reinterpret_cast<int *>(ldsRaw) + threadIdx.x;
    context.stack.overflowStride = (size_t)gridDim.x * kBlock;
    context.stack.overflow = p.stackOverflow + ((size_t)blockIdx.x * kBlock + threadIdx.x);
    context.maxStack = p.maxStack;
    context.scene = &scene;
    context.materials = materials;
    context.primMedium = scene.primMedium;
    context.media = scene.media;

    const int lane = threadIdx.x & 63;
    const unsigned int waveId = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const uint64_t seed = ((uint64_t)p.seedHi << 32) | p.seedLo;

    // ---- work units, as in k_path_small
    unsigned int queue = waveId % (unsigned int)p.nQueues, queuesTried = 0;
    unsigned int reservedNext = 0, reservedEnd = 0;
    auto takeUnits = [&](bool want) -> unsigned int {
        unsigned int mine = 0xFFFFFFFFu;
        unsigned long long wanting = __ballot(want);
        while (wanting != 0ull) {
            if (reservedNext == reservedEnd) {
                if (queuesTried >= (unsigned int)p.nQueues) { break; }
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

    // one ray query: the all-triangles intersector (SMALL) or the per-lane walk of the 4-wide tree
    auto query = [&](int mode, V3 origin, V3 direction, float tfar, RayHit *hit, VolumeEvents *events) -> bool {
        if constexpr (SMALL) { return volumeQuerySmall<QUADS>(context, (const f2 *)smallTris.data, p.smallQuads, p.smallKappaT, mode, origin, direction, tfar, hit, events); }
        else { return volumeQuery<STACK>(context, mode, origin, direction, tfar, hit, events); }
    };

    // The reference traces the ray (vertex, BSDF sample) twice: directSampleBSDF asks for the nearest NON-container surface
    // (testVolumetricIntersect) and the path's next segment for the nearest surface of any kind (testIntersect).  One
    // traversal answers both: the volumetric query also records the nearest container it skipped (VolumeEvents).
    RayHit segmentHit;
    segmentHit.t = 0.f; segmentHit.u = 0.f; segmentHit.v = 0.f; segmentHit.prim = -1;
    bool segmentFound = false, segmentKnown = false;

    // What a vertex asks of the scene, all along rays that leave isect.point: DirectLightingHelper::Ld
    // (src/direct_lighting_helper.cpp:37-187; `direct`: the bounce counts) -- the occlusion query of its light sample and the
    // volumetric closest-hit query of its BSDF sample -- and the path's next segment (`needSegment`: the loop goes on), which
    // is that same BSDF-sample ray under Scene::testIntersect.  ONE pass answers all of it: the closest-hit query runs for
    // every lane, per lane in the mode its vertex needs (a vertex without direct lighting -- a container's surface, an
    // emitter, a bounce outside the window -- asks the regular question; the others get the regular answer from the
    // container the volumetric query skipped), paired with the light sample's occlusion ray (volumeQueryPairSmall).
    // The reference's statements in the reference's order on every value; only the queries moved.
    auto vertexLighting = [&](const Isect &isect, int medium, const DMaterial &material, const BSDFSample &bsdfSample, Rng &random,
                              bool direct, bool needSegment) -> Rgb {
        SHADE_REGION(5, direct);   // (profile builds) direct lighting at a vertex
        const bool lit = direct && material.type != PATHED_MAT_PASSTHROUGH && isBlack(matEmit(material));
        // directSampleLights, :74-134, up to its occlusion query
        bool wantShadow = false;
        DLight light;
        light.kind = 0; light.index = 0;
        SurfaceSample surfaceSample;
        surfaceSample.point = isect.point; surfaceSample.normal = v3(0.f, 0.f, 0.f); surfaceSample.invPDF = 1.f; surfaceSample.solidAngle = 1;
        int lightMaterial = 0;
        float invPDF = 1.f, lightDistance = 0.f;
        V3 lightDirection = v3(0.f, 0.f, 0.f), wiWorld = bsdfSample.wiWorld;
        if (lit && !volumeIsDelta(material) && scene.nLights != 0) {
            const int lightCount = scene.nLights;
            int lightIndex = (int)floorf(random.next() * lightCount);
            lightIndex = imin(lightIndex, lightCount - 1);
            light = scene.lights[lightIndex];
            if (TRAITS::triangleLights && light.kind == 0) {
                const TriShade tri = loadTriCorners(scene, light.index);
                surfaceSample = triangleSample(tri.p0, tri.p1, tri.p2, random);
                lightMaterial = tri.material;
            } else if (TRAITS::spheres && light.kind == 1) {
                const DSphere sphere = scene.spheres[light.index];
                surfaceSample = sphereSample<TRAITS::pairedTrig>(v3(sphere.centerSample[0], sphere.centerSample[1], sphere.centerSample[2]), sphere.radius, isect.point, random);
                lightMaterial = sphere.material;
            } else if (TRAITS::env) {
                surfaceSample = envSample<TRAITS::pairedTrig>(scene.env, isect.point, random);
            } else {   // not reached: a light of a kind the instantiation's scene set does not contain
                surfaceSample.point = isect.point; surfaceSample.normal = v3(0.f, 0.f, 0.f); surfaceSample.invPDF = 1.f; surfaceSample.solidAngle = 1;
            }
            const float lightChoicePDF = 1.f / lightCount;
            invPDF = surfaceSample.invPDF * (1.f / lightChoicePDF);
            lightDirection = surfaceSample.point - isect.point;
            wiWorld = normalized(lightDirection);
            if (!(dot(surfaceSample.normal, wiWorld) >= 0.f)) {
                lightDistance = length(lightDirection);
                wantShadow = true;
            }
        }

        // the queries
        const bool wantClosest = lit || needSegment;
        const int closestMode = lit ? kQueryVolumeClosest : kQueryRegular;
        RayHit bounceHit;
        bounceHit.t = 0.f; bounceHit.u = 0.f; bounceHit.v = 0.f; bounceHit.prim = -1;
        bool found = false, occluded = false;
        VolumeEvents skipped, events;
        eventsClear(skipped);
        eventsClear(events);
        // directSampleLights, :74-134, from its occlusion query on
        auto finishLight = [&]() -> Rgb {
            if (!wantShadow || occluded) { return rgb(0.f); }
            const Rgb transmittance = rayTransmission(context.media, isect.point, wiWorld, events, medium);
            float pdf;
            if (surfaceSample.solidAngle) {
                pdf = 1.f / invPDF;
            } else {
                const V3 lightWoForPdf = -normalized(lightDirection);
                const float distance2 = lightDistance * lightDistance;
                const float projectedArea = smax(0.f, dot(surfaceSample.normal, lightWoForPdf));
                pdf = (1.f / invPDF) * distance2 / projectedArea;
            }
            float brdfPDF;
            const Rgb f = materialF<TRAITS>(material, isect, wiWorld, &brdfPDF);
            const float lightWeight = (1 * pdf) / (1 * pdf + 1 * brdfPDF);
            const V3 lightWo = -normalized(lightDirection);
            Rgb emitted;
            if (TRAITS::env && light.kind == 2) { emitted = envEmit(scene.env, lightWo); }
            else { emitted = matEmit(materials[lightMaterial]); }
            return emitted
                * transmittance
                * lightWeight
                * f
                * fabsf(dot(isect.shadingNormal, wiWorld))
                / pdf;
        };
        Rgb lightContribution = rgb(0.f);
        if constexpr (SMALL) {
            SHADE_REGION(7, wantClosest);   // the closest-hit query of the BSDF sample / next segment
            SHADE_REGION(6, wantShadow);    // the light sample's occlusion query (same pass)
            if (__ballot(wantShadow) != 0ull) {
                volumeQueryPairSmall<QUADS>(context, (const f2 *)smallTris.data, p.smallQuads, p.smallKappaT, isect.point,
                                     wantClosest, closestMode, bsdfSample.wiWorld, &bounceHit, &found, &skipped,
                                     wantShadow, wiWorld, lightDistance - 1e-3f, &occluded, &events);
            } else if (__ballot(wantClosest) != 0ull) {
                // no lane of the wave has a light sample to test (delta materials, containers): the one-ray pass
                if (wantClosest) { found = query(closestMode, isect.point, bsdfSample.wiWorld, PATHED_TFAR, &bounceHit, &skipped); }
            }
            lightContribution = finishLight();
        } else {
            // a per-lane walk of the tree per query: the light sample is finished before the second walk starts
            if (wantShadow) {
                SHADE_REGION(6, true);
                RayHit unused;
                occluded = query(kQueryVolumeOccluded, isect.point, wiWorld, lightDistance - 1e-3f, &unused, &events);
            }
            lightContribution = finishLight();
            if (wantClosest) {
                SHADE_REGION(7, true);
                found = query(closestMode, isect.point, bsdfSample.wiWorld, PATHED_TFAR, &bounceHit, &skipped);
            }
        }

        Rgb result = rgb(0.f);
        result = result + lightContribution;

        // the path's next segment: the nearest surface of any kind along the BSDF sample
        segmentHit = bounceHit;
        segmentFound = found;
        if (lit && skipped.containerPrim >= 0) {
            const bool nearer = !found || skipped.containerT < bounceHit.t
                || (skipped.containerT == bounceHit.t && skipped.containerPrim < bounceHit.prim);
            if (nearer) {
                segmentHit.t = skipped.containerT; segmentHit.u = skipped.containerU; segmentHit.v = skipped.containerV;
                segmentHit.prim = skipped.containerPrim;
                segmentFound = true;
            }
        }
        segmentKnown = wantClosest;

        // directSampleBSDF, :136-187: the query skips containers; no transmittance is applied (as in the reference)
        Rgb bsdfTerm = rgb(0.f);
        if (lit) {
            if (found) {
                const Isect bounce = makeIsect<TRAITS>(scene, isect.point, bsdfSample.wiWorld,
                                               make_float4(bounceHit.t, bounceHit.u, bounceHit.v, intAsFloat(bounceHit.prim)));
                const Rgb emit = matEmit(materials[bounce.material]);
                if (!isBlack(emit) && dot(bounce.wo, bounce.shadingNormal) >= 0.f) {
                    const float lightPDF = lightsPDF<TRAITS>(scene, isect.point, bounce);
                    const float brdfWeight = volumeIsDelta(material) ? 1.f : (1 * bsdfSample.pdf) / (1 * bsdfSample.pdf + 1 * lightPDF);
                    bsdfTerm = emit * brdfWeight * bsdfSample.throughput * fabsf(dot(isect.shadingNormal, bsdfSample.wiWorld)) / bsdfSample.pdf;
                }
            } else {
                const Rgb environmentLight = environmentL<TRAITS>(scene, bsdfSample.wiWorld);
                if (!isBlack(environmentLight)) {
                    const float lightPDF = envEmitPDF(scene.env, bsdfSample.wiWorld) / scene.nLights;
                    const float brdfWeight = volumeIsDelta(material) ? 1.f : (1 * bsdfSample.pdf) / (1 * bsdfSample.pdf + 1 * lightPDF);
                    bsdfTerm = environmentLight * brdfWeight * bsdfSample.throughput * fabsf(dot(isect.shadingNormal, bsdfSample.wiWorld)) / bsdfSample.pdf;
                }
            }
        }
        result = result + bsdfTerm;
        return result;
    };

    // VolumePathTracer::scatter -> HomogeneousMedium::integrate -> VolumeHelper::directSampleLights
    // (src/volume_path_tracer.cpp:114-131, src/homogeneous_medium.cpp:36-66, src/volume_helper.cpp:12-69)
    auto scatter = [&](int medium, V3 entry, V3 exitPoint, Rng &random) -> Rgb {
        if (medium < 0) { return rgb(0.f); }
        const float sigmaT = context.media[medium].sigmaT[0];
        const V3 travel = exitPoint - entry;
        const float distance = length(travel);
        const float xi = random.next();
        const float sampleT = -logf(1 - xi) / sigmaT;
        if (sampleT >= distance) { return rgb(0.f); }
        const V3 samplePoint = entry + normalized(travel) * sampleT;
        if (scene.nLights == 0) { return rgb(0.f); }
        const int lightCount = scene.nLights;
        int lightIndex = (int)floorf(random.next() * lightCount);
        lightIndex = imin(lightIndex, lightCount - 1);
        const DLight light = scene.lights[lightIndex];
        SurfaceSample surfaceSample;
        int lightMaterial = 0;
        if (TRAITS::triangleLights && light.kind == 0) {
            const TriShade tri = loadTriCorners(scene, light.index);
            surfaceSample = triangleSample(tri.p0, tri.p1, tri.p2, random);
            lightMaterial = tri.material;
        } else if (TRAITS::spheres && light.kind == 1) {
            const DSphere sphere = scene.spheres[light.index];
            surfaceSample = sphereSample<TRAITS::pairedTrig>(v3(sphere.centerSample[0], sphere.centerSample[1], sphere.centerSample[2]), sphere.radius, samplePoint, random);
            lightMaterial = sphere.material;
        } else if (TRAITS::env) {
            surfaceSample = envSample<TRAITS::pairedTrig>(scene.env, samplePoint, random);
        } else {   // not reached
            surfaceSample.point = samplePoint; surfaceSample.normal = v3(0.f, 0.f, 0.f); surfaceSample.invPDF = 1.f; surfaceSample.solidAngle = 1;
        }
        const float lightChoicePDF = 1.f / lightCount;
        const float invPDF = surfaceSample.invPDF * (1.f / lightChoicePDF);
        const V3 sampleDirection = surfaceSample.point - samplePoint;
        const V3 wiWorld = normalized(sampleDirection);
        if (dot(surfaceSample.normal, wiWorld) >= 0.f) { return rgb(0.f); }
        const float lightDistance = length(sampleDirection);
        VolumeEvents events;
        RayHit unused;
        SHADE_REGION(4, true);   // medium event: its light sample's occlusion query
        if (query(kQueryVolumeOccluded, samplePoint, wiWorld, lightDistance - 1e-3f, &unused, &events)) { return rgb(0.f); }
        float pdf;
        if (surfaceSample.solidAngle) {
            pdf = 1.f / invPDF;
        } else {
            const V3 lightWoForPdf = -normalized(sampleDirection);
            const float distance2 = lightDistance * lightDistance;
            const float projectedArea = smax(0.f, dot(surfaceSample.normal, lightWoForPdf));
            pdf = (1.f / invPDF) * distance2 / projectedArea;
        }
        const V3 lightWo = -normalized(sampleDirection);
        Rgb shadowTransmittance = rgb(0.f);
        if (events.count == 1) { shadowTransmittance = mediumTransmittance(context.media[medium], samplePoint, samplePoint + wiWorld * events.t0); }
        else if (events.count >= 2) { shadowTransmittance = mediumTransmittance(context.media[medium], samplePoint + wiWorld * events.t0, samplePoint + wiWorld * events.t1); }
        Rgb emitted;
        if (TRAITS::env && light.kind == 2) { emitted = envEmit(scene.env, lightWo); }
        else { emitted = matEmit(materials[lightMaterial]); }
        const float fourPi = (float)(4.f * 3.14159265358979323846);   // `4.f * M_PI` is a double, Color::operator/ takes a float
        return emitted * shadowTransmittance * 1.f / fourPi / pdf;
    };

    // one camera sample: SampleIntegrator::samplePixel, src/sample_integrator.cpp:10-78
    auto samplePixel = [&](uint32_t pixel, uint32_t sample) -> Rgb {
        segmentKnown = false;
        Rng random;
        makeKey(seed, pixel, sample, &random.k0, &random.k1);
        random.dimension = 0;
        const int width = scene.camera.resX;
        const int row = (int)fastDivide((unsigned int)pixel, p.divWidth);   // pixel / width, exactly
        const int col = (int)pixel - row * width;
        const float jitterX = random.next() - 0.5f;
        const float jitterY = random.next() - 0.5f;
        V3 rayOrigin, rayDirection;
        cameraRay(scene.camera, row + jitterY, col + jitterX, &rayOrigin, &rayDirection);

        Rgb color = rgb(0.f);
        RayHit hit;
        SHADE_REGION(1, true);   // camera ray query
        if (!query(kQueryRegular, rayOrigin, rayDirection, PATHED_TFAR, &hit, nullptr)) {
            return color + environmentL<TRAITS>(scene, rayDirection);
        }
        Isect last = makeIsect<TRAITS>(scene, rayOrigin, rayDirection, make_float4(hit.t, hit.u, hit.v, intAsFloat(hit.prim)));
        if (checkCounts(p.startBounce, p.lastBounce, 0)) {
            const DMaterial &first = materials[last.material];
            const bool backside = dot(last.normal, last.wo) < 0.f;
            if (!isBlack(matEmit(first)) && !backside) { color = color + matEmit(first); }
            if (first.type == PATHED_MAT_PASSTHROUGH) {
                // what is seen through the container, src/sample_integrator.cpp:35-51
                VolumeEvents events;
                RayHit through;
                SHADE_REGION(8, true);   // what is seen through a container
                const bool found = query(kQueryVolumeClosest, rayOrigin, rayDirection, PATHED_TFAR, &through, &events);
                const Rgb transmittance = rayTransmission(context.media, rayOrigin, rayDirection, events, -1);
                if (found) { color = color + matEmit(materials[primMaterial(context, through.prim)]) * transmittance; }
                else { color = color + environmentL<TRAITS>(scene, rayDirection) * transmittance; }
            }
        }

        // ---- VolumePathTracer::L, src/volume_path_tracer.cpp:14-99
        int medium = -1;
        random.dimension = vertexBase(1);
        prepareLobes<TRAITS>(materials[last.material], last);
        BSDFSample bsdfSample = volumeMaterialSample<TRAITS>(materials[last.material], last, random);
        Rgb result = rgb(0.f);
        {
            const bool direct = checkCounts(p.startBounce, p.lastBounce, 1);
            random.dimension = vertexBase(1) + 3;
            const Rgb Ld = vertexLighting(last, medium, materials[last.material], bsdfSample, random, direct, !checkDone(p.lastBounce, 2));
            if (direct) { result = Ld; }
        }
        Rgb modulation = rgb(1.f);
        for (int bounce = 2; !checkDone(p.lastBounce, bounce); bounce++) {
            SHADE_REGION(2, true);   // bounce-loop iterations
            // refraction: the medium changes (:43-51)
            if (dot(last.wo, bsdfSample.wiWorld) < 0.f) {
                if (dot(last.normal, bsdfSample.wiWorld) < 0.f) { medium = context.primMedium[last.prim]; }
                else { medium = -1; }
            }
            if (segmentKnown) {
                // the previous vertex's pass has answered it (vertexLighting)
                segmentKnown = false;
                if (!segmentFound) { break; }
                hit = segmentHit;
            } else {
                break;   // not reached: every vertex whose path goes on has asked for its segment (needSegment)
            }
            Isect next = makeIsect<TRAITS>(scene, last.point, bsdfSample.wiWorld, make_float4(hit.t, hit.u, hit.v, intAsFloat(hit.prim)));
            const float invPDF = 1.f / bsdfSample.pdf;
            const float cosTheta = fabsf(dot(last.shadingNormal, bsdfSample.wiWorld));
            modulation = modulation * (bsdfSample.throughput * cosTheta * invPDF);

            random.dimension = mediumBase(bounce);
            const Rgb Ls = scatter(medium, last.point, next.point, random);
            result = result + Ls * modulation;
            if (medium >= 0) { modulation = modulation * mediumTransmittance(context.media[medium], last.point, next.point); }
            else { modulation = modulation * rgb(1.f); }
            if (isBlack(modulation)) { break; }

            random.dimension = vertexBase(bounce);
            prepareLobes<TRAITS>(materials[next.material], next);
            bsdfSample = volumeMaterialSample<TRAITS>(materials[next.material], next, random);
            last = next;
            {
                const bool direct = checkCounts(p.startBounce, p.lastBounce, bounce);
                random.dimension = vertexBase(bounce) + 3;
                const Rgb Ld = vertexLighting(last, medium, materials[last.material], bsdfSample, random, direct, !checkDone(p.lastBounce, bounce + 1));
                if (direct) { result = result + Ld * modulation; }
            }
        }
        return color + result;
    };

    unsigned long long samplesDone = 0;
    while (true) {
        const unsigned int unit = takeUnits(true);
        if (__ballot(unit != 0xFFFFFFFFu) == 0ull) { break; }
        if (unit != 0xFFFFFFFFu) {
            uint32_t pixel, first, end;
            unitSamples(p, unit, &pixel, &first, &end);
            float4 partial = make_float4(0.f, 0.f, 0.f, 0.f);
            for (uint32_t sample = first; sample < end; sample++) {
                SHADE_REGION(0, true);   // samples
                const Rgb color = samplePixel(pixel, sample);
                // radianceLookup += color, src/sample_integrator.cpp:61-63; non-finite samples dropped
                if (isfinite(color.r) && isfinite(color.g) && isfinite(color.b)) {
                    partial.x += color.r;
                    partial.y += color.g;
                    partial.z += color.b;
                } else {
                    atomicAdd(&p.stats[kStatDropped], 1ull);
                }
                samplesDone++;
            }
            p.state.chunkBuf[partialIndex(p, unit)] = partial;
        }
    }
    (void)samplesDone;
}

// ------------------------------------------------------------------------- scene set-up
// The per-triangle shading records (device_scene.h: kTriShadeQuads) gathered on the device from the
// vertex arrays: pure copies, so the table is the one the host loop used to build, without writing
// and uploading 128 bytes per triangle from one host thread.
__global__ __launch_bounds__(kBlock) void k_build_tri_shade(
    const float *positions, const float *normals, const float *uvs, const uint32_t *indices, const int *triMaterial,
    uint32_t nTriangles, float4 *triShade, float4 *triCompact)
{
    // EIGHT lanes per triangle, one per float4 of its record: a wave's store is 1 KiB of contiguous bytes (one thread per
    // triangle wrote 16 bytes at a 128-byte stride per instruction and ran at < 1 TB/s: 1.46 ms for 5.2 M triangles, the
    // larger half of a refit; now a third of that)
    const size_t thread = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t i = thread >> 3;
    const int q = (int)(thread & 7);
    if (i >= nTriangles) { return; }
    const int corner = q < 3 ? q : q < 6 ? q - 3 : 0;   // the vertex this lane's xyz comes from
    const size_t vertex = indices[3 * i + corner];
    const float *source = (q < 3 ? positions : normals) + 3 * vertex;
    float x = 0.f, y = 0.f, z = 0.f, w = 0.f;
    if (q < 6) { x = source[0]; y = source[1]; z = source[2]; }
    // .w: q0 material, q1 uv0.u, q2 uv0.v, q3 uv1.u, q4 uv1.v, q5 uv2.u; q6.x = uv2.v
    if (q == 0) { w = intAsFloat(triMaterial[i]); }
    else if (q <= 6) {
        const int uvCorner = (q - 1) >> 1, component = (q - 1) & 1;
        const float value = uvs[2 * (size_t)indices[3 * i + uvCorner] + component];
        if (q == 6) { x = value; } else { w = value; }
    }
    triShade[(size_t)kTriShadeQuads * i + q] = make_float4(x, y, z, w);
    // the 16-byte record (makeIsect): geometric normal + material, read for the triangles of the scene's plain ranges.
    // Lanes 0..2 of the group hold the corners: lane 0 collects them.
    const int lane = threadIdx.x & 63, base = lane & ~7;
    const float x1 = __shfl(x, base + 1), y1 = __shfl(y, base + 1), z1 = __shfl(z, base + 1);
    const float x2 = __shfl(x, base + 2), y2 = __shfl(y, base + 2), z2 = __shfl(z, base + 2);
    if (q == 0) {
        const V3 normal = triangleNormal(v3(x, y, z), v3(x1, y1, z1), v3(x2, y2, z2));
        triCompact[i] = make_float4(normal.x, normal.y, normal.z, w);
    }
}

// ------------------------------------------------------------------------- refit
// New vertex positions over an UNCHANGED topology: what rtcCommitScene (reference src/scene.cpp:39) does again for an animated
// mesh, without rebuilding.  One thread per 4-wide node, level-synchronous like the builders' own fitting pass (lbvh.hip:
// k_lbvh_fit_pass): a pass gives every node whose inner children had their bounds BEFORE the pass its four child boxes --
// leaf children straight from the moved vertices (and rewrites their (v0, prim) (e1) (e2) records), inner children from the
// UNPADDED bounds kept beside the nodes -- pads them exactly as the builders do (bvh_build.h padBox) and records its own
// unpadded bounds; double-buffered flags keep a node from being read in the pass that writes it, kernel boundaries make a
// pass visible to the next.  As many passes as the tree has 4-wide levels.  Boxes are the ones a fresh build over the same
// topology would store; hits do not depend on them anyway (the triangle test decides).
__global__ __launch_bounds__(kBlock) void k_refit_pass(
    float4 *nodes, int nNodes, float4 *leafTris, const float *positions, const uint32_t *indices, const DSphere *spheres,
    float4 *boundsLo, float4 *boundsHi, const unsigned char *readyIn, unsigned char *readyOut)
{
    const int n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= nNodes) { return; }
    if (readyIn[n]) { readyOut[n] = 1; return; }
    float4 *node = nodes + (size_t)8 * n;
    const float4 refWords = node[6];
    const int refs[4] = { floatAsInt(refWords.x), floatAsInt(refWords.y), floatAsInt(refWords.z), floatAsInt(refWords.w) };
    for (int k = 0; k < 4; k++) {
        if (refs[k] >= 0 && !readyIn[refs[k]]) { readyOut[n] = 0; return; }   // an inner child is not fitted yet
    }
    const float inf = __builtin_huge_valf();
    float lo[3][4], hi[3][4];
    float ownLo[3] = { inf, inf, inf }, ownHi[3] = { -inf, -inf, -inf };
    for (int k = 0; k < 4; k++) {
        const int ref = refs[k];
        float low[3] = { inf, inf, inf }, high[3] = { -inf, -inf, -inf };
        const bool present = ref != kEmptyChild;
        if (ref >= 0) {
            const float4 a = boundsLo[ref], b = boundsHi[ref];
            low[0] = a.x; low[1] = a.y; low[2] = a.z; high[0] = b.x; high[1] = b.y; high[2] = b.z;
        } else if (present) {
            const int code = -ref - 1;
            const int first = code >> 3, count = code & 7;
            if (count == 0) {
                // a sphere leaf (bvh_build.h: its bounds are padded once on their own)
                const DSphere sphere = spheres[first - 1];
                const float radius = fabsf(sphere.radius);
                for (int a = 0; a < 3; a++) {
                    const float reach = radius * 1.00001f + 1e-5f * fabsf(sphere.centerWorld[a]);
                    low[a] = sphere.centerWorld[a] - reach;
                    high[a] = sphere.centerWorld[a] + reach;
                }
            }
            for (int t = 0; t < count; t++) {
                float4 *record = leafTris + (size_t)3 * (first + t);
                const int prim = floatAsInt(record[0].w);
                const float *v0 = positions + 3 * (size_t)indices[3 * (size_t)prim + 0];
                const float *v1 = positions + 3 * (size_t)indices[3 * (size_t)prim + 1];
                const float *v2 = positions + 3 * (size_t)indices[3 * (size_t)prim + 2];
                record[0] = make_float4(v0[0], v0[1], v0[2], intAsFloat(prim));
                record[1] = make_float4(v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2], 0.f);
                record[2] = make_float4(v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2], 0.f);
                for (int a = 0; a < 3; a++) {
                    low[a] = fminf(low[a], fminf(v0[a], fminf(v1[a], v2[a])));
                    high[a] = fmaxf(high[a], fmaxf(v0[a], fmaxf(v1[a], v2[a])));
                }
            }
        }
        for (int a = 0; a < 3; a++) {
            const float pad = 1e-5f * fmaxf(1.f, fmaxf(fabsf(low[a]), fabsf(high[a])));
            lo[a][k] = present ? low[a] - pad : 0.f;
            hi[a][k] = present ? high[a] + pad : 0.f;
            if (present) { ownLo[a] = fminf(ownLo[a], low[a]); ownHi[a] = fmaxf(ownHi[a], high[a]); }
        }
    }
    for (int a = 0; a < 3; a++) {
        node[a] = make_float4(lo[a][0], lo[a][1], lo[a][2], lo[a][3]);
        node[3 + a] = make_float4(hi[a][0], hi[a][1], hi[a][2], hi[a][3]);
    }
    boundsLo[n] = make_float4(ownLo[0], ownLo[1], ownLo[2], 0.f);
    boundsHi[n] = make_float4(ownHi[0], ownHi[1], ownHi[2], 0.f);
    readyOut[n] = 1;
}

// ------------------------------------------------------------------------- bandwidth probe
// What this box's HBM actually delivers to a plain streaming kernel: the second denominator beside
// the 8 TB/s spec figure.  16 bytes per lane per access, grid-stride.
__global__ __launch_bounds__(kBlock) void k_stream_read(const float4 *source, size_t count, float *sink)
{
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride) {
        const float4 v = source[i];
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    // never true for the zero-filled probe buffer, but keeps the loads alive
    if (sum.x + sum.y + sum.z + sum.w == 12345.678f) { *sink = sum.x; }
}

__global__ __launch_bounds__(kBlock) void k_stream_copy(const float4 *source, float4 *target, size_t count)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride) { target[i] = source[i]; }
}

// ------------------------------------------------------------------------- VALU issue-rate probe
// Denominator of the "VALU issue" bound (pathed_hip_measure_valu).  Every lane keeps eight
// independent accumulator chains, so consecutive v_fma_f32 never wait for each other; inline asm
// keeps the compiler from folding or vectorising them.  MIXED adds one v_rcp_f32 + v_sqrt_f32 pair per
// six v_fma_f32 (a path tracer's normalisations; those two issue at a quarter of the FMA rate).
static const int kValuProbeUnroll = 48;   // VALU instructions per loop iteration
// MODE 0: v_fma_f32 with one VGPR source (the accumulator) and an SGPR for multiplier and addend: no VGPR bank
//         conflicts, nothing but issue limits the rate
// MODE 1: six such v_fma_f32 + one v_rcp_f32 + one v_sqrt_f32 per eight instructions
// MODE 2: v_fma_f32 with three VGPR sources (accumulator, multiplier, addend in registers)
// MODE 3: v_pk_fma_f32 (two fp32 FMAs per lane and instruction)
// MODE 4: v_mul_lo_u32 (the random stream's multiplies)
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_valu_probe(int iterations, float seed, float *sink)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    float a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float m = 0.999f, c = 0.001f;
    f2 p0 = { a0, a1 }, p1 = { a2, a3 }, p2 = { a4, a5 }, p3 = { a6, a7 }, p4 = { a1, a0 }, p5 = { a3, a2 }, p6 = { a5, a4 }, p7 = { a7, a6 };
    const f2 pm = { m, m }, pc = { c, c };
    unsigned int i0 = threadIdx.x + 1u, i1 = i0 + 1u, i2 = i0 + 2u, i3 = i0 + 3u, i4 = i0 + 4u, i5 = i0 + 5u, i6 = i0 + 6u, i7 = i0 + 7u;
    const unsigned int im = 0x846ca68bu;
    for (int i = 0; i < iterations; i++) {
        #pragma unroll
        for (int k = 0; k < kValuProbeUnroll / 8; k++) {
            if (MODE == 0) {
                asm volatile(
                    "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                    "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m));
            } else if (MODE == 1) {
                asm volatile(
                    "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n"
                    "v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n"
                    "v_rcp_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m));
            } else if (MODE == 2) {
                asm volatile(
                    "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                    "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (MODE == 3) {
                asm volatile(
                    "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                    "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                    : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));
            } else {
                asm volatile(
                    "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                    "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                    : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "s"(im));
            }
        }
    }
    float total = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    total += ((p0.x + p1.y) + (p2.x + p3.y)) + ((p4.x + p5.y) + (p6.x + p7.y));
    total += (float)(((i0 ^ i1) ^ (i2 ^ i3)) ^ ((i4 ^ i5) ^ (i6 ^ i7)));
    if (total == 12345.678f) { *sink = total; }   // keeps the chains alive
}

// dst[i] += src[i]: the fan-in of per-GPU radiance sums inside one process (pathed_hip_accum_add)
__global__ __launch_bounds__(kBlock) void k_accum_add(float *dst, const float *src, size_t count)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride) { dst[i] += src[i]; }
}

}  // namespace pathed
