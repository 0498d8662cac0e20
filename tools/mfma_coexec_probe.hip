// Do matrix instructions of one wave overlap with VALU instructions of the other waves on the same SIMD -- for the
// f32-input MFMA (v_mfma_f32_32x32x2_f32) as for the bf16 one (v_mfma_f32_32x32x16_bf16)?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_coexec_probe tools/mfma_coexec_probe.hip
// Every wave runs `iterations` of: VALU independent v_fma_f32 (8 chains) + MFMA matrix instructions (two accumulators).
// Grid: 256 CUs x 4 blocks x 256 threads (four waves per SIMD, what k_path_small runs at).  Prints ms per configuration and
// the cycles per iteration per wave-round they amount to at the measured clock; "sum" means no overlap, "max" full overlap.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));

// ROLES: 0 every wave runs both parts; 1 waves in odd slots of their SIMD run only the matrix part (twice), the others
// only the VALU part (twice): the same work per SIMD when the slots split evenly (counted in `census`);
// 2 every wave runs both, odd-slot waves start half an iteration late (VALU first)
template <int VALU, int MFMA_F32, int MFMA_BF16, int ROLES = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void probe(int iterations, float seed, float *sink, unsigned int *census = nullptr)
{
    unsigned int hwId;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwId));
    const bool odd = (hwId & 1u) != 0u;   // HW_ID[3:0] = wave slot within the SIMD
    if (census && (threadIdx.x & 63) == 0) { atomicAdd(&census[odd ? 1 : 0], 1u); }
    float c[8];
    for (int i = 0; i < 8; i++) { c[i] = seed + (float)i + (float)threadIdx.x * 1e-3f; }
    f16v acc0, acc1;
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const float a = seed * 0.5f, b = seed * 0.25f;
    bf8v pa, pb;
    for (int r = 0; r < 8; r++) { pa[r] = (__bf16)(seed * (float)r); pb[r] = (__bf16)(seed + (float)r); }
    const float m = 1.0001f, k = 1e-6f;
    const int valuRepeats = ROLES == 1 ? (odd ? 0 : 2) : 1;
    const int mfmaRepeats = ROLES == 1 ? (odd ? 2 : 0) : 1;
    if (ROLES == 2 && odd) {
#pragma unroll
        for (int v = 0; v < VALU / 8; v++) {
#pragma unroll
            for (int i = 0; i < 8; i++) { c[i] = __builtin_fmaf(c[i], m, k); }
        }
    }
    for (int it = 0; it < iterations; it++) {
        for (int repeat = 0; repeat < valuRepeats; repeat++) {
#pragma unroll
            for (int v = 0; v < VALU / 8; v++) {
#pragma unroll
                for (int i = 0; i < 8; i++) { c[i] = __builtin_fmaf(c[i], m, k); }
            }
        }
        for (int repeat = 0; repeat < mfmaRepeats; repeat++) {
#pragma unroll
            for (int q = 0; q < MFMA_F32 / 2; q++) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < MFMA_BF16 / 2; q++) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pb, pa, acc1, 0, 0, 0);
            }
        }
    }
    float total = 0.f;
    for (int i = 0; i < 8; i++) { total += c[i]; }
    for (int r = 0; r < 16; r++) { total += acc0[r] + acc1[r]; }
    if (total == 123.456f) { sink[0] = total; }
}

// the same work with the matrix instructions spread evenly through the wave's VALU stream (never two in a row):
// GROUPS x (VALU / GROUPS v_fma_f32, then one matrix instruction), order pinned by scheduling barriers
template <int VALU, int GROUPS, bool BF16, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void interleaved(int iterations, float seed, float *sink)
{
    float c[8];
    for (int i = 0; i < 8; i++) { c[i] = seed + (float)i + (float)threadIdx.x * 1e-3f; }
    f16v acc[4];
    for (int q = 0; q < 4; q++) { for (int r = 0; r < 16; r++) { acc[q][r] = 0.f; } }
    const float a = seed * 0.5f, b = seed * 0.25f;
    bf8v pa, pb;
    for (int r = 0; r < 8; r++) { pa[r] = (__bf16)(seed * (float)r); pb[r] = (__bf16)(seed + (float)r); }
    const float m = 1.0001f, k = 1e-6f;
    for (int it = 0; it < iterations; it++) {
#pragma unroll
        for (int g = 0; g < GROUPS; g++) {
#pragma unroll
            for (int v = 0; v < VALU / GROUPS / 8; v++) {
#pragma unroll
                for (int i = 0; i < 8; i++) { c[i] = __builtin_fmaf(c[i], m, k); }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (BF16) { acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc[g & 3], 0, 0, 0); }
            else { acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g & 3], 0, 0, 0); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float total = 0.f;
    for (int i = 0; i < 8; i++) { total += c[i]; }
    for (int q = 0; q < 4; q++) { for (int r = 0; r < 16; r++) { total += acc[q][r]; } }
    if (total == 123.456f) { sink[0] = total; }
}

template <int VALU, int GROUPS, bool BF16, int WAVES>
static double runInterleaved(const char *name, int iterations, float *sink)
{
    hipEvent_t start, stop;
    hipEventCreate(&start); hipEventCreate(&stop);
    const dim3 grid(256 * WAVES), block(256);
    hipLaunchKernelGGL((interleaved<VALU, GROUPS, BF16, WAVES>), grid, block, 0, nullptr, iterations / 10, 1.f, sink);
    hipDeviceSynchronize();
    double best = 1e30;
    for (int repeat = 0; repeat < 3; repeat++) {
        hipEventRecord(start, nullptr);
        hipLaunchKernelGGL((interleaved<VALU, GROUPS, BF16, WAVES>), grid, block, 0, nullptr, iterations, 1.f, sink);
        hipEventRecord(stop, nullptr);
        hipEventSynchronize(stop);
        float ms = 0.f;
        hipEventElapsedTime(&ms, start, stop);
        if (ms < best) { best = ms; }
    }
    const double nsPerRound = best * 1e6 / iterations;
    std::printf("%-62s %8.3f ms   %8.1f ns per round of %d waves (%6.0f cycles at 2.4 GHz)\n", name, best, nsPerRound, WAVES, nsPerRound * 2.4);
    return best;
}

template <int VALU, int MFMA_F32, int MFMA_BF16, int ROLES = 0>
static double run(const char *name, int iterations, float *sink)
{
    static unsigned int *census = nullptr;
    if (!census) { hipMalloc(&census, 8); }
    hipMemset(census, 0, 8);
    hipEvent_t start, stop;
    hipEventCreate(&start); hipEventCreate(&stop);
    const dim3 grid(256 * 4), block(256);
    hipLaunchKernelGGL((probe<VALU, MFMA_F32, MFMA_BF16, ROLES>), grid, block, 0, nullptr, iterations / 10, 1.f, sink, census);
    hipDeviceSynchronize();
    unsigned int slots[2] = { 0, 0 };
    hipMemcpy(slots, census, 8, hipMemcpyDeviceToHost);
    double best = 1e30;
    for (int repeat = 0; repeat < 3; repeat++) {
        hipEventRecord(start, nullptr);
        hipLaunchKernelGGL((probe<VALU, MFMA_F32, MFMA_BF16, ROLES>), grid, block, 0, nullptr, iterations, 1.f, sink, (unsigned int *)nullptr);
        hipEventRecord(stop, nullptr);
        hipEventSynchronize(stop);
        float ms = 0.f;
        hipEventElapsedTime(&ms, start, stop);
        if (ms < best) { best = ms; }
    }
    // four waves per SIMD: a "round" = one iteration of each of the four waves
    const double nsPerRound = best * 1e6 / iterations;
    std::printf("%-62s %8.3f ms   %8.1f ns per round of four waves (%6.0f cycles at 2.4 GHz)  even / odd slots %u / %u\n", name, best, nsPerRound, nsPerRound * 2.4, slots[0], slots[1]);
    return best;
}

int main()
{
    float *sink;
    hipMalloc(&sink, 4);
    const int iterations = 20000;
    const double valu = run<192, 0, 0>("192 v_fma_f32", iterations, sink);
    const double f32 = run<0, 4, 0>("4 v_mfma_f32_32x32x2_f32", iterations, sink);
    const double both32 = run<192, 4, 0>("192 v_fma_f32 + 4 v_mfma_f32_32x32x2_f32", iterations, sink);
    const double bf = run<0, 0, 8>("8 v_mfma_f32_32x32x16_bf16", iterations, sink);
    const double bothbf = run<192, 0, 8>("192 v_fma_f32 + 8 v_mfma_f32_32x32x16_bf16", iterations, sink);
    const double bf4 = run<0, 0, 4>("4 v_mfma_f32_32x32x16_bf16", iterations, sink);
    const double bothbf4 = run<192, 0, 4>("192 v_fma_f32 + 4 v_mfma_f32_32x32x16_bf16", iterations, sink);
    run<192, 4, 0, 1>("roles split: 2 x 192 v_fma_f32 | 2 x 4 f32 MFMA", iterations, sink);
    run<192, 0, 8, 1>("roles split: 2 x 192 v_fma_f32 | 2 x 8 bf16 MFMA", iterations, sink);
    run<192, 0, 4, 1>("roles split: 2 x 192 v_fma_f32 | 2 x 4 bf16 MFMA", iterations, sink);
    run<192, 4, 0, 2>("staggered: 192 v_fma_f32 + 4 f32 MFMA", iterations, sink);
    run<192, 0, 8, 2>("staggered: 192 v_fma_f32 + 8 bf16 MFMA", iterations, sink);
    run<192, 0, 4, 2>("staggered: 192 v_fma_f32 + 4 bf16 MFMA", iterations, sink);
    run<96, 2, 0, 0>("fine-grained: 96 v_fma_f32 + 2 f32 MFMA (x2 iterations)", iterations * 2, sink);
    run<96, 0, 4, 0>("fine-grained: 96 v_fma_f32 + 4 bf16 MFMA (x2 iterations)", iterations * 2, sink);
    run<96, 0, 2, 0>("fine-grained: 96 v_fma_f32 + 2 bf16 MFMA (x2 iterations)", iterations * 2, sink);
    runInterleaved<192, 4, false, 4>("interleaved, 4 waves: 4 x (48 v_fma_f32, 1 f32 MFMA)", iterations, sink);
    runInterleaved<192, 4, true, 4>("interleaved, 4 waves: 4 x (48 v_fma_f32, 1 bf16 MFMA)", iterations, sink);
    runInterleaved<192, 8, true, 4>("interleaved, 4 waves: 8 x (24 v_fma_f32, 1 bf16 MFMA)", iterations, sink);
    runInterleaved<192, 2, true, 4>("interleaved, 4 waves: 2 x (96 v_fma_f32, 1 bf16 MFMA)", iterations, sink);
    runInterleaved<192, 1, true, 4>("interleaved, 4 waves: 1 x (192 v_fma_f32, 1 bf16 MFMA)", iterations, sink);
    runInterleaved<192, 1, false, 4>("interleaved, 4 waves: 1 x (192 v_fma_f32, 1 f32 MFMA)", iterations, sink);
    runInterleaved<192, 4, false, 1>("interleaved, 1 wave: 4 x (48 v_fma_f32, 1 f32 MFMA)", iterations, sink);
    runInterleaved<192, 4, true, 1>("interleaved, 1 wave: 4 x (48 v_fma_f32, 1 bf16 MFMA)", iterations, sink);
    runInterleaved<192, 24, true, 1>("interleaved, 1 wave: 24 x (8 v_fma_f32, 1 bf16 MFMA)", iterations, sink);
    runInterleaved<192, 24, true, 2>("interleaved, 2 waves: 24 x (8 v_fma_f32, 1 bf16 MFMA)", iterations, sink);
    runInterleaved<192, 24, false, 1>("interleaved, 1 wave: 24 x (8 v_fma_f32, 1 f32 MFMA)", iterations, sink);
    runInterleaved<192, 1, true, 1>("interleaved, 1 wave: 1 x (192 v_fma_f32, 1 bf16 MFMA)", iterations, sink);
    runInterleaved<192, 1, true, 2>("interleaved, 2 waves: 1 x (192 v_fma_f32, 1 bf16 MFMA)", iterations, sink);
    std::printf("f32 MFMA:  both / (valu + mfma) = %.3f   both / max = %.3f\n", both32 / (valu + f32), both32 / (valu > f32 ? valu : f32));
    std::printf("bf16 MFMA (8): both / (valu + mfma) = %.3f   both / max = %.3f\n", bothbf / (valu + bf), bothbf / (valu > bf ? valu : bf));
    std::printf("bf16 MFMA (4): both / (valu + mfma) = %.3f   both / max = %.3f\n", bothbf4 / (valu + bf4), bothbf4 / (valu > bf4 ? valu : bf4));
    return 0;
}
