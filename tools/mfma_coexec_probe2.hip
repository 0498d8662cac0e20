// Second, stricter co-execution probe: the instruction streams are written in inline assembly so that the compiler can
// neither pack the v_fma_f32 into v_pk_fma_f32 nor regroup the matrix instructions (it did both to mfma_coexec_probe.hip).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_coexec_probe2 tools/mfma_coexec_probe2.hip
// Per loop iteration a wave issues GROUPS x (FMAS v_fma_f32 on eight independent VGPR chains, then MFMAS matrix
// instructions on rotating accumulators).  WAVES waves per SIMD.  Reported: cycles (at 2.4 GHz) per iteration of ONE SIMD
// (all its waves), next to what the VALU part and the matrix part take alone.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define FMA8 \
    "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
    "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"

template <int FMA_OCTETS, int MFMAS, bool BF16, int GROUPS, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void probe(int iterations, float seed, float *sink)
{
    float c0 = seed, c1 = seed + 1.f, c2 = seed + 2.f, c3 = seed + 3.f, c4 = seed + 4.f, c5 = seed + 5.f, c6 = seed + 6.f, c7 = seed + 7.f + threadIdx.x;
    const float m = 1.0001f, k = 1e-6f;
    f16v acc0, acc1, acc2, acc3;
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
    f4v pa, pb;
    for (int r = 0; r < 4; r++) { pa[r] = seed * 1e-3f * (float)r; pb[r] = seed * 1e-3f + (float)r; }
    const float a = seed * 0.5f, b = seed * 0.25f;
    for (int it = 0; it < iterations; it++) {
#pragma unroll
        for (int g = 0; g < GROUPS; g++) {
#pragma unroll
            for (int v = 0; v < FMA_OCTETS; v++) {
                asm volatile(FMA8 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(m), "v"(k));
            }
#pragma unroll
            for (int q = 0; q < MFMAS; q++) {
                if (BF16) {
                    switch ((g * MFMAS + q) & 3) {
                    case 0: asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(pa), "v"(pb)); break;
                    case 1: asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(pa), "v"(pb)); break;
                    case 2: asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc2) : "v"(pa), "v"(pb)); break;
                    default: asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc3) : "v"(pa), "v"(pb)); break;
                    }
                } else {
                    switch ((g * MFMAS + q) & 3) {
                    case 0: asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b)); break;
                    case 1: asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b)); break;
                    case 2: asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc2) : "v"(a), "v"(b)); break;
                    default: asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc3) : "v"(a), "v"(b)); break;
                    }
                }
            }
        }
    }
    float total = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    for (int r = 0; r < 16; r++) { total += acc0[r] + acc1[r] + acc2[r] + acc3[r]; }
    if (total == 123.456f) { sink[0] = total; }
}

template <int FMA_OCTETS, int MFMAS, bool BF16, int GROUPS, int WAVES>
static double run(int iterations, float *sink)
{
    hipEvent_t start, stop;
    hipEventCreate(&start); hipEventCreate(&stop);
    const dim3 grid(256 * WAVES), block(256);
    hipLaunchKernelGGL((probe<FMA_OCTETS, MFMAS, BF16, GROUPS, WAVES>), grid, block, 0, nullptr, iterations / 10, 1.f, sink);
    hipDeviceSynchronize();
    double best = 1e30;
    for (int repeat = 0; repeat < 3; repeat++) {
        hipEventRecord(start, nullptr);
        hipLaunchKernelGGL((probe<FMA_OCTETS, MFMAS, BF16, GROUPS, WAVES>), grid, block, 0, nullptr, iterations, 1.f, sink);
        hipEventRecord(stop, nullptr);
        hipEventSynchronize(stop);
        float ms = 0.f;
        hipEventElapsedTime(&ms, start, stop);
        if (ms < best) { best = ms; }
    }
    return best * 1e6 / iterations * 2.4;   // cycles per iteration of one SIMD at 2.4 GHz
}

template <int FMA_OCTETS, int MFMAS, bool BF16, int GROUPS, int WAVES>
static void row(int iterations, float *sink)
{
    const double valu = run<FMA_OCTETS, 0, BF16, GROUPS, WAVES>(iterations, sink);
    const double mfma = run<0, MFMAS, BF16, GROUPS, WAVES>(iterations, sink);
    const double both = run<FMA_OCTETS, MFMAS, BF16, GROUPS, WAVES>(iterations, sink);
    const int nMfma = MFMAS * GROUPS * WAVES;
    std::printf("%d waves x %2d x (%3d v_fma_f32, %d %s MFMA): VALU alone %7.0f  MFMA alone %7.0f  both %7.0f cycles   both / sum %.3f  both / max %.3f   cost per MFMA beside the VALU stream %5.1f cycles (of %d)\n",
                WAVES, GROUPS, FMA_OCTETS * 8, MFMAS, BF16 ? "bf16" : "f32 ", valu, mfma, both, both / (valu + mfma), both / (valu > mfma ? valu : mfma),
                (both - valu) / nMfma, BF16 ? 32 : 64);
}

int main()
{
    float *sink;
    hipMalloc(&sink, 4);
    const int n = 20000;
    row<24, 4, false, 1, 4>(n, sink);
    row<24, 4, true, 1, 4>(n, sink);
    row<24, 1, false, 1, 4>(n, sink);
    row<24, 1, true, 1, 4>(n, sink);
    row<6, 1, false, 4, 4>(n, sink);
    row<6, 1, true, 4, 4>(n, sink);
    row<3, 1, true, 8, 4>(n, sink);
    row<1, 1, true, 24, 4>(n, sink);
    row<6, 1, false, 4, 1>(n, sink);
    row<6, 1, true, 4, 1>(n, sink);
    row<1, 1, true, 24, 1>(n, sink);
    row<1, 1, false, 24, 1>(n, sink);
    row<2, 1, false, 12, 1>(n, sink);
    row<1, 1, true, 24, 2>(n, sink);
    row<2, 1, false, 12, 2>(n, sink);
    row<6, 1, false, 4, 2>(n, sink);
    return 0;
}
