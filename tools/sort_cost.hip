// What would sorting the ray queue cost?  Times rocPRIM's radix sort of N (16-bit key, 32-bit index)
// pairs -- the cheapest sort that groups rays by (direction octant, 12-bit origin Morton code) -- and
// the gather of N 32-byte ray records through the sorted index, with HIP events.
// Build: hipcc -O3 --offload-arch=gfx950 tools/sort_cost.hip -o tools/bin/sort_cost ; run on a GPU box.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#define CHECK(x) do { hipError_t s_ = (x); if (s_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(s_)); return 1; } } while (0)

__global__ void k_gather(const float4 *rayO, const float4 *rayD, const unsigned int *order, size_t n, float4 *outO, float4 *outD)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) { return; }
    const unsigned int from = order[i];
    outO[i] = rayO[from];
    outD[i] = rayD[from];
}

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : ((size_t)1 << 22);
    std::vector<unsigned int> keys(n), values(n);
    unsigned int state = 12345u;
    for (size_t i = 0; i < n; i++) { state = state * 1664525u + 1013904223u; keys[i] = (state >> 8) & 0x7FFFu; values[i] = (unsigned int)i; }
    unsigned int *keysIn, *keysOut, *valuesIn, *valuesOut;
    float4 *rayO, *rayD, *outO, *outD;
    CHECK(hipMalloc(&keysIn, n * 4)); CHECK(hipMalloc(&keysOut, n * 4)); CHECK(hipMalloc(&valuesIn, n * 4)); CHECK(hipMalloc(&valuesOut, n * 4));
    CHECK(hipMalloc(&rayO, n * 16)); CHECK(hipMalloc(&rayD, n * 16)); CHECK(hipMalloc(&outO, n * 16)); CHECK(hipMalloc(&outD, n * 16));
    CHECK(hipMemcpy(keysIn, keys.data(), n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(valuesIn, values.data(), n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(rayO, 0, n * 16)); CHECK(hipMemset(rayD, 0, n * 16));
    size_t bytes = 0;
    CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keysIn, keysOut, valuesIn, valuesOut, n, 0u, 15u, nullptr));
    void *temporary;
    CHECK(hipMalloc(&temporary, bytes));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int round = 0; round < 2; round++) {   // second round is the timed one
        CHECK(hipEventRecord(a, nullptr));
        for (int r = 0; r < 10; r++) { CHECK(rocprim::radix_sort_pairs(temporary, bytes, keysIn, keysOut, valuesIn, valuesOut, n, 0u, 15u, nullptr)); }
        CHECK(hipEventRecord(b, nullptr));
        CHECK(hipEventSynchronize(b));
        float sortMs = 0.f;
        CHECK(hipEventElapsedTime(&sortMs, a, b));
        CHECK(hipEventRecord(a, nullptr));
        for (int r = 0; r < 10; r++) { hipLaunchKernelGGL(k_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, rayO, rayD, valuesOut, n, outO, outD); }
        CHECK(hipEventRecord(b, nullptr));
        CHECK(hipEventSynchronize(b));
        float gatherMs = 0.f;
        CHECK(hipEventElapsedTime(&gatherMs, a, b));
        if (round == 1) {
            printf("%zu rays: radix sort of (15-bit key, index) pairs %.1f us; gather of 32-byte ray records through the sorted index %.1f us\n",
                   n, sortMs * 100.f, gatherMs * 100.f);
        }
    }
    return 0;
}
