// Are sincosf's two results the bits of sinf and cosf (ocml, gfx950)?  Sweeps every float in [0, 8) plus a coarse sweep
// of the rest of the finite range: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/bin/sincos_identity tools/sincos_identity.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

__global__ void compare(unsigned int first, unsigned int count, unsigned int stride, unsigned long long *differences, unsigned int *example)
{
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) { return; }
    const unsigned int bits = first + i * stride;
    const float x = __uint_as_float(bits);
    float s, c;
    sincosf(x, &s, &c);
    const float s1 = sinf(x), c1 = cosf(x);
    if (__float_as_uint(s) != __float_as_uint(s1) || __float_as_uint(c) != __float_as_uint(c1)) {
        if (!(s != s && s1 != s1 && c != c && c1 != c1)) {   // NaN payloads aside
            atomicAdd(differences, 1ull);
            *example = bits;
        }
    }
}

int main()
{
    unsigned long long *differences; unsigned int *example;
    hipMalloc(&differences, 8); hipMalloc(&example, 4);
    hipMemset(differences, 0, 8); hipMemset(example, 0, 4);
    const unsigned int eight = 0x41000000u;   // 8.0f: every float in [0, 8)
    for (unsigned int sign = 0; sign < 2; sign++) {
        compare<<<(eight + 255) / 256, 256>>>(sign << 31, eight, 1, differences, example);
    }
    compare<<<(0x7F800000u / 97 + 255) / 256, 256>>>(0u, 0x7F800000u / 97, 97, differences, example);   // the rest, every 97th
    hipDeviceSynchronize();
    unsigned long long host = 0; unsigned int bits = 0;
    hipMemcpy(&host, differences, 8, hipMemcpyDeviceToHost); hipMemcpy(&bits, example, 4, hipMemcpyDeviceToHost);
    float x; std::memcpy(&x, &bits, 4);
    std::printf("sincosf vs sinf / cosf: %llu inputs differ of %u + %u (example %g)\n", host, 2 * eight, 0x7F800000u / 97, host ? x : 0.0);
    return host ? 1 : 0;
}
