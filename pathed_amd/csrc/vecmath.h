// fp32 vector helpers for the device code.
//
// Two families, on purpose:
//   * "reference-order" ops (dot, cross, normalized, ...) evaluate exactly like the
//     reference's Vector3/Point3/Color classes (src/vector.cpp, src/point.cpp,
//     src/color.cpp): separate multiplies and adds, left to right.  The library is
//     compiled with -ffp-contract=off so hipcc does not fuse them.
//   * "x" ops (xdot, xcross) are the intersector's arithmetic: explicit fmaf, one
//     v_fma_f32 each.  They are part of the intersector specification in DESIGN.md.
#pragma once

#include <hip/hip_runtime.h>

namespace pathed {

struct V3 {
    float x, y, z;
};

__host__ __device__ inline V3 v3(float x, float y, float z) { V3 v = { x, y, z }; return v; }
__host__ __device__ inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__host__ __device__ inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__host__ __device__ inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
__host__ __device__ inline V3 operator*(V3 a, float t) { return v3(a.x * t, a.y * t, a.z * t); }
__host__ __device__ inline bool operator==(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

__host__ __device__ inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

__host__ __device__ inline V3 cross(V3 a, V3 b)
{
    return v3((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x));
}

__host__ __device__ inline float length(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }

__host__ __device__ inline V3 normalized(V3 a)
{
    const float norm = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    return v3(a.x / norm, a.y / norm, a.z / norm);
}

// Vector3::reflect: (normal * dot(normal) * 2) - *this
__host__ __device__ inline V3 reflect(V3 v, V3 normal) { return (normal * dot(v, normal) * 2.f) - v; }

__host__ __device__ inline V3 xcross(V3 a, V3 b)
{
    return v3(
        fmaf(a.y, b.z, -(a.z * b.y)),
        fmaf(a.z, b.x, -(a.x * b.z)),
        fmaf(a.x, b.y, -(a.y * b.x)));
}

__host__ __device__ inline float xdot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }

struct Rgb {
    float r, g, b;
};

__host__ __device__ inline Rgb rgb(float r, float g, float b) { Rgb c = { r, g, b }; return c; }
__host__ __device__ inline Rgb rgb(float v) { return rgb(v, v, v); }
__host__ __device__ inline Rgb operator+(Rgb a, Rgb b) { return rgb(a.r + b.r, a.g + b.g, a.b + b.b); }
__host__ __device__ inline Rgb operator*(Rgb a, Rgb b) { return rgb(a.r * b.r, a.g * b.g, a.b * b.b); }
__host__ __device__ inline Rgb operator*(Rgb a, float t) { return rgb(a.r * t, a.g * t, a.b * t); }
// Color::operator/(float) multiplies by the reciprocal (src/color.cpp:128-135)
__host__ __device__ inline Rgb operator/(Rgb a, float t) { const float inv = 1.f / t; return a * inv; }
__host__ __device__ inline bool isBlack(Rgb c) { return c.r == 0.f && c.g == 0.f && c.b == 0.f; }

// std::max / std::min semantics (the reference uses them, and they differ from
// fmaxf/fminf when an operand is NaN): max(a,b) = (a < b) ? b : a, min(a,b) = (b < a) ? b : a
__host__ __device__ inline float smax(float a, float b) { return (a < b) ? b : a; }
__host__ __device__ inline float smin(float a, float b) { return (b < a) ? b : a; }
__host__ __device__ inline int imin(int a, int b) { return (b < a) ? b : a; }
__host__ __device__ inline int imax(int a, int b) { return (a < b) ? b : a; }

// util::clamp, include/util.h:49-51: std::min(highest, std::max(value, lowest))
__host__ __device__ inline float clampf(float value, float lowest, float highest)
{
    return smin(highest, smax(value, lowest));
}

__device__ inline float4 loadF4(const float4 *p) { return *p; }

__device__ inline int floatAsInt(float f) { return __float_as_int(f); }
__device__ inline float intAsFloat(int i) { return __int_as_float(i); }

}  // namespace pathed
