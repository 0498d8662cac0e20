// 4x4 fp32 transform with the reference's exact construction order.
// Reference: parseTransform (src/scene_parser.cpp:690-793) builds the matrix by
// PRE-multiplying scale, rotations, translate onto identity with matrix::*
// helpers (src/matrix.cpp:43-162) and builds the inverse analytically;
// Transform::apply (src/transform.cpp:64-104) evaluates rows left to right.
#pragma once

#include <cmath>
#include <cstring>

namespace pathed {

struct Mat4 {
    float m[4][4];

    static Mat4 identity()
    {
        Mat4 r;
        for (int i = 0; i < 4; i++) {
            for (int j = 0; j < 4; j++) { r.m[i][j] = (i == j) ? 1.f : 0.f; }
        }
        return r;
    }

    // result = A * B, accumulated from 0.f in k order (reference src/matrix.cpp:69-80)
    static Mat4 multiply(const Mat4 &a, const Mat4 &b)
    {
        Mat4 r;
        for (int row = 0; row < 4; row++) {
            for (int col = 0; col < 4; col++) {
                float sum = 0.f;
                for (int i = 0; i < 4; i++) { sum += a.m[row][i] * b.m[i][col]; }
                r.m[row][col] = sum;
            }
        }
        return r;
    }

    void preScale(float x, float y, float z)
    {
        Mat4 s = identity();
        s.m[0][0] = x; s.m[1][1] = y; s.m[2][2] = z;
        *this = multiply(s, *this);
    }

    void preTranslate(float x, float y, float z)
    {
        Mat4 t = identity();
        t.m[0][3] = x; t.m[1][3] = y; t.m[2][3] = z;
        *this = multiply(t, *this);
    }

    void preRotateX(float theta)
    {
        Mat4 r = identity();
        r.m[1][1] = cosf(theta); r.m[1][2] = -sinf(theta);
        r.m[2][1] = sinf(theta); r.m[2][2] = cosf(theta);
        *this = multiply(r, *this);
    }

    void preRotateY(float theta)
    {
        Mat4 r = identity();
        r.m[0][0] = cosf(theta); r.m[0][2] = sinf(theta);
        r.m[2][0] = -sinf(theta); r.m[2][2] = cosf(theta);
        *this = multiply(r, *this);
    }

    void preRotateZ(float theta)
    {
        Mat4 r = identity();
        r.m[0][0] = cosf(theta); r.m[0][1] = -sinf(theta);
        r.m[1][0] = sinf(theta); r.m[1][1] = cosf(theta);
        *this = multiply(r, *this);
    }

    void applyPoint(const float p[3], float out[3]) const
    {
        const float x = p[0], y = p[1], z = p[2];
        out[0] = m[0][0] * x + m[0][1] * y + m[0][2] * z + m[0][3];
        out[1] = m[1][0] * x + m[1][1] * y + m[1][2] * z + m[1][3];
        out[2] = m[2][0] * x + m[2][1] * y + m[2][2] * z + m[2][3];
    }

    void applyVector(const float v[3], float out[3]) const
    {
        const float x = v[0], y = v[1], z = v[2];
        out[0] = m[0][0] * x + m[0][1] * y + m[0][2] * z;
        out[1] = m[1][0] * x + m[1][1] * y + m[1][2] * z;
        out[2] = m[2][0] * x + m[2][1] * y + m[2][2] * z;
    }

    void toArray(float out[16]) const { std::memcpy(out, m, sizeof m); }
};

struct Transform {
    Mat4 matrix = Mat4::identity();
    Mat4 inverse = Mat4::identity();
};

}  // namespace pathed
