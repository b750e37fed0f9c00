// Host-side linear algebra for the per-frame camera set-up (the role "ml.h" from NVIDIA-RTX/MathLib plays in
// reference Source/InstanceImpl.cpp:339-470; MathLib is NOT vendored in the reference, so these are our own
// definitions -- see DESIGN.md "restated MathLib").
//
// Conventions (reference Include/NRDSettings.h:90-94): matrices are column-major, vectors are columns,
// clip = viewToClip * view. m.c[j] is column j; element (row i, col j) is m.c[j].v[i].
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

namespace nrdhost {

struct Vec3 {
    float x = 0, y = 0, z = 0;
};
struct Vec4 {
    float v[4] = {0, 0, 0, 0};
    float& operator[](int i) { return v[i]; }
    float operator[](int i) const { return v[i]; }
};

inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }

struct Mat4 {
    Vec4 c[4];

    static Mat4 Identity() {
        Mat4 m;
        for (int j = 0; j < 4; j++)
            m.c[j].v[j] = 1.0f;
        return m;
    }
    static Mat4 FromColumnMajor(const float* p) {
        Mat4 m;
        for (int j = 0; j < 4; j++)
            for (int i = 0; i < 4; i++)
                m.c[j].v[i] = p[j * 4 + i];
        return m;
    }
    float at(int row, int col) const { return c[col].v[row]; }
    float& at(int row, int col) { return c[col].v[row]; }
    bool operator==(const Mat4& o) const { return memcmp(this, &o, sizeof(Mat4)) == 0; }
    bool operator!=(const Mat4& o) const { return !(*this == o); }
};

// a * b (apply b first)
inline Mat4 Mul(const Mat4& a, const Mat4& b) {
    Mat4 r;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) {
            float s = 0.0f;
            for (int k = 0; k < 4; k++)
                s += a.at(i, k) * b.at(k, j);
            r.at(i, j) = s;
        }
    return r;
}

inline Mat4 Transposed(const Mat4& m) {
    Mat4 r;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++)
            r.at(i, j) = m.at(j, i);
    return r;
}

// Inverse of a rigid transform [R | t]: [R^T | -R^T t]
inline Mat4 InvertRigid(const Mat4& m) {
    Mat4 r = Mat4::Identity();
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            r.at(i, j) = m.at(j, i);
    for (int i = 0; i < 3; i++)
        r.at(i, 3) = -(r.at(i, 0) * m.at(0, 3) + r.at(i, 1) * m.at(1, 3) + r.at(i, 2) * m.at(2, 3));
    return r;
}

// General inverse (double precision cofactor expansion, rounded once to fp32).
inline Mat4 Invert(const Mat4& mm) {
    double m[16], inv[16];
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++)
            m[j * 4 + i] = mm.c[j].v[i];

    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];

    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    double invDet = det != 0.0 ? 1.0 / det : 0.0;

    Mat4 r;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++)
            r.c[j].v[i] = (float)(inv[j * 4 + i] * invDet);
    return r;
}

// What reference InstanceImpl.cpp:392,445-451 asks of ml.h's DecomposeProjection for a D3D-style projection
// (depth = z / w). Derivation: with uv = ndc * (0.5, -0.5) + 0.5 and clip.w = +/-z,
//   Xv.xy = (uv * frustum.zw + frustum.xy) * viewZ must invert the projection.
struct ProjectionInfo {
    bool isLeftHanded = true;
    bool isOrtho = false;
    float frustum[4] = {0, 0, 0, 0}; // ( -(1+P02)/P00, (1-P12)/P11, 2/P00, -2/P11 ) for LH perspective
    float projectY = 1.0f;           // |P11|
};

inline ProjectionInfo DecomposeProjection(const Mat4& p) {
    ProjectionInfo r;
    r.isOrtho = p.at(3, 3) == 1.0f && p.at(3, 2) == 0.0f;
    r.isLeftHanded = r.isOrtho ? (p.at(2, 2) >= 0.0f) : (p.at(3, 2) > 0.0f);

    float p00 = p.at(0, 0), p11 = p.at(1, 1);
    if (!r.isOrtho) {
        float s = r.isLeftHanded ? 1.0f : -1.0f; // clip.w = s * z
        float p02 = p.at(0, 2) * s, p12 = p.at(1, 2) * s;
        r.frustum[0] = -(1.0f + p02) / p00;
        r.frustum[1] = (1.0f - p12) / p11;
        r.frustum[2] = 2.0f / p00;
        r.frustum[3] = -2.0f / p11;
    } else {
        // Xv.xy = uv * frustum.zw + frustum.xy (no depth scaling); reconstruction flips sign via orthoMode
        float p03 = p.at(0, 3), p13 = p.at(1, 3);
        r.frustum[0] = -(1.0f + p03) / p00;
        r.frustum[1] = (1.0f - p13) / p11;
        r.frustum[2] = 2.0f / p00;
        r.frustum[3] = -2.0f / p11;
    }
    r.projectY = std::fabs(p11);
    return r;
}

// ---- low-discrepancy helpers used for the per-frame kernel rotators (reference InstanceImpl.cpp:339-349).
// MathLib's Sequence::Weyl1D / Bayer4x4 restated: additive recurrence with the 24-bit golden-ratio
// increment; 4x4 ordered-dither index advanced by the frame index.
inline float Weyl1D(float p, uint32_t n) {
    float t = p + float((n * 10368889u) & 0x00FFFFFFu) / 16777216.0f;
    return t - std::floor(t);
}

// NRD_MATHLIB_BAYER_REVERSEBITS (default 0): MathLib is not vendored in the reference tree, so Sequence::Bayer4x4ui is a restatement; the round-5 reviewer recalls a MathLib
// default ML_BAYER_REVERSEBITS that advances the dither index by ReverseBits4( frameIndex ) instead of frameIndex. Unverifiable here; 1 selects that alternative in every place the
// function is restated (product host + device, oracle, both MathLib stand-ins under oracle/ref) -- it changes the dither PHASE per frame, nothing else (INTEGRATION.md "MathLib").
#ifndef NRD_MATHLIB_BAYER_REVERSEBITS
#define NRD_MATHLIB_BAYER_REVERSEBITS 0
#endif
inline uint32_t BayerFrameOffset(uint32_t frameIndex) {
#if NRD_MATHLIB_BAYER_REVERSEBITS
    const uint32_t v = frameIndex & 0xFu;
    return ((v & 1u) << 3) | ((v & 2u) << 1) | ((v & 4u) >> 1) | ((v & 8u) >> 3); // ReverseBits4
#else
    return frameIndex;
#endif
}
inline uint32_t Bayer4x4ui(uint32_t x, uint32_t y, uint32_t frameIndex) {
    x &= 3u;
    y &= 3u;
    uint32_t a = 2068378560u * (1u - (x >> 1)) + 1500172770u * (x >> 1);
    uint32_t b = (y + ((x & 1u) << 2)) << 2;
    return ((a >> b) + BayerFrameOffset(frameIndex)) & 0xFu;
}

// round 5: i / 16, "RESULT: [0; 1)" -- the form both the builder's and the round-4 reviewer's recollection of NVIDIA-RTX/MathLib agree on (until then (i + 0.5) / 16; MathLib is not vendored: unpinned either way)
inline float Bayer4x4(uint32_t x, uint32_t y, uint32_t frameIndex) { return float(Bayer4x4ui(x, y, frameIndex)) / 16.0f; }

// Rotator = (cos, sin, -sin, cos); v' = v.x * r.xz + v.y * r.yw  (reference Common.hlsli:465 default (1,0,0,1))
inline Vec4 GetRotator(float angle) {
    float ca = (float)std::cos((double)angle), sa = (float)std::sin((double)angle);
    Vec4 r;
    r.v[0] = ca;
    r.v[1] = sa;
    r.v[2] = -sa;
    r.v[3] = ca;
    return r;
}

inline Vec4 CombineRotators(const Vec4& r1, const Vec4& r2) {
    // r1.xyxy * r2.xxzz + r1.zwzw * r2.yyww
    Vec4 r;
    r.v[0] = r1[0] * r2[0] + r1[2] * r2[1];
    r.v[1] = r1[1] * r2[0] + r1[3] * r2[1];
    r.v[2] = r1[0] * r2[2] + r1[2] * r2[3];
    r.v[3] = r1[1] * r2[2] + r1[3] * r2[3];
    return r;
}

constexpr float kPi = 3.14159265358979323846f;
inline float Radians(float deg) { return deg * (kPi / 180.0f); }

} // namespace nrdhost
