// ORACLE/_ref -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
// Stand-in for the HOST side of NVIDIA-RTX/MathLib (ml.h), which the reference fetches unpinned at configure time (CMakeLists.txt:118-127) and which is absent from
// /root/reference: "parity unpinned" for THIS FILE ALONE. It exists so that the reference's own host sources -- Source/InstanceImpl.cpp, Wrapper.cpp, Reblur.cpp, Relax.cpp,
// Sigma.cpp, Reference.cpp, Timer.cpp and Source/Denoisers/*.hpp, compiled where they lie by oracle/ref/host/Makefile -- link into oracle/_ref/libnrdhost.so, the executable form
// of the reference's dispatch compiler that tests/test_ref_host.py holds the product's host (raytracingdenoiser_amd/csrc/host) against: pool layouts, pipelines, the resource
// list and grid of every dispatch, and every constant buffer, for all 19 denoisers.
// Only what those sources use is here (types float2 / float3 / float4 / float4x4 / uint2 / int2, DecomposeProjection, Rotate, min / max / clamp / saturate / lerp / radians).
// Conventions restated (the same ones DESIGN.md section 4.1 "The restated MathLib" anchors on what the reference does pin):
//   float4x4( c0, c1, c2, c3 ): columns; m * v = sum c_i * v_i; element aRC = row R of column C; GetRowN; InvertOrtho = inverse of a rigid transform;
//   DecomposeProjection( D3D ): frustum = ( -(1 + P02) / P00, (1 - P12) / P11, 2 / P00, -2 / P11 ) with P02 / P12 negated for right-handed matrices -- what
//   Geometry::ReconstructViewPosition( uv, frustum, viewZ ) needs to invert the projection --, project[ 1 ] = | P11 |, PROJ_ORTHO = ( P33 == 1 && P32 == 0 ),
//   PROJ_LEFT_HANDED = clip.w has the sign of view z (ortho: P22 >= 0);
//   sizeof( float3 ) == 16 (InstanceImpl.h:79 relies on it); the fourth lane of float3::xmm is 0.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>

struct v4f {
    float f[4];
};

struct float2 {
    float x, y;
    float2() : x(0), y(0) {}
    float2(float a, float b) : x(a), y(b) {}
};
struct uint2 {
    uint32_t x, y;
    uint2() : x(0), y(0) {}
    template <class A, class B> uint2(A a, B b) : x((uint32_t)a), y((uint32_t)b) {}
};
struct int2 {
    int32_t x, y;
    int2() : x(0), y(0) {}
    template <class A, class B> int2(A a, B b) : x((int32_t)a), y((int32_t)b) {}
};

struct float4;
struct alignas(16) float3 {
    union {
        struct {
            float x, y, z;
        };
        v4f xmm;
    };
    float3() { xmm = v4f{{0, 0, 0, 0}}; }
    float3(float a, float b, float c) { xmm = v4f{{a, b, c, 0}}; }
    float3(const float3& o) { xmm = o.xmm; }
    explicit float3(const float4& v);
    float3& operator=(const float3& o) {
        xmm = o.xmm;
        return *this;
    }
    static float3 Zero() { return float3(); }
    float3 operator-() const { return float3(-x, -y, -z); }
    float3 operator-(const float3& o) const { return float3(x - o.x, y - o.y, z - o.z); }
    float3 operator+(const float3& o) const { return float3(x + o.x, y + o.y, z + o.z); }
    float3 operator*(float s) const { return float3(x * s, y * s, z * s); }
};

struct alignas(16) float4 {
    union {
        struct {
            float x, y, z, w;
        };
        float a[4];
        v4f xmm;
        float3 xyz;
    };
    float4() { xmm = v4f{{0, 0, 0, 0}}; }
    float4(float X, float Y, float Z, float W) { xmm = v4f{{X, Y, Z, W}}; }
    explicit float4(const float* p) { memcpy(a, p, 16); }
    float4(const v4f& v) { xmm = v; }
    float4(const float4& o) { xmm = o.xmm; }
    float4& operator=(const float4& o) {
        xmm = o.xmm;
        return *this;
    }
    float4& operator=(const v4f& v) {
        xmm = v;
        return *this;
    }
    static float4 Zero() { return float4(); }
    float4 operator-() const { return float4(-x, -y, -z, -w); }
    float4 operator*(const float4& o) const { return float4(x * o.x, y * o.y, z * o.z, w * o.w); }
    float4 operator+(const float4& o) const { return float4(x + o.x, y + o.y, z + o.z, w + o.w); }
    float4 operator*(float s) const { return float4(x * s, y * s, z * s, w * s); }
};
inline float3::float3(const float4& v) { xmm = v4f{{v.x, v.y, v.z, 0}}; }

struct alignas(16) float4x4 {
    union {
        float4 cols[4]; // (col0 .. col3 below: an anonymous struct may not hold members with constructors)
        struct { // aRC: row R, column C
            float a00, a10, a20, a30, a01, a11, a21, a31, a02, a12, a22, a32, a03, a13, a23, a33;
        };
        float m[16];
    };
    float4x4() { memset(m, 0, sizeof(m)); }
    float4x4(const float4& c0, const float4& c1, const float4& c2, const float4& c3) {
        cols[0] = c0;
        cols[1] = c1;
        cols[2] = c2;
        cols[3] = c3;
    }
    float4x4(const float4x4& o) { memcpy(m, o.m, sizeof(m)); }
    float4x4& operator=(const float4x4& o) {
        memcpy(m, o.m, sizeof(m));
        return *this;
    }
    static float4x4 Identity() { return float4x4(float4(1, 0, 0, 0), float4(0, 1, 0, 0), float4(0, 0, 1, 0), float4(0, 0, 0, 1)); }
    float4& operator[](int i) { return cols[i]; }
    const float4& operator[](int i) const { return cols[i]; }
    bool operator!=(const float4x4& o) const { return memcmp(m, o.m, sizeof(m)) != 0; }
    float at(int r, int c) const { return m[c * 4 + r]; }
    float4 GetRow0() const { return float4(a00, a01, a02, a03); }
    float4 GetRow1() const { return float4(a10, a11, a12, a13); }
    float4 GetRow2() const { return float4(a20, a21, a22, a23); }
    void Transpose() {
        float t[16];
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++)
                t[r * 4 + c] = m[c * 4 + r];
        memcpy(m, t, sizeof(m));
    }
    void SetTranslation(const float3& t) {
        cols[3].x = t.x;
        cols[3].y = t.y;
        cols[3].z = t.z;
    }
    // inverse of a rigid transform: the rotation transposed, translation' = -( R^T t )
    void InvertOrtho() {
        float4x4 r;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                r.m[j * 4 + i] = m[i * 4 + j];
        for (int i = 0; i < 3; i++)
            r.m[12 + i] = -(r.m[0 + i] * cols[3].x + r.m[4 + i] * cols[3].y + r.m[8 + i] * cols[3].z);
        r.m[3] = r.m[7] = r.m[11] = 0.0f;
        r.m[15] = 1.0f;
        *this = r;
    }
    // general inverse (Gauss-Jordan with partial pivoting in double precision, rounded once)
    void Invert() {
        double A[4][8];
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) {
                A[r][c] = m[c * 4 + r];
                A[r][4 + c] = r == c ? 1.0 : 0.0;
            }
        for (int i = 0; i < 4; i++) {
            int p = i;
            for (int r = i + 1; r < 4; r++)
                if (std::fabs(A[r][i]) > std::fabs(A[p][i]))
                    p = r;
            for (int c = 0; c < 8; c++)
                std::swap(A[i][c], A[p][c]);
            double d = A[i][i];
            for (int c = 0; c < 8; c++)
                A[i][c] /= d;
            for (int r = 0; r < 4; r++)
                if (r != i) {
                    double f = A[r][i];
                    for (int c = 0; c < 8; c++)
                        A[r][c] -= f * A[i][c];
                }
        }
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++)
                m[c * 4 + r] = (float)A[r][4 + c];
    }
    float4 operator*(const float4& v) const { return cols[0] * v.x + cols[1] * v.y + cols[2] * v.z + cols[3] * v.w; }
    float4x4 operator*(const float4x4& o) const { return float4x4(*this * o.cols[0], *this * o.cols[1], *this * o.cols[2], *this * o.cols[3]); }
};
#define col0 cols[0]
#define col1 cols[1]
#define col2 cols[2]
#define col3 cols[3]

// the 3x3 part applied to a direction
inline float3 Rotate(const float4x4& m, const float3& v) {
    float4 r = m.cols[0] * v.x + m.cols[1] * v.y + m.cols[2] * v.z;
    return float3(r.x, r.y, r.z);
}

enum eStyle { STYLE_D3D, STYLE_OGL };
enum eProjectionFlag : uint32_t { PROJ_ORTHO = 0x1, PROJ_REVERSED_Z = 0x2, PROJ_LEFT_HANDED = 0x4 };

inline void DecomposeProjection(eStyle, eStyle, const float4x4& p, uint32_t* flags, float* settings15, float* unproject2, float* frustum4, float* project3, float* safeNearZ) {
    (void)settings15;
    (void)unproject2;
    (void)safeNearZ;
    const bool ortho = p.at(3, 3) == 1.0f && p.at(3, 2) == 0.0f;
    const bool leftHanded = ortho ? (p.at(2, 2) >= 0.0f) : (p.at(3, 2) > 0.0f);
    if (flags)
        *flags = (ortho ? PROJ_ORTHO : 0u) | (leftHanded ? PROJ_LEFT_HANDED : 0u);
    const float p00 = p.at(0, 0), p11 = p.at(1, 1);
    if (frustum4) {
        const float s = leftHanded ? 1.0f : -1.0f;
        const float ox = ortho ? p.at(0, 3) : p.at(0, 2) * s, oy = ortho ? p.at(1, 3) : p.at(1, 2) * s;
        frustum4[0] = -(1.0f + ox) / p00;
        frustum4[1] = (1.0f - oy) / p11;
        frustum4[2] = 2.0f / p00;
        frustum4[3] = -2.0f / p11;
    }
    if (project3) {
        project3[0] = std::fabs(p00);
        project3[1] = std::fabs(p11);
        project3[2] = 0.0f;
    }
}

// scalar helpers the host sources call unqualified
template <class T> inline T min(T a, T b) { return a < b ? a : b; }
template <class T> inline T max(T a, T b) { return a > b ? a : b; }
template <class T> inline T clamp(T x, T a, T b) { return x < a ? a : (x > b ? b : x); }
inline float saturate(float x) { return clamp(x, 0.0f, 1.0f); }
inline float lerp(float a, float b, float t) { return a + (b - a) * t; }
inline float radians(float deg) { return deg * (3.14159265358979323846f / 180.0f); } // (one multiplication by the constant; the other association differs by an ulp: a MathLib ambiguity)
template <class T> inline void Swap(T& a, T& b) {
    T t = a;
    a = b;
    b = t;
}
using std::abs;
using std::log;
