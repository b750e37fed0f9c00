// ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may load anything under oracle/. The shipped library (raytracingdenoiser_amd/lib/libNRD_hip.so) never does.
//
// HLSL-flavoured scalar / vector vocabulary for the CPU restatement of the NRD shader arithmetic, under the numerics contract of the product
// (DESIGN.md "Numerics"), which is what makes a bit-for-bit comparison with the GPU possible:
//   * + - * and fused multiply-add are IEEE-754 binary32. This directory is compiled by clang with -ffp-contract=on -mfma: `a * b + c` written as
//     ONE expression is a single fma (ISO C "FP_CONTRACT ON"), nothing else is fused -- the same front end makes the same decisions for the
//     device sources. The expressions here are therefore written in the order and the shape of the HLSL they restate.
//   * rcp / division / sqrt / rsqrt / exp2 / log2 follow gfx950's v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 / v_exp_f32 / v_log_f32 (each within 1 ulp of the
//     correctly rounded result) through per-mantissa deviation tables measured on the device: oracle/hw_math.h. `a / b` of a shader is
//     Div(a, b) = a * rcp(b) here and on the device. IEEE mode (oracle_set_ieee_mode) replaces the five instructions by their reference
//     results: "the HLSL math on an IEEE machine", the oracle without any knowledge of the device.
//   * UNORM texel decoding (tex.h) is the exact quotient k / (2^n - 1) on both sides.
// Written independently of the HIP device header; the two must agree bit for bit, which is what the parity tests check.
//
// PARITY: pinned to the reference's own shader text since round 4. The reference ships no CPU implementation, no tests and no golden vectors, but its
// shaders compile as C++ over a small HLSL shim (oracle/ref/ -> oracle/_ref/libnrdref.so, built from the sources where they lie under /root/reference), and
// every pass of this restatement is compared with them on identical inputs (tests/test_ref_parity.py; fixtures recorded from them: tests/golden/ref_text_*.npz,
// tests/test_ref_golden.py); the reference's host sources compile too (oracle/ref/host/ -> oracle/_ref/libnrdhost.so) and pin the dispatch lists and constant blocks the passes are
// handed (tests/test_ref_host.py). What stays "parity unpinned" is MathLib alone (NVIDIA-RTX/MathLib, fetched unpinned at configure time -- reference
// CMakeLists.txt:118-127 -- and absent): definitions marked [ml] restate it from its public behaviour and from anchors inside the reference.
#pragma once

#include "hw_math.h"

#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

// ------------------------------------------------------------------------------------------------ scalars
inline uint32_t asuint(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    return u;
}
inline float asfloat(uint32_t u) {
    float x;
    memcpy(&x, &u, 4);
    return x;
}
inline float min(float a, float b) { return a < b ? a : b; }
inline float max(float a, float b) { return a > b ? a : b; }
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline float clamp(float x, float a, float b) { return min(max(x, a), b); }
inline int clamp(int x, int a, int b) { return min(max(x, a), b); }
inline float saturate(float x) { return min(max(x, 0.0f), 1.0f); }
inline float lerp(float a, float b, float t) { return a + (b - a) * t; }
inline float step(float edge, float x) { return x >= edge ? 1.0f : 0.0f; }
// the transcendental instructions of the device (oracle/hw_math.h)
inline float HwSqrt(float x) { return hwmath::HwSqrt(x); }
inline float HwRsq(float x) { return hwmath::HwRsq(x); }
inline float Rcp(float x) { return hwmath::HwRcp(x); }
inline float rcp(float x) { return hwmath::HwRcp(x); }
// the shaders' a / b. ORC_STRICT_IEEE (oracle/Makefile: liboracle_strict.so, compiled without contraction) is the build that is held against the reference's own shader
// text (oracle/_ref): there a division is a division, and what remains between the two is the re-association of a few expressions (DESIGN.md "Numerics")
#ifdef ORC_STRICT_IEEE
inline float Div(float a, float b) { return a / b; }
#else
inline float Div(float a, float b) { return a * Rcp(b); }
#endif
// strict build: HLSL leaves rsqrt's rounding open; the build that is compared with the reference's compiled text (oracle/ref/hlsl_shim.h: 1.0f / sqrtf(x)) takes the same
// two-rounding form, so that what remains between the two is association, not the choice of this one primitive
#ifdef ORC_STRICT_IEEE
inline float rsqrt(float x) { return 1.0f / sqrtf(x); }
#else
inline float rsqrt(float x) { return HwRsq(x); }
#endif
inline float frac(float x) { return x - floorf(x); }

// 2^x = v_exp_f32(1 + frac(x)) * 2^(floor(x) - 1), the instruction seeing [1, 2] only (oracle/hw_math.h)
inline float exp2(float x) {
    x = clamp(x, -125.0f, 125.0f);
    const float fl = floorf(x);
    const float t = 1.0f + (x - fl);
    return ldexpf(hwmath::HwExp2OnOneTwo(t), (int)fl - 1);
}

// log2(x) = e + v_log_f32(m), x = m * 2^e, m in [1, 2); zero, denormals, negative numbers and NaN return -126 (= log2 of the smallest normal); +inf 128
inline float log2(float x) {
    if (!(x >= 1.17549435e-38f))
        return -126.0f;
    const uint32_t bits = asuint(x);
    const int e = (int)(bits >> 23) - 127;
    const float m = asfloat((bits & 0x007FFFFFu) | 0x3F800000u);
    return float(e) + hwmath::HwLog2OnMantissa(m);
}

// round 5 (csrc/hip/nrdmath.h Exp2NonPos / SatExp2 / ExpNegAbs): 2^x for x <= 0 as 2 * v_exp_f32(x - 1); the strict build keeps the reference's plain forms
inline float Exp2NonPos(float x) {
#ifdef ORC_STRICT_IEEE
    return exp2(x);
#else
    const float t = x - 1.0f;
    return 2.0f * hwmath::HwExp2OnNegative(t);
#endif
}
inline float SatExp2(float x) {
#ifdef ORC_STRICT_IEEE
    return saturate(exp2(x));
#else
    return Exp2NonPos(min(x, 0.0f));
#endif
}
inline float exp(float x) { return exp2(x * 1.44269504f); }
inline float ExpNegAbs(float w) {
#ifdef ORC_STRICT_IEEE
    return exp(-fabsf(w));
#else
    const float t = -fabsf(w) * 1.44269504f - 1.0f; // ONE fused multiply-add (-ffp-contract=on), as on the device
    return 2.0f * hwmath::HwExp2OnNegative(t);
#endif
}
inline float log(float x) { return log2(x) * 0.69314718f; }
inline float pow(float x, float y) { return x <= 0.0f ? 0.0f : exp2(y * log2(x)); }

// atan: reduction to |t| <= tan(pi/8), degree-9 odd polynomial (Cephes atanf coefficients)
inline float atan(float x) {
    float a = fabsf(x);
    float base = 0.0f, t = a;
    if (a > 2.41421356f) {
        base = 1.57079633f;
        t = -Rcp(a);
    } else if (a > 0.41421356f) {
        base = 0.78539816f;
        t = Div(a - 1.0f, a + 1.0f);
    }
    float z = t * t;
    float p = 8.05374449538e-2f;
    p = p * z - 1.38776856032e-1f;
    p = p * z + 1.99777106478e-1f;
    p = p * z - 3.33329491539e-1f;
    float r = base + (p * z * t + t);
    return x < 0.0f ? -r : r;
}

// ------------------------------------------------------------------------------------------------ vectors
struct float2 {
    float x, y;
    float2() : x(0), y(0) {}
    float2(float a) : x(a), y(a) {}
    float2(float a, float b) : x(a), y(b) {}
};
struct float3 {
    float x, y, z;
    float3() : x(0), y(0), z(0) {}
    float3(float a) : x(a), y(a), z(a) {}
    float3(float a, float b, float c) : x(a), y(b), z(c) {}
    float2 xy() const { return float2(x, y); }
};
struct float4 {
    float x, y, z, w;
    float4() : x(0), y(0), z(0), w(0) {}
    float4(float a) : x(a), y(a), z(a), w(a) {}
    float4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    float4(float3 v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    float3 xyz() const { return float3(x, y, z); }
    float2 xy() const { return float2(x, y); }
    float2 zw() const { return float2(z, w); }
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
struct int2 {
    int x, y;
    int2() : x(0), y(0) {}
    int2(int a, int b) : x(a), y(b) {}
};
struct uint4 {
    uint32_t x, y, z, w;
};

#define ORC_OP2(op)                                                                       \
    inline float2 operator op(float2 a, float2 b) { return float2(a.x op b.x, a.y op b.y); } \
    inline float2 operator op(float2 a, float b) { return float2(a.x op b, a.y op b); }      \
    inline float2 operator op(float a, float2 b) { return float2(a op b.x, a op b.y); }
#define ORC_OP3(op)                                                                                    \
    inline float3 operator op(float3 a, float3 b) { return float3(a.x op b.x, a.y op b.y, a.z op b.z); } \
    inline float3 operator op(float3 a, float b) { return float3(a.x op b, a.y op b, a.z op b); }        \
    inline float3 operator op(float a, float3 b) { return float3(a op b.x, a op b.y, a op b.z); }
#define ORC_OP4(op)                                                                                                 \
    inline float4 operator op(float4 a, float4 b) { return float4(a.x op b.x, a.y op b.y, a.z op b.z, a.w op b.w); } \
    inline float4 operator op(float4 a, float b) { return float4(a.x op b, a.y op b, a.z op b, a.w op b); }          \
    inline float4 operator op(float a, float4 b) { return float4(a op b.x, a op b.y, a op b.z, a op b.w); }
ORC_OP2(+) ORC_OP2(-) ORC_OP2(*) ORC_OP2(/)
ORC_OP3(+) ORC_OP3(-) ORC_OP3(*) ORC_OP3(/)
ORC_OP4(+) ORC_OP4(-) ORC_OP4(*) ORC_OP4(/)
// operator/ above is the IEEE division of the front-end helpers (application side, as include/NRD.hip.h); the passes divide with Div
inline float2 Div(float2 a, float2 b) { return float2(Div(a.x, b.x), Div(a.y, b.y)); }
inline float2 Div(float2 a, float b) {
#ifdef ORC_STRICT_IEEE
    return float2(a.x / b, a.y / b);
#else
    float r = Rcp(b);
    return float2(a.x * r, a.y * r);
#endif
}
inline float3 Div(float3 a, float b) {
#ifdef ORC_STRICT_IEEE
    return float3(a.x / b, a.y / b, a.z / b);
#else
    float r = Rcp(b);
    return float3(a.x * r, a.y * r, a.z * r);
#endif
}
inline float4 Div(float4 a, float b) {
#ifdef ORC_STRICT_IEEE
    return float4(a.x / b, a.y / b, a.z / b, a.w / b);
#else
    float r = Rcp(b);
    return float4(a.x * r, a.y * r, a.z * r, a.w * r);
#endif
}
inline float4 Div(float4 a, float4 b) { return float4(Div(a.x, b.x), Div(a.y, b.y), Div(a.z, b.z), Div(a.w, b.w)); }
// x / c for a literal c: the arithmetic contract multiplies by the reciprocal constant (DESIGN.md "Numerics"), the reference text divides
#ifdef ORC_STRICT_IEEE
#define DivConst(x, c) ((x) / (c))
#else
#define DivConst(x, c) ((x) * (1.0f / (c)))
#endif
// a * s + c with ONE rounding per component (the overloaded operators round the product first): the accumulations of the tap loops, as on the device
inline float Mad(float a, float s, float c) { return a * s + c; }
inline float2 Mad(float2 a, float s, float2 c) { return float2(a.x * s + c.x, a.y * s + c.y); }
inline float2 Mad(float2 a, float2 s, float2 c) { return float2(a.x * s.x + c.x, a.y * s.y + c.y); }
inline float3 Mad(float3 a, float s, float3 c) { return float3(a.x * s + c.x, a.y * s + c.y, a.z * s + c.z); }
inline float4 Mad(float4 a, float s, float4 c) { return float4(a.x * s + c.x, a.y * s + c.y, a.z * s + c.z, a.w * s + c.w); }
// a * wa + b * wb (+ c * wc + d * wd): per component the very expression a scalar blend is written as, so vector and scalar blends fuse alike
inline float WSum(float a, float wa, float b, float wb) { return a * wa + b * wb; }
inline float4 WSum(float4 a, float wa, float4 b, float wb) { return float4(a.x * wa + b.x * wb, a.y * wa + b.y * wb, a.z * wa + b.z * wb, a.w * wa + b.w * wb); }
inline float WSum(float a, float wa, float b, float wb, float c, float wc, float d, float wd) { return a * wa + b * wb + c * wc + d * wd; }
inline float4 WSum(float4 a, float wa, float4 b, float wb, float4 c, float wc, float4 d, float wd) {
    return float4(a.x * wa + b.x * wb + c.x * wc + d.x * wd, a.y * wa + b.y * wb + c.y * wc + d.y * wd, a.z * wa + b.z * wb + c.z * wc + d.z * wd, a.w * wa + b.w * wb + c.w * wc + d.w * wd);
}
inline float4 Mad(float4 a, float4 s, float4 c) { return float4(a.x * s.x + c.x, a.y * s.y + c.y, a.z * s.z + c.z, a.w * s.w + c.w); }
inline float3 operator-(float3 a) { return float3(-a.x, -a.y, -a.z); }
inline float2 operator-(float2 a) { return float2(-a.x, -a.y); }
inline float2& operator+=(float2& a, float2 b) { return a = a + b; }
inline float3& operator+=(float3& a, float3 b) { return a = a + b; }
inline float4& operator+=(float4& a, float4 b) { return a = a + b; }
inline float2& operator*=(float2& a, float2 b) { return a = a * b; }
inline float2& operator*=(float2& a, float b) { return a = a * b; }
inline float3& operator*=(float3& a, float b) { return a = a * b; }
inline float3& operator*=(float3& a, float3 b) { return a = a * b; }
inline float4& operator*=(float4& a, float b) { return a = a * b; }
inline float4& operator*=(float4& a, float4 b) { return a = a * b; }
inline float2& operator/=(float2& a, float b) { return a = a / b; }
inline float3& operator/=(float3& a, float b) { return a = a / b; }
inline float4& operator-=(float4& a, float b) { return a = a - b; }

inline float dot(float2 a, float2 b) { return a.x * b.x + a.y * b.y; }
inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float dot(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
inline float sum(float4 a) { return a.x + a.y + a.z + a.w; } // dot( a, 1.0 )
inline float sum(float3 a) { return a.x + a.y + a.z; }
inline float length(float2 v) { return HwSqrt(dot(v, v)); }
inline float length(float3 v) { return HwSqrt(dot(v, v)); }
inline float3 normalize(float3 v) { return v * rsqrt(dot(v, v)); }
inline float3 cross(float3 a, float3 b) { return float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float3 reflect(float3 i, float3 n) { return i - n * (2.0f * dot(n, i)); }
inline float2 lerp(float2 a, float2 b, float t) { return float2(lerp(a.x, b.x, t), lerp(a.y, b.y, t)); }
inline float3 lerp(float3 a, float3 b, float t) { return float3(lerp(a.x, b.x, t), lerp(a.y, b.y, t), lerp(a.z, b.z, t)); }
inline float4 lerp(float4 a, float4 b, float t) { return float4(lerp(a.x, b.x, t), lerp(a.y, b.y, t), lerp(a.z, b.z, t), lerp(a.w, b.w, t)); }
inline float2 saturate(float2 v) { return float2(saturate(v.x), saturate(v.y)); }
inline float2 floor(float2 v) { return float2(floorf(v.x), floorf(v.y)); }
inline float2 abs(float2 v) { return float2(fabsf(v.x), fabsf(v.y)); }
inline float3 abs(float3 v) { return float3(fabsf(v.x), fabsf(v.y), fabsf(v.z)); }
inline float4 abs(float4 v) { return float4(fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w)); }
inline float abs(float v) { return fabsf(v); }
inline float3 max(float3 v, float s) { return float3(max(v.x, s), max(v.y, s), max(v.z, s)); }
inline float2 max(float2 a, float2 b) { return float2(max(a.x, b.x), max(a.y, b.y)); }
inline float2 min(float2 a, float2 b) { return float2(min(a.x, b.x), min(a.y, b.y)); }
inline float3 step(float3 edge, float x) { return float3(step(edge.x, x), step(edge.y, x), step(edge.z, x)); }
inline float4 step(float4 edge, float4 x) { return float4(step(edge.x, x.x), step(edge.y, x.y), step(edge.z, x.z), step(edge.w, x.w)); }
inline float2 step(float2 edge, float2 x) { return float2(step(edge.x, x.x), step(edge.y, x.y)); }

// 4x4 matrix as stored in the constant buffers: 16 floats, column-major
struct float4x4 {
    float m[16];
    float at(int row, int col) const { return m[col * 4 + row]; }
};

} // namespace orc
