// Device math for the NRD pass chain: HLSL intrinsics with pinned definitions, the restated NVIDIA-RTX/MathLib subset the shaders call
// ("ml.hlsli" -- NOT vendored in the reference, fetched unpinned at configure time, reference CMakeLists.txt:118-127), and the NRD.hlsli
// front-end / back-end codecs.
//
// Numerics contract (DESIGN.md "Numerics"): ONE arithmetic, fast on gfx950 and reproducible bit for bit on a CPU.
//   * + - * and fused multiply-add are IEEE-754 binary32, correctly rounded on both machines. Which multiply-adds are fused is decided by the
//     SOURCE, not by an optimiser: this file is compiled with -ffp-contract=on (ISO C "FP_CONTRACT ON": `a * b + c` written as ONE expression is a
//     single fma; nothing is fused across statements, function calls or overloaded vector operators). The CPU oracle is compiled by the same
//     front end with the same flag.
//   * 1/x is the hardware's v_rcp_f32, a / b is a * v_rcp_f32(b) (Div); sqrt and 1/sqrt are v_sqrt_f32 / v_rsq_f32; 2^x and log2 x go through
//     v_exp_f32 / v_log_f32 on ONE binade with the range reduction spelled out (Exp2 / Log2 below). Each instruction is within 1 ulp of the
//     correctly rounded result; the oracle reproduces it exactly from deviation tables measured on the device (oracle/hw_math.h).
//   * No IEEE division, no libm call, no reassociation, fp32 denormals kept (the five instructions flush theirs, and so does the oracle).
// The CPU oracle (oracle/) restates the same definitions independently; the two are compared bit for bit, because a loose tolerance would be
// amplified by the chain's many thresholds (tap snapping, `> 11.5` material tests, fp16 ties, 6-bit counters).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../common/encoding.h"

namespace nrdhip {

#ifndef NRD_D
#    define NRD_D __device__ __forceinline__
#endif

// ------------------------------------------------------------------------------------------------ constants
#define NRD_FP16_MAX 65504.0f
#define NRD_PI 3.14159265358979323846f
#define NRD_EPS 1e-6f
#define NRD_INF 1e6f
// reference Common.hlsli:76-85
#if NRD_NORMAL_ENCODING < NRD_NORMAL_ENCODING_R10G10B10A2_UNORM
#define NRD_NORMAL_ENCODING_ERROR (1.5f / 255.0f)
#elif NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_R10G10B10A2_UNORM
#define NRD_NORMAL_ENCODING_ERROR (0.75f / 255.0f)
#else
#define NRD_NORMAL_ENCODING_ERROR (0.5f / 255.0f)
#endif
#define NRD_ROUGHNESS_SENSITIVITY 0.01f
#define NRD_EXP_WEIGHT_DEFAULT_SCALE 3.0f
#define NRD_CATROM_SHARPNESS 0.5f
#define NRD_DISOCCLUSION_THRESHOLD 0.02f
#define NRD_MAX_PERCENT_OF_LOBE_VOLUME 0.75f
#define NRD_CURVATURE_Z_THRESHOLD 0.1f

// ------------------------------------------------------------------------------------------------ scalar intrinsics
#if defined(NRD_NATIVE_MINMAX) && NRD_NATIVE_MINMAX
// EXPERIMENT (tools/build_variant.py, not the product, not mirrored by the oracle): v_min_f32 / v_max_f32 -- what an HLSL compiler emits for min / max. Differs from the
// compare + select below only when an operand is NaN (the other operand is returned) or the operands are zeros of opposite sign; priced in DESIGN.md section 8.
NRD_D float Min(float a, float b) { return __builtin_fminf(a, b); }
NRD_D float Max(float a, float b) { return __builtin_fmaxf(a, b); }
#else
NRD_D float Min(float a, float b) { return a < b ? a : b; }
NRD_D float Max(float a, float b) { return a > b ? a : b; }
#endif
NRD_D float Clamp(float x, float a, float b) { return Min(Max(x, a), b); }
NRD_D float Sat(float x) { return Min(Max(x, 0.0f), 1.0f); }
NRD_D float Lerp(float a, float b, float t) { return a + (b - a) * t; }
NRD_D float Step(float edge, float x) { return x >= edge ? 1.0f : 0.0f; }
// v_rcp_f32. The empty asm keeps the argument opaque to the optimiser: LLVM folds this intrinsic of a compile-time constant into the CORRECTLY
// ROUNDED reciprocal, which is not what the instruction returns for 11 % of the mantissas (profiles/r03_a_hw_tables_report.txt) -- a constant
// that reaches a division after inlining would then differ from the oracle by one ulp. The other four intrinsics are not folded.
#ifndef NRD_OPAQUE_VALUE
#define NRD_OPAQUE_VALUE(x) asm("" : "+v"(x))
#endif
NRD_D float Rcp(float x) {
    NRD_OPAQUE_VALUE(x);
    return __builtin_amdgcn_rcpf(x);
}
// a / b of the contract: one multiplication by the hardware reciprocal (2 instructions instead of the ~11 of a correctly rounded division).
// Where this is NOT the IEEE quotient (the oracle reproduces every case bit for bit: tests/test_numerics.py "div_edge_cases"):
//   * v_rcp_f32 flushes: |b| > 2^126 gives rcp = +-0 and a quotient of 0 (IEEE: a small non-zero number); a denormal b gives rcp = +-inf, so
//     a / b = +-inf for a != 0 (IEEE agrees up to overflow) and 0 / denormal = 0 * inf = NaN (IEEE: 0);
//   * Div(x, x) is 1 only up to one ulp (x * rcp(x) rounds twice);
//   * the call sites of the passes keep their denominators away from both ends (Max(.., eps), +NRD_EPS, PositiveRcp), as the reference's shaders do for the
//     same reason -- HLSL's a / b is a * rcp(b) on every GPU the reference runs on, with the same flush behaviour.
NRD_D float Div(float a, float b) { return a * Rcp(b); }
// sqrt and 1/sqrt: v_sqrt_f32, v_rsq_f32 (within 1 ulp of the correctly rounded result, denormals flushed)
NRD_D float Sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
NRD_D float Rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
NRD_D float Abs(float x) { return fabsf(x); }
NRD_D float Floor(float x) { return floorf(x); }
NRD_D float Frac(float x) { return x - floorf(x); }
NRD_D uint32_t AsUint(float x) { return __float_as_uint(x); }
NRD_D float AsFloat(uint32_t x) { return __uint_as_float(x); }

// ------------------------------------------------------------------------------------------------ transcendentals
// 2^x = v_exp_f32(1 + frac(x)) * 2^(floor(x) - 1): the instruction only ever sees an argument in [1, 2] (x - floor(x) is exact; adding 1 rounds it to
// 2^-23, a relative error of 2^-24 ln 2 in the result). v_exp_f32 of the raw argument would be one instruction shorter, but for negative and for
// small arguments it is NOT this reduced form (it keeps extra internal bits) and cannot be tabulated for the oracle.
NRD_D float Exp2(float x) {
    x = Clamp(x, -125.0f, 125.0f);
    const float fl = floorf(x);
    const float t = 1.0f + (x - fl);
    return ldexpf(__builtin_amdgcn_exp2f(t), (int)fl - 1);
}
// log2(x) = e + v_log_f32(m), x = m * 2^e with m in [1, 2) (one fp32 addition). Anything that is not a positive normal number -- zero, denormals, negative
// numbers, NaN -- returns -126 = log2 of the smallest normal (straight-line code: the result is selected, not branched to; +inf returns 128)
NRD_D float Log2(float x) {
    const uint32_t bits = AsUint(x);
    const int e = (int)(bits >> 23) - 127;
    const float m = AsFloat((bits & 0x007FFFFFu) | 0x3F800000u);
    const float r = float(e) + __builtin_amdgcn_logf(m);
    return x >= 1.17549435e-38f ? r : -126.0f;
}

// 2^x for x <= 0 (round 5): 2 * v_exp_f32(x - 1) -- one subtraction in front of the instruction instead of the eight-instruction reduction above. The instruction then only
// ever sees an argument <= -1, where it is a sign-magnitude function of ONE binade (tools/hw_exp_neg.hip, profiles/r05_b_hw_exp_neg_report.txt: v_exp_f32(-w) ==
// v_exp_f32(-(1 + frac(w))) * 2^-(floor(w) - 1) for every w >= 1, 0 mismatches over six binades x 2^23 mantissas, results below 2^-126 flushed to 0), so the oracle reproduces
// it from one more deviation table over [-2, -1] (oracle/hw_exp2neg.i8.z). x - 1 rounds x to 2^-24 for |x| < 1 -- the same precision class as 1 + frac(x) above.
// Callers guarantee x <= 0 (or -0, NaN): a positive argument would leave the measured domain (the oracle aborts on it).
NRD_D float Exp2NonPos(float x) {
    const float t = x - 1.0f;
    return 2.0f * __builtin_amdgcn_exp2f(t);
}
// saturate(2^x) for any x: 2^min(x, 0) (2^x <= 1 for x <= 0, and saturate(2^x) = 1 = 2^0 for x > 0)
NRD_D float SatExp2(float x) { return Exp2NonPos(Min(x, 0.0f)); }
// e^-|w|: the argument of the instruction is ONE fused multiply-add, fma(-|w|, log2 e, -1) (sign and magnitude are free source modifiers)
NRD_D float ExpNegAbs(float w) {
    const float t = -Abs(w) * 1.44269504f - 1.0f;
    return 2.0f * __builtin_amdgcn_exp2f(t);
}
NRD_D float Exp(float x) { return Exp2(x * 1.44269504f); }
NRD_D float Log(float x) { return Log2(x) * 0.69314718f; }
// x^y for x >= 0 (0^y = 0 for the y > 0 the chain uses)
NRD_D float Pow(float x, float y) { return x <= 0.0f ? 0.0f : Exp2(y * Log2(x)); }

// atan(x): Cephes-style reduction to |t| <= tan(pi/8) and a degree-9 odd polynomial (abs. error ~1e-7)
NRD_D float Atan(float x) {
    float a = Abs(x);
    float base = 0.0f;
    float t = a;
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
NRD_D float2 F2(float x, float y) { return make_float2(x, y); }
NRD_D float3 F3(float x, float y, float z) { return make_float3(x, y, z); }
NRD_D float3 F3(float x) { return make_float3(x, x, x); }
NRD_D float4 F4(float x, float y, float z, float w) { return make_float4(x, y, z, w); }
NRD_D float4 F4(float3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
NRD_D float4 F4(float x) { return make_float4(x, x, x, x); }
NRD_D float3 Xyz(float4 v) { return make_float3(v.x, v.y, v.z); }

NRD_D float2 operator+(float2 a, float2 b) { return F2(a.x + b.x, a.y + b.y); }
NRD_D float2 operator-(float2 a, float2 b) { return F2(a.x - b.x, a.y - b.y); }
NRD_D float2 operator*(float2 a, float2 b) { return F2(a.x * b.x, a.y * b.y); }
NRD_D float2 operator*(float2 a, float b) { return F2(a.x * b, a.y * b); }
NRD_D float2 Div(float2 a, float2 b) { return F2(Div(a.x, b.x), Div(a.y, b.y)); }
NRD_D float2 Div(float2 a, float b) { float r = Rcp(b); return F2(a.x * r, a.y * r); }
NRD_D float2 operator+(float2 a, float b) { return F2(a.x + b, a.y + b); }
NRD_D float2 operator-(float2 a, float b) { return F2(a.x - b, a.y - b); }

NRD_D float3 operator+(float3 a, float3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
NRD_D float3 operator-(float3 a, float3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
NRD_D float3 operator-(float3 a) { return F3(-a.x, -a.y, -a.z); }
NRD_D float3 operator*(float3 a, float3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
NRD_D float3 operator*(float3 a, float b) { return F3(a.x * b, a.y * b, a.z * b); }
NRD_D float3 Div(float3 a, float b) { float r = Rcp(b); return F3(a.x * r, a.y * r, a.z * r); }

NRD_D float4 operator+(float4 a, float4 b) { return F4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
NRD_D float4 operator-(float4 a, float4 b) { return F4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
NRD_D float4 operator*(float4 a, float4 b) { return F4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
NRD_D float4 operator*(float4 a, float b) { return F4(a.x * b, a.y * b, a.z * b, a.w * b); }
NRD_D float4 Div(float4 a, float b) { float r = Rcp(b); return F4(a.x * r, a.y * r, a.z * r, a.w * r); }
NRD_D float4 Div(float4 a, float4 b) { return F4(Div(a.x, b.x), Div(a.y, b.y), Div(a.z, b.z), Div(a.w, b.w)); }
NRD_D float4 operator-(float4 a, float b) { return F4(a.x - b, a.y - b, a.z - b, a.w - b); }
// a * s + c with ONE rounding per component. The overloaded operators above cannot fuse (`c + a * s` rounds the product first: the contraction rule of
// this build looks at expressions, not through function calls), so the accumulations of the tap loops spell the fused form out -- on both sides.
NRD_D float Mad(float a, float s, float c) { return a * s + c; }
NRD_D float2 Mad(float2 a, float s, float2 c) { return F2(a.x * s + c.x, a.y * s + c.y); }
NRD_D float2 Mad(float2 a, float2 s, float2 c) { return F2(a.x * s.x + c.x, a.y * s.y + c.y); }
NRD_D float3 Mad(float3 a, float s, float3 c) { return F3(a.x * s + c.x, a.y * s + c.y, a.z * s + c.z); }
NRD_D float4 Mad(float4 a, float s, float4 c) { return F4(a.x * s + c.x, a.y * s + c.y, a.z * s + c.z, a.w * s + c.w); }
// a * wa + b * wb (+ c * wc + d * wd): per component the very expression a scalar blend is written as, so vector and scalar blends fuse alike
NRD_D float WSum(float a, float wa, float b, float wb) { return a * wa + b * wb; }
NRD_D float4 WSum(float4 a, float wa, float4 b, float wb) { return F4(a.x * wa + b.x * wb, a.y * wa + b.y * wb, a.z * wa + b.z * wb, a.w * wa + b.w * wb); }
NRD_D float WSum(float a, float wa, float b, float wb, float c, float wc, float d, float wd) { return a * wa + b * wb + c * wc + d * wd; }
NRD_D float4 WSum(float4 a, float wa, float4 b, float wb, float4 c, float wc, float4 d, float wd) {
    return F4(a.x * wa + b.x * wb + c.x * wc + d.x * wd, a.y * wa + b.y * wb + c.y * wc + d.y * wd, a.z * wa + b.z * wb + c.z * wc + d.z * wd, a.w * wa + b.w * wb + c.w * wc + d.w * wd);
}
NRD_D float4 Mad(float4 a, float4 s, float4 c) { return F4(a.x * s.x + c.x, a.y * s.y + c.y, a.z * s.z + c.z, a.w * s.w + c.w); }

// component-wise select: `cond ? a : b` on two vector LVALUES is an lvalue conditional, which the compiler implements as a
// select between the addresses of two stack copies (scratch memory traffic); selecting per component keeps it in registers
NRD_D float2 Select(bool c, float2 a, float2 b) { return F2(c ? a.x : b.x, c ? a.y : b.y); }
NRD_D float3 Select(bool c, float3 a, float3 b) { return F3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }
NRD_D float4 Select(bool c, float4 a, float4 b) { return F4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w); }

NRD_D float Dot(float2 a, float2 b) { return a.x * b.x + a.y * b.y; }
NRD_D float Dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
NRD_D float Dot(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
NRD_D float Sum(float4 a) { return a.x + a.y + a.z + a.w; } // dot( a, 1.0 )
NRD_D float Length(float2 v) { return Sqrt(Dot(v, v)); }
NRD_D float Length(float3 v) { return Sqrt(Dot(v, v)); }
NRD_D float LengthSquared(float3 v) { return Dot(v, v); }
NRD_D float LengthSquared(float2 v) { return Dot(v, v); }
NRD_D float3 Normalize(float3 v) { return v * Rsqrt(Dot(v, v)); }
NRD_D float3 Cross(float3 a, float3 b) { return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
NRD_D float3 Reflect(float3 i, float3 n) { return i - n * (2.0f * Dot(n, i)); }
NRD_D float3 Lerp(float3 a, float3 b, float t) { return F3(Lerp(a.x, b.x, t), Lerp(a.y, b.y, t), Lerp(a.z, b.z, t)); }
NRD_D float4 Lerp(float4 a, float4 b, float t) { return F4(Lerp(a.x, b.x, t), Lerp(a.y, b.y, t), Lerp(a.z, b.z, t), Lerp(a.w, b.w, t)); }
NRD_D float2 Lerp(float2 a, float2 b, float t) { return F2(Lerp(a.x, b.x, t), Lerp(a.y, b.y, t)); }
NRD_D float2 Sat(float2 v) { return F2(Sat(v.x), Sat(v.y)); }
NRD_D float2 Floor(float2 v) { return F2(floorf(v.x), floorf(v.y)); }
NRD_D float2 Abs(float2 v) { return F2(Abs(v.x), Abs(v.y)); }
NRD_D float3 Abs(float3 v) { return F3(Abs(v.x), Abs(v.y), Abs(v.z)); }
NRD_D float4 Abs(float4 v) { return F4(Abs(v.x), Abs(v.y), Abs(v.z), Abs(v.w)); }
NRD_D float3 Max(float3 v, float s) { return F3(Max(v.x, s), Max(v.y, s), Max(v.z, s)); }
NRD_D float4 Step(float4 edge, float4 x) { return F4(Step(edge.x, x.x), Step(edge.y, x.y), Step(edge.z, x.z), Step(edge.w, x.w)); }
NRD_D float3 Step(float3 edge, float x) { return F3(Step(edge.x, x), Step(edge.y, x), Step(edge.z, x)); }

// ------------------------------------------------------------------------------------------------ Math::
NRD_D float LinearStep(float a, float b, float x) { return Sat(Div(x - a, b - a)); }
NRD_D float SmoothStep01(float x) {
    x = Sat(x);
    return x * x * (3.0f - 2.0f * x);
}
NRD_D float SmoothStep(float a, float b, float x) { return SmoothStep01(LinearStep(a, b, x)); }
NRD_D float Sqrt01(float x) { return Sqrt(Sat(x)); }
// saturate(x)^y for y >= 0 (every call site: constant exponents, and a product of factors in [0, 1] x [1, 32]): log2 of a number in (0, 1] is <= 0, so the product
// goes to Exp2NonPos; the min guards the domain of the instruction's table, not the mathematics
NRD_D float Pow01(float x, float y) {
    x = Sat(x);
    return x <= 0.0f ? 0.0f : Exp2NonPos(Min(y * Log2(x), 0.0f));
}
NRD_D float PositiveRcp(float x) { return Rcp(Max(x, 1e-15f)); }
NRD_D float AcosApprox(float x) { return 1.41421356f * Sqrt(Sat(1.0f - x)); } // sqrt(2) * sqrt(saturate(1 - x))
NRD_D float Pow5(float x) {                                                  // BRDF::Pow5 = (1 - x)^5 on saturated input
    float t = Sat(1.0f - x);
    float t2 = t * t;
    return t2 * t2 * t;
}

// ------------------------------------------------------------------------------------------------ Geometry::
// 4x4 matrices are 16 floats, column-major (csrc/common/pass_constants.h)
NRD_D float3 RotateVector(const float* m, float3 v) { // (float3x3)M * v
    return F3(m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z, m[2] * v.x + m[6] * v.y + m[10] * v.z);
}
NRD_D float3 RotateVectorInverse(const float* m, float3 v) { // transpose((float3x3)M) * v
    return F3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
NRD_D float3 AffineTransform(const float* m, float3 p) {
    return F3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13], m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
NRD_D float4 ProjectiveTransform(const float* m, float3 p) {
    return F4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13], m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
        m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}
// clip -> uv with y flipped; points behind the camera are sent far off-screen
NRD_D float2 GetScreenUv(const float* worldToClip, float3 X) {
    float4 clip = ProjectiveTransform(worldToClip, X);
    float2 uv = F2(Div(clip.x, clip.w) * 0.5f + 0.5f, Div(clip.y, clip.w) * -0.5f + 0.5f);
    return clip.w < 0.0f ? F2(99999.0f, 99999.0f) : uv;
}
// inverse of the above for a known viewZ: Xv.xy = (uv * frustum.zw + frustum.xy) * viewZ (perspective)
NRD_D float3 ReconstructViewPosition(float2 uv, float4 frustum, float viewZ, float orthoMode) {
    float s = viewZ * (1.0f - Abs(orthoMode)) + orthoMode;
    return F3((uv.x * frustum.z + frustum.x) * s, (uv.y * frustum.w + frustum.y) * s, viewZ);
}
// rotator = (cos, sin, -sin, cos): v' = v.x * r.xz + v.y * r.yw
NRD_D float2 RotateVector(float4 r, float2 v) { return F2(v.x * r.x + v.y * r.y, v.x * r.z + v.y * r.w); }
NRD_D float4 ScaleRotator(float4 r, float2 s) { return F4(r.x * s.x, r.y * s.x, r.z * s.y, r.w * s.y); }
// branch-free orthonormal basis (Duff et al. 2017): T, B such that {T, B, N} is right-handed
NRD_D void GetBasis(float3 N, float3& T, float3& B) {
    float sz = N.z >= 0.0f ? 1.0f : -1.0f;
    float a = Rcp(sz + N.z);
    float ya = N.y * a;
    float b = N.x * ya;
    float c = N.x * sz;
    T = F3(c * N.x * a - 1.0f, sz * b, c);
    B = F3(b, N.y * ya - sz, N.y);
}

// ------------------------------------------------------------------------------------------------ Color:: / Packing:: / Sequence::
NRD_D float Luminance(float3 c) { return c.x * 0.2126f + c.y * 0.7152f + c.z * 0.0722f; }
// MathLib BRDF::ConvertBaseColorMetalnessToAlbedoRf0 (dielectric Rf0 = 0.04) and BRDF::EnvironmentTerm_Rtg (reference NRD.hlsli:490-517:
// "Ray Tracing Gems" ch. 32 eq. 4, GGX VNDF + Schlick); the polynomial rows are summed left to right, rcp is an exact division
NRD_D void ConvertBaseColorMetalnessToAlbedoRf0(float3 baseColor, float metalness, float3& albedo, float3& Rf0) {
    float k = Sat(1.0f - metalness);
    albedo = F3(baseColor.x * k, baseColor.y * k, baseColor.z * k);
    Rf0 = F3(0.04f + (baseColor.x - 0.04f) * metalness, 0.04f + (baseColor.y - 0.04f) * metalness, 0.04f + (baseColor.z - 0.04f) * metalness);
}
NRD_D float3 EnvironmentTerm_Rtg(float3 Rf0, float NoV, float roughness) {
    float m = Sat(roughness * roughness);
    float x1 = NoV, x2 = NoV * NoV, x3 = NoV * x2;
    float y1 = m, y3 = m * (m * m);
    float biasNum = (0.99044f + -1.28514f * x1) + (1.29678f + -0.755907f * x1) * y1;
    float biasDen = (1.0f + 2.92338f * x1 + 59.4188f * x3) + (20.3225f + -27.0302f * x1 + 222.592f * x3) * y1 + (121.563f + 626.13f * x1 + 316.627f * x3) * y3;
    float scaleNum = (0.0365463f + 3.32707f * x1) + (9.0632f + -9.04756f * x1) * y1;
    float scaleDen = (1.0f + 3.59685f * x2 + -1.36772f * x3) + (9.04401f + -16.3174f * x2 + 9.22949f * x3) * y1 + (5.56589f + 19.7886f * x2 + -20.2123f * x3) * y3;
    float bias = biasNum * Rcp(Max(biasDen, 1e-6f));
    float scale = scaleNum * Rcp(Max(scaleDen, 1e-6f));
    return F3(Sat(Rf0.x * scale + bias), Sat(Rf0.y * scale + bias), Sat(Rf0.z * scale + bias));
}
NRD_D float ColorClamp(float m1, float sigma, float x) { return Clamp(x, m1 - sigma, m1 + sigma); }
NRD_D float3 LinearToYCoCg(float3 c) { // reference NRD.hlsli:356-363
    return F3(c.x * 0.25f + c.y * 0.5f + c.z * 0.25f, c.x * 0.5f + c.y * 0.0f + c.z * -0.5f, c.x * -0.25f + c.y * 0.5f + c.z * -0.25f);
}
NRD_D float3 YCoCgToLinear(float3 c) { // reference NRD.hlsli:365-375
    float t = c.x - c.z;
    return F3(Max(t + c.y, 0.0f), Max(c.x + c.z, 0.0f), Max(t - c.y, 0.0f));
}
NRD_D uint32_t CheckerBoard(uint32_t x, uint32_t y, uint32_t frameIndex) { return ((x ^ y) ^ frameIndex) & 1u; }
// reference Common.hlsli:297-307: a tap landing on a pixel without data moves one pixel left / right, alternating with the tap counter
// ("pos" = a pixel centre; mode 2 = checkerboard off)
NRD_D float2 ApplyCheckerboardShift(float2 pos, uint32_t mode, uint32_t counter, uint32_t frameIndex) {
    uint32_t checkerboard = CheckerBoard((uint32_t)(pos.x + 16384.0f), (uint32_t)(pos.y + 16384.0f), frameIndex);
    float shift = (counter & 1u) == 0 ? -1.0f : 1.0f;
    pos.x += shift * ((checkerboard != mode && mode != 2u) ? 1.0f : 0.0f);
    return pos;
}

// NRD_MATHLIB_BAYER_REVERSEBITS (default 0): MathLib is not vendored in the reference tree, so Sequence::Bayer4x4ui is a restatement; the round-5 reviewer recalls a MathLib
// default ML_BAYER_REVERSEBITS that advances the dither index by ReverseBits4( frameIndex ) instead of frameIndex. Unverifiable here; 1 selects that alternative in every place the
// function is restated (product host + device, oracle, both MathLib stand-ins under oracle/ref) -- it changes the dither PHASE per frame, nothing else (INTEGRATION.md "MathLib").
#ifndef NRD_MATHLIB_BAYER_REVERSEBITS
#define NRD_MATHLIB_BAYER_REVERSEBITS 0
#endif
NRD_D uint32_t BayerFrameOffset(uint32_t frameIndex) {
#if NRD_MATHLIB_BAYER_REVERSEBITS
    const uint32_t v = frameIndex & 0xFu;
    return ((v & 1u) << 3) | ((v & 2u) << 1) | ((v & 4u) >> 1) | ((v & 8u) >> 3); // ReverseBits4
#else
    return frameIndex;
#endif
}
NRD_D uint32_t Bayer4x4ui(uint32_t x, uint32_t y, uint32_t frameIndex) {
    x &= 3u;
    y &= 3u;
    uint32_t a = 2068378560u * (1u - (x >> 1)) + 1500172770u * (x >> 1);
    uint32_t b = (y + ((x & 1u) << 2)) << 2;
    return ((a >> b) + BayerFrameOffset(frameIndex)) & 0xFu;
}
// round 5: i / 16, "RESULT: [0; 1)" -- the form both the builder's and the round-4 reviewer's recollection of NVIDIA-RTX/MathLib agree on (until then (i + 0.5) / 16; MathLib is not vendored: unpinned either way)
NRD_D float Bayer4x4(uint32_t x, uint32_t y, uint32_t frameIndex) { return float(Bayer4x4ui(x, y, frameIndex)) * 0.0625f; }

// Rng::Hash -- OUR definition (MathLib's is unavailable): PCG-style state seeded from (pixel, frame)
struct RngHash {
    uint32_t state;
    NRD_D void Initialize(uint32_t x, uint32_t y, uint32_t frameIndex) {
        uint32_t s = x * 0x9E3779B1u ^ (y * 0x85EBCA77u + 0xC2B2AE3Du) ^ (frameIndex * 0x27D4EB2Fu + 0x165667B1u);
        s ^= s >> 15;
        s *= 0x2C1B3C6Du;
        s ^= s >> 12;
        s *= 0x297A2D39u;
        s ^= s >> 15;
        state = s;
    }
    NRD_D uint32_t Next() {
        state = state * 747796405u + 2891336453u;
        uint32_t w = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
        return (w >> 22) ^ w;
    }
    NRD_D float GetFloat() { return float(Next() >> 8) * (1.0f / 16777216.0f); }
    NRD_D float2 GetFloat2() {
        float a = GetFloat();
        float b = GetFloat();
        return F2(a, b);
    }
};

// ------------------------------------------------------------------------------------------------ Filtering::
struct Bilinear {
    float2 origin;  // integer texel coords of the top-left tap
    float2 weights; // fractional position inside the 2x2 footprint
};
NRD_D Bilinear GetBilinearFilter(float2 uv, float2 texSize) {
    float2 t = uv * texSize - 0.5f;
    Bilinear r;
    r.origin = Floor(t);
    r.weights = t - r.origin;
    return r;
}
// order: (0,0) (1,0) (0,1) (1,1)
NRD_D float4 GetBilinearCustomWeights(Bilinear f, float4 customWeights) {
    float2 oneMinus = F2(1.0f - f.weights.x, 1.0f - f.weights.y);
    float4 w = customWeights;
    w.x *= oneMinus.x * oneMinus.y;
    w.y *= f.weights.x * oneMinus.y;
    w.z *= oneMinus.x * f.weights.y;
    w.w *= f.weights.x * f.weights.y;
    return w;
}
NRD_D float ApplyBilinearFilter(float s00, float s10, float s01, float s11, Bilinear f) {
    return Lerp(Lerp(s00, s10, f.weights.x), Lerp(s01, s11, f.weights.x), f.weights.y);
}
NRD_D float ApplyBilinearCustomWeights(float s00, float s10, float s01, float s11, float4 w) {
    float sum = w.x + w.y + w.z + w.w;
    float r = s00 * w.x + s10 * w.y + s01 * w.z + s11 * w.w;
    return sum < 0.0001f ? 0.0f : Div(r, sum);
}
// top-left texel of the 4x4 Catmull-Rom footprint (reference REBLUR_TemporalAccumulation.hlsli:152-171)
NRD_D float2 GetCatmullRomOrigin(float2 uv, float2 texSize) {
    float2 t = uv * texSize - 0.5f;
    return Floor(t) - 1.0f;
}
NRD_D float GetModifiedRoughnessFromNormalVariance(float linearRoughness, float3 nonNormalizedAverageNormal) {
    float l = Length(nonNormalizedAverageNormal);
    float kappa = Div(Sat(1.0f - l * l), Max(l * (3.0f - l * l), 1e-15f));
    return Sqrt(Sat(linearRoughness * linearRoughness + kappa));
}

// ------------------------------------------------------------------------------------------------ ImportanceSampling::
NRD_D float GetSpecularLobeTanHalfAngle(float linearRoughness, float percentOfVolume) {
    float r = Sat(linearRoughness);
    float p = Sat(percentOfVolume);
    float m = r * r;
    return m * Sqrt(Div(p, 1.0f - p + NRD_EPS)); // the "fixed" MathLib form (see reference Reblur.cpp:384, RELAX_Common.hlsli:113-122)
}
NRD_D float GetSpecularDominantFactor(float NoV, float linearRoughness) { // G2 fit, reference NRD.hlsli:386-392
    float a = 0.298475f * Log(39.4115f - 39.0029f * linearRoughness);
    float f = Pow01(1.0f - NoV, 10.8649f) * (1.0f - a) + a;
    return Sat(f);
}
NRD_D float4 GetSpecularDominantDirection(float3 N, float3 V, float linearRoughness) {
    float NoV = Abs(Dot(N, V));
    float f = GetSpecularDominantFactor(NoV, linearRoughness);
    float3 R = Reflect(-V, N);
    float3 D = Normalize(Lerp(N, R, f));
    return F4(D, f);
}

// ------------------------------------------------------------------------------------------------ NRD.hlsli codecs
NRD_D float3 SafeNormalize(float3 v) { return v * Rsqrt(Dot(v, v) + 1e-9f); }
NRD_D float3 DecodeUnitVectorOct(float2 p) { // unsigned input, not normalised; reference NRD.hlsli:333-343
    float px = p.x * 2.0f - 1.0f, py = p.y * 2.0f - 1.0f;
    float3 n = F3(px, py, 1.0f - Abs(px) - Abs(py));
    float t = Sat(-n.z);
    n.x -= t * (Step(0.0f, n.x) * 2.0f - 1.0f);
    n.y -= t * (Step(0.0f, n.y) * 2.0f - 1.0f);
    return n;
}
// the IN_NORMAL_ROUGHNESS texel as the texture unit returns it -> (N, linear roughness), materialID; reference NRD.hlsli:600-637
// (R10G10B10A2_UNORM: oct normal, roughness, 2-bit material id; the four RGBA encodings: the normal in xyz -- biased for the UNORM ones --, roughness in w, no material id)
NRD_D float4 UnpackNormalAndRoughness(float4 p, float& materialID) {
#if NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_R10G10B10A2_UNORM
    float3 n = DecodeUnitVectorOct(F2(p.x, p.y));
    float r = p.z;
    materialID = p.w * 3.0f;
#else
#if NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_RGBA8_UNORM || NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_RGBA16_UNORM
    float3 n = F3(p.x * 2.0f - 1.0f, p.y * 2.0f - 1.0f, p.z * 2.0f - 1.0f);
#else
    float3 n = F3(p.x, p.y, p.z);
#endif
    float r = p.w;
    materialID = 0.0f;
#endif
    n = SafeNormalize(n);
#if NRD_ROUGHNESS_ENCODING == NRD_ROUGHNESS_ENCODING_SQRT_LINEAR
    r *= r;
#elif NRD_ROUGHNESS_ENCODING == NRD_ROUGHNESS_ENCODING_SQ_LINEAR
    r = Sqrt(Sat(r));
#endif
    return F4(n, r);
}
NRD_D float4 UnpackNormalAndRoughness(float4 p) {
    float unused;
    return UnpackNormalAndRoughness(p, unused);
}
NRD_D float GetHitDistanceNormalization(float viewZ, float4 hitDistParams, float roughness) { // reference NRD.hlsli:520-523
    return (hitDistParams.x + Abs(viewZ) * hitDistParams.y) * Lerp(1.0f, hitDistParams.z, SatExp2(hitDistParams.w * roughness * roughness));
}

} // namespace nrdhip
