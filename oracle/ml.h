// ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/hlsl.h).
//
// Restatement of the library layers the NRD shaders sit on:
//   [ml]   NVIDIA-RTX/MathLib "ml.hlsli" namespaces (Math, Geometry, Filtering, Color, Packing, Sequence, Rng,
//          ImportanceSampling, BRDF) -- not present in /root/reference; restated, see SURVEY.md section 8c
//   [nrd]  reference Shaders/Include/NRD.hlsli (public encodings) -- file:line cited per function
//   [com]  reference Shaders/Include/Common.hlsli (shared shader helpers) -- file:line cited per function
#pragma once

#include "hlsl.h"

namespace orc {

constexpr float NRD_FP16_MAX = 65504.0f;
constexpr float NRD_EPS = 1e-6f;
constexpr float NRD_INF = 1e6f;
// the library's G-buffer encoding: a build configuration of the whole stack (CMakeLists.txt:28-29; oracle/Makefile SUFFIX / EXTRA), default R10G10B10A2_UNORM + LINEAR
#ifndef NRD_NORMAL_ENCODING
#define NRD_NORMAL_ENCODING 2
#endif
#ifndef NRD_ROUGHNESS_ENCODING
#define NRD_ROUGHNESS_ENCODING 1
#endif
constexpr float NRD_NORMAL_ENCODING_ERROR = (NRD_NORMAL_ENCODING < 2 ? 1.5f : NRD_NORMAL_ENCODING == 2 ? 0.75f : 0.5f) / 255.0f; // [com] Common.hlsli:76-85
constexpr bool NRD_STOCHASTIC_BILINEAR = NRD_NORMAL_ENCODING == 2; // REBLUR_USE_STF == 1 && R10G10B10A2 (Common.hlsli:76-85, 359-372): otherwise gLinearClamp at the unmodified uv
constexpr float NRD_ROUGHNESS_SENSITIVITY = 0.01f;           // [com] Common.hlsli:66
constexpr float NRD_EXP_WEIGHT_DEFAULT_SCALE = 3.0f;         // [com] Common.hlsli:65
constexpr float NRD_CATROM_SHARPNESS = 0.5f;                 // [com] Common.hlsli:63
constexpr float NRD_DISOCCLUSION_THRESHOLD = 0.02f;          // [com] Common.hlsli:62
constexpr float NRD_MAX_PERCENT_OF_LOBE_VOLUME = 0.75f;      // [com] Common.hlsli:69
constexpr float NRD_CURVATURE_Z_THRESHOLD = 0.1f;            // [com] Common.hlsli:67

// ================================================================================================ [ml] Math
namespace Math {
inline float LinearStep(float a, float b, float x) { return saturate(Div(x - a, b - a)); }
inline float SmoothStep01(float x) {
    x = saturate(x);
    return x * x * (3.0f - 2.0f * x);
}
inline float SmoothStep(float a, float b, float x) { return SmoothStep01(LinearStep(a, b, x)); }
inline float4 SmoothStep(float a, float b, float4 x) { return float4(SmoothStep(a, b, x.x), SmoothStep(a, b, x.y), SmoothStep(a, b, x.z), SmoothStep(a, b, x.w)); }
inline float Sqrt01(float x) { return HwSqrt(saturate(x)); }
inline float Pow01(float x, float y) { // csrc/hip/nrdmath.h Pow01 (round 5): exponents are >= 0 at every call site
#ifdef ORC_STRICT_IEEE
    return pow(saturate(x), y);
#else
    x = saturate(x);
    return x <= 0.0f ? 0.0f : Exp2NonPos(min(y * log2(x), 0.0f));
#endif
}
inline float PositiveRcp(float x) { return Rcp(max(x, 1e-15f)); }
inline float AcosApprox(float x) { return 1.41421356f * HwSqrt(saturate(1.0f - x)); }
inline float LengthSquared(float3 v) { return dot(v, v); }
inline float LengthSquared(float2 v) { return dot(v, v); }
inline float Rsqrt(float x) { return rsqrt(x); }
} // namespace Math

// ================================================================================================ [ml] Geometry
namespace Geometry {
inline float3 RotateVector(const float4x4& M, float3 v) {
    const float* m = M.m;
    return float3(m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z, m[2] * v.x + m[6] * v.y + m[10] * v.z);
}
inline float3 RotateVectorInverse(const float4x4& M, float3 v) {
    const float* m = M.m;
    return float3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
inline float3 AffineTransform(const float4x4& M, float3 p) {
    const float* m = M.m;
    return float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13], m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
inline float4 ProjectiveTransform(const float4x4& M, float3 p) {
    const float* m = M.m;
    return float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13], m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
        m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}
inline float2 GetScreenUv(const float4x4& worldToClip, float3 X) {
    float4 clip = ProjectiveTransform(worldToClip, X);
    float2 uv = float2((Div(clip.x, clip.w)) * 0.5f + 0.5f, (Div(clip.y, clip.w)) * -0.5f + 0.5f);
    return clip.w < 0.0f ? float2(99999.0f) : uv;
}
inline float3 ReconstructViewPosition(float2 uv, float4 frustum, float viewZ = 1.0f, float orthoMode = 0.0f) {
    float s = viewZ * (1.0f - fabsf(orthoMode)) + orthoMode;
    return float3((uv.x * frustum.z + frustum.x) * s, (uv.y * frustum.w + frustum.y) * s, viewZ);
}
inline float2 RotateVector(float4 rotator, float2 v) { return float2(v.x * rotator.x + v.y * rotator.y, v.x * rotator.z + v.y * rotator.w); }
inline float4 ScaleRotator(float4 r, float2 s) { return float4(r.x * s.x, r.y * s.x, r.z * s.y, r.w * s.y); }
inline void GetBasis(float3 N, float3& T, float3& B) {
    float sz = N.z >= 0.0f ? 1.0f : -1.0f;
    float a = Rcp(sz + N.z);
    float ya = N.y * a;
    float b = N.x * ya;
    float c = N.x * sz;
    T = float3(c * N.x * a - 1.0f, sz * b, c);
    B = float3(b, N.y * ya - sz, N.y);
}
} // namespace Geometry

// ================================================================================================ [ml] Color / Sequence / Rng
namespace Color {
inline float Luminance(float3 c) { return c.x * 0.2126f + c.y * 0.7152f + c.z * 0.0722f; } // [nrd] NRD.hlsli:350-354
// BRDF::ConvertBaseColorMetalnessToAlbedoRf0 [ml, restated: dielectric Rf0 = 0.04] and BRDF::EnvironmentTerm_Rtg [nrd] NRD.hlsli:490-517 (anchor:
// "Ray Tracing Gems" ch. 32 eq. 4); matrix rows are evaluated left to right, rcp is an exact division
inline void ConvertBaseColorMetalnessToAlbedoRf0(float3 baseColor, float metalness, float3& albedo, float3& Rf0) {
    float k = fminf(fmaxf(1.0f - metalness, 0.0f), 1.0f);
    albedo = float3(baseColor.x * k, baseColor.y * k, baseColor.z * k);
    Rf0 = float3(0.04f + (baseColor.x - 0.04f) * metalness, 0.04f + (baseColor.y - 0.04f) * metalness, 0.04f + (baseColor.z - 0.04f) * metalness);
}
inline float3 EnvironmentTerm_Rtg(float3 Rf0, float NoV, float roughness) {
    float m = fminf(fmaxf(roughness * roughness, 0.0f), 1.0f);
    float x1 = NoV, x2 = NoV * NoV, x3 = NoV * x2;
    float y1 = m, y2 = m * m, y3 = m * y2;
    float biasNum = (0.99044f + -1.28514f * x1) + (1.29678f + -0.755907f * x1) * y1;
    float biasDen = (1.0f + 2.92338f * x1 + 59.4188f * x3) + (20.3225f + -27.0302f * x1 + 222.592f * x3) * y1 + (121.563f + 626.13f * x1 + 316.627f * x3) * y3;
    float scaleNum = (0.0365463f + 3.32707f * x1) + (9.0632f + -9.04756f * x1) * y1;
    float scaleDen = (1.0f + 3.59685f * x2 + -1.36772f * x3) + (9.04401f + -16.3174f * x2 + 9.22949f * x3) * y1 + (5.56589f + 19.7886f * x2 + -20.2123f * x3) * y3;
    float bias = biasNum * (Rcp(fmaxf(biasDen, 1e-6f)));
    float scale = scaleNum * (Rcp(fmaxf(scaleDen, 1e-6f)));
    (void)y2;
    auto sat = [](float v) { return fminf(fmaxf(v, 0.0f), 1.0f); };
    return float3(sat(Rf0.x * scale + bias), sat(Rf0.y * scale + bias), sat(Rf0.z * scale + bias));
}
inline float Clamp(float m1, float sigma, float x) { return clamp(x, m1 - sigma, m1 + sigma); }
} // namespace Color

namespace Sequence {
inline uint32_t CheckerBoard(uint32_t x, uint32_t y, uint32_t frameIndex) { return ((x ^ y) ^ frameIndex) & 1u; }
// [ml] Math::ReverseBits4 and Color::ColorizeZucconi (validation overlays only). The colour ramp is A. Zucconi's six-coefficient fit of the visible
// spectrum ("Improving the Rainbow", 2017) evaluated at x in [0, 1]; MathLib's exact variant is unavailable (parity unpinned, like the rest of [ml]).
inline uint32_t ReverseBits4(uint32_t x) { return ((x & 1u) << 3) | ((x & 2u) << 1) | ((x & 4u) >> 1) | ((x & 8u) >> 3); }
inline float ZucconiBump(float x, float yoffset) { return saturate((1.0f - x * x) - yoffset); }
inline float3 ColorizeZucconi(float x) {
    x = saturate(x);
    return float3(ZucconiBump(3.54585104f * (x - 0.69549072f), 0.02312639f) + ZucconiBump(3.90307140f * (x - 0.11748627f), 0.84897130f),
        ZucconiBump(2.93225262f * (x - 0.49228336f), 0.15225084f) + ZucconiBump(3.21182957f * (x - 0.86755042f), 0.88445281f),
        ZucconiBump(2.41593945f * (x - 0.27699880f), 0.52607955f) + ZucconiBump(3.96587128f * (x - 0.66077860f), 0.73949448f));
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
inline float Bayer4x4(uint32_t x, uint32_t y, uint32_t frameIndex) { return float(Bayer4x4ui(x, y, frameIndex)) * 0.0625f; } // (round 5: i / 16, csrc/hip/nrdmath.h)
} // namespace Sequence

// Our own hash RNG (MathLib's Rng::Hash is unavailable): seeded per (pixel, frame), PCG output function
struct RngHash {
    uint32_t state = 0;
    void Initialize(uint32_t x, uint32_t y, uint32_t frameIndex) {
        uint32_t s = x * 0x9E3779B1u ^ (y * 0x85EBCA77u + 0xC2B2AE3Du) ^ (frameIndex * 0x27D4EB2Fu + 0x165667B1u);
        s ^= s >> 15;
        s *= 0x2C1B3C6Du;
        s ^= s >> 12;
        s *= 0x297A2D39u;
        s ^= s >> 15;
        state = s;
    }
    uint32_t Next() {
        state = state * 747796405u + 2891336453u;
        uint32_t w = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
        return (w >> 22) ^ w;
    }
    float GetFloat() { return float(Next() >> 8) * (1.0f / 16777216.0f); }
    float2 GetFloat2() {
        float a = GetFloat();
        float b = GetFloat();
        return float2(a, b);
    }
};

// ================================================================================================ [ml] Filtering
namespace Filtering {
struct Bilinear {
    float2 origin;
    float2 weights;
};
inline Bilinear GetBilinearFilter(float2 uv, float2 texSize) {
    float2 t = uv * texSize - 0.5f;
    Bilinear r;
    r.origin = floor(t);
    r.weights = t - r.origin;
    return r;
}
inline float4 GetBilinearCustomWeights(Bilinear f, float4 customWeights) {
    float2 oneMinus = float2(1.0f - f.weights.x, 1.0f - f.weights.y);
    float4 w = customWeights;
    w.x *= oneMinus.x * oneMinus.y;
    w.y *= f.weights.x * oneMinus.y;
    w.z *= oneMinus.x * f.weights.y;
    w.w *= f.weights.x * f.weights.y;
    return w;
}
inline float ApplyBilinearFilter(float s00, float s10, float s01, float s11, Bilinear f) {
    return lerp(lerp(s00, s10, f.weights.x), lerp(s01, s11, f.weights.x), f.weights.y);
}
inline float ApplyBilinearCustomWeights(float s00, float s10, float s01, float s11, float4 w) {
    float sumw = w.x + w.y + w.z + w.w;
    float r = s00 * w.x + s10 * w.y + s01 * w.z + s11 * w.w;
    return sumw < 0.0001f ? 0.0f : Div(r, sumw);
}
// top-left texel of the 4x4 Catmull-Rom footprint; [com] REBLUR_TemporalAccumulation.hlsli:152-171 fixes the convention
inline float2 GetCatmullRomOrigin(float2 uv, float2 texSize) {
    float2 t = uv * texSize - 0.5f;
    return floor(t) - 1.0f;
}
inline float GetModifiedRoughnessFromNormalVariance(float linearRoughness, float3 nonNormalizedAverageNormal) {
    float l = length(nonNormalizedAverageNormal);
    float kappa = Div(saturate(1.0f - l * l), max(l * (3.0f - l * l), 1e-15f));
    return HwSqrt(saturate(linearRoughness * linearRoughness + kappa));
}
} // namespace Filtering

// ================================================================================================ [ml] ImportanceSampling / BRDF
namespace ImportanceSampling {
inline float GetSpecularLobeTanHalfAngle(float linearRoughness, float percentOfVolume) {
    float r = saturate(linearRoughness);
    float p = saturate(percentOfVolume);
    float m = r * r;
    return m * HwSqrt(Div(p, 1.0f - p + NRD_EPS));
}
inline float GetSpecularDominantFactor(float NoV, float linearRoughness) { // [nrd] NRD.hlsli:386-392
    float a = 0.298475f * log(39.4115f - 39.0029f * linearRoughness);
    float f = Math::Pow01(1.0f - NoV, 10.8649f) * (1.0f - a) + a;
    return saturate(f);
}
inline float4 GetSpecularDominantDirection(float3 N, float3 V, float linearRoughness) { // [nrd] NRD.hlsli:394-400
    float NoV = fabsf(dot(N, V));
    float f = GetSpecularDominantFactor(NoV, linearRoughness);
    float3 R = reflect(-V, N);
    float3 D = normalize(lerp(N, R, f));
    return float4(D, f);
}
} // namespace ImportanceSampling

namespace BRDF {
inline float Pow5(float x) { // [nrd] NRD.hlsli:408-411, as explicit products
    float t = saturate(1.0f - x);
    float t2 = t * t;
    return t2 * t2 * t;
}
} // namespace BRDF

// ================================================================================================ [nrd] NRD.hlsli
inline float3 _NRD_SafeNormalize(float3 v) { return v * rsqrt(dot(v, v) + 1e-9f); } // NRD.hlsli:316-319
inline float3 _NRD_DecodeUnitVector(float2 p) {                                        // NRD.hlsli:333-343 (unsigned, not normalised)
    p = p * 2.0f - 1.0f;
    float3 n = float3(p.x, p.y, 1.0f - fabsf(p.x) - fabsf(p.y));
    float t = saturate(-n.z);
    n.x -= t * (step(0.0f, n.x) * 2.0f - 1.0f);
    n.y -= t * (step(0.0f, n.y) * 2.0f - 1.0f);
    return n;
}
inline float2 _NRD_EncodeUnitVector(float3 v) { // NRD.hlsli:322-330 (unsigned)
    v /= fabsf(v.x) + fabsf(v.y) + fabsf(v.z); // front end (application side): IEEE division, as include/NRD.hip.h
    float2 octWrap = float2((1.0f - fabsf(v.y)) * (step(0.0f, v.x) * 2.0f - 1.0f), (1.0f - fabsf(v.x)) * (step(0.0f, v.y) * 2.0f - 1.0f));
    float2 r = v.z >= 0.0f ? float2(v.x, v.y) : octWrap;
    return r * 0.5f + 0.5f;
}
inline float3 _NRD_LinearToYCoCg(float3 c) { // NRD.hlsli:356-363
    return float3(c.x * 0.25f + c.y * 0.5f + c.z * 0.25f, c.x * 0.5f + c.y * 0.0f + c.z * -0.5f, c.x * -0.25f + c.y * 0.5f + c.z * -0.25f);
}
inline float3 _NRD_YCoCgToLinear(float3 c) { // NRD.hlsli:365-375
    float t = c.x - c.z;
    return float3(max(t + c.y, 0.0f), max(c.x + c.z, 0.0f), max(t - c.y, 0.0f));
}
inline float _REBLUR_GetHitDistanceNormalization(float viewZ, float4 hitDistParams, float roughness) { // NRD.hlsli:520-523
    return (hitDistParams.x + fabsf(viewZ) * hitDistParams.y) * lerp(1.0f, hitDistParams.z, SatExp2(hitDistParams.w * roughness * roughness));
}
// NRD.hlsli:600-637
inline float4 NRD_FrontEnd_UnpackNormalAndRoughness(float4 p, float& materialID) {
    float4 r;
#if NRD_NORMAL_ENCODING == 2
    float3 n = _NRD_DecodeUnitVector(float2(p.x, p.y));
    r.w = p.z;
    materialID = p.w * 3.0f;
#else
#if NRD_NORMAL_ENCODING == 0 || NRD_NORMAL_ENCODING == 3
    float3 n = float3(p.x * 2.0f - 1.0f, p.y * 2.0f - 1.0f, p.z * 2.0f - 1.0f);
#else
    float3 n = p.xyz();
#endif
    r.w = p.w;
    materialID = 0.0f;
#endif
    n = _NRD_SafeNormalize(n);
#if NRD_ROUGHNESS_ENCODING == 2
    r.w *= r.w;
#elif NRD_ROUGHNESS_ENCODING == 0
    r.w = HwSqrt(saturate(r.w));
#endif
    return float4(n, r.w);
}
inline float4 NRD_FrontEnd_UnpackNormalAndRoughness(float4 p) {
    float unused;
    return NRD_FrontEnd_UnpackNormalAndRoughness(p, unused);
}
inline float4 NRD_FrontEnd_PackNormalAndRoughness(float3 N, float roughness, float materialID) { // NRD.hlsli:640-667
#if NRD_ROUGHNESS_ENCODING == 2
    roughness = HwSqrt(saturate(roughness));
#elif NRD_ROUGHNESS_ENCODING == 0
    roughness *= roughness;
#endif
#if NRD_NORMAL_ENCODING == 2
    float2 e = _NRD_EncodeUnitVector(N);
    return float4(e.x, e.y, roughness, saturate(materialID / 3.0f));
#else
    N = N / max(fabsf(N.x), max(fabsf(N.y), fabsf(N.z))); // best fit (optional)
#if NRD_NORMAL_ENCODING == 0 || NRD_NORMAL_ENCODING == 3
    N = N * 0.5f + 0.5f;
#endif
    return float4(N, roughness);
#endif
}
inline float REBLUR_FrontEnd_GetNormHitDist(float hitDist, float viewZ, float4 hitDistParams, float roughness) { // NRD.hlsli:722-727
    return saturate(hitDist / _REBLUR_GetHitDistanceNormalization(viewZ, hitDistParams, roughness));
}
inline float4 REBLUR_FrontEnd_PackRadianceAndNormHitDist(float3 radiance, float normHitDist) { // NRD.hlsli:732-743 (sanitised inputs)
    return float4(_NRD_LinearToYCoCg(radiance), normHitDist);
}
inline float4 REBLUR_BackEnd_UnpackRadianceAndNormHitDist(float4 d) { return float4(_NRD_YCoCgToLinear(d.xyz()), d.w); } // NRD.hlsli:863-868

// ================================================================================================ [com] Common.hlsli
static const float3 g_Special6[6] = { // Common.hlsli:170-179 (performance mode); 0.5 * sqrt(3) and 0.15 * sqrt(3) rounded to fp32
    float3(-0.8660254f, -0.5f, 1.0f), float3(0.0f, 1.0f, 1.0f), float3(0.8660254f, -0.5f, 1.0f),
    float3(0.0f, -0.3f, 0.3f), float3(0.25980762f, 0.15f, 0.3f), float3(-0.25980762f, 0.15f, 0.3f)};
static const float3 g_Special8[8] = { // Common.hlsli:181-192
    float3(-1.0f, 0.0f, 1.0f), float3(0.0f, 1.0f, 1.0f), float3(1.0f, 0.0f, 1.0f), float3(0.0f, -1.0f, 1.0f),
    float3(-0.25f * 1.41421356f, 0.25f * 1.41421356f, 0.5f), float3(0.25f * 1.41421356f, 0.25f * 1.41421356f, 0.5f),
    float3(0.25f * 1.41421356f, -0.25f * 1.41421356f, 0.5f), float3(-0.25f * 1.41421356f, -0.25f * 1.41421356f, 0.5f)};

inline float PixelRadiusToWorld(float unproject, float orthoMode, float pixelRadius, float viewZ) { // :237-240
    return pixelRadius * unproject * lerp(viewZ, 1.0f, fabsf(orthoMode));
}
inline float GetFrustumSize(float minRectDimMulUnproject, float orthoMode, float viewZ) { return minRectDimMulUnproject * lerp(viewZ, 1.0f, fabsf(orthoMode)); } // :242-248
inline float GetHitDistFactor(float hitDist, float frustumSize) { return saturate(Div(hitDist, frustumSize)); }                                                // :250-253
inline float IsInScreenNearest(float2 uv) { return (uv.x > 0.0f && uv.y > 0.0f && uv.x < 1.0f && uv.y < 1.0f) ? 1.0f : 0.0f; }                          // :281-284
inline float4 IsInScreenBilinear(float2 footprintOrigin, float2 rectSize) {                                                                              // :288-296
    float4 p = float4(footprintOrigin.x, footprintOrigin.y, footprintOrigin.x + 1.0f, footprintOrigin.y + 1.0f);
    float4 r = float4(p.x >= 0.0f ? 1.0f : 0.0f, p.y >= 0.0f ? 1.0f : 0.0f, p.z >= 0.0f ? 1.0f : 0.0f, p.w >= 0.0f ? 1.0f : 0.0f);
    r *= float4(p.x < rectSize.x ? 1.0f : 0.0f, p.y < rectSize.y ? 1.0f : 0.0f, p.z < rectSize.x ? 1.0f : 0.0f, p.w < rectSize.y ? 1.0f : 0.0f);
    return float4(r.x * r.y, r.z * r.y, r.x * r.w, r.z * r.w); // r.xzxz * r.yyww
}
inline float GetSpecMagicCurve(float roughness, float power = 0.25f) { // :312-318
    float f = 1.0f - Exp2NonPos(-200.0f * roughness * roughness);
    f *= Math::Pow01(roughness, power);
    return f;
}
inline float ComputeParallaxInPixels(float3 X, float2 uvForZeroParallax, const float4x4& mWorldToClip, float2 rectSize) { // :320-333
    float2 uv = Geometry::GetScreenUv(mWorldToClip, X);
    float2 parallaxInUv = uv - uvForZeroParallax;
    return length(parallaxInUv * rectSize);
}
inline float ApplyThinLensEquation(float O, float curvature) { return Div(O, 2.0f * curvature * O + 1.0f); } // :404-409

// Virtual (reflected) position, NRD_USE_SPECULAR_MOTION_V2 = 1 branch; :411-461
inline float3 GetXvirtual(float hitDist, float curvature, float3 X, float3 Xprev, float3 N, float3 V, float roughness) {
    float4 D = ImportanceSampling::GetSpecularDominantDirection(N, V, roughness);
    float3 Iw = V;

    float3 reflectionRay = D.xyz() * hitDist;
    float3 T, B;
    Geometry::GetBasis(N, T, B);
    float3 O = float3(dot(T, reflectionRay), dot(B, reflectionRay), dot(N, reflectionRay)); // basis rows * ray
    O.z = -O.z;

    float mag = Rcp(2.0f * curvature * O.z - 1.0f);
    float f = length(X);
    f *= 1.0f - fabsf(dot(N, V));
    f *= max(curvature, 0.0f);
    mag *= Rcp(1.0f + f);

    float3 I = O * mag;
    Iw *= length(I);

    float closenessToSurface = saturate(Div(length(Iw), hitDist + NRD_EPS));
    float3 origin = lerp(Xprev, X, closenessToSurface * D.w);
    return origin - Iw * D.w;
}

// :465-482
inline float2 GetKernelSampleCoordinates(const float4x4& mToClip, float3 offset, float3 X, float3 T, float3 B, float4 rotator) {
    float2 o = Geometry::RotateVector(rotator, float2(offset.x, offset.y));
    float3 p = Mad(B, o.y, Mad(T, o.x, X));
    float4 clip4 = Geometry::ProjectiveTransform(mToClip, p);
    float3 clip = float3(clip4.x, clip4.y, clip4.w);
    clip.x = Div(clip.x, clip.z);
    clip.y = Div(clip.y, clip.z);
    clip.y = -clip.y;
    return float2(clip.x * 0.5f + 0.5f, clip.y * 0.5f + 0.5f);
}
// round 5 (csrc/hip/reblur_device.h KernelProjection): the same position, affine in the unrotated Poisson offset and expanded once per pixel; the strict build keeps the
// reference's expression above
struct KernelProjection {
    float3 c0, u, v;
};
inline KernelProjection MakeKernelProjection(const float4x4& M, float3 X, float3 T, float3 B, float4 r) {
    const float* m = M.m;
    const float3 Tr = Mad(B, r.z, T * r.x), Br = Mad(B, r.w, T * r.y);
    KernelProjection k;
    k.c0 = float3(m[0] * X.x + m[4] * X.y + m[8] * X.z + m[12], m[1] * X.x + m[5] * X.y + m[9] * X.z + m[13], m[3] * X.x + m[7] * X.y + m[11] * X.z + m[15]);
    k.u = float3(m[0] * Tr.x + m[4] * Tr.y + m[8] * Tr.z, m[1] * Tr.x + m[5] * Tr.y + m[9] * Tr.z, m[3] * Tr.x + m[7] * Tr.y + m[11] * Tr.z);
    k.v = float3(m[0] * Br.x + m[4] * Br.y + m[8] * Br.z, m[1] * Br.x + m[5] * Br.y + m[9] * Br.z, m[3] * Br.x + m[7] * Br.y + m[11] * Br.z);
    return k;
}
inline float2 KernelSampleUv(const KernelProjection& k, float ox, float oy) {
    float3 clip = Mad(k.v, oy, Mad(k.u, ox, k.c0));
    clip.x = Div(clip.x, clip.z);
    clip.y = Div(clip.y, clip.z);
    clip.y = -clip.y;
    return float2(clip.x * 0.5f + 0.5f, clip.y * 0.5f + 0.5f);
}
inline float GetNormalWeightParam(float nonLinearAccumSpeed, float lobeAngleFraction, float roughness = 1.0f) { // :486-499
    float percentOfVolume = NRD_MAX_PERCENT_OF_LOBE_VOLUME * lerp(lobeAngleFraction, 1.0f, nonLinearAccumSpeed);
    float tanHalfAngle = ImportanceSampling::GetSpecularLobeTanHalfAngle(roughness, percentOfVolume);
    float angle = atan(tanHalfAngle);
    angle = max(angle, NRD_NORMAL_ENCODING_ERROR);
    return Rcp(angle);
}
inline float2 GetGeometryWeightParams(float planeDistSensitivity, float frustumSize, float3 Xv, float3 Nv) { // :501-508
    float norm = planeDistSensitivity * frustumSize;
    float a = Rcp(norm);
    float b = dot(Nv, Xv) * a;
    return float2(a, -b);
}
inline float2 GetHitDistanceWeightParams(float hitDist, float nonLinearAccumSpeed, float roughness = 1.0f) { // :510-521
    float smc = GetSpecMagicCurve(roughness);
    float norm = lerp(0.0005f, 1.0f, min(nonLinearAccumSpeed, smc));
    float a = Rcp(norm);
    float b = hitDist * a;
    return float2(a, -b);
}
inline float2 GetRoughnessWeightParams(float roughness, float fraction, float sensitivity = NRD_ROUGHNESS_SENSITIVITY) { // :523-529
    float a = Rcp(lerp(sensitivity, 1.0f, saturate(roughness * fraction)));
    float b = roughness * a;
    return float2(a, -b);
}
inline float2 GetRelaxedRoughnessWeightParams(float m, float fraction = 1.0f, float sensitivity = NRD_ROUGHNESS_SENSITIVITY) { // :531-540
    float a = Rcp(lerp(sensitivity, 1.0f, lerp(m * m, m, fraction)));
    float b = m * a;
    return float2(a, -b);
}
inline float ExpApprox(float x) { return rcp(x * x - x + 1.0f); }                                                                      // :548-549
inline float ComputeExponentialWeight(float x, float px, float py) { return ExpApprox(-NRD_EXP_WEIGHT_DEFAULT_SCALE * fabsf(x * px + py)); } // :554-555
inline float ComputeNonExponentialWeight(float x, float px, float py) { return Math::SmoothStep(1.0f, 0.0f, fabsf(x * px + py)); }           // :559-560
inline float ComputeNonExponentialWeightWithSigma(float x, float px, float py, float sigma) { return Math::SmoothStep(1.0f, 0.0f, fabsf(x * px + py) - sigma * px); } // :562-563
inline float ComputeWeight(float x, float px, float py) { return ComputeNonExponentialWeight(x, px, py); }                                   // :565-569
inline float GetGaussianWeight(float r) { return exp(-0.66f * r * r); }                                                                      // :571-574
inline float GetEncodingAwareNormalWeight(float3 Ncurr, float3 Nprev, float maxAngle, float curvatureAngle, float thresholdAngle) { // :578-589 (remap = false)
    float cosa = dot(Ncurr, Nprev);
    float angle = Math::AcosApprox(cosa);
    return Math::SmoothStep01(1.0f - Div(angle - curvatureAngle - thresholdAngle, maxAngle));
}
inline float GetDisocclusionThreshold(float disocclusionThreshold, float frustumSize, float NoV) { // :593-596
    return frustumSize * saturate(Div(disocclusionThreshold, max(0.01f, NoV)));
}

} // namespace orc
