// NRD front-end / back-end functions for HIP application kernels -- the device-side counterpart of the reference's
// Shaders/Include/NRD.hlsli, which applications include in THEIR shaders to pack the denoiser inputs and to unpack / resolve its
// outputs. Same function names, argument order and meaning, so a HIP path tracer feeds and consumes the MI355X back-end without
// leaving the device:
//
//   general      NRD_FrontEnd_PackNormalAndRoughness / UnpackNormalAndRoughness   NRD.hlsli:597-667   (+ the R10G10B10A2 word codec)
//                NRD_MaterialFactors                                              NRD.hlsli:676-687
//                NRD_FrontEnd_SpecHitDistAveraging_{Begin,Add,End}, TrimHitDistance NRD.hlsli:693-716
//   REBLUR       REBLUR_FrontEnd_GetNormHitDist, PackRadianceAndNormHitDist, PackSh, PackDirectionalOcclusion   NRD.hlsli:722-790
//                REBLUR_BackEnd_UnpackRadianceAndNormHitDist, UnpackSh, UnpackDirectionalOcclusion             NRD.hlsli:863-903
//   RELAX        RELAX_FrontEnd_PackRadianceAndHitDist, PackSh; RELAX_BackEnd_UnpackRadiance, UnpackSh         NRD.hlsli:796-820, 905-924
//   SIGMA        SIGMA_FrontEnd_PackPenumbra (2 overloads), PackTranslucency; SIGMA_BackEnd_UnpackShadow        NRD.hlsli:828-855, 931
//   SG / SH      NRD_SG, NRD_SG_ExtractColor / Direction / RoughnessAA, NRD_SG_Rotate, NRD_SG_ResolveDiffuse / Specular,
//                NRD_SH_ResolveDiffuse / Specular, NRD_SG_ReJitter                                               NRD.hlsli:541-586, 937-1111
//   misc         NRD_IsValidRadiance, REBLUR_GetHitDist, NRD_GetNormalizedStrandThickness                         NRD.hlsli:1136-1162
//
// Build configuration = the library's (nrd::GetLibraryDesc().normalEncoding / roughnessEncoding): define NRD_NORMAL_ENCODING (0..4) and NRD_ROUGHNESS_ENCODING (0..2) to the
// values the linked libNRD_hip.so was built with before including this file, as the reference asks for NRDEncoding.hlsli (NRD.hlsli:290-309); undefined, they take the
// library's defaults R10G10B10A2_UNORM (oct-packed normal, 2 bits of material id) and LINEAR. All functions are __host__ __device__ (the same code serves a CPU reference of the
// application) and use only fp32 arithmetic; "sanitize" keeps the reference default (true).
// Include from a .hip / hipcc translation unit; needs nothing from libNRD_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>

#define NRD_HIP_FN __host__ __device__ inline

// normal encoding variants (match nrd::NormalEncoding) and roughness encoding variants (match nrd::RoughnessEncoding): reference NRD.hlsli:298-309
#define NRD_NORMAL_ENCODING_RGBA8_UNORM 0
#define NRD_NORMAL_ENCODING_RGBA8_SNORM 1
#define NRD_NORMAL_ENCODING_R10G10B10A2_UNORM 2 // supports material ID bits
#define NRD_NORMAL_ENCODING_RGBA16_UNORM 3
#define NRD_NORMAL_ENCODING_RGBA16_SNORM 4
#define NRD_ROUGHNESS_ENCODING_SQ_LINEAR 0   // linearRoughness * linearRoughness
#define NRD_ROUGHNESS_ENCODING_LINEAR 1      // linearRoughness
#define NRD_ROUGHNESS_ENCODING_SQRT_LINEAR 2 // sqrt( linearRoughness )
#ifndef NRD_NORMAL_ENCODING
#define NRD_NORMAL_ENCODING NRD_NORMAL_ENCODING_R10G10B10A2_UNORM
#endif
#ifndef NRD_ROUGHNESS_ENCODING
#define NRD_ROUGHNESS_ENCODING NRD_ROUGHNESS_ENCODING_LINEAR
#endif

#define NRD_FP16_MAX 65504.0f
#define NRD_PI 3.14159265358979323846f
#define NRD_EPS 1e-6f
#define NRD_INF 1e6f
#define NRD_REJITTER_VIEWZ_THRESHOLD 0.01f
#define NRD_ROUGHNESS_EPS 0.03162277660168379f // sqrt( sqrt( NRD_EPS ) )
#define NRD_MATERIAL_FACTOR_MIN_SCALE 0.02f
#define NRD_ROUGHNESS_FACTOR_MIN_SCALE 0.1f

//=================================================================================================================================
// PRIVATE HELPERS (small vector algebra on HIP's float2 / float3 / float4)
//=================================================================================================================================

namespace nrd_hip_detail {

NRD_HIP_FN float3 f3(float x, float y, float z) { return make_float3(x, y, z); }
NRD_HIP_FN float3 add(float3 a, float3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
NRD_HIP_FN float3 sub(float3 a, float3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
NRD_HIP_FN float3 mul(float3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
NRD_HIP_FN float3 mul(float3 a, float3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
NRD_HIP_FN float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
NRD_HIP_FN float length(float3 a) { return sqrtf(dot(a, a)); }
NRD_HIP_FN float3 normalize(float3 a) { return mul(a, 1.0f / sqrtf(dot(a, a))); }
NRD_HIP_FN float saturate(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
NRD_HIP_FN float3 saturate(float3 a) { return f3(saturate(a.x), saturate(a.y), saturate(a.z)); }
NRD_HIP_FN float lerp(float a, float b, float t) { return a + (b - a) * t; }
NRD_HIP_FN float3 lerp(float3 a, float3 b, float t) { return f3(lerp(a.x, b.x, t), lerp(a.y, b.y, t), lerp(a.z, b.z, t)); }
NRD_HIP_FN float3 lerp(float3 a, float3 b, float3 t) { return f3(lerp(a.x, b.x, t.x), lerp(a.y, b.y, t.y), lerp(a.z, b.z, t.z)); }
NRD_HIP_FN float3 clamp(float3 a, float lo, float hi) { return f3(fminf(fmaxf(a.x, lo), hi), fminf(fmaxf(a.y, lo), hi), fminf(fmaxf(a.z, lo), hi)); }
NRD_HIP_FN float3 reflect(float3 i, float3 n) { return sub(i, mul(n, 2.0f * dot(n, i))); }
NRD_HIP_FN float step(float edge, float x) { return x >= edge ? 1.0f : 0.0f; }
NRD_HIP_FN bool isInvalid(float x) { return isnan(x) || isinf(x); }
NRD_HIP_FN bool isInvalid(float3 a) { return isInvalid(a.x) || isInvalid(a.y) || isInvalid(a.z); }
NRD_HIP_FN uint32_t toUnorm(float x, float maxValue) { return (uint32_t)floorf(saturate(x) * maxValue + 0.5f); }

} // namespace nrd_hip_detail

NRD_HIP_FN float3 _NRD_SafeNormalize(float3 v) {
    using namespace nrd_hip_detail;
    return mul(v, 1.0f / sqrtf(dot(v, v) + 1e-9f));
}

// Oct packing
NRD_HIP_FN float2 _NRD_EncodeUnitVector(float3 v, const bool bSigned) {
    using namespace nrd_hip_detail;
    float s = fabsf(v.x) + fabsf(v.y) + fabsf(v.z);
    v = f3(v.x / s, v.y / s, v.z / s);
    float2 octWrap = make_float2((1.0f - fabsf(v.y)) * (step(0.0f, v.x) * 2.0f - 1.0f), (1.0f - fabsf(v.x)) * (step(0.0f, v.y) * 2.0f - 1.0f));
    float2 r = v.z >= 0.0f ? make_float2(v.x, v.y) : octWrap;
    return bSigned ? r : make_float2(r.x * 0.5f + 0.5f, r.y * 0.5f + 0.5f);
}

NRD_HIP_FN float3 _NRD_DecodeUnitVector(float2 p, const bool bSigned, const bool bNormalize) {
    using namespace nrd_hip_detail;
    if (!bSigned)
        p = make_float2(p.x * 2.0f - 1.0f, p.y * 2.0f - 1.0f);
    float3 n = f3(p.x, p.y, 1.0f - fabsf(p.x) - fabsf(p.y));
    float t = saturate(-n.z);
    n.x -= t * (step(0.0f, n.x) * 2.0f - 1.0f);
    n.y -= t * (step(0.0f, n.y) * 2.0f - 1.0f);
    return bNormalize ? normalize(n) : n;
}

NRD_HIP_FN float _NRD_Luminance(float3 linearColor) { return nrd_hip_detail::dot(linearColor, make_float3(0.2126f, 0.7152f, 0.0722f)); }

NRD_HIP_FN float3 _NRD_LinearToYCoCg(float3 color) {
    using namespace nrd_hip_detail;
    float Y = dot(color, f3(0.25f, 0.5f, 0.25f));
    float Co = dot(color, f3(0.5f, 0.0f, -0.5f));
    float Cg = dot(color, f3(-0.25f, 0.5f, -0.25f));
    return f3(Y, Co, Cg);
}

NRD_HIP_FN float3 _NRD_YCoCgToLinear(float3 color) {
    float t = color.x - color.z;
    float3 r;
    r.y = color.x + color.z;
    r.x = t + color.y;
    r.z = t - color.y;
    return make_float3(fmaxf(r.x, 0.0f), fmaxf(r.y, 0.0f), fmaxf(r.z, 0.0f));
}

NRD_HIP_FN float3 _NRD_YCoCgToLinear_Corrected(float Y, float Y0, float2 CoCg) {
    Y = fmaxf(Y, 0.0f);
    float k = (Y + NRD_EPS) / (Y0 + NRD_EPS);
    return _NRD_YCoCgToLinear(make_float3(Y, CoCg.x * k, CoCg.y * k));
}

NRD_HIP_FN float _NRD_GetSpecularDominantFactor(float NoV, float roughness) {
    float a = 0.298475f * logf(39.4115f - 39.0029f * roughness);
    float dominantFactor = powf(nrd_hip_detail::saturate(1.0f - NoV), 10.8649f) * (1.0f - a) + a;
    return nrd_hip_detail::saturate(dominantFactor);
}

NRD_HIP_FN float3 _NRD_GetSpecularDominantDirection(float3 N, float3 V, float dominantFactor) {
    using namespace nrd_hip_detail;
    float3 R = reflect(mul(V, -1.0f), N);
    return normalize(lerp(N, R, dominantFactor));
}

NRD_HIP_FN float _NRD_GetSpecMagicCurve(float roughness) { return 1.0f - exp2f(-30.0f * roughness * roughness); }

NRD_HIP_FN float _NRD_Pow5(float x) {
    float t = nrd_hip_detail::saturate(1.0f - x), t2 = t * t;
    return t2 * t2 * t;
}

NRD_HIP_FN float _NRD_FresnelTerm(float Rf0, float VoNH) { return Rf0 + (1.0f - Rf0) * _NRD_Pow5(VoNH); }

NRD_HIP_FN float _NRD_DistributionTerm(float roughness, float NoH) {
    float m = roughness * roughness;
    float m2 = m * m;
    float t = (NoH * m2 - NoH) * NoH + 1.0f;
    float a = m / t;
    float d = a * a;
    return d / NRD_PI;
}

NRD_HIP_FN float _NRD_GeometryTerm(float roughness, float NoL, float NoV) {
    using namespace nrd_hip_detail;
    float m = roughness * roughness;
    float m2 = m * m;
    float a = NoL + sqrtf(saturate((NoL - m2 * NoL) * NoL + m2));
    float b = NoV + sqrtf(saturate((NoV - m2 * NoV) * NoV + m2));
    return 1.0f / fmaxf(a * b, NRD_EPS);
}

NRD_HIP_FN float _NRD_DiffuseTerm(float roughness, float NoL, float NoV, float VoH) {
    float m = roughness * roughness;
    float f = 2.0f * VoH * VoH * m - 0.5f;
    float FdV = f * _NRD_Pow5(NoV) + 1.0f;
    float FdL = f * _NRD_Pow5(NoL) + 1.0f;
    float d = FdV * FdL;
    return d / NRD_PI;
}

NRD_HIP_FN float2 _NRD_ComputeBrdfs(float3 Ld, float3 Ls, float3 N, float3 V, float Rf0, float roughness) {
    using namespace nrd_hip_detail;
    float2 result;
    float NoV = fabsf(dot(N, V));
    { // Diffuse
        float3 H = normalize(add(Ld, V));
        float NoL = saturate(dot(N, Ld));
        float VoH = saturate(dot(V, H));
        float F = _NRD_FresnelTerm(Rf0, VoH);
        float Kdiff = _NRD_DiffuseTerm(roughness, NoL, NoV, VoH);
        result.x = (1.0f - F) * Kdiff * NoL;
    }
    { // Specular
        float3 H = normalize(add(Ls, V));
        H = normalize(lerp(N, H, roughness));
        float NoL = saturate(dot(N, Ls));
        float NoH = saturate(dot(N, H));
        float VoH = saturate(dot(V, H));
        float F = _NRD_FresnelTerm(Rf0, VoH);
        float D = _NRD_DistributionTerm(roughness, NoH);
        float G = _NRD_GeometryTerm(roughness, NoL, NoV);
        result.y = F * D * G * NoL;
    }
    return result;
}

NRD_HIP_FN float3 _NRD_EnvironmentTerm_Rtg(float3 Rf0, float NoV, float roughness) {
    using namespace nrd_hip_detail;
    float m = saturate(roughness * roughness);
    float X[4] = {1.0f, NoV, NoV * NoV, 0.0f};
    X[3] = NoV * X[2];
    float Y[4] = {1.0f, m, m * m, 0.0f};
    Y[3] = m * Y[2];
    // mul( M, v ) with row-major HLSL matrices
    float m1x = 0.99044f * X[0] + -1.28514f * X[1], m1y = 1.29678f * X[0] + -0.755907f * X[1];
    float m2x = 1.0f * X[0] + 2.92338f * X[1] + 59.4188f * X[3], m2y = 20.3225f * X[0] + -27.0302f * X[1] + 222.592f * X[3], m2z = 121.563f * X[0] + 626.13f * X[1] + 316.627f * X[3];
    float m3x = 0.0365463f * X[0] + 3.32707f * X[1], m3y = 9.0632f * X[0] + -9.04756f * X[1];
    float m4x = 1.0f * X[0] + 3.59685f * X[2] + -1.36772f * X[3], m4y = 9.04401f * X[0] + -16.3174f * X[2] + 9.22949f * X[3], m4z = 5.56589f * X[0] + 19.7886f * X[2] + -20.2123f * X[3];
    float bias = (m1x * Y[0] + m1y * Y[1]) * (1.0f / fmaxf(m2x * Y[0] + m2y * Y[1] + m2z * Y[3], NRD_EPS));
    float scale = (m3x * Y[0] + m3y * Y[1]) * (1.0f / fmaxf(m4x * Y[0] + m4y * Y[1] + m4z * Y[3], NRD_EPS));
    return saturate(f3(Rf0.x * scale + bias, Rf0.y * scale + bias, Rf0.z * scale + bias));
}

NRD_HIP_FN float _REBLUR_GetHitDistanceNormalization(float viewZ, float4 hitDistParams, float roughness) {
    using namespace nrd_hip_detail;
    return (hitDistParams.x + fabsf(viewZ) * hitDistParams.y) * lerp(1.0f, hitDistParams.z, saturate(exp2f(hitDistParams.w * roughness * roughness)));
}

//=================================================================================================================================
// SPHERICAL GAUSSIAN
//=================================================================================================================================

struct NRD_SG {
    float c0;
    float2 chroma;
    float normHitDist;
    float3 c1;
    float sharpness;
};

NRD_HIP_FN NRD_SG _NRD_SG_Create(float3 radiance, float3 direction, float normHitDist) {
    float3 YCoCg = _NRD_LinearToYCoCg(radiance);
    NRD_SG sg;
    sg.c0 = YCoCg.x;
    sg.chroma = make_float2(YCoCg.y, YCoCg.z);
    sg.c1 = nrd_hip_detail::mul(direction, YCoCg.x);
    sg.normHitDist = normHitDist;
    sg.sharpness = 0.0f;
    return sg;
}

NRD_HIP_FN float3 _NRD_SG_ExtractDirection(NRD_SG sg) {
    using namespace nrd_hip_detail;
    float l = fmaxf(length(sg.c1), NRD_EPS);
    return f3(sg.c1.x / l, sg.c1.y / l, sg.c1.z / l);
}

NRD_HIP_FN float _NRD_SG_IntegralApprox(NRD_SG sg) { return 2.0f * NRD_PI * (sg.c0 / sg.sharpness); }

NRD_HIP_FN float _NRD_SG_Integral(NRD_SG sg) { return _NRD_SG_IntegralApprox(sg) * (1.0f - expf(-2.0f * sg.sharpness)); }

NRD_HIP_FN float _NRD_SG_InnerProduct(NRD_SG a, NRD_SG b) {
    using namespace nrd_hip_detail;
    float d = length(add(mul(_NRD_SG_ExtractDirection(a), a.sharpness), mul(_NRD_SG_ExtractDirection(b), b.sharpness)));
    float c = expf(d - a.sharpness - b.sharpness);
    c *= 1.0f - expf(-2.0f * d);
    c /= fmaxf(d, NRD_EPS);
    return NRD_PI * saturate(2.0f * c * a.c0) * b.c0;
}

//=================================================================================================================================
// FRONT-END - GENERAL
//=================================================================================================================================

// IN_NORMAL_ROUGHNESS (the channel values of the texel as a texture unit returns them) => X                 reference NRD.hlsli:600-637
NRD_HIP_FN float4 NRD_FrontEnd_UnpackNormalAndRoughness(float4 p, float& materialID) {
    float3 n;
    float r;
#if (NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_R10G10B10A2_UNORM)
    n = _NRD_DecodeUnitVector(make_float2(p.x, p.y), false, false);
    r = p.z;
    materialID = p.w * 3.0f;
#else
#if (NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_RGBA8_UNORM || NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_RGBA16_UNORM)
    n = make_float3(p.x * 2.0f - 1.0f, p.y * 2.0f - 1.0f, p.z * 2.0f - 1.0f);
#else
    n = make_float3(p.x, p.y, p.z);
#endif
    r = p.w;
    materialID = 0.0f;
#endif
    n = _NRD_SafeNormalize(n);
#if (NRD_ROUGHNESS_ENCODING == NRD_ROUGHNESS_ENCODING_SQRT_LINEAR)
    r *= r;
#elif (NRD_ROUGHNESS_ENCODING == NRD_ROUGHNESS_ENCODING_SQ_LINEAR)
    r = sqrtf(nrd_hip_detail::saturate(r));
#endif
    return make_float4(n.x, n.y, n.z, r);
}
NRD_HIP_FN float4 NRD_FrontEnd_UnpackNormalAndRoughness(float4 p) {
    float unused;
    return NRD_FrontEnd_UnpackNormalAndRoughness(p, unused);
}

// X => IN_NORMAL_ROUGHNESS (channel values; store with NRD_StoreNormalRoughnessTexel)                         reference NRD.hlsli:640-667
NRD_HIP_FN float4 NRD_FrontEnd_PackNormalAndRoughness(float3 N, float roughness, float materialID) {
#if (NRD_ROUGHNESS_ENCODING == NRD_ROUGHNESS_ENCODING_SQRT_LINEAR)
    roughness = sqrtf(nrd_hip_detail::saturate(roughness));
#elif (NRD_ROUGHNESS_ENCODING == NRD_ROUGHNESS_ENCODING_SQ_LINEAR)
    roughness *= roughness;
#endif
#if (NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_R10G10B10A2_UNORM)
    float2 e = _NRD_EncodeUnitVector(N, false);
    return make_float4(e.x, e.y, roughness, nrd_hip_detail::saturate(materialID / 3.0f));
#else
    (void)materialID; // these encodings carry none
    const float m = fmaxf(fabsf(N.x), fmaxf(fabsf(N.y), fabsf(N.z))); // best fit (optional)
    N = make_float3(N.x / m, N.y / m, N.z / m);
#if (NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_RGBA8_UNORM || NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_RGBA16_UNORM)
    N = make_float3(N.x * 0.5f + 0.5f, N.y * 0.5f + 0.5f, N.z * 0.5f + 0.5f);
#endif
    return make_float4(N.x, N.y, N.z, roughness);
#endif
}

// the texel word of an R10_G10_B10_A2_UNORM plane (what nrdHipBindResource expects for IN_NORMAL_ROUGHNESS)
NRD_HIP_FN uint32_t NRD_StoreR10G10B10A2(float4 unorm) {
    using namespace nrd_hip_detail;
    return toUnorm(unorm.x, 1023.0f) | (toUnorm(unorm.y, 1023.0f) << 10) | (toUnorm(unorm.z, 1023.0f) << 20) | (toUnorm(unorm.w, 3.0f) << 30);
}
NRD_HIP_FN float4 NRD_LoadR10G10B10A2(uint32_t word) {
    return make_float4(float(word & 0x3FFu) / 1023.0f, float((word >> 10) & 0x3FFu) / 1023.0f, float((word >> 20) & 0x3FFu) / 1023.0f, float(word >> 30) / 3.0f);
}

// the texel of the format nrdHipBindResource expects for IN_NORMAL_ROUGHNESS under the library's encoding: RGBA8_UNORM / RGBA8_SNORM / R10_G10_B10_A2_UNORM (one 32-bit word,
// channel x in the low bits), RGBA16_UNORM / RGBA16_SNORM (64 bits, channel x in the low 16). UNORM: floor(saturate(v) * max + 0.5); SNORM: clamp, scale, round half away from 0
#if (NRD_NORMAL_ENCODING <= NRD_NORMAL_ENCODING_R10G10B10A2_UNORM)
typedef uint32_t NRD_NormalRoughnessTexel;
#else
typedef uint64_t NRD_NormalRoughnessTexel;
#endif
NRD_HIP_FN NRD_NormalRoughnessTexel NRD_StoreNormalRoughnessTexel(float4 p) {
    using namespace nrd_hip_detail;
    const float v[4] = {p.x, p.y, p.z, p.w};
    (void)v;
#if (NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_R10G10B10A2_UNORM)
    return NRD_StoreR10G10B10A2(p);
#elif (NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_RGBA8_UNORM)
    return toUnorm(p.x, 255.0f) | (toUnorm(p.y, 255.0f) << 8) | (toUnorm(p.z, 255.0f) << 16) | (toUnorm(p.w, 255.0f) << 24);
#elif (NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_RGBA16_UNORM)
    return (uint64_t)toUnorm(p.x, 65535.0f) | ((uint64_t)toUnorm(p.y, 65535.0f) << 16) | ((uint64_t)toUnorm(p.z, 65535.0f) << 32) | ((uint64_t)toUnorm(p.w, 65535.0f) << 48);
#else
    const float scale = NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_RGBA8_SNORM ? 127.0f : 32767.0f;
    const int bits = NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_RGBA8_SNORM ? 8 : 16;
    NRD_NormalRoughnessTexel t = 0;
    for (int k = 0; k < 4; k++) {
        const float c = fminf(fmaxf(v[k], -1.0f), 1.0f) * scale;
        const int32_t i = c >= 0.0f ? (int32_t)floorf(c + 0.5f) : -(int32_t)floorf(-c + 0.5f);
        t |= (NRD_NormalRoughnessTexel)((uint32_t)i & ((1u << bits) - 1u)) << (bits * k);
    }
    return t;
#endif
}
NRD_HIP_FN float4 NRD_LoadNormalRoughnessTexel(NRD_NormalRoughnessTexel t) {
#if (NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_R10G10B10A2_UNORM)
    return NRD_LoadR10G10B10A2(t);
#elif (NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_RGBA8_UNORM)
    return make_float4(float(t & 0xFFu) / 255.0f, float((t >> 8) & 0xFFu) / 255.0f, float((t >> 16) & 0xFFu) / 255.0f, float(t >> 24) / 255.0f);
#elif (NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_RGBA8_SNORM)
    return make_float4(fmaxf(float((int8_t)(t & 0xFFu)) / 127.0f, -1.0f), fmaxf(float((int8_t)((t >> 8) & 0xFFu)) / 127.0f, -1.0f), fmaxf(float((int8_t)((t >> 16) & 0xFFu)) / 127.0f, -1.0f),
        fmaxf(float((int8_t)(t >> 24)) / 127.0f, -1.0f));
#elif (NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_RGBA16_UNORM)
    return make_float4(float(t & 0xFFFFu) / 65535.0f, float((t >> 16) & 0xFFFFu) / 65535.0f, float((t >> 32) & 0xFFFFu) / 65535.0f, float(t >> 48) / 65535.0f);
#else
    return make_float4(fmaxf(float((int16_t)(t & 0xFFFFu)) / 32767.0f, -1.0f), fmaxf(float((int16_t)((t >> 16) & 0xFFFFu)) / 32767.0f, -1.0f),
        fmaxf(float((int16_t)((t >> 32) & 0xFFFFu)) / 32767.0f, -1.0f), fmaxf(float((int16_t)(t >> 48)) / 32767.0f, -1.0f));
#endif
}

// material de-modulation factors: divide irradiance by them before NRD, multiply the denoised radiance by them after
NRD_HIP_FN void NRD_MaterialFactors(float3 N, float3 V, float3 albedo, float3 Rf0, float roughness, float3& diffFactor, float3& specFactor) {
    using namespace nrd_hip_detail;
    float NoV = fabsf(dot(N, V));
    float3 Fenv = _NRD_EnvironmentTerm_Rtg(Rf0, NoV, roughness);
    const float3 lo = f3(NRD_MATERIAL_FACTOR_MIN_SCALE, NRD_MATERIAL_FACTOR_MIN_SCALE, NRD_MATERIAL_FACTOR_MIN_SCALE), one = f3(1.0f, 1.0f, 1.0f);

    diffFactor = mul(sub(one, Fenv), albedo);
    diffFactor = lerp(lo, one, diffFactor);

    specFactor = Fenv;
    specFactor = mul(specFactor, lerp(NRD_ROUGHNESS_FACTOR_MIN_SCALE, 1.0f, roughness));
    specFactor = lerp(lo, one, specFactor);
}

//=================================================================================================================================
// FRONT-END - SPECULAR HIT DISTANCE AVERAGING ( in case of rpp > 1 )
//=================================================================================================================================

NRD_HIP_FN float NRD_FrontEnd_SpecHitDistAveraging_Begin() { return NRD_INF; }
NRD_HIP_FN float NRD_FrontEnd_TrimHitDistance(float hitDist, float threshold) { return hitDist < threshold ? 0.0f : hitDist; }
NRD_HIP_FN void NRD_FrontEnd_SpecHitDistAveraging_Add(float& accumulatedSpecHitDist, float hitDist) {
    accumulatedSpecHitDist = fminf(accumulatedSpecHitDist, hitDist == 0.0f ? NRD_INF : hitDist);
}
NRD_HIP_FN void NRD_FrontEnd_SpecHitDistAveraging_End(float& accumulatedSpecHitDist) { accumulatedSpecHitDist = accumulatedSpecHitDist == NRD_INF ? 0.0f : accumulatedSpecHitDist; }

//=================================================================================================================================
// FRONT-END - REBLUR
//=================================================================================================================================

// hitDistParams = ReblurSettings::hitDistanceParameters {A, B, C, D}
NRD_HIP_FN float REBLUR_FrontEnd_GetNormHitDist(float hitDist, float viewZ, float4 hitDistParams, float roughness = 1.0f) {
    float f = _REBLUR_GetHitDistanceNormalization(viewZ, hitDistParams, roughness);
    return nrd_hip_detail::saturate(hitDist / f);
}

// X => IN_DIFF_RADIANCE_HITDIST / IN_SPEC_RADIANCE_HITDIST
NRD_HIP_FN float4 REBLUR_FrontEnd_PackRadianceAndNormHitDist(float3 radiance, float normHitDist, bool sanitize = true) {
    using namespace nrd_hip_detail;
    if (sanitize) {
        radiance = isInvalid(radiance) ? f3(0.0f, 0.0f, 0.0f) : clamp(radiance, 0.0f, NRD_FP16_MAX);
        normHitDist = isInvalid(normHitDist) ? 0.0f : saturate(normHitDist);
    }
    radiance = _NRD_LinearToYCoCg(radiance);
    return make_float4(radiance.x, radiance.y, radiance.z, normHitDist);
}

// X => IN_DIFF_SH0 / IN_SPEC_SH0 (return value) and IN_DIFF_SH1 / IN_SPEC_SH1 (out1)
NRD_HIP_FN float4 REBLUR_FrontEnd_PackSh(float3 radiance, float normHitDist, float3 direction, float4& out1, bool sanitize = true) {
    using namespace nrd_hip_detail;
    if (sanitize) {
        radiance = isInvalid(radiance) ? f3(0.0f, 0.0f, 0.0f) : clamp(radiance, 0.0f, NRD_FP16_MAX);
        normHitDist = isInvalid(normHitDist) ? 0.0f : saturate(normHitDist);
        direction = isInvalid(direction) ? f3(0.0f, 0.0f, 0.0f) : clamp(direction, -1.0f, 1.0f);
    }
    NRD_SG sg = _NRD_SG_Create(radiance, direction, normHitDist);
    out1 = make_float4(sg.c1.x, sg.c1.y, sg.c1.z, sg.sharpness);
    return make_float4(sg.c0, sg.chroma.x, sg.chroma.y, sg.normHitDist);
}

// X => IN_DIFF_DIRECTION_HITDIST
NRD_HIP_FN float4 REBLUR_FrontEnd_PackDirectionalOcclusion(float3 direction, float normHitDist, bool sanitize = true) {
    using namespace nrd_hip_detail;
    if (sanitize) {
        direction = isInvalid(direction) ? f3(0.0f, 0.0f, 0.0f) : clamp(direction, -1.0f, 1.0f);
        normHitDist = isInvalid(normHitDist) ? 0.0f : saturate(normHitDist);
    }
    NRD_SG sg = _NRD_SG_Create(f3(normHitDist, normHitDist, normHitDist), direction, normHitDist);
    return make_float4(sg.c1.x, sg.c1.y, sg.c1.z, sg.c0);
}

//=================================================================================================================================
// FRONT-END - RELAX
//=================================================================================================================================

// X => IN_DIFF_RADIANCE_HITDIST / IN_SPEC_RADIANCE_HITDIST
NRD_HIP_FN float4 RELAX_FrontEnd_PackRadianceAndHitDist(float3 radiance, float hitDist, bool sanitize = true) {
    using namespace nrd_hip_detail;
    if (sanitize) {
        radiance = isInvalid(radiance) ? f3(0.0f, 0.0f, 0.0f) : clamp(radiance, 0.0f, NRD_FP16_MAX);
        hitDist = isInvalid(hitDist) ? 0.0f : fminf(fmaxf(hitDist, 0.0f), NRD_FP16_MAX);
    }
    return make_float4(radiance.x, radiance.y, radiance.z, hitDist);
}

// X => IN_DIFF_SH0 / IN_SPEC_SH0 (return value) and IN_DIFF_SH1 / IN_SPEC_SH1 (out1)
NRD_HIP_FN float4 RELAX_FrontEnd_PackSh(float3 radiance, float hitDist, float3 direction, float4& out1, bool sanitize = true) {
    using namespace nrd_hip_detail;
    if (sanitize) {
        radiance = isInvalid(radiance) ? f3(0.0f, 0.0f, 0.0f) : clamp(radiance, 0.0f, NRD_FP16_MAX);
        hitDist = isInvalid(hitDist) ? 0.0f : fminf(fmaxf(hitDist, 0.0f), NRD_FP16_MAX);
        direction = isInvalid(direction) ? f3(0.0f, 0.0f, 0.0f) : clamp(direction, -1.0f, 1.0f);
    }
    float l = _NRD_Luminance(radiance);
    out1 = make_float4(direction.x * l, direction.y * l, direction.z * l, 0.0f);
    return make_float4(radiance.x, radiance.y, radiance.z, hitDist);
}

//=================================================================================================================================
// FRONT-END - SIGMA
//=================================================================================================================================

// directional light: X => IN_PENUMBRA
NRD_HIP_FN float SIGMA_FrontEnd_PackPenumbra(float distanceToOccluder, float tanOfLightAngularRadius) {
    float penumbraSize = distanceToOccluder * tanOfLightAngularRadius;
    float penumbraRadius = penumbraSize * 0.5f;
    return distanceToOccluder >= NRD_FP16_MAX ? NRD_FP16_MAX : fminf(penumbraRadius, 32768.0f);
}

// area / point light at a finite distance: X => IN_PENUMBRA
NRD_HIP_FN float SIGMA_FrontEnd_PackPenumbra(float distanceToOccluder, float distanceToLight, float lightSize) {
    float penumbraSize = lightSize * distanceToOccluder / fmaxf(distanceToLight - distanceToOccluder, NRD_EPS);
    float penumbraRadius = penumbraSize * 0.5f;
    return distanceToOccluder >= NRD_FP16_MAX ? NRD_FP16_MAX : fminf(penumbraRadius, 32768.0f);
}

// X => IN_TRANSLUCENCY
NRD_HIP_FN float4 SIGMA_FrontEnd_PackTranslucency(float distanceToOccluder, float3 translucency) {
    float3 t = nrd_hip_detail::saturate(translucency);
    return make_float4(distanceToOccluder >= NRD_FP16_MAX ? 1.0f : 0.0f, t.x, t.y, t.z);
}

//=================================================================================================================================
// BACK-END
//=================================================================================================================================

// OUT_DIFF_RADIANCE_HITDIST / OUT_SPEC_RADIANCE_HITDIST => X
NRD_HIP_FN float4 REBLUR_BackEnd_UnpackRadianceAndNormHitDist(float4 data) {
    float3 c = _NRD_YCoCgToLinear(make_float3(data.x, data.y, data.z));
    return make_float4(c.x, c.y, c.z, data.w);
}

// OUT_*_SH0 / OUT_*_SH1 => X
NRD_HIP_FN NRD_SG REBLUR_BackEnd_UnpackSh(float4 sh0, float4 sh1) {
    NRD_SG sg;
    sg.c0 = sh0.x;
    sg.chroma = make_float2(sh0.y, sh0.z);
    sg.normHitDist = sh0.w;
    sg.c1 = make_float3(sh1.x, sh1.y, sh1.z);
    sg.sharpness = sh1.w;
    return sg;
}

// OUT_DIFF_DIRECTION_HITDIST => X
NRD_HIP_FN NRD_SG REBLUR_BackEnd_UnpackDirectionalOcclusion(float4 data) {
    NRD_SG sg;
    sg.c0 = data.w;
    sg.chroma = make_float2(0.0f, 0.0f);
    sg.normHitDist = data.w;
    sg.c1 = make_float3(data.x, data.y, data.z);
    sg.sharpness = 0.0f;
    return sg;
}

NRD_HIP_FN float4 RELAX_BackEnd_UnpackRadiance(float4 color) { return color; }
NRD_HIP_FN NRD_SG RELAX_BackEnd_UnpackSh(float4 sh0, float4 sh1) { return REBLUR_BackEnd_UnpackSh(sh0, sh1); }

// OUT_SHADOW_TRANSLUCENCY => X ( .x = shadow, .yzw = translucent shadow for SIGMA_SHADOW_TRANSLUCENCY )
NRD_HIP_FN float SIGMA_BackEnd_UnpackShadow(float shadow) { return shadow * shadow; }
NRD_HIP_FN float4 SIGMA_BackEnd_UnpackShadow(float4 shadow) { return make_float4(shadow.x * shadow.x, shadow.y * shadow.y, shadow.z * shadow.z, shadow.w * shadow.w); }

//=================================================================================================================================
// BACK-END - HIGH QUALITY RESOLVE
//=================================================================================================================================

NRD_HIP_FN float3 NRD_SG_ExtractColor(NRD_SG sg) { return _NRD_YCoCgToLinear(make_float3(sg.c0, sg.chroma.x, sg.chroma.y)); }
NRD_HIP_FN float3 NRD_SG_ExtractDirection(NRD_SG sg) { return _NRD_SG_ExtractDirection(sg); }
NRD_HIP_FN float NRD_SG_ExtractRoughnessAA(NRD_SG sg) { return sg.sharpness; }

// rotation = 3 rows of a 3x3 matrix ( mul( rotation, v ) )
NRD_HIP_FN void NRD_SG_Rotate(NRD_SG& sg, float3 row0, float3 row1, float3 row2) {
    using namespace nrd_hip_detail;
    sg.c1 = f3(dot(row0, sg.c1), dot(row1, sg.c1), dot(row2, sg.c1));
}

NRD_HIP_FN float3 NRD_SG_ResolveDiffuse(NRD_SG sg, float3 N) {
    using namespace nrd_hip_detail;
    sg.sharpness = 4.0f;

    float c0 = 0.36f;
    float c1 = 1.0f / (4.0f * c0);

    float e = expf(-sg.sharpness);
    float e2 = e * e;
    float r = 1.0f / sg.sharpness;

    float scale = 1.0f + 2.0f * e2 - r;
    float bias = (e - e2) * r - e2;

    float NoL = dot(N, _NRD_SG_ExtractDirection(sg));
    float x = sqrtf(saturate(1.0f - scale));
    float x0 = c0 * NoL;
    float x1 = c1 * x;

    float n = x0 + x1;

    float y = saturate(NoL);
    if (fabsf(x0) <= x1)
        y = n * n / x;

    float Y = scale * y + bias;
    Y *= _NRD_SG_IntegralApprox(sg);

    return _NRD_YCoCgToLinear_Corrected(Y, sg.c0, sg.chroma);
}

NRD_HIP_FN float3 NRD_SG_ResolveSpecular(NRD_SG sg, float3 N, float3 V, float roughness) {
    using namespace nrd_hip_detail;
    roughness = fmaxf(roughness, NRD_ROUGHNESS_EPS);
    sg.sharpness = 2.0f;

    float3 H = normalize(add(_NRD_SG_ExtractDirection(sg), V));
    H = normalize(lerp(N, H, roughness));

    float m = roughness * roughness;
    float m2 = m * m;

    NRD_SG ndf;
    ndf.c0 = 1.0f / (NRD_PI * m2);
    ndf.c1 = H;
    ndf.sharpness = 2.0f / fmaxf(m2, NRD_EPS);
    ndf.chroma = make_float2(0.0f, 0.0f);
    ndf.normHitDist = 0.0f;

    ndf.c0 *= lerp(1.0f, 0.75f * 2.0f * NRD_PI, m2);

    NRD_SG ndfWarped;
    ndfWarped.c0 = ndf.c0;
    ndfWarped.c1 = reflect(mul(V, -1.0f), ndf.c1);
    ndfWarped.sharpness = ndf.sharpness / fmaxf(4.0f * fabsf(dot(ndf.c1, V)), NRD_EPS);
    ndfWarped.chroma = make_float2(0.0f, 0.0f);
    ndfWarped.normHitDist = 0.0f;

    float NoV = fabsf(dot(N, V));
    float NoL = saturate(dot(N, ndfWarped.c1));

    ndfWarped.c0 *= NoL;
    ndfWarped.c0 *= _NRD_GeometryTerm(roughness, NoL, NoV);

    float Y = _NRD_SG_InnerProduct(ndfWarped, sg);

    return _NRD_YCoCgToLinear_Corrected(Y, sg.c0, sg.chroma);
}

// Re-jittering ( SH / SG variants only ): returns the {diffuse, specular} scale that brings back the high-frequency normal details
// lost by resolving at the denoised (low-frequency) normal. Neighbours: e = (+1, 0), w = (-1, 0), n = (0, +1), s = (0, -1).
NRD_HIP_FN float2 NRD_SG_ReJitter(NRD_SG diffSg, NRD_SG specSg, float3 Rf0, float3 V, float roughness, float Z, float Ze, float Zw, float Zn, float Zs, float3 N, float3 Ne, float3 Nw,
    float3 Nn, float3 Ns) {
    using namespace nrd_hip_detail;
    roughness = fmaxf(roughness, NRD_ROUGHNESS_EPS);
    float rf0 = _NRD_Luminance(Rf0);

    float3 Ld = _NRD_SG_ExtractDirection(diffSg);
    float3 Ls = _NRD_SG_ExtractDirection(specSg);

    float smc = _NRD_GetSpecMagicCurve(roughness);
    Ls = normalize(lerp(V, Ls, smc));

    float2 brdfCenter = _NRD_ComputeBrdfs(Ld, Ls, N, V, rf0, roughness);

    float2 brdfAverage = _NRD_ComputeBrdfs(Ld, Ls, Ne, V, rf0, roughness);
    float2 t = _NRD_ComputeBrdfs(Ld, Ls, Nn, V, rf0, roughness);
    brdfAverage = make_float2(brdfAverage.x + t.x, brdfAverage.y + t.y);
    t = _NRD_ComputeBrdfs(Ld, Ls, Nw, V, rf0, roughness);
    brdfAverage = make_float2(brdfAverage.x + t.x, brdfAverage.y + t.y);
    t = _NRD_ComputeBrdfs(Ld, Ls, Ns, V, rf0, roughness);
    brdfAverage = make_float2(brdfAverage.x + t.x, brdfAverage.y + t.y);

    float NoV = fabsf(dot(N, V));
    float zThreshold = NRD_REJITTER_VIEWZ_THRESHOLD * fabsf(Z) / (NoV * 0.95f + 0.05f);

    uint32_t sum = fabsf(Ze - Z) < zThreshold && dot(Ne, N) > 0.0f ? 1u : 0u;
    sum += fabsf(Zn - Z) < zThreshold && dot(Nn, N) > 0.0f ? 1u : 0u;
    sum += fabsf(Zw - Z) < zThreshold && dot(Nw, N) > 0.0f ? 1u : 0u;
    sum += fabsf(Zs - Z) < zThreshold && dot(Ns, N) > 0.0f ? 1u : 0u;

    float2 f = make_float2((brdfCenter.x * 4.0f + NRD_EPS) / (brdfAverage.x + NRD_EPS), (brdfCenter.y * 4.0f + NRD_EPS) / (brdfAverage.y + NRD_EPS));
    if (sum != 4u)
        return make_float2(1.0f, 1.0f);
    return make_float2(fminf(fmaxf(f.x, 1.0f / NRD_PI), NRD_PI), fminf(fmaxf(f.y, 1.0f / NRD_PI), NRD_PI));
}

//=================================================================================================================================
// BACK-END - SPHERICAL HARMONICS RESOLVE ( cheaper, lower quality )
//=================================================================================================================================

NRD_HIP_FN float3 NRD_SH_ResolveDiffuse(NRD_SG sh, float3 N) {
    float Y = nrd_hip_detail::dot(N, sh.c1) + 0.5f * sh.c0;
    return _NRD_YCoCgToLinear_Corrected(Y, sh.c0, sh.chroma);
}

NRD_HIP_FN float3 NRD_SH_ResolveSpecular(NRD_SG sh, float3 N, float3 V, float roughness) {
    using namespace nrd_hip_detail;
    float NoV = fabsf(dot(N, V));
    float f = _NRD_GetSpecularDominantFactor(NoV, roughness);
    float3 D = _NRD_GetSpecularDominantDirection(N, V, f);
    float Y = dot(D, sh.c1) + 0.5f * sh.c0;
    return _NRD_YCoCgToLinear_Corrected(Y, sh.c0, sh.chroma);
}

//=================================================================================================================================
// MISC ( NRD.hlsli:1136-1162 )
//=================================================================================================================================

NRD_HIP_FN bool _NRD_IsInvalid(float x) { return nrd_hip_detail::isInvalid(x); }
NRD_HIP_FN bool _NRD_IsInvalid(float3 x) { return nrd_hip_detail::isInvalid(x); }

// Needs to be used to avoid summing up NAN/INF values in many rays per pixel scenarios
NRD_HIP_FN bool NRD_IsValidRadiance(float3 radiance) { return !_NRD_IsInvalid(radiance); }

// Scales normalized hit distance back to real length
NRD_HIP_FN float REBLUR_GetHitDist(float normHitDist, float viewZ, float4 hitDistParams, float roughness) {
    return normHitDist * _REBLUR_GetHitDistanceNormalization(viewZ, hitDistParams, roughness);
}

// Normalized strand thickness factor in range [0; 1]: 0 - thick enough (in pixels) for a stable multi-pixel projection, 1 - "pixel soup".
// pixelSize = size of a pixel in world units at "viewZ" = gUnproject * ( isOrtho ? 1.0 : abs( viewZ ) )
NRD_HIP_FN float NRD_GetNormalizedStrandThickness(float strandThickness, float pixelSize) { return pixelSize / (pixelSize + strandThickness); }
