// REBLUR device-side shared pieces: compile-time settings, storage packing, per-signal helpers, history fetch.
//   settings  : reference Shaders/Include/REBLUR_Config.hlsli:13-98
//   packing   : reference Shaders/Include/REBLUR_Common.hlsli:13-80
//   helpers   : reference Shaders/Include/REBLUR_Common.hlsli:84-361, Common.hlsli:226-656
// Every formula keeps the operation order pinned in DESIGN.md "Numerics" so results are reproducible bit-for-bit.
#pragma once

#include "../common/pass_constants.h"
#include "nrdmath.h"
#include "planes.h"

// The REBLUR launchers reject orthographic projections (CheckSupported), so gOrthoMode is 0 in every kernel that runs: the arithmetic keeps the reference's expressions
// with the COMPILE-TIME constant (round 5: read from the constants at run time until then, which kept both arms of every `ortho ? a : b` alive -- ~3 % of
// TemporalAccumulation's instructions; the compiler only folds what is value-identical, x * 1 and the selects, never x + 0)
#ifndef NRD_ORTHO_MODE
#define NRD_ORTHO_MODE(c) 0.0f
#endif

namespace nrdhip {

typedef nrdc::ReblurConstants ReblurCB;

#define REBLUR_MAX_ACCUM_FRAME_NUM 63.0f
#define REBLUR_MAX_MATERIALID_NUM 15.0f
#define REBLUR_PRE_BLUR_FRACTION_SCALE 2.0f
#define REBLUR_PRE_BLUR_NON_LINEAR_ACCUM_SPEED (1.0f / (1.0f + 10.0f))
#define REBLUR_BLUR_FRACTION_SCALE 1.0f
#define REBLUR_POST_BLUR_FRACTION_SCALE 0.5f
#define REBLUR_POST_BLUR_RADIUS_SCALE 2.0f
#define REBLUR_NORMAL_ULP NRD_NORMAL_ENCODING_ERROR
#define REBLUR_ALMOST_ZERO_ANGLE 0.01745240643728351f // cos( 89 deg )
#define REBLUR_FIREFLY_SUPPRESSOR_MAX_RELATIVE_INTENSITY 38.0f
#define REBLUR_FIREFLY_SUPPRESSOR_RADIUS_SCALE 0.1f
#define REBLUR_FIREFLY_SUPPRESSOR_FAST_RELATIVE_INTENSITY 4.0f
#define REBLUR_ANTI_FIREFLY_FILTER_RADIUS 4
#define REBLUR_ANTI_FIREFLY_SIGMA_SCALE 2.0f
#define REBLUR_ROUGHNESS_SENSITIVITY_IN_TA (NRD_ROUGHNESS_SENSITIVITY * 0.3f)
#define REBLUR_SAMPLES_PER_FRAME 1.0f
#define REBLUR_MAX_PERCENT_OF_LOBE_VOLUME_FOR_PRE_PASS 0.3f
#define REBLUR_COLOR_CLAMPING_SIGMA_SCALE 2.0f           // radiance signals
#define REBLUR_COLOR_CLAMPING_SIGMA_SCALE_OCCLUSION 1.0f // REBLUR_OCCLUSION (reference REBLUR_Config.hlsli:94-98)

enum SpatialMode { PRE_BLUR = 0, BLUR = 1, POST_BLUR = 2 };

NRD_D float4 ToF4(nrdc::F4 v) { return F4(v.x, v.y, v.z, v.w); }
NRD_D float3 ToF3(nrdc::F4 v) { return F3(v.x, v.y, v.z); }
NRD_D float2 ToF2(nrdc::F2 v) { return F2(v.x, v.y); }

// ---- Poisson-like 8-tap kernel (reference Common.hlsli:181-192) ------------------------------------------------------
// GetGaussianWeight( offset.z ) = Exp( -0.66 z^2 ) only ever sees z = 1 and z = 0.5: the two results of OUR Exp() are
// baked in as bit patterns (0x3f04505f, 0x3f590f8f; tests/test_numerics.py re-derives them on the GPU).
#define REBLUR_GAUSSIAN_WEIGHT_Z1 0.5168513655662537f
#define REBLUR_GAUSSIAN_WEIGHT_Z05 0.8478936553001404f
#define REBLUR_GAUSSIAN_WEIGHT_Z03 0.9423297643661499f // 0x3f713c86: the inner ring of g_Special6 (performance mode)
// g_Special6 (reference Common.hlsli:170-179): 0.5 * sqrt(3) and 0.15 * sqrt(3) rounded to fp32
__device__ __constant__ const float g_Special6[6][3] = {{-0.8660254f, -0.5f, 1.0f}, {0.0f, 1.0f, 1.0f}, {0.8660254f, -0.5f, 1.0f},
    {0.0f, -0.3f, 0.3f}, {0.25980762f, 0.15f, 0.3f}, {-0.25980762f, 0.15f, 0.3f}};
__device__ __constant__ const float g_Special8[8][3] = {{-1.0f, 0.0f, 1.0f}, {0.0f, 1.0f, 1.0f}, {1.0f, 0.0f, 1.0f}, {0.0f, -1.0f, 1.0f},
    {-0.25f * 1.41421356f, 0.25f * 1.41421356f, 0.5f}, {0.25f * 1.41421356f, 0.25f * 1.41421356f, 0.5f}, {0.25f * 1.41421356f, -0.25f * 1.41421356f, 0.5f},
    {-0.25f * 1.41421356f, -0.25f * 1.41421356f, 0.5f}};

// ---- storage packing ----------------------------------------------------------------------------------------------
NRD_D uint32_t PackInternalData(float diffAccumSpeed, float specAccumSpeed, float materialID) {
    float tx = Div(diffAccumSpeed, REBLUR_MAX_ACCUM_FRAME_NUM), ty = Div(specAccumSpeed, REBLUR_MAX_ACCUM_FRAME_NUM), tz = Div(materialID, REBLUR_MAX_MATERIALID_NUM);
    uint32_t p = (uint32_t)floorf(Sat(tx) * 63.0f + 0.5f);
    p |= (uint32_t)floorf(Sat(ty) * 63.0f + 0.5f) << 6;
    p |= (uint32_t)floorf(Sat(tz) * 15.0f + 0.5f) << 12;
    return p;
}
NRD_D float3 UnpackInternalData(uint32_t p) {
    float3 t = F3(NRD_DIV_63(float(p & 63u)), NRD_DIV_63(float((p >> 6) & 63u)), NRD_DIV_15(float((p >> 12) & 15u)));
    t.x *= REBLUR_MAX_ACCUM_FRAME_NUM;
    t.y *= REBLUR_MAX_ACCUM_FRAME_NUM;
    t.z *= REBLUR_MAX_MATERIALID_NUM;
    return t;
}
// DATA1: RG8_UNORM for diffuse+specular, R8_UNORM for a single signal (both channels alias)
template <bool DIFF, bool SPEC>
NRD_D void StoreData1(const Plane& p, int x, int y, float diffAccumSpeed, float specAccumSpeed) {
    float rx = Sat(Div(diffAccumSpeed, REBLUR_MAX_ACCUM_FRAME_NUM)), ry = Sat(Div(specAccumSpeed, REBLUR_MAX_ACCUM_FRAME_NUM));
    if (DIFF && SPEC)
        StoreRG8Unorm(p, x, y, F2(rx, ry));
    else
        StoreR8Unorm(p, x, y, DIFF ? rx : ry);
}
template <bool DIFF, bool SPEC>
NRD_D float2 LoadData1(const Plane& p, int x, int y) {
    float2 v;
    if (DIFF && SPEC)
        v = LoadRG8Unorm(p, x, y);
    else {
        float s = LoadR8Unorm(p, x, y);
        v = DIFF ? F2(s, 0.0f) : F2(s, s); // single-signal: .y aliases .x for specular; diffuse never reads .y
    }
    return v * REBLUR_MAX_ACCUM_FRAME_NUM;
}
NRD_D uint32_t PackData2(float fbits, float curvature, float virtualHistoryAmount) {
    uint32_t p = (uint32_t)(fbits + 0.5f);
    p |= (uint32_t)(Sat(virtualHistoryAmount) * 255.0f + 0.5f) << 8;
    p |= (uint32_t)FloatToHalfBits(curvature) << 16;
    return p;
}
NRD_D float2 UnpackData2(uint32_t p, uint32_t& bits) {
    bits = p & 0xFFu;
    return F2(NRD_DIV_255(float((p >> 8) & 0xFFu)), HalfBitsToFloat((uint16_t)(p >> 16)));
}

// ---- helpers ------------------------------------------------------------------------------------------------------
NRD_D float UnpackViewZ(const ReblurCB& c, float z) { return Abs(z * c.gViewZScale); }
NRD_D float3 GetViewVector(const ReblurCB& c, float3 X, bool isViewSpace = false) {
    return NRD_ORTHO_MODE(c) == 0.0f ? Normalize(-X) : (isViewSpace ? F3(0.0f, 0.0f, -1.0f) : ToF3(c.gViewVectorWorld));
}
NRD_D float3 GetViewVectorPrev(const ReblurCB& c, float3 Xprev, float3 cameraDelta) {
    return NRD_ORTHO_MODE(c) == 0.0f ? Normalize(cameraDelta - Xprev) : ToF3(c.gViewVectorWorldPrev);
}
NRD_D float PixelRadiusToWorld(float unproject, float orthoMode, float pixelRadius, float viewZ) { return pixelRadius * unproject * Lerp(viewZ, 1.0f, Abs(orthoMode)); }
NRD_D float GetFrustumSize(float minRectDimMulUnproject, float orthoMode, float viewZ) { return minRectDimMulUnproject * Lerp(viewZ, 1.0f, Abs(orthoMode)); }
NRD_D float GetHitDistFactor(float hitDist, float frustumSize) { return Sat(Div(hitDist, frustumSize)); }
NRD_D float IsInScreenNearest(float2 uv) { return (uv.x > 0.0f && uv.y > 0.0f && uv.x < 1.0f && uv.y < 1.0f) ? 1.0f : 0.0f; }
NRD_D float4 IsInScreenBilinear(float2 footprintOrigin, float2 rectSize) {
    float4 p = F4(footprintOrigin.x, footprintOrigin.y, footprintOrigin.x + 1.0f, footprintOrigin.y + 1.0f);
    float4 r = F4(p.x >= 0.0f ? 1.0f : 0.0f, p.y >= 0.0f ? 1.0f : 0.0f, p.z >= 0.0f ? 1.0f : 0.0f, p.w >= 0.0f ? 1.0f : 0.0f);
    r = r * F4(p.x < rectSize.x ? 1.0f : 0.0f, p.y < rectSize.y ? 1.0f : 0.0f, p.z < rectSize.x ? 1.0f : 0.0f, p.w < rectSize.y ? 1.0f : 0.0f);
    return F4(r.x * r.y, r.z * r.y, r.x * r.w, r.z * r.w);
}
NRD_D float GetSpecMagicCurve(float roughness, float power = 0.25f) {
    float f = 1.0f - Exp2NonPos(-200.0f * roughness * roughness);
    f *= Pow01(roughness, power);
    return f;
}
NRD_D float ComputeParallaxInPixels(float3 X, float2 uvForZeroParallax, const float* mWorldToClip, float2 rectSize) {
    float2 uv = GetScreenUv(mWorldToClip, X);
    float2 parallaxInUv = uv - uvForZeroParallax;
    return Length(parallaxInUv * rectSize);
}
NRD_D float3 GetXvirtual(float hitDist, float curvature, float3 X, float3 Xprev, float3 N, float3 V, float roughness) {
    float4 D = GetSpecularDominantDirection(N, V, roughness);
    float3 Iw = V;

    float3 reflectionRay = Xyz(D) * hitDist;
    float3 T, B;
    GetBasis(N, T, B);
    float3 O = F3(Dot(T, reflectionRay), Dot(B, reflectionRay), Dot(N, reflectionRay));
    O.z = -O.z;

    float mag = Rcp(2.0f * curvature * O.z - 1.0f);
    float f = Length(X);
    f *= 1.0f - Abs(Dot(N, V));
    f *= Max(curvature, 0.0f);
    mag *= Rcp(1.0f + f);

    float3 I = O * mag;
    Iw = Iw * Length(I);

    float closenessToSurface = Sat(Div(Length(Iw), hitDist + NRD_EPS));
    float3 origin = Lerp(Xprev, X, closenessToSurface * D.w);
    return origin - Iw * D.w;
}
NRD_D float2 GetKernelSampleCoordinates(const float* mToClip, float3 offset, float3 X, float3 T, float3 B, float4 rotator) {
    float2 o = RotateVector(rotator, F2(offset.x, offset.y));
    float3 p = Mad(B, o.y, Mad(T, o.x, X));
    float4 clip4 = ProjectiveTransform(mToClip, p);
    float3 clip = F3(clip4.x, clip4.y, clip4.w);
    clip.x = Div(clip.x, clip.z);
    clip.y = Div(clip.y, clip.z);
    clip.y = -clip.y;
    return F2(clip.x * 0.5f + 0.5f, clip.y * 0.5f + 0.5f);
}
// The same, expanded once per pixel (round 5): the clip-space position of a world-space kernel tap is affine in the UNROTATED Poisson offset (ox, oy) --
//   p = X + T * (ox r.x + oy r.y) + B * (ox r.z + oy r.w) = X + ox * (T r.x + B r.z) + oy * (T r.y + B r.w),   clip = M p = c0 + ox * u + oy * v
// -- so the rotation, the basis and the projection cost 42 operations per PIXEL and a tap is 6 fused multiply-adds instead of 22 operations (8 taps: 90 instead of 176).
// A re-association of the reference's expression (Common.hlsli:465-482): the oracle states the same form, the strict build keeps the reference's.
struct KernelProjection {
    float3 c0, u, v; // (clip.x, clip.y, clip.w) of the centre and per unit of ox / oy
};
NRD_D KernelProjection MakeKernelProjection(const float* m, float3 X, float3 T, float3 B, float4 r) {
    const float3 Tr = Mad(B, r.z, T * r.x), Br = Mad(B, r.w, T * r.y);
    KernelProjection k;
    k.c0 = F3(m[0] * X.x + m[4] * X.y + m[8] * X.z + m[12], m[1] * X.x + m[5] * X.y + m[9] * X.z + m[13], m[3] * X.x + m[7] * X.y + m[11] * X.z + m[15]);
    k.u = F3(m[0] * Tr.x + m[4] * Tr.y + m[8] * Tr.z, m[1] * Tr.x + m[5] * Tr.y + m[9] * Tr.z, m[3] * Tr.x + m[7] * Tr.y + m[11] * Tr.z);
    k.v = F3(m[0] * Br.x + m[4] * Br.y + m[8] * Br.z, m[1] * Br.x + m[5] * Br.y + m[9] * Br.z, m[3] * Br.x + m[7] * Br.y + m[11] * Br.z);
    return k;
}
NRD_D float2 KernelSampleUv(const KernelProjection& k, float ox, float oy) {
    float3 clip = Mad(k.v, oy, Mad(k.u, ox, k.c0));
    clip.x = Div(clip.x, clip.z);
    clip.y = Div(clip.y, clip.z);
    clip.y = -clip.y;
    return F2(clip.x * 0.5f + 0.5f, clip.y * 0.5f + 0.5f);
}
NRD_D float GetNormalWeightParam(float nonLinearAccumSpeed, float lobeAngleFraction, float roughness = 1.0f) {
    float percentOfVolume = NRD_MAX_PERCENT_OF_LOBE_VOLUME * Lerp(lobeAngleFraction, 1.0f, nonLinearAccumSpeed);
    float tanHalfAngle = GetSpecularLobeTanHalfAngle(roughness, percentOfVolume);
    float angle = Atan(tanHalfAngle);
    angle = Max(angle, NRD_NORMAL_ENCODING_ERROR);
    return Rcp(angle);
}
NRD_D float2 GetGeometryWeightParams(float planeDistSensitivity, float frustumSize, float3 Xv, float3 Nv) {
    float norm = planeDistSensitivity * frustumSize;
    float a = Rcp(norm);
    float b = Dot(Nv, Xv) * a;
    return F2(a, -b);
}
NRD_D float2 GetHitDistanceWeightParams(float hitDist, float nonLinearAccumSpeed, float roughness = 1.0f) {
    float smc = GetSpecMagicCurve(roughness);
    float norm = Lerp(0.0005f, 1.0f, Min(nonLinearAccumSpeed, smc));
    float a = Rcp(norm);
    float b = hitDist * a;
    return F2(a, -b);
}
NRD_D float2 GetRoughnessWeightParams(float roughness, float fraction, float sensitivity = NRD_ROUGHNESS_SENSITIVITY) {
    float a = Rcp(Lerp(sensitivity, 1.0f, Sat(roughness * fraction)));
    float b = roughness * a;
    return F2(a, -b);
}
NRD_D float2 GetRelaxedRoughnessWeightParams(float m, float fraction = 1.0f, float sensitivity = NRD_ROUGHNESS_SENSITIVITY) {
    float a = Rcp(Lerp(sensitivity, 1.0f, Lerp(m * m, m, fraction)));
    float b = m * a;
    return F2(a, -b);
}
NRD_D float ExpApprox(float x) { return Rcp(x * x - x + 1.0f); }
NRD_D float ComputeExponentialWeight(float x, float px, float py) { return ExpApprox(-NRD_EXP_WEIGHT_DEFAULT_SCALE * Abs(x * px + py)); }
// SmoothStep(1, 0, t) = SmoothStep01(LinearStep(1, 0, t)) and LinearStep(1, 0, t) = saturate(Div(t - 1, -1)) = saturate(1 - t) EXACTLY: v_rcp_f32(-1) is -1
// and negation commutes with rounding. Written as 1 - t it is one instruction (a subtraction with the clamp modifier) instead of three, same bits.
NRD_D float SmoothStepOneToZero(float t) { return SmoothStep01(1.0f - t); }
NRD_D float ComputeNonExponentialWeight(float x, float px, float py) { return SmoothStepOneToZero(Abs(x * px + py)); }
NRD_D float ComputeNonExponentialWeightWithSigma(float x, float px, float py, float sigma) { return SmoothStepOneToZero(Abs(x * px + py) - sigma * px); }
NRD_D float ComputeWeight(float x, float px, float py) { return ComputeNonExponentialWeight(x, px, py); }
NRD_D float GetGaussianWeight(float r) { return Exp(-0.66f * r * r); }
NRD_D float GetEncodingAwareNormalWeight(float3 Ncurr, float3 Nprev, float maxAngle, float curvatureAngle, float thresholdAngle) {
    float cosa = Dot(Ncurr, Nprev);
    float angle = AcosApprox(cosa);
    return SmoothStep01(1.0f - Div(angle - curvatureAngle - thresholdAngle, maxAngle));
}
NRD_D float GetDisocclusionThreshold(float disocclusionThreshold, float frustumSize, float NoV) { return frustumSize * Sat(Div(disocclusionThreshold, Max(0.01f, NoV))); }
#if NRD_EXPERIMENT_NO_MATERIALS // A/B builds only: what a compile-time "no material test in this frame" flag would buy (tools/build_variant.py)
NRD_D bool CompareMaterials(float, float, float) { return true; }
#else
NRD_D bool CompareMaterials(float m0, float m, float minm) { return Max(m0, minm) == Max(m, minm); }
#endif

NRD_D float GetMinAllowedLimitForHitDistNonLinearAccumSpeed(const ReblurCB& c, float roughness) {
    float frameNum = 0.5f * GetSpecMagicCurve(roughness) * c.gMaxAccumulatedFrameNum;
    return Rcp(1.0f + frameNum);
}
NRD_D float GetFadeBasedOnAccumulatedFrames(const ReblurCB& c, float accumSpeed) {
    float a = c.gHistoryFixFrameNum * 2.0f * (1.0f / 3.0f) + 1e-6f;
    float b = c.gHistoryFixFrameNum * 4.0f * (1.0f / 3.0f) + 2e-6f;
    return LinearStep(a, b, accumSpeed);
}
template <typename CB> // reference REBLUR_Common.hlsli:111-124; hasData = false for the empty pixels of a checkerboarded input
NRD_D float GetNonLinearAccumSpeed(const CB& c, float accumSpeed, float maxAccumSpeed, float confidence, bool hasData) {
    float nonLinearAccumSpeed = Max(1.0f - confidence, Rcp(1.0f + Min(accumSpeed, maxAccumSpeed)));
    if (!hasData)
        nonLinearAccumSpeed *= Lerp(1.0f - c.gCheckerboardResolveAccumSpeed, 1.0f, nonLinearAccumSpeed);
    return nonLinearAccumSpeed;
}
NRD_D float RemapRoughnessToResponsiveFactor(const ReblurCB& c, float roughness) {
    float amount = Div(roughness + NRD_EPS, c.gResponsiveAccumulationRoughnessThreshold + NRD_EPS);
    return SmoothStep01(amount);
}
NRD_D float GetLumaScale(float currLuma, float newLuma) { return Div(newLuma + NRD_EPS, currLuma + NRD_EPS); }
NRD_D float4 MixHistoryAndCurrent(const ReblurCB& c, float4 history, float4 current, float f, float roughness = 1.0f) {
    float4 r;
    r.x = Lerp(history.x, current.x, f);
    r.y = Lerp(history.y, current.y, f);
    r.z = Lerp(history.z, current.z, f);
    r.w = Lerp(history.w, current.w, Max(f, GetMinAllowedLimitForHitDistNonLinearAccumSpeed(c, roughness)));
    return r;
}
NRD_D float GetLuma(float4 v) { return v.x; }
NRD_D float4 ChangeLuma(float4 v, float newLuma) {
    float s = GetLumaScale(GetLuma(v), newLuma);
    return F4(v.x * s, v.y * s, v.z * s, v.w);
}
NRD_D float4 ClampNegativeToZero(float4 v) {
    float3 rgb = LinearToYCoCg(YCoCgToLinear(Xyz(v)));
    return F4(rgb, Sat(v.w));
}
// REBLUR_TYPE kinds (reference REBLUR_Common.hlsli:148-215): 0 radiance (float4, RGBA16F), 1 occlusion (float = the normalised hit distance
// alone, R16_UNORM), 2 directional occlusion (float4 whose "luma" is .w, RGBA16_SNORM; otherwise the radiance arithmetic)
enum { SIGNAL_RADIANCE = 0, SIGNAL_OCCLUSION = 1, SIGNAL_DIRECTIONAL_OCCLUSION = 2 };
struct DirOcc { // distinct type so that GetLuma / ChangeLuma / ClampNegativeToZero overload
    float4 v;
};
NRD_D DirOcc MakeDirOcc(float4 v) {
    DirOcc r;
    r.v = v;
    return r;
}
NRD_D DirOcc operator+(DirOcc a, DirOcc b) { return MakeDirOcc(a.v + b.v); }
NRD_D DirOcc operator*(DirOcc a, float b) { return MakeDirOcc(a.v * b); }
NRD_D DirOcc Mad(DirOcc a, float s, DirOcc c) { return MakeDirOcc(Mad(a.v, s, c.v)); }
NRD_D DirOcc Lerp(DirOcc a, DirOcc b, float t) { return MakeDirOcc(Lerp(a.v, b.v, t)); }
NRD_D DirOcc Select(bool c, DirOcc a, DirOcc b) { return MakeDirOcc(Select(c, a.v, b.v)); }
NRD_D float Select(bool c, float a, float b) { return c ? a : b; }
NRD_D float ExtractHitDist(float4 v) { return v.w; }
NRD_D float ExtractHitDist(float v) { return v; }
NRD_D float ExtractHitDist(DirOcc s) { return s.v.w; }
NRD_D float GetLuma(float v) { return v; }
NRD_D float GetLuma(DirOcc s) { return s.v.w; }
NRD_D float ChangeLuma(float, float newLuma) { return newLuma; }
NRD_D DirOcc ChangeLuma(DirOcc s, float newLuma) {
    float k = GetLumaScale(s.v.w, newLuma);
    return MakeDirOcc(F4(s.v.x * k, s.v.y * k, s.v.z * k, newLuma));
}
NRD_D float ClampNegativeToZero(float v) { return Sat(v); }
NRD_D DirOcc ClampNegativeToZero(DirOcc s) { return ChangeLuma(s, Sat(s.v.w)); }
NRD_D float MixHistoryAndCurrent(const ReblurCB& c, float history, float current, float f, float roughness = 1.0f) {
    return Lerp(history, current, Max(f, GetMinAllowedLimitForHitDistNonLinearAccumSpeed(c, roughness)));
}
NRD_D DirOcc MixHistoryAndCurrent(const ReblurCB& c, DirOcc history, DirOcc current, float f, float roughness = 1.0f) {
    return MakeDirOcc(MixHistoryAndCurrent(c, history.v, current.v, f, roughness));
}
NRD_D float ComputeAntilag(const ReblurCB& c, float history, float avg, float sigma, float accumSpeed) {
    float h = history, a = avg;
    float s = sigma * c.gAntilagParams.x;
    float magic = c.gAntilagParams.y * c.gFramerateScale * c.gFramerateScale;
    float hc = ColorClamp(a, s, h);
    float d = Div(Abs(h - hc), Max(h, hc) + NRD_EPS);
    return Rcp(1.0f + Div(d * accumSpeed, magic));
}
NRD_D void GetKernelBasis(float3 D, float3 N, float3& T, float3& B) {
    GetBasis(N, T, B);
    if (Abs(Dot(D, N)) < 0.999f) {
        float3 R = Reflect(-D, N);
        T = Normalize(Cross(N, R));
        B = Cross(R, T);
    }
}
NRD_D float2 GetTemporalAccumulationParams(const ReblurCB& c, float isInScreenMulFootprintQuality, float accumSpeed) {
    accumSpeed *= REBLUR_SAMPLES_PER_FRAME;
    float w = isInScreenMulFootprintQuality;
    w *= Div(accumSpeed, 1.0f + accumSpeed);
    return F2(w, 1.0f + 3.0f * c.gFramerateScale * w);
}

// ---- decoded guides ----------------------------------------------------------------------------------------------------
// NRD_FrontEnd_UnpackNormalAndRoughness costs ~60 VALU instructions (oct decode, rsqrt = sqrt + divide) and the spatial passes
// evaluate it at every tap of every pass. The executor therefore decodes IN_NORMAL_ROUGHNESS ONCE per frame into a float4 plane
// (N.xyz as computed by UnpackNormalAndRoughness, w = the roughness AS A FLOAT with the two materialID bits of the packed texel in bits 31:30 -- a
// roughness is in [0, 1], so its float has both clear) and the kernels fetch 16 bytes instead of re-deriving the normal: same values bit for bit, 4x
// the tap bytes (L2-served), ~55 instructions less per tap -- the kernels are VALU-bound (profiles/), not bandwidth-bound. Round 5: the roughness used
// to travel as its 10-bit integer and cost every specular tap a mask, a conversion and the exact division by 1023 (5 instructions; now one v_and).
// The encodings without material bits (every NRD_NORMAL_ENCODING but R10G10B10A2) keep the plain float there: materialID is 0 by definition (NRD.hlsli:617).
#if NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_R10G10B10A2_UNORM
NRD_D float DecodedRoughness(uint32_t w) { return AsFloat(w & 0x3FFFFFFFu); }
NRD_D float DecodedMaterialID(uint32_t w) { return NRD_DIV_3(float(w >> 30)) * 3.0f; } // = p.w * 3 of UnpackNormalAndRoughness
#else
NRD_D float DecodedRoughness(uint32_t w) { return AsFloat(w); }
NRD_D float DecodedMaterialID(uint32_t) { return 0.0f; }
#endif

// ---- the normal / roughness texel of the library's encoding (nrdmath.h NRD_NORMAL_ENCODING) -------------------------------------------------------------
// NrRaw = one undecoded texel of IN_NORMAL_ROUGHNESS, and of REBLUR's PREV_NORMAL_ROUGHNESS (reference Reblur.cpp:52-62: the pool plane follows the encoding):
//   encoding                0 RGBA8_UNORM   1 RGBA8_SNORM   2 R10G10B10A2_UNORM   3 RGBA16_UNORM   4 RGBA16_SNORM
//   IN_NORMAL_ROUGHNESS     RGBA8_UNORM     RGBA8_SNORM     R10_G10_B10_A2_UNORM  RGBA16_UNORM     RGBA16_SNORM      (the format the application binds: executor.hip)
//   PREV_NORMAL_ROUGHNESS   RGBA8_UNORM     RGBA8_SNORM     R10_G10_B10_A2_UNORM  RGBA16_UNORM     RGBA16_SFLOAT
// Decode*NormalRoughnessTexel return the float4 the reference's texture unit would hand to NRD_FrontEnd_UnpackNormalAndRoughness.
#if NRD_NORMAL_ENCODING <= NRD_NORMAL_ENCODING_R10G10B10A2_UNORM
typedef uint32_t NrRaw;
NRD_D NrRaw NrRawZero() { return 0u; }
#else
typedef uint2 NrRaw;
NRD_D NrRaw NrRawZero() { return make_uint2(0u, 0u); }
#endif
constexpr uint32_t NR_TEXEL_BYTES = (uint32_t)sizeof(NrRaw);
NRD_D NrRaw LoadNrRaw(const Plane& p, int x, int y) { return *TexelPtr<const NrRaw>(p, x, y); }
NRD_D void StoreNrRaw(const Plane& p, int x, int y, NrRaw v) { *TexelPtr<NrRaw>(p, x, y) = v; }
NRD_D float4 DecodeInNormalRoughnessTexel(NrRaw raw) {
#if NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_RGBA8_UNORM
    return DecodeRGBA8Unorm(raw);
#elif NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_RGBA8_SNORM
    return DecodeRGBA8Snorm(raw);
#elif NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_R10G10B10A2_UNORM
    return DecodeR10G10B10A2(raw);
#elif NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_RGBA16_UNORM
    return DecodeRGBA16Unorm(raw);
#else
    return DecodeRGBA16Snorm(raw);
#endif
}
NRD_D float4 DecodePrevNormalRoughnessTexel(NrRaw raw) {
#if NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_RGBA16_SNORM
    return DecodeRGBA16Float(raw);
#else
    return DecodeInNormalRoughnessTexel(raw);
#endif
}
// REBLUR PostBlur forwards the texel it read to PREV_NORMAL_ROUGHNESS (REBLUR_PostBlur.hlsli:47): a copy of the bits where the two planes share the format,
// a conversion of the four SNORM16 values to fp16 for encoding 4
NRD_D NrRaw InToPrevNormalRoughnessTexel(NrRaw raw) {
#if NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_RGBA16_SNORM
    const float4 v = DecodeRGBA16Snorm(raw);
    return make_uint2(FloatsToHalf2Bits(v.x, v.y), FloatsToHalf2Bits(v.z, v.w));
#else
    return raw;
#endif
}
// the roughness channel of a PREV_NORMAL_ROUGHNESS texel AS STORED (REBLUR_TemporalAccumulation.hlsli:463-467: GatherBlue for R10G10B10A2, GatherAlpha otherwise;
// not passed through the roughness encoding, as in the reference)
NRD_D float PrevNormalRoughnessTexelRoughness(NrRaw raw) {
#if NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_R10G10B10A2_UNORM
    return DecodeR10G10B10A2(raw).z;
#else
    return DecodePrevNormalRoughnessTexel(raw).w;
#endif
}
NRD_D float4 LoadInNormalRoughnessTexel(const Plane& p, int x, int y) { return DecodeInNormalRoughnessTexel(LoadNrRaw(p, x, y)); }

NRD_D float4 EncodeDecodedNormalRoughness(NrRaw raw) {
    float unused;
    float4 nr = UnpackNormalAndRoughness(DecodeInNormalRoughnessTexel(raw), unused);
#if NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_R10G10B10A2_UNORM
    return F4(nr.x, nr.y, nr.z, AsFloat(AsUint(nr.w) | (raw & 0xC0000000u)));
#else
    return nr;
#endif
}
NRD_D float4 DecodedToNormalRoughness(float4 d, float& materialID) {
    const uint32_t bits = AsUint(d.w);
    materialID = DecodedMaterialID(bits);
    return F4(d.x, d.y, d.z, DecodedRoughness(bits));
}
NRD_D float4 LoadDecodedNormalRoughness(const Plane& decoded, int x, int y, float& materialID) { return DecodedToNormalRoughness(LoadRGBA32F(decoded, x, y), materialID); }
// REBLUR lists keep the decoded guides as TWO planes (passes.h viewPos + roughnessWord): float4 (normal, viewZ) and the roughness | material word at a quarter of the
// pitch -- the spatial taps want exactly these (a diffuse tap one 16-byte load, a specular tap 4 bytes more), and every other REBLUR kernel reads its normal + roughness from
// the same pair, so the float4 (normal, word) plane of the RELAX lists is neither written nor read here (round 5: 16 B/px less per frame, one 59 MB plane less in the L2).
struct NormalRoughnessGuide {
    Plane nz, word;
};
NRD_D void ShareSize(NormalRoughnessGuide& g, const Plane& ref) { ShareSize(g.nz, ref), ShareSize(g.word, ref); }
inline bool SameSize(const NormalRoughnessGuide& g, const Plane& ref) { return SameSize(g.nz, ref) && SameSize(g.word, ref); }
// launcher side: nullptr, or why the pair the executor handed over cannot be used
inline const char* MakeNormalRoughnessGuide(const PassArgs& a, NormalRoughnessGuide& g) {
    g.nz = a.viewPos;
    g.word = a.roughnessWord;
    if (!g.nz.ptr || !g.word.ptr)
        return "REBLUR: the decoded guide planes are missing (IN_NORMAL_ROUGHNESS / IN_VIEWZ not bound?)";
    if (g.word.w != g.nz.w || g.word.h != g.nz.h || g.word.pitch * 4u != g.nz.pitch)
        return "REBLUR: internal error: the roughness-word plane does not match the (normal, viewZ) guide plane";
    return nullptr;
}
NRD_D float4 LoadDecodedNormalRoughness(const NormalRoughnessGuide& g, int x, int y, float& materialID, float& viewZ) {
    const uint32_t offset = TexelOffset(g.nz, x, y, 16u, true);
    const float4 t = *(const float4*)(g.nz.ptr + offset);
    const uint32_t w = *(const uint32_t*)(g.word.ptr + (offset >> 2));
    materialID = DecodedMaterialID(w);
    viewZ = t.w;
    return F4(t.x, t.y, t.z, DecodedRoughness(w));
}
NRD_D float4 LoadDecodedNormalRoughness(const NormalRoughnessGuide& g, int x, int y, float& materialID) {
    float unused;
    return LoadDecodedNormalRoughness(g, x, y, materialID, unused);
}
NRD_D float4 LoadDecodedNormalRoughness(const NormalRoughnessGuide& g, int x, int y) {
    float unused, unused2;
    return LoadDecodedNormalRoughness(g, x, y, unused, unused2);
}
NRD_D float4 LoadDecodedNormalRoughness(const Plane& decoded, int x, int y) {
    float unused;
    return LoadDecodedNormalRoughness(decoded, x, y, unused);
}
// Texture2D::Load semantics: outside the plane the packed texel reads as 0
NRD_D float4 LoadDecodedNormalRoughnessOrZero(const Plane& decoded, int x, int y, float& materialID) {
    if (InBounds(decoded, x, y))
        return LoadDecodedNormalRoughness(decoded, x, y, materialID);
    return UnpackNormalAndRoughness(F4(0.0f), materialID);
}

// ---- clamp-addressed fetches (what a clamp sampler / gather does at the border) ------------------------------------------
// clamp(x, a, b) for a <= b (every call site: 0 <= size - 1, or a non-empty rect). The compiler cannot know a <= b and emits v_max_i32 + v_min_i32; the median
// of three is ONE instruction and the same value. (The 4x4 / 12-texel footprints of the temporal passes clamp ~100 coordinates per pixel: 6 % of
// TemporalAccumulation's instructions.) Compile-time constant arguments keep the plain form so that they still fold.
#ifndef NRD_MED3_I32 // (the CPU emulation of these sources under tests/emu uses the plain form)
#define NRD_MED3_I32(r, x, a, b) asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(a), "v"(b))
#endif
NRD_D int ClampI(int x, int a, int b) {
    if (__builtin_constant_p(x) || (__builtin_constant_p(a) && __builtin_constant_p(b)))
        return x < a ? a : (x > b ? b : x);
    int r;
    NRD_MED3_I32(r, x, a, b);
    return r;
}
NRD_D float FetchClampedR32F(const Plane& p, int x, int y) { return LoadR32F(p, ClampI(x, 0, p.w - 1), ClampI(y, 0, p.h - 1)); }
NRD_D float FetchClampedR16F(const Plane& p, int x, int y) { return LoadR16F(p, ClampI(x, 0, p.w - 1), ClampI(y, 0, p.h - 1)); }
NRD_D float4 FetchClampedRGBA16F(const Plane& p, int x, int y) { return LoadRGBA16F(p, ClampI(x, 0, p.w - 1), ClampI(y, 0, p.h - 1)); }
NRD_D float4 FetchClampedR10G10B10A2(const Plane& p, int x, int y) { return LoadR10G10B10A2(p, ClampI(x, 0, p.w - 1), ClampI(y, 0, p.h - 1)); }
NRD_D uint32_t FetchClampedR16U(const Plane& p, int x, int y) { return LoadR16U(p, ClampI(x, 0, p.w - 1), ClampI(y, 0, p.h - 1)); }
// zero outside (Texture2D::Load / operator[])
NRD_D float4 LoadRGBA16FOrZero(const Plane& p, int x, int y) { return InBounds(p, x, y) ? LoadRGBA16F(p, x, y) : F4(0.0f); }
NRD_D float LoadR16FOrZero(const Plane& p, int x, int y) { return InBounds(p, x, y) ? LoadR16F(p, x, y) : 0.0f; }
NRD_D float4 LoadR10G10B10A2OrZero(const Plane& p, int x, int y) { return InBounds(p, x, y) ? LoadR10G10B10A2(p, x, y) : F4(0.0f); }

// ---- row-vector fetches of small footprints ----------------------------------------------------------------------------------------
// The temporal passes read 2x2 and 4x4 texel footprints at data-dependent positions. One load instruction per texel costs the L1 / address path as much
// as a 16-byte one (profiles/r02_c_gather_bench.txt), so an INTERIOR footprint (no coordinate needs clamping) is fetched row by row with one vector load
// per row: gfx950 takes dwordx2 / dwordx4 loads at any element-aligned address. Footprints that touch the border keep the per-texel clamped loads.
struct alignas(4) F32x4U { float v[4]; };
struct alignas(4) F32x2U { float v[2]; };
struct alignas(4) U32x2U { uint32_t v[2]; };
struct alignas(2) U16x4U { uint16_t v[4]; };
struct alignas(2) U16x2U { uint16_t v[2]; };
NRD_D bool FootprintIsInterior(const Plane& p, int x, int y, int nx, int ny) { return x >= 0 && y >= 0 && x + nx <= p.w && y + ny <= p.h; }
NRD_D float4 LoadRowR32Fx4(const Plane& p, int x, int y) {
    const F32x4U r = *(const F32x4U*)TexelPtr<const float>(p, x, y);
    return F4(r.v[0], r.v[1], r.v[2], r.v[3]);
}
NRD_D float2 LoadRowR32Fx2(const Plane& p, int x, int y) {
    const F32x2U r = *(const F32x2U*)TexelPtr<const float>(p, x, y);
    return F2(r.v[0], r.v[1]);
}
NRD_D void LoadRowR32Ux2(const Plane& p, int x, int y, uint32_t& a, uint32_t& b) {
    const U32x2U r = *(const U32x2U*)TexelPtr<const uint32_t>(p, x, y);
    a = r.v[0], b = r.v[1];
}
// two texels of a row in the library's normal encoding (NrRaw: 4 or 8 bytes per texel)
NRD_D void LoadRowNrRawx2(const Plane& p, int x, int y, NrRaw& a, NrRaw& b) {
#if NRD_NORMAL_ENCODING <= NRD_NORMAL_ENCODING_R10G10B10A2_UNORM
    LoadRowR32Ux2(p, x, y, a, b);
#else
    uint32_t v[4];
    __builtin_memcpy(v, TexelPtr<const NrRaw>(p, x, y), 16);
    a = make_uint2(v[0], v[1]), b = make_uint2(v[2], v[3]);
#endif
}
NRD_D void LoadRowR16Ux4(const Plane& p, int x, int y, uint32_t* out) {
    const U16x4U r = *(const U16x4U*)TexelPtr<const uint16_t>(p, x, y);
    out[0] = r.v[0], out[1] = r.v[1], out[2] = r.v[2], out[3] = r.v[3];
}
// undecoded rows: the halves are split where the values are used, so that a batch of requests contains no ALU work on loaded data (any such
// instruction makes the wave wait for every request issued before it)
NRD_D uint32_t LoadRowR16x2Raw(const Plane& p, int x, int y) {
    uint32_t v;
    __builtin_memcpy(&v, TexelPtr<const uint16_t>(p, x, y), 4);
    return v;
}
NRD_D uint2 LoadRowR16x4Raw(const Plane& p, int x, int y) {
    uint32_t v[2];
    __builtin_memcpy(v, TexelPtr<const uint16_t>(p, x, y), 8);
    return make_uint2(v[0], v[1]);
}
NRD_D void LoadRowR16Ux2(const Plane& p, int x, int y, uint32_t& a, uint32_t& b) {
    const U16x2U r = *(const U16x2U*)TexelPtr<const uint16_t>(p, x, y);
    a = r.v[0], b = r.v[1];
}

// SampleLevel( gNearestClamp, uv, 0 ): texel index
NRD_D int2 NearestTexel(const Plane& p, float2 uv) {
    return make_int2(ClampI((int)floorf(uv.x * float(p.w)), 0, p.w - 1), ClampI((int)floorf(uv.y * float(p.h)), 0, p.h - 1));
}

// SampleLevel( gLinearClamp, ... ) with the position in TEXEL units; weights in fp32, fixed evaluation order
struct LinearTaps {
    int x0, y0;
    float w00, w10, w01, w11;
};
NRD_D LinearTaps MakeLinearTaps(float2 pos) {
    float tx = pos.x - 0.5f, ty = pos.y - 0.5f;
    float fx0 = floorf(tx), fy0 = floorf(ty);
    float fx = tx - fx0, fy = ty - fy0;
    LinearTaps t;
    t.x0 = (int)fx0;
    t.y0 = (int)fy0;
    t.w00 = (1.0f - fx) * (1.0f - fy);
    t.w10 = fx * (1.0f - fy);
    t.w01 = (1.0f - fx) * fy;
    t.w11 = fx * fy;
    return t;
}
NRD_D float4 SampleLinearRGBA16F(const Plane& p, float2 pos) {
    LinearTaps t = MakeLinearTaps(pos);
    float4 s00 = FetchClampedRGBA16F(p, t.x0, t.y0), s10 = FetchClampedRGBA16F(p, t.x0 + 1, t.y0), s01 = FetchClampedRGBA16F(p, t.x0, t.y0 + 1),
           s11 = FetchClampedRGBA16F(p, t.x0 + 1, t.y0 + 1);
    return s00 * t.w00 + s10 * t.w10 + s01 * t.w01 + s11 * t.w11;
}
// ---- multi-GPU: how far from its own row does a pixel read LAST FRAME'S planes? -------------------------------------------------------------------------
// A row-strip rank holds last frame's planes only on its strip + a halo. The surface-motion reprojection is bounded by what nrdHipMeasureMotionRows measures, but the specular
// passes also read at the VIRTUAL-motion position and at look-back taps behind it (REBLUR_TemporalAccumulation.hlsli:455-610, RELAX_TemporalAccumulation.hlsli:420-520), whose
// distance depends on hit distances and curvature: no a-priori bound (profiles/r06_i_scaling_model_relax_ds_sh.json: one rank of four read 50+ rows beyond a 9-row surface motion).
// So the kernels REPORT it: uvY = the sample's uv.y in the previous frame's rect (clamped: a sample outside the screen is rejected, whatever its texels hold), the wave's maximum
// of |uvY * rectHeight - row| goes into *word with one atomicMax on the float's bits. The wave reduction walks the active lanes with v_readlane (exact under divergence: lanes
// that left the kernel do not take part) and only those that exceed what the word already holds: in steady state no lane does and the cost is a load, a compare and a ballot.
NRD_D float HistoryReachRows(float uvY, float rectHeight, int row) {
    const float d = fabsf(fminf(fmaxf(uvY, 0.0f), 1.0f) * rectHeight - (float(row) + 0.5f));
    return uvY == uvY ? d : 0.0f; // (a NaN position -- the direction of a zero motion -- is rejected by the in-screen tests like one outside the screen)
}
NRD_D void TrackHistoryReach(uint32_t* word, float rows) {
    if (!word) // (uniform: a kernel argument)
        return;
    const uint32_t bits = __float_as_uint(rows);
#ifdef NRD_EMU
    if (bits > *word) // (the CPU emulation of these sources has no wave intrinsics outside convergent code)
        atomicMax(word, bits);
#else
    const uint32_t seen = __builtin_nontemporal_load(word);
    unsigned long long mask = __ballot(bits > seen);
    if (!mask)
        return;
    uint32_t m = 0u;
    while (mask) {
        const int lane = __builtin_ctzll(mask);
        m = max(m, (uint32_t)__builtin_amdgcn_readlane((int)bits, lane));
        mask &= mask - 1ull;
    }
    if (bits == m) // the lane(s) holding the maximum
        atomicMax(word, m);
#endif
}

NRD_D NrRaw FetchClampedNrRaw(const Plane& p, int x, int y) { return LoadNrRaw(p, ClampI(x, 0, p.w - 1), ClampI(y, 0, p.h - 1)); }
// gPrev_Normal_Roughness.SampleLevel( gLinearClamp, ... ) of the encodings without the stochastic tap (REBLUR_TemporalAccumulation.hlsli:473, 593): the ENCODED texels are blended
NRD_D float4 SampleLinearPrevNormalRoughness(const Plane& p, float2 pos) {
    LinearTaps t = MakeLinearTaps(pos);
    float4 s00 = DecodePrevNormalRoughnessTexel(FetchClampedNrRaw(p, t.x0, t.y0)), s10 = DecodePrevNormalRoughnessTexel(FetchClampedNrRaw(p, t.x0 + 1, t.y0)),
           s01 = DecodePrevNormalRoughnessTexel(FetchClampedNrRaw(p, t.x0, t.y0 + 1)), s11 = DecodePrevNormalRoughnessTexel(FetchClampedNrRaw(p, t.x0 + 1, t.y0 + 1));
    return s00 * t.w00 + s10 * t.w10 + s01 * t.w01 + s11 * t.w11;
}
NRD_D float SampleLinearR16F(const Plane& p, float2 pos) {
    LinearTaps t = MakeLinearTaps(pos);
    float s00, s10, s01, s11;
    if (FootprintIsInterior(p, t.x0, t.y0, 2, 2)) {
        uint32_t a, b, c2, d;
        LoadRowR16Ux2(p, t.x0, t.y0, a, b);
        LoadRowR16Ux2(p, t.x0, t.y0 + 1, c2, d);
        s00 = HalfBitsToFloat((uint16_t)a), s10 = HalfBitsToFloat((uint16_t)b), s01 = HalfBitsToFloat((uint16_t)c2), s11 = HalfBitsToFloat((uint16_t)d);
    } else {
        s00 = FetchClampedR16F(p, t.x0, t.y0), s10 = FetchClampedR16F(p, t.x0 + 1, t.y0), s01 = FetchClampedR16F(p, t.x0, t.y0 + 1), s11 = FetchClampedR16F(p, t.x0 + 1, t.y0 + 1);
    }
    return s00 * t.w00 + s10 * t.w10 + s01 * t.w01 + s11 * t.w11;
}

// ---- history fetch: Catmull-Rom with fallback to custom-weight bilinear -----------------------------------------------------
// The shaders realise Catmull-Rom over the 4x4-minus-corners footprint as 5 bilinear texture fetches. A texture unit does the
// 20 texel reads for free; here every read is address arithmetic + a load + fp16 conversions, and TemporalAccumulation does
// 6 such fetches per pixel. Each fetch is therefore restated on the texels it actually blends (the other taps of fetches 0, 1,
// 3, 4 have weight exactly 0): 12 distinct texels, (k, j) = floor(samplePos - 0.5), clamp-addressed like the sampler;
//   fetch 0 / 4 : rows j-1 / j+2, columns k, k+1, fraction tc.x        fetch 1 / 3 : columns k-1 / k+2, rows j, j+1, fraction tc.y
//   fetch 2     : central 2x2 at (tc.x, tc.y)
// and the clamped coordinates are computed once per footprint and shared by every plane fetched through it.
struct HistoryFilter {
    float4 w;       // weights of fetches 0..3 (bicubic) or the custom bilinear weights of the 2x2 footprint
    float w4, sum;
    float4 cw;      // the Catmull-Rom weights of fetches 0..3 and (cw4) of fetch 4, before the choice
    float cw4;
    float2 tc;      // bilinear fractions inside the central 2x2
    int x[4], y[4]; // clamped texel coordinates k-1 .. k+2, j-1 .. j+2
    int ox, oy;     // origin of the shaders' Load-based bilinear path: int( centerPos ), truncation
    float4 bw;
    bool useBicubic;
};
// Two steps: the geometry (texel coordinates, fractions, Catmull-Rom weights) needs only the sample position -- TemporalAccumulation requests the texels
// of a footprint as soon as the reprojected position exists --, the choice between Catmull-Rom and the custom-weight bilinear fallback needs the
// occlusion tests that come later. Same arithmetic as the single-step form.
NRD_D HistoryFilter MakeHistoryGeometry(float2 samplePos, const Plane& dims) {
    const float S = NRD_CATROM_SHARPNESS;
    HistoryFilter h;
    float2 origin = Floor(samplePos - 0.5f);
    float2 centerPos = origin + 0.5f;
    float2 f = Sat(samplePos - centerPos);
    float2 w0 = f * (f * (f * -S + 2.0f * S) - S);
    float2 w1 = f * (f * (f * (2.0f - S) - (3.0f - S))) + 1.0f;
    float2 w2 = f * (f * (f * -(2.0f - S) + (3.0f - 2.0f * S)) + S);
    float2 w3 = f * (f * (f * S - S));
    float2 w12 = w1 + w2;
    h.cw = F4(w12.x * w0.y, w0.x * w12.y, w12.x * w12.y, w3.x * w12.y);
    h.cw4 = w12.x * w3.y;
    h.tc = Div(w2, w12);
    const int kx = (int)origin.x, ky = (int)origin.y;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        h.x[i] = ClampI(kx - 1 + i, 0, dims.w - 1);
        h.y[i] = ClampI(ky - 1 + i, 0, dims.h - 1);
    }
    h.ox = (int)centerPos.x;
    h.oy = (int)centerPos.y;
    return h;
}
NRD_D void SetHistoryWeights(HistoryFilter& h, float4 bilinearCustomWeights, bool useBicubic) {
    h.w = Select(useBicubic, h.cw, bilinearCustomWeights); // component-wise: `?:` on HIP vector types selects between ADDRESSES and pins the struct to scratch
    h.w4 = useBicubic ? h.cw4 : 0.0f;
    h.sum = Sum(h.w) + h.w4;
    h.bw = bilinearCustomWeights;
    h.useBicubic = useBicubic;
}
NRD_D HistoryFilter MakeHistoryFilter(float2 samplePos, float4 bilinearCustomWeights, bool useBicubic, const Plane& dims) {
    HistoryFilter h = MakeHistoryGeometry(samplePos, dims);
    SetHistoryWeights(h, bilinearCustomWeights, useBicubic);
    return h;
}
// V = float4 / float; load(plane, x, y) reads one in-bounds texel
template <typename V, typename LoadFn>
NRD_D V FetchHistoryGeneric(const HistoryFilter& h, const Plane& tex, LoadFn load, V zero) {
    const V t00 = load(tex, h.x[1], h.y[1]), t10 = load(tex, h.x[2], h.y[1]), t01 = load(tex, h.x[1], h.y[2]), t11 = load(tex, h.x[2], h.y[2]);
    V color;
    if (h.useBicubic) {
        const float fx = h.tc.x, fy = h.tc.y, gx = 1.0f - fx, gy = 1.0f - fy;
        V s0 = WSum(load(tex, h.x[1], h.y[0]), gx, load(tex, h.x[2], h.y[0]), fx);
        V s1 = WSum(load(tex, h.x[0], h.y[1]), gy, load(tex, h.x[0], h.y[2]), fy);
        V s2 = WSum(t00, gx * gy, t10, fx * gy, t01, gx * fy, t11, fx * fy);
        V s3 = WSum(load(tex, h.x[3], h.y[1]), gy, load(tex, h.x[3], h.y[2]), fy);
        V s4 = WSum(load(tex, h.x[1], h.y[3]), gx, load(tex, h.x[2], h.y[3]), fx);
        color = s0 * h.w.x;
        color = Mad(s1, h.w.y, color);
        color = Mad(s2, h.w.z, color);
        color = Mad(s3, h.w.w, color);
        color = Mad(s4, h.w4, color);
    } else {
        color = t00 * h.w.x;
        color = Mad(t10, h.w.y, color);
        color = Mad(t01, h.w.z, color);
        color = Mad(t11, h.w.w, color);
    }
    return h.sum < 0.0001f ? zero : Div(color, h.sum);
}
// Two / four horizontally adjacent RGBA16F texels with one 16-byte request each (8-byte aligned: legal for global_load_dwordx4).
// The temporal passes are limited by the number of L1 requests, not by bytes: one request per texel pair halves them.
struct alignas(8) RGBA16Fx2Raw {
    uint32_t v[4];
};
NRD_D float4 DecodeRGBA16F(uint32_t lo, uint32_t hi) {
    return F4(HalfBitsToFloat((uint16_t)(lo & 0xFFFFu)), HalfBitsToFloat((uint16_t)(lo >> 16)), HalfBitsToFloat((uint16_t)(hi & 0xFFFFu)), HalfBitsToFloat((uint16_t)(hi >> 16)));
}
NRD_D void LoadRGBA16Fx2(const Plane& p, int x, int y, float4& a, float4& b) {
    const RGBA16Fx2Raw raw = *(const RGBA16Fx2Raw*)TexelPtr<const uint2>(p, x, y);
    a = DecodeRGBA16F(raw.v[0], raw.v[1]);
    b = DecodeRGBA16F(raw.v[2], raw.v[3]);
}
NRD_D bool HistoryFootprintIsInterior(const HistoryFilter& h) { return h.x[3] - h.x[0] == 3 && h.y[3] - h.y[0] == 3; }
// The 12 texels of an interior footprint as 2 + 4 + 4 + 2 contiguous runs (6 requests of 16 bytes), undecoded. `loaded` is false for a footprint
// that touches the border of the plane (then the fetch reads its clamped texels one by one when it is evaluated).
struct Raw4 { // plain scalars (not a HIP vector type, whose union members end up in scratch when assigned under a condition)
    uint32_t x, y, z, w;
};
struct HistoryTexelsRGBA16F {
    Raw4 a, b0, b1, c0, c1, d; // two RGBA16F texels each
    bool loaded;
};
NRD_D Raw4 LoadRGBA16Fx2Raw(const Plane& p, int x, int y) {
    const RGBA16Fx2Raw raw = *(const RGBA16Fx2Raw*)TexelPtr<const uint2>(p, x, y);
    Raw4 r;
    r.x = raw.v[0], r.y = raw.v[1], r.z = raw.v[2], r.w = raw.v[3];
    return r;
}
NRD_D void PrefetchHistoryRGBA16F(const HistoryFilter& h, const Plane& tex, HistoryTexelsRGBA16F& t) {
    t.loaded = HistoryFootprintIsInterior(h);
    t.a = t.b0 = t.b1 = t.c0 = t.c1 = t.d = Raw4{0u, 0u, 0u, 0u};
    if (t.loaded) {
        t.a = LoadRGBA16Fx2Raw(tex, h.x[1], h.y[0]);
        t.b0 = LoadRGBA16Fx2Raw(tex, h.x[0], h.y[1]);
        t.b1 = LoadRGBA16Fx2Raw(tex, h.x[2], h.y[1]);
        t.c0 = LoadRGBA16Fx2Raw(tex, h.x[0], h.y[2]);
        t.c1 = LoadRGBA16Fx2Raw(tex, h.x[2], h.y[2]);
        t.d = LoadRGBA16Fx2Raw(tex, h.x[1], h.y[3]);
    }
}
// Window kernels (kernels_reblur_ta.hip MODE 1, kernels_relax_ta.hip): the two-step fetch with the texels taken from an LDS window of the previous frame (row
// stride winStride, first texel = plane texel (wx0, wy0)) instead of being requested from memory. A footprint on the border of
// the plane needs no special case: texel (i, j) is the texel at the clamped coordinate (h.x[i], h.y[j]) -- what FetchHistoryGeneric reads one by one -- and
// the blend of the row-loaded path is the same arithmetic as the generic one.
NRD_D void WindowHistoryTexels(const HistoryFilter& h, const uint2* win, int wx0, int wy0, int winStride, HistoryTexelsRGBA16F& t) {
    int xo[4], yo[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
        xo[i] = h.x[i] - wx0, yo[i] = (h.y[i] - wy0) * winStride;
    const uint2 a0 = win[yo[0] + xo[1]], a1 = win[yo[0] + xo[2]];
    const uint2 b0 = win[yo[1] + xo[0]], b1 = win[yo[1] + xo[1]], b2 = win[yo[1] + xo[2]], b3 = win[yo[1] + xo[3]];
    const uint2 c0 = win[yo[2] + xo[0]], c1 = win[yo[2] + xo[1]], c2 = win[yo[2] + xo[2]], c3 = win[yo[2] + xo[3]];
    const uint2 d0 = win[yo[3] + xo[1]], d1 = win[yo[3] + xo[2]];
    t.a = Raw4{a0.x, a0.y, a1.x, a1.y};
    t.b0 = Raw4{b0.x, b0.y, b1.x, b1.y}, t.b1 = Raw4{b2.x, b2.y, b3.x, b3.y};
    t.c0 = Raw4{c0.x, c0.y, c1.x, c1.y}, t.c1 = Raw4{c2.x, c2.y, c3.x, c3.y};
    t.d = Raw4{d0.x, d0.y, d1.x, d1.y};
    t.loaded = true;
}
NRD_D float4 FetchHistoryRGBA16F(const HistoryFilter& h, const Plane& tex, const HistoryTexelsRGBA16F& t) {
    if (!t.loaded)
        return FetchHistoryGeneric<float4>(h, tex, [](const Plane& p, int x, int y) { return LoadRGBA16F(p, x, y); }, F4(0.0f));
    const float4 b1 = DecodeRGBA16F(t.b0.z, t.b0.w), b2 = DecodeRGBA16F(t.b1.x, t.b1.y);
    const float4 c1 = DecodeRGBA16F(t.c0.z, t.c0.w), c2 = DecodeRGBA16F(t.c1.x, t.c1.y);
    float4 color;
    if (h.useBicubic) {
        const float4 a0 = DecodeRGBA16F(t.a.x, t.a.y), a1 = DecodeRGBA16F(t.a.z, t.a.w);
        const float4 b0 = DecodeRGBA16F(t.b0.x, t.b0.y), b3 = DecodeRGBA16F(t.b1.z, t.b1.w);
        const float4 c0 = DecodeRGBA16F(t.c0.x, t.c0.y), c3 = DecodeRGBA16F(t.c1.z, t.c1.w);
        const float4 d0 = DecodeRGBA16F(t.d.x, t.d.y), d1 = DecodeRGBA16F(t.d.z, t.d.w);
        const float fx = h.tc.x, fy = h.tc.y, gx = 1.0f - fx, gy = 1.0f - fy;
        float4 s0 = WSum(a0, gx, a1, fx);
        float4 s1 = WSum(b0, gy, c0, fy);
        float4 s2 = WSum(b1, gx * gy, b2, fx * gy, c1, gx * fy, c2, fx * fy);
        float4 s3 = WSum(b3, gy, c3, fy);
        float4 s4 = WSum(d0, gx, d1, fx);
        color = s0 * h.w.x;
        color = Mad(s1, h.w.y, color);
        color = Mad(s2, h.w.z, color);
        color = Mad(s3, h.w.w, color);
        color = Mad(s4, h.w4, color);
    } else { // the custom-weight bilinear fallback blends the central 2x2 (FetchHistoryGeneric, same order)
        color = b1 * h.w.x;
        color = Mad(b2, h.w.y, color);
        color = Mad(c1, h.w.z, color);
        color = Mad(c2, h.w.w, color);
    }
    return h.sum < 0.0001f ? F4(0.0f) : Div(color, h.sum);
}
NRD_D float4 FetchHistoryRGBA16F(const HistoryFilter& h, const Plane& tex) {
    // interior footprint (no coordinate was clamped): the 12 texels are 2 + 4 + 4 + 2 contiguous runs -> 6 requests instead of 12
    const bool interior = HistoryFootprintIsInterior(h);
    if (!(interior && h.useBicubic))
        return FetchHistoryGeneric<float4>(h, tex, [](const Plane& p, int x, int y) { return LoadRGBA16F(p, x, y); }, F4(0.0f));
    HistoryTexelsRGBA16F t;
    PrefetchHistoryRGBA16F(h, tex, t);
    return FetchHistoryRGBA16F(h, tex, t);
}
NRD_D float FetchHistoryR16F(const HistoryFilter& h, const Plane& tex) {
    // interior bicubic footprint: the 12 texels are rows of 2 + 4 + 4 + 2 -> four row loads (4 / 8 / 8 / 4 bytes) instead of twelve 2-byte ones
    const bool interior = h.x[3] - h.x[0] == 3 && h.y[3] - h.y[0] == 3;
    if (!(interior && h.useBicubic))
        return FetchHistoryGeneric<float>(h, tex, [](const Plane& p, int x, int y) { return LoadR16F(p, x, y); }, 0.0f);
    uint32_t a0, a1, b[4], c2[4], d0, d1;
    LoadRowR16Ux2(tex, h.x[1], h.y[0], a0, a1);
    LoadRowR16Ux4(tex, h.x[0], h.y[1], b);
    LoadRowR16Ux4(tex, h.x[0], h.y[2], c2);
    LoadRowR16Ux2(tex, h.x[1], h.y[3], d0, d1);
#define NRD_H(v) HalfBitsToFloat((uint16_t)(v))
    const float fx = h.tc.x, fy = h.tc.y, gx = 1.0f - fx, gy = 1.0f - fy;
    float s0 = NRD_H(a0) * gx + NRD_H(a1) * fx;
    float s1 = NRD_H(b[0]) * gy + NRD_H(c2[0]) * fy;
    float s2 = NRD_H(b[1]) * (gx * gy) + NRD_H(b[2]) * (fx * gy) + NRD_H(c2[1]) * (gx * fy) + NRD_H(c2[2]) * (fx * fy);
    float s3 = NRD_H(b[3]) * gy + NRD_H(c2[3]) * fy;
    float s4 = NRD_H(d0) * gx + NRD_H(d1) * fx;
#undef NRD_H
    float color = s0 * h.w.x;
    color = color + s1 * h.w.y;
    color = color + s2 * h.w.z;
    color = color + s3 * h.w.w;
    color = color + s4 * h.w4;
    return h.sum < 0.0001f ? 0.0f : Div(color, h.sum);
}
// custom-weight bilinear fetch of an RGBA16F plane (the SH1 histories; reference REBLUR_Common.hlsli:350-361 fetches them this way)
NRD_D float4 FetchHistoryBilinearRGBA16F(const HistoryFilter& h, const Plane& tex) {
    auto at = [&](int x, int y) { return InBounds(tex, x, y) ? LoadRGBA16F(tex, x, y) : F4(0.0f); };
    float4 color = at(h.ox, h.oy) * h.bw.x;
    color = Mad(at(h.ox + 1, h.oy), h.bw.y, color);
    color = Mad(at(h.ox, h.oy + 1), h.bw.z, color);
    color = Mad(at(h.ox + 1, h.oy + 1), h.bw.w, color);
    float s = Sum(h.bw);
    return s < 0.0001f ? F4(0.0f) : Div(color, s);
}
struct BilinearTexelsR16F { // the 2x2 footprint of the Load-based bilinear path as two undecoded 4-byte rows; `loaded` as above
    uint32_t r0, r1;
    bool loaded;
};
NRD_D void PrefetchBilinearR16F(const Plane& tex, int ox, int oy, BilinearTexelsR16F& t) {
    t.loaded = FootprintIsInterior(tex, ox, oy, 2, 2);
    t.r0 = t.r1 = 0u;
    if (t.loaded) {
        t.r0 = LoadRowR16x2Raw(tex, ox, oy);
        t.r1 = LoadRowR16x2Raw(tex, ox, oy + 1);
    }
}
NRD_D float FetchHistoryBilinearR16F(const HistoryFilter& h, const Plane& tex, const BilinearTexelsR16F& t) {
    float s00, s10, s01, s11;
    if (t.loaded) { // two 4-byte row loads instead of four 2-byte ones (same texels)
        s00 = HalfBitsToFloat((uint16_t)(t.r0 & 0xFFFFu)), s10 = HalfBitsToFloat((uint16_t)(t.r0 >> 16)), s01 = HalfBitsToFloat((uint16_t)(t.r1 & 0xFFFFu)), s11 = HalfBitsToFloat((uint16_t)(t.r1 >> 16));
    } else {
        s00 = LoadR16FOrZero(tex, h.ox, h.oy), s10 = LoadR16FOrZero(tex, h.ox + 1, h.oy), s01 = LoadR16FOrZero(tex, h.ox, h.oy + 1), s11 = LoadR16FOrZero(tex, h.ox + 1, h.oy + 1);
    }
    float color = s00 * h.bw.x;
    color += s10 * h.bw.y;
    color += s01 * h.bw.z;
    color += s11 * h.bw.w;
    float s = Sum(h.bw);
    return s < 0.0001f ? 0.0f : Div(color, s);
}
NRD_D float FetchHistoryBilinearR16F(const HistoryFilter& h, const Plane& tex) {
    BilinearTexelsR16F t;
    PrefetchBilinearR16F(tex, h.ox, h.oy, t);
    return FetchHistoryBilinearR16F(h, tex, t);
}


// REBLUR_TYPE and its storage (reference Reblur.cpp:38-45): radiance + hit distance in RGBA16F with an R16F fast history; the hit distance
// alone in R16_UNORM with an R16_UNORM fast history (occlusion family); direction + hit distance in RGBA16_SNORM with an R16_UNORM fast
// history (directional occlusion)
template <int KIND>
struct ReblurSignal;
template <>
struct ReblurSignal<SIGNAL_RADIANCE> {
    typedef float4 type;
    static NRD_D float4 Zero() { return F4(0.0f); }
    static NRD_D float4 Load(const Plane& p, int x, int y) { return LoadRGBA16F(p, x, y); }
    // Denanify( w, s ) of the tap loops (reference Common.hlsli:219: `w == 0 ? 0 : s`): the selection is made on the two raw dwords, before the fp16 -> fp32
    // conversions -- 2 selects instead of 4, and the conversions then sit directly in front of the accumulating multiply-adds, where the compiler folds
    // them into v_fma_mix_f32 (the conversion of an fp16 value is exact, so cvt + fma and fma_mix are the same number). Same values bit for bit.
    static NRD_D float4 LoadOrZero(const Plane& p, int x, int y, bool zero) {
#if NRD_EXPERIMENT_LEGACY_DENANIFY // A/B builds only (tools/build_variant.py): the selection after the conversions, as up to round 3
        return Select(zero, F4(0.0f), LoadRGBA16F(p, x, y));
#endif
        uint2 raw = *TexelPtr<const uint2>(p, x, y);
        raw.x = zero ? 0u : raw.x;
        raw.y = zero ? 0u : raw.y;
        return F4(HalfBitsToFloat((uint16_t)(raw.x & 0xFFFFu)), HalfBitsToFloat((uint16_t)(raw.x >> 16)), HalfBitsToFloat((uint16_t)(raw.y & 0xFFFFu)), HalfBitsToFloat((uint16_t)(raw.y >> 16)));
    }
    static NRD_D void Store(const Plane& p, int x, int y, float4 v) { StoreRGBA16F(p, x, y, v); }
    static NRD_D float4 WithHitDist(float4 s, float hitDist) { return F4(s.x, s.y, s.z, hitDist); }
    static NRD_D float4 FetchHistory(const HistoryFilter& h, const Plane& tex) { return FetchHistoryRGBA16F(h, tex); }
    static NRD_D float LoadFast(const Plane& p, int x, int y) { return LoadR16F(p, x, y); }
    static NRD_D void StoreFast(const Plane& p, int x, int y, float v) { StoreR16F(p, x, y, v); }
    static NRD_D float FetchFastBilinear(const HistoryFilter& h, const Plane& tex) { return FetchHistoryBilinearR16F(h, tex); }
    // two-step fetches (texels requested early, blended late)
    typedef HistoryTexelsRGBA16F HistoryTexels;
    typedef BilinearTexelsR16F FastTexels;
    static NRD_D void PrefetchHistory(const HistoryFilter& h, const Plane& tex, HistoryTexels& t, bool enable) { // enable = false: blend from single-texel loads
        t.loaded = false;
        t.a = t.b0 = t.b1 = t.c0 = t.c1 = t.d = Raw4{0u, 0u, 0u, 0u};
        if (enable)
            PrefetchHistoryRGBA16F(h, tex, t);
    }
    static NRD_D float4 FetchHistory(const HistoryFilter& h, const Plane& tex, const HistoryTexels& t) { return FetchHistoryRGBA16F(h, tex, t); }
    static NRD_D void PrefetchFast(const HistoryFilter& h, const Plane& tex, FastTexels& t) { PrefetchBilinearR16F(tex, h.ox, h.oy, t); }
    static NRD_D float FetchFastBilinear(const HistoryFilter& h, const Plane& tex, const FastTexels& t) { return FetchHistoryBilinearR16F(h, tex, t); }
};
NRD_D float FetchHistoryBilinearR16Unorm(const HistoryFilter& h, const Plane& tex) {
    auto at = [&](int x, int y) { return InBounds(tex, x, y) ? LoadR16Unorm(tex, x, y) : 0.0f; };
    float color = at(h.ox, h.oy) * h.bw.x;
    color += at(h.ox + 1, h.oy) * h.bw.y;
    color += at(h.ox, h.oy + 1) * h.bw.z;
    color += at(h.ox + 1, h.oy + 1) * h.bw.w;
    float s = Sum(h.bw);
    return s < 0.0001f ? 0.0f : Div(color, s);
}
template <>
struct ReblurSignal<SIGNAL_OCCLUSION> {
    typedef float type;
    static NRD_D float Zero() { return 0.0f; }
    static NRD_D float Load(const Plane& p, int x, int y) { return LoadR16Unorm(p, x, y); }
    static NRD_D float LoadOrZero(const Plane& p, int x, int y, bool zero) { return zero ? 0.0f : LoadR16Unorm(p, x, y); }
    static NRD_D void Store(const Plane& p, int x, int y, float v) { StoreR16Unorm(p, x, y, v); }
    static NRD_D float WithHitDist(float, float hitDist) { return hitDist; }
    static NRD_D float FetchHistory(const HistoryFilter& h, const Plane& tex) {
        return FetchHistoryGeneric<float>(h, tex, [](const Plane& p, int x, int y) { return LoadR16Unorm(p, x, y); }, 0.0f);
    }
    static NRD_D float LoadFast(const Plane& p, int x, int y) { return LoadR16Unorm(p, x, y); }
    static NRD_D void StoreFast(const Plane& p, int x, int y, float v) { StoreR16Unorm(p, x, y, v); }
    static NRD_D float FetchFastBilinear(const HistoryFilter& h, const Plane& tex) { return FetchHistoryBilinearR16Unorm(h, tex); }
    // no early requests for this storage: the two-step interface falls back to the plain fetch
    struct HistoryTexels {};
    struct FastTexels {};
    static NRD_D void PrefetchHistory(const HistoryFilter&, const Plane&, HistoryTexels&, bool) {}
    static NRD_D type FetchHistory(const HistoryFilter& h, const Plane& tex, const HistoryTexels&) { return FetchHistory(h, tex); }
    static NRD_D void PrefetchFast(const HistoryFilter&, const Plane&, FastTexels&) {}
    static NRD_D float FetchFastBilinear(const HistoryFilter& h, const Plane& tex, const FastTexels&) { return FetchFastBilinear(h, tex); }
};
template <>
struct ReblurSignal<SIGNAL_DIRECTIONAL_OCCLUSION> {
    typedef DirOcc type;
    static NRD_D DirOcc Zero() { return MakeDirOcc(F4(0.0f)); }
    static NRD_D DirOcc Load(const Plane& p, int x, int y) { return MakeDirOcc(LoadRGBA16Snorm(p, x, y)); }
    static NRD_D DirOcc LoadOrZero(const Plane& p, int x, int y, bool zero) { return Select(zero, Zero(), Load(p, x, y)); }
    static NRD_D void Store(const Plane& p, int x, int y, DirOcc s) { StoreRGBA16Snorm(p, x, y, s.v); }
    static NRD_D DirOcc WithHitDist(DirOcc s, float hitDist) { return MakeDirOcc(F4(s.v.x, s.v.y, s.v.z, hitDist)); }
    static NRD_D DirOcc FetchHistory(const HistoryFilter& h, const Plane& tex) {
        return MakeDirOcc(FetchHistoryGeneric<float4>(h, tex, [](const Plane& p, int x, int y) { return LoadRGBA16Snorm(p, x, y); }, F4(0.0f)));
    }
    static NRD_D float LoadFast(const Plane& p, int x, int y) { return LoadR16Unorm(p, x, y); }
    static NRD_D void StoreFast(const Plane& p, int x, int y, float v) { StoreR16Unorm(p, x, y, v); }
    static NRD_D float FetchFastBilinear(const HistoryFilter& h, const Plane& tex) { return FetchHistoryBilinearR16Unorm(h, tex); }
    // no early requests for this storage: the two-step interface falls back to the plain fetch
    struct HistoryTexels {};
    struct FastTexels {};
    static NRD_D void PrefetchHistory(const HistoryFilter&, const Plane&, HistoryTexels&, bool) {}
    static NRD_D type FetchHistory(const HistoryFilter& h, const Plane& tex, const HistoryTexels&) { return FetchHistory(h, tex); }
    static NRD_D void PrefetchFast(const HistoryFilter&, const Plane&, FastTexels&) {}
    static NRD_D float FetchFastBilinear(const HistoryFilter& h, const Plane& tex, const FastTexels&) { return FetchFastBilinear(h, tex); }
};

} // namespace nrdhip
