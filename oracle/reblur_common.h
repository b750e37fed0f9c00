// ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/hlsl.h).
// REBLUR shared pieces: constant block, storage packing, small helpers.
//   constants : reference Shaders/Include/REBLUR_Config.hlsli:113-186
//   helpers   : reference Shaders/Include/REBLUR_Common.hlsli:13-274
#pragma once

#include "ml.h"
#include "tex.h"

namespace orc {

struct ReblurCB {
    float4x4 gWorldToClip, gViewToClip, gViewToWorld, gWorldToViewPrev, gWorldToClipPrev, gWorldPrevToWorld;
    float4 gRotatorPre, gRotator, gRotatorPost, gFrustum, gFrustumPrev, gCameraDelta, gHitDistParams, gViewVectorWorld, gViewVectorWorldPrev, gMvScale;
    float2 gAntilagParams, gResourceSize, gResourceSizeInv, gResourceSizeInvPrev, gRectSize, gRectSizeInv, gRectSizePrev, gResolutionScale, gResolutionScalePrev,
        gRectOffset, gSpecProbabilityThresholdsForMvModification, gJitter;
    uint32_t gPrintfAt[2], gRectOrigin[2];
    int gRectSizeMinusOne[2];
    float gDisocclusionThreshold, gDisocclusionThresholdAlternate, gCameraAttachedReflectionMaterialID, gStrandMaterialID, gStrandThickness,
        gStabilizationStrength, gHitDistStabilizationStrength, gDebug, gOrthoMode, gUnproject, gDenoisingRange, gPlaneDistSensitivity, gFramerateScale,
        gMinBlurRadius, gMaxBlurRadius, gDiffPrepassBlurRadius, gSpecPrepassBlurRadius, gMaxAccumulatedFrameNum, gMaxFastAccumulatedFrameNum, gAntiFirefly,
        gLobeAngleFraction, gRoughnessFraction, gResponsiveAccumulationRoughnessThreshold, gHistoryFixFrameNum, gHistoryFixBasePixelStride,
        gMinRectDimMulUnproject, gUsePrepassNotOnlyForSpecularMotionEstimation, gSplitScreen, gSplitScreenPrev, gCheckerboardResolveAccumSpeed, gViewZScale,
        gFireflySuppressorMinRelativeScale, gMinHitDistanceWeight, gDiffMinMaterial, gSpecMinMaterial;
    uint32_t gHasHistoryConfidence, gHasDisocclusionThresholdMix, gDiffCheckerboard, gSpecCheckerboard, gFrameIndex, gIsRectChanged, gResetHistory;
};
static_assert(sizeof(ReblurCB) == 832, "REBLUR constant block");

// REBLUR_Config.hlsli:58-61
constexpr float REBLUR_MAX_ACCUM_FRAME_NUM = 63.0f; // 6 bits
constexpr float REBLUR_MAX_MATERIALID_NUM = 15.0f;  // 4 bits
// REBLUR_Config.hlsli:63-98
constexpr float REBLUR_PRE_BLUR_FRACTION_SCALE = 2.0f;
constexpr float REBLUR_PRE_BLUR_NON_LINEAR_ACCUM_SPEED = 1.0f / (1.0f + 10.0f);
constexpr float REBLUR_BLUR_FRACTION_SCALE = 1.0f;
constexpr float REBLUR_POST_BLUR_FRACTION_SCALE = 0.5f;
constexpr float REBLUR_POST_BLUR_RADIUS_SCALE = 2.0f;
constexpr float REBLUR_NORMAL_ULP = NRD_NORMAL_ENCODING_ERROR;
constexpr float REBLUR_ALMOST_ZERO_ANGLE = 0.01745240643728351f; // cos( 89 deg )
constexpr float REBLUR_FIREFLY_SUPPRESSOR_MAX_RELATIVE_INTENSITY = 38.0f;
constexpr float REBLUR_FIREFLY_SUPPRESSOR_RADIUS_SCALE = 0.1f;
constexpr float REBLUR_FIREFLY_SUPPRESSOR_FAST_RELATIVE_INTENSITY = 4.0f;
constexpr int REBLUR_ANTI_FIREFLY_FILTER_RADIUS = 4;
constexpr float REBLUR_ANTI_FIREFLY_SIGMA_SCALE = 2.0f;
constexpr float REBLUR_ROUGHNESS_SENSITIVITY_IN_TA = NRD_ROUGHNESS_SENSITIVITY * 0.3f;
constexpr float REBLUR_SAMPLES_PER_FRAME = 1.0f;
constexpr float REBLUR_MAX_PERCENT_OF_LOBE_VOLUME_FOR_PRE_PASS = 0.3f;
constexpr float REBLUR_COLOR_CLAMPING_SIGMA_SCALE = 2.0f;           // radiance signals
constexpr float REBLUR_COLOR_CLAMPING_SIGMA_SCALE_OCCLUSION = 1.0f; // REBLUR_OCCLUSION (REBLUR_Config.hlsli:94-98)

enum SpatialMode { PRE_BLUR = 0, BLUR = 1, POST_BLUR = 2 };

// ---- storage packing: REBLUR_Common.hlsli:13-80 ; Packing::RgbaToUint / UintToRgba with 6,6,4,0 bits [ml] ---------
inline uint32_t PackInternalData(float diffAccumSpeed, float specAccumSpeed, float materialID) {
    float tx = Div(diffAccumSpeed, REBLUR_MAX_ACCUM_FRAME_NUM), ty = Div(specAccumSpeed, REBLUR_MAX_ACCUM_FRAME_NUM), tz = Div(materialID, REBLUR_MAX_MATERIALID_NUM);
    uint32_t p = (uint32_t)floorf(saturate(tx) * 63.0f + 0.5f);
    p |= (uint32_t)floorf(saturate(ty) * 63.0f + 0.5f) << 6;
    p |= (uint32_t)floorf(saturate(tz) * 15.0f + 0.5f) << 12;
    return p;
}
inline float3 UnpackInternalData(uint32_t p) {
    float3 t = float3(float(p & 63u) / 63.0f, float((p >> 6) & 63u) / 63.0f, float((p >> 12) & 15u) / 15.0f); // UNORM decode: the exact quotient (tex.h)
    t.x *= REBLUR_MAX_ACCUM_FRAME_NUM;
    t.y *= REBLUR_MAX_ACCUM_FRAME_NUM;
    t.z *= REBLUR_MAX_MATERIALID_NUM;
    return t;
}
// DATA1 is RG8 (diffuse+specular) or R8 (single signal: both channels alias .x)
inline float2 PackData1(float diffAccumSpeed, float specAccumSpeed, bool hasDiff) {
    float2 r = float2(saturate(Div(diffAccumSpeed, REBLUR_MAX_ACCUM_FRAME_NUM)), saturate(Div(specAccumSpeed, REBLUR_MAX_ACCUM_FRAME_NUM)));
    if (!hasDiff)
        r.x = r.y;
    return r;
}
inline float2 UnpackData1(float4 texel, bool hasDiff) {
    float2 p = float2(texel.x, texel.y);
    if (!hasDiff)
        p.y = p.x;
    return p * REBLUR_MAX_ACCUM_FRAME_NUM;
}
inline uint32_t PackData2(float fbits, float curvature, float virtualHistoryAmount) {
    uint32_t p = (uint32_t)(fbits + 0.5f);
    p |= (uint32_t)(saturate(virtualHistoryAmount) * 255.0f + 0.5f) << 8;
    p |= f32tof16(curvature) << 16;
    return p;
}
inline float2 UnpackData2(uint32_t p, uint32_t& bits) {
    bits = p & 0xFFu;
    return float2(float((p >> 8) & 0xFFu) / 255.0f, f16tof32(p >> 16)); // UNORM decode: the exact quotient
}

// ---- helpers: REBLUR_Common.hlsli:84-274 --------------------------------------------------------------------------
inline float UnpackViewZ(const ReblurCB& c, float z) { return fabsf(z * c.gViewZScale); } // Common.hlsli:233
inline float3 GetViewVector(const ReblurCB& c, float3 X, bool isViewSpace = false) {
    return c.gOrthoMode == 0.0f ? normalize(-X) : (isViewSpace ? float3(0, 0, -1) : c.gViewVectorWorld.xyz());
}
inline float3 GetViewVectorPrev(const ReblurCB& c, float3 Xprev, float3 cameraDelta) {
    return c.gOrthoMode == 0.0f ? normalize(cameraDelta - Xprev) : c.gViewVectorWorldPrev.xyz();
}
inline float GetMinAllowedLimitForHitDistNonLinearAccumSpeed(const ReblurCB& c, float roughness) {
    float frameNum = 0.5f * GetSpecMagicCurve(roughness) * c.gMaxAccumulatedFrameNum;
    return Rcp(1.0f + frameNum);
}
inline float GetFadeBasedOnAccumulatedFrames(const ReblurCB& c, float accumSpeed) {
    float a = DivConst(c.gHistoryFixFrameNum * 2.0f, 3.0f) + 1e-6f;
    float b = DivConst(c.gHistoryFixFrameNum * 4.0f, 3.0f) + 2e-6f;
    return Math::LinearStep(a, b, accumSpeed);
}
inline float GetNonLinearAccumSpeed(const ReblurCB& c, float accumSpeed, float maxAccumSpeed, float confidence, bool hasData) { // REBLUR_Common.hlsli:111-124
    float nonLinearAccumSpeed = max(1.0f - confidence, Rcp(1.0f + min(accumSpeed, maxAccumSpeed)));
    if (!hasData)
        nonLinearAccumSpeed *= lerp(1.0f - c.gCheckerboardResolveAccumSpeed, 1.0f, nonLinearAccumSpeed);
    return nonLinearAccumSpeed;
}
inline float RemapRoughnessToResponsiveFactor(const ReblurCB& c, float roughness) {
    float amount = Div(roughness + NRD_EPS, c.gResponsiveAccumulationRoughnessThreshold + NRD_EPS);
    return Math::SmoothStep01(amount);
}
inline float GetLumaScale(float currLuma, float newLuma) { return Div(newLuma + NRD_EPS, currLuma + NRD_EPS); }
inline float4 MixHistoryAndCurrent(const ReblurCB& c, float4 history, float4 current, float f, float roughness = 1.0f) {
    float4 r;
    r.x = lerp(history.x, current.x, f);
    r.y = lerp(history.y, current.y, f);
    r.z = lerp(history.z, current.z, f);
    r.w = lerp(history.w, current.w, max(f, GetMinAllowedLimitForHitDistNonLinearAccumSpeed(c, roughness)));
    return r;
}
inline float GetLuma(float4 v) { return v.x; } // REBLUR_USE_YCOCG = 1
inline float4 ChangeLuma(float4 v, float newLuma) {
    float s = GetLumaScale(GetLuma(v), newLuma);
    return float4(v.x * s, v.y * s, v.z * s, v.w);
}
inline float4 ClampNegativeToZero(float4 v) {
    float3 rgb = _NRD_LinearToYCoCg(_NRD_YCoCgToLinear(v.xyz()));
    return float4(rgb, saturate(v.w));
}
// REBLUR_TYPE (REBLUR_Config.hlsli:100-105) and its overloaded helpers (REBLUR_Common.hlsli:148-215). Three kinds of signal:
//   0 radiance            float4 (YCoCg radiance, normalised hit distance), RGBA16F
//   1 occlusion           float  (the normalised hit distance alone), R16_UNORM                      -- REBLUR_OCCLUSION
//   2 directional occl.   float4 (direction * hit distance?, normalised hit distance), RGBA16_SNORM  -- REBLUR_DIRECTIONAL_OCCLUSION:
//                         same arithmetic as radiance except that the "luma" of the signal is its .w
enum { SIGNAL_RADIANCE = 0, SIGNAL_OCCLUSION = 1, SIGNAL_DIRECTIONAL_OCCLUSION = 2 };
struct DirOcc { // distinct type so that GetLuma / ChangeLuma / ClampNegativeToZero can be overloaded
    float4 v;
    DirOcc() {}
    explicit DirOcc(float a) : v(a) {}
    explicit DirOcc(float4 a) : v(a) {}
    operator float4() const { return v; }
};
inline DirOcc operator+(DirOcc a, DirOcc b) { return DirOcc(a.v + b.v); }
inline DirOcc operator*(DirOcc a, float b) { return DirOcc(a.v * b); }
inline DirOcc Mad(DirOcc a, float s, DirOcc c) { return DirOcc(Mad(a.v, s, c.v)); }
inline DirOcc lerp(DirOcc a, DirOcc b, float t) { return DirOcc(lerp(a.v, b.v, t)); }

template <int KIND> struct ReblurSignal;
template <> struct ReblurSignal<SIGNAL_RADIANCE> {
    typedef float4 type;
    static float4 From(float4 texel) { return texel; }
    static float4 WithHitDist(float4 s, float hitDist) { return float4(s.x, s.y, s.z, hitDist); }
};
template <> struct ReblurSignal<SIGNAL_OCCLUSION> {
    typedef float type;
    static float From(float4 texel) { return texel.x; }
    static float WithHitDist(float, float hitDist) { return hitDist; }
};
template <> struct ReblurSignal<SIGNAL_DIRECTIONAL_OCCLUSION> {
    typedef DirOcc type;
    static DirOcc From(float4 texel) { return DirOcc(texel); }
    static DirOcc WithHitDist(DirOcc s, float hitDist) { return DirOcc(float4(s.v.x, s.v.y, s.v.z, hitDist)); }
};
template <typename S> struct SignalKind { enum { value = SIGNAL_RADIANCE }; };
template <> struct SignalKind<float> { enum { value = SIGNAL_OCCLUSION }; };
template <> struct SignalKind<DirOcc> { enum { value = SIGNAL_DIRECTIONAL_OCCLUSION }; };

inline float ExtractHitDist(float4 v) { return v.w; }
inline float ExtractHitDist(float v) { return v; }
inline float ExtractHitDist(DirOcc s) { return s.v.w; }
inline float GetLuma(float v) { return v; }
inline float GetLuma(DirOcc s) { return s.v.w; }
inline float ChangeLuma(float, float newLuma) { return newLuma; }
inline DirOcc ChangeLuma(DirOcc s, float newLuma) {
    float k = GetLumaScale(s.v.w, newLuma);
    return DirOcc(float4(s.v.x * k, s.v.y * k, s.v.z * k, newLuma));
}
inline float ClampNegativeToZero(float v) { return saturate(v); } // ClampNegativeHitDistToZero
inline DirOcc ClampNegativeToZero(DirOcc s) { return ChangeLuma(s, saturate(s.v.w)); }
inline float MixHistoryAndCurrent(const ReblurCB& c, float history, float current, float f, float roughness = 1.0f) {
    return lerp(history, current, max(f, GetMinAllowedLimitForHitDistNonLinearAccumSpeed(c, roughness)));
}
inline DirOcc MixHistoryAndCurrent(const ReblurCB& c, DirOcc history, DirOcc current, float f, float roughness = 1.0f) {
    return DirOcc(MixHistoryAndCurrent(c, history.v, current.v, f, roughness));
}

inline float ComputeAntilag(const ReblurCB& c, float history, float avg, float sigma, float accumSpeed) { // REBLUR_ANTILAG_MODE = 2
    float h = history, a = avg;
    float s = sigma * c.gAntilagParams.x;
    float magic = c.gAntilagParams.y * c.gFramerateScale * c.gFramerateScale;
    float hc = Color::Clamp(a, s, h);
    float d = Div(fabsf(h - hc), max(h, hc) + NRD_EPS);
    return Rcp(1.0f + Div(d * accumSpeed, magic));
}
inline void GetKernelBasis(float3 D, float3 N, float3& T, float3& B) {
    Geometry::GetBasis(N, T, B);
    if (fabsf(dot(D, N)) < 0.999f) {
        float3 R = reflect(-D, N);
        T = normalize(cross(N, R));
        B = cross(R, T);
    }
}
inline float2 GetTemporalAccumulationParams(const ReblurCB& c, float isInScreenMulFootprintQuality, float accumSpeed) {
    accumSpeed *= REBLUR_SAMPLES_PER_FRAME;
    float w = isInScreenMulFootprintQuality;
    w *= Div(accumSpeed, 1.0f + accumSpeed);
    return float2(w, 1.0f + 3.0f * c.gFramerateScale * w);
}
inline bool CompareMaterials(float m0, float m, float minm) { return max(m0, minm) == max(m, minm); } // Common.hlsli:226-230

// ---- history fetch: Common.hlsli:602-656 + REBLUR_Common.hlsli:305-361 ----------------------------------------------
// The shaders realise Catmull-Rom over the 4x4-minus-corners footprint as 5 bilinear texture fetches (or -- if !useBicubic --
// the 2x2 footprint with custom weights). Each of those fetches is restated here on the texels it actually blends:
//   fetch 0 / 4 : rows j-1 / j+2, columns k, k+1, horizontal fraction tc.x      (their vertical fraction is exactly 0)
//   fetch 1 / 3 : columns k-1 / k+2, rows j, j+1, vertical fraction tc.y        (their horizontal fraction is exactly 0)
//   fetch 2     : the central 2x2 at fractions (tc.x, tc.y)
// i.e. 12 distinct texels with (k, j) = floor(samplePos - 0.5); texel coordinates clamp to the plane like the sampler does.
// The fractions are tc itself (a texture unit would quantise them to 8 bits; we keep fp32).
struct HistoryFilter {
    float4 w;      // weights of fetches 0..3 (bicubic) or the custom bilinear weights of the 2x2 footprint
    float w4;      // weight of fetch 4 (0 without bicubic)
    float sum;
    float2 tc;     // bilinear fractions inside the central 2x2
    int kx, ky;    // (k, j): floor-based origin of the central 2x2 (clamp-addressed fetches)
    int ox, oy;    // origin as the shaders' Load-based bilinear path computes it: int( centerPos ), truncation
    float4 bw;     // custom bilinear weights
    bool useBicubic;
};
inline HistoryFilter MakeHistoryFilter(float2 samplePos, float4 bilinearCustomWeights, bool useBicubic) {
    const float S = NRD_CATROM_SHARPNESS;
    HistoryFilter h;
    float2 origin = floor(samplePos - 0.5f);
    float2 centerPos = origin + 0.5f;
    float2 f = saturate(samplePos - centerPos);
    float2 w0 = f * (f * (-S * f + 2.0f * S) - S);
    float2 w1 = f * (f * ((2.0f - S) * f - (3.0f - S))) + 1.0f;
    float2 w2 = f * (f * (-(2.0f - S) * f + (3.0f - 2.0f * S)) + S);
    float2 w3 = f * (f * (S * f - S));
    float2 w12 = w1 + w2;
    float4 w = float4(w12.x * w0.y, w0.x * w12.y, w12.x * w12.y, w3.x * w12.y);
    float w4 = w12.x * w3.y;
    h.w = useBicubic ? w : bilinearCustomWeights;
    h.w4 = useBicubic ? w4 : 0.0f;
    h.sum = sum(h.w) + h.w4;
    h.tc = Div(w2, w12);
    h.kx = (int)origin.x;
    h.ky = (int)origin.y;
    h.ox = (int)centerPos.x; // int3( centerPos, 0 ): truncation of k + 0.5
    h.oy = (int)centerPos.y;
    h.bw = bilinearCustomWeights;
    h.useBicubic = useBicubic;
    return h;
}
// V = float4 (RGBA planes) or float (single-channel planes: `get` picks .x). The scalar instantiation is NOT the .x of the vector one: with
// -ffp-contract=on `a * b + c` on scalars is one fma, while the overloaded vector operators round the product first (hlsl.h) -- exactly as the
// device's FetchHistoryGeneric<float4> / <float> and its scalar row-load paths do.
template <typename V, typename Get>
inline V FetchHistoryT(const HistoryFilter& h, const Tex& tex, Get get, V zero) {
    auto T = [&](int dx, int dy) { return get(tex.FetchClamped(h.kx + dx, h.ky + dy)); };
    V color;
    if (h.useBicubic) {
        float fx = h.tc.x, fy = h.tc.y, gx = 1.0f - fx, gy = 1.0f - fy;
        V s0 = WSum(T(0, -1), gx, T(1, -1), fx);
        V s1 = WSum(T(-1, 0), gy, T(-1, 1), fy);
        V s2 = WSum(T(0, 0), gx * gy, T(1, 0), fx * gy, T(0, 1), gx * fy, T(1, 1), fx * fy);
        V s3 = WSum(T(2, 0), gy, T(2, 1), fy);
        V s4 = WSum(T(0, 2), gx, T(1, 2), fx);
        color = s0 * h.w.x;
        color = Mad(s1, h.w.y, color);
        color = Mad(s2, h.w.z, color);
        color = Mad(s3, h.w.w, color);
        color = Mad(s4, h.w4, color);
    } else {
        color = T(0, 0) * h.w.x;
        color = Mad(T(1, 0), h.w.y, color);
        color = Mad(T(0, 1), h.w.z, color);
        color = Mad(T(1, 1), h.w.w, color);
    }
    return h.sum < 0.0001f ? zero : Div(color, h.sum);
}
inline float4 FetchHistoryColor(const HistoryFilter& h, const Tex& tex) {
    return FetchHistoryT<float4>(h, tex, [](float4 t) { return t; }, float4(0.0f));
}
inline float FetchHistoryScalar(const HistoryFilter& h, const Tex& tex) {
    return FetchHistoryT<float>(h, tex, [](float4 t) { return t.x; }, 0.0f);
}
template <typename V, typename Get>
inline V FetchHistoryBilinearT(const HistoryFilter& h, const Tex& tex, Get get, V zero) {
    V color = get(tex.Load(h.ox, h.oy)) * h.bw.x;
    color = Mad(get(tex.Load(h.ox + 1, h.oy)), h.bw.y, color);
    color = Mad(get(tex.Load(h.ox, h.oy + 1)), h.bw.z, color);
    color = Mad(get(tex.Load(h.ox + 1, h.oy + 1)), h.bw.w, color);
    float s = sum(h.bw);
    return s < 0.0001f ? zero : Div(color, s);
}
inline float4 FetchHistoryBilinear(const HistoryFilter& h, const Tex& tex) {
    return FetchHistoryBilinearT<float4>(h, tex, [](float4 t) { return t; }, float4(0.0f));
}
inline float FetchHistoryBilinearScalar(const HistoryFilter& h, const Tex& tex) {
    return FetchHistoryBilinearT<float>(h, tex, [](float4 t) { return t.x; }, 0.0f);
}
// the history of a signal in its own type (REBLUR_TYPE)
inline float4 FetchSignalHistory(const HistoryFilter& h, const Tex& tex, float4) { return FetchHistoryColor(h, tex); }
inline float FetchSignalHistory(const HistoryFilter& h, const Tex& tex, float) { return FetchHistoryScalar(h, tex); }
inline DirOcc FetchSignalHistory(const HistoryFilter& h, const Tex& tex, DirOcc) { return DirOcc(FetchHistoryColor(h, tex)); }

} // namespace orc
