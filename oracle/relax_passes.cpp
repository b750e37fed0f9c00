// ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/hlsl.h).
//
// CPU restatement of the RELAX passes, one function per reference shader, templated over the six variants
// (diffuse / specular / both, with and without SH):
//   RELAX_ClassifyTiles           reference Shaders/Source/RELAX_ClassifyTiles.cs.hlsl:19-53
//   RELAX_*_PrePass               reference Shaders/Include/RELAX_PrePass.hlsli:13-346
//   RELAX_*_TemporalAccumulation  reference Shaders/Include/RELAX_TemporalAccumulation.hlsli:10-931
//   RELAX_*_HistoryFix            reference Shaders/Include/RELAX_HistoryFix.hlsli:11-160
//   RELAX_*_HistoryClamping       reference Shaders/Include/RELAX_HistoryClamping.hlsli:10-330
//   RELAX_*_AtrousSmem            reference Shaders/Include/RELAX_AtrousSmem.hlsli:10-455
//   RELAX_*_Atrous                reference Shaders/Include/RELAX_Atrous.hlsli:10-240
//   RELAX_*_SplitScreen           reference Shaders/Include/RELAX_SplitScreen.hlsli:10-52
// Helpers: reference Shaders/Include/RELAX_Common.hlsli. Plane binding order: reference Source/Denoisers/Relax_*.hpp.
// Group-shared tiles of the shaders are read straight from the planes here (with the same coordinate clamping).
// Where the reference reads group-shared memory it never wrote (AtrousSmem in all-sky tiles) this restatement DEFINES the
// value as zero; the HIP kernels do the same.
#include "ml.h"
#include "passes.h"
#include "reblur_common.h" // CompareMaterials, HistoryFilter (shared Common.hlsli pieces)

namespace orc {
namespace {

struct RelaxCB { // reference Shaders/Include/RELAX_Config.hlsli:21-99
    float4x4 gWorldToClip, gWorldToClipPrev, gWorldToViewPrev, gWorldPrevToWorld;
    float4 gRotatorPre, gFrustumRight, gFrustumUp, gFrustumForward, gPrevFrustumRight, gPrevFrustumUp, gPrevFrustumForward, gCameraDelta, gMvScale;
    float2 gJitter, gResolutionScale, gRectOffset, gResourceSizeInv, gResourceSize, gRectSizeInv, gRectSizePrev, gResourceSizeInvPrev;
    uint32_t gPrintfAt[2], gRectOrigin[2];
    int gRectSize[2];
    float gSpecMaxAccumulatedFrameNum, gSpecMaxFastAccumulatedFrameNum, gDiffMaxAccumulatedFrameNum, gDiffMaxFastAccumulatedFrameNum;
    float gDisocclusionThreshold, gDisocclusionThresholdAlternate, gCameraAttachedReflectionMaterialID, gStrandMaterialID, gStrandThickness;
    float gRoughnessFraction, gSpecVarianceBoost, gSplitScreen, gDiffBlurRadius, gSpecBlurRadius, gDepthThreshold, gLobeAngleFraction, gSpecLobeAngleSlack;
    float gHistoryFixEdgeStoppingNormalPower, gRoughnessEdgeStoppingRelaxation, gNormalEdgeStoppingRelaxation, gColorBoxSigmaScale;
    float gHistoryAccelerationAmount, gHistoryResetTemporalSigmaScale, gHistoryResetSpatialSigmaScale, gHistoryResetAmount, gDenoisingRange;
    float gSpecPhiLuminance, gDiffPhiLuminance, gDiffMaxLuminanceRelativeDifference, gSpecMaxLuminanceRelativeDifference, gLuminanceEdgeStoppingRelaxation;
    float gConfidenceDrivenRelaxationMultiplier, gConfidenceDrivenLuminanceEdgeStoppingRelaxation, gConfidenceDrivenNormalEdgeStoppingRelaxation;
    float gDebug, gOrthoMode, gUnproject, gFramerateScale, gCheckerboardResolveAccumSpeed, gJitterDelta, gHistoryFixFrameNum, gHistoryFixBasePixelStride;
    float gHistoryThreshold, gViewZScale, gMinHitDistanceWeight, gDiffMinMaterial, gSpecMinMaterial;
    uint32_t gRoughnessEdgeStoppingEnabled, gFrameIndex, gDiffCheckerboard, gSpecCheckerboard, gHasHistoryConfidence, gHasDisocclusionThresholdMix, gResetHistory;
    // a-trous passes only
    uint32_t gStepSize, gIsLastPass;
};
static_assert(sizeof(RelaxCB) == 712, "RELAX constant block");

constexpr float RELAX_NORMAL_ULP = 1.5f / 255.0f;                 // RELAX_Config.hlsli:15
constexpr float RELAX_MAX_ACCUM_FRAME_NUM = 255.0f;               // RELAX_Config.hlsli:17
constexpr float RELAX_ANTILAG_ACCELERATION_AMOUNT_SCALE = 10.0f;  // RELAX_Config.hlsli:18
constexpr float HALF_PI = 1.57079633f;

static const float3 g_Poisson8[8] = { // reference Shaders/Include/Poisson.hlsli:40-50
    float3(-0.4706069f, -0.4427112f, +0.6461146f), float3(-0.9057375f, +0.3003471f, +0.9542373f), float3(-0.3487388f, +0.4037880f, +0.5335386f),
    float3(+0.1023042f, +0.6439373f, +0.6520134f), float3(+0.5699277f, +0.3513750f, +0.6695386f), float3(+0.2939128f, -0.1131226f, +0.3149309f),
    float3(+0.7836658f, -0.4208784f, +0.8895339f), float3(+0.1564120f, -0.8198990f, +0.8346850f)};

// ---- small vector helpers missing from hlsl.h
inline float3 vmin(float3 a, float3 b) { return float3(min(a.x, b.x), min(a.y, b.y), min(a.z, b.z)); }
inline float3 vmax(float3 a, float3 b) { return float3(max(a.x, b.x), max(a.y, b.y), max(a.z, b.z)); }
inline float3 vsqrt(float3 a) { return float3(HwSqrt(a.x), HwSqrt(a.y), HwSqrt(a.z)); }
inline float4 vmax0(float4 a) { return float4(max(a.x, 0.0f), max(a.y, 0.0f), max(a.z, 0.0f), max(a.w, 0.0f)); }
inline float4 vclamp(float4 a, float lo, float hi) { return float4(clamp(a.x, lo, hi), clamp(a.y, lo, hi), clamp(a.z, lo, hi), clamp(a.w, lo, hi)); }
inline float Cmp(bool b) { return b ? 1.0f : 0.0f; }

// [ml] Color::RgbToYCoCg / YCoCgToRgb (MathLib; unclamped inverse)
inline float3 RgbToYCoCg(float3 c) { return float3(c.x * 0.25f + c.y * 0.5f + c.z * 0.25f, c.x * 0.5f + c.y * 0.0f + c.z * -0.5f, c.x * -0.25f + c.y * 0.5f + c.z * -0.25f); }
inline float3 YCoCgToRgb(float3 c) {
    float t = c.x - c.z;
    return float3(t + c.y, c.x + c.z, t - c.y);
}

// ---- RELAX_Common.hlsli
inline float UnpackViewZ(const RelaxCB& c, float z) { return fabsf(z * c.gViewZScale); } // Common.hlsli:233
inline float4 UnpackPrevNormalRoughness(float4 p) {                                        // RELAX_Common.hlsli:10-17
    return float4(_NRD_SafeNormalize(float3(p.x * 2.0f - 1.0f, p.y * 2.0f - 1.0f, p.z * 2.0f - 1.0f)), p.w);
}
inline float4 PackPrevNormalRoughness(float4 nr) { return float4(nr.x * 0.5f + 0.5f, nr.y * 0.5f + 0.5f, nr.z * 0.5f + 0.5f, nr.w); } // :19-26
inline float BilinearWithCustomWeightsImmediateFloat(float s00, float s10, float s01, float s11, float4 w) { // :28-39
    float o = s00 * w.x;
    o += s10 * w.y;
    o += s01 * w.z;
    o += s11 * w.w;
    float sumWeights = sum(w);
    return sumWeights < 0.0001f ? 0.0f : o * rcp(sumWeights);
}
inline float4 BilinearWithCustomWeightsFloat4(const Tex& tex, int ox, int oy, float4 w) { // :55-66
    float4 o = tex.Load(ox, oy) * w.x;
    o = Mad(tex.Load(ox + 1, oy), w.y, o);
    o = Mad(tex.Load(ox, oy + 1), w.z, o);
    o = Mad(tex.Load(ox + 1, oy + 1), w.w, o);
    float sumWeights = sum(w);
    return sumWeights < 0.0001f ? float4(0.0f) : o * rcp(sumWeights);
}
inline float3 WorldPosFromClip(const RelaxCB& c, float4 R, float4 U, float4 F, float2 clip, float viewZ) { // :68-102
    float3 dir = F.xyz() + R.xyz() * clip.x - U.xyz() * clip.y;
    float3 ortho = viewZ * F.xyz() + R.xyz() * clip.x - U.xyz() * clip.y;
    return c.gOrthoMode == 0.0f ? viewZ * dir : ortho;
}
inline float3 GetCurrentWorldPosFromClipSpaceXY(const RelaxCB& c, float2 clip, float viewZ) { return WorldPosFromClip(c, c.gFrustumRight, c.gFrustumUp, c.gFrustumForward, clip, viewZ); }
inline float3 GetCurrentWorldPosFromPixelPos(const RelaxCB& c, int px, int py, float viewZ) {
    float2 clip = float2(float(px) + 0.5f, float(py) + 0.5f) * c.gRectSizeInv * 2.0f - 1.0f;
    return GetCurrentWorldPosFromClipSpaceXY(c, clip, viewZ);
}
inline float3 GetPreviousWorldPosFromClipSpaceXY(const RelaxCB& c, float2 clip, float viewZ) {
    return WorldPosFromClip(c, c.gPrevFrustumRight, c.gPrevFrustumUp, c.gPrevFrustumForward, clip, viewZ);
}
inline float3 GetPreviousWorldPosFromPixelPos(const RelaxCB& c, int px, int py, float viewZ) {
    float2 clip = float2(float(px) + 0.5f, float(py) + 0.5f) * float2(Rcp(c.gRectSizePrev.x), Rcp(c.gRectSizePrev.y)) * 2.0f - 1.0f;
    return GetPreviousWorldPosFromClipSpaceXY(c, clip, viewZ);
}
inline float GetPlaneDistanceWeight(float3 centerWorldPos, float3 centerNormal, float centerViewZ, float3 sampleWorldPos, float threshold) { // :104-109
    float d = fabsf(dot(sampleWorldPos - centerWorldPos, centerNormal));
    return Div(d, centerViewZ) > threshold ? 0.0f : 1.0f;
}
inline float GetPlaneDistanceWeight_Atrous(float3 centerWorldPos, float3 centerNormal, float3 sampleWorldPos, float threshold) { // :111-116
    float d = fabsf(dot(sampleWorldPos - centerWorldPos, centerNormal));
    return d < threshold ? 1.0f : 0.0f;
}
inline float GetSpecLobeTanHalfAngle(float roughness, float percentOfVolume = 0.75f) { // :118-126 (the "old" lobe formula)
    roughness = saturate(roughness);
    percentOfVolume = saturate(percentOfVolume);
    return Div(roughness * roughness * percentOfVolume, 1.0f - percentOfVolume + NRD_EPS);
}
inline float2 GetNormalWeightParams_ATrous(float roughness, float numFramesInHistory, float specularReprojectionConfidence, float normalEdgeStoppingRelaxation,
    float specularLobeAngleFraction, float specularLobeAngleSlack) { // :128-148
    float relaxation = saturate(DivConst(numFramesInHistory, 5.0f));
    relaxation *= lerp(1.0f, specularReprojectionConfidence, normalEdgeStoppingRelaxation);
    float f = 0.9f + 0.1f * relaxation;
    float angle = atan(GetSpecLobeTanHalfAngle(roughness, specularLobeAngleFraction));
    angle *= 10.0f - 9.0f * relaxation;
    angle += specularLobeAngleSlack;
    angle = min(HALF_PI, angle);
    return float2(angle, f);
}
inline float GetSpecularNormalWeight_ATrous(float2 params0, float3 n0, float3 n, float3 v0, float3 v) { // :150-159
    float cosaN = dot(n0, n);
    float cosaV = dot(v0, v);
    float cosa = min(cosaN, cosaV);
    float a = Math::AcosApprox(cosa);
    a = Math::SmoothStep(0.0f, params0.x, a);
    return saturate(1.0f - a * params0.y);
}
inline float GetNormalWeightParam2(float roughness, float angleFraction) { // :162-168
    float angle = atan(GetSpecLobeTanHalfAngle(roughness, angleFraction));
    return Rcp(max(angle, RELAX_NORMAL_ULP));
}
inline float GetBilateralWeight(float z, float zc) { return Math::LinearStep(0.03f, 0.0f, fabsf(z - zc) * rcp(max(z, zc))); } // :171-172

// Common.hlsli:578-589 including the RELAX-only remap
inline float GetEncodingAwareNormalWeightR(float3 Ncurr, float3 Nprev, float maxAngle, float curvatureAngle, float thresholdAngle, bool remap) {
    float w = GetEncodingAwareNormalWeight(Ncurr, Nprev, maxAngle, curvatureAngle, thresholdAngle);
    if (remap)
        w = Math::SmoothStep(0.05f, 0.95f, w);
    return w;
}
// Geometry::GetScreenUv( M, X, false ): no back-projection override
inline float2 ScreenUvNoKill(const float4x4& worldToClip, float3 X) {
    float4 clip = Geometry::ProjectiveTransform(worldToClip, X);
    return float2((Div(clip.x, clip.w)) * 0.5f + 0.5f, (Div(clip.y, clip.w)) * -0.5f + 0.5f);
}
// Common.hlsli:297-307
inline float2 ApplyCheckerboardShift(float2 pos, uint32_t mode, uint32_t counter, uint32_t frameIndex) {
    float2 posPositive = pos + 16384.0f;
    uint32_t checkerboard = Sequence::CheckerBoard((uint32_t)posPositive.x, (uint32_t)posPositive.y, frameIndex);
    float shift = ((counter & 1u) == 0u) ? -1.0f : 1.0f;
    pos.x += shift * Cmp(checkerboard != mode && mode != 2u);
    return pos;
}
inline float2 ClampUvToViewport(const RelaxCB& c, float2 uv) { return min(uv * c.gResolutionScale, c.gResolutionScale - 0.5f * c.gResourceSizeInv); }
inline float4 Denanify(float w, float4 x) { return w == 0.0f ? float4(0.0f) : x; }

// walks DispatchDesc::resources in binding order
struct Cursor {
    Tex* t;
    uint32_t i = 0;
    Tex* next() { return &t[i++]; }
};
// one radiance signal (specular or diffuse) of a pass
struct Sig {
    Tex *in = nullptr, *inSh = nullptr, *prev = nullptr, *prevSh = nullptr, *fast = nullptr, *fastSh = nullptr, *noisy = nullptr, *confidence = nullptr;
    Tex *out = nullptr, *outSh = nullptr, *outFast = nullptr, *outFastSh = nullptr;
};

// ================================================================================================ ClassifyTiles
void ClassifyTiles(const PassIO& io) {
    const RelaxCB& c = *(const RelaxCB*)io.constants;
    const Tex& gIn_ViewZ = io.t[0];
    Tex& gOut_Tiles = io.t[1];
    const int tilesW = (c.gRectSize[0] + 15) / 16, tilesH = (c.gRectSize[1] + 15) / 16;
    for (int ty = 0; ty < tilesH; ty++)
        for (int tx = 0; tx < tilesW; tx++) {
            uint32_t sky = 0;
            for (int j = 0; j < 16; j++)
                for (int i = 0; i < 16; i++)
                    sky += fabsf(gIn_ViewZ.Load(tx * 16 + i, ty * 16 + j).x) > c.gDenoisingRange ? 1 : 0;
            gOut_Tiles.Store(tx, ty, sky == 256 ? 1.0f : 0.0f);
        }
}

// ================================================================================================ HitDistReconstruction
// reference Shaders/Include/RELAX_HitDistReconstruction.hlsli:10-160. BORDER = 1 -> 3x3, 2 -> 5x5. Hit distances are (spec, diff);
// the window is read at rect-clamped coordinates like the LDS preload. NOTE (kept): the roughness weight is fed the CENTER
// roughness, so it evaluates to exactly 1.
template <bool DIFF, bool SPEC, int BORDER>
void HitDistReconstruction(const PassIO& io) {
    const RelaxCB& c = *(const RelaxCB*)io.constants;
    Cursor cur{io.t};
    Sig spec, diff;
    const Tex& gIn_Tiles = *cur.next();
    if (SPEC) spec.in = cur.next();
    if (DIFF) diff.in = cur.next();
    const Tex& gIn_Normal_Roughness = *cur.next();
    const Tex& gIn_ViewZ = *cur.next();
    if (SPEC) spec.out = cur.next();
    if (DIFF) diff.out = cur.next();
    const int rectW = c.gRectSize[0], rectH = c.gRectSize[1];
    const int ox = (int)c.gRectOrigin[0], oy = (int)c.gRectOrigin[1];

#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < rectH; py++)
        for (int px = 0; px < rectW; px++) {
            if (gIn_Tiles.Load(px >> 4, py >> 4).x != 0.0f)
                continue;
            auto Cx = [&](int x) { return clamp(x, 0, rectW - 1); };
            auto Cy = [&](int y) { return clamp(y, 0, rectH - 1); };
            auto ViewZ = [&](int x, int y) { return UnpackViewZ(c, gIn_ViewZ.Load(ox + Cx(x), oy + Cy(y)).x); };
            auto HitDist = [&](int x, int y) { return float2(SPEC ? spec.in->Load(Cx(x), Cy(y)).w : c.gDenoisingRange, DIFF ? diff.in->Load(Cx(x), Cy(y)).w : c.gDenoisingRange); };
            const float centerViewZ = ViewZ(px, py);
            if (centerViewZ > c.gDenoisingRange)
                continue;

            float2 pixelUv = float2(float(px) + 0.5f, float(py) + 0.5f) * c.gRectSizeInv;
            float4 normalAndRoughness = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.Load(ox + px, oy + py));
            float3 centerNormal = normalAndRoughness.xyz();
            float centerRoughness = normalAndRoughness.w;
            float2 centerHitDist = HitDist(px, py);

            float2 relaxedRoughnessWeightParams = GetRelaxedRoughnessWeightParams(centerRoughness * centerRoughness);
            float specularNormalWeightParam = GetNormalWeightParam(1.0f, 1.0f, centerRoughness);
            float diffuseNormalWeightParam = GetNormalWeightParam(1.0f, 1.0f);

            float sumSpecularWeight = 1000.0f * Cmp(centerHitDist.x != 0.0f);
            float sumSpecularHitDist = centerHitDist.x * sumSpecularWeight;
            float sumDiffuseWeight = 1000.0f * Cmp(centerHitDist.y != 0.0f);
            float sumDiffuseHitDist = centerHitDist.y * sumDiffuseWeight;

            for (int dy = 0; dy <= BORDER * 2; dy++)
                for (int dx = 0; dx <= BORDER * 2; dx++) {
                    int ix = dx - BORDER, iy = dy - BORDER;
                    if (ix == 0 && iy == 0)
                        continue;
                    float2 o = float2(float(ix), float(iy));
                    float3 sampleNormal = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.Load(ox + Cx(px + ix), oy + Cy(py + iy))).xyz();
                    float2 sampleHitDist = HitDist(px + ix, py + iy);
                    float sampleViewZ = ViewZ(px + ix, py + iy);
                    float cosa = dot(centerNormal, sampleNormal);
                    float angle = Math::AcosApprox(cosa);

                    float w = IsInScreenNearest(pixelUv + o * c.gRectSizeInv);
                    w *= Cmp(sampleViewZ < c.gDenoisingRange);
                    w *= GetGaussianWeight(length(o) * 0.5f);
                    w *= GetBilateralWeight(sampleViewZ, centerViewZ);

                    if (SPEC) {
                        float specularWeight = w;
                        specularWeight *= ComputeExponentialWeight(angle, specularNormalWeightParam, 0.0f);
                        specularWeight *= ComputeExponentialWeight(normalAndRoughness.w * normalAndRoughness.w, relaxedRoughnessWeightParams.x, relaxedRoughnessWeightParams.y);
                        float sampleSpecularHitDist = specularWeight == 0.0f ? 0.0f : sampleHitDist.x; // Denanify
                        specularWeight *= Cmp(sampleSpecularHitDist != 0.0f);
                        sumSpecularHitDist += sampleSpecularHitDist * specularWeight;
                        sumSpecularWeight += specularWeight;
                    }
                    if (DIFF) {
                        float diffuseWeight = w;
                        diffuseWeight *= ComputeExponentialWeight(angle, diffuseNormalWeightParam, 0.0f);
                        float sampleDiffuseHitDist = diffuseWeight == 0.0f ? 0.0f : sampleHitDist.y; // Denanify
                        diffuseWeight *= Cmp(sampleDiffuseHitDist != 0.0f);
                        sumDiffuseHitDist += diffuseWeight == 0.0f ? 0.0f : sampleDiffuseHitDist * diffuseWeight;
                        sumDiffuseWeight += diffuseWeight;
                    }
                }

            if (SPEC) {
                sumSpecularHitDist = Div(sumSpecularHitDist, max(sumSpecularWeight, 1e-6f));
                spec.out->Store(px, py, float4(spec.in->Load(px, py).xyz(), sumSpecularHitDist));
            }
            if (DIFF) {
                sumDiffuseHitDist = Div(sumDiffuseHitDist, max(sumDiffuseWeight, 1e-6f));
                diff.out->Store(px, py, float4(diff.in->Load(px, py).xyz(), sumDiffuseHitDist));
            }
        }
}

// ================================================================================================ PrePass
template <bool DIFF, bool SPEC, bool SH>
void PrePass(const PassIO& io) {
    const RelaxCB& c = *(const RelaxCB*)io.constants;
    Cursor cur{io.t};
    Sig spec, diff;
    const Tex& gIn_Tiles = *cur.next();
    if (SPEC) spec.in = cur.next();
    if (DIFF) diff.in = cur.next();
    const Tex& gIn_Normal_Roughness = *cur.next();
    const Tex& gIn_ViewZ = *cur.next();
    if (SH && SPEC) spec.inSh = cur.next();
    if (SH && DIFF) diff.inSh = cur.next();
    if (SPEC) spec.out = cur.next();
    if (DIFF) diff.out = cur.next();
    if (SH && SPEC) spec.outSh = cur.next();
    if (SH && DIFF) diff.outSh = cur.next();

    const int rectW = c.gRectSize[0], rectH = c.gRectSize[1];
    const int ox = (int)c.gRectOrigin[0], oy = (int)c.gRectOrigin[1];
    const float2 rectSize = float2(float(rectW), float(rectH));

#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < rectH; py++)
        for (int px = 0; px < rectW; px++) {
            if (gIn_Tiles.Load(px >> 4, py >> 4).x != 0.0f)
                continue;
            float centerViewZ = UnpackViewZ(c, gIn_ViewZ.Load(ox + px, oy + py).x);
            if (centerViewZ > c.gDenoisingRange)
                continue;

            // Checkerboard resolve weights
            uint32_t checkerboard = Sequence::CheckerBoard((uint32_t)px, (uint32_t)py, c.gFrameIndex);
            int cbx0 = max(px - 1, 0), cbx1 = min(px + 1, rectW - 1);
            float materialID0 = 0.0f, materialID1 = 0.0f;
            float2 checkerboardResolveWeights = float2(1.0f);
            bool anyCheckerboard = (SPEC && c.gSpecCheckerboard != 2u) || (DIFF && c.gDiffCheckerboard != 2u);
            if (anyCheckerboard) {
                float viewZ0 = UnpackViewZ(c, gIn_ViewZ.Load(ox + cbx0, oy + py).x);
                float viewZ1 = UnpackViewZ(c, gIn_ViewZ.Load(ox + cbx1, oy + py).x);
                NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.Load(ox + cbx0, oy + py), materialID0);
                NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.Load(ox + cbx1, oy + py), materialID1);
                checkerboardResolveWeights = float2(GetBilateralWeight(viewZ0, centerViewZ), GetBilateralWeight(viewZ1, centerViewZ));
                checkerboardResolveWeights.x = (viewZ0 > c.gDenoisingRange || px < 1) ? 0.0f : checkerboardResolveWeights.x;
                checkerboardResolveWeights.y = (viewZ1 > c.gDenoisingRange || px > rectW - 2) ? 0.0f : checkerboardResolveWeights.y;
            }
            cbx0 >>= 1;
            cbx1 >>= 1;

            float centerMaterialID;
            float4 centerNormalRoughness = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.Load(ox + px, oy + py), centerMaterialID);
            float3 centerNormal = centerNormalRoughness.xyz();
            float centerRoughness = centerNormalRoughness.w;
            float3 centerWorldPos = GetCurrentWorldPosFromPixelPos(c, px, py, centerViewZ);
            float4 rotator = c.gRotatorPre; // GetBlurKernelRotation( NRD_FRAME, ... )
            float2 pixelUv = float2(float(px) + 0.5f, float(py) + 0.5f) * c.gRectSizeInv;

            // shared per-tap geometry fetch
            auto TapUv = [&](int i, float blurRadius, uint32_t checkerboardMode, float2& uv, float2& uvScaled, float2& checkerboardUvScaled) {
                float3 offset = g_Poisson8[i];
                uv = pixelUv * rectSize + Geometry::RotateVector(rotator, offset.xy()) * blurRadius;
                uv = floor(uv) + 0.5f;
                uv = ApplyCheckerboardShift(uv, checkerboardMode, (uint32_t)i, c.gFrameIndex) * c.gRectSizeInv;
                uvScaled = ClampUvToViewport(c, uv);
                checkerboardUvScaled = float2(uvScaled.x * (checkerboardMode != 2u ? 0.5f : 1.0f), uvScaled.y);
            };

            if (DIFF) {
                bool diffHasData = true;
                int dpx = px;
                if (c.gDiffCheckerboard != 2u) {
                    diffHasData = checkerboard == c.gDiffCheckerboard;
                    dpx >>= 1;
                }
                float4 diffuseIllumination = diff.in->Load(dpx, py);
                float4 diffuseSH = SH ? diff.inSh->Load(dpx, py) : float4(0.0f);
                if (!diffHasData) {
                    float2 wc = checkerboardResolveWeights;
                    wc.x *= Cmp(CompareMaterials(centerMaterialID, materialID0, c.gDiffMinMaterial));
                    wc.y *= Cmp(CompareMaterials(centerMaterialID, materialID1, c.gDiffMinMaterial));
                    wc *= Math::PositiveRcp(wc.x + wc.y);
                    float4 d0 = Denanify(wc.x, diff.in->Load(cbx0, py));
                    float4 d1 = Denanify(wc.y, diff.in->Load(cbx1, py));
                    diffuseIllumination = d0 * wc.x + d1 * wc.y;
                    if (SH) {
                        float4 d0SH = Denanify(wc.x, diff.inSh->Load(cbx0, py));
                        float4 d1SH = Denanify(wc.y, diff.inSh->Load(cbx1, py));
                        diffuseSH = d0SH * wc.x + d1SH * wc.y;
                    }
                }

                if (c.gDiffBlurRadius > 0.0f) {
                    float frustumSize = PixelRadiusToWorld(c.gUnproject, c.gOrthoMode, float(min(rectW, rectH)), centerViewZ);
                    float hitDist = diffuseIllumination.w == 0.0f ? 1.0f : diffuseIllumination.w;
                    float hitDistFactor = GetHitDistFactor(hitDist, frustumSize);
                    float blurRadius = c.gDiffBlurRadius * hitDistFactor;
                    if (diffuseIllumination.w == 0.0f)
                        blurRadius = max(blurRadius, 1.0f);

                    float normalWeightParam = GetNormalWeightParam2(1.0f, 0.25f * c.gLobeAngleFraction);
                    float2 hitDistanceWeightParams = GetHitDistanceWeightParams(diffuseIllumination.w, 1.0f / 9.0f);
                    float weightSum = 1.0f;
                    float diffMinHitDistanceWeight = c.gMinHitDistanceWeight;

                    for (int i = 0; i < 8; i++) {
                        float2 uv, uvScaled, cbUv;
                        TapUv(i, blurRadius, c.gDiffCheckerboard, uv, uvScaled, cbUv);

                        float sampleMaterialID;
                        float3 sampleNormal = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.SampleNearest(uvScaled + c.gRectOffset), sampleMaterialID).xyz();
                        float sampleViewZ = UnpackViewZ(c, gIn_ViewZ.SampleNearest(uvScaled + c.gRectOffset).x);
                        float3 sampleWorldPos = GetCurrentWorldPosFromClipSpaceXY(c, uv * 2.0f - 1.0f, sampleViewZ);

                        float sampleWeight = IsInScreenNearest(uv);
                        sampleWeight *= Cmp(sampleViewZ < c.gDenoisingRange);
                        sampleWeight *= Cmp(CompareMaterials(centerMaterialID, sampleMaterialID, c.gDiffMinMaterial));
                        sampleWeight *= GetPlaneDistanceWeight(centerWorldPos, centerNormal, c.gOrthoMode == 0.0f ? centerViewZ : 1.0f, sampleWorldPos, c.gDepthThreshold);
                        float angle = Math::AcosApprox(dot(centerNormal, sampleNormal));
                        sampleWeight *= ComputeWeight(angle, normalWeightParam, 0.0f);

                        float4 sampleDiffuseIllumination = Denanify(sampleWeight, diff.in->SampleNearest(cbUv));
                        sampleWeight *= lerp(diffMinHitDistanceWeight, 1.0f, ComputeExponentialWeight(sampleDiffuseIllumination.w, hitDistanceWeightParams.x, hitDistanceWeightParams.y));
                        sampleWeight *= GetGaussianWeight(g_Poisson8[i].z);

                        weightSum += sampleWeight;
                        diffuseIllumination = Mad(sampleDiffuseIllumination, sampleWeight, diffuseIllumination);
                        if (SH) {
                            float4 sampleDiffuseSH = Denanify(sampleWeight, diff.inSh->SampleNearest(cbUv));
                            diffuseSH = Mad(sampleDiffuseSH, sampleWeight, diffuseSH);
                        }
                    }
                    diffuseIllumination = Div(diffuseIllumination, weightSum);
                    if (SH)
                        diffuseSH = Div(diffuseSH, weightSum);
                }
                diff.out->Store(px, py, vclamp(diffuseIllumination, 0.0f, NRD_FP16_MAX));
                if (SH)
                    diff.outSh->Store(px, py, vclamp(diffuseSH, -NRD_FP16_MAX, NRD_FP16_MAX));
            }

            if (SPEC) {
                bool specHasData = true;
                int spx = px;
                if (c.gSpecCheckerboard != 2u) {
                    specHasData = checkerboard == c.gSpecCheckerboard;
                    spx >>= 1;
                }
                float4 specularIllumination = spec.in->Load(spx, py);
                float4 specularSH = SH ? spec.inSh->Load(spx, py) : float4(0.0f);
                if (!specHasData) {
                    float2 wc = checkerboardResolveWeights;
                    wc.x *= Cmp(CompareMaterials(centerMaterialID, materialID0, c.gSpecMinMaterial));
                    wc.y *= Cmp(CompareMaterials(centerMaterialID, materialID1, c.gSpecMinMaterial));
                    wc *= Math::PositiveRcp(wc.x + wc.y);
                    float4 s0 = Denanify(wc.x, spec.in->Load(cbx0, py));
                    float4 s1 = Denanify(wc.y, spec.in->Load(cbx1, py));
                    specularIllumination = s0 * wc.x + s1 * wc.y;
                    if (SH) {
                        float4 s0SH = Denanify(wc.x, spec.inSh->Load(cbx0, py));
                        float4 s1SH = Denanify(wc.y, spec.inSh->Load(cbx1, py));
                        specularSH = s0SH * wc.x + s1SH * wc.y;
                    }
                }
                specularIllumination.w = max(0.0f, min(c.gDenoisingRange, specularIllumination.w));

                if (c.gSpecBlurRadius > 0.0f) {
                    float3 viewVector = c.gOrthoMode == 0.0f ? normalize(-centerWorldPos) : c.gFrustumForward.xyz();
                    float4 D = ImportanceSampling::GetSpecularDominantDirection(centerNormal, viewVector, centerRoughness);
                    float NoD = fabsf(dot(centerNormal, D.xyz()));

                    float frustumSize = PixelRadiusToWorld(c.gUnproject, c.gOrthoMode, float(min(rectW, rectH)), centerViewZ);
                    float hitDist = specularIllumination.w == 0.0f ? 1.0f : specularIllumination.w;
                    float hitDistFactor = GetHitDistFactor(hitDist * NoD, frustumSize);

                    float smc = GetSpecMagicCurve(centerRoughness);
                    float blurRadius = c.gSpecBlurRadius * hitDistFactor * smc;
                    float lobeTanHalfAngle = ImportanceSampling::GetSpecularLobeTanHalfAngle(centerRoughness, 0.75f);
                    float lobeRadius = hitDist * NoD * lobeTanHalfAngle;
                    float minBlurRadius = Div(lobeRadius, PixelRadiusToWorld(c.gUnproject, c.gOrthoMode, 1.0f, centerViewZ + hitDist * D.w));
                    blurRadius = min(blurRadius, minBlurRadius);
                    if (specularIllumination.w == 0.0f)
                        blurRadius = max(blurRadius, 1.0f);

                    float normalWeightParam = GetNormalWeightParam2(centerRoughness, 0.5f * c.gLobeAngleFraction);
                    float2 hitDistanceWeightParams = GetHitDistanceWeightParams(specularIllumination.w, 1.0f / 9.0f, centerRoughness);
                    float2 roughnessWeightParams = GetRoughnessWeightParams(centerRoughness, c.gRoughnessFraction);

                    float specMinHitDistanceWeight = specularIllumination.w == 0.0f ? 1.0f : c.gMinHitDistanceWeight * smc;
                    float specularHitT = specularIllumination.w == 0.0f ? c.gDenoisingRange : specularIllumination.w;
                    float minHitT = specularHitT == 0.0f ? NRD_INF : specularHitT;
                    float weightSum = 1.0f;
                    float3 rgb = specularIllumination.xyz();

                    for (int i = 0; i < 8; i++) {
                        float2 uv, uvScaled, cbUv;
                        TapUv(i, blurRadius, c.gSpecCheckerboard, uv, uvScaled, cbUv);

                        float sampleMaterialID;
                        float4 sampleNormalRoughness = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.SampleNearest(uvScaled + c.gRectOffset), sampleMaterialID);
                        float3 sampleNormal = sampleNormalRoughness.xyz();
                        float sampleRoughness = sampleNormalRoughness.w;
                        float sampleViewZ = UnpackViewZ(c, gIn_ViewZ.SampleNearest(uvScaled + c.gRectOffset).x);

                        float sampleWeight = IsInScreenNearest(uv);
                        sampleWeight *= Cmp(sampleViewZ < c.gDenoisingRange);
                        sampleWeight *= Cmp(CompareMaterials(centerMaterialID, sampleMaterialID, c.gSpecMinMaterial));
                        sampleWeight *= ComputeWeight(sampleRoughness, roughnessWeightParams.x, roughnessWeightParams.y);
                        float angle = Math::AcosApprox(dot(centerNormal, sampleNormal));
                        sampleWeight *= ComputeWeight(angle, normalWeightParam, 0.0f);

                        float3 sampleWorldPos = GetCurrentWorldPosFromClipSpaceXY(c, uv * 2.0f - 1.0f, sampleViewZ);
                        sampleWeight *= GetPlaneDistanceWeight(centerWorldPos, centerNormal, c.gOrthoMode == 0.0f ? centerViewZ : 1.0f, sampleWorldPos, c.gDepthThreshold);

                        float4 sampleSpecularIllumination = Denanify(sampleWeight, spec.in->SampleNearest(cbUv));
                        sampleWeight *= lerp(specMinHitDistanceWeight, 1.0f, ComputeExponentialWeight(sampleSpecularIllumination.w, hitDistanceWeightParams.x, hitDistanceWeightParams.y));
                        sampleWeight *= GetGaussianWeight(g_Poisson8[i].z);

                        // samples close to the reflection contact should not be pre-blurred
                        float d = length(sampleWorldPos - centerWorldPos);
                        float h = sampleSpecularIllumination.w;
                        float t = Div(h, specularIllumination.w + d);
                        sampleWeight *= lerp(saturate(t), 1.0f, Math::LinearStep(0.5f, 1.0f, centerRoughness));

                        weightSum += sampleWeight;
                        rgb = Mad(sampleSpecularIllumination.xyz(), sampleWeight, rgb);
                        if (SH) {
                            float4 sampleSpecularSH = Denanify(sampleWeight, spec.inSh->SampleNearest(cbUv));
                            specularSH = Mad(sampleSpecularSH, sampleWeight, specularSH);
                        }
                        if (sampleWeight != 0.0f)
                            minHitT = min(minHitT, sampleSpecularIllumination.w == 0.0f ? NRD_INF : sampleSpecularIllumination.w);
                    }
                    rgb = Div(rgb, weightSum);
                    specularIllumination = float4(rgb, minHitT == NRD_INF ? 0.0f : minHitT);
                    if (SH)
                        specularSH = Div(specularSH, weightSum);
                }
                spec.out->Store(px, py, vclamp(specularIllumination, 0.0f, NRD_FP16_MAX));
                if (SH)
                    spec.outSh->Store(px, py, vclamp(specularSH, -NRD_FP16_MAX, NRD_FP16_MAX));
            }
        }
}

// ================================================================================================ TemporalAccumulation
template <bool DIFF, bool SPEC, bool SH>
void TemporalAccumulation(const PassIO& io) {
    const RelaxCB& c = *(const RelaxCB*)io.constants;
    Cursor cur{io.t};
    Sig spec, diff;
    const Tex& gIn_Tiles = *cur.next();
    if (SPEC) spec.in = cur.next();
    if (DIFF) diff.in = cur.next();
    const Tex& gIn_Mv = *cur.next();
    const Tex& gIn_Normal_Roughness = *cur.next();
    const Tex& gIn_ViewZ = *cur.next();
    if (SPEC) spec.fast = cur.next();
    if (DIFF) diff.fast = cur.next();
    if (SPEC) spec.prev = cur.next();
    if (DIFF) diff.prev = cur.next();
    const Tex& gPrev_Normal_Roughness = *cur.next();
    const Tex& gPrev_ViewZ = *cur.next();
    const Tex* gPrev_SpecHitDist = SPEC ? cur.next() : nullptr;
    const Tex& gPrev_HistoryLength = *cur.next();
    const Tex& gPrev_MaterialID = *cur.next();
    if (SPEC) spec.confidence = cur.next();
    if (DIFF) diff.confidence = cur.next();
    const Tex& gIn_DisocclusionThresholdMix = *cur.next();
    if (SH && SPEC) spec.inSh = cur.next();
    if (SH && DIFF) diff.inSh = cur.next();
    if (SH && SPEC) spec.fastSh = cur.next();
    if (SH && DIFF) diff.fastSh = cur.next();
    if (SH && SPEC) spec.prevSh = cur.next();
    if (SH && DIFF) diff.prevSh = cur.next();
    if (SPEC) spec.out = cur.next();
    if (DIFF) diff.out = cur.next();
    if (SPEC) spec.outFast = cur.next();
    if (DIFF) diff.outFast = cur.next();
    Tex* gOut_SpecHitDist = SPEC ? cur.next() : nullptr;
    Tex& gOut_HistoryLength = *cur.next();
    Tex* gOut_SpecReprojectionConfidence = SPEC ? cur.next() : nullptr;
    if (SH && SPEC) spec.outSh = cur.next();
    if (SH && DIFF) diff.outSh = cur.next();
    if (SH && SPEC) spec.outFastSh = cur.next();
    if (SH && DIFF) diff.outFastSh = cur.next();

    const int rectW = c.gRectSize[0], rectH = c.gRectSize[1];
    const int ox = (int)c.gRectOrigin[0], oy = (int)c.gRectOrigin[1];
    const float2 rectSize = float2(float(rectW), float(rectH));
    const float2 resolutionScalePrev = c.gRectSizePrev * c.gResourceSizeInvPrev;
    auto PrevSize = [](const Tex& t) { return float2(float(t.W()), float(t.H())); };

#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < rectH; py++)
        for (int px = 0; px < rectW; px++) {
            if (gIn_Tiles.Load(px >> 4, py >> 4).x != 0.0f)
                continue;
            float currentLinearZ = UnpackViewZ(c, gIn_ViewZ.Load(ox + px, oy + py).x);
            if (currentLinearZ > c.gDenoisingRange)
                continue;

            // the group-shared tile: normal (xyz) + specular hitT (w) at rect-clamped coordinates
            auto Shared = [&](int x, int y) {
                x = clamp(x, 0, rectW - 1);
                y = clamp(y, 0, rectH - 1);
                float4 v = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.Load(ox + x, oy + y));
                if (SPEC)
                    v.w = spec.in->Load(x, y).w;
                return v;
            };

            float currentMaterialID;
            float4 currentNormalRoughness = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.Load(ox + px, oy + py), currentMaterialID);
            float3 currentNormal = currentNormalRoughness.xyz();
            float currentRoughness = currentNormalRoughness.w;

            float3 currentWorldPos = GetCurrentWorldPosFromPixelPos(c, px, py, currentLinearZ);
            float3 currentViewVector = c.gOrthoMode == 0.0f ? currentWorldPos : currentLinearZ * normalize(c.gFrustumForward.xyz());
            float3 V = -normalize(currentViewVector);
            float NoV = fabsf(dot(currentNormal, V));

            // previous position
            float2 pixelUv = float2(float(px) + 0.5f, float(py) + 0.5f) * c.gRectSizeInv;
            float4 mvRaw = gIn_Mv.Load(ox + px, oy + py);
            float3 mv = mvRaw.xyz() * c.gMvScale.xyz();
            float3 prevWorldPos = currentWorldPos;
            float2 prevUVSMB = pixelUv + mv.xy();
            if (c.gMvScale.w == 0.0f) {
                if (c.gMvScale.z == 0.0f)
                    mv.z = Geometry::AffineTransform(c.gWorldToViewPrev, currentWorldPos).z - currentLinearZ;
                prevWorldPos = GetPreviousWorldPosFromClipSpaceXY(c, prevUVSMB * 2.0f - 1.0f, currentLinearZ + mv.z) + c.gCameraDelta.xyz();
            } else {
                prevWorldPos += mv;
                prevUVSMB = Geometry::GetScreenUv(c.gWorldToClipPrev, prevWorldPos);
            }

            // noisy inputs
            float3 diffuseIllumination = DIFF ? diff.in->Load(px, py).xyz() : float3(0.0f);
            float4 diffuseSH = (DIFF && SH) ? diff.inSh->Load(px, py) : float4(0.0f);
            float4 specularIllumination = SPEC ? spec.in->Load(px, py) : float4(0.0f);
            float4 specularSH = (SPEC && SH) ? spec.inSh->Load(px, py) : float4(0.0f);

            // average normal and min hit distance in 3x3
            float hitTM1 = Shared(px, py).w;
            float minHitDist3x3 = hitTM1 == 0.0f ? NRD_INF : hitTM1;
            float3 currentNormalAveraged = currentNormal;
            for (int i = -1; i <= 1; i++)
                for (int j = -1; j <= 1; j++) {
                    if (i == 0 && j == 0)
                        continue;
                    float4 normalSpecHitT = Shared(px + i, py + j);
                    minHitDist3x3 = min(minHitDist3x3, normalSpecHitT.w == 0.0f ? NRD_INF : normalSpecHitT.w);
                    currentNormalAveraged += normalSpecHitT.xyz();
                }
            currentNormalAveraged = DivConst(currentNormalAveraged, 9.0f);

            float currentRoughnessModified = SPEC ? Filtering::GetModifiedRoughnessFromNormalVariance(currentRoughness, currentNormalAveraged) : 0.0f;

            float specular1stMoment = Color::Luminance(specularIllumination.xyz());
            float specular2ndMoment = specular1stMoment * specular1stMoment;
            float diffuse1stMoment = Color::Luminance(diffuseIllumination);
            float diffuse2ndMoment = diffuse1stMoment * diffuse1stMoment;

            // surface parallax
            float smbParallaxInPixels1 = ComputeParallaxInPixels(prevWorldPos + c.gCameraDelta.xyz(), c.gOrthoMode == 0.0f ? prevUVSMB : pixelUv, c.gWorldToClipPrev, rectSize);
            float smbParallaxInPixels2 = ComputeParallaxInPixels(prevWorldPos - c.gCameraDelta.xyz(), c.gOrthoMode == 0.0f ? pixelUv : prevUVSMB, c.gWorldToClip, rectSize);
            float smbParallaxInPixelsMax = max(smbParallaxInPixels1, smbParallaxInPixels2);
            float smbParallaxInPixelsMin = min(smbParallaxInPixels1, smbParallaxInPixels2);

            float pixelSize = PixelRadiusToWorld(c.gUnproject, c.gOrthoMode, 1.0f, currentLinearZ);

            // disocclusion threshold
            float disocclusionThresholdMix = 0.0f;
            if (currentMaterialID == c.gStrandMaterialID)
                disocclusionThresholdMix = Div(pixelSize, pixelSize + c.gStrandThickness); // NRD_GetNormalizedStrandThickness, NRD.hlsli:1158-1161 (round 4: was restated as saturate( thickness / pixelSize ); found by the per-pass comparison with the reference text)
            if (c.gHasDisocclusionThresholdMix)
                disocclusionThresholdMix = gIn_DisocclusionThresholdMix.Load(ox + px, oy + py).x;
            float disocclusionThreshold = lerp(c.gDisocclusionThreshold, c.gDisocclusionThresholdAlternate, disocclusionThresholdMix);

            // ---------------------------------------------------------------- surface motion based history (loadSurfaceMotionBasedPrevData)
            float footprintQuality, historyLength, SMBReprojectionFound;
            float4 prevDiffuseIllumAnd2ndMomentSMB(0.0f), prevDiffuseSH(0.0f), prevDiffuseResponsiveSH(0.0f);
            float3 prevDiffuseResponsiveSMB(0.0f);
            float4 prevSpecularIllumAnd2ndMomentSMB(0.0f), prevSpecularSMBSH(0.0f), prevSpecularSMBResponsiveSH(0.0f);
            float3 prevSpecularResponsiveSMB(0.0f);
            float prevReflectionHitTSMB = 0.0f;
            {
                float3 smbNormal = normalize(currentNormalAveraged);
                float2 prevPixelPosFloat = prevUVSMB * c.gRectSizePrev;
                float2 originF = floor(prevPixelPosFloat - 0.5f);
                int bx = (int)originF.x, by = (int)originF.y;
                float2 bilinearWeights = float2(frac(prevPixelPosFloat.x - 0.5f), frac(prevPixelPosFloat.y - 0.5f));

                // the 12-tap footprint (4x4 without corners), coordinates clamped to the plane like a clamp-sampler gather
                auto Z = [&](int dx, int dy) { return UnpackViewZ(c, gPrev_ViewZ.FetchClamped(bx + dx, by + dy).x); };
                auto M = [&](int dx, int dy) { return gPrev_MaterialID.FetchClamped(bx + dx, by + dy).x * 255.0f; };

                float frustumSize = pixelSize * float(min(rectW, rectH));
                float disocclusionThresholdSlopeScale = Rcp(lerp(lerp(0.05f, 1.0f, NoV), 1.0f, saturate(DivConst(smbParallaxInPixelsMax, 30.0f))));
                float4 smbDisocclusionThreshold = float4(saturate(disocclusionThreshold * disocclusionThresholdSlopeScale) * frustumSize);
                smbDisocclusionThreshold *= IsInScreenBilinear(originF, c.gRectSizePrev);
                smbDisocclusionThreshold -= NRD_EPS;

                float3 prevViewPos = Geometry::AffineTransform(c.gWorldToViewPrev, prevWorldPos);
                float minMaterialID = min(c.gSpecMinMaterial, c.gDiffMinMaterial);
                auto Valid = [&](int dx, int dy, float threshold) {
                    float v = step(fabsf(Z(dx, dy) - prevViewPos.z), threshold);
                    return v * Cmp(CompareMaterials(currentMaterialID, M(dx, dy), minMaterialID));
                };
                // quads in the reference's order; the last listed tap of each group is a bilinear tap (see tapsValid*.* picks)
                float3 tapsValid0 = float3(Valid(0, -1, smbDisocclusionThreshold.x), Valid(-1, 0, smbDisocclusionThreshold.x), Valid(0, 0, smbDisocclusionThreshold.x));
                float3 tapsValid1 = float3(Valid(1, -1, smbDisocclusionThreshold.y), Valid(1, 0, smbDisocclusionThreshold.y), Valid(2, 0, smbDisocclusionThreshold.y));
                float3 tapsValid2 = float3(Valid(-1, 1, smbDisocclusionThreshold.z), Valid(0, 1, smbDisocclusionThreshold.z), Valid(0, 2, smbDisocclusionThreshold.z));
                float3 tapsValid3 = float3(Valid(1, 1, smbDisocclusionThreshold.w), Valid(2, 1, smbDisocclusionThreshold.w), Valid(1, 2, smbDisocclusionThreshold.w));

                float bicubicFootprintValid = sum(tapsValid0 + tapsValid1 + tapsValid2 + tapsValid3) > 11.5f ? 1.0f : 0.0f;
                float4 bilinearTapsValid = float4(tapsValid0.z, tapsValid1.y, tapsValid2.y, tapsValid3.x);

                // averaged previous normal
                float3 prevNormalFlat = UnpackPrevNormalRoughness(gPrev_Normal_Roughness.SampleLinearTexel(float2(float(bx) + 1.0f, float(by) + 1.0f))).xyz();
                prevNormalFlat = Geometry::RotateVector(c.gWorldPrevToWorld, prevNormalFlat);
                if (dot(smbNormal, prevNormalFlat) < 0.0f) { // back-facing history
                    bilinearTapsValid = float4(0.0f);
                    bicubicFootprintValid = 0.0f;
                }

                Filtering::Bilinear bilinear;
                bilinear.weights = bilinearWeights;
                float4 bilinearCustomWeights = Filtering::GetBilinearCustomWeights(bilinear, bilinearTapsValid);
                bool useBicubic = bicubicFootprintValid > 0.0f;

                HistoryFilter hf = MakeHistoryFilter(prevPixelPosFloat, bilinearCustomWeights, useBicubic);
                if (DIFF) {
                    prevDiffuseIllumAnd2ndMomentSMB = vmax0(FetchHistoryColor(hf, *diff.prev));
                    prevDiffuseResponsiveSMB = vmax0(FetchHistoryColor(hf, *diff.fast)).xyz();
                }
                if (SPEC) {
                    prevSpecularIllumAnd2ndMomentSMB = vmax0(FetchHistoryColor(hf, *spec.prev));
                    prevSpecularResponsiveSMB = vmax0(FetchHistoryColor(hf, *spec.fast)).xyz();
                }
                if (SH) {
                    if (DIFF) {
                        prevDiffuseSH = BilinearWithCustomWeightsFloat4(*diff.prevSh, bx, by, bilinearCustomWeights);
                        prevDiffuseResponsiveSH = BilinearWithCustomWeightsFloat4(*diff.fastSh, bx, by, bilinearCustomWeights);
                    }
                    if (SPEC) {
                        prevSpecularSMBSH = BilinearWithCustomWeightsFloat4(*spec.prevSh, bx, by, bilinearCustomWeights);
                        prevSpecularSMBResponsiveSH = BilinearWithCustomWeightsFloat4(*spec.fastSh, bx, by, bilinearCustomWeights);
                    }
                }

                historyLength = 255.0f * BilinearWithCustomWeightsImmediateFloat(gPrev_HistoryLength.FetchClamped(bx, by).x, gPrev_HistoryLength.FetchClamped(bx + 1, by).x,
                                             gPrev_HistoryLength.FetchClamped(bx, by + 1).x, gPrev_HistoryLength.FetchClamped(bx + 1, by + 1).x, bilinearCustomWeights);
                if (SPEC) {
                    prevReflectionHitTSMB = BilinearWithCustomWeightsImmediateFloat(gPrev_SpecHitDist->FetchClamped(bx, by).x, gPrev_SpecHitDist->FetchClamped(bx + 1, by).x,
                        gPrev_SpecHitDist->FetchClamped(bx, by + 1).x, gPrev_SpecHitDist->FetchClamped(bx + 1, by + 1).x, bilinearCustomWeights);
                    prevReflectionHitTSMB = max(0.001f, prevReflectionHitTSMB);
                }

                SMBReprojectionFound = bicubicFootprintValid > 0.0f ? 2.0f : 1.0f;
                footprintQuality = bicubicFootprintValid > 0.0f ? 1.0f : sum(bilinearCustomWeights);
                bool anyValid = bilinearTapsValid.x != 0.0f || bilinearTapsValid.y != 0.0f || bilinearTapsValid.z != 0.0f || bilinearTapsValid.w != 0.0f;
                if (!anyValid) {
                    SMBReprojectionFound = 0.0f;
                    footprintQuality = 0.0f;
                }
            }

            historyLength = historyLength + 1.0f;
            historyLength = min(RELAX_MAX_ACCUM_FRAME_NUM, historyLength);

            // avoid footprint stretching due to the changed viewing angle
            float3 Vprev = c.gOrthoMode == 0.0f ? -normalize(prevWorldPos - c.gCameraDelta.xyz()) : -normalize(c.gPrevFrustumForward.xyz());
            float NoVprev = fabsf(dot(currentNormal, Vprev));
            float sizeQuality = Div(NoVprev + 1e-3f, NoV + 1e-3f);
            sizeQuality *= sizeQuality;
            sizeQuality *= sizeQuality;
            footprintQuality *= lerp(0.1f, 1.0f, saturate(sizeQuality + fabsf(c.gOrthoMode)));

            if (footprintQuality < 1.0f) {
                historyLength *= HwSqrt(footprintQuality);
                historyLength = max(historyLength, 1.0f);
            }
            historyLength = c.gResetHistory != 0 ? 1.0f : historyLength;

            float maxAccumulatedFrameNum = 1.0f + ((DIFF && SPEC) ? max(c.gDiffMaxAccumulatedFrameNum, c.gSpecMaxAccumulatedFrameNum)
                                                                  : (DIFF ? c.gDiffMaxAccumulatedFrameNum : c.gSpecMaxAccumulatedFrameNum));
            historyLength = min(historyLength, maxAccumulatedFrameNum);

            uint32_t checkerboard = Sequence::CheckerBoard((uint32_t)px, (uint32_t)py, c.gFrameIndex);

            if (DIFF) {
                float diffMaxAccumulatedFrameNum = c.gDiffMaxAccumulatedFrameNum;
                float diffMaxFastAccumulatedFrameNum = c.gDiffMaxFastAccumulatedFrameNum;
                if (c.gHasHistoryConfidence) {
                    float inDiffConfidence = diff.confidence->Load(ox + px, oy + py).x;
                    diffMaxAccumulatedFrameNum *= inDiffConfidence;
                    diffMaxFastAccumulatedFrameNum *= inDiffConfidence;
                }
                float diffHistoryLength = historyLength;
                float diffuseAlpha = SMBReprojectionFound > 0.0f ? max(Rcp(diffMaxAccumulatedFrameNum + 1.0f), Rcp(diffHistoryLength)) : 1.0f;
                float diffuseAlphaResponsive = SMBReprojectionFound > 0.0f ? max(Rcp(diffMaxFastAccumulatedFrameNum + 1.0f), Rcp(diffHistoryLength)) : 1.0f;

                bool diffHasData = true;
                if (c.gDiffCheckerboard != 2u)
                    diffHasData = checkerboard == c.gDiffCheckerboard;
                if (!diffHasData && diffHistoryLength > 1.0f) {
                    diffuseAlpha *= 1.0f - c.gCheckerboardResolveAccumSpeed;
                    diffuseAlphaResponsive *= 1.0f - c.gCheckerboardResolveAccumSpeed;
                }

                float4 accumulated = lerp(prevDiffuseIllumAnd2ndMomentSMB, float4(diffuseIllumination, diffuse2ndMoment), diffuseAlpha);
                float3 accumulatedResponsive = lerp(prevDiffuseResponsiveSMB, diffuseIllumination, diffuseAlphaResponsive);
                diff.out->Store(px, py, accumulated);
                diff.outFast->Store(px, py, float4(accumulatedResponsive, 0.0f));
                if (SH) {
                    diff.outSh->Store(px, py, lerp(prevDiffuseSH, diffuseSH, diffuseAlpha));
                    diff.outFastSh->Store(px, py, lerp(prevDiffuseResponsiveSH, diffuseSH, diffuseAlphaResponsive));
                }
            }

            gOut_HistoryLength.Store(px, py, DivConst(historyLength, 255.0f));

            if (SPEC) {
                float specMaxAccumulatedFrameNum = c.gSpecMaxAccumulatedFrameNum;
                float specMaxFastAccumulatedFrameNum = c.gSpecMaxFastAccumulatedFrameNum;
                if (c.gHasHistoryConfidence) {
                    float inSpecConfidence = spec.confidence->Load(ox + px, oy + py).x;
                    specMaxAccumulatedFrameNum *= inSpecConfidence;
                    specMaxFastAccumulatedFrameNum *= inSpecConfidence;
                }
                float specHistoryLength = historyLength;
                float specHistoryFrames = min(specMaxAccumulatedFrameNum, specHistoryLength);
                float specHistoryResponsiveFrames = min(specMaxFastAccumulatedFrameNum, specHistoryLength);

                float hitDist = minHitDist3x3 == NRD_INF ? 0.0f : minHitDist3x3;

                // curvature along the direction of motion
                float curvature;
                {
                    float2 uvForZeroParallax = c.gOrthoMode == 0.0f ? prevUVSMB : pixelUv;
                    float2 deltaUv = uvForZeroParallax - Geometry::GetScreenUv(c.gWorldToClipPrev, prevWorldPos + c.gCameraDelta.xyz());
                    deltaUv *= rectSize;
                    deltaUv = Div(deltaUv, max(smbParallaxInPixels1, 1.0f / 256.0f));

                    auto Edge = [&](float2 duv, int sx, int sy, float3& nOut, float3& xOut) {
                        float3 x = GetCurrentWorldPosFromClipSpaceXY(c, (pixelUv + duv * c.gRectSizeInv) * 2.0f - 1.0f, 1.0f);
                        float3 v = c.gOrthoMode == 0.0f ? normalize(-x) : c.gFrustumForward.xyz();
                        float3 o = c.gOrthoMode == 0.0f ? float3(0.0f) : x;
                        xOut = o + Div(v * dot(currentWorldPos - o, currentNormal), dot(currentNormal, v)); // line-plane intersection
                        nOut = Shared(px + sx, py + sy).xyz();
                    };
                    float3 n10, x10, n01, x01;
                    Edge(float2(1.0f, 0.0f), 1, 0, n10, x10);
                    Edge(float2(0.0f, 1.0f), 0, 1, n01, x01);

                    float2 w = abs(deltaUv) + 1.0f / 256.0f;
                    w = Div(w, w.x + w.y);
                    float3 x = x10 * w.x + x01 * w.y;
                    float3 n = normalize(n10 * w.x + n01 * w.y);

                    // high parallax: flatten the surface
                    float deltaUvLenFixed = smbParallaxInPixelsMin;
                    deltaUvLenFixed *= 1.0f; // NRD_USE_HIGH_PARALLAX_CURVATURE_SILHOUETTE_FIX = 0
                    deltaUvLenFixed *= 1.0f + c.gFramerateScale * Sequence::Bayer4x4((uint32_t)px, (uint32_t)py, c.gFrameIndex);

                    float2 motionUvHigh = pixelUv + deltaUvLenFixed * deltaUv * c.gRectSizeInv;
                    motionUvHigh = (floor(motionUvHigh * rectSize) + 0.5f) * c.gRectSizeInv;
                    if (deltaUvLenFixed > 1.0f && IsInScreenNearest(motionUvHigh) != 0.0f) {
                        float2 uvScaled = ClampUvToViewport(c, motionUvHigh) + c.gRectOffset;
                        float zHigh = UnpackViewZ(c, gIn_ViewZ.SampleNearest(uvScaled).x);
                        float3 xHigh = GetCurrentWorldPosFromClipSpaceXY(c, motionUvHigh * 2.0f - 1.0f, zHigh);
                        float3 nHigh = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.SampleNearest(uvScaled)).xyz();
                        float zError = fabsf(zHigh - currentLinearZ) * rcp(max(zHigh, currentLinearZ));
                        bool cmp = zError < NRD_CURVATURE_Z_THRESHOLD;
                        n = cmp ? nHigh : n;
                        x = cmp ? xHigh : x;
                    }

                    float3 edge = x - currentWorldPos;
                    float edgeLenSq = Math::LengthSquared(edge);
                    curvature = dot(n - currentNormal, edge) * Math::PositiveRcp(edgeLenSq);
                }

                float hitDistFocused = ApplyThinLensEquation(hitDist, curvature);

                // ---------------------------------------------------------------- virtual motion based history (loadVirtualMotionBasedPrevData)
                float4 prevSpecularIllumAnd2ndMomentVMB(0.0f), prevSpecularResponsiveVMB(0.0f), prevSpecularVMBSH(0.0f), prevSpecularVMBResponsiveSH(0.0f);
                float3 prevNormalVMB = currentNormal;
                float prevRoughnessVMB = 0.0f, prevReflectionHitTVMB = c.gDenoisingRange, VMBReprojectionFound;
                float2 prevUVVMB;
                {
                    float3 virtualViewVector = normalize(currentViewVector) * hitDistFocused;
                    float3 prevVirtualWorldPos = prevWorldPos + virtualViewVector;

                    prevUVVMB = ScreenUvNoKill(c.gWorldToClipPrev, prevVirtualWorldPos);
                    prevUVVMB = currentMaterialID == c.gCameraAttachedReflectionMaterialID ? prevUVSMB : prevUVVMB;

                    float2 prevVirtualPixelPosFloat = prevUVVMB * c.gRectSizePrev;
                    float2 originF = floor(prevVirtualPixelPosFloat - 0.5f);
                    int bx = (int)originF.x, by = (int)originF.y;
                    float2 bilinearWeights = float2(frac(prevVirtualPixelPosFloat.x - 0.5f), frac(prevVirtualPixelPosFloat.y - 0.5f));

                    float3 currentWorldPosShifted = currentWorldPos - c.gCameraDelta.xyz();

                    float4 vmbDisocclusionThreshold = float4(disocclusionThreshold * (c.gOrthoMode == 0.0f ? currentLinearZ : 1.0f));
                    vmbDisocclusionThreshold *= IsInScreenBilinear(originF, c.gRectSizePrev);
                    vmbDisocclusionThreshold -= NRD_EPS;

                    auto TapValid = [&](int dx, int dy, float threshold) {
                        float z = UnpackViewZ(c, gPrev_ViewZ.FetchClamped(bx + dx, by + dy).x);
                        float3 prevWorldPosInTap = GetPreviousWorldPosFromPixelPos(c, bx + dx, by + dy, z);
                        float3 posDiff = currentWorldPosShifted - prevWorldPosInTap;
                        float maxPlaneDistance = fabsf(dot(posDiff, currentNormal));
                        float valid = maxPlaneDistance > threshold ? 0.0f : 1.0f; // isReprojectionTapValid
                        float m = gPrev_MaterialID.FetchClamped(bx + dx, by + dy).x * 255.0f;
                        return valid * Cmp(CompareMaterials(currentMaterialID, m, c.gSpecMinMaterial));
                    };
                    float4 bilinearTapsValid = float4(TapValid(0, 0, vmbDisocclusionThreshold.x), TapValid(1, 0, vmbDisocclusionThreshold.y), TapValid(0, 1, vmbDisocclusionThreshold.z),
                        TapValid(1, 1, vmbDisocclusionThreshold.w));
                    bool anyValid = bilinearTapsValid.x != 0.0f || bilinearTapsValid.y != 0.0f || bilinearTapsValid.z != 0.0f || bilinearTapsValid.w != 0.0f;
                    bool allValid = bilinearTapsValid.x != 0.0f && bilinearTapsValid.y != 0.0f && bilinearTapsValid.z != 0.0f && bilinearTapsValid.w != 0.0f;

                    if (anyValid) {
                        Filtering::Bilinear bilinear;
                        bilinear.weights = bilinearWeights;
                        float4 bilinearCustomWeights = Filtering::GetBilinearCustomWeights(bilinear, bilinearTapsValid);
                        bool useBicubic = SMBReprojectionFound == 2.0f && allValid;

                        HistoryFilter hf = MakeHistoryFilter(prevVirtualPixelPosFloat, bilinearCustomWeights, useBicubic);
                        prevSpecularIllumAnd2ndMomentVMB = vmax0(FetchHistoryColor(hf, *spec.prev));
                        prevSpecularResponsiveVMB = vmax0(FetchHistoryColor(hf, *spec.fast));
                        if (SH) {
                            prevSpecularVMBSH = BilinearWithCustomWeightsFloat4(*spec.prevSh, bx, by, bilinearCustomWeights);
                            prevSpecularVMBResponsiveSH = BilinearWithCustomWeightsFloat4(*spec.fastSh, bx, by, bilinearCustomWeights);
                        }

                        prevReflectionHitTVMB = gPrev_SpecHitDist->SampleLinearTexelScalar(prevUVVMB * resolutionScalePrev * PrevSize(*gPrev_SpecHitDist));
                        prevReflectionHitTVMB = max(0.001f, prevReflectionHitTVMB);

                        float4 prevNormalRoughness = UnpackPrevNormalRoughness(gPrev_Normal_Roughness.SampleLinearTexel(prevUVVMB * resolutionScalePrev * PrevSize(gPrev_Normal_Roughness)));
                        prevNormalVMB = Geometry::RotateVector(c.gWorldPrevToWorld, prevNormalRoughness.xyz());
                        prevRoughnessVMB = prevNormalRoughness.w;
                    }
                    VMBReprojectionFound = allValid ? 1.0f : 0.0f;
                }

                // amount of virtual motion
                float4 D = ImportanceSampling::GetSpecularDominantDirection(currentNormal, V, currentRoughnessModified);
                float virtualHistoryAmount = VMBReprojectionFound * D.w;
                virtualHistoryAmount *= c.gOrthoMode == 0.0f ? 1.0f : 0.75f;
                virtualHistoryAmount *= Cmp(dot(prevNormalVMB, currentNormalAveraged) > 0.0f);

                float2 uvDiff = prevUVVMB - prevUVSMB;
                float uvDiffLengthInPixels = length(uvDiff * rectSize);

                float tanCurvature = fabsf(curvature * pixelSize);
                tanCurvature *= max(Div(uvDiffLengthInPixels, max(NoV, 0.01f)), 1.0f);
                float curvatureAngle = atan(tanCurvature);

                float lobeHalfAngle = max(atan(GetSpecLobeTanHalfAngle(currentRoughnessModified)), RELAX_NORMAL_ULP);
                float normalWeight = GetEncodingAwareNormalWeightR(currentNormal, prevNormalVMB, lobeHalfAngle, curvatureAngle, RELAX_NORMAL_ULP, true);
                virtualHistoryAmount *= lerp(1.0f - saturate(uvDiffLengthInPixels), 1.0f, normalWeight);

                float2 relaxedRoughnessWeightParams = GetRelaxedRoughnessWeightParams(currentRoughness * currentRoughness, c.gRoughnessFraction);
                float virtualRoughnessWeight = ComputeWeight(prevRoughnessVMB * prevRoughnessVMB, relaxedRoughnessWeightParams.x, relaxedRoughnessWeightParams.y);
                virtualRoughnessWeight = lerp(1.0f - saturate(uvDiffLengthInPixels), 1.0f, virtualRoughnessWeight);
                virtualHistoryAmount *= c.gOrthoMode == 0.0f ? virtualRoughnessWeight : 1.0f;
                float specVMBConfidence = virtualRoughnessWeight * 0.9f + 0.1f;

                // look back 1 and 2 frames
                uvDiff *= Math::Rsqrt(Math::LengthSquared(uvDiff));
                uvDiff = Div(uvDiff, c.gRectSizePrev);
                uvDiff *= saturate(DivConst(uvDiffLengthInPixels, 0.1f)) + uvDiffLengthInPixels * 0.5f;
                float2 backUV1 = prevUVVMB + 1.0f * uvDiff;
                float2 backUV2 = prevUVVMB + 2.0f * uvDiff;
                float4 backNormalRoughness1 = UnpackPrevNormalRoughness(gPrev_Normal_Roughness.SampleLinearTexel(backUV1 * resolutionScalePrev * PrevSize(gPrev_Normal_Roughness)));
                float4 backNormalRoughness2 = UnpackPrevNormalRoughness(gPrev_Normal_Roughness.SampleLinearTexel(backUV2 * resolutionScalePrev * PrevSize(gPrev_Normal_Roughness)));
                float3 backNormal1 = Geometry::RotateVector(c.gWorldPrevToWorld, backNormalRoughness1.xyz());
                float3 backNormal2 = Geometry::RotateVector(c.gWorldPrevToWorld, backNormalRoughness2.xyz());
                float prevPrevNormalWeight = IsInScreenNearest(backUV1) != 0.0f ? GetEncodingAwareNormalWeightR(prevNormalVMB, backNormal1, lobeHalfAngle, curvatureAngle * 2.0f, RELAX_NORMAL_ULP, true) : 1.0f;
                prevPrevNormalWeight *= IsInScreenNearest(backUV2) != 0.0f ? GetEncodingAwareNormalWeightR(prevNormalVMB, backNormal2, lobeHalfAngle, curvatureAngle * 3.0f, RELAX_NORMAL_ULP, true) : 1.0f;
                virtualHistoryAmount *= 0.33f + 0.67f * prevPrevNormalWeight;
                specVMBConfidence *= 0.33f + 0.67f * prevPrevNormalWeight;
                float rw = ComputeWeight(backNormalRoughness1.w * backNormalRoughness1.w, relaxedRoughnessWeightParams.x, relaxedRoughnessWeightParams.y);
                rw *= ComputeWeight(backNormalRoughness2.w * backNormalRoughness2.w, relaxedRoughnessWeightParams.x, relaxedRoughnessWeightParams.y);
                virtualHistoryAmount *= c.gOrthoMode == 0.0f ? rw * 0.9f + 0.1f : 1.0f;

                // hit distance confidence
                float SMC = GetSpecMagicCurve(currentRoughnessModified);
                float hitDistC = lerp(specularIllumination.w, prevReflectionHitTSMB, SMC);
                float hitDist1 = ApplyThinLensEquation(hitDistC, curvature);
                float hitDist2 = ApplyThinLensEquation(prevReflectionHitTVMB, curvature);
                float maxDist = max(hitDist1, hitDist2);
                float dHitT = fabsf(hitDist1 - hitDist2);
                float dHitTMultiplier = lerp(20.0f, 0.0f, SMC);
                float virtualHistoryHitDistConfidence = 1.0f - saturate(Div(dHitTMultiplier * dHitT, currentLinearZ + maxDist));
                virtualHistoryHitDistConfidence = lerp(virtualHistoryHitDistConfidence, 1.0f, SMC);

                // virtual UV discrepancy
                float3 virtualWorldPos = GetXvirtual(hitDist, curvature, currentWorldPos, prevWorldPos, currentNormal, V, currentRoughness);
                float virtualWorldPosLength = length(virtualWorldPos);
                float hitDistForTrackingPrev = prevSpecularResponsiveVMB.w;
                float3 prevVirtualWorldPos2 = GetXvirtual(hitDistForTrackingPrev, curvature, currentWorldPos, prevWorldPos, currentNormal, V, currentRoughness);
                float virtualWorldPosLengthPrev = length(prevVirtualWorldPos2);
                float2 prevUVVMBTest = ScreenUvNoKill(c.gWorldToClipPrev, prevVirtualWorldPos2);
                prevUVVMBTest = currentMaterialID == c.gCameraAttachedReflectionMaterialID ? prevUVSMB : prevUVVMBTest;

                float lobeTanHalfAngle = GetSpecLobeTanHalfAngle(currentRoughness, 0.6f);
                lobeTanHalfAngle = max(lobeTanHalfAngle, 0.5f * c.gRectSizeInv.x);
                float unproj1 = Div(min(hitDist, hitDistForTrackingPrev), PixelRadiusToWorld(c.gUnproject, c.gOrthoMode, 1.0f, max(virtualWorldPosLength, virtualWorldPosLengthPrev)));
                float lobeRadiusInPixels = lobeTanHalfAngle * unproj1;
                float deltaParallaxInPixels = length((prevUVVMBTest - prevUVVMB) * rectSize);
                virtualHistoryHitDistConfidence *= Math::SmoothStep(lobeRadiusInPixels + 0.25f, 0.0f, deltaParallaxInPixels);

                // surface motion signal
                float specSMBConfidence = (SMBReprojectionFound > 0.0f ? 1.0f : 0.0f) * GetEncodingAwareNormalWeightR(V, Vprev, Div(lobeHalfAngle * NoV, c.gFramerateScale), 0.0f, 0.0f, false);
                float specSMBAlpha = 1.0f - specSMBConfidence;
                float specSMBResponsiveAlpha = 1.0f - specSMBConfidence;
                specSMBAlpha = max(specSMBAlpha, Rcp(1.0f + specHistoryFrames));
                specSMBResponsiveAlpha = max(specSMBAlpha, Rcp(1.0f + specHistoryResponsiveFrames));

                bool specHasData = true;
                if (c.gSpecCheckerboard != 2u)
                    specHasData = checkerboard == c.gSpecCheckerboard;
                if (!specHasData && smbParallaxInPixelsMax < 0.5f) {
                    specSMBAlpha *= 1.0f - c.gCheckerboardResolveAccumSpeed * (SMBReprojectionFound > 0.0f ? 1.0f : 0.0f);
                    specSMBResponsiveAlpha *= 1.0f - c.gCheckerboardResolveAccumSpeed * (SMBReprojectionFound > 0.0f ? 1.0f : 0.0f);
                }

                float3 accumulatedSpecularSMB = lerp(prevSpecularIllumAnd2ndMomentSMB.xyz(), specularIllumination.xyz(), specSMBAlpha);
                float accumulatedSpecularSMBHitT = lerp(prevReflectionHitTSMB, specularIllumination.w, max(specSMBAlpha, 0.1f));
                float accumulatedSpecularM2SMB = lerp(prevSpecularIllumAnd2ndMomentSMB.w, specular2ndMoment, specSMBAlpha);
                float3 accumulatedSpecularSMBResponsive = lerp(prevSpecularResponsiveSMB, specularIllumination.xyz(), specSMBResponsiveAlpha);

                // virtual motion signal
                float specVMBAlpha = 1.0f - specVMBConfidence;
                float specVMBResponsiveAlpha = 1.0f - specVMBConfidence * virtualHistoryHitDistConfidence;
                float specVMBHitTAlpha = specVMBResponsiveAlpha;
                specVMBAlpha = max(specVMBAlpha, Rcp(1.0f + specHistoryFrames));
                specVMBResponsiveAlpha = max(specVMBResponsiveAlpha, Rcp(1.0f + specHistoryResponsiveFrames));
                specVMBHitTAlpha = max(specVMBHitTAlpha, Rcp(1.0f + specHistoryFrames));
                if (!specHasData && smbParallaxInPixelsMax < 0.5f) {
                    float k = 1.0f - c.gCheckerboardResolveAccumSpeed * (VMBReprojectionFound > 0.0f ? 1.0f : 0.0f);
                    specVMBAlpha *= k;
                    specVMBResponsiveAlpha *= k;
                    specVMBHitTAlpha *= k;
                }

                float3 accumulatedSpecularVMB = lerp(prevSpecularIllumAnd2ndMomentVMB.xyz(), specularIllumination.xyz(), specVMBAlpha);
                float accumulatedSpecularVMBHitT = lerp(prevReflectionHitTVMB, specularIllumination.w, max(specVMBHitTAlpha, 0.1f));
                float accumulatedSpecularM2VMB = lerp(prevSpecularIllumAnd2ndMomentVMB.w, specular2ndMoment, specVMBAlpha);
                float3 accumulatedSpecularVMBResponsive = lerp(prevSpecularResponsiveVMB.xyz(), specularIllumination.xyz(), specVMBResponsiveAlpha);

                // fall back to surface motion if virtual motion doesn't go well
                virtualHistoryAmount *= saturate(Div(specVMBConfidence, specSMBConfidence + NRD_EPS));

                float accumulatedReflectionHitT = lerp(accumulatedSpecularSMBHitT, accumulatedSpecularVMBHitT, virtualHistoryAmount);
                float3 accumulatedSpecularIllumination = lerp(accumulatedSpecularSMB, accumulatedSpecularVMB, virtualHistoryAmount);
                float3 accumulatedSpecularIlluminationResponsive = lerp(accumulatedSpecularSMBResponsive, accumulatedSpecularVMBResponsive, virtualHistoryAmount);
                float accumulatedSpecular2ndMoment = lerp(accumulatedSpecularM2SMB, accumulatedSpecularM2VMB, virtualHistoryAmount);

                if (SH) {
                    float4 accumulatedSpecularSMBSH = lerp(prevSpecularSMBSH, specularSH, specSMBAlpha);
                    float4 accumulatedSpecularSMBResponsiveSH = lerp(prevSpecularSMBResponsiveSH, specularSH, specSMBResponsiveAlpha);
                    float4 accumulatedSpecularVMBSH = lerp(prevSpecularVMBSH, specularSH, specVMBAlpha);
                    float4 accumulatedSpecularVMBResponsiveSH = lerp(prevSpecularVMBResponsiveSH, specularSH, specVMBResponsiveAlpha);
                    float4 accumulatedSpecularSH = lerp(accumulatedSpecularSMBSH, accumulatedSpecularVMBSH, virtualHistoryAmount);
                    float4 accumulatedSpecularResponsiveSH = lerp(accumulatedSpecularSMBResponsiveSH, accumulatedSpecularVMBResponsiveSH, virtualHistoryAmount);
                    spec.outSh->Store(px, py, float4(accumulatedSpecularSH.xyz(), currentRoughnessModified));
                    spec.outFastSh->Store(px, py, accumulatedSpecularResponsiveSH);
                }

                float specularHistoryConfidence = lerp(specSMBConfidence, specVMBConfidence, virtualHistoryAmount);
                if (accumulatedSpecular2ndMoment == 0.0f)
                    accumulatedSpecular2ndMoment = c.gSpecVarianceBoost * (1.0f - specularHistoryConfidence);

                spec.out->Store(px, py, float4(accumulatedSpecularIllumination, accumulatedSpecular2ndMoment));
                spec.outFast->Store(px, py, float4(accumulatedSpecularIlluminationResponsive, hitDist));
                gOut_SpecHitDist->Store(px, py, accumulatedReflectionHitT);
                gOut_SpecReprojectionConfidence->Store(px, py, specularHistoryConfidence);
            }
        }
}

// ================================================================================================ HistoryFix
template <bool DIFF, bool SPEC, bool SH>
void HistoryFix(const PassIO& io) {
    const RelaxCB& c = *(const RelaxCB*)io.constants;
    Cursor cur{io.t};
    Sig spec, diff;
    const Tex& gIn_Tiles = *cur.next();
    if (SPEC) spec.in = cur.next();
    if (DIFF) diff.in = cur.next();
    const Tex& gIn_HistoryLength = *cur.next();
    const Tex& gIn_Normal_Roughness = *cur.next();
    const Tex& gIn_ViewZ = *cur.next();
    if (SH && SPEC) spec.inSh = cur.next();
    if (SH && DIFF) diff.inSh = cur.next();
    if (SPEC) spec.out = cur.next();
    if (DIFF) diff.out = cur.next();
    if (SH && SPEC) spec.outSh = cur.next();
    if (SH && DIFF) diff.outSh = cur.next();

    const int rectW = c.gRectSize[0], rectH = c.gRectSize[1];

#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < rectH; py++)
        for (int px = 0; px < rectW; px++) {
            if (gIn_Tiles.Load(px >> 4, py >> 4).x != 0.0f)
                continue;
            float centerViewZ = UnpackViewZ(c, gIn_ViewZ.Load(px, py).x);
            float historyLength = 255.0f * gIn_HistoryLength.Load(px, py).x;
            if (centerViewZ > c.gDenoisingRange || (historyLength > c.gHistoryFixFrameNum || c.gHistoryFixFrameNum == 1.0f))
                continue;

            float centerMaterialID;
            float4 centerNormalRoughness = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.Load(px, py), centerMaterialID);
            float3 centerNormal = centerNormalRoughness.xyz();
            float centerRoughness = centerNormalRoughness.w;
            float3 centerWorldPos = GetCurrentWorldPosFromPixelPos(c, px, py, centerViewZ);
            float3 centerV = -normalize(centerWorldPos);
            float depthThreshold = c.gDepthThreshold * (c.gOrthoMode == 0.0f ? centerViewZ : 1.0f);

            float4 diffuseSum = DIFF ? diff.in->Load(px, py) : float4(0.0f);
            float4 diffuseSumSH = (DIFF && SH) ? diff.inSh->Load(px, py) : float4(0.0f);
            float diffuseWSum = 1.0f;
            float4 specularSum = SPEC ? spec.in->Load(px, py) : float4(0.0f);
            float4 specularSumSH = (SPEC && SH) ? spec.inSh->Load(px, py) : float4(0.0f);
            float roughnessModified = specularSumSH.w;
            float specularWSum = 1.0f;
            float2 specularNormalWeightParams = SPEC ? GetNormalWeightParams_ATrous(centerRoughness, 5.0f, 1.0f, 0.0f, c.gLobeAngleFraction, c.gSpecLobeAngleSlack) : float2(0.0f);

            float r = Div(c.gHistoryFixBasePixelStride, 1.0f + historyLength);
            r = floorf(r + 0.5f);

            for (int j = -2; j <= 2; j++)
                for (int i = -2; i <= 2; i++) {
                    int dx = (int)(float(i) * r), dy = (int)(float(j) * r);
                    int sx = px + dx, sy = py + dy;
                    bool isInside = sx >= 0 && sy >= 0 && sx < rectW && sy < rectH;
                    if (i == 0 && j == 0)
                        continue;

                    float sampleMaterialID;
                    float3 sampleNormal = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.Load(sx, sy), sampleMaterialID).xyz();
                    float sampleViewZ = UnpackViewZ(c, gIn_ViewZ.Load(sx, sy).x);
                    float3 sampleWorldPos = GetCurrentWorldPosFromPixelPos(c, sx, sy, sampleViewZ);
                    float geometryWeight = GetPlaneDistanceWeight_Atrous(centerWorldPos, centerNormal, sampleWorldPos, depthThreshold);

                    if (DIFF) {
                        float diffuseW = geometryWeight;
                        diffuseW *= pow(max(0.01f, dot(centerNormal, sampleNormal)), max(c.gHistoryFixEdgeStoppingNormalPower, 0.01f)); // getDiffuseNormalWeight
                        diffuseW = isInside ? diffuseW : 0.0f;
                        diffuseW *= Cmp(CompareMaterials(sampleMaterialID, centerMaterialID, c.gDiffMinMaterial));
                        if (diffuseW > 1e-4f) {
                            diffuseSum += diff.in->Load(sx, sy) * diffuseW;
                            if (SH)
                                diffuseSumSH += diff.inSh->Load(sx, sy) * diffuseW;
                            diffuseWSum += diffuseW;
                        }
                    }
                    if (SPEC) {
                        float3 sampleV = -normalize(sampleWorldPos + c.gRoughnessEdgeStoppingRelaxation * centerWorldPos);
                        float specularW = geometryWeight;
                        specularW *= GetSpecularNormalWeight_ATrous(specularNormalWeightParams, centerNormal, sampleNormal, centerV, sampleV);
                        specularW = isInside ? specularW : 0.0f;
                        specularW *= Cmp(CompareMaterials(sampleMaterialID, centerMaterialID, c.gSpecMinMaterial));
                        if (specularW > 1e-4f) {
                            specularSum += spec.in->Load(sx, sy) * specularW;
                            if (SH)
                                specularSumSH += spec.inSh->Load(sx, sy) * specularW;
                            specularWSum += specularW;
                        }
                    }
                }

            if (DIFF) {
                diff.out->Store(px, py, Div(diffuseSum, diffuseWSum));
                if (SH)
                    diff.outSh->Store(px, py, Div(diffuseSumSH, diffuseWSum));
            }
            if (SPEC) {
                spec.out->Store(px, py, Div(specularSum, specularWSum));
                if (SH)
                    spec.outSh->Store(px, py, float4(Div(specularSumSH.xyz(), specularWSum), roughnessModified));
            }
        }
}

// ================================================================================================ HistoryClamping
// one signal's clamping + anti-lag; returns the clamping factor (used to blend the SH planes)
struct ClampOut {
    float4 slow, fast;
    float clampingFactor;
};
inline ClampOut ClampSignal(const RelaxCB& c, bool isSpec, float3 fastM1, float3 fastM2, float3 noisyM1, float noisyM2, float4 fastCenterYCoCg, float4 slowIn, float3 noisyCenter,
    float historyLength) {
    float maxFast = isSpec ? c.gSpecMaxFastAccumulatedFrameNum : c.gDiffMaxFastAccumulatedFrameNum;
    float maxSlow = isSpec ? c.gSpecMaxAccumulatedFrameNum : c.gDiffMaxAccumulatedFrameNum;

    float3 sigma = vsqrt(vmax(float3(0.0f), fastM2 - fastM1 * fastM1));
    float3 colorMin = fastM1 - c.gColorBoxSigmaScale * sigma;
    float3 colorMax = fastM1 + c.gColorBoxSigmaScale * sigma;
    colorMin = vmin(colorMin, fastCenterYCoCg.xyz());
    colorMax = vmax(colorMax, fastCenterYCoCg.xyz());

    float3 slowYCoCg = RgbToYCoCg(slowIn.xyz());
    float3 clampedYCoCg = slowYCoCg;
    if (maxFast < maxSlow)
        clampedYCoCg = vmin(vmax(slowYCoCg, colorMin), colorMax);
    float3 clamped = YCoCgToRgb(clampedYCoCg);

    float4 outSlow = float4(clamped, slowIn.w);
    float3 fastCenter = YCoCgToRgb(fastCenterYCoCg.xyz());
    float4 outFast = float4(fastCenter, isSpec ? fastCenterYCoCg.w : 0.0f);
    if (historyLength <= c.gHistoryFixFrameNum)
        outSlow = isSpec ? outFast : float4(outFast.xyz(), outSlow.w);

    float clampingFactor = (clampedYCoCg.x - slowYCoCg.x) == 0.0f ? 0.0f : saturate(Div(clampedYCoCg.x - slowYCoCg.x, fastCenterYCoCg.x - slowYCoCg.x));
    if (historyLength <= c.gHistoryFixFrameNum)
        clampingFactor = 1.0f;

    // history acceleration based on (responsive - normal); 3x weaker for specular
    float historyDifferenceL = (isSpec ? 0.33f * RELAX_ANTILAG_ACCELERATION_AMOUNT_SCALE : RELAX_ANTILAG_ACCELERATION_AMOUNT_SCALE) * c.gHistoryAccelerationAmount *
                               Color::Luminance(abs(fastCenter - slowIn.xyz()));
    historyDifferenceL *= clampingFactor;
    if (historyLength <= c.gHistoryFixFrameNum)
        historyDifferenceL = 0.0f;

    float3 distanceToNoisy = noisyM1 - fastCenter;
    float distanceToNoisyL = Color::Luminance(abs(distanceToNoisy));
    float3 acceleration = distanceToNoisyL == 0.0f ? float3(0.0f) : Div(distanceToNoisy * historyDifferenceL, distanceToNoisyL);
    float accelerationL = Color::Luminance(abs(acceleration));
    float ratio = accelerationL == 0.0f ? 0.0f : Div(distanceToNoisyL, accelerationL);
    if (ratio < 1.0f)
        acceleration *= ratio;
    if (ratio <= 0.0f)
        acceleration = float3(0.0f);

    float3 slowRgb = outSlow.xyz() + acceleration;
    float3 fastRgb = outFast.xyz() + acceleration;

    // history reset
    float slowL = Color::Luminance(slowIn.xyz());
    float noisyL = Color::Luminance(noisyM1);
    float temporalSigma = c.gHistoryResetTemporalSigmaScale * HwSqrt(max(0.0f, noisyM2 - noisyL * noisyL));
    float spatialSigma = c.gHistoryResetSpatialSigmaScale * sigma.x;
    float resetAmount = Div((isSpec ? 0.5f * c.gHistoryResetAmount : c.gHistoryResetAmount) * max(0.0f, fabsf(slowL - noisyL) - spatialSigma - temporalSigma), 1.0e-6f + max(slowL, noisyL) + spatialSigma + temporalSigma);
    resetAmount = saturate(resetAmount);
    slowRgb = lerp(slowRgb, noisyCenter, resetAmount);
    fastRgb = lerp(fastRgb, noisyCenter, resetAmount);

    // 2nd moment correction
    float outL = Color::Luminance(slowRgb);
    float momentCorrection = outL * outL - slowL * slowL;
    float a = max(0.0f, outSlow.w + momentCorrection);

    ClampOut o;
    o.slow = float4(slowRgb, a);
    o.fast = float4(fastRgb, outFast.w);
    o.clampingFactor = clampingFactor;
    return o;
}

template <bool DIFF, bool SPEC, bool SH>
void HistoryClamping(const PassIO& io) {
    const RelaxCB& c = *(const RelaxCB*)io.constants;
    Cursor cur{io.t};
    Sig spec, diff;
    const Tex& gIn_Tiles = *cur.next();
    const Tex& gIn_ViewZ = *cur.next();
    if (SPEC) spec.noisy = cur.next();
    if (DIFF) diff.noisy = cur.next();
    if (SPEC) spec.in = cur.next();
    if (DIFF) diff.in = cur.next();
    if (SPEC) spec.fast = cur.next();
    if (DIFF) diff.fast = cur.next();
    const Tex& gIn_HistoryLength = *cur.next();
    if (SH && SPEC) spec.inSh = cur.next();
    if (SH && DIFF) diff.inSh = cur.next();
    if (SH && SPEC) spec.fastSh = cur.next();
    if (SH && DIFF) diff.fastSh = cur.next();
    if (SPEC) spec.out = cur.next();
    if (DIFF) diff.out = cur.next();
    if (SPEC) spec.outFast = cur.next();
    if (DIFF) diff.outFast = cur.next();
    Tex& gOut_HistoryLength = *cur.next();
    if (SH && SPEC) spec.outSh = cur.next();
    if (SH && DIFF) diff.outSh = cur.next();
    if (SH && SPEC) spec.outFastSh = cur.next();
    if (SH && DIFF) diff.outFastSh = cur.next();

    const int rectW = c.gRectSize[0], rectH = c.gRectSize[1];

#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < rectH; py++)
        for (int px = 0; px < rectW; px++) {
            if (gIn_Tiles.Load(px >> 4, py >> 4).x != 0.0f)
                continue;
            // group-shared data, read at rect-clamped coordinates; NOTE: raw viewZ (no scale, no abs) as in the reference
            auto IsValid = [&](int x, int y) { return Cmp(gIn_ViewZ.Load(clamp(x, 0, rectW - 1), clamp(y, 0, rectH - 1)).x < c.gDenoisingRange); };
            if (IsValid(px, py) == 0.0f)
                continue;

            float historyLength = 255.0f * gIn_HistoryLength.Load(px, py).x;

            struct Moments {
                float3 fastM1 = float3(0.0f), fastM2 = float3(0.0f), noisyM1 = float3(0.0f);
                float noisyM2 = 0.0f;
            } ms, md;
            float sum = 0.0f;
            for (int dx = -2; dx <= 2; dx++)
                for (int dy = -2; dy <= 2; dy++) {
                    int x = clamp(px + dx, 0, rectW - 1), y = clamp(py + dy, 0, rectH - 1);
                    float w = IsValid(x, y);
                    if (w != 0.0f) {
                        auto Accumulate = [&](const Sig& s, Moments& m) {
                            float3 sampleYCoCg = RgbToYCoCg(s.fast->Load(x, y).xyz());
                            m.fastM1 += sampleYCoCg;
                            m.fastM2 += sampleYCoCg * sampleYCoCg;
                            float3 noisy = s.noisy->Load(x, y).xyz();
                            float noisyLuminance = Color::Luminance(noisy);
                            m.noisyM1 += noisy;
                            m.noisyM2 += noisyLuminance * noisyLuminance;
                        };
                        if (SPEC) Accumulate(spec, ms);
                        if (DIFF) Accumulate(diff, md);
                        sum += w;
                    }
                }

            auto Resolve = [&](const Sig& s, Moments& m, bool isSpec) {
                m.fastM1 = Div(m.fastM1, sum);
                m.fastM2 = Div(m.fastM2, sum);
                m.noisyM1 = Div(m.noisyM1, sum);
                m.noisyM2 = Div(m.noisyM2, sum);
                float4 fastCenter = s.fast->Load(px, py);
                float4 fastCenterYCoCg = float4(RgbToYCoCg(fastCenter.xyz()), fastCenter.w);
                ClampOut o = ClampSignal(c, isSpec, m.fastM1, m.fastM2, m.noisyM1, m.noisyM2, fastCenterYCoCg, s.in->Load(px, py), s.noisy->Load(px, py).xyz(), historyLength);
                s.out->Store(px, py, o.slow);
                s.outFast->Store(px, py, o.fast);
                if (SH) {
                    float4 sh = s.inSh->Load(px, py), shFast = s.fastSh->Load(px, py);
                    s.outSh->Store(px, py, lerp(sh, shFast, o.clampingFactor));
                    s.outFastSh->Store(px, py, shFast);
                }
            };
            if (SPEC) Resolve(spec, ms, true);
            if (DIFF) Resolve(diff, md, false);

            gOut_HistoryLength.Store(px, py, DivConst(historyLength, 255.0f));
        }
}

// ================================================================================================ A-trous
// per-pixel confidence-driven parameters shared by both a-trous flavours
struct AtrousParams {
    float specularPhiLIlluminationInv, specularLuminanceWeightRelaxation, specularNormalWeightParamSimplified;
    float2 specularNormalWeightParams, roughnessWeightParams;
    float diffusePhiLIlluminationInv, diffuseLuminanceWeightRelaxation, diffuseNormalWeightParam;
};

template <bool DIFF, bool SPEC, bool SH>
void AtrousSmem(const PassIO& io) {
    const RelaxCB& c = *(const RelaxCB*)io.constants;
    Cursor cur{io.t};
    Sig spec, diff;
    const Tex& gIn_Tiles = *cur.next();
    if (SPEC) spec.in = cur.next();
    if (DIFF) diff.in = cur.next();
    const Tex& gIn_HistoryLength = *cur.next();
    const Tex* gIn_SpecReprojectionConfidence = SPEC ? cur.next() : nullptr;
    const Tex& gIn_Normal_Roughness = *cur.next();
    const Tex& gIn_ViewZ = *cur.next();
    if (SPEC) spec.confidence = cur.next();
    if (DIFF) diff.confidence = cur.next();
    if (SH && SPEC) spec.inSh = cur.next();
    if (SH && DIFF) diff.inSh = cur.next();
    if (SPEC) spec.out = cur.next();
    if (DIFF) diff.out = cur.next();
    Tex& gOut_NormalRoughness = *cur.next();
    Tex& gOut_MaterialID = *cur.next();
    Tex& gOut_ViewZ = *cur.next();
    if (SH && SPEC) spec.outSh = cur.next();
    if (SH && DIFF) diff.outSh = cur.next();

    const int rectW = c.gRectSize[0], rectH = c.gRectSize[1];
    const int ox = (int)c.gRectOrigin[0], oy = (int)c.gRectOrigin[1];
    const int gridW = (rectW + 7) / 8 * 8, gridH = (rectH + 7) / 8 * 8; // every launched thread stores the "previous frame" guides

#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < gridH; py++)
        for (int px = 0; px < gridW; px++) {
            float isSky = gIn_Tiles.Load(px >> 4, py >> 4).x;

            // group-shared tile entries, at rect-clamped coordinates
            auto Cx = [&](int x) { return clamp(x, 0, rectW - 1); };
            auto Cy = [&](int y) { return clamp(y, 0, rectH - 1); };
            auto SNormalRoughness = [&](int x, int y, float& materialID) { return NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.Load(Cx(x), Cy(y)), materialID); };
            auto SWorldPos = [&](int x, int y) { return GetCurrentWorldPosFromPixelPos(c, Cx(x), Cy(y), UnpackViewZ(c, gIn_ViewZ.Load(Cx(x), Cy(y)).x)); };

            float viewZpacked = gIn_ViewZ.Load(px, py).x;
            gOut_ViewZ.Store(px, py, viewZpacked);

            float centerMaterialID = 0.0f;
            float4 normalRoughness = float4(0.0f); // all-sky tile: the shader reads unwritten group-shared memory; defined as zero here
            if (isSky == 0.0f)
                normalRoughness = SNormalRoughness(px, py, centerMaterialID);
            float centerViewZ = UnpackViewZ(c, viewZpacked);
            if (centerViewZ > c.gDenoisingRange)
                normalRoughness = float4(1.0f / 255.0f);
            gOut_NormalRoughness.Store(px, py, PackPrevNormalRoughness(normalRoughness));
            if (NRD_NORMAL_ENCODING == 2) // RELAX_AtrousSmem.hlsli:139-141: only the R10G10B10A2 encoding carries material IDs
                gOut_MaterialID.Store(px, py, DivConst(centerMaterialID, 255.0f));

            if (isSky != 0.0f || px >= rectW || py >= rectH)
                continue;
            if (centerViewZ > c.gDenoisingRange)
                continue;

            float3 centerWorldPos = SWorldPos(px, py);
            float3 centerNormal = normalRoughness.xyz();
            float centerRoughness = normalRoughness.w;
            float historyLength = 255.0f * gIn_HistoryLength.Load(px, py).x;

            auto S = [&](const Sig& s, int x, int y) { return s.in->Load(Cx(x), Cy(y)); };
            auto SSh = [&](const Sig& s, int x, int y) { return s.inSh->Load(Cx(x), Cy(y)); };

            if (historyLength >= c.gHistoryThreshold) {
                // 3x3 gaussian-filtered variance
                static const float kernel[2][2] = {{1.0f / 4.0f, 1.0f / 8.0f}, {1.0f / 8.0f, 1.0f / 16.0f}};
                float4 specularSum(0.0f), diffuseSum(0.0f);
                for (int dx = -1; dx <= 1; dx++)
                    for (int dy = -1; dy <= 1; dy++) {
                        float k = kernel[dx < 0 ? -dx : dx][dy < 0 ? -dy : dy];
                        if (SPEC) specularSum += S(spec, px + dx, py + dy) * k;
                        if (DIFF) diffuseSum += S(diff, px + dx, py + dy) * k;
                    }
                float specular1stMomentV = Color::Luminance(specularSum.xyz());
                float centerSpecularVar = max(0.0f, specularSum.w - specular1stMomentV * specular1stMomentV);
                float diffuse1stMomentV = Color::Luminance(diffuseSum.xyz());
                float centerDiffuseVar = max(0.0f, diffuseSum.w - diffuse1stMomentV * diffuse1stMomentV);

                float diffuseLobeAngleFraction = c.gLobeAngleFraction;

                float centerSpecularLuminance = 0.0f, specularPhiLIlluminationInv = 0.0f, specularLuminanceWeightRelaxation = 0.0f, specularNormalWeightParamSimplified = 0.0f;
                float2 roughnessWeightParams(0.0f), specularNormalWeightParams(0.0f);
                float roughnessModified = 0.0f;
                float3 centerV(0.0f);
                if (SPEC) {
                    centerSpecularLuminance = Color::Luminance(S(spec, px, py).xyz());
                    specularPhiLIlluminationInv = Rcp(max(1.0e-4f, c.gSpecPhiLuminance * HwSqrt(centerSpecularVar)));
                    roughnessWeightParams = GetRoughnessWeightParams(centerRoughness, c.gRoughnessFraction);
                    float diffuseLobeAngleFractionForSimplifiedSpecularNormalWeight = diffuseLobeAngleFraction;
                    float specularLobeAngleFraction = c.gLobeAngleFraction;
                    float specularReprojectionConfidence = gIn_SpecReprojectionConfidence->Load(px, py).x;
                    specularLuminanceWeightRelaxation = lerp(1.0f, specularReprojectionConfidence, c.gLuminanceEdgeStoppingRelaxation);
                    if (c.gHasHistoryConfidence) {
                        float specConfidenceDrivenRelaxation = saturate(c.gConfidenceDrivenRelaxationMultiplier * (1.0f - spec.confidence->Load(ox + px, oy + py).x));
                        float r = saturate(specConfidenceDrivenRelaxation * c.gConfidenceDrivenNormalEdgeStoppingRelaxation);
                        diffuseLobeAngleFractionForSimplifiedSpecularNormalWeight = lerp(diffuseLobeAngleFraction, 1.0f, r);
                        specularLobeAngleFraction = lerp(specularLobeAngleFraction, 1.0f, r);
                        r = saturate(specConfidenceDrivenRelaxation * c.gConfidenceDrivenLuminanceEdgeStoppingRelaxation);
                        specularLuminanceWeightRelaxation *= 1.0f - r;
                    }
                    specularNormalWeightParamSimplified = GetNormalWeightParam2(1.0f, diffuseLobeAngleFractionForSimplifiedSpecularNormalWeight);
                    specularNormalWeightParams = GetNormalWeightParams_ATrous(centerRoughness, historyLength, specularReprojectionConfidence, c.gNormalEdgeStoppingRelaxation,
                        specularLobeAngleFraction, c.gSpecLobeAngleSlack);
                    if (SH)
                        roughnessModified = SSh(spec, px, py).w;
                    centerV = -normalize(centerWorldPos);
                }

                float centerDiffuseLuminance = 0.0f, diffusePhiLIlluminationInv = 0.0f, diffuseLuminanceWeightRelaxation = 1.0f, diffuseNormalWeightParam = 0.0f;
                if (DIFF) {
                    centerDiffuseLuminance = Color::Luminance(S(diff, px, py).xyz());
                    diffusePhiLIlluminationInv = Rcp(max(1.0e-4f, c.gDiffPhiLuminance * HwSqrt(centerDiffuseVar)));
                    if (c.gHasHistoryConfidence) {
                        float diffConfidenceDrivenRelaxation = saturate(c.gConfidenceDrivenRelaxationMultiplier * (1.0f - diff.confidence->Load(ox + px, oy + py).x));
                        float r = saturate(diffConfidenceDrivenRelaxation * c.gConfidenceDrivenNormalEdgeStoppingRelaxation);
                        diffuseLobeAngleFraction = lerp(diffuseLobeAngleFraction, 1.0f, r);
                        r = saturate(diffConfidenceDrivenRelaxation * c.gConfidenceDrivenLuminanceEdgeStoppingRelaxation);
                        diffuseLuminanceWeightRelaxation = 1.0f - r;
                    }
                    diffuseNormalWeightParam = GetNormalWeightParam2(1.0f, diffuseLobeAngleFraction);
                }

                float sumWSpecular = 0.0f, sumWDiffuse = 0.0f;
                float4 sumSpecular(0.0f), sumSpecularSH(0.0f), sumDiffuse(0.0f), sumDiffuseSH(0.0f);
                static const float kernelWeightGaussian3x3[2] = {0.44198f, 0.27901f};
                float depthThreshold = c.gDepthThreshold * (c.gOrthoMode == 0.0f ? centerViewZ : 1.0f);

                for (int cx = -1; cx <= 1; cx++)
                    for (int cy = -1; cy <= 1; cy++) {
                        int qx = px + cx, qy = py + cy;
                        bool isCenter = cx == 0 && cy == 0;
                        bool isInside = qx >= 0 && qy >= 0 && qx < rectW && qy < rectH;
                        float kernelW = isInside ? kernelWeightGaussian3x3[cx < 0 ? -cx : cx] * kernelWeightGaussian3x3[cy < 0 ? -cy : cy] : 0.0f;

                        float sampleMaterialID;
                        float4 sampleNormalRoughness = SNormalRoughness(qx, qy, sampleMaterialID);
                        float3 sampleNormal = sampleNormalRoughness.xyz();
                        float sampleRoughness = sampleNormalRoughness.w;
                        float3 sampleWorldPos = SWorldPos(qx, qy);

                        float geometryW = GetPlaneDistanceWeight_Atrous(centerWorldPos, centerNormal, sampleWorldPos, depthThreshold);
                        geometryW *= kernelW;

                        if (SPEC) {
                            float angles = Math::AcosApprox(dot(centerNormal, sampleNormal));
                            float3 sampleV = -normalize(sampleWorldPos + c.gRoughnessEdgeStoppingRelaxation * centerWorldPos);
                            float normalWSpecularSimplified = ComputeWeight(angles, specularNormalWeightParamSimplified, 0.0f);
                            float normalWSpecular = GetSpecularNormalWeight_ATrous(specularNormalWeightParams, centerNormal, sampleNormal, centerV, sampleV);
                            float roughnessWSpecular = ComputeWeight(sampleRoughness, roughnessWeightParams.x, roughnessWeightParams.y);

                            float4 sampleSpecular = S(spec, qx, qy);
                            float sampleSpecularLuminance = Color::Luminance(sampleSpecular.xyz());
                            float specularLuminanceW = fabsf(centerSpecularLuminance - sampleSpecularLuminance) * specularPhiLIlluminationInv;
                            specularLuminanceW = min(c.gSpecMaxLuminanceRelativeDifference, specularLuminanceW);
                            specularLuminanceW *= specularLuminanceWeightRelaxation;

                            float wSpecular = geometryW * ExpNegAbs(specularLuminanceW);
                            wSpecular *= c.gRoughnessEdgeStoppingEnabled ? (normalWSpecular * roughnessWSpecular) : normalWSpecularSimplified;
                            wSpecular = isCenter ? kernelW : wSpecular;
                            wSpecular *= Cmp(CompareMaterials(sampleMaterialID, centerMaterialID, c.gSpecMinMaterial));

                            sumWSpecular += wSpecular;
                            sumSpecular = Mad(sampleSpecular, wSpecular, sumSpecular);
                            if (SH)
                                sumSpecularSH = Mad(SSh(spec, qx, qy), wSpecular, sumSpecularSH);
                        }
                        if (DIFF) {
                            float angled = Math::AcosApprox(dot(centerNormal, sampleNormal));
                            float normalWDiffuse = ComputeWeight(angled, diffuseNormalWeightParam, 0.0f);

                            float4 sampleDiffuse = S(diff, qx, qy);
                            float sampleDiffuseLuminance = Color::Luminance(sampleDiffuse.xyz());
                            float diffuseLuminanceW = fabsf(centerDiffuseLuminance - sampleDiffuseLuminance) * diffusePhiLIlluminationInv;
                            diffuseLuminanceW = min(c.gDiffMaxLuminanceRelativeDifference, diffuseLuminanceW);
                            diffuseLuminanceW *= diffuseLuminanceWeightRelaxation;

                            float wDiffuse = geometryW * normalWDiffuse * ExpNegAbs(diffuseLuminanceW);
                            wDiffuse = isCenter ? kernelW : wDiffuse;
                            wDiffuse *= Cmp(CompareMaterials(sampleMaterialID, centerMaterialID, c.gDiffMinMaterial));

                            sumWDiffuse += wDiffuse;
                            sumDiffuse = Mad(sampleDiffuse, wDiffuse, sumDiffuse);
                            if (SH)
                                sumDiffuseSH = Mad(SSh(diff, qx, qy), wDiffuse, sumDiffuseSH);
                        }
                    }

                if (SPEC) {
                    sumWSpecular = max(sumWSpecular, 1e-6f);
                    sumSpecular = Div(sumSpecular, sumWSpecular);
                    float m1 = Color::Luminance(sumSpecular.xyz());
                    float variance = max(0.0f, sumSpecular.w - m1 * m1);
                    spec.out->Store(px, py, float4(sumSpecular.xyz(), variance));
                    if (SH)
                        spec.outSh->Store(px, py, float4(Div(sumSpecularSH.xyz(), sumWSpecular), roughnessModified));
                }
                if (DIFF) {
                    sumWDiffuse = max(sumWDiffuse, 1e-6f);
                    sumDiffuse = Div(sumDiffuse, sumWDiffuse);
                    float m1 = Color::Luminance(sumDiffuse.xyz());
                    float variance = max(0.0f, sumDiffuse.w - m1 * m1);
                    diff.out->Store(px, py, float4(sumDiffuse.xyz(), variance));
                    if (SH)
                        diff.outSh->Store(px, py, Div(sumDiffuseSH, sumWDiffuse));
                }
            } else {
                // spatial variance estimation over 5x5
                float sumWSpecular = 0.0f, sumSpecular1stMoment = 0.0f, sumSpecular2ndMoment = 0.0f;
                float3 sumSpecularIllumination(0.0f);
                float4 sumSpecularSH(0.0f);
                float sumWDiffuse = 0.0f, sumDiffuse1stMoment = 0.0f, sumDiffuse2ndMoment = 0.0f;
                float3 sumDiffuseIllumination(0.0f);
                float4 sumDiffuseSH(0.0f);

                float diffuseNormalWeightParam = GetNormalWeightParam2(1.0f, c.gLobeAngleFraction);

                for (int cx = -2; cx <= 2; cx++)
                    for (int cy = -2; cy <= 2; cy++) {
                        int qx = px + cx, qy = py + cy;
                        float sampleMaterialID;
                        float3 sampleNormal = SNormalRoughness(qx, qy, sampleMaterialID).xyz();

                        float depthW = 1.0f;
                        float angle = Math::AcosApprox(dot(centerNormal, sampleNormal));
                        float normalW = ComputeWeight(angle, diffuseNormalWeightParam, 0.0f);

                        if (SPEC) {
                            float4 sampleSpecular = S(spec, qx, qy);
                            float sample1stMoment = Color::Luminance(sampleSpecular.xyz());
                            float specularW = normalW * depthW;
                            specularW *= Cmp(CompareMaterials(sampleMaterialID, centerMaterialID, c.gSpecMinMaterial));
                            sumWSpecular += specularW;
                            sumSpecularIllumination = Mad(sampleSpecular.xyz(), specularW, sumSpecularIllumination);
                            sumSpecular1stMoment += sample1stMoment * specularW;
                            sumSpecular2ndMoment += sampleSpecular.w * specularW;
                            if (SH)
                                sumSpecularSH = Mad(SSh(spec, qx, qy), specularW, sumSpecularSH);
                        }
                        if (DIFF) {
                            float4 sampleDiffuse = S(diff, qx, qy);
                            float sample1stMoment = Color::Luminance(sampleDiffuse.xyz());
                            float diffuseW = normalW * depthW;
                            diffuseW *= Cmp(CompareMaterials(sampleMaterialID, centerMaterialID, c.gDiffMinMaterial));
                            sumWDiffuse += diffuseW;
                            sumDiffuseIllumination = Mad(sampleDiffuse.xyz(), diffuseW, sumDiffuseIllumination);
                            sumDiffuse1stMoment += sample1stMoment * diffuseW;
                            sumDiffuse2ndMoment += sampleDiffuse.w * diffuseW;
                            if (SH)
                                sumDiffuseSH = Mad(SSh(diff, qx, qy), diffuseW, sumDiffuseSH);
                        }
                    }

                float boost = max(1.0f, Div(4.0f, historyLength + 1.0f));
                if (SPEC) {
                    sumWSpecular = max(sumWSpecular, 1e-6f);
                    sumSpecularIllumination = Div(sumSpecularIllumination, sumWSpecular);
                    sumSpecular1stMoment = Div(sumSpecular1stMoment, sumWSpecular);
                    sumSpecular2ndMoment = Div(sumSpecular2ndMoment, sumWSpecular);
                    float variance = max(0.0f, sumSpecular2ndMoment - sumSpecular1stMoment * sumSpecular1stMoment);
                    variance *= boost;
                    spec.out->Store(px, py, float4(sumSpecularIllumination, variance));
                    if (SH) {
                        float roughnessModified = SSh(spec, px, py).w;
                        spec.outSh->Store(px, py, float4(Div(sumSpecularSH.xyz(), sumWSpecular), roughnessModified));
                    }
                }
                if (DIFF) {
                    sumWDiffuse = max(sumWDiffuse, 1e-6f);
                    sumDiffuseIllumination = Div(sumDiffuseIllumination, sumWDiffuse);
                    sumDiffuse1stMoment = Div(sumDiffuse1stMoment, sumWDiffuse);
                    sumDiffuse2ndMoment = Div(sumDiffuse2ndMoment, sumWDiffuse);
                    float variance = max(0.0f, sumDiffuse2ndMoment - sumDiffuse1stMoment * sumDiffuse1stMoment);
                    variance *= boost;
                    diff.out->Store(px, py, float4(sumDiffuseIllumination, variance));
                    if (SH)
                        diff.outSh->Store(px, py, Div(sumDiffuseSH, sumWDiffuse));
                }
            }
        }
}

template <bool DIFF, bool SPEC, bool SH>
void Atrous(const PassIO& io) {
    const RelaxCB& c = *(const RelaxCB*)io.constants;
    Cursor cur{io.t};
    Sig spec, diff;
    const Tex& gIn_Tiles = *cur.next();
    if (SPEC) spec.in = cur.next();
    if (DIFF) diff.in = cur.next();
    const Tex& gIn_HistoryLength = *cur.next();
    const Tex* gIn_SpecReprojectionConfidence = SPEC ? cur.next() : nullptr;
    const Tex& gIn_Normal_Roughness = *cur.next();
    const Tex& gIn_ViewZ = *cur.next();
    if (SPEC) spec.confidence = cur.next();
    if (DIFF) diff.confidence = cur.next();
    if (SH && SPEC) spec.inSh = cur.next();
    if (SH && DIFF) diff.inSh = cur.next();
    if (SPEC) spec.out = cur.next();
    if (DIFF) diff.out = cur.next();
    if (SH && SPEC) spec.outSh = cur.next();
    if (SH && DIFF) diff.outSh = cur.next();

    const int rectW = c.gRectSize[0], rectH = c.gRectSize[1];
    const int ox = (int)c.gRectOrigin[0], oy = (int)c.gRectOrigin[1];
    const int stepSize = (int)c.gStepSize;

#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < rectH; py++)
        for (int px = 0; px < rectW; px++) {
            if (gIn_Tiles.Load(px >> 4, py >> 4).x != 0.0f)
                continue;
            float centerViewZ = UnpackViewZ(c, gIn_ViewZ.Load(px, py).x);
            if (centerViewZ > c.gDenoisingRange)
                continue;

            float centerMaterialID;
            float4 centerNormalRoughness = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.Load(px, py), centerMaterialID);
            float3 centerNormal = centerNormalRoughness.xyz();
            float centerRoughness = centerNormalRoughness.w;
            float historyLength = 255.0f * gIn_HistoryLength.Load(px, py).x;

            float diffuseLobeAngleFraction = Div(c.gLobeAngleFraction, HwSqrt(float(c.gStepSize)));
            if (SH)
                diffuseLobeAngleFraction = Rcp(HwSqrt(float(c.gStepSize)));
            diffuseLobeAngleFraction = lerp(0.99f, diffuseLobeAngleFraction, saturate(DivConst(historyLength, 5.0f)));

            float4 centerSpecular(0.0f), centerSpecularSH(0.0f), sumSpecular(0.0f), sumSpecularSH(0.0f);
            float centerSpecularLuminance = 0.0f, specularPhiLIlluminationInv = 0.0f, specularLuminanceWeightRelaxation = 1.0f, specularNormalWeightParamSimplified = 0.0f;
            float2 roughnessWeightParams(0.0f), specularNormalWeightParams(0.0f);
            float sumWSpecular = 0.44198f * 0.44198f, roughnessModified = 0.0f;
            if (SPEC) {
                centerSpecular = spec.in->Load(px, py);
                centerSpecularLuminance = Color::Luminance(centerSpecular.xyz());
                float centerSpecularVar = centerSpecular.w;
                specularPhiLIlluminationInv = Rcp(max(1.0e-4f, c.gSpecPhiLuminance * HwSqrt(centerSpecularVar)));

                roughnessWeightParams = GetRoughnessWeightParams(centerRoughness, c.gRoughnessFraction);
                float diffuseLobeAngleFractionForSimplifiedSpecularNormalWeight = diffuseLobeAngleFraction;
                float specularLobeAngleFraction = c.gLobeAngleFraction;
                float specularReprojectionConfidence = gIn_SpecReprojectionConfidence->Load(px, py).x;
                if (c.gStepSize <= 4)
                    specularLuminanceWeightRelaxation = lerp(1.0f, specularReprojectionConfidence, c.gLuminanceEdgeStoppingRelaxation);
                if (c.gHasHistoryConfidence) {
                    float specConfidenceDrivenRelaxation = saturate(c.gConfidenceDrivenRelaxationMultiplier * (1.0f - spec.confidence->Load(ox + px, oy + py).x));
                    float r = saturate(specConfidenceDrivenRelaxation * c.gConfidenceDrivenNormalEdgeStoppingRelaxation);
                    diffuseLobeAngleFractionForSimplifiedSpecularNormalWeight = lerp(diffuseLobeAngleFraction, 1.0f, r);
                    specularLobeAngleFraction = lerp(specularLobeAngleFraction, 1.0f, r);
                    r = saturate(specConfidenceDrivenRelaxation * c.gConfidenceDrivenLuminanceEdgeStoppingRelaxation);
                    specularLuminanceWeightRelaxation *= 1.0f - r;
                }
                specularNormalWeightParamSimplified = GetNormalWeightParam2(1.0f, diffuseLobeAngleFractionForSimplifiedSpecularNormalWeight);
                specularNormalWeightParams = GetNormalWeightParams_ATrous(centerRoughness, historyLength, specularReprojectionConfidence, c.gNormalEdgeStoppingRelaxation,
                    specularLobeAngleFraction, c.gSpecLobeAngleSlack);

                sumSpecular = centerSpecular * float4(sumWSpecular, sumWSpecular, sumWSpecular, sumWSpecular * sumWSpecular);
                if (SH) {
                    centerSpecularSH = spec.inSh->Load(px, py);
                    sumSpecularSH = centerSpecularSH * sumWSpecular;
                    roughnessModified = centerSpecularSH.w;
                }
            }

            float4 centerDiffuse(0.0f), sumDiffuse(0.0f), sumDiffuseSH(0.0f);
            float centerDiffuseLuminance = 0.0f, diffusePhiLIlluminationInv = 0.0f, diffuseLuminanceWeightRelaxation = 1.0f, diffuseNormalWeightParam = 0.0f;
            float sumWDiffuse = 0.44198f * 0.44198f;
            if (DIFF) {
                centerDiffuse = diff.in->Load(px, py);
                centerDiffuseLuminance = Color::Luminance(centerDiffuse.xyz());
                float centerDiffuseVar = centerDiffuse.w;
                diffusePhiLIlluminationInv = Rcp(max(1.0e-4f, c.gDiffPhiLuminance * HwSqrt(centerDiffuseVar)));
                if (c.gHasHistoryConfidence) {
                    float diffConfidenceDrivenRelaxation = saturate(c.gConfidenceDrivenRelaxationMultiplier * (1.0f - diff.confidence->Load(ox + px, oy + py).x));
                    float r = saturate(diffConfidenceDrivenRelaxation * c.gConfidenceDrivenNormalEdgeStoppingRelaxation);
                    diffuseLobeAngleFraction = lerp(diffuseLobeAngleFraction, 1.0f, r);
                    r = saturate(diffConfidenceDrivenRelaxation * c.gConfidenceDrivenLuminanceEdgeStoppingRelaxation);
                    diffuseLuminanceWeightRelaxation = 1.0f - r;
                }
                diffuseNormalWeightParam = GetNormalWeightParam2(1.0f, diffuseLobeAngleFraction);
                sumDiffuse = centerDiffuse * float4(sumWDiffuse, sumWDiffuse, sumWDiffuse, sumWDiffuse * sumWDiffuse);
                if (SH)
                    sumDiffuseSH = diff.inSh->Load(px, py) * sumWDiffuse;
            }

            float3 centerWorldPos = GetCurrentWorldPosFromPixelPos(c, px, py, centerViewZ);
            float3 centerV = -normalize(centerWorldPos);
            static const float kernelWeightGaussian3x3[2] = {0.44198f, 0.27901f};
            float depthThreshold = c.gDepthThreshold * (c.gOrthoMode == 0.0f ? centerViewZ : 1.0f);

            // random offsets against ringing at large steps
            int offx = 0, offy = 0;
            if (c.gStepSize > 4) {
                RngHash rng;
                rng.Initialize((uint32_t)px, (uint32_t)py, c.gFrameIndex);
                float2 rnd = rng.GetFloat2();
                offx = (int)(float(c.gStepSize) * 0.5f * (rnd.x - 0.5f));
                offy = (int)(float(c.gStepSize) * 0.5f * (rnd.y - 0.5f));
            }

            for (int yy = -1; yy <= 1; yy++)
                for (int xx = -1; xx <= 1; xx++) {
                    if (xx == 0 && yy == 0)
                        continue;
                    int qx = px + offx + xx * stepSize, qy = py + offy + yy * stepSize;
                    bool isInside = qx >= 0 && qy >= 0 && qx < rectW && qy < rectH;
                    float kernelW = kernelWeightGaussian3x3[xx < 0 ? -xx : xx] * kernelWeightGaussian3x3[yy < 0 ? -yy : yy];

                    float sampleMaterialID;
                    float4 sampleNormalRoughness = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.Load(qx, qy), sampleMaterialID);
                    float3 sampleNormal = sampleNormalRoughness.xyz();
                    float sampleRoughness = sampleNormalRoughness.w;
                    float sampleViewZ = UnpackViewZ(c, gIn_ViewZ.Load(qx, qy).x);
                    float3 sampleWorldPos = GetCurrentWorldPosFromPixelPos(c, qx, qy, sampleViewZ);

                    float geometryW = GetPlaneDistanceWeight_Atrous(centerWorldPos, centerNormal, sampleWorldPos, depthThreshold);
                    geometryW *= kernelW;
                    geometryW *= Cmp(isInside && sampleViewZ < c.gDenoisingRange);

                    if (SPEC) {
                        float3 sampleV = -normalize(sampleWorldPos + c.gRoughnessEdgeStoppingRelaxation * centerWorldPos);
                        float angles = Math::AcosApprox(dot(centerNormal, sampleNormal));
                        float normalWSpecularSimplified = ComputeWeight(angles, specularNormalWeightParamSimplified, 0.0f);
                        float normalWSpecular = GetSpecularNormalWeight_ATrous(specularNormalWeightParams, centerNormal, sampleNormal, centerV, sampleV);
                        float roughnessWSpecular = ComputeWeight(sampleRoughness, roughnessWeightParams.x, roughnessWeightParams.y);

                        float wSpecular = geometryW * (c.gRoughnessEdgeStoppingEnabled ? (normalWSpecular * roughnessWSpecular) : normalWSpecularSimplified);
                        wSpecular *= Cmp(CompareMaterials(sampleMaterialID, centerMaterialID, c.gSpecMinMaterial));
                        if (wSpecular > 1e-4f) {
                            float4 sampleSpecular = spec.in->Load(qx, qy);
                            float sampleSpecularLuminance = Color::Luminance(sampleSpecular.xyz());
                            float specularLuminanceW = fabsf(centerSpecularLuminance - sampleSpecularLuminance) * specularPhiLIlluminationInv;
                            specularLuminanceW = min(c.gSpecMaxLuminanceRelativeDifference, specularLuminanceW);
                            specularLuminanceW *= specularLuminanceWeightRelaxation;
                            wSpecular *= ExpNegAbs(specularLuminanceW);

                            sumWSpecular += wSpecular;
                            sumSpecular = Mad(sampleSpecular, float4(wSpecular, wSpecular, wSpecular, wSpecular * wSpecular), sumSpecular);
                            if (SH)
                                sumSpecularSH = Mad(spec.inSh->Load(qx, qy), wSpecular, sumSpecularSH);
                        }
                    }
                    if (DIFF) {
                        float angled = Math::AcosApprox(dot(centerNormal, sampleNormal));
                        float normalWDiffuse = ComputeWeight(angled, diffuseNormalWeightParam, 0.0f);
                        float wDiffuse = geometryW * normalWDiffuse;
                        wDiffuse *= Cmp(CompareMaterials(sampleMaterialID, centerMaterialID, c.gDiffMinMaterial));
                        if (wDiffuse > 1e-4f) {
                            float4 sampleDiffuse = diff.in->Load(qx, qy);
                            float sampleDiffuseLuminance = Color::Luminance(sampleDiffuse.xyz());
                            float diffuseLuminanceW = fabsf(centerDiffuseLuminance - sampleDiffuseLuminance) * diffusePhiLIlluminationInv;
                            diffuseLuminanceW = min(c.gDiffMaxLuminanceRelativeDifference, diffuseLuminanceW);
                            diffuseLuminanceW *= diffuseLuminanceWeightRelaxation;
                            wDiffuse *= ExpNegAbs(diffuseLuminanceW);

                            sumWDiffuse += wDiffuse;
                            sumDiffuse = Mad(sampleDiffuse, float4(wDiffuse, wDiffuse, wDiffuse, wDiffuse * wDiffuse), sumDiffuse);
                            if (SH)
                                sumDiffuseSH = Mad(diff.inSh->Load(qx, qy), wDiffuse, sumDiffuseSH);
                        }
                    }
                }

            if (SPEC) {
                float4 filtered = Div(sumSpecular, float4(sumWSpecular, sumWSpecular, sumWSpecular, sumWSpecular * sumWSpecular));
                if (SH) {
                    if (c.gIsLastPass == 1)
                        filtered = float4(_NRD_LinearToYCoCg(filtered.xyz()), filtered.w);
                    spec.outSh->Store(px, py, float4(Div(sumSpecularSH.xyz(), sumWSpecular), roughnessModified));
                }
                spec.out->Store(px, py, filtered);
            }
            if (DIFF) {
                float4 filtered = Div(sumDiffuse, float4(sumWDiffuse, sumWDiffuse, sumWDiffuse, sumWDiffuse * sumWDiffuse));
                if (SH) {
                    if (c.gIsLastPass == 1)
                        filtered = float4(_NRD_LinearToYCoCg(filtered.xyz()), filtered.w);
                    diff.outSh->Store(px, py, Div(sumDiffuseSH, sumWDiffuse));
                }
                diff.out->Store(px, py, filtered);
            }
        }
}

// ================================================================================================ Copy / AntiFirefly
// (only dispatched when RelaxSettings::enableAntiFirefly: history -> user output plane -> RCRS filter -> history)
template <bool DIFF, bool SPEC>
void Copy(const PassIO& io) { // reference Shaders/Include/RELAX_Copy.hlsli:10-24
    const RelaxCB& c = *(const RelaxCB*)io.constants;
    Cursor cur{io.t};
    Sig spec, diff;
    if (SPEC) spec.in = cur.next();
    if (DIFF) diff.in = cur.next();
    if (SPEC) spec.out = cur.next();
    if (DIFF) diff.out = cur.next();
    const int gridW = (c.gRectSize[0] + 7) / 8 * 8, gridH = (c.gRectSize[1] + 7) / 8 * 8;
    for (int py = 0; py < gridH; py++)
        for (int px = 0; px < gridW; px++) {
            if (SPEC) spec.out->Store(px, py, spec.in->Load(px, py));
            if (DIFF) diff.out->Store(px, py, diff.in->Load(px, py));
        }
}

// cross-bilateral rank-conditioned rank-selection over the 3x3 neighbourhood; reference Shaders/Include/RELAX_AntiFirefly.hlsli:10-210
template <bool DIFF, bool SPEC>
void AntiFirefly(const PassIO& io) {
    const RelaxCB& c = *(const RelaxCB*)io.constants;
    Cursor cur{io.t};
    Sig spec, diff;
    const Tex& gIn_Tiles = *cur.next();
    if (SPEC) spec.in = cur.next();
    if (DIFF) diff.in = cur.next();
    const Tex& gIn_Normal_Roughness = *cur.next();
    const Tex& gIn_ViewZ = *cur.next();
    if (SPEC) spec.out = cur.next();
    if (DIFF) diff.out = cur.next();
    const int rectW = c.gRectSize[0], rectH = c.gRectSize[1];

#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < rectH; py++)
        for (int px = 0; px < rectW; px++) {
            if (gIn_Tiles.Load(px >> 4, py >> 4).x != 0.0f)
                continue;
            if (UnpackViewZ(c, gIn_ViewZ.Load(px, py).x) > c.gDenoisingRange)
                continue;
            auto MaterialID = [&](int x, int y) {
                float m;
                NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.Load(x, y), m);
                return m;
            };
            const float centerMaterialID = MaterialID(px, py);
            auto Filter = [&](const Sig& s, float minMaterial) {
                float4 center = s.in->Load(px, py);
                float centerLuminance = Color::Luminance(center.xyz());
                float maxLuminance = -1.0f, minLuminance = 1.0e6f;
                int maxX = px, maxY = py, minX = px, minY = py;
                for (int yy = -1; yy <= 1; yy++)
                    for (int xx = -1; xx <= 1; xx++) {
                        int qx = px + xx, qy = py + yy;
                        if ((xx == 0 && yy == 0) || qx < 0 || qy < 0 || qx >= rectW || qy >= rectH)
                            continue;
                        float luminance = Color::Luminance(s.in->Load(qx, qy).xyz());
                        if (CompareMaterials(MaterialID(qx, qy), centerMaterialID, minMaterial)) {
                            if (luminance > maxLuminance) {
                                maxLuminance = luminance;
                                maxX = qx, maxY = qy;
                            }
                            if (luminance < minLuminance) {
                                minLuminance = luminance;
                                minX = qx, minY = qy;
                            }
                        }
                    }
                int sx = px, sy = py;
                if (centerLuminance > maxLuminance)
                    sx = maxX, sy = maxY;
                if (centerLuminance < minLuminance)
                    sx = minX, sy = minY;
                s.out->Store(px, py, float4(s.in->Load(sx, sy).xyz(), center.w));
            };
            if (SPEC) Filter(spec, c.gSpecMinMaterial);
            if (DIFF) Filter(diff, c.gDiffMinMaterial);
        }
}

// ================================================================================================ SplitScreen
template <bool DIFF, bool SPEC, bool SH>
void SplitScreen(const PassIO& io) {
    const RelaxCB& c = *(const RelaxCB*)io.constants;
    Cursor cur{io.t};
    Sig spec, diff;
    const Tex& gIn_ViewZ = *cur.next();
    if (DIFF) diff.in = cur.next();
    if (SPEC) spec.in = cur.next();
    if (SH && DIFF) diff.inSh = cur.next();
    if (SH && SPEC) spec.inSh = cur.next();
    if (DIFF) diff.out = cur.next();
    if (SPEC) spec.out = cur.next();
    if (SH && DIFF) diff.outSh = cur.next();
    if (SH && SPEC) spec.outSh = cur.next();

    const int rectW = c.gRectSize[0], rectH = c.gRectSize[1];
    for (int py = 0; py < rectH; py++)
        for (int px = 0; px < rectW; px++) {
            float2 pixelUv = float2(float(px) + 0.5f, float(py) + 0.5f) * c.gRectSizeInv;
            if (pixelUv.x > c.gSplitScreen)
                continue;
            float viewZ = UnpackViewZ(c, gIn_ViewZ.Load((int)c.gRectOrigin[0] + px, (int)c.gRectOrigin[1] + py).x);
            float keep = Cmp(viewZ < c.gDenoisingRange);
            auto Pass = [&](const Sig& s, uint32_t checkerboardMode) {
                int cx = px >> (checkerboardMode != 2u ? 1 : 0);
                float4 v = s.in->Load(cx, py);
                if (SH)
                    v = float4(_NRD_LinearToYCoCg(v.xyz()), v.w);
                s.out->Store(px, py, v * keep);
                if (SH)
                    s.outSh->Store(px, py, s.inSh->Load(cx, py) * keep);
            };
            if (DIFF) Pass(diff, c.gDiffCheckerboard);
            if (SPEC) Pass(spec, c.gSpecCheckerboard);
        }
}

} // namespace

#define RELAX_VARIANT(name, D, S, H)                                         \
    {"RELAX_" name "_PrePass.cs", PrePass<D, S, H>},                         \
    {"RELAX_" name "_TemporalAccumulation.cs", TemporalAccumulation<D, S, H>}, \
    {"RELAX_" name "_HistoryFix.cs", HistoryFix<D, S, H>},                   \
    {"RELAX_" name "_HistoryClamping.cs", HistoryClamping<D, S, H>},         \
    {"RELAX_" name "_AtrousSmem.cs", AtrousSmem<D, S, H>},                   \
    {"RELAX_" name "_Atrous.cs", Atrous<D, S, H>},                           \
    {"RELAX_" name "_Copy.cs", Copy<D, S>},                                    \
    {"RELAX_" name "_AntiFirefly.cs", AntiFirefly<D, S>},                      \
    {"RELAX_" name "_SplitScreen.cs", SplitScreen<D, S, H>}

#define RELAX_HITDIST(name, D, S)                                                          \
    {"RELAX_" name "_HitDistReconstruction.cs", HitDistReconstruction<D, S, 1>},           \
    {"RELAX_" name "_HitDistReconstruction_5x5.cs", HitDistReconstruction<D, S, 2>}

// ================================================================================================ Validation
// reference Shaders/Source/RELAX_Validation.cs.hlsl:32-207 without the text overlay (see the REBLUR twin in reblur_passes.cpp)
static void RelaxValidation(const PassIO& io) {
    const RelaxCB& c = *(const RelaxCB*)io.constants;
    const Tex &gIn_Normal_Roughness = io.t[0], &gIn_ViewZ = io.t[1], &gIn_Mv = io.t[2], &gIn_HistoryLength = io.t[3];
    Tex& gOut_Validation = io.t[4];
    const float VIEWPORT_SIZE = 0.25f;
#pragma omp parallel for schedule(static)
    for (int py = 0; py < gOut_Validation.H(); py++)
        for (int px = 0; px < gOut_Validation.W(); px++) {
            if (c.gResetHistory != 0) {
                gOut_Validation.Store(px, py, float4(0.0f));
                continue;
            }
            float2 pixelUv = Div(float2(float(px) + 0.5f, float(py) + 0.5f), c.gResourceSize);
            float2 scaled = pixelUv * 4.0f; // / VIEWPORT_SIZE
            float2 viewportId = floor(scaled);
            float2 viewportUv = scaled - viewportId;
            float viewportIndex = viewportId.y * 4.0f + viewportId.x; // / VIEWPORT_SIZE
            float2 viewportUvScaled = viewportUv * c.gResolutionScale;

            float4 normalAndRoughness = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.SampleNearest(viewportUvScaled + c.gRectOffset));
            float viewZ = UnpackViewZ(c, gIn_ViewZ.SampleNearest(viewportUvScaled + c.gRectOffset).x);
            float4 mvRaw = gIn_Mv.SampleNearest(viewportUvScaled + c.gRectOffset);
            float3 mv = float3(mvRaw.x * c.gMvScale.x, mvRaw.y * c.gMvScale.y, mvRaw.z * c.gMvScale.z);
            float historyLength = 255.0f * gIn_HistoryLength.SampleNearest(viewportUvScaled).x - 1.0f;

            float3 N = normalAndRoughness.xyz();
            float3 X = GetCurrentWorldPosFromClipSpaceXY(c, viewportUv * 2.0f - 1.0f, abs(viewZ));
            bool isInf = abs(viewZ) > c.gDenoisingRange;
            bool checkerboard = Sequence::CheckerBoard((uint32_t)px >> 2, (uint32_t)py >> 2, 0) != 0;
            float notInf = isInf ? 0.0f : 1.0f;

            float4 result = gOut_Validation.Load(px, py);
            if (viewportIndex == 0.0f) {
                result = float4(N * 0.5f + 0.5f, 1.0f);
            } else if (viewportIndex == 1.0f) {
                result = float4(float3(normalAndRoughness.w), 1.0f);
            } else if (viewportIndex == 2.0f) {
                float f = Div(0.1f * abs(viewZ), 1.0f + 0.1f * abs(viewZ));
                float3 color = viewZ < 0.0f ? float3(0, 0, 1) : float3(0, 1, 0);
                result = float4(isInf ? float3(1, 0, 0) : color * f, 1.0f);
            } else if (viewportIndex == 3.0f) {
                float2 viewportUvPrevExpected = Geometry::GetScreenUv(c.gWorldToClipPrev, X);
                float2 viewportUvPrev = viewportUv + float2(mv.x, mv.y);
                if (c.gMvScale.w != 0.0f)
                    viewportUvPrev = Geometry::GetScreenUv(c.gWorldToClipPrev, X + mv);
                float2 uvDelta = (viewportUvPrev - viewportUvPrevExpected) * float2(float(c.gRectSize[0]), float(c.gRectSize[1]));
                result = float4(IsInScreenNearest(viewportUvPrev) != 0.0f ? float3(abs(uvDelta.x), abs(uvDelta.y), 0.0f) : float3(0, 0, 1), 1.0f);
            } else if (viewportIndex == 4.0f) {
                float2 dim = float2(Div(0.5f * c.gResourceSize.y, c.gResourceSize.x), 0.5f);
                float2 remappedUv = Div(viewportUv - (1.0f - dim), dim);
                if (remappedUv.x > 0.0f && remappedUv.y > 0.0f) {
                    float2 dimInPixels = c.gResourceSize * VIEWPORT_SIZE * dim;
                    float2 uv = c.gJitter + 0.5f;
                    float2 su = saturate(uv);
                    bool isValid = su.x == uv.x && su.y == uv.y;
                    int ax = (int)(su.x * dimInPixels.x), ay = (int)(su.y * dimInPixels.y);
                    int bx = (int)(remappedUv.x * dimInPixels.x), by = (int)(remappedUv.y * dimInPixels.y);
                    int dx = ax - bx < 0 ? bx - ax : ax - bx, dy = ay - by < 0 ? by - ay : ay - by;
                    if (dx <= 1 && dy <= 1 && isValid)
                        result.x = result.y = result.z = 0.66f;
                    if (dx <= 3 && dy <= 3 && !isValid)
                        result.x = 1.0f, result.y = 0.0f, result.z = 0.0f;
                } else {
                    float roundingErrorCorrection = abs(viewZ) * 0.001f;
                    float3 v = X + roundingErrorCorrection;
                    result.x = frac(v.x) * notInf, result.y = frac(v.y) * notInf, result.z = frac(v.z) * notInf;
                }
                result.w = 1.0f;
            } else if (viewportIndex == 8.0f) {
                float f = 1.0f - saturate(Div(historyLength, max(max(c.gDiffMaxAccumulatedFrameNum, c.gSpecMaxAccumulatedFrameNum), 1.0f)));
                f = checkerboard && historyLength < 2.0f ? 0.75f : f;
                result = float4(Sequence::ColorizeZucconi(viewportUv.y > 0.95f ? 1.0f - viewportUv.x : f * notInf), 1.0f);
            }
            gOut_Validation.Store(px, py, result);
        }
}

const PassEntry* GetRelaxPasses(uint32_t& n) {
    static const PassEntry k[] = {
        {"RELAX_ClassifyTiles.cs", ClassifyTiles},
        {"RELAX_Validation.cs", RelaxValidation},
        RELAX_HITDIST("Diffuse", true, false),
        RELAX_HITDIST("Specular", false, true),
        RELAX_HITDIST("DiffuseSpecular", true, true),
        RELAX_VARIANT("Diffuse", true, false, false),
        RELAX_VARIANT("DiffuseSh", true, false, true),
        RELAX_VARIANT("Specular", false, true, false),
        RELAX_VARIANT("SpecularSh", false, true, true),
        RELAX_VARIANT("DiffuseSpecular", true, true, false),
        RELAX_VARIANT("DiffuseSpecularSh", true, true, true),
    };
    n = sizeof(k) / sizeof(k[0]);
    return k;
}

} // namespace orc
