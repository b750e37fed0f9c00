// ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/hlsl.h).
//
// CPU restatement of the REBLUR pass chain, all ten denoisers of the family from one set of templates: radiance + hit distance (REBLUR_DIFFUSE / _SPECULAR /
// _DIFFUSE_SPECULAR), their _SH variants, the _OCCLUSION variants and _DIFFUSE_DIRECTIONAL_OCCLUSION; quality and performance mode, checkerboard modes,
// hit-distance reconstruction 3x3 / 5x5, anti-firefly, with and without temporal stabilisation, and the optional inputs (history confidence, disocclusion
// threshold mix, base colour / metalness for the specular motion-vector patch). One function per reference shader; every pixel is independent inside a
// pass, so the LDS preloads of the shaders become clamped plane reads here. Pinned pass by pass against the reference's own shader text: tests/test_ref_parity.py.
//   ClassifyTiles           reference Shaders/Source/REBLUR_ClassifyTiles.cs.hlsl:20-55
//   PrePass                 reference Shaders/Include/REBLUR_PrePass.hlsli:11-108
//   spatial filters         reference Shaders/Include/REBLUR_Common_DiffuseSpatialFilter.hlsli:22-213,
//                                     REBLUR_Common_SpecularSpatialFilter.hlsli:22-277
//   TemporalAccumulation    reference Shaders/Include/REBLUR_TemporalAccumulation.hlsli:11-931
//   HistoryFix              reference Shaders/Include/REBLUR_HistoryFix.hlsli:11-463
//   Blur / PostBlur         reference Shaders/Include/REBLUR_Blur.hlsli:11-74, REBLUR_PostBlur.hlsli:11-78
//   TemporalStabilization   reference Shaders/Include/REBLUR_TemporalStabilization.hlsli:11-367
//   SplitScreen             reference Shaders/Include/REBLUR_SplitScreen.hlsli:11-46
// Binding order of planes = reference Source/Denoisers/Reblur_{Diffuse,Specular,DiffuseSpecular}.hpp.
#include "passes.h"
#include "reblur_common.h"

#include <cstdio>
#include <cstdlib>

namespace orc {

namespace {

struct Cursor { // walks io.t in binding order
    const PassIO& io;
    uint32_t k = 0;
    explicit Cursor(const PassIO& i) : io(i) {}
    Tex* next() { return &io.t[k++]; }
    Tex* nextIf(bool cond) { return cond ? &io.t[k++] : nullptr; }
};

// ================================================================================================ ClassifyTiles
void ClassifyTiles(const PassIO& io) {
    const ReblurCB& c = *(const ReblurCB*)io.constants;
    const Tex& gIn_ViewZ = io.t[0];
    Tex& gOut_Tiles = io.t[1];
    const int tilesW = ((int)c.gRectSize.x + 15) / 16, tilesH = ((int)c.gRectSize.y + 15) / 16; // one group per 16x16 tile of the RECT (dynamic resolution)
#pragma omp parallel for schedule(static)
    for (int ty = 0; ty < tilesH; ty++)
        for (int tx = 0; tx < tilesW; tx++) {
            int sum = 0;
            for (int j = 0; j < 16; j++)
                for (int i = 0; i < 16; i++) {
                    float viewZ = UnpackViewZ(c, gIn_ViewZ.Load(tx * 16 + i, ty * 16 + j).x);
                    sum += viewZ > c.gDenoisingRange ? 1 : 0;
                }
            gOut_Tiles.Store(tx, ty, sum == 256 ? 1.0f : 0.0f);
        }
}

// ================================================================================================ spatial filters
struct SpatialCtx { // what PrePass / Blur / PostBlur share per pixel
    int px, py;
    float2 pixelUv;
    float viewZ, roughness, materialID, NoV, frustumSize;
    float3 N, Nv, Xv, Vv;
    float4 rotator;
    float2 data1; // accumulated frames (diff, spec); unused by the pre-pass
    // Plane distance of a tap: the shaders evaluate dot(Nv, Xv(tap)) with Xv = ReconstructViewPosition(uv, frustum, z) (REBLUR_Common_*SpatialFilter.hlsli
    // "ComputeWeight( dot( Nv, Xvs ), ... )"). Every tap position is a pixel centre, uv = (k + 0.5) * rectSizeInv with integer k, and the perspective
    // Xv is linear in uv, so the product is expanded once per PIXEL: dot(Nv, Xv((k + 0.5) * rectSizeInv, z)) = z * (k.x * geo.x + k.y * geo.y + geo.z).
    // This expanded form is the specification of both sides (3 operations per tap instead of 13); orthographic projection is rejected up front.
    float3 geo;
    bool perf;    // REBLUR_PERFORMANCE_MODE (REBLUR_Config.hlsli:196-238): 6 taps of g_Special6, screen-space sampling for specular too
    // checkerboard resolve of the pre-pass (REBLUR_PrePass.hlsli:43-56): left / right neighbour columns in the half-width input and their weights
    int cbX0 = 0, cbX1 = 0;
    float2 wc = float2(0.0f);
};

// Common.hlsli:297-307: a tap that lands on a pixel without data moves one pixel left / right (alternating with the tap counter); "pos" is a pixel centre
inline float2 ApplyCheckerboardShift(float2 pos, uint32_t mode, uint32_t counter, uint32_t frameIndex) {
    float2 posPositive = pos + 16384.0f;
    uint32_t checkerboard = Sequence::CheckerBoard((uint32_t)posPositive.x, (uint32_t)posPositive.y, frameIndex);
    float shift = (counter & 1u) == 0 ? -1.0f : 1.0f;
    pos.x += shift * ((checkerboard != mode && mode != 2) ? 1.0f : 0.0f);
    return pos;
}

template <typename S> // S = REBLUR_TYPE: float4 (radiance + hit distance) or float (occlusion: hit distance only)
S DiffuseSpatialFilterTaps(const ReblurCB& c, SpatialMode mode, const SpatialCtx& s, S diff, const Tex& gIn_Diff, const Tex& gIn_ViewZ, const Tex& gIn_Normal_Roughness,
    float4* diffSh, const Tex* gIn_DiffSh, float& sum) { // REBLUR_SH: the SH1 plane is filtered with the same weights (all 4 components)
    constexpr int KIND = SignalKind<S>::value;
    constexpr bool OCC = KIND == SIGNAL_OCCLUSION;
    typedef ReblurSignal<KIND> Sig;

    float fractionScale = 1.0f, radiusScale = 1.0f;
    if (mode == PRE_BLUR)
        fractionScale = REBLUR_PRE_BLUR_FRACTION_SCALE;
    else if (mode == BLUR)
        fractionScale = REBLUR_BLUR_FRACTION_SCALE;
    else {
        radiusScale = REBLUR_POST_BLUR_RADIUS_SCALE;
        fractionScale = REBLUR_POST_BLUR_FRACTION_SCALE;
    }

    // Hit distance factor
    float hitDistScale = _REBLUR_GetHitDistanceNormalization(s.viewZ, c.gHitDistParams, 1.0f);
    float hitDist = ExtractHitDist(diff) * hitDistScale;
    float hitDistFactor = GetHitDistFactor(hitDist, s.frustumSize);

    // Blur radius
    float diffNonLinearAccumSpeed, blurRadius, areaFactor;
    if (mode == PRE_BLUR) {
        diffNonLinearAccumSpeed = REBLUR_PRE_BLUR_NON_LINEAR_ACCUM_SPEED;
        blurRadius = c.gDiffPrepassBlurRadius;
        areaFactor = hitDistFactor;
    } else {
        float boost = 1.0f - GetFadeBasedOnAccumulatedFrames(c, s.data1.x);
        boost *= 1.0f - BRDF::Pow5(s.NoV);
        diffNonLinearAccumSpeed = Rcp(1.0f + REBLUR_SAMPLES_PER_FRAME * (1.0f - boost) * s.data1.x);
        blurRadius = c.gMaxBlurRadius;
        areaFactor = hitDistFactor * diffNonLinearAccumSpeed;
    }
    blurRadius *= Math::Sqrt01(areaFactor);
    blurRadius *= radiusScale;
    blurRadius = max(blurRadius, c.gMinBlurRadius);

    // Weights
    float2 geometryWeightParams = GetGeometryWeightParams(c.gPlaneDistSensitivity, s.frustumSize, s.Xv, s.Nv);
    float normalWeightParam = Div(GetNormalWeightParam(diffNonLinearAccumSpeed, c.gLobeAngleFraction), fractionScale);
    float2 hitDistanceWeightParams = GetHitDistanceWeightParams(ExtractHitDist(diff), diffNonLinearAccumSpeed);
    float minHitDistWeight = c.gMinHitDistanceWeight * fractionScale;
    if (mode != PRE_BLUR && !OCC) // REBLUR_Common_DiffuseSpatialFilter.hlsli:76
        minHitDistWeight *= HwSqrt(diffNonLinearAccumSpeed);

    // Screen-space sampling (REBLUR_USE_SCREEN_SPACE_SAMPLING_FOR_DIFFUSE = 1)
    float2 skew = float2(1.0f);
    if (mode != PRE_BLUR) {
        skew = lerp(float2(1.0f - fabsf(s.Nv.x), 1.0f - fabsf(s.Nv.y)), float2(1.0f), s.NoV);
        skew = Div(skew, max(skew.x, skew.y));
    }
    skew *= c.gRectSizeInv * blurRadius;
    float4 scaledRotator = Geometry::ScaleRotator(s.rotator, skew);

    const int sampleNum = s.perf ? 6 : 8;
    for (int n = 0; n < sampleNum; n++) {
        float3 offset = s.perf ? g_Special6[n] : g_Special8[n];
        float2 uv = s.pixelUv + Geometry::RotateVector(scaledRotator, float2(offset.x, offset.y));
        uv = floor(uv * c.gRectSize) + 0.5f; // snap to the pixel centre
        if (mode == PRE_BLUR)
            uv = ApplyCheckerboardShift(uv, c.gDiffCheckerboard, (uint32_t)n, c.gFrameIndex);
        const float2 k = uv - 0.5f; // the tap's pixel (exact: a pixel centre minus one half)
        uv *= c.gRectSizeInv;
        float2 uvScaled = min(uv * c.gResolutionScale, c.gResolutionScale - 0.5f * c.gResourceSizeInv); // ClampUvToViewport
        float2 checkerboardUvScaled = uvScaled; // checkerboarded inputs live in the left half of the plane
        if (mode == PRE_BLUR && c.gDiffCheckerboard != 2)
            checkerboardUvScaled.x *= 0.5f;

        float zs = UnpackViewZ(c, gIn_ViewZ.SampleNearest(uvScaled).x);
        float materialIDs;
        float4 Ns = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.SampleNearest(uvScaled), materialIDs);

        float angle = Math::AcosApprox(dot(s.N, Ns.xyz()));

        float w = IsInScreenNearest(uv);
        w *= ComputeWeight(zs * (k.x * s.geo.x + (k.y * s.geo.y + s.geo.z)), geometryWeightParams.x, geometryWeightParams.y); // dot( Nv, Xvs ), expanded (SpatialCtx::geo)
        w *= CompareMaterials(s.materialID, materialIDs, c.gDiffMinMaterial) ? 1.0f : 0.0f;
        w *= ComputeWeight(angle, normalWeightParam, 0.0f);

        S smp = Sig::From(gIn_Diff.SampleNearest(checkerboardUvScaled));
        smp = w == 0.0f ? S(0.0f) : smp; // Denanify

        w *= lerp(minHitDistWeight, 1.0f, ComputeExponentialWeight(ExtractHitDist(smp), hitDistanceWeightParams.x, hitDistanceWeightParams.y));
        w *= GetGaussianWeight(offset.z);

        sum += w;
        diff = Mad(smp, w, diff);
        if (diffSh) {
            float4 sh = gIn_DiffSh->SampleNearest(checkerboardUvScaled);
            sh = w == 0.0f ? float4(0.0f) : sh;
            *diffSh = Mad(sh, w, *diffSh);
        }
    }

    float invSum = Math::PositiveRcp(sum);
    diff = diff * invSum;
    if (diffSh)
        *diffSh *= invSum;
    return diff;
}

// "sum" = 1 when the centre pixel carries data, 0 for the empty pixels of a checkerboarded input (pre-pass only)
template <typename S>
S DiffuseSpatialFilter(const ReblurCB& c, SpatialMode mode, const SpatialCtx& s, S diff, const Tex& gIn_Diff, const Tex& gIn_ViewZ, const Tex& gIn_Normal_Roughness,
    float4* diffSh = nullptr, const Tex* gIn_DiffSh = nullptr, float sum = 1.0f) {
    typedef ReblurSignal<SignalKind<S>::value> Sig;
    if (!(mode == PRE_BLUR && c.gDiffPrepassBlurRadius == 0.0f))
        diff = DiffuseSpatialFilterTaps<S>(c, mode, s, diff, gIn_Diff, gIn_ViewZ, gIn_Normal_Roughness, diffSh, gIn_DiffSh, sum);
    if (mode == PRE_BLUR && sum == 0.0f) { // checkerboard resolve, if the pre-pass failed (REBLUR_Common_DiffuseSpatialFilter.hlsli:177-199)
        S s0 = Sig::From(gIn_Diff.Load(s.cbX0, s.py)), s1 = Sig::From(gIn_Diff.Load(s.cbX1, s.py));
        s0 = s.wc.x == 0.0f ? S(0.0f) : s0;
        s1 = s.wc.y == 0.0f ? S(0.0f) : s1;
        diff = s0 * s.wc.x + s1 * s.wc.y;
        if (diffSh) {
            float4 sh0 = gIn_DiffSh->Load(s.cbX0, s.py), sh1 = gIn_DiffSh->Load(s.cbX1, s.py);
            sh0 = s.wc.x == 0.0f ? float4(0.0f) : sh0;
            sh1 = s.wc.y == 0.0f ? float4(0.0f) : sh1;
            *diffSh = sh0 * s.wc.x + sh1 * s.wc.y;
        }
    }
    return diff;
}

// returns the filtered signal; for the pre-pass also produces hitDistForTracking (written only if the radius != 0)
template <typename S>
S SpecularSpatialFilterTaps(const ReblurCB& c, SpatialMode mode, const SpatialCtx& s, S spec, const Tex& gIn_Spec, const Tex& gIn_ViewZ,
    const Tex& gIn_Normal_Roughness, Tex* gOut_SpecHitDistForTracking, float4* specSh, const Tex* gIn_SpecSh, float& sum) { // REBLUR_SH: .xyz only (.w = roughness for AA)
    constexpr int KIND = SignalKind<S>::value;
    constexpr bool OCC = KIND == SIGNAL_OCCLUSION;
    typedef ReblurSignal<KIND> Sig;
    float smc = GetSpecMagicCurve(s.roughness);

    RngHash rng;
    if (mode == PRE_BLUR)
        rng.Initialize((uint32_t)s.px, (uint32_t)s.py, c.gFrameIndex);

    float fractionScale = 1.0f, radiusScale = 1.0f;
    if (mode == PRE_BLUR)
        fractionScale = REBLUR_PRE_BLUR_FRACTION_SCALE;
    else if (mode == BLUR)
        fractionScale = REBLUR_BLUR_FRACTION_SCALE;
    else {
        radiusScale = REBLUR_POST_BLUR_RADIUS_SCALE;
        fractionScale = REBLUR_POST_BLUR_FRACTION_SCALE;
    }

    // Hit distance factor
    float4 Dv = ImportanceSampling::GetSpecularDominantDirection(s.Nv, s.Vv, s.roughness);
    float NoD = fabsf(dot(s.Nv, Dv.xyz()));
    float hitDistScale = _REBLUR_GetHitDistanceNormalization(s.viewZ, c.gHitDistParams, s.roughness);
    float hitDist = ExtractHitDist(spec) * hitDistScale;
    float hitDistFactor = GetHitDistFactor(hitDist, s.frustumSize);

    // Blur radius
    float hitDistForTracking = 0.0f, specNonLinearAccumSpeed, blurRadius, areaFactor;
    if (mode == PRE_BLUR) {
        specNonLinearAccumSpeed = REBLUR_PRE_BLUR_NON_LINEAR_ACCUM_SPEED;
        hitDistForTracking = hitDist == 0.0f ? NRD_INF : hitDist;
        blurRadius = c.gSpecPrepassBlurRadius;
        areaFactor = s.roughness * hitDistFactor;
    } else {
        float boost = 1.0f - GetFadeBasedOnAccumulatedFrames(c, s.data1.y);
        boost *= 1.0f - BRDF::Pow5(s.NoV);
        boost *= smc;
        specNonLinearAccumSpeed = Rcp(1.0f + REBLUR_SAMPLES_PER_FRAME * (1.0f - boost) * s.data1.y);
        blurRadius = c.gMaxBlurRadius;
        areaFactor = s.roughness * hitDistFactor * specNonLinearAccumSpeed;
    }
    blurRadius *= Math::Sqrt01(areaFactor);

    if (mode == PRE_BLUR) {
        float lobeTanHalfAngle = ImportanceSampling::GetSpecularLobeTanHalfAngle(s.roughness, REBLUR_MAX_PERCENT_OF_LOBE_VOLUME_FOR_PRE_PASS);
        float lobeRadius = hitDist * NoD * lobeTanHalfAngle;
        float minBlurRadius = Div(lobeRadius, PixelRadiusToWorld(c.gUnproject, c.gOrthoMode, 1.0f, s.viewZ + hitDist * Dv.w));
        blurRadius = min(blurRadius, minBlurRadius);
    }
    blurRadius *= radiusScale;
    blurRadius = max(blurRadius, c.gMinBlurRadius * smc);

    // Weights
    float roughnessFractionScaled = saturate(c.gRoughnessFraction * fractionScale);
    float2 geometryWeightParams = GetGeometryWeightParams(c.gPlaneDistSensitivity, s.frustumSize, s.Xv, s.Nv);
    float normalWeightParam = Div(GetNormalWeightParam(specNonLinearAccumSpeed, c.gLobeAngleFraction, s.roughness), fractionScale);
    float2 roughnessWeightParams = GetRoughnessWeightParams(s.roughness, roughnessFractionScaled);
    float2 hitDistanceWeightParams = GetHitDistanceWeightParams(ExtractHitDist(spec), specNonLinearAccumSpeed, s.roughness);
    float minHitDistWeight = c.gMinHitDistanceWeight * fractionScale * smc;
    if (mode != PRE_BLUR && !OCC) // REBLUR_Common_SpecularSpatialFilter.hlsli:98
        minHitDistWeight *= HwSqrt(specNonLinearAccumSpeed);

    // Sampling set-up: screen space for the pre-pass (and for every pass in performance mode), world space along the (bent) lobe otherwise
    const bool screenSpace = mode == PRE_BLUR || s.perf; // REBLUR_USE_SCREEN_SPACE_SAMPLING_FOR_SPECULAR
    float4 scaledRotator = float4(0.0f);
    float3 T, B;
    if (screenSpace) {
        float2 skew = c.gRectSizeInv * blurRadius;
        scaledRotator = Geometry::ScaleRotator(s.rotator, skew);
    } else {
        float bentFactor = HwSqrt(hitDistFactor);
        float skewFactor = lerp(0.25f + 0.75f * s.roughness, 1.0f, NoD);
        skewFactor = lerp(skewFactor, 1.0f, specNonLinearAccumSpeed);
        skewFactor = lerp(1.0f, skewFactor, bentFactor);
        float3 bentDv = normalize(lerp(s.Nv, Dv.xyz(), bentFactor));
        GetKernelBasis(bentDv, s.Nv, T, B);
        float worldRadius = PixelRadiusToWorld(c.gUnproject, c.gOrthoMode, blurRadius, s.viewZ);
        T *= worldRadius * skewFactor;
        B *= Div(worldRadius, skewFactor);
    }
#ifndef ORC_STRICT_IEEE
    KernelProjection kernelProjection = {};
    if (!screenSpace)
        kernelProjection = MakeKernelProjection(c.gViewToClip, s.Xv, T, B, s.rotator);
#endif

    const int sampleNum = s.perf ? 6 : 8;
    for (int n = 0; n < sampleNum; n++) {
        float3 offset = s.perf ? g_Special6[n] : g_Special8[n];
        float2 uv;
        if (screenSpace)
            uv = s.pixelUv + Geometry::RotateVector(scaledRotator, float2(offset.x, offset.y));
        else
#ifdef ORC_STRICT_IEEE
            uv = GetKernelSampleCoordinates(c.gViewToClip, offset, s.Xv, T, B, s.rotator);
#else
            uv = KernelSampleUv(kernelProjection, offset.x, offset.y);
#endif

        uv = floor(uv * c.gRectSize) + 0.5f;
        if (mode == PRE_BLUR)
            uv = ApplyCheckerboardShift(uv, c.gSpecCheckerboard, (uint32_t)n, c.gFrameIndex);
        const float2 k = uv - 0.5f; // the tap's pixel (exact: a pixel centre minus one half)
        uv *= c.gRectSizeInv;
        float2 uvScaled = min(uv * c.gResolutionScale, c.gResolutionScale - 0.5f * c.gResourceSizeInv);
        float2 checkerboardUvScaled = uvScaled;
        if (mode == PRE_BLUR && c.gSpecCheckerboard != 2)
            checkerboardUvScaled.x *= 0.5f;

        float zs = UnpackViewZ(c, gIn_ViewZ.SampleNearest(uvScaled).x);
        float materialIDs;
        float4 Ns = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.SampleNearest(uvScaled), materialIDs);

        float angle = Math::AcosApprox(dot(s.N, Ns.xyz()));
        float3 Xvs = Geometry::ReconstructViewPosition(uv, c.gFrustum, zs, c.gOrthoMode);

        float w = IsInScreenNearest(uv);
        w *= ComputeWeight(zs * (k.x * s.geo.x + (k.y * s.geo.y + s.geo.z)), geometryWeightParams.x, geometryWeightParams.y); // dot( Nv, Xvs ), expanded (SpatialCtx::geo)
        w *= CompareMaterials(s.materialID, materialIDs, c.gSpecMinMaterial) ? 1.0f : 0.0f;
        w *= ComputeWeight(angle, normalWeightParam, 0.0f);
        w *= ComputeWeight(Ns.w, roughnessWeightParams.x, roughnessWeightParams.y);

        S smp = Sig::From(gIn_Spec.SampleNearest(checkerboardUvScaled));
        smp = w == 0.0f ? S(0.0f) : smp;

        if (mode == PRE_BLUR) {
            // stochastic min hit distance for tracking, ignoring zeros
            float hs = ExtractHitDist(smp) * _REBLUR_GetHitDistanceNormalization(zs, c.gHitDistParams, Ns.w);
            float d = length(Xvs - s.Xv) + NRD_EPS;
            float geometryWeight = w * saturate(Div(hs, d));
            if (rng.GetFloat() < geometryWeight)
                hitDistForTracking = min(hitDistForTracking, hs);

            w *= c.gUsePrepassNotOnlyForSpecularMotionEstimation;

            // samples close to the reflection contact should not be blurred
            float t = Div(hs, d + hitDist);
            w *= lerp(saturate(t), 1.0f, Math::LinearStep(0.5f, 1.0f, s.roughness));
        }
        w *= lerp(minHitDistWeight, 1.0f, ComputeExponentialWeight(ExtractHitDist(smp), hitDistanceWeightParams.x, hitDistanceWeightParams.y));
        w *= GetGaussianWeight(offset.z);

        sum += w;
        spec = Mad(smp, w, spec);
        if (specSh) {
            float4 sh = gIn_SpecSh->SampleNearest(checkerboardUvScaled);
            sh = w == 0.0f ? float4(0.0f) : sh;
            specSh->x += sh.x * w, specSh->y += sh.y * w, specSh->z += sh.z * w;
        }
    }

    float invSum = Math::PositiveRcp(sum);
    spec = spec * invSum;
    if (specSh)
        specSh->x *= invSum, specSh->y *= invSum, specSh->z *= invSum;

    if (mode == PRE_BLUR)
        gOut_SpecHitDistForTracking->Store(s.px, s.py, hitDistForTracking == NRD_INF ? 0.0f : hitDistForTracking);
    return spec;
}

template <typename S>
S SpecularSpatialFilter(const ReblurCB& c, SpatialMode mode, const SpatialCtx& s, S spec, const Tex& gIn_Spec, const Tex& gIn_ViewZ,
    const Tex& gIn_Normal_Roughness, Tex* gOut_SpecHitDistForTracking, float4* specSh = nullptr, const Tex* gIn_SpecSh = nullptr, float sum = 1.0f) {
    typedef ReblurSignal<SignalKind<S>::value> Sig;
    if (!(mode == PRE_BLUR && c.gSpecPrepassBlurRadius == 0.0f))
        spec = SpecularSpatialFilterTaps<S>(c, mode, s, spec, gIn_Spec, gIn_ViewZ, gIn_Normal_Roughness, gOut_SpecHitDistForTracking, specSh, gIn_SpecSh, sum);
    if (mode == PRE_BLUR && sum == 0.0f) { // checkerboard resolve, if the pre-pass failed (REBLUR_Common_SpecularSpatialFilter.hlsli:224-246; all 4 SH components)
        S s0 = Sig::From(gIn_Spec.Load(s.cbX0, s.py)), s1 = Sig::From(gIn_Spec.Load(s.cbX1, s.py));
        s0 = s.wc.x == 0.0f ? S(0.0f) : s0;
        s1 = s.wc.y == 0.0f ? S(0.0f) : s1;
        spec = s0 * s.wc.x + s1 * s.wc.y;
        if (specSh) {
            float4 sh0 = gIn_SpecSh->Load(s.cbX0, s.py), sh1 = gIn_SpecSh->Load(s.cbX1, s.py);
            sh0 = s.wc.x == 0.0f ? float4(0.0f) : sh0;
            sh1 = s.wc.y == 0.0f ? float4(0.0f) : sh1;
            *specSh = sh0 * s.wc.x + sh1 * s.wc.y;
        }
    }
    return spec;
}

// fills the per-pixel context; returns false on the early-outs (sky tile / outside rect / beyond denoising range)
bool MakeSpatialCtx(const ReblurCB& c, int px, int py, const Tex& gIn_Tiles, const Tex& gIn_ViewZ, const Tex& gIn_Normal_Roughness, float4 rotator, bool perf, SpatialCtx& s, float* viewZpackedOut = nullptr) {
    s.perf = perf;
    float isSky = gIn_Tiles.Load(px >> 4, py >> 4).x;
    if (isSky != 0.0f || px > c.gRectSizeMinusOne[0] || py > c.gRectSizeMinusOne[1])
        return false;
    float viewZpacked = gIn_ViewZ.Load(px, py).x;
    if (viewZpackedOut)
        *viewZpackedOut = viewZpacked;
    s.viewZ = UnpackViewZ(c, viewZpacked);
    if (s.viewZ > c.gDenoisingRange)
        return false;

    float4 normalAndRoughness = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.Load(px, py), s.materialID);
    s.px = px;
    s.py = py;
    s.N = normalAndRoughness.xyz();
    s.Nv = Geometry::RotateVectorInverse(c.gViewToWorld, s.N);
    s.roughness = normalAndRoughness.w;
    s.pixelUv = float2(float(px) + 0.5f, float(py) + 0.5f) * c.gRectSizeInv;
    s.Xv = Geometry::ReconstructViewPosition(s.pixelUv, c.gFrustum, s.viewZ, c.gOrthoMode);
    s.Vv = GetViewVector(c, s.Xv, true);
    s.NoV = fabsf(dot(s.Nv, s.Vv));
    s.frustumSize = GetFrustumSize(c.gMinRectDimMulUnproject, c.gOrthoMode, s.viewZ);
    s.rotator = rotator; // NRD_FRAME rotator mode: the per-frame base rotator, no per-pixel component
    s.data1 = float2(0.0f);
    {
        const float4 f = c.gFrustum;
        const float2 r = c.gRectSizeInv;
        s.geo = float3(s.Nv.x * f.z * r.x, s.Nv.y * f.w * r.y, s.Nv.x * (0.5f * r.x * f.z + f.x) + s.Nv.y * (0.5f * r.y * f.w + f.y) + s.Nv.z);
    }
    return true;
}

// ================================================================================================ PrePass
template <bool DIFF, bool SPEC, bool PERF, bool SH, int KIND>
void PrePass(const PassIO& io) {
    typedef ReblurSignal<KIND> Sig;
    typedef typename Sig::type S;
    const ReblurCB& c = *(const ReblurCB*)io.constants;
    Cursor cur(io);
    const Tex& gIn_Tiles = *cur.next();
    const Tex& gIn_Normal_Roughness = *cur.next();
    const Tex& gIn_ViewZ = *cur.next();
    const Tex* gIn_Diff = cur.nextIf(DIFF);
    const Tex* gIn_Spec = cur.nextIf(SPEC);
    const Tex* gIn_DiffSh = cur.nextIf(DIFF && SH);
    const Tex* gIn_SpecSh = cur.nextIf(SPEC && SH);
    Tex* gOut_Diff = cur.nextIf(DIFF);
    Tex* gOut_Spec = cur.nextIf(SPEC);
    Tex* gOut_SpecHitDistForTracking = cur.nextIf(SPEC);
    Tex* gOut_DiffSh = cur.nextIf(DIFF && SH);
    Tex* gOut_SpecSh = cur.nextIf(SPEC && SH);

#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < (int)c.gRectSize.y; py++)
        for (int px = 0; px < (int)c.gRectSize.x; px++) {
            SpatialCtx s;
            if (!MakeSpatialCtx(c, px, py, gIn_Tiles, gIn_ViewZ, gIn_Normal_Roughness, c.gRotatorPre, PERF, s))
                continue;
            // checkerboard resolve weights (REBLUR_PrePass.hlsli:43-56)
            const uint32_t checkerboard = Sequence::CheckerBoard((uint32_t)px, (uint32_t)py, c.gFrameIndex);
            {
                int x0 = max(px - 1, 0), x1 = min(px + 1, c.gRectSizeMinusOne[0]);
                float viewZ0 = UnpackViewZ(c, gIn_ViewZ.Load(x0, py).x), viewZ1 = UnpackViewZ(c, gIn_ViewZ.Load(x1, py).x);
                float thr = GetDisocclusionThreshold(NRD_DISOCCLUSION_THRESHOLD, s.frustumSize, s.NoV);
                float2 wc = float2(thr >= fabsf(viewZ0 - s.viewZ) ? 1.0f : 0.0f, thr >= fabsf(viewZ1 - s.viewZ) ? 1.0f : 0.0f); // GetDisocclusionWeight = step
                wc.x = (viewZ0 > c.gDenoisingRange || px < 1) ? 0.0f : wc.x;
                wc.y = (viewZ1 > c.gDenoisingRange || px >= c.gRectSizeMinusOne[0]) ? 0.0f : wc.y;
                wc *= Math::PositiveRcp(wc.x + wc.y);
                s.wc = wc;
                s.cbX0 = x0 >> 1;
                s.cbX1 = x1 >> 1;
            }
            if (DIFF) {
                const int pos = c.gDiffCheckerboard == 2 ? px : px >> 1;
                float sum = 1.0f;
                S diff = Sig::From(gIn_Diff->Load(pos, py));
                float4 diffSh = SH ? gIn_DiffSh->Load(pos, py) : float4(0.0f);
                if (c.gDiffCheckerboard != 2 && checkerboard != c.gDiffCheckerboard) {
                    sum = 0.0f;
                    diff = S(0.0f);
                    diffSh = float4(0.0f);
                }
                diff = DiffuseSpatialFilter<S>(c, PRE_BLUR, s, diff, *gIn_Diff, gIn_ViewZ, gIn_Normal_Roughness, SH ? &diffSh : nullptr, gIn_DiffSh, sum);
                gOut_Diff->Store(px, py, diff);
                if (SH)
                    gOut_DiffSh->Store(px, py, diffSh);
            }
            if (SPEC) {
                const int pos = c.gSpecCheckerboard == 2 ? px : px >> 1;
                float sum = 1.0f;
                S spec = Sig::From(gIn_Spec->Load(pos, py));
                float4 specSh = SH ? gIn_SpecSh->Load(pos, py) : float4(0.0f);
                if (c.gSpecCheckerboard != 2 && checkerboard != c.gSpecCheckerboard) {
                    sum = 0.0f;
                    spec = S(0.0f);
                    specSh = float4(0.0f);
                }
                spec = SpecularSpatialFilter<S>(c, PRE_BLUR, s, spec, *gIn_Spec, gIn_ViewZ, gIn_Normal_Roughness, gOut_SpecHitDistForTracking, SH ? &specSh : nullptr, gIn_SpecSh, sum);
                gOut_Spec->Store(px, py, spec);
                if (SH)
                    gOut_SpecSh->Store(px, py, specSh);
            }
        }
}

// ================================================================================================ Blur
template <bool DIFF, bool SPEC, bool PERF, int KIND, bool SH>
void Blur(const PassIO& io) {
    typedef ReblurSignal<KIND> Sig;
    constexpr bool OCC = KIND == SIGNAL_OCCLUSION;
    typedef typename Sig::type S;
    const ReblurCB& c = *(const ReblurCB*)io.constants;
    Cursor cur(io);
    const Tex& gIn_Tiles = *cur.next();
    const Tex& gIn_Normal_Roughness = *cur.next();
    const Tex& gIn_Data1 = *cur.next();
    const Tex* gIn_Diff = cur.nextIf(DIFF);
    const Tex* gIn_Spec = cur.nextIf(SPEC);
    const Tex& gIn_ViewZ = *cur.next();
    const Tex* gIn_DiffSh = cur.nextIf(DIFF && SH);
    const Tex* gIn_SpecSh = cur.nextIf(SPEC && SH);
    Tex* gOut_Diff = cur.nextIf(DIFF);
    Tex* gOut_Spec = cur.nextIf(SPEC);
    Tex& gOut_ViewZ = *cur.next();
    Tex* gOut_DiffSh = cur.nextIf(DIFF && SH);
    Tex* gOut_SpecSh = cur.nextIf(SPEC && SH);

#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < (int)c.gRectSize.y; py++)
        for (int px = 0; px < (int)c.gRectSize.x; px++) {
            // the copy of viewZ into PREV_VIEWZ happens BEFORE the denoising-range early-out (REBLUR_Blur.hlsli:22-27)
            float isSky = gIn_Tiles.Load(px >> 4, py >> 4).x;
            if (isSky != 0.0f || px > c.gRectSizeMinusOne[0] || py > c.gRectSizeMinusOne[1])
                continue;
            gOut_ViewZ.Store(px, py, gIn_ViewZ.Load(px, py).x);

            SpatialCtx s;
            if (!MakeSpatialCtx(c, px, py, gIn_Tiles, gIn_ViewZ, gIn_Normal_Roughness, c.gRotator, PERF, s))
                continue;
            s.data1 = UnpackData1(gIn_Data1.Load(px, py), DIFF);
            if (DIFF) {
                S diff = Sig::From(gIn_Diff->Load(px, py));
                float4 diffSh = SH ? gIn_DiffSh->Load(px, py) : float4(0.0f);
                diff = DiffuseSpatialFilter<S>(c, BLUR, s, diff, *gIn_Diff, gIn_ViewZ, gIn_Normal_Roughness, SH ? &diffSh : nullptr, gIn_DiffSh);
                gOut_Diff->Store(px, py, diff);
                if (SH)
                    gOut_DiffSh->Store(px, py, diffSh);
            }
            if (SPEC) {
                S spec = Sig::From(gIn_Spec->Load(px, py));
                float4 specSh = SH ? gIn_SpecSh->Load(px, py) : float4(0.0f);
                spec = SpecularSpatialFilter<S>(c, BLUR, s, spec, *gIn_Spec, gIn_ViewZ, gIn_Normal_Roughness, nullptr, SH ? &specSh : nullptr, gIn_SpecSh);
                gOut_Spec->Store(px, py, spec);
                if (SH)
                    gOut_SpecSh->Store(px, py, specSh);
            }
        }
}

// ================================================================================================ PostBlur
template <bool DIFF, bool SPEC, bool NO_TS, bool PERF, int KIND, bool SH>
void PostBlur(const PassIO& io) {
    typedef ReblurSignal<KIND> Sig;
    constexpr bool OCC = KIND == SIGNAL_OCCLUSION;
    typedef typename Sig::type S;
    const ReblurCB& c = *(const ReblurCB*)io.constants;
    Cursor cur(io);
    const Tex& gIn_Tiles = *cur.next();
    const Tex& gIn_Normal_Roughness = *cur.next();
    const Tex& gIn_Data1 = *cur.next();
    const Tex* gIn_Diff = cur.nextIf(DIFF);
    const Tex* gIn_Spec = cur.nextIf(SPEC);
    const Tex& gIn_ViewZ = *cur.next(); // PREV_VIEWZ written by Blur
    const Tex* gIn_DiffSh = cur.nextIf(DIFF && SH);
    const Tex* gIn_SpecSh = cur.nextIf(SPEC && SH);
    Tex& gOut_Normal_Roughness = *cur.next();
    Tex* gOut_Diff = cur.nextIf(DIFF);
    Tex* gOut_Spec = cur.nextIf(SPEC);
    Tex* gOut_InternalData = cur.nextIf(NO_TS);
    Tex* gOut_DiffCopy = cur.nextIf(NO_TS && DIFF && !OCC); // no copy in the occlusion family (the output itself is next frame's history)
    Tex* gOut_SpecCopy = cur.nextIf(NO_TS && SPEC && !OCC);
    Tex* gOut_DiffShCopy = cur.nextIf(NO_TS && DIFF && SH);
    Tex* gOut_SpecShCopy = cur.nextIf(NO_TS && SPEC && SH);
    Tex* gOut_DiffSh = cur.nextIf(DIFF && SH);
    Tex* gOut_SpecSh = cur.nextIf(SPEC && SH);

#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < (int)c.gRectSize.y; py++)
        for (int px = 0; px < (int)c.gRectSize.x; px++) {
            SpatialCtx s;
            if (!MakeSpatialCtx(c, px, py, gIn_Tiles, gIn_ViewZ, gIn_Normal_Roughness, c.gRotatorPost, PERF, s))
                continue;
            s.data1 = UnpackData1(gIn_Data1.Load(px, py), DIFF);

            // REBLUR_PostBlur.hlsli:47 stores the float4 it loaded: the same format on both sides for encodings 0..3 (the texel round-trips exactly: a copy of its
            // 4 / 8 bytes), SNORM16 -> fp16 for encoding 4 (Reblur.cpp:52-62)
            if (NRD_NORMAL_ENCODING == 4)
                gOut_Normal_Roughness.Store(px, py, gIn_Normal_Roughness.Load(px, py));
            else
                gOut_Normal_Roughness.CopyTexelFrom(gIn_Normal_Roughness, px, py, NRD_NORMAL_ENCODING <= 2 ? 4 : 8);
            if (NO_TS)
                gOut_InternalData->StoreUint(px, py, PackInternalData(s.data1.x + 1.0f, s.data1.y + 1.0f, s.materialID));

            if (DIFF) {
                S diff = Sig::From(gIn_Diff->Load(px, py));
                float4 diffSh = SH ? gIn_DiffSh->Load(px, py) : float4(0.0f);
                diff = DiffuseSpatialFilter<S>(c, POST_BLUR, s, diff, *gIn_Diff, gIn_ViewZ, gIn_Normal_Roughness, SH ? &diffSh : nullptr, gIn_DiffSh);
                gOut_Diff->Store(px, py, diff);
                if (SH)
                    gOut_DiffSh->Store(px, py, diffSh);
                if (NO_TS && !OCC)
                    gOut_DiffCopy->Store(px, py, diff);
                if (NO_TS && SH)
                    gOut_DiffShCopy->Store(px, py, diffSh);
            }
            if (SPEC) {
                S spec = Sig::From(gIn_Spec->Load(px, py));
                float4 specSh = SH ? gIn_SpecSh->Load(px, py) : float4(0.0f);
                spec = SpecularSpatialFilter<S>(c, POST_BLUR, s, spec, *gIn_Spec, gIn_ViewZ, gIn_Normal_Roughness, nullptr, SH ? &specSh : nullptr, gIn_SpecSh);
                gOut_Spec->Store(px, py, spec);
                if (SH)
                    gOut_SpecSh->Store(px, py, specSh);
                if (NO_TS && !OCC)
                    gOut_SpecCopy->Store(px, py, spec);
                if (NO_TS && SH)
                    gOut_SpecShCopy->Store(px, py, specSh);
            }
        }
}

// ================================================================================================ TemporalAccumulation
template <bool DIFF, bool SPEC, bool PERF, int KIND, bool SH>
void TemporalAccumulation(const PassIO& io) {
    typedef ReblurSignal<KIND> Sig;
    constexpr bool OCC = KIND == SIGNAL_OCCLUSION;
    typedef typename Sig::type S;
    const ReblurCB& c = *(const ReblurCB*)io.constants;
    Cursor cur(io);
    const Tex& gIn_Tiles = *cur.next();
    const Tex& gIn_Normal_Roughness = *cur.next();
    const Tex& gIn_ViewZ = *cur.next();
    const Tex& gIn_Mv = *cur.next();
    const Tex& gPrev_ViewZ = *cur.next();
    const Tex& gPrev_Normal_Roughness = *cur.next();
    const Tex& gPrev_InternalData = *cur.next();
    const Tex& gIn_DisocclusionThresholdMix = *cur.next(); // a dummy plane unless gHasDisocclusionThresholdMix
    const Tex* gIn_DiffConfidence = cur.nextIf(DIFF);      // dummies unless gHasHistoryConfidence
    const Tex* gIn_SpecConfidence = cur.nextIf(SPEC);
    const Tex* gIn_Diff = cur.nextIf(DIFF);
    const Tex* gIn_Spec = cur.nextIf(SPEC);
    const Tex* gHistory_Diff = cur.nextIf(DIFF);
    const Tex* gHistory_Spec = cur.nextIf(SPEC);
    const Tex* gHistory_DiffFast = cur.nextIf(DIFF);
    const Tex* gHistory_SpecFast = cur.nextIf(SPEC);
    const Tex* gPrev_SpecHitDistForTracking = cur.nextIf(SPEC);
    const Tex* gIn_SpecHitDistForTracking = cur.nextIf(SPEC && !OCC); // written by the pre-pass, which the occlusion family does not have
    const Tex* gIn_DiffSh = cur.nextIf(DIFF && SH);
    const Tex* gIn_SpecSh = cur.nextIf(SPEC && SH);
    const Tex* gHistory_DiffSh = cur.nextIf(DIFF && SH);
    const Tex* gHistory_SpecSh = cur.nextIf(SPEC && SH);
    Tex* gOut_Diff = cur.nextIf(DIFF);
    Tex* gOut_Spec = cur.nextIf(SPEC);
    Tex* gOut_DiffFast = cur.nextIf(DIFF);
    Tex* gOut_SpecFast = cur.nextIf(SPEC);
    Tex* gOut_SpecHitDistForTracking = cur.nextIf(SPEC);
    Tex& gOut_Data1 = *cur.next();
    Tex* gOut_Data2 = cur.nextIf(!OCC); // REBLUR_TemporalAccumulation.hlsli:822-824
    Tex* gOut_DiffSh = cur.nextIf(DIFF && SH);
    Tex* gOut_SpecSh = cur.nextIf(SPEC && SH);

    const int rw = c.gRectSizeMinusOne[0], rh = c.gRectSizeMinusOne[1];

#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py <= rh; py++)
        for (int px = 0; px <= rw; px++) {
            float isSky = gIn_Tiles.Load(px >> 4, py >> 4).x;
            if (isSky != 0.0f)
                continue;
            float viewZ = UnpackViewZ(c, gIn_ViewZ.Load(px, py).x);
            if (viewZ > c.gDenoisingRange)
                continue;

            // "shared memory": unpacked normal/roughness and hit distance for tracking at clamped positions
            auto sNR = [&](int x, int y) { return NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.Load(clamp(x, 0, rw), clamp(y, 0, rh))); };
            auto sHitDistForTracking = [&](int x, int y) {
                x = clamp(x, 0, rw);
                y = clamp(y, 0, rh);
                const int shift = (OCC && c.gSpecCheckerboard != 2) ? 1 : 0; // REBLUR_TemporalAccumulation.hlsli:21-27
                float hitDist = (OCC || c.gSpecPrepassBlurRadius == 0.0f) ? ExtractHitDist(Sig::From(gIn_Spec->Load(x >> shift, y))) : gIn_SpecHitDistForTracking->Load(x, y).x;
                return hitDist == 0.0f ? NRD_INF : hitDist;
            };

            // Current position
            float2 pixelUv = float2(float(px) + 0.5f, float(py) + 0.5f) * c.gRectSizeInv;
            float3 Xv = Geometry::ReconstructViewPosition(pixelUv, c.gFrustum, viewZ, c.gOrthoMode);
            float3 X = Geometry::RotateVector(c.gViewToWorld, Xv);

            // Hit distance for tracking, averaged normal and roughness variance over 3x3
            float3 Navg = float3(0.0f);
            float hitDistForTracking = NRD_INF, roughnessM1 = 0.0f, roughnessM2 = 0.0f;
            for (int j = 0; j <= 2; j++)
                for (int i = 0; i <= 2; i++) {
                    float4 nr = sNR(px - 1 + i, py - 1 + j);
                    if (i < 2 && j < 2)
                        Navg += nr.xyz();
                    if (SPEC) {
                        hitDistForTracking = min(hitDistForTracking, sHitDistForTracking(px - 1 + i, py - 1 + j));
                        float roughnessSq = nr.w * nr.w;
                        roughnessM1 += roughnessSq;
                        roughnessM2 += roughnessSq * roughnessSq;
                    }
                }
            Navg = Navg * 0.25f;

            float materialID;
            float4 normalAndRoughness = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.Load(px, py), materialID);
            float3 N = normalAndRoughness.xyz();
            float roughness = normalAndRoughness.w;

            float roughnessModified = 0.0f, roughnessSigma = 0.0f, hitDistNormalization = 0.0f;
            RngHash rng;
            if (SPEC) {
                roughnessModified = Filtering::GetModifiedRoughnessFromNormalVariance(roughness, Navg);
                roughnessM1 = DivConst(roughnessM1, 9.0f);
                roughnessM2 = DivConst(roughnessM2, 9.0f);
                roughnessSigma = HwSqrt(fabsf(roughnessM2 - roughnessM1 * roughnessM1)); // GetStdDev

                rng.Initialize((uint32_t)px, (uint32_t)py, c.gFrameIndex);

                hitDistForTracking = hitDistForTracking == NRD_INF ? 0.0f : hitDistForTracking;
                hitDistNormalization = _REBLUR_GetHitDistanceNormalization(viewZ, c.gHitDistParams, roughness);
                hitDistForTracking *= (OCC || c.gSpecPrepassBlurRadius == 0.0f) ? hitDistNormalization : 1.0f;
                gOut_SpecHitDistForTracking->Store(px, py, hitDistForTracking);
            }

            // Previous position and surface motion uv
            float4 mvRaw = gIn_Mv.Load(px, py);
            float3 mv = float3(mvRaw.x, mvRaw.y, mvRaw.z) * c.gMvScale.xyz();
            float3 Xprev = X;
            float2 smbPixelUv = pixelUv + float2(mv.x, mv.y);
            if (c.gMvScale.w == 0.0f) {
                if (c.gMvScale.z == 0.0f)
                    mv.z = Geometry::AffineTransform(c.gWorldToViewPrev, X).z - viewZ;
                float viewZprev = viewZ + mv.z;
                float3 Xvprevlocal = Geometry::ReconstructViewPosition(smbPixelUv, c.gFrustumPrev, viewZprev, c.gOrthoMode);
                Xprev = Geometry::RotateVectorInverse(c.gWorldToViewPrev, Xvprevlocal) + c.gCameraDelta.xyz();
            } else {
                Xprev += mv;
                smbPixelUv = Geometry::GetScreenUv(c.gWorldToClipPrev, Xprev);
            }

            // Previous viewZ over the 4x4 Catmull-Rom footprint; q<k> = quad k in (0,0)(1,0)(0,1)(1,1) order
            float2 catromOrigin = Filtering::GetCatmullRomOrigin(smbPixelUv, c.gRectSizePrev);
            int cx = (int)catromOrigin.x, cy = (int)catromOrigin.y;
            auto quadZ = [&](int ox, int oy) {
                return float4(gPrev_ViewZ.FetchClamped(cx + ox, cy + oy).x, gPrev_ViewZ.FetchClamped(cx + ox + 1, cy + oy).x, gPrev_ViewZ.FetchClamped(cx + ox, cy + oy + 1).x,
                    gPrev_ViewZ.FetchClamped(cx + ox + 1, cy + oy + 1).x);
            };
            float4 smbViewZ0 = quadZ(0, 0), smbViewZ1 = quadZ(2, 0), smbViewZ2 = quadZ(0, 2), smbViewZ3 = quadZ(2, 2);
            float3 prevViewZ0 = float3(UnpackViewZ(c, smbViewZ0.y), UnpackViewZ(c, smbViewZ0.z), UnpackViewZ(c, smbViewZ0.w));
            float3 prevViewZ1 = float3(UnpackViewZ(c, smbViewZ1.x), UnpackViewZ(c, smbViewZ1.z), UnpackViewZ(c, smbViewZ1.w));
            float3 prevViewZ2 = float3(UnpackViewZ(c, smbViewZ2.x), UnpackViewZ(c, smbViewZ2.y), UnpackViewZ(c, smbViewZ2.w));
            float3 prevViewZ3 = float3(UnpackViewZ(c, smbViewZ3.x), UnpackViewZ(c, smbViewZ3.y), UnpackViewZ(c, smbViewZ3.z));

            // Previous normal averaged over the valid pixels of the 2x2 footprint
            Filtering::Bilinear smbBilinearFilter = Filtering::GetBilinearFilter(smbPixelUv, c.gRectSizePrev);
            float3 smbNavg;
            {
                int bx = (int)smbBilinearFilter.origin.x, by = (int)smbBilinearFilter.origin.y; // taps outside the plane load 0 (our pinned choice; D3D ftou would clamp a negative origin)
                float sumw = 0.0f;
                float w = prevViewZ0.z < c.gDenoisingRange ? 1.0f : 0.0f;
                smbNavg = NRD_FrontEnd_UnpackNormalAndRoughness(gPrev_Normal_Roughness.Load(bx, by)).xyz() * w;
                sumw += w;
                w = prevViewZ1.y < c.gDenoisingRange ? 1.0f : 0.0f;
                smbNavg += NRD_FrontEnd_UnpackNormalAndRoughness(gPrev_Normal_Roughness.Load(bx + 1, by)).xyz() * w;
                sumw += w;
                w = prevViewZ2.y < c.gDenoisingRange ? 1.0f : 0.0f;
                smbNavg += NRD_FrontEnd_UnpackNormalAndRoughness(gPrev_Normal_Roughness.Load(bx, by + 1)).xyz() * w;
                sumw += w;
                w = prevViewZ3.x < c.gDenoisingRange ? 1.0f : 0.0f;
                smbNavg += NRD_FrontEnd_UnpackNormalAndRoughness(gPrev_Normal_Roughness.Load(bx + 1, by + 1)).xyz() * w;
                sumw += w;
                smbNavg = Div(smbNavg, sumw == 0.0f ? 1.0f : sumw);
            }
            smbNavg = Geometry::RotateVector(c.gWorldPrevToWorld, smbNavg);

            // Parallax
            float smbParallaxInPixels1 = ComputeParallaxInPixels(Xprev + c.gCameraDelta.xyz(), c.gOrthoMode == 0.0f ? smbPixelUv : pixelUv, c.gWorldToClipPrev, c.gRectSize);
            float smbParallaxInPixels2 = ComputeParallaxInPixels(Xprev - c.gCameraDelta.xyz(), c.gOrthoMode == 0.0f ? pixelUv : smbPixelUv, c.gWorldToClip, c.gRectSize);
            float smbParallaxInPixelsMax = max(smbParallaxInPixels1, smbParallaxInPixels2);
            float smbParallaxInPixelsMin = min(smbParallaxInPixels1, smbParallaxInPixels2);

            // Disocclusion: threshold
            float pixelSize = PixelRadiusToWorld(c.gUnproject, c.gOrthoMode, 1.0f, viewZ);
            float frustumSize = GetFrustumSize(c.gMinRectDimMulUnproject, c.gOrthoMode, viewZ);

            float disocclusionThresholdMix = 0.0f;
            if (materialID == c.gStrandMaterialID)
                disocclusionThresholdMix = Div(pixelSize, pixelSize + c.gStrandThickness); // NRD_GetNormalizedStrandThickness, NRD.hlsli:1158-1161 (round 4: was restated as saturate( thickness / pixelSize ); found by the per-pass comparison with the reference text)
            if (c.gHasDisocclusionThresholdMix)
                disocclusionThresholdMix = gIn_DisocclusionThresholdMix.Load((int)c.gRectOrigin[0] + px, (int)c.gRectOrigin[1] + py).x;
            float disocclusionThreshold = lerp(c.gDisocclusionThreshold, c.gDisocclusionThresholdAlternate, disocclusionThresholdMix);

            float smallParallax = Math::LinearStep(0.25f, 0.0f, smbParallaxInPixelsMax);
            disocclusionThreshold += 0.05f * smallParallax;

            float3 V = GetViewVector(c, X);
            float NoV = fabsf(dot(N, V));
            float NoVstrict = lerp(NoV, 1.0f, saturate(DivConst(smbParallaxInPixelsMax, 30.0f)));
            float4 smbDisocclusionThreshold = float4(GetDisocclusionThreshold(disocclusionThreshold, frustumSize, NoVstrict));
            smbDisocclusionThreshold *= dot(smbNavg, Navg) > REBLUR_ALMOST_ZERO_ANGLE - 0.25f * smallParallax ? 1.0f : 0.0f;
            smbDisocclusionThreshold *= IsInScreenBilinear(smbBilinearFilter.origin, c.gRectSizePrev);
            smbDisocclusionThreshold -= NRD_EPS;

            // Disocclusion: plane distance
            float3 Xvprev = Geometry::AffineTransform(c.gWorldToViewPrev, Xprev);
            float3 smbOcclusion0 = step(abs(prevViewZ0 - Xvprev.z), smbDisocclusionThreshold.x);
            float3 smbOcclusion1 = step(abs(prevViewZ1 - Xvprev.z), smbDisocclusionThreshold.y);
            float3 smbOcclusion2 = step(abs(prevViewZ2 - Xvprev.z), smbDisocclusionThreshold.z);
            float3 smbOcclusion3 = step(abs(prevViewZ3 - Xvprev.z), smbDisocclusionThreshold.w);

            // Disocclusion: materialID (R10G10B10A2 encoding carries one)
            auto quadU = [&](int ox, int oy, uint32_t q[4]) {
                q[0] = gPrev_InternalData.FetchUintClamped(cx + ox, cy + oy);
                q[1] = gPrev_InternalData.FetchUintClamped(cx + ox + 1, cy + oy);
                q[2] = gPrev_InternalData.FetchUintClamped(cx + ox, cy + oy + 1);
                q[3] = gPrev_InternalData.FetchUintClamped(cx + ox + 1, cy + oy + 1);
            };
            uint32_t id0[4], id1[4], id2[4], id3[4];
            quadU(0, 0, id0), quadU(2, 0, id1), quadU(0, 2, id2), quadU(2, 2, id3);
            float minMaterialID = min(c.gSpecMinMaterial, c.gDiffMinMaterial);
            auto matCmp = [&](uint32_t p) { return CompareMaterials(materialID, UnpackInternalData(p).z, minMaterialID) ? 1.0f : 0.0f; };
            smbOcclusion0 = smbOcclusion0 * float3(matCmp(id0[1]), matCmp(id0[2]), matCmp(id0[3]));
            smbOcclusion1 = smbOcclusion1 * float3(matCmp(id1[0]), matCmp(id1[2]), matCmp(id1[3]));
            smbOcclusion2 = smbOcclusion2 * float3(matCmp(id2[0]), matCmp(id2[1]), matCmp(id2[3]));
            smbOcclusion3 = smbOcclusion3 * float3(matCmp(id3[0]), matCmp(id3[1]), matCmp(id3[2]));
            uint32_t smbInternalData[4] = {id0[3], id1[2], id2[1], id3[0]};

            // 2x2 occlusion weights
            float4 smbOcclusionWeights = Filtering::GetBilinearCustomWeights(smbBilinearFilter, float4(smbOcclusion0.z, smbOcclusion1.y, smbOcclusion2.y, smbOcclusion3.x));
            bool smbAllowCatRom = sum(smbOcclusion0 + smbOcclusion1 + smbOcclusion2 + smbOcclusion3) > 11.5f && !PERF && KIND != SIGNAL_DIRECTIONAL_OCCLUSION; // REBLUR_USE_CATROM_FOR_SURFACE_MOTION_IN_TA

            float fbits = smbOcclusion0.z * 1.0f;
            fbits += smbOcclusion1.y * 2.0f;
            fbits += smbOcclusion2.y * 4.0f;
            fbits += smbOcclusion3.x * 8.0f;

            // Accumulation speed
            float3 internalData00 = UnpackInternalData(smbInternalData[0]), internalData10 = UnpackInternalData(smbInternalData[1]);
            float3 internalData01 = UnpackInternalData(smbInternalData[2]), internalData11 = UnpackInternalData(smbInternalData[3]);
            float diffAccumSpeed = 0.0f, smbSpecAccumSpeed = 0.0f;
            if (DIFF)
                diffAccumSpeed = Filtering::ApplyBilinearCustomWeights(internalData00.x, internalData10.x, internalData01.x, internalData11.x, smbOcclusionWeights);
            if (SPEC)
                smbSpecAccumSpeed = Filtering::ApplyBilinearCustomWeights(internalData00.y, internalData10.y, internalData01.y, internalData11.y, smbOcclusionWeights);

            // Footprint quality
            float3 smbVprev = GetViewVectorPrev(c, Xprev, c.gCameraDelta.xyz());
            float NoVprev = fabsf(dot(N, smbVprev));
            float sizeQuality = Div(NoVprev + 1e-3f, NoV + 1e-3f);
            sizeQuality *= sizeQuality;
            sizeQuality = lerp(0.1f, 1.0f, saturate(sizeQuality));

            float smbFootprintQuality = Filtering::ApplyBilinearFilter(smbOcclusion0.z, smbOcclusion1.y, smbOcclusion2.y, smbOcclusion3.x, smbBilinearFilter);
            smbFootprintQuality = Math::Sqrt01(smbFootprintQuality);
            smbFootprintQuality *= sizeQuality;

            const float2 smbSamplePos = saturate(smbPixelUv) * c.gRectSizePrev;

            // Checkerboard resolve (REBLUR_TemporalAccumulation.hlsli:307-321): only the occlusion family resolves here (the others did in the pre-pass)
            const uint32_t checkerboard = Sequence::CheckerBoard((uint32_t)px, (uint32_t)py, c.gFrameIndex);
            const bool diffHasData = c.gDiffCheckerboard == 2 || checkerboard == c.gDiffCheckerboard;
            const bool specHasData = c.gSpecCheckerboard == 2 || checkerboard == c.gSpecCheckerboard;
            int cbX0 = 0, cbX1 = 0;
            float2 wc = float2(0.0f);
            if (OCC) {
                int x0 = max(px - 1, 0), x1 = min(px + 1, c.gRectSizeMinusOne[0]);
                float viewZ0 = UnpackViewZ(c, gIn_ViewZ.Load(x0, py).x), viewZ1 = UnpackViewZ(c, gIn_ViewZ.Load(x1, py).x);
                float thr = GetDisocclusionThreshold(NRD_DISOCCLUSION_THRESHOLD, frustumSize, NoV);
                wc = float2(thr >= fabsf(viewZ0 - viewZ) ? 1.0f : 0.0f, thr >= fabsf(viewZ1 - viewZ) ? 1.0f : 0.0f);
                wc.x = (viewZ0 > c.gDenoisingRange || px < 1) ? 0.0f : wc.x;
                wc.y = (viewZ1 > c.gDenoisingRange || px >= c.gRectSizeMinusOne[0]) ? 0.0f : wc.y;
                wc *= Math::PositiveRcp(wc.x + wc.y);
                cbX0 = x0 >> 1;
                cbX1 = x1 >> 1;
            }

            // ---------------------------------------------------------------------------------------------- specular
            float specAccumSpeed = 0.0f, curvature = 0.0f, virtualHistoryAmount = 0.0f;
            if (SPEC) {
                float specHistoryConfidence = smbFootprintQuality;
                if (c.gHasHistoryConfidence)
                    specHistoryConfidence *= gIn_SpecConfidence->Load((int)c.gRectOrigin[0] + px, (int)c.gRectOrigin[1] + py).x;
                smbSpecAccumSpeed *= lerp(specHistoryConfidence, 1.0f, Rcp(1.0f + smbSpecAccumSpeed));
                smbSpecAccumSpeed = min(smbSpecAccumSpeed, c.gMaxAccumulatedFrameNum);

                S spec = Sig::From(gIn_Spec->Load((OCC && c.gSpecCheckerboard != 2) ? px >> 1 : px, py));
                if (OCC && !specHasData) {
                    S s0 = Sig::From(gIn_Spec->Load(cbX0, py)), s1 = Sig::From(gIn_Spec->Load(cbX1, py));
                    s0 = wc.x == 0.0f ? S(0.0f) : s0;
                    s1 = wc.y == 0.0f ? S(0.0f) : s1;
                    spec = s0 * wc.x + s1 * wc.y;
                }

                // Curvature estimation along predicted motion
                {
                    float2 uvForZeroParallax = c.gOrthoMode == 0.0f ? smbPixelUv : pixelUv;
                    float2 deltaUv = uvForZeroParallax - Geometry::GetScreenUv(c.gWorldToClipPrev, Xprev + c.gCameraDelta.xyz());
                    deltaUv *= c.gRectSize;
                    deltaUv = Div(deltaUv, max(smbParallaxInPixels1, 1.0f / 256.0f));

                    // 10 edge
                    float3 n10, x10;
                    {
                        float3 xv = Geometry::ReconstructViewPosition(pixelUv + float2(1, 0) * c.gRectSizeInv, c.gFrustum, 1.0f, c.gOrthoMode);
                        float3 x = Geometry::RotateVector(c.gViewToWorld, xv);
                        float3 v = GetViewVector(c, x);
                        float3 o = c.gOrthoMode == 0.0f ? float3(0.0f) : x;
                        x10 = o + Div(v * dot(X - o, N), dot(N, v));
                        n10 = sNR(px + 1, py).xyz();
                    }
                    // 01 edge
                    float3 n01, x01;
                    {
                        float3 xv = Geometry::ReconstructViewPosition(pixelUv + float2(0, 1) * c.gRectSizeInv, c.gFrustum, 1.0f, c.gOrthoMode);
                        float3 x = Geometry::RotateVector(c.gViewToWorld, xv);
                        float3 v = GetViewVector(c, x);
                        float3 o = c.gOrthoMode == 0.0f ? float3(0.0f) : x;
                        x01 = o + Div(v * dot(X - o, N), dot(N, v));
                        n01 = sNR(px, py + 1).xyz();
                    }
                    // Mix
                    float2 w = abs(deltaUv) + 1.0f / 256.0f;
                    w = Div(w, w.x + w.y);
                    float3 x = x10 * w.x + x01 * w.y;
                    float3 n = normalize(n10 * w.x + n01 * w.y);

                    // High parallax: flatten the surface on fast motion
                    float deltaUvLenFixed = smbParallaxInPixelsMin;
                    deltaUvLenFixed *= 1.0f + c.gFramerateScale * Sequence::Bayer4x4((uint32_t)px, (uint32_t)py, c.gFrameIndex);

                    float2 motionUvHigh = pixelUv + deltaUvLenFixed * deltaUv * c.gRectSizeInv;
                    motionUvHigh = (floor(motionUvHigh * c.gRectSize) + 0.5f) * c.gRectSizeInv;

                    if (deltaUvLenFixed > 1.0f && IsInScreenNearest(motionUvHigh) != 0.0f) {
                        float2 uvScaled = min(motionUvHigh * c.gResolutionScale, c.gResolutionScale - 0.5f * c.gResourceSizeInv);
                        float zHigh = UnpackViewZ(c, gIn_ViewZ.SampleNearest(uvScaled).x);
                        float3 xHigh = Geometry::ReconstructViewPosition(motionUvHigh, c.gFrustum, zHigh, c.gOrthoMode);
                        xHigh = Geometry::RotateVector(c.gViewToWorld, xHigh);
                        float3 nHigh = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.SampleNearest(uvScaled)).xyz();

                        float zError = fabsf(zHigh - viewZ) * rcp(max(zHigh, viewZ));
                        bool cmp = zError < NRD_CURVATURE_Z_THRESHOLD;
                        n = cmp ? nHigh : n;
                        x = cmp ? xHigh : x;
                    }

                    float3 edge = x - X;
                    float edgeLenSq = Math::LengthSquared(edge);
                    curvature = dot(n - N, edge) * Math::PositiveRcp(edgeLenSq);
                }

                // Virtual motion - coordinates
                float3 Xvirtual = GetXvirtual(hitDistForTracking, curvature, X, Xprev, N, V, roughness);
                float XvirtualLength = length(Xvirtual);

                float2 vmbPixelUv = Geometry::GetScreenUv(c.gWorldToClipPrev, Xvirtual);
                vmbPixelUv = materialID == c.gCameraAttachedReflectionMaterialID ? smbPixelUv : vmbPixelUv;

                float2 vmbDelta = vmbPixelUv - smbPixelUv;
                float vmbPixelsTraveled = length(vmbDelta * c.gRectSize);

                // Virtual motion - roughness (gather of the previous roughness = blue channel of the packed texel)
                Filtering::Bilinear vmbBilinearFilter = Filtering::GetBilinearFilter(vmbPixelUv, c.gRectSizePrev);
                int vx = (int)vmbBilinearFilter.origin.x, vy = (int)vmbBilinearFilter.origin.y;
                float2 relaxedRoughnessWeightParams = GetRelaxedRoughnessWeightParams(roughness * roughness, c.gRoughnessFraction, REBLUR_ROUGHNESS_SENSITIVITY_IN_TA);
                // (REBLUR_TemporalAccumulation.hlsli:463-467: GatherBlue for R10G10B10A2, GatherAlpha for the RGBA encodings -- the STORED roughness either way)
                constexpr int RC = NRD_NORMAL_ENCODING == 2 ? 2 : 3;
                float4 vmbRoughness = float4(gPrev_Normal_Roughness.FetchClamped(vx, vy)[RC], gPrev_Normal_Roughness.FetchClamped(vx + 1, vy)[RC],
                    gPrev_Normal_Roughness.FetchClamped(vx, vy + 1)[RC], gPrev_Normal_Roughness.FetchClamped(vx + 1, vy + 1)[RC]);
                float4 roughnessWeight;
                for (int k = 0; k < 4; k++)
                    roughnessWeight[k] = ComputeNonExponentialWeightWithSigma(vmbRoughness[k] * vmbRoughness[k], relaxedRoughnessWeightParams.x, relaxedRoughnessWeightParams.y, roughnessSigma);
                float jitterFriendly = Math::SmoothStep(1.0f, 0.0f, smbParallaxInPixelsMax);
                for (int k = 0; k < 4; k++)
                    roughnessWeight[k] = lerp(jitterFriendly, 1.0f, roughnessWeight[k]);
                float virtualHistoryRoughnessBasedConfidence = Filtering::ApplyBilinearFilter(roughnessWeight.x, roughnessWeight.y, roughnessWeight.z, roughnessWeight.w, vmbBilinearFilter);

                // Virtual motion - normal: parallax. Stochastic nearest tap of the bilinear footprint (REBLUR_USE_STF = 1)
                auto stochasticBilinearFetch = [&](float2 uv) {
                    if (!NRD_STOCHASTIC_BILINEAR) // Common.hlsli:76-85, 359-372: gLinearClamp at the unmodified uv, no draws (every encoding but R10G10B10A2)
                        return NRD_FrontEnd_UnpackNormalAndRoughness(gPrev_Normal_Roughness.SampleLinearTexel(uv * c.gResolutionScalePrev * float2(float(gPrev_Normal_Roughness.W()), float(gPrev_Normal_Roughness.H()))));
                    Filtering::Bilinear f = Filtering::GetBilinearFilter(uv, c.gRectSizePrev);
                    float2 rnd = rng.GetFloat2();
                    f.origin += step(rnd, f.weights);
                    float2 uvs = (Div(f.origin + 0.5f, c.gRectSizePrev)) * c.gResolutionScalePrev;
                    return NRD_FrontEnd_UnpackNormalAndRoughness(gPrev_Normal_Roughness.SampleNearest(uvs));
                };
                float4 vmbNormalAndRoughness = stochasticBilinearFetch(vmbPixelUv);
                float3 vmbN = Geometry::RotateVector(c.gWorldPrevToWorld, vmbNormalAndRoughness.xyz());
                float Dfactor = ImportanceSampling::GetSpecularDominantFactor(NoV, roughness);
                float virtualHistoryNormalBasedConfidence = Rcp(1.0f + 0.5f * Dfactor * saturate(length(N - vmbN) - REBLUR_NORMAL_ULP) * vmbPixelsTraveled);

                // Patch "smbNavg" if "smb" motion is invalid
                smbNavg = smbFootprintQuality == 0.0f ? vmbN : smbNavg;

                // Virtual motion - disocclusion: plane distance and roughness
                float4 vmbOcclusion;
                {
                    float4 vmbOcclusionThreshold = float4(disocclusionThreshold * frustumSize);
                    vmbOcclusionThreshold *= lerp(0.25f, 1.0f, NoV);
                    vmbOcclusionThreshold *= dot(vmbN, N) > REBLUR_ALMOST_ZERO_ANGLE ? 1.0f : 0.0f;
                    vmbOcclusionThreshold *= dot(vmbN, smbNavg) > REBLUR_ALMOST_ZERO_ANGLE ? 1.0f : 0.0f;
                    vmbOcclusionThreshold *= IsInScreenBilinear(vmbBilinearFilter.origin, c.gRectSizePrev);
                    vmbOcclusionThreshold -= NRD_EPS;

                    float4 vmbViewZ = float4(UnpackViewZ(c, gPrev_ViewZ.FetchClamped(vx, vy).x), UnpackViewZ(c, gPrev_ViewZ.FetchClamped(vx + 1, vy).x),
                        UnpackViewZ(c, gPrev_ViewZ.FetchClamped(vx, vy + 1).x), UnpackViewZ(c, gPrev_ViewZ.FetchClamped(vx + 1, vy + 1).x));
                    float3 vmbVv = Geometry::ReconstructViewPosition(vmbPixelUv, c.gFrustumPrev, 1.0f);
                    float3 vmbV = Geometry::RotateVectorInverse(c.gWorldToViewPrev, vmbVv);
                    float NoXcurr = dot(N, Xprev - c.gCameraDelta.xyz());
                    float4 NoXprev = (N.x * vmbV.x + N.y * vmbV.y) * (c.gOrthoMode == 0.0f ? vmbViewZ : float4(c.gOrthoMode)) + N.z * vmbV.z * vmbViewZ;
                    float4 vmbPlaneDist = abs(NoXprev - NoXcurr);

                    vmbOcclusion = step(vmbPlaneDist, vmbOcclusionThreshold);
                    vmbOcclusion *= step(float4(0.5f), roughnessWeight);
                }

                // Virtual motion - disocclusion: materialID
                float3 vmbInternalData00 = UnpackInternalData(gPrev_InternalData.FetchUintClamped(vx, vy));
                float3 vmbInternalData10 = UnpackInternalData(gPrev_InternalData.FetchUintClamped(vx + 1, vy));
                float3 vmbInternalData01 = UnpackInternalData(gPrev_InternalData.FetchUintClamped(vx, vy + 1));
                float3 vmbInternalData11 = UnpackInternalData(gPrev_InternalData.FetchUintClamped(vx + 1, vy + 1));
                vmbOcclusion.x *= CompareMaterials(materialID, vmbInternalData00.z, c.gSpecMinMaterial) ? 1.0f : 0.0f;
                vmbOcclusion.y *= CompareMaterials(materialID, vmbInternalData10.z, c.gSpecMinMaterial) ? 1.0f : 0.0f;
                vmbOcclusion.z *= CompareMaterials(materialID, vmbInternalData01.z, c.gSpecMinMaterial) ? 1.0f : 0.0f;
                vmbOcclusion.w *= CompareMaterials(materialID, vmbInternalData11.z, c.gSpecMinMaterial) ? 1.0f : 0.0f;

                fbits += vmbOcclusion.x * 16.0f;
                fbits += vmbOcclusion.y * 32.0f;
                fbits += vmbOcclusion.z * 64.0f;
                fbits += vmbOcclusion.w * 128.0f;

                // Virtual motion - accumulation speed
                float4 vmbOcclusionWeights = Filtering::GetBilinearCustomWeights(vmbBilinearFilter, vmbOcclusion);
                float vmbSpecAccumSpeed = Filtering::ApplyBilinearCustomWeights(vmbInternalData00.y, vmbInternalData10.y, vmbInternalData01.y, vmbInternalData11.y, vmbOcclusionWeights);

                float vmbFootprintQuality = Filtering::ApplyBilinearFilter(vmbOcclusion.x, vmbOcclusion.y, vmbOcclusion.z, vmbOcclusion.w, vmbBilinearFilter);
                vmbFootprintQuality = Math::Sqrt01(vmbFootprintQuality);
                vmbSpecAccumSpeed *= lerp(vmbFootprintQuality, 1.0f, Rcp(1.0f + vmbSpecAccumSpeed));

                bool vmbAllowCatRom = sum(vmbOcclusion) > 3.5f && !PERF && KIND != SIGNAL_DIRECTIONAL_OCCLUSION; // REBLUR_USE_CATROM_FOR_VIRTUAL_MOTION_IN_TA
                vmbAllowCatRom = vmbAllowCatRom && smbAllowCatRom;

                // How many radians can the travelled pixels be?
                float curvatureAngleTan = pixelSize * fabsf(curvature);
                curvatureAngleTan *= max(Div(vmbPixelsTraveled, max(NoV, 0.01f)), 1.0f);
                curvatureAngleTan *= 2.0f;
                float curvatureAngle = atan(curvatureAngleTan);

                float percentOfVolume = Div(NRD_MAX_PERCENT_OF_LOBE_VOLUME, 1.0f + vmbSpecAccumSpeed);
                float lobeTanHalfAngle = ImportanceSampling::GetSpecularLobeTanHalfAngle(roughnessModified, percentOfVolume);
                float lobeHalfAngle = atan(lobeTanHalfAngle);
                lobeHalfAngle = max(lobeHalfAngle, NRD_NORMAL_ENCODING_ERROR);

                // Virtual motion - normal: lobe overlapping
                float normalWeight = GetEncodingAwareNormalWeight(N, vmbN, lobeHalfAngle, curvatureAngle, REBLUR_NORMAL_ULP);
                normalWeight = lerp(Math::SmoothStep(1.0f, 0.0f, vmbPixelsTraveled), 1.0f, normalWeight);
                virtualHistoryNormalBasedConfidence = min(virtualHistoryNormalBasedConfidence, normalWeight);

                // Virtual history amount
                virtualHistoryAmount = Math::SmoothStep(0.05f, 0.95f, Dfactor);
                virtualHistoryAmount *= virtualHistoryNormalBasedConfidence;

                // Virtual motion - virtual parallax difference
                float virtualHistoryParallaxBasedConfidence;
                {
                    float hitDistForTrackingPrev = gPrev_SpecHitDistForTracking->SampleLinearTexelScalar(vmbPixelUv * c.gResolutionScalePrev * float2(float(gPrev_SpecHitDistForTracking->W()), float(gPrev_SpecHitDistForTracking->H())));
                    float3 XvirtualPrev = GetXvirtual(hitDistForTrackingPrev, curvature, X, Xprev, N, V, roughness);

                    float2 vmbPixelUvPrev = Geometry::GetScreenUv(c.gWorldToClipPrev, XvirtualPrev);
                    vmbPixelUvPrev = materialID == c.gCameraAttachedReflectionMaterialID ? smbPixelUv : vmbPixelUvPrev;

                    float pixelSizeAtXvirtual = PixelRadiusToWorld(c.gUnproject, c.gOrthoMode, 1.0f, XvirtualLength);
                    float r = Div((lobeTanHalfAngle + curvatureAngle) * min(hitDistForTracking, hitDistForTrackingPrev), pixelSizeAtXvirtual);
                    float d = length((vmbPixelUvPrev - vmbPixelUv) * c.gRectSize);

                    r = max(r, 0.1f);
                    virtualHistoryParallaxBasedConfidence = Math::LinearStep(r, 0.0f, d);
                }

                // Virtual motion - normal & roughness prev-prev tests (1 iteration)
                float stepBetweenTaps = min(vmbPixelsTraveled * c.gFramerateScale, 2.0f) + vmbPixelsTraveled * 1.0f;
                vmbDelta *= Math::Rsqrt(Math::LengthSquared(vmbDelta));
                vmbDelta = Div(vmbDelta, c.gRectSizePrev);

                relaxedRoughnessWeightParams = GetRelaxedRoughnessWeightParams(vmbNormalAndRoughness.w * vmbNormalAndRoughness.w, c.gRoughnessFraction, REBLUR_ROUGHNESS_SENSITIVITY_IN_TA);
                {
                    const float i = 1.0f;
                    float2 vmbPixelUvPrev = vmbPixelUv + vmbDelta * i * stepBetweenTaps;
                    float4 vmbNormalAndRoughnessPrev = stochasticBilinearFetch(vmbPixelUvPrev);

                    float2 w;
                    w.x = GetEncodingAwareNormalWeight(vmbNormalAndRoughness.xyz(), vmbNormalAndRoughnessPrev.xyz(), lobeHalfAngle, curvatureAngle * (1.0f + i * stepBetweenTaps), REBLUR_NORMAL_ULP);
                    w.y = ComputeNonExponentialWeightWithSigma(vmbNormalAndRoughnessPrev.w * vmbNormalAndRoughnessPrev.w, relaxedRoughnessWeightParams.x, relaxedRoughnessWeightParams.y, roughnessSigma);
                    if (NRD_STOCHASTIC_BILINEAR)
                        w = lerp(float2(1.0f), w, saturate(stepBetweenTaps)); // cures "StochasticBilinear" issues (`:599-602`)
                    w = IsInScreenNearest(vmbPixelUvPrev) != 0.0f ? w : float2(1.0f);

                    virtualHistoryNormalBasedConfidence = min(virtualHistoryNormalBasedConfidence, w.x);
                    virtualHistoryRoughnessBasedConfidence = min(virtualHistoryRoughnessBasedConfidence, w.y);
                }

                // Virtual history confidence
                float virtualHistoryConfidenceForSmbRelaxation = virtualHistoryNormalBasedConfidence * virtualHistoryRoughnessBasedConfidence;
                float virtualHistoryConfidence = virtualHistoryNormalBasedConfidence * virtualHistoryRoughnessBasedConfidence * virtualHistoryParallaxBasedConfidence;
                virtualHistoryAmount *= virtualHistoryRoughnessBasedConfidence;

                // Sample surface history
                HistoryFilter smbFilter = MakeHistoryFilter(smbSamplePos, smbOcclusionWeights, smbAllowCatRom);
                S smbSpecHistory = FetchSignalHistory(smbFilter, *gHistory_Spec, S());
                float smbSpecFastHistory = FetchHistoryBilinearScalar(smbFilter, *gHistory_SpecFast);

                // Surface motion confidence
                float surfaceHistoryConfidence;
                {
                    float a = atan(Div(smbParallaxInPixelsMax * pixelSize, length(X)));
                    float nonLinearAccumSpeed = Rcp(1.0f + smbSpecAccumSpeed);
                    float h = lerp(ExtractHitDist(smbSpecHistory), ExtractHitDist(spec), nonLinearAccumSpeed) * hitDistNormalization;

                    float tana0 = ImportanceSampling::GetSpecularLobeTanHalfAngle(roughnessModified, NRD_MAX_PERCENT_OF_LOBE_VOLUME);
                    tana0 *= lerp(NoV, 1.0f, roughnessModified);
                    tana0 *= nonLinearAccumSpeed;
                    tana0 = Div(tana0, GetHitDistFactor(h, frustumSize) + NRD_EPS);

                    float a0 = atan(tana0);
                    a0 = max(a0, NRD_NORMAL_ENCODING_ERROR);

                    float f = Math::LinearStep(a0, 0.0f, a);
                    surfaceHistoryConfidence = Math::Pow01(f, 4.0f);
                }

                // Responsive accumulation
                float2 maxResponsiveFrameNum;
                {
                    float responsiveFactor = RemapRoughnessToResponsiveFactor(c, roughness);
                    float smc = GetSpecMagicCurve(roughnessModified);
                    float2 f = float2(dot(N, normalize(smbNavg)), dot(N, vmbN));
                    float e = lerp(32.0f, 1.0f, smc) * (1.0f - responsiveFactor);
                    f = lerp(smc, 1.0f, responsiveFactor) * float2(Math::Pow01(f.x, e), Math::Pow01(f.y, e));
                    maxResponsiveFrameNum = max(c.gMaxAccumulatedFrameNum * f, float2(c.gHistoryFixFrameNum));
                }

                // Surface motion: max allowed frames
                float smbMaxFrameNum = c.gMaxAccumulatedFrameNum;
                smbMaxFrameNum *= surfaceHistoryConfidence;
                smbMaxFrameNum = min(smbMaxFrameNum, maxResponsiveFrameNum.x);

                float smbBoostedMaxFrameNum = max(smbMaxFrameNum, c.gHistoryFixFrameNum * (1.0f - virtualHistoryConfidenceForSmbRelaxation));
                float smbSpecAccumSpeedBoosted = min(smbSpecAccumSpeed, smbBoostedMaxFrameNum);

                // Virtual motion: max allowed frames
                float vmbMaxFrameNum = c.gMaxAccumulatedFrameNum;
                vmbMaxFrameNum *= virtualHistoryConfidence;
                vmbMaxFrameNum = min(vmbMaxFrameNum, maxResponsiveFrameNum.y);

                smbSpecAccumSpeed = min(smbSpecAccumSpeed, smbMaxFrameNum);
                vmbSpecAccumSpeed = min(vmbSpecAccumSpeed, vmbMaxFrameNum);

                // Fallback to "smb" if "vmb" history is short (works in both directions)
                float magic = vmbSpecAccumSpeed > smbSpecAccumSpeed ? 8.0f : 0.5f;
                virtualHistoryAmount *= 1.0f + Div(vmbSpecAccumSpeed - smbSpecAccumSpeed, magic * max(vmbSpecAccumSpeed, smbSpecAccumSpeed) + 1.0f);
                virtualHistoryAmount = saturate(virtualHistoryAmount);

                // Sample virtual history
                HistoryFilter vmbFilter = MakeHistoryFilter(saturate(vmbPixelUv) * c.gRectSizePrev, vmbOcclusionWeights, vmbAllowCatRom);
                S vmbSpecHistory = FetchSignalHistory(vmbFilter, *gHistory_Spec, S());
                float vmbSpecFastHistory = FetchHistoryBilinearScalar(vmbFilter, *gHistory_SpecFast);

                smbSpecHistory = ClampNegativeToZero(smbSpecHistory);
                vmbSpecHistory = ClampNegativeToZero(vmbSpecHistory);

                // Accumulation
                float smbSpecNonLinearAccumSpeed = Rcp(1.0f + smbSpecAccumSpeed);
                float vmbSpecNonLinearAccumSpeed = Rcp(1.0f + vmbSpecAccumSpeed);
                if (!specHasData) {
                    smbSpecNonLinearAccumSpeed *= lerp(1.0f - c.gCheckerboardResolveAccumSpeed, 1.0f, smbSpecNonLinearAccumSpeed);
                    vmbSpecNonLinearAccumSpeed *= lerp(1.0f - c.gCheckerboardResolveAccumSpeed, 1.0f, vmbSpecNonLinearAccumSpeed);
                }

                S smbSpec = MixHistoryAndCurrent(c, smbSpecHistory, spec, smbSpecNonLinearAccumSpeed, roughnessModified);
                S vmbSpec = MixHistoryAndCurrent(c, vmbSpecHistory, spec, vmbSpecNonLinearAccumSpeed, roughnessModified);
                S specResult = lerp(smbSpec, vmbSpec, virtualHistoryAmount);

                float4 specShResult = float4(0.0f);
                if (SH) { // REBLUR_TemporalAccumulation.hlsli:742-751; the SH history is fetched with the custom-weight bilinear filter (REBLUR_Common.hlsli:350-361)
                    float4 smbSpecShHistory = FetchHistoryBilinear(smbFilter, *gHistory_SpecSh);
                    float4 vmbSpecShHistory = FetchHistoryBilinear(vmbFilter, *gHistory_SpecSh);
                    float4 specSh = gIn_SpecSh->Load(px, py);
                    float4 smbShSpec = lerp(smbSpecShHistory, specSh, smbSpecNonLinearAccumSpeed);
                    float4 vmbShSpec = lerp(vmbSpecShHistory, specSh, vmbSpecNonLinearAccumSpeed);
                    specShResult = lerp(smbShSpec, vmbShSpec, virtualHistoryAmount);
                    specShResult.w = roughnessModified; // assists AA during the SG resolve; never blurred
                }

                specAccumSpeed = lerp(smbSpecAccumSpeedBoosted, vmbSpecAccumSpeed, virtualHistoryAmount);
                S specHistory = lerp(smbSpecHistory, vmbSpecHistory, virtualHistoryAmount);

                // Firefly suppressor (not in the occlusion family: REBLUR_TemporalAccumulation.hlsli:757, 788)
                float specMaxRelativeIntensity = 0.0f, specAntifireflyFactor = 0.0f;
                if (KIND == SIGNAL_RADIANCE) {
                    specMaxRelativeIntensity = c.gFireflySuppressorMinRelativeScale + Div(REBLUR_FIREFLY_SUPPRESSOR_MAX_RELATIVE_INTENSITY, specAccumSpeed + 1.0f);
                    specAntifireflyFactor = specAccumSpeed * c.gMaxBlurRadius * REBLUR_FIREFLY_SUPPRESSOR_RADIUS_SCALE;
                    specAntifireflyFactor = Div(specAntifireflyFactor, 1.0f + specAntifireflyFactor);

                    float specLumaResult = GetLuma(specResult);
                    float specLumaClamped = min(specLumaResult, GetLuma(specHistory) * specMaxRelativeIntensity);
                    specLumaClamped = lerp(specLumaResult, specLumaClamped, specAntifireflyFactor);
                    specResult = ChangeLuma(specResult, specLumaClamped);
                    if (SH) {
                        float k = GetLumaScale(length(specShResult.xyz()), specLumaClamped);
                        specShResult.x *= k, specShResult.y *= k, specShResult.z *= k;
                    }
                }

                gOut_Spec->Store(px, py, specResult);
                if (SH)
                    gOut_SpecSh->Store(px, py, specShResult);

                // Fast history
                float smbSpecFastNonLinearAccumSpeed = GetNonLinearAccumSpeed(c, smbSpecAccumSpeed, c.gMaxFastAccumulatedFrameNum, surfaceHistoryConfidence, specHasData);
                float vmbSpecFastNonLinearAccumSpeed = GetNonLinearAccumSpeed(c, vmbSpecAccumSpeed, c.gMaxFastAccumulatedFrameNum, virtualHistoryConfidence, specHasData);
                float smbSpecFast = lerp(smbSpecFastHistory, GetLuma(spec), smbSpecFastNonLinearAccumSpeed);
                float vmbSpecFast = lerp(vmbSpecFastHistory, GetLuma(spec), vmbSpecFastNonLinearAccumSpeed);
                float specFastResult = lerp(smbSpecFast, vmbSpecFast, virtualHistoryAmount);

                if (KIND == SIGNAL_RADIANCE) {
                    float specFastClamped = min(specFastResult, GetLuma(specHistory) * specMaxRelativeIntensity * REBLUR_FIREFLY_SUPPRESSOR_FAST_RELATIVE_INTENSITY);
                    specFastResult = lerp(specFastResult, specFastClamped, specAntifireflyFactor);
                }
                gOut_SpecFast->Store(px, py, specFastResult);
            }

            // Output: 4+4 occlusion bits, curvature, virtual history amount (R32_UINT, or the low byte only in R8_UINT)
            if (!OCC)
                gOut_Data2->StoreUint(px, py, PackData2(fbits, curvature, virtualHistoryAmount));

            // ---------------------------------------------------------------------------------------------- diffuse
            if (DIFF) {
                float diffHistoryConfidence = smbFootprintQuality;
                if (c.gHasHistoryConfidence)
                    diffHistoryConfidence *= gIn_DiffConfidence->Load((int)c.gRectOrigin[0] + px, (int)c.gRectOrigin[1] + py).x;
                diffAccumSpeed *= lerp(diffHistoryConfidence, 1.0f, Rcp(1.0f + diffAccumSpeed));
                diffAccumSpeed = min(diffAccumSpeed, c.gMaxAccumulatedFrameNum);

                S diff = Sig::From(gIn_Diff->Load((OCC && c.gDiffCheckerboard != 2) ? px >> 1 : px, py));
                if (OCC && !diffHasData) {
                    S d0 = Sig::From(gIn_Diff->Load(cbX0, py)), d1 = Sig::From(gIn_Diff->Load(cbX1, py));
                    d0 = wc.x == 0.0f ? S(0.0f) : d0;
                    d1 = wc.y == 0.0f ? S(0.0f) : d1;
                    diff = d0 * wc.x + d1 * wc.y;
                }

                HistoryFilter smbFilter = MakeHistoryFilter(smbSamplePos, smbOcclusionWeights, smbAllowCatRom);
                S smbDiffHistory = FetchSignalHistory(smbFilter, *gHistory_Diff, S());
                float smbDiffFastHistory = FetchHistoryBilinearScalar(smbFilter, *gHistory_DiffFast);
                smbDiffHistory = ClampNegativeToZero(smbDiffHistory);

                float diffNonLinearAccumSpeed = Rcp(1.0f + diffAccumSpeed);
                if (!diffHasData)
                    diffNonLinearAccumSpeed *= lerp(1.0f - c.gCheckerboardResolveAccumSpeed, 1.0f, diffNonLinearAccumSpeed);
                S diffResult = MixHistoryAndCurrent(c, smbDiffHistory, diff, diffNonLinearAccumSpeed);
                float4 diffShResult = float4(0.0f);
                if (SH) { // REBLUR_TemporalAccumulation.hlsli:883-886
                    float4 smbDiffShHistory = FetchHistoryBilinear(smbFilter, *gHistory_DiffSh);
                    diffShResult = MixHistoryAndCurrent(c, smbDiffShHistory, gIn_DiffSh->Load(px, py), diffNonLinearAccumSpeed);
                }

                // Firefly suppressor (not in the occlusion family: REBLUR_TemporalAccumulation.hlsli:889, 918)
                float diffMaxRelativeIntensity = 0.0f, diffAntifireflyFactor = 0.0f;
                if (KIND == SIGNAL_RADIANCE) {
                    diffMaxRelativeIntensity = c.gFireflySuppressorMinRelativeScale + Div(REBLUR_FIREFLY_SUPPRESSOR_MAX_RELATIVE_INTENSITY, diffAccumSpeed + 1.0f);
                    diffAntifireflyFactor = diffAccumSpeed * c.gMaxBlurRadius * REBLUR_FIREFLY_SUPPRESSOR_RADIUS_SCALE;
                    diffAntifireflyFactor = Div(diffAntifireflyFactor, 1.0f + diffAntifireflyFactor);

                    float diffLumaResult = GetLuma(diffResult);
                    float diffLumaClamped = min(diffLumaResult, GetLuma(smbDiffHistory) * diffMaxRelativeIntensity);
                    diffLumaClamped = lerp(diffLumaResult, diffLumaClamped, diffAntifireflyFactor);
                    diffResult = ChangeLuma(diffResult, diffLumaClamped);
                    if (SH) {
                        float k = GetLumaScale(length(diffShResult.xyz()), diffLumaClamped);
                        diffShResult.x *= k, diffShResult.y *= k, diffShResult.z *= k;
                    }
                }
                gOut_Diff->Store(px, py, diffResult);
                if (SH)
                    gOut_DiffSh->Store(px, py, diffShResult);

                // Fast history
                float diffFastAccumSpeed = min(diffAccumSpeed, c.gMaxFastAccumulatedFrameNum);
                float diffFastNonLinearAccumSpeed = Rcp(1.0f + diffFastAccumSpeed);
                if (!diffHasData)
                    diffFastNonLinearAccumSpeed *= lerp(1.0f - c.gCheckerboardResolveAccumSpeed, 1.0f, diffFastNonLinearAccumSpeed);
                float diffFastResult = lerp(smbDiffFastHistory, GetLuma(diff), diffFastNonLinearAccumSpeed);
                if (KIND == SIGNAL_RADIANCE) {
                    float diffFastClamped = min(diffFastResult, GetLuma(smbDiffHistory) * diffMaxRelativeIntensity * REBLUR_FIREFLY_SUPPRESSOR_FAST_RELATIVE_INTENSITY);
                    diffFastResult = lerp(diffFastResult, diffFastClamped, diffAntifireflyFactor);
                }
                gOut_DiffFast->Store(px, py, diffFastResult);
            }

            float2 d1 = PackData1(diffAccumSpeed, specAccumSpeed, DIFF);
            gOut_Data1.Store(px, py, float4(d1.x, d1.y, 0.0f, 0.0f));
        }
}

// ================================================================================================ HistoryFix
// one signal (diffuse or specular) of the history-fix pass; returns the fixed signal and writes the fast history
template <typename S>
S HistoryFixSignal(const ReblurCB& c, bool isSpec, bool perf, int px, int py, S sig, float frameNum, float strideBase, float roughness, float viewZ, float materialID,
    float3 N, float3 Nv, float3 Xv, float2 pixelUv, float frustumSize, const Tex& gIn_ViewZ, const Tex& gIn_Normal_Roughness, const Tex& gIn_Data1, bool hasDiff,
    const Tex& gIn_Signal, const Tex& gIn_Fast, Tex& gOut_Fast, float4* sh = nullptr, const Tex* gIn_Sh = nullptr) { // REBLUR_SH: SH1 plane rides along (specular: .xyz only)
    constexpr int KIND = SignalKind<S>::value;
    constexpr bool OCC = KIND == SIGNAL_OCCLUSION;
    typedef ReblurSignal<KIND> Sig;
    const int rw = c.gRectSizeMinusOne[0], rh = c.gRectSizeMinusOne[1];
    float smc = GetSpecMagicCurve(roughness);

    // Stride between taps
    float stride = strideBase * (frameNum < c.gHistoryFixFrameNum ? 1.0f : 0.0f);
    if (isSpec)
        stride *= lerp(0.5f, 1.0f, smc);
    stride = floorf(stride);

    // History reconstruction: 5x5 minus centre minus corners, sparse
    if (stride != 0.0f) {
        int stridei = (int)(stride + 0.5f);
        float nonLinearAccumSpeed = Rcp(1.0f + frameNum);
        float r = isSpec ? roughness : 1.0f;

        float normalWeightParam = GetNormalWeightParam(nonLinearAccumSpeed, c.gLobeAngleFraction, r);
        float2 geometryWeightParams = GetGeometryWeightParams(c.gPlaneDistSensitivity, frustumSize, Xv, Nv);
        float2 relaxedRoughnessWeightParams = GetRelaxedRoughnessWeightParams(roughness * roughness, HwSqrt(c.gRoughnessFraction));

        float hitDistScale = _REBLUR_GetHitDistanceNormalization(viewZ, c.gHitDistParams, r);
        float hitDist = ExtractHitDist(sig) * hitDistScale;
        float hitDistFactor = GetHitDistFactor(hitDist, frustumSize);
        float2 hitDistanceWeightParams = GetHitDistanceWeightParams(hitDistFactor, nonLinearAccumSpeed, r);

        float sumw = 1.0f + frameNum;
        if (perf) // REBLUR_HistoryFix.hlsli:88-90 / 292-294
            sumw = 1.0f + Rcp(1.0f + c.gMaxAccumulatedFrameNum) - nonLinearAccumSpeed;
        sig = sig * sumw;
        if (sh) {
            sh->x *= sumw, sh->y *= sumw, sh->z *= sumw;
            if (!isSpec)
                sh->w *= sumw;
        }

        for (int j = -2; j <= 2; j++)
            for (int i = -2; i <= 2; i++) {
                if ((i == 0 && j == 0) || (::abs(i) + ::abs(j) == 4))
                    continue;

                float2 uv = pixelUv + float2(float(i), float(j)) * stride * c.gRectSizeInv;
                int sx = clamp(px + i * stridei, 0, rw), sy = clamp(py + j * stridei, 0, rh);

                float zs = UnpackViewZ(c, gIn_ViewZ.Load(sx, sy).x);
                float materialIDs;
                float4 Ns = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.Load(sx, sy), materialIDs);

                float angle = Math::AcosApprox(dot(Ns.xyz(), N));
                float3 Xvs = Geometry::ReconstructViewPosition(uv, c.gFrustum, zs, c.gOrthoMode);

                float w = IsInScreenNearest(uv);
                w *= ComputeWeight(dot(Nv, Xvs), geometryWeightParams.x, geometryWeightParams.y);
                w *= CompareMaterials(materialID, materialIDs, isSpec ? c.gSpecMinMaterial : c.gDiffMinMaterial) ? 1.0f : 0.0f;
                w *= ComputeExponentialWeight(angle, normalWeightParam, 0.0f);
                if (isSpec)
                    w *= ComputeExponentialWeight(Ns.w * Ns.w, relaxedRoughnessWeightParams.x, relaxedRoughnessWeightParams.y);

                if (!perf) {
                    float2 d1 = UnpackData1(gIn_Data1.Load(sx, sy), hasDiff);
                    w *= 1.0f + (isSpec ? d1.y : d1.x);
                }

                S smp = Sig::From(gIn_Signal.Load(sx, sy));
                smp = w == 0.0f ? S(0.0f) : smp;

                float hs = ExtractHitDist(smp) * hitDistScale;
                float hsFactor = GetHitDistFactor(hs, frustumSize);
                w *= ComputeExponentialWeight(hsFactor, hitDistanceWeightParams.x, hitDistanceWeightParams.y);

                if (isSpec) { // low roughness: hit distances work as a non-noisy guide
                    float d = Div(fabsf(hitDist - hs), max(hitDist, hs) + 0.001f);
                    float b = Math::LinearStep(0.03f, 0.05f, roughness);
                    w *= Math::SmoothStep(0.2f + b, 0.05f + b, d);
                }

                sumw += w;
                sig = Mad(smp, w, sig);
                if (sh) {
                    float4 t = gIn_Sh->Load(sx, sy);
                    t = w == 0.0f ? float4(0.0f) : t;
                    sh->x += t.x * w, sh->y += t.y * w, sh->z += t.z * w;
                    if (!isSpec)
                        sh->w += t.w * w;
                }
            }

        sumw = Math::PositiveRcp(sumw);
        sig = sig * sumw;
        if (sh) {
            sh->x *= sumw, sh->y *= sumw, sh->z *= sumw;
            if (!isSpec)
                sh->w *= sumw;
        }
    }

    // Local variance of the fast history over 5x5 (clamped reads = the shader's LDS preload)
    auto sLuma = [&](int x, int y) { return gIn_Fast.Load(clamp(x, 0, rw), clamp(y, 0, rh)).x; };
    float center = sLuma(px, py);
    float m1 = center, m2 = center * center;

    float f = saturate(Div(frameNum, c.gHistoryFixFrameNum + NRD_EPS));
    if (isSpec)
        f = lerp(1.0f, f, smc);
    center = lerp(GetLuma(sig), center, f);
    gOut_Fast.Store(px, py, center);

    for (int j = 0; j <= 4; j++)
        for (int i = 0; i <= 4; i++) {
            if (i == 2 && j == 2)
                continue;
            float d = sLuma(px - 2 + i, py - 2 + j);
            m1 += d;
            m2 += d * d;
        }

    float luma = GetLuma(sig);

    // Anti-firefly: 9x9 minus the central 3x3 (REBLUR_USE_ANTIFIREFLY = 0 in the occlusion family)
    if (c.gAntiFirefly != 0.0f && KIND == SIGNAL_RADIANCE) {
        float am1 = 0.0f, am2 = 0.0f;
        const int R = perf ? 3 : REBLUR_ANTI_FIREFLY_FILTER_RADIUS; // REBLUR_Config.hlsli:236-237
        for (int j = -R; j <= R; j++)
            for (int i = -R; i <= R; i++) {
                if (::abs(i) <= 1 && ::abs(j) <= 1)
                    continue;
                float d = gIn_Fast.Load(clamp(px + i, 0, rw), clamp(py + j, 0, rh)).x;
                am1 += d;
                am2 += d * d;
            }
        float invNorm = Rcp(float((R * 2 + 1) * (R * 2 + 1) - 3 * 3));
        am1 *= invNorm;
        am2 *= invNorm;
        float sigma = HwSqrt(fabsf(am2 - am1 * am1)) * REBLUR_ANTI_FIREFLY_SIGMA_SCALE;
        luma = clamp(luma, am1 - sigma, am1 + sigma);
    }

    // Fast-history clamping
    m1 = DivConst(m1, 25.0f);
    m2 = DivConst(m2, 25.0f);
    float sigma = HwSqrt(fabsf(m2 - m1 * m1)) * (KIND != SIGNAL_RADIANCE ? REBLUR_COLOR_CLAMPING_SIGMA_SCALE_OCCLUSION : REBLUR_COLOR_CLAMPING_SIGMA_SCALE);
    float lumaClamped = clamp(luma, m1 - sigma, m1 + sigma);
    luma = lerp(lumaClamped, luma, Rcp(1.0f + (c.gMaxFastAccumulatedFrameNum < c.gMaxAccumulatedFrameNum ? 1.0f : 0.0f) * frameNum * 2.0f));

    if (sh) { // REBLUR_HistoryFix.hlsli:247-249
        float k = GetLumaScale(length(sh->xyz()), luma);
        sh->x *= k, sh->y *= k, sh->z *= k;
    }
    return ChangeLuma(sig, luma);
}

template <bool DIFF, bool SPEC, bool PERF, int KIND, bool SH>
void HistoryFix(const PassIO& io) {
    typedef ReblurSignal<KIND> Sig;
    constexpr bool OCC = KIND == SIGNAL_OCCLUSION;
    typedef typename Sig::type S;
    const ReblurCB& c = *(const ReblurCB*)io.constants;
    Cursor cur(io);
    const Tex& gIn_Tiles = *cur.next();
    const Tex& gIn_Normal_Roughness = *cur.next();
    const Tex& gIn_Data1 = *cur.next();
    const Tex& gIn_ViewZ = *cur.next();
    const Tex* gIn_Diff = cur.nextIf(DIFF);
    const Tex* gIn_Spec = cur.nextIf(SPEC);
    const Tex* gIn_DiffFast = cur.nextIf(DIFF);
    const Tex* gIn_SpecFast = cur.nextIf(SPEC);
    const Tex* gIn_DiffSh = cur.nextIf(DIFF && SH);
    const Tex* gIn_SpecSh = cur.nextIf(SPEC && SH);
    Tex* gOut_Diff = cur.nextIf(DIFF);
    Tex* gOut_Spec = cur.nextIf(SPEC);
    Tex* gOut_DiffFast = cur.nextIf(DIFF);
    Tex* gOut_SpecFast = cur.nextIf(SPEC);
    Tex* gOut_DiffSh = cur.nextIf(DIFF && SH);
    Tex* gOut_SpecSh = cur.nextIf(SPEC && SH);

#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py <= c.gRectSizeMinusOne[1]; py++)
        for (int px = 0; px <= c.gRectSizeMinusOne[0]; px++) {
            float isSky = gIn_Tiles.Load(px >> 4, py >> 4).x;
            if (isSky != 0.0f)
                continue;
            float viewZ = UnpackViewZ(c, gIn_ViewZ.Load(px, py).x);
            if (viewZ > c.gDenoisingRange)
                continue;

            float materialID;
            float4 normalAndRoughness = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.Load(px, py), materialID);
            float3 N = normalAndRoughness.xyz();
            float roughness = normalAndRoughness.w;

            float frustumSize = GetFrustumSize(c.gMinRectDimMulUnproject, c.gOrthoMode, viewZ);
            float2 pixelUv = float2(float(px) + 0.5f, float(py) + 0.5f) * c.gRectSizeInv;
            float3 Xv = Geometry::ReconstructViewPosition(pixelUv, c.gFrustum, viewZ, c.gOrthoMode);
            float3 Nv = Geometry::RotateVectorInverse(c.gViewToWorld, N);
            float2 frameNum = UnpackData1(gIn_Data1.Load(px, py), DIFF);
            float2 stride = Div(c.gHistoryFixBasePixelStride, 2.0f + frameNum);

            if (DIFF) {
                float4 diffSh = SH ? gIn_DiffSh->Load(px, py) : float4(0.0f);
                S diff = HistoryFixSignal<S>(c, false, PERF, px, py, Sig::From(gIn_Diff->Load(px, py)), frameNum.x, stride.x, roughness, viewZ, materialID, N, Nv, Xv, pixelUv, frustumSize,
                    gIn_ViewZ, gIn_Normal_Roughness, gIn_Data1, DIFF, *gIn_Diff, *gIn_DiffFast, *gOut_DiffFast, SH ? &diffSh : nullptr, gIn_DiffSh);
                gOut_Diff->Store(px, py, diff);
                if (SH)
                    gOut_DiffSh->Store(px, py, diffSh);
            }
            if (SPEC) {
                float4 specSh = SH ? gIn_SpecSh->Load(px, py) : float4(0.0f);
                S spec = HistoryFixSignal<S>(c, true, PERF, px, py, Sig::From(gIn_Spec->Load(px, py)), frameNum.y, stride.y, roughness, viewZ, materialID, N, Nv, Xv, pixelUv, frustumSize,
                    gIn_ViewZ, gIn_Normal_Roughness, gIn_Data1, DIFF, *gIn_Spec, *gIn_SpecFast, *gOut_SpecFast, SH ? &specSh : nullptr, gIn_SpecSh);
                gOut_Spec->Store(px, py, spec);
                if (SH)
                    gOut_SpecSh->Store(px, py, specSh);
            }
        }
}

// ================================================================================================ TemporalStabilization
template <bool DIFF, bool SPEC, bool PERF, bool SH, int KIND>
void TemporalStabilization(const PassIO& io) {
    typedef ReblurSignal<KIND> Sig;
    typedef typename Sig::type S;
    const ReblurCB& c = *(const ReblurCB*)io.constants;
    Cursor cur(io);
    const Tex& gIn_Tiles = *cur.next();
    const Tex& gIn_Normal_Roughness = *cur.next();
    const Tex* gIn_BaseColor_Metalness = cur.nextIf(SPEC); // a dummy unless CommonSettings::isBaseColorMetalnessAvailable
    const Tex& gIn_ViewZ = *cur.next(); // PREV_VIEWZ (already holds this frame's viewZ)
    const Tex& gIn_Data1 = *cur.next();
    const Tex& gIn_Data2 = *cur.next();
    const Tex* gIn_Diff = cur.nextIf(DIFF);
    const Tex* gIn_Spec = cur.nextIf(SPEC);
    const Tex* gHistory_DiffLumaStabilized = cur.nextIf(DIFF);
    const Tex* gHistory_SpecLumaStabilized = cur.nextIf(SPEC);
    const Tex* gIn_SpecHitDistForTracking = cur.nextIf(SPEC);
    const Tex* gIn_DiffSh = cur.nextIf(DIFF && SH);
    const Tex* gIn_SpecSh = cur.nextIf(SPEC && SH);
    Tex& gInOut_Mv = *cur.next();
    Tex& gOut_InternalData = *cur.next();
    Tex* gOut_Diff = cur.nextIf(DIFF);
    Tex* gOut_Spec = cur.nextIf(SPEC);
    Tex* gOut_DiffLumaStabilized = cur.nextIf(DIFF);
    Tex* gOut_SpecLumaStabilized = cur.nextIf(SPEC);
    Tex* gOut_DiffSh = cur.nextIf(DIFF && SH);
    Tex* gOut_SpecSh = cur.nextIf(SPEC && SH);

    const int rw = c.gRectSizeMinusOne[0], rh = c.gRectSizeMinusOne[1];

#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py <= rh; py++)
        for (int px = 0; px <= rw; px++) {
            float isSky = gIn_Tiles.Load(px >> 4, py >> 4).x;
            if (isSky != 0.0f)
                continue;
            float viewZ = UnpackViewZ(c, gIn_ViewZ.Load(px, py).x);
            if (viewZ > c.gDenoisingRange)
                continue;

            // Position
            float2 pixelUv = float2(float(px) + 0.5f, float(py) + 0.5f) * c.gRectSizeInv;
            float3 Xv = Geometry::ReconstructViewPosition(pixelUv, c.gFrustum, viewZ, c.gOrthoMode);
            float3 X = Geometry::RotateVector(c.gViewToWorld, Xv);

            // Previous position and surface motion uv
            float4 inMv = gInOut_Mv.Load(px, py);
            float3 mv = float3(inMv.x, inMv.y, inMv.z) * c.gMvScale.xyz();
            float3 Xprev = X;
            float2 smbPixelUv = pixelUv + float2(mv.x, mv.y);
            if (c.gMvScale.w == 0.0f) {
                if (c.gMvScale.z == 0.0f)
                    mv.z = Geometry::AffineTransform(c.gWorldToViewPrev, X).z - viewZ;
                float viewZprev = viewZ + mv.z;
                float3 Xvprevlocal = Geometry::ReconstructViewPosition(smbPixelUv, c.gFrustumPrev, viewZprev, c.gOrthoMode);
                Xprev = Geometry::RotateVectorInverse(c.gWorldToViewPrev, Xvprevlocal) + c.gCameraDelta.xyz();
            } else {
                Xprev += mv;
                smbPixelUv = Geometry::GetScreenUv(c.gWorldToClipPrev, Xprev);
            }

            float materialID;
            float4 normalAndRoughness = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.Load(px, py), materialID);
            float3 N = normalAndRoughness.xyz();
            float roughness = normalAndRoughness.w;

            uint32_t bits;
            float2 data1 = UnpackData1(gIn_Data1.Load(px, py), DIFF);
            float2 data2 = UnpackData2(gIn_Data2.LoadUint(px, py), bits);

            // Surface motion footprint
            Filtering::Bilinear smbBilinearFilter = Filtering::GetBilinearFilter(smbPixelUv, c.gRectSizePrev);
            float4 smbOcclusion = float4((bits & 1u) ? 1.0f : 0.0f, (bits & 2u) ? 1.0f : 0.0f, (bits & 4u) ? 1.0f : 0.0f, (bits & 8u) ? 1.0f : 0.0f);
            float4 smbOcclusionWeights = Filtering::GetBilinearCustomWeights(smbBilinearFilter, smbOcclusion);
            bool smbAllowCatRom = sum(smbOcclusion) > 3.5f && !PERF; // REBLUR_USE_CATROM_FOR_SURFACE_MOTION_IN_TS
            float smbFootprintQuality = Filtering::ApplyBilinearFilter(smbOcclusion.x, smbOcclusion.y, smbOcclusion.z, smbOcclusion.w, smbBilinearFilter);
            smbFootprintQuality = Math::Sqrt01(smbFootprintQuality);

            const float2 smbSamplePos = saturate(smbPixelUv) * c.gRectSizePrev;

            // 3x3 luma statistics (clamped reads = LDS preload)
            auto stats = [&](const Tex& tex, float& luma, float& m1, float& sigma) {
                auto sL = [&](int x, int y) { return GetLuma(Sig::From(tex.Load(clamp(x, 0, rw), clamp(y, 0, rh)))); };
                luma = sL(px, py);
                float M1 = luma, M2 = luma * luma, mn = NRD_INF, mx = -NRD_INF;
                for (int j = 0; j <= 2; j++)
                    for (int i = 0; i <= 2; i++) {
                        if (i == 1 && j == 1)
                            continue;
                        float d = sL(px - 1 + i, py - 1 + j);
                        M1 += d;
                        M2 += d * d;
                        mn = min(mn, d);
                        mx = max(mx, d);
                    }
                M1 = DivConst(M1, 9.0f);
                M2 = DivConst(M2, 9.0f);
                m1 = M1;
                sigma = HwSqrt(fabsf(M2 - M1 * M1));
                if (!PERF && c.gMaxBlurRadius != 0.0f) // RCRS (not in performance mode)
                    luma = clamp(luma, mn, mx);
            };

            if (DIFF) {
                float diffLuma, diffLumaM1, diffLumaSigma;
                stats(*gIn_Diff, diffLuma, diffLumaM1, diffLumaSigma);

                HistoryFilter smbFilter = MakeHistoryFilter(smbSamplePos, smbOcclusionWeights, smbAllowCatRom);
                float smbDiffLumaHistory = FetchHistoryScalar(smbFilter, *gHistory_DiffLumaStabilized);
                smbDiffLumaHistory = max(smbDiffLumaHistory, 0.0f);

                float diffAntilag = ComputeAntilag(c, smbDiffLumaHistory, diffLumaM1, diffLumaSigma, smbFootprintQuality * data1.x);

                float2 diffTemporalAccumulationParams = GetTemporalAccumulationParams(c, smbFootprintQuality, data1.x);
                float diffHistoryWeight = diffTemporalAccumulationParams.x;
                diffHistoryWeight *= diffAntilag;
                diffHistoryWeight *= pixelUv.x >= c.gSplitScreen ? 1.0f : 0.0f;
                diffHistoryWeight *= smbPixelUv.x >= c.gSplitScreenPrev ? 1.0f : 0.0f;

                smbDiffLumaHistory = Color::Clamp(diffLumaM1, diffLumaSigma * diffTemporalAccumulationParams.y, smbDiffLumaHistory);
                float diffLumaStabilized = lerp(diffLuma, smbDiffLumaHistory, min(diffHistoryWeight, c.gStabilizationStrength));

                S diff = Sig::From(gIn_Diff->Load(px, py));
                diff = ChangeLuma(diff, diffLumaStabilized);
                gOut_Diff->Store(px, py, diff);
                gOut_DiffLumaStabilized->Store(px, py, diffLumaStabilized);
                if (SH) { // REBLUR_TemporalStabilization.hlsli:166-176
                    float4 diffSh = gIn_DiffSh->Load(px, py);
                    float k = GetLumaScale(length(diffSh.xyz()), diffLumaStabilized);
                    gOut_DiffSh->Store(px, py, float4(diffSh.x * k, diffSh.y * k, diffSh.z * k, diffSh.w));
                }

                data1.x += 1.0f;
                float diffMinAccumSpeed = min(data1.x, c.gHistoryFixFrameNum);
                data1.x = lerp(diffMinAccumSpeed, data1.x, diffAntilag);
            }

            if (SPEC) {
                float specLuma, specLumaM1, specLumaSigma;
                stats(*gIn_Spec, specLuma, specLumaM1, specLumaSigma);

                float virtualHistoryAmount = data2.x;
                float curvature = data2.y;

                S spec = Sig::From(gIn_Spec->Load(px, py));
                float hitDistForTracking = ExtractHitDist(spec) * _REBLUR_GetHitDistanceNormalization(viewZ, c.gHitDistParams, roughness);
                if (c.gSpecPrepassBlurRadius != 0.0f)
                    hitDistForTracking = min(hitDistForTracking, gIn_SpecHitDistForTracking->Load(px, py).x);

                // Virtual motion
                float3 V = GetViewVector(c, X);
                float3 Xvirtual = GetXvirtual(hitDistForTracking, curvature, X, Xprev, N, V, roughness);
                float2 vmbPixelUv = Geometry::GetScreenUv(c.gWorldToClipPrev, Xvirtual);
                vmbPixelUv = materialID == c.gCameraAttachedReflectionMaterialID ? pixelUv : vmbPixelUv;

                // Modify MVs if requested (REBLUR_TemporalStabilization.hlsli:250-285): where the surface is mostly specular, IN_MV is bent towards
                // the motion of the reflected world (x = 2 without IN_BASECOLOR_METALNESS: off)
                if (c.gSpecProbabilityThresholdsForMvModification.x < 1.0f) {
                    float NoV = fabsf(dot(N, V));
                    float4 baseColorMetalness = gIn_BaseColor_Metalness->Load((int)c.gRectOrigin[0] + px, (int)c.gRectOrigin[1] + py);
                    float3 albedo, Rf0;
                    Color::ConvertBaseColorMetalnessToAlbedoRf0(baseColorMetalness.xyz(), baseColorMetalness.w, albedo, Rf0);
                    float3 Fenv = Color::EnvironmentTerm_Rtg(Rf0, NoV, roughness);
                    float lumSpec = Color::Luminance(Fenv);
                    float lumDiff = Color::Luminance(float3(albedo.x * (1.0f - Fenv.x), albedo.y * (1.0f - Fenv.y), albedo.z * (1.0f - Fenv.z)));
                    float specProb = Div(lumSpec, lumDiff + lumSpec + NRD_EPS);
                    float f = Math::SmoothStep(c.gSpecProbabilityThresholdsForMvModification.x, c.gSpecProbabilityThresholdsForMvModification.y, specProb);
                    f *= 1.0f - GetSpecMagicCurve(roughness);
                    f *= 1.0f - Math::Sqrt01(fabsf(curvature));
                    if (f != 0.0f) {
                        float3 specMv = Xvirtual - X; // world-space delta
                        if (c.gMvScale.w == 0.0f) {
                            specMv.x = vmbPixelUv.x - pixelUv.x;
                            specMv.y = vmbPixelUv.y - pixelUv.y;
                            specMv.z = Geometry::AffineTransform(c.gWorldToViewPrev, Xvirtual).z - viewZ;
                        }
                        // only .xy for 2D, .xyz for 2.5D and 3D MVs
                        float3 newMv = float3(Div(specMv.x, c.gMvScale.x), Div(specMv.y, c.gMvScale.y), c.gMvScale.z == 0.0f ? inMv.z : Div(specMv.z, c.gMvScale.z));
                        inMv.x = lerp(inMv.x, newMv.x, f);
                        inMv.y = lerp(inMv.y, newMv.y, f);
                        inMv.z = lerp(inMv.z, newMv.z, f);
                        gInOut_Mv.Store((int)c.gRectOrigin[0] + px, (int)c.gRectOrigin[1] + py, inMv);
                    }
                }

                HistoryFilter smbFilter = MakeHistoryFilter(smbSamplePos, smbOcclusionWeights, smbAllowCatRom);
                float smbSpecLumaHistory = FetchHistoryScalar(smbFilter, *gHistory_SpecLumaStabilized);

                // Virtual motion footprint
                Filtering::Bilinear vmbBilinearFilter = Filtering::GetBilinearFilter(vmbPixelUv, c.gRectSizePrev);
                float4 vmbOcclusion = float4((bits & 16u) ? 1.0f : 0.0f, (bits & 32u) ? 1.0f : 0.0f, (bits & 64u) ? 1.0f : 0.0f, (bits & 128u) ? 1.0f : 0.0f);
                float4 vmbOcclusionWeights = Filtering::GetBilinearCustomWeights(vmbBilinearFilter, vmbOcclusion);
                bool vmbAllowCatRom = sum(vmbOcclusion) > 3.5f && !PERF; // REBLUR_USE_CATROM_FOR_VIRTUAL_MOTION_IN_TS
                float vmbFootprintQuality = Filtering::ApplyBilinearFilter(vmbOcclusion.x, vmbOcclusion.y, vmbOcclusion.z, vmbOcclusion.w, vmbBilinearFilter);
                vmbFootprintQuality = Math::Sqrt01(vmbFootprintQuality);

                HistoryFilter vmbFilter = MakeHistoryFilter(saturate(vmbPixelUv) * c.gRectSizePrev, vmbOcclusionWeights, vmbAllowCatRom);
                float vmbSpecLumaHistory = FetchHistoryScalar(vmbFilter, *gHistory_SpecLumaStabilized);

                smbSpecLumaHistory = max(smbSpecLumaHistory, 0.0f);
                vmbSpecLumaHistory = max(vmbSpecLumaHistory, 0.0f);

                float specLumaHistory = lerp(smbSpecLumaHistory, vmbSpecLumaHistory, virtualHistoryAmount);

                float footprintQuality = lerp(smbFootprintQuality, vmbFootprintQuality, virtualHistoryAmount);
                float specAntilag = ComputeAntilag(c, specLumaHistory, specLumaM1, specLumaSigma, footprintQuality * data1.y);

                float2 specTemporalAccumulationParams = GetTemporalAccumulationParams(c, footprintQuality, data1.y);
                float specHistoryWeight = specTemporalAccumulationParams.x;
                specHistoryWeight *= specAntilag;
                specHistoryWeight *= pixelUv.x >= c.gSplitScreen ? 1.0f : 0.0f;
                specHistoryWeight *= virtualHistoryAmount != 1.0f ? (smbPixelUv.x >= c.gSplitScreenPrev ? 1.0f : 0.0f) : 1.0f;
                specHistoryWeight *= virtualHistoryAmount != 0.0f ? (vmbPixelUv.x >= c.gSplitScreenPrev ? 1.0f : 0.0f) : 1.0f;

                float responsiveFactor = RemapRoughnessToResponsiveFactor(c, roughness);
                float smc = GetSpecMagicCurve(roughness);
                float acceleration = lerp(smc, 1.0f, 0.5f + responsiveFactor * 0.5f);
                specHistoryWeight *= materialID == c.gStrandMaterialID ? 0.5f : acceleration;

                specLumaHistory = Color::Clamp(specLumaM1, specLumaSigma * specTemporalAccumulationParams.y, specLumaHistory);
                float specLumaStabilized = lerp(specLuma, specLumaHistory, min(specHistoryWeight, c.gStabilizationStrength));

                spec = ChangeLuma(spec, specLumaStabilized);
                gOut_Spec->Store(px, py, spec);
                gOut_SpecLumaStabilized->Store(px, py, specLumaStabilized);
                if (SH) { // REBLUR_TemporalStabilization.hlsli:346-356
                    float4 specSh = gIn_SpecSh->Load(px, py);
                    float k = GetLumaScale(length(specSh.xyz()), specLumaStabilized);
                    gOut_SpecSh->Store(px, py, float4(specSh.x * k, specSh.y * k, specSh.z * k, specSh.w));
                }

                data1.y += 1.0f;
                float specMinAccumSpeed = min(data1.y, c.gHistoryFixFrameNum);
                data1.y = lerp(specMinAccumSpeed, data1.y, specAntilag);
            }

            gOut_InternalData.StoreUint(px, py, PackInternalData(data1.x, data1.y, materialID));
        }
}

// ================================================================================================ HitDistReconstruction
// reference Shaders/Include/REBLUR_HitDistReconstruction.hlsli:10-160 (REBLUR_USE_DECOMPRESSED_HIT_DIST_IN_RECONSTRUCTION = 0,
// non-performance mode). BORDER = 1 -> 3x3, 2 -> 5x5 window; the window is read at rect-clamped coordinates like the LDS preload.
template <bool DIFF, bool SPEC, int BORDER, bool PERF, int KIND>
void HitDistReconstruction(const PassIO& io) {
    typedef ReblurSignal<KIND> Sig;
    constexpr bool OCC = KIND == SIGNAL_OCCLUSION;
    const ReblurCB& c = *(const ReblurCB*)io.constants;
    Cursor cur(io);
    const Tex& gIn_Tiles = *cur.next();
    const Tex& gIn_Normal_Roughness = *cur.next();
    const Tex& gIn_ViewZ = *cur.next();
    const Tex* gIn_Diff = cur.nextIf(DIFF);
    const Tex* gIn_Spec = cur.nextIf(SPEC);
    Tex* gOut_Diff = cur.nextIf(DIFF);
    Tex* gOut_Spec = cur.nextIf(SPEC);
    const int rw = c.gRectSizeMinusOne[0], rh = c.gRectSizeMinusOne[1];
    const int ox = (int)c.gRectOrigin[0], oy = (int)c.gRectOrigin[1];

#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py <= rh; py++)
        for (int px = 0; px <= rw; px++) {
            if (gIn_Tiles.Load(px >> 4, py >> 4).x != 0.0f)
                continue;
            auto ViewZ = [&](int x, int y) { return UnpackViewZ(c, gIn_ViewZ.Load(ox + clamp(x, 0, rw), oy + clamp(y, 0, rh)).x); };
            auto NormalRoughness = [&](int x, int y) { return NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.Load(ox + clamp(x, 0, rw), oy + clamp(y, 0, rh))); };
            auto HitDist = [&](int x, int y) {
                x = clamp(x, 0, rw), y = clamp(y, 0, rh);
                return float2(DIFF ? ExtractHitDist(Sig::From(gIn_Diff->Load(x, y))) : 0.0f, SPEC ? ExtractHitDist(Sig::From(gIn_Spec->Load(x, y))) : 0.0f);
            };
            const float centerZ = ViewZ(px, py);
            if (centerZ > c.gDenoisingRange)
                continue;

            float4 normalAndRoughness = NormalRoughness(px, py);
            float3 N = normalAndRoughness.xyz();
            float roughness = normalAndRoughness.w;

            float2 pixelUv = float2(float(px) + 0.5f, float(py) + 0.5f) * c.gRectSizeInv;
            float3 Xv = Geometry::ReconstructViewPosition(pixelUv, c.gFrustum, centerZ, c.gOrthoMode);
            float3 Nv = Geometry::RotateVectorInverse(c.gViewToWorld, N);
            float frustumSize = GetFrustumSize(c.gMinRectDimMulUnproject, c.gOrthoMode, centerZ);

            float2 geometryWeightParams = GetGeometryWeightParams(c.gPlaneDistSensitivity, frustumSize, Xv, Nv);
            float2 relaxedRoughnessWeightParams = GetRelaxedRoughnessWeightParams(roughness * roughness);
            float diffNormalWeightParam = GetNormalWeightParam(1.0f, 1.0f);
            float specNormalWeightParam = GetNormalWeightParam(1.0f, 1.0f, roughness);

            float2 center = HitDist(px, py);
            float2 sum = float2(center.x != 0.0f ? 1000.0f : 0.0f, center.y != 0.0f ? 1000.0f : 0.0f);
            center = center * sum;

            for (int j = 0; j <= BORDER * 2; j++)
                for (int i = 0; i <= BORDER * 2; i++) {
                    float2 o = float2(float(i - BORDER), float(j - BORDER));
                    if (o.x == 0.0f && o.y == 0.0f)
                        continue;
                    int sx = px + i - BORDER, sy = py + j - BORDER;
                    float2 data = HitDist(sx, sy);
                    float dataZ = ViewZ(sx, sy);

                    float w = IsInScreenNearest(pixelUv + o * c.gRectSizeInv);
                    w *= GetGaussianWeight(length(o) * 0.5f);

                    float2 uv = pixelUv + o * c.gRectSizeInv;
                    float3 Xvs = Geometry::ReconstructViewPosition(uv, c.gFrustum, dataZ, c.gOrthoMode);
                    w *= ComputeWeight(dot(Nv, Xvs), geometryWeightParams.x, geometryWeightParams.y);

                    float2 ww = float2(w);
                    if (!PERF) {
                        float4 sampleNormalAndRoughness = NormalRoughness(sx, sy);
                        float cosa = dot(N, sampleNormalAndRoughness.xyz());
                        float angle = Math::AcosApprox(cosa);
                        ww.x *= ComputeExponentialWeight(angle, diffNormalWeightParam, 0.0f);
                        ww.y *= ComputeExponentialWeight(angle, specNormalWeightParam, 0.0f);
                        ww.y *= ComputeExponentialWeight(sampleNormalAndRoughness.w * sampleNormalAndRoughness.w, relaxedRoughnessWeightParams.x, relaxedRoughnessWeightParams.y);
                    }

                    data.x = ww.x == 0.0f ? 0.0f : data.x; // Denanify
                    data.y = ww.y == 0.0f ? 0.0f : data.y;
                    ww = ww * float2(data.x != 0.0f ? 1.0f : 0.0f, data.y != 0.0f ? 1.0f : 0.0f);

                    center = Mad(data, ww, center);
                    sum += ww;
                }
            center = Div(center, max(sum, float2(NRD_EPS)));

            if (DIFF)
                gOut_Diff->Store(px, py, Sig::WithHitDist(Sig::From(gIn_Diff->Load(px, py)), center.x));
            if (SPEC)
                gOut_Spec->Store(px, py, Sig::WithHitDist(Sig::From(gIn_Spec->Load(px, py)), center.y));
        }
}

// ================================================================================================ SplitScreen
template <bool DIFF, bool SPEC, bool SH>
void SplitScreen(const PassIO& io) {
    const ReblurCB& c = *(const ReblurCB*)io.constants;
    Cursor cur(io);
    const Tex& gIn_ViewZ = *cur.next();
    const Tex* gIn_Diff = cur.nextIf(DIFF);
    const Tex* gIn_Spec = cur.nextIf(SPEC);
    const Tex* gIn_DiffSh = cur.nextIf(DIFF && SH);
    const Tex* gIn_SpecSh = cur.nextIf(SPEC && SH);
    Tex* gOut_Diff = cur.nextIf(DIFF);
    Tex* gOut_Spec = cur.nextIf(SPEC);
    Tex* gOut_DiffSh = cur.nextIf(DIFF && SH);
    Tex* gOut_SpecSh = cur.nextIf(SPEC && SH);
#pragma omp parallel for schedule(static)
    for (int py = 0; py <= c.gRectSizeMinusOne[1]; py++)
        for (int px = 0; px <= c.gRectSizeMinusOne[0]; px++) {
            float2 pixelUv = float2(float(px) + 0.5f, float(py) + 0.5f) * c.gRectSizeInv;
            if (pixelUv.x > c.gSplitScreen)
                continue;
            float viewZ = UnpackViewZ(c, gIn_ViewZ.Load(px, py).x);
            float keep = viewZ < c.gDenoisingRange ? 1.0f : 0.0f;
            const int dx = c.gDiffCheckerboard != 2 ? px >> 1 : px, sx = c.gSpecCheckerboard != 2 ? px >> 1 : px; // checkerboarded inputs: left half
            if (DIFF)
                gOut_Diff->Store(px, py, gIn_Diff->Load(dx, py) * keep);
            if (SPEC)
                gOut_Spec->Store(px, py, gIn_Spec->Load(sx, py) * keep);
            if (DIFF && SH)
                gOut_DiffSh->Store(px, py, gIn_DiffSh->Load(dx, py) * keep);
            if (SPEC && SH)
                gOut_SpecSh->Store(px, py, gIn_SpecSh->Load(sx, py) * keep);
        }
}

} // namespace

// quality and performance ("REBLUR_Perf_*", REBLUR_PERFORMANCE_MODE) permutations of one signal family; the SH family ("Sh": an SH1 plane per
// signal rides along) reuses the radiance family's hit-distance reconstruction; the occlusion family has no pre-pass / stabilisation
#define REBLUR_PASSES(PREFIX, NAME, D, S, P)                                                            \
    {PREFIX NAME "_HitDistReconstruction.cs", HitDistReconstruction<D, S, 1, P, false>},               \
    {PREFIX NAME "_HitDistReconstruction_5x5.cs", HitDistReconstruction<D, S, 2, P, false>},           \
    {PREFIX NAME "_PrePass.cs", PrePass<D, S, P, false, 0>},                                              \
    {PREFIX NAME "_TemporalAccumulation.cs", TemporalAccumulation<D, S, P, false, false>},             \
    {PREFIX NAME "_HistoryFix.cs", HistoryFix<D, S, P, false, false>},                                 \
    {PREFIX NAME "_Blur.cs", Blur<D, S, P, false, false>},                                             \
    {PREFIX NAME "_PostBlur.cs", PostBlur<D, S, false, P, false, false>},                              \
    {PREFIX NAME "_PostBlur_NoTemporalStabilization.cs", PostBlur<D, S, true, P, false, false>},       \
    {PREFIX NAME "_TemporalStabilization.cs", TemporalStabilization<D, S, P, false, 0>},                  \
    {PREFIX NAME "Sh_PrePass.cs", PrePass<D, S, P, true, 0>},                                             \
    {PREFIX NAME "Sh_TemporalAccumulation.cs", TemporalAccumulation<D, S, P, false, true>},            \
    {PREFIX NAME "Sh_HistoryFix.cs", HistoryFix<D, S, P, false, true>},                                \
    {PREFIX NAME "Sh_Blur.cs", Blur<D, S, P, false, true>},                                            \
    {PREFIX NAME "Sh_PostBlur.cs", PostBlur<D, S, false, P, false, true>},                             \
    {PREFIX NAME "Sh_PostBlur_NoTemporalStabilization.cs", PostBlur<D, S, true, P, false, true>},      \
    {PREFIX NAME "Sh_TemporalStabilization.cs", TemporalStabilization<D, S, P, true, 0>},                 \
    {PREFIX NAME "Occlusion_HitDistReconstruction.cs", HitDistReconstruction<D, S, 1, P, true>},       \
    {PREFIX NAME "Occlusion_HitDistReconstruction_5x5.cs", HitDistReconstruction<D, S, 2, P, true>},   \
    {PREFIX NAME "Occlusion_TemporalAccumulation.cs", TemporalAccumulation<D, S, P, true, false>},     \
    {PREFIX NAME "Occlusion_HistoryFix.cs", HistoryFix<D, S, P, true, false>},                         \
    {PREFIX NAME "Occlusion_Blur.cs", Blur<D, S, P, true, false>},                                     \
    {PREFIX NAME "Occlusion_PostBlur_NoTemporalStabilization.cs", PostBlur<D, S, true, P, true, false>},
// REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION: the diffuse chain on RGBA16_SNORM (direction, hit distance) texels whose "luma" is .w
#define REBLUR_DIRECTIONAL_OCCLUSION_PASSES(PREFIX, P)                                                                                  \
    {PREFIX "DiffuseDirectionalOcclusion_PrePass.cs", PrePass<true, false, P, false, 2>},                                              \
    {PREFIX "DiffuseDirectionalOcclusion_TemporalAccumulation.cs", TemporalAccumulation<true, false, P, 2, false>},                    \
    {PREFIX "DiffuseDirectionalOcclusion_HistoryFix.cs", HistoryFix<true, false, P, 2, false>},                                        \
    {PREFIX "DiffuseDirectionalOcclusion_Blur.cs", Blur<true, false, P, 2, false>},                                                    \
    {PREFIX "DiffuseDirectionalOcclusion_PostBlur.cs", PostBlur<true, false, false, P, 2, false>},                                     \
    {PREFIX "DiffuseDirectionalOcclusion_PostBlur_NoTemporalStabilization.cs", PostBlur<true, false, true, P, 2, false>},              \
    {PREFIX "DiffuseDirectionalOcclusion_TemporalStabilization.cs", TemporalStabilization<true, false, P, false, 2>},
#define REBLUR_FAMILY(NAME, D, S)                                                                      \
    REBLUR_PASSES("REBLUR_", NAME, D, S, false)                                                        \
    REBLUR_PASSES("REBLUR_Perf_", NAME, D, S, true)                                                    \
    {"REBLUR_" NAME "_SplitScreen.cs", SplitScreen<D, S, false>},                                      \
    {"REBLUR_" NAME "Sh_SplitScreen.cs", SplitScreen<D, S, true>},

// ================================================================================================ Validation
// reference Shaders/Source/REBLUR_Validation.cs.hlsl:32-356 without the text overlay (MathLib's Text module and its font are not in the reference tree):
// a 4 x 4 grid of viewports over OUT_VALIDATION -- normals, roughness, viewZ, motion-vector error, world units / jitter / rotators, virtual-history amount,
// accumulated frames, hit distances; viewports nothing writes keep their previous content (the pass reads its own output).
struct ReblurValidationCB {
    ReblurCB shared;
    uint32_t gHasDiffuse, gHasSpecular;
};
static void ReblurValidation(const PassIO& io) {
    const ReblurValidationCB& vc = *(const ReblurValidationCB*)io.constants;
    const ReblurCB& c = vc.shared;
    const Tex &gIn_Normal_Roughness = io.t[0], &gIn_ViewZ = io.t[1], &gIn_Mv = io.t[2], &gIn_Data1 = io.t[3], &gIn_Data2 = io.t[4], &gIn_Diff = io.t[5], &gIn_Spec = io.t[6];
    Tex& gOut_Validation = io.t[7];
    const float VIEWPORT_SIZE = 0.25f;
    static const float3 special8[8] = {float3(-1.0f, 0.0f, 1.0f), float3(0.0f, 1.0f, 1.0f), float3(1.0f, 0.0f, 1.0f), float3(0.0f, -1.0f, 1.0f),
        float3(-0.25f * 1.41421356f, 0.25f * 1.41421356f, 0.5f), float3(0.25f * 1.41421356f, 0.25f * 1.41421356f, 0.5f), float3(0.25f * 1.41421356f, -0.25f * 1.41421356f, 0.5f),
        float3(-0.25f * 1.41421356f, -0.25f * 1.41421356f, 0.5f)};
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
            float4 diff = gIn_Diff.SampleNearest(viewportUvScaled * float2(c.gDiffCheckerboard != 2 ? 0.5f : 1.0f, 1.0f));
            float4 spec = gIn_Spec.SampleNearest(viewportUvScaled * float2(c.gSpecCheckerboard != 2 ? 0.5f : 1.0f, 1.0f));
            float4 d1 = gIn_Data1.SampleNearest(viewportUvScaled);
            float2 data1 = float2(d1.x, d1.y);
            if (!vc.gHasDiffuse || !vc.gHasSpecular) // single-signal denoisers store one channel (R8_UNORM)
                data1.y = data1.x;
            data1 = data1 * REBLUR_MAX_ACCUM_FRAME_NUM;
            uint32_t bits;
            float2 data2 = UnpackData2(gIn_Data2.LoadUint((int)(viewportUvScaled.x * c.gResourceSize.x), (int)(viewportUvScaled.y * c.gResourceSize.y)), bits);

            float3 N = normalAndRoughness.xyz();
            float3 Xv = Geometry::ReconstructViewPosition(viewportUv, c.gFrustum, abs(viewZ), c.gOrthoMode);
            float3 X = Geometry::RotateVector(c.gViewToWorld, Xv);
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
                float2 uvDelta = (viewportUvPrev - viewportUvPrevExpected) * c.gRectSize;
                result = float4(IsInScreenNearest(viewportUvPrev) != 0.0f ? float3(abs(uvDelta.x), abs(uvDelta.y), 0.0f) : float3(0, 0, 1), 1.0f);
            } else if (viewportIndex == 4.0f) {
                float2 dim = float2(Div(0.5f * c.gResourceSize.y, c.gResourceSize.x), 0.5f);
                float2 dimInPixels = c.gResourceSize * VIEWPORT_SIZE * dim;
                float2 remappedUv = Div(viewportUv - (1.0f - dim), dim);
                float2 remappedUv2 = Div(viewportUv - float2(1.0f - dim.x, 0.0f), dim);
                if (remappedUv.x > 0.0f && remappedUv.y > 0.0f) {
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
                } else if (remappedUv2.x > 0.0f && remappedUv2.y > 0.0f) {
                    float scale = 0.5f;
                    scale *= float(Sequence::ReverseBits4(c.gFrameIndex)) * 0.0625f;
                    int bx = (int)(remappedUv2.x * dimInPixels.x), by = (int)(remappedUv2.y * dimInPixels.y);
                    const float4 rot[3] = {c.gRotatorPre, c.gRotator, c.gRotatorPost};
                    for (int n = 0; n < 8; n++) {
                        float3 offset = special8[n] * scale;
                        for (int k = 0; k < 3; k++) {
                            float2 uv = 0.5f + Geometry::RotateVector(rot[k], float2(offset.x, offset.y));
                            float2 su = saturate(uv);
                            int ax = (int)(su.x * dimInPixels.x), ay = (int)(su.y * dimInPixels.y);
                            int dx = ax - bx < 0 ? bx - ax : ax - bx, dy = ay - by < 0 ? by - ay : ay - by;
                            result[k] += (dx <= 1 && dy <= 1) ? 1.0f : 0.0f;
                        }
                    }
                    result = c.gFrameIndex % 256 == 0 ? float4(0.0f) : float4(saturate(result.x), saturate(result.y), saturate(result.z), saturate(result.w));
                } else {
                    float roundingErrorCorrection = abs(viewZ) * 0.001f;
                    float3 v = X + roundingErrorCorrection;
                    result.x = frac(v.x) * notInf, result.y = frac(v.y) * notInf, result.z = frac(v.z) * notInf;
                }
                result.w = 1.0f;
            } else if (viewportIndex == 7.0f && vc.gHasSpecular) {
                result = float4(float3(data2.x * notInf), 1.0f);
            } else if ((viewportIndex == 8.0f && vc.gHasDiffuse) || (viewportIndex == 11.0f && vc.gHasSpecular)) {
                float frames = viewportIndex == 8.0f ? data1.x : data1.y;
                float f = 1.0f - saturate(Div(frames, max(c.gMaxAccumulatedFrameNum, 1.0f)));
                f = checkerboard && frames < 1.0f ? 0.75f : f;
                result = float4(Sequence::ColorizeZucconi(viewportUv.y > 0.95f ? 1.0f - viewportUv.x : f * notInf), 1.0f);
            } else if ((viewportIndex == 12.0f && vc.gHasDiffuse) || (viewportIndex == 15.0f && vc.gHasSpecular)) {
                float h = viewportIndex == 12.0f ? diff.w : spec.w;
                float3 v = h == 0.0f ? float3(1, 0, 0) : (h != saturate(h) ? float3(1, 0, 1) : float3(h));
                result = float4(v * notInf, 1.0f);
            }
            gOut_Validation.Store(px, py, result);
        }
}

const PassEntry* GetReblurPasses(uint32_t& n) {
    static const PassEntry k[] = {
        {"REBLUR_ClassifyTiles.cs", ClassifyTiles},
        {"REBLUR_Validation.cs", ReblurValidation},
        REBLUR_FAMILY("Diffuse", true, false)
        REBLUR_FAMILY("Specular", false, true)
        REBLUR_FAMILY("DiffuseSpecular", true, true)
        REBLUR_DIRECTIONAL_OCCLUSION_PASSES("REBLUR_", false)
        REBLUR_DIRECTIONAL_OCCLUSION_PASSES("REBLUR_Perf_", true)
    };
    n = sizeof(k) / sizeof(k[0]);
    return k;
}

} // namespace orc
