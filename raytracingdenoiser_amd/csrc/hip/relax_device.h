// RELAX device-side shared pieces: constants view, plane-binding cursor, RELAX_Common.hlsli helpers.
//   settings : reference Shaders/Include/RELAX_Config.hlsli:13-18
//   helpers  : reference Shaders/Include/RELAX_Common.hlsli:10-196
// The Common.hlsli helpers shared with REBLUR (weights, history filter, clamp-addressed fetches) come from reblur_device.h.
// Operation order is pinned (DESIGN.md "Numerics"): the CPU oracle restates the same arithmetic independently and the
// parity tests compare bit-for-bit.
#pragma once

#include "passes.h"
#include "reblur_device.h"

#include <cstring>

namespace nrdhip {

typedef nrdc::RelaxAtrousConstants RelaxCB; // shared block + (gStepSize, gIsLastPass); the tail is only valid for a-trous passes

#define RELAX_NORMAL_ULP (1.5f / 255.0f)
#define RELAX_MAX_ACCUM_FRAME_NUM 255.0f
#define RELAX_ANTILAG_ACCELERATION_AMOUNT_SCALE 10.0f
#define RELAX_HALF_PI 1.57079633f

constexpr int RELAX_TILE_X = 32;
constexpr int RELAX_TILE_Y = 8;

// every RELAX dispatch carries at least the 704-byte shared block; the launchers copy it into a RelaxCB kernel argument
inline RelaxCB LoadRelaxConstants(const PassArgs& a) {
    RelaxCB c;
    memset(&c, 0, sizeof(c));
    memcpy(&c, a.constants, a.constantsSize < sizeof(c) ? a.constantsSize : sizeof(c));
    return c;
}

inline const char* CheckSupportedRelax(const PassArgs& a) {
    if (!a.constants || a.constantsSize < sizeof(nrdc::RelaxConstants))
        return "RELAX: constant block missing";
    const nrdc::RelaxConstants& c = *(const nrdc::RelaxConstants*)a.constants;
    if (c.gRectOrigin.x != 0 || c.gRectOrigin.y != 0) // the executor moves the rect of the guide inputs to (0, 0) and zeroes this field (executor.hip "shifted rect")
        return "internal error: a pass was handed a non-zero rectOrigin";
    if (c.gOrthoMode != 0.0f)
        return "RELAX: orthographic projection is not supported (SURVEY.md section 8c)";
    return nullptr;
}

// walks DispatchDesc::resources in binding order (same walk as the host tables in csrc/host/denoiser_relax.cpp)
struct PlaneCursor {
    const Plane* p;
    uint32_t num, i;
    bool overflow;
    explicit PlaneCursor(const PassArgs& a) : p(a.planes), num(a.planesNum), i(0), overflow(false) {}
    Plane next() {
        if (i >= num) {
            overflow = true;
            return Plane{};
        }
        return p[i++];
    }
    bool complete() const { return !overflow && i == num; }
};

// planes of one radiance signal in a pass (unused members stay null)
struct SignalPlanes {
    Plane in, inSh, prev, prevSh, fast, fastSh, noisy, confidence;
    Plane out, outSh, outFast, outFastSh;
};

// ---- extra codecs ------------------------------------------------------------------------------------------------
NRD_D float4 FetchClampedRGBA8Unorm(const Plane& p, int x, int y) { return LoadRGBA8Unorm(p, ClampI(x, 0, p.w - 1), ClampI(y, 0, p.h - 1)); }
NRD_D float FetchClampedR8Unorm(const Plane& p, int x, int y) { return LoadR8Unorm(p, ClampI(x, 0, p.w - 1), ClampI(y, 0, p.h - 1)); }
NRD_D float4 SampleLinearRGBA8Unorm(const Plane& p, float2 pos) {
    LinearTaps t = MakeLinearTaps(pos);
    float4 s00 = FetchClampedRGBA8Unorm(p, t.x0, t.y0), s10 = FetchClampedRGBA8Unorm(p, t.x0 + 1, t.y0), s01 = FetchClampedRGBA8Unorm(p, t.x0, t.y0 + 1),
           s11 = FetchClampedRGBA8Unorm(p, t.x0 + 1, t.y0 + 1);
    return s00 * t.w00 + s10 * t.w10 + s01 * t.w01 + s11 * t.w11;
}
NRD_D float LoadR32FOrZero(const Plane& p, int x, int y) { return InBounds(p, x, y) ? LoadR32F(p, x, y) : 0.0f; }
NRD_D float LoadR8UnormOrZero(const Plane& p, int x, int y) { return InBounds(p, x, y) ? LoadR8Unorm(p, x, y) : 0.0f; }

NRD_D float4 operator*(float a, float4 b) { return F4(a * b.x, a * b.y, a * b.z, a * b.w); }
NRD_D float3 operator*(float a, float3 b) { return F3(a * b.x, a * b.y, a * b.z); }
NRD_D float3 Min3(float3 a, float3 b) { return F3(Min(a.x, b.x), Min(a.y, b.y), Min(a.z, b.z)); }
NRD_D float3 Max3(float3 a, float3 b) { return F3(Max(a.x, b.x), Max(a.y, b.y), Max(a.z, b.z)); }
NRD_D float3 Sqrt3(float3 a) { return F3(Sqrt(a.x), Sqrt(a.y), Sqrt(a.z)); }
NRD_D float4 Max0(float4 a) { return F4(Max(a.x, 0.0f), Max(a.y, 0.0f), Max(a.z, 0.0f), Max(a.w, 0.0f)); }
NRD_D float4 Clamp4(float4 a, float lo, float hi) { return F4(Clamp(a.x, lo, hi), Clamp(a.y, lo, hi), Clamp(a.z, lo, hi), Clamp(a.w, lo, hi)); }
NRD_D float Cmp(bool b) { return b ? 1.0f : 0.0f; }

// [ml] Color::RgbToYCoCg / YCoCgToRgb (unclamped inverse)
NRD_D float3 RgbToYCoCg(float3 c) { return LinearToYCoCg(c); }
NRD_D float3 YCoCgToRgb(float3 c) {
    float t = c.x - c.z;
    return F3(t + c.y, c.x + c.z, t - c.y);
}

// ---- RELAX_Common.hlsli ------------------------------------------------------------------------------------------
NRD_D float RelaxUnpackViewZ(const RelaxCB& c, float z) { return Abs(z * c.shared.gViewZScale); }
NRD_D float4 UnpackPrevNormalRoughness(float4 p) { return F4(SafeNormalize(F3(p.x * 2.0f - 1.0f, p.y * 2.0f - 1.0f, p.z * 2.0f - 1.0f)), p.w); }
NRD_D float4 PackPrevNormalRoughness(float4 nr) { return F4(nr.x * 0.5f + 0.5f, nr.y * 0.5f + 0.5f, nr.z * 0.5f + 0.5f, nr.w); }
NRD_D float BilinearWithCustomWeightsImmediateFloat(float s00, float s10, float s01, float s11, float4 w) {
    float o = s00 * w.x;
    o += s10 * w.y;
    o += s01 * w.z;
    o += s11 * w.w;
    float sumWeights = Sum(w);
    return sumWeights < 0.0001f ? 0.0f : o * Rcp(sumWeights);
}
NRD_D float4 BilinearWithCustomWeightsRGBA16F(const Plane& tex, int ox, int oy, float4 w) {
    float4 s00, s10, s01, s11;
    if (FootprintIsInterior(tex, ox, oy, 2, 2)) { // two 16-byte row loads instead of four 8-byte ones (same texels)
        LoadRGBA16Fx2(tex, ox, oy, s00, s10);
        LoadRGBA16Fx2(tex, ox, oy + 1, s01, s11);
    } else {
        s00 = LoadRGBA16FOrZero(tex, ox, oy), s10 = LoadRGBA16FOrZero(tex, ox + 1, oy), s01 = LoadRGBA16FOrZero(tex, ox, oy + 1), s11 = LoadRGBA16FOrZero(tex, ox + 1, oy + 1);
    }
    float4 o = s00 * w.x;
    o = Mad(s10, w.y, o);
    o = Mad(s01, w.z, o);
    o = Mad(s11, w.w, o);
    float sumWeights = Sum(w);
    return sumWeights < 0.0001f ? F4(0.0f) : o * Rcp(sumWeights);
}
// perspective only (CheckSupportedRelax): X = viewZ * (forward + right * clip.x - up * clip.y)
NRD_D float3 WorldPosFromClip(float3 R, float3 U, float3 Fw, float2 clip, float viewZ) { return viewZ * (Fw + R * clip.x - U * clip.y); }
NRD_D float3 GetCurrentWorldPosFromClipSpaceXY(const RelaxCB& c, float2 clip, float viewZ) {
    return WorldPosFromClip(ToF3(c.shared.gFrustumRight), ToF3(c.shared.gFrustumUp), ToF3(c.shared.gFrustumForward), clip, viewZ);
}
NRD_D float3 GetCurrentWorldPosFromPixelPos(const RelaxCB& c, int px, int py, float viewZ) {
    float2 clip = F2(float(px) + 0.5f, float(py) + 0.5f) * ToF2(c.shared.gRectSizeInv) * 2.0f - 1.0f;
    return GetCurrentWorldPosFromClipSpaceXY(c, clip, viewZ);
}
NRD_D float3 GetPreviousWorldPosFromClipSpaceXY(const RelaxCB& c, float2 clip, float viewZ) {
    return WorldPosFromClip(ToF3(c.shared.gPrevFrustumRight), ToF3(c.shared.gPrevFrustumUp), ToF3(c.shared.gPrevFrustumForward), clip, viewZ);
}
NRD_D float3 GetPreviousWorldPosFromPixelPos(const RelaxCB& c, int px, int py, float viewZ) {
    float2 rcpSize = F2(Rcp(c.shared.gRectSizePrev.x), Rcp(c.shared.gRectSizePrev.y));
    float2 clip = F2(float(px) + 0.5f, float(py) + 0.5f) * rcpSize * 2.0f - 1.0f;
    return GetPreviousWorldPosFromClipSpaceXY(c, clip, viewZ);
}
NRD_D float GetPlaneDistanceWeight(float3 centerWorldPos, float3 centerNormal, float centerViewZ, float3 sampleWorldPos, float threshold) {
    float d = Abs(Dot(sampleWorldPos - centerWorldPos, centerNormal));
    return Div(d, centerViewZ) > threshold ? 0.0f : 1.0f;
}
NRD_D float GetPlaneDistanceWeight_Atrous(float3 centerWorldPos, float3 centerNormal, float3 sampleWorldPos, float threshold) {
    float d = Abs(Dot(sampleWorldPos - centerWorldPos, centerNormal));
    return d < threshold ? 1.0f : 0.0f;
}
NRD_D float GetSpecLobeTanHalfAngleOld(float roughness, float percentOfVolume = 0.75f) { // RELAX keeps the pre-fix lobe formula
    roughness = Sat(roughness);
    percentOfVolume = Sat(percentOfVolume);
    return Div(roughness * roughness * percentOfVolume, 1.0f - percentOfVolume + NRD_EPS);
}
NRD_D float2 GetNormalWeightParams_ATrous(float roughness, float numFramesInHistory, float specularReprojectionConfidence, float normalEdgeStoppingRelaxation,
    float specularLobeAngleFraction, float specularLobeAngleSlack) {
    float relaxation = Sat(numFramesInHistory * (1.0f / 5.0f));
    relaxation *= Lerp(1.0f, specularReprojectionConfidence, normalEdgeStoppingRelaxation);
    float f = 0.9f + 0.1f * relaxation;
    float angle = Atan(GetSpecLobeTanHalfAngleOld(roughness, specularLobeAngleFraction));
    angle *= 10.0f - 9.0f * relaxation;
    angle += specularLobeAngleSlack;
    angle = Min(RELAX_HALF_PI, angle);
    return F2(angle, f);
}
NRD_D float GetSpecularNormalWeight_ATrous(float2 params0, float3 n0, float3 n, float3 v0, float3 v) {
    float cosaN = Dot(n0, n);
    float cosaV = Dot(v0, v);
    float cosa = Min(cosaN, cosaV);
    float a = AcosApprox(cosa);
    a = SmoothStep(0.0f, params0.x, a);
    return Sat(1.0f - a * params0.y);
}
NRD_D float GetNormalWeightParam2(float roughness, float angleFraction) {
    float angle = Atan(GetSpecLobeTanHalfAngleOld(roughness, angleFraction));
    return Rcp(Max(angle, RELAX_NORMAL_ULP));
}
NRD_D float GetEncodingAwareNormalWeightR(float3 Ncurr, float3 Nprev, float maxAngle, float curvatureAngle, float thresholdAngle, bool remap) {
    float w = GetEncodingAwareNormalWeight(Ncurr, Nprev, maxAngle, curvatureAngle, thresholdAngle);
    if (remap)
        w = SmoothStep(0.05f, 0.95f, w);
    return w;
}
NRD_D float2 ScreenUvNoKill(const float* worldToClip, float3 X) {
    float4 clip = ProjectiveTransform(worldToClip, X);
    return F2((Div(clip.x, clip.w)) * 0.5f + 0.5f, (Div(clip.y, clip.w)) * -0.5f + 0.5f);
}
NRD_D float2 RelaxClampUvToViewport(const RelaxCB& c, float2 uv) {
    float2 a = uv * ToF2(c.shared.gResolutionScale);
    float2 b = ToF2(c.shared.gResolutionScale) - ToF2(c.shared.gResourceSizeInv) * 0.5f;
    return F2(Min(a.x, b.x), Min(a.y, b.y));
}
NRD_D float ApplyThinLensEquation(float O, float curvature) { return Div(O, 2.0f * curvature * O + 1.0f); } // reference Common.hlsli:404-409
NRD_D float4 Denanify(float w, float4 x) { return w == 0.0f ? F4(0.0f) : x; }
// Denanify( w, tex[ p ] ) of the tap loops, with the selection made on the two raw dwords (2 selects instead of 4; the fp16 -> fp32 conversions then feed the
// accumulating multiply-adds directly and fold into v_fma_mix_f32: same values bit for bit -- reblur_device.h ReblurSignal::LoadOrZero has the measurement)
NRD_D float4 LoadDenanifiedRGBA16F(float w, const Plane& p, int x, int y) {
    uint2 raw = *TexelPtr<const uint2>(p, x, y);
    raw.x = w == 0.0f ? 0u : raw.x;
    raw.y = w == 0.0f ? 0u : raw.y;
    return F4(HalfBitsToFloat((uint16_t)(raw.x & 0xFFFFu)), HalfBitsToFloat((uint16_t)(raw.x >> 16)), HalfBitsToFloat((uint16_t)(raw.y & 0xFFFFu)), HalfBitsToFloat((uint16_t)(raw.y >> 16)));
}

// true when any of the 16x16 tiles overlapped by this workgroup's 32x8 block has geometry
NRD_D bool RelaxBlockHasGeometry(const Plane& tiles, int blockX, int blockY) {
    const int tileY = (blockY * RELAX_TILE_Y) >> 4, tileX0 = (blockX * RELAX_TILE_X) >> 4;
    bool any = false;
    for (int t = 0; t < RELAX_TILE_X / 16; t++)
        if (tileX0 + t < tiles.w && tileY < tiles.h)
            any |= LoadR8Unorm(tiles, tileX0 + t, tileY) == 0.0f;
    return any;
}

} // namespace nrdhip
