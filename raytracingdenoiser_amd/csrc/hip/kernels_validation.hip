// Validation overlays (CommonSettings::enableValidation) as HIP kernels for gfx950.
//   REBLUR_Validation   reference Shaders/Source/REBLUR_Validation.cs.hlsl:32-356
//   RELAX_Validation    reference Shaders/Source/RELAX_Validation.cs.hlsl:32-207
// A 4 x 4 grid of viewports over OUT_VALIDATION (RGBA8_UNORM), each showing one diagnostic: world normals, roughness, viewZ, the error of the motion
// vectors against the camera reprojection, world units / jitter / kernel rotators, virtual-history amount, accumulated frames, hit distances. The pass reads
// its own output: viewports nothing writes keep what they held. NOT restated: the text labels -- they come from MathLib's Text module, whose font data is not
// part of the reference tree; the diagnostics themselves are complete.
// One thread per output pixel, 32 x 8 workgroups; every plane is point-sampled, so there is nothing to stage: the pass is a few scattered 4..8-byte loads and
// one 4-byte read-modify-write per pixel (a debug feature: 0 cost unless enabled).
#include "../common/pass_constants.h"
#include "relax_device.h"

namespace nrdhip {

namespace {

constexpr float VIEWPORT_SIZE = 0.25f;
enum : uint8_t { FMT_R8_UNORM = 0, FMT_R8_UINT = 2, FMT_RG8_UNORM = 4, FMT_R32_UINT = 28 };

// Math::ReverseBits4 and Color::ColorizeZucconi of MathLib, restated (oracle/ml.h holds the same definitions; parity unpinned): the colour ramp is
// A. Zucconi's six-coefficient fit of the visible spectrum evaluated at x in [0, 1]
NRD_D uint32_t ReverseBits4(uint32_t x) { return ((x & 1u) << 3) | ((x & 2u) << 1) | ((x & 4u) >> 1) | ((x & 8u) >> 3); }
NRD_D float ZucconiBump(float x, float yoffset) { return Sat((1.0f - x * x) - yoffset); }
NRD_D float3 ColorizeZucconi(float x) {
    x = Sat(x);
    return F3(ZucconiBump(3.54585104f * (x - 0.69549072f), 0.02312639f) + ZucconiBump(3.90307140f * (x - 0.11748627f), 0.84897130f),
        ZucconiBump(2.93225262f * (x - 0.49228336f), 0.15225084f) + ZucconiBump(3.21182957f * (x - 0.86755042f), 0.88445281f),
        ZucconiBump(2.41593945f * (x - 0.27699880f), 0.52607955f) + ZucconiBump(3.96587128f * (x - 0.66077860f), 0.73949448f));
}

// SampleLevel( gNearestClamp, uv, 0 ) of a plane whose format is only known at launch (the pass serves every signal family)
NRD_D float4 SampleNearestAny(const Plane& p, uint32_t format, float2 uv) {
    const int2 t = NearestTexel(p, uv);
    switch (format) {
        case FORMAT_RGBA16_SFLOAT: return LoadRGBA16F(p, t.x, t.y);
        case FORMAT_RGBA16_SNORM: return LoadRGBA16Snorm(p, t.x, t.y);
        case FORMAT_R16_UNORM: return F4(LoadR16Unorm(p, t.x, t.y), 0.0f, 0.0f, 0.0f);
        case FMT_RG8_UNORM: {
            const float2 v = LoadRG8Unorm(p, t.x, t.y);
            return F4(v.x, v.y, 0.0f, 0.0f);
        }
        case FMT_R8_UNORM: return F4(LoadR8Unorm(p, t.x, t.y), 0.0f, 0.0f, 0.0f);
        default: return F4(0.0f);
    }
}

struct Viewport {
    float2 uv, uvScaled;
    float index;
};
NRD_D Viewport MakeViewport(int px, int py, float2 resourceSize, float2 resolutionScale) {
    Viewport v;
    const float2 pixelUv = Div(F2(float(px) + 0.5f, float(py) + 0.5f), resourceSize);
    const float2 scaled = pixelUv * 4.0f; // / VIEWPORT_SIZE
    const float2 id = Floor(scaled);
    v.uv = scaled - id;
    v.index = id.y * 4.0f + id.x; // / VIEWPORT_SIZE
    v.uvScaled = v.uv * resolutionScale;
    return v;
}

// the "jitter" square of viewport 4 (both denoisers)
NRD_D void JitterMark(float2 jitter, float2 remappedUv, float2 dimInPixels, float4& result) {
    const float2 uv = jitter + 0.5f;
    const float2 su = Sat(uv);
    const bool isValid = su.x == uv.x && su.y == uv.y;
    const int ax = (int)(su.x * dimInPixels.x), ay = (int)(su.y * dimInPixels.y);
    const int bx = (int)(remappedUv.x * dimInPixels.x), by = (int)(remappedUv.y * dimInPixels.y);
    const int dx = ax - bx < 0 ? bx - ax : ax - bx, dy = ay - by < 0 ? by - ay : ay - by;
    if (dx <= 1 && dy <= 1 && isValid)
        result.x = result.y = result.z = 0.66f;
    if (dx <= 3 && dy <= 3 && !isValid)
        result.x = 1.0f, result.y = 0.0f, result.z = 0.0f;
}

struct ReblurValidationPlanes {
    Plane normalRoughness, viewZ, mv, data1, data2, diff, spec, out;
    uint32_t data1Format, data2Format, diffFormat, specFormat;
};

__global__ __launch_bounds__(256) void ReblurValidationKernel(nrdc::ReblurValidationConstants vc, ReblurValidationPlanes P) {
    const ReblurCB& c = vc.shared;
    const int px = blockIdx.x * 32 + (threadIdx.x & 31), py = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (!InBounds(P.out, px, py))
        return;
    if (c.gResetHistory != 0) {
        StoreRGBA8Unorm(P.out, px, py, F4(0.0f));
        return;
    }
    const Viewport v = MakeViewport(px, py, ToF2(c.gResourceSize), ToF2(c.gResolutionScale));
    const float2 guideUv = v.uvScaled + ToF2(c.gRectOffset);

    const int2 tn = NearestTexel(P.normalRoughness, guideUv);
    const float4 normalAndRoughness = UnpackNormalAndRoughness(LoadInNormalRoughnessTexel(P.normalRoughness, tn.x, tn.y));
    const int2 tz = NearestTexel(P.viewZ, guideUv);
    const float viewZ = UnpackViewZ(c, LoadR32F(P.viewZ, tz.x, tz.y));
    const int2 tm = NearestTexel(P.mv, guideUv);
    const float4 mvRaw = LoadRGBA16F(P.mv, tm.x, tm.y);
    const float3 mv = F3(mvRaw.x * c.gMvScale.x, mvRaw.y * c.gMvScale.y, mvRaw.z * c.gMvScale.z);
    const float4 diff = SampleNearestAny(P.diff, P.diffFormat, v.uvScaled * F2(c.gDiffCheckerboard != 2 ? 0.5f : 1.0f, 1.0f));
    const float4 spec = SampleNearestAny(P.spec, P.specFormat, v.uvScaled * F2(c.gSpecCheckerboard != 2 ? 0.5f : 1.0f, 1.0f));
    const float4 d1 = SampleNearestAny(P.data1, P.data1Format, v.uvScaled);
    float2 data1 = F2(d1.x, d1.y);
    if (!vc.gHasDiffuse || !vc.gHasSpecular) // single-signal denoisers store one channel (R8_UNORM)
        data1.y = data1.x;
    data1 = data1 * REBLUR_MAX_ACCUM_FRAME_NUM;
    const int dx2 = (int)(v.uvScaled.x * c.gResourceSize.x), dy2 = (int)(v.uvScaled.y * c.gResourceSize.y);
    uint32_t packed2 = 0, bits;
    if (InBounds(P.data2, dx2, dy2))
        packed2 = P.data2Format == FMT_R32_UINT ? LoadR32U(P.data2, dx2, dy2) : LoadR8U(P.data2, dx2, dy2);
    const float2 data2 = UnpackData2(packed2, bits);

    const float3 N = Xyz(normalAndRoughness);
    const float3 Xv = ReconstructViewPosition(v.uv, ToF4(c.gFrustum), Abs(viewZ), c.gOrthoMode);
    const float3 X = RotateVector(c.gViewToWorld, Xv);
    const bool isInf = Abs(viewZ) > c.gDenoisingRange;
    const bool checkerboard = CheckerBoard((uint32_t)px >> 2, (uint32_t)py >> 2, 0) != 0;
    const float notInf = isInf ? 0.0f : 1.0f;

    float4 result = LoadRGBA8Unorm(P.out, px, py);
    if (v.index == 0.0f) {
        result = F4(N * 0.5f + F3(0.5f), 1.0f);
    } else if (v.index == 1.0f) {
        result = F4(F3(normalAndRoughness.w), 1.0f);
    } else if (v.index == 2.0f) {
        const float f = Div(0.1f * Abs(viewZ), 1.0f + 0.1f * Abs(viewZ));
        const float3 color = viewZ < 0.0f ? F3(0.0f, 0.0f, 1.0f) : F3(0.0f, 1.0f, 0.0f);
        result = F4(isInf ? F3(1.0f, 0.0f, 0.0f) : color * f, 1.0f);
    } else if (v.index == 3.0f) {
        const float2 expected = GetScreenUv(c.gWorldToClipPrev, X);
        float2 prev = v.uv + F2(mv.x, mv.y);
        if (c.gMvScale.w != 0.0f)
            prev = GetScreenUv(c.gWorldToClipPrev, X + mv);
        const float2 uvDelta = (prev - expected) * ToF2(c.gRectSize);
        result = F4(IsInScreenNearest(prev) != 0.0f ? F3(Abs(uvDelta.x), Abs(uvDelta.y), 0.0f) : F3(0.0f, 0.0f, 1.0f), 1.0f);
    } else if (v.index == 4.0f) {
        const float2 dim = F2(Div(0.5f * c.gResourceSize.y, c.gResourceSize.x), 0.5f);
        const float2 dimInPixels = ToF2(c.gResourceSize) * VIEWPORT_SIZE * dim;
        const float2 remappedUv = Div(v.uv - (F2(1.0f, 1.0f) - dim), dim);
        const float2 remappedUv2 = Div(v.uv - F2(1.0f - dim.x, 0.0f), dim);
        if (remappedUv.x > 0.0f && remappedUv.y > 0.0f) {
            JitterMark(ToF2(c.gJitter), remappedUv, dimInPixels, result);
        } else if (remappedUv2.x > 0.0f && remappedUv2.y > 0.0f) {
            float scale = 0.5f;
            scale *= float(ReverseBits4(c.gFrameIndex)) * 0.0625f;
            const int bx = (int)(remappedUv2.x * dimInPixels.x), by = (int)(remappedUv2.y * dimInPixels.y);
            const float4 rot[3] = {ToF4(c.gRotatorPre), ToF4(c.gRotator), ToF4(c.gRotatorPost)};
            float acc[3] = {result.x, result.y, result.z};
            for (int n = 0; n < 8; n++) {
                const float3 offset = F3(g_Special8[n][0], g_Special8[n][1], g_Special8[n][2]) * scale;
                for (int k = 0; k < 3; k++) {
                    const float2 uv = F2(0.5f, 0.5f) + RotateVector(rot[k], F2(offset.x, offset.y));
                    const float2 su = Sat(uv);
                    const int ax = (int)(su.x * dimInPixels.x), ay = (int)(su.y * dimInPixels.y);
                    const int dx = ax - bx < 0 ? bx - ax : ax - bx, dy = ay - by < 0 ? by - ay : ay - by;
                    acc[k] += (dx <= 1 && dy <= 1) ? 1.0f : 0.0f;
                }
            }
            result = c.gFrameIndex % 256 == 0 ? F4(0.0f) : F4(Sat(acc[0]), Sat(acc[1]), Sat(acc[2]), Sat(result.w));
        } else {
            const float roundingErrorCorrection = Abs(viewZ) * 0.001f;
            const float3 w = X + F3(roundingErrorCorrection);
            result.x = Frac(w.x) * notInf, result.y = Frac(w.y) * notInf, result.z = Frac(w.z) * notInf;
        }
        result.w = 1.0f;
    } else if (v.index == 7.0f && vc.gHasSpecular) {
        result = F4(F3(data2.x * notInf), 1.0f);
    } else if ((v.index == 8.0f && vc.gHasDiffuse) || (v.index == 11.0f && vc.gHasSpecular)) {
        const float frames = v.index == 8.0f ? data1.x : data1.y;
        float f = 1.0f - Sat(Div(frames, Max(c.gMaxAccumulatedFrameNum, 1.0f)));
        f = checkerboard && frames < 1.0f ? 0.75f : f;
        result = F4(ColorizeZucconi(v.uv.y > 0.95f ? 1.0f - v.uv.x : f * notInf), 1.0f);
    } else if ((v.index == 12.0f && vc.gHasDiffuse) || (v.index == 15.0f && vc.gHasSpecular)) {
        const float h = v.index == 12.0f ? diff.w : spec.w;
        const float3 col = h == 0.0f ? F3(1.0f, 0.0f, 0.0f) : (h != Sat(h) ? F3(1.0f, 0.0f, 1.0f) : F3(h));
        result = F4(col * notInf, 1.0f);
    }
    StoreRGBA8Unorm(P.out, px, py, result);
}

const char* LaunchReblurValidation(const PassArgs& a) {
    if (a.planesNum != 8 || !a.constants || a.constantsSize < sizeof(nrdc::ReblurValidationConstants))
        return "REBLUR_Validation: unexpected resources or constants";
    const nrdc::ReblurValidationConstants& vc = *(const nrdc::ReblurValidationConstants*)a.constants;
    if (vc.shared.gOrthoMode != 0.0f)
        return "REBLUR: orthographic projection is not supported (SURVEY.md section 8c)";
    if (a.formats[7] != FORMAT_RGBA8_UNORM)
        return "REBLUR_Validation: OUT_VALIDATION must be RGBA8_UNORM";
    ReblurValidationPlanes P = {a.planes[0], a.planes[1], a.planes[2], a.planes[3], a.planes[4], a.planes[5], a.planes[6], a.planes[7], a.formats[3], a.formats[4], a.formats[5], a.formats[6]};
    LaunchPass(a, ReblurValidationKernel, GridFor(P.out.w, P.out.h, 32, 8), dim3(256), vc, P);
    return nullptr;
}

struct RelaxValidationPlanes {
    Plane normalRoughness, viewZ, mv, historyLength, out;
};

__global__ __launch_bounds__(256) void RelaxValidationKernel(RelaxCB c, RelaxValidationPlanes P) {
    const int px = blockIdx.x * 32 + (threadIdx.x & 31), py = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (!InBounds(P.out, px, py))
        return;
    if (c.shared.gResetHistory != 0) {
        StoreRGBA8Unorm(P.out, px, py, F4(0.0f));
        return;
    }
    const Viewport v = MakeViewport(px, py, ToF2(c.shared.gResourceSize), ToF2(c.shared.gResolutionScale));
    const float2 guideUv = v.uvScaled + ToF2(c.shared.gRectOffset);

    const int2 tn = NearestTexel(P.normalRoughness, guideUv);
    const float4 normalAndRoughness = UnpackNormalAndRoughness(LoadInNormalRoughnessTexel(P.normalRoughness, tn.x, tn.y));
    const int2 tz = NearestTexel(P.viewZ, guideUv);
    const float viewZ = RelaxUnpackViewZ(c, LoadR32F(P.viewZ, tz.x, tz.y));
    const int2 tm = NearestTexel(P.mv, guideUv);
    const float4 mvRaw = LoadRGBA16F(P.mv, tm.x, tm.y);
    const float3 mv = F3(mvRaw.x * c.shared.gMvScale.x, mvRaw.y * c.shared.gMvScale.y, mvRaw.z * c.shared.gMvScale.z);
    const int2 th = NearestTexel(P.historyLength, v.uvScaled);
    const float historyLength = 255.0f * LoadR8Unorm(P.historyLength, th.x, th.y) - 1.0f;

    const float3 N = Xyz(normalAndRoughness);
    const float3 X = GetCurrentWorldPosFromClipSpaceXY(c, v.uv * 2.0f - 1.0f, Abs(viewZ));
    const bool isInf = Abs(viewZ) > c.shared.gDenoisingRange;
    const bool checkerboard = CheckerBoard((uint32_t)px >> 2, (uint32_t)py >> 2, 0) != 0;
    const float notInf = isInf ? 0.0f : 1.0f;

    float4 result = LoadRGBA8Unorm(P.out, px, py);
    if (v.index == 0.0f) {
        result = F4(N * 0.5f + F3(0.5f), 1.0f);
    } else if (v.index == 1.0f) {
        result = F4(F3(normalAndRoughness.w), 1.0f);
    } else if (v.index == 2.0f) {
        const float f = Div(0.1f * Abs(viewZ), 1.0f + 0.1f * Abs(viewZ));
        const float3 color = viewZ < 0.0f ? F3(0.0f, 0.0f, 1.0f) : F3(0.0f, 1.0f, 0.0f);
        result = F4(isInf ? F3(1.0f, 0.0f, 0.0f) : color * f, 1.0f);
    } else if (v.index == 3.0f) {
        const float2 expected = GetScreenUv(c.shared.gWorldToClipPrev, X);
        float2 prev = v.uv + F2(mv.x, mv.y);
        if (c.shared.gMvScale.w != 0.0f)
            prev = GetScreenUv(c.shared.gWorldToClipPrev, X + mv);
        const float2 uvDelta = (prev - expected) * F2(float(c.shared.gRectSize.x), float(c.shared.gRectSize.y));
        result = F4(IsInScreenNearest(prev) != 0.0f ? F3(Abs(uvDelta.x), Abs(uvDelta.y), 0.0f) : F3(0.0f, 0.0f, 1.0f), 1.0f);
    } else if (v.index == 4.0f) {
        const float2 dim = F2(Div(0.5f * c.shared.gResourceSize.y, c.shared.gResourceSize.x), 0.5f);
        const float2 remappedUv = Div(v.uv - (F2(1.0f, 1.0f) - dim), dim);
        if (remappedUv.x > 0.0f && remappedUv.y > 0.0f) {
            JitterMark(ToF2(c.shared.gJitter), remappedUv, ToF2(c.shared.gResourceSize) * VIEWPORT_SIZE * dim, result);
        } else {
            const float roundingErrorCorrection = Abs(viewZ) * 0.001f;
            const float3 w = X + F3(roundingErrorCorrection);
            result.x = Frac(w.x) * notInf, result.y = Frac(w.y) * notInf, result.z = Frac(w.z) * notInf;
        }
        result.w = 1.0f;
    } else if (v.index == 8.0f) {
        float f = 1.0f - Sat(Div(historyLength, Max(Max(c.shared.gDiffMaxAccumulatedFrameNum, c.shared.gSpecMaxAccumulatedFrameNum), 1.0f)));
        f = checkerboard && historyLength < 2.0f ? 0.75f : f;
        result = F4(ColorizeZucconi(v.uv.y > 0.95f ? 1.0f - v.uv.x : f * notInf), 1.0f);
    }
    StoreRGBA8Unorm(P.out, px, py, result);
}

const char* LaunchRelaxValidation(const PassArgs& a) {
    if (const char* e = CheckSupportedRelax(a))
        return e;
    if (a.planesNum != 5)
        return "RELAX_Validation: unexpected resource count";
    if (a.formats[4] != FORMAT_RGBA8_UNORM)
        return "RELAX_Validation: OUT_VALIDATION must be RGBA8_UNORM";
    RelaxValidationPlanes P = {a.planes[0], a.planes[1], a.planes[2], a.planes[3], a.planes[4]};
    RelaxCB c = LoadRelaxConstants(a);
    LaunchPass(a, RelaxValidationKernel, GridFor(P.out.w, P.out.h, 32, 8), dim3(256), c, P);
    return nullptr;
}

} // namespace

const PassEntry* GetValidationPasses(uint32_t& num) {
    static const PassEntry k[] = {
        {"REBLUR_Validation.cs", LaunchReblurValidation},
        {"RELAX_Validation.cs", LaunchRelaxValidation},
    };
    num = sizeof(k) / sizeof(k[0]);
    return k;
}

} // namespace nrdhip
