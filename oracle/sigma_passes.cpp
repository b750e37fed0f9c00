// ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/hlsl.h).
//
// CPU restatement of the SIGMA_SHADOW and SIGMA_SHADOW_TRANSLUCENCY pass chains. One template per pass, instantiated for
// SIGMA_TYPE = float (shadow) and float4 (shadow + translucent colour, "#ifdef SIGMA_TRANSLUCENT" in the reference).
//   ClassifyTiles           reference Shaders/Include/SIGMA_ClassifyTiles.hlsli:11-81
//   SmoothTiles             reference Shaders/Include/SIGMA_SmoothTiles.hlsli:11-48
//   Copy                    reference Shaders/Include/SIGMA_Copy.hlsli:11-24
//   Blur / PostBlur         reference Shaders/Include/SIGMA_Blur.hlsli:11-268 (SIGMA_FIRST_PASS / not)
//   TemporalStabilization   reference Shaders/Include/SIGMA_TemporalStabilization.hlsli:11-226
//   SplitScreen             reference Shaders/Include/SIGMA_SplitScreen.hlsli:11-35
//   helpers                 reference Shaders/Include/SIGMA_Common.hlsli:11-125, SIGMA_Config.hlsli:13-36
// Binding order of planes = reference Source/Denoisers/Sigma_Shadow.hpp:50-155 / Sigma_ShadowTranslucency.hpp:50-158.
#include "passes.h"
#include "reblur_common.h" // MakeHistoryFilter (shared Common.hlsli:602-656 machinery), CompareMaterials etc.

namespace orc {

namespace {

struct SigmaCB { // reference Shaders/Include/SIGMA_Config.hlsli:45-80
    float4x4 gWorldToView, gViewToClip, gWorldToClipPrev, gWorldToViewPrev;
    float4 gRotator, gRotatorPost, gViewVectorWorld, gLightDirectionView, gFrustum, gFrustumPrev, gCameraDelta, gMvScale;
    float2 gResourceSizeInv, gResourceSizeInvPrev, gRectSize, gRectSizeInv, gRectSizePrev, gResolutionScale, gRectOffset;
    uint32_t gPrintfAt[2], gRectOrigin[2];
    int gRectSizeMinusOne[2], gTilesSizeMinusOne[2];
    float gOrthoMode, gUnproject, gDenoisingRange, gPlaneDistSensitivity, gStabilizationStrength, gDebug, gSplitScreen, gViewZScale, gMinRectDimMulUnproject;
    uint32_t gFrameIndex, gIsRectChanged;
};
static_assert(sizeof(SigmaCB) == 516, "SIGMA constant block");

constexpr float SIGMA_MAX_PIXEL_RADIUS = 32.0f;
constexpr float SIGMA_TS_SIGMA_SCALE = 3.0f;
constexpr float SIGMA_MAX_ACCUM_FRAME_NUM = 7.0f;
constexpr int BORDER = 2;

inline float UnpackViewZ(const SigmaCB& c, float z) { return fabsf(z * c.gViewZScale); }
inline bool IsLit(float p) { return p >= NRD_FP16_MAX; }
inline float PackShadow(float s) { return Math::Sqrt01(s); }
inline float UnpackShadow(float s) { return s * s; } // SIGMA_BackEnd_UnpackShadow, NRD.hlsli:931
inline float4 PackShadow(float4 s) { return float4(Math::Sqrt01(s.x), Math::Sqrt01(s.y), Math::Sqrt01(s.z), Math::Sqrt01(s.w)); }
inline float4 UnpackShadow(float4 s) { return s * s; }

// SIGMA_TYPE (SIGMA_Config.hlsli:39-43) and the handful of component-wise intrinsics the passes apply to it
template <bool TRANSLUCENT> struct SigmaType;
template <> struct SigmaType<false> {
    typedef float type;
    static float From(float4 v) { return v.x; }
    static float X(float v) { return v; }
};
template <> struct SigmaType<true> {
    typedef float4 type;
    static float4 From(float4 v) { return v; }
    static float X(float4 v) { return v.x; }
};
inline float StdDev(float m1, float m2) { return HwSqrt(fabsf(m2 - m1 * m1)); } // GetStdDev, Common.hlsli:227
inline float4 StdDev(float4 m1, float4 m2) { return float4(StdDev(m1.x, m2.x), StdDev(m1.y, m2.y), StdDev(m1.z, m2.z), StdDev(m1.w, m2.w)); }
inline float Clamp(float x, float a, float b) { return clamp(x, a, b); }
inline float4 Clamp(float4 x, float4 a, float4 b) { return float4(clamp(x.x, a.x, b.x), clamp(x.y, a.y, b.y), clamp(x.z, a.z, b.z), clamp(x.w, a.w, b.w)); }
inline float Saturate(float x) { return saturate(x); }
inline float4 Saturate(float4 x) { return float4(saturate(x.x), saturate(x.y), saturate(x.z), saturate(x.w)); }
inline float3 GetViewVectorV(const SigmaCB& c, float3 X) { return c.gOrthoMode == 0.0f ? normalize(-X) : float3(0, 0, -1); }

// SIGMA_Common.hlsli:21-33 (5x5 radius-estimation kernel => minimum radius 2)
inline float GetKernelRadiusInPixels(float hitDist, float unprojectZ, float scale = 1.0f) {
    float unclampedRadius = Div(hitDist, unprojectZ);
    unclampedRadius *= scale;
    float minRadius = min(unclampedRadius, 2.0f);
    return clamp(unclampedRadius, minRadius, SIGMA_MAX_PIXEL_RADIUS);
}
inline float AreBothLitOrUnlit(float penumbra1, float penumbra2) { return ((penumbra1 == 0.0f) == (penumbra2 == 0.0f)) ? 1.0f : 0.0f; }

// SIGMA_Common.hlsli:45-92 restated in scalar form; returns channel .y of the bicubically filtered RG8 tile map
inline void BicubicAxis(float f, float& w0, float& w1, float& wz) {
    const float k = 1.0f / 6.0f;
    float f2 = f * f, f3 = f2 * f;
    float phix = k * (-1.0f * f3 + 3.0f * f2 + -3.0f * f + 1.0f);
    float phiy = k * (3.0f * f3 + -6.0f * f2 + 0.0f * f + 4.0f);
    float phiz = k * (-3.0f * f3 + 3.0f * f2 + 3.0f * f + 1.0f);
    float phiw = k * (1.0f * f3 + 0.0f * f2 + 0.0f * f + 0.0f);
    w0 = 1.0f + 1.0f * f + -Div(1.0f * phiy, phix + phiy);
    w1 = 1.0f + -1.0f * f + Div(1.0f * phiw, phiz + phiw);
    wz = phix + phiy;
}
inline float TextureCubicY(const Tex& tex, float2 uv) {
    float2 size = float2(float(tex.W()), float(tex.H()));
    float dx = -Rcp(size.x), dy = -Rcp(size.y);
    float2 t = uv * size - 0.5f;
    float2 f = float2(frac(t.x), frac(t.y));
    float xw0, xw1, xwz, yw0, yw1, ywz;
    BicubicAxis(f.x, xw0, xw1, xwz);
    BicubicAxis(f.y, yw0, yw1, ywz);
    // uv_10_00 = uv.xyxy + (1,1,-1,-1) * xw.xxyy * (dx,-0,dx,-0)  ->  x coordinates only move
    float u10 = uv.x + 1.0f * xw0 * dx, u00 = uv.x + -1.0f * xw1 * dx;
    // uv_11_01 = uv_10_00 + yw.x * (-0,dy,-0,dy) ; uv_10_00 -= yw.y * (-0,dy,-0,dy)
    float v1 = uv.y + yw0 * dy, v0 = uv.y - yw1 * dy;
    float c00 = tex.SampleLinearTexel(float2(u00, v0) * size).y;
    float c10 = tex.SampleLinearTexel(float2(u10, v0) * size).y;
    float c01 = tex.SampleLinearTexel(float2(u00, v1) * size).y;
    float c11 = tex.SampleLinearTexel(float2(u10, v1) * size).y;
    float tx = ywz, ty = xwz; // return float2( yw.z, xw.z )
    c00 = lerp(c00, c01, tx);
    c10 = lerp(c10, c11, tx);
    return lerp(c00, c10, ty);
}

// ================================================================================================ ClassifyTiles
template <bool TRANSLUCENT>
void ClassifyTiles(const PassIO& io) {
    const SigmaCB& c = *(const SigmaCB*)io.constants;
    uint32_t k = 0;
    const Tex& gIn_ViewZ = io.t[k++];
    const Tex& gIn_Penumbra = io.t[k++];
    const Tex* gIn_Shadow_Translucency = TRANSLUCENT ? &io.t[k++] : nullptr;
    Tex& gOut_Tiles = io.t[k++];
    // one thread group per 16x16 tile of the RECT (the dispatch grid of Sigma.cpp: ceil( rectSize / 16 )): tiles of the plane beyond it are left alone, as in the reference
    // (round 4: this loop used to walk the whole resource-sized tile plane; found by the per-pass comparison under dynamic resolution)
    const int tilesW = min(gOut_Tiles.W(), ((int)c.gRectSize.x + 15) / 16), tilesH = min(gOut_Tiles.H(), ((int)c.gRectSize.y + 15) / 16);
#pragma omp parallel for schedule(static)
    for (int ty = 0; ty < tilesH; ty++)
        for (int tx = 0; tx < tilesW; tx++) {
            uint32_t lit = 0, umbra = 0, inf = 0;
            float maxRadius = 0.0f;
            for (int j = 0; j < 16; j++)
                for (int i = 0; i < 16; i++) {
                    int x = tx * 16 + i, y = ty * 16 + j;
                    float h = gIn_Penumbra.Load(x, y).x;
                    float viewZ = UnpackViewZ(c, gIn_ViewZ.Load(x, y).x);
                    bool isInf = viewZ > c.gDenoisingRange, isShadow = h == 0.0f, isLitP = IsLit(h);
                    bool isOpaque = true;
                    if (TRANSLUCENT) {
                        float4 t = gIn_Shadow_Translucency->Load(x, y);
                        isOpaque = Color::Luminance(float3(t.y, t.z, t.w)) < 0.003f;
                    }
                    lit += (isLitP || isInf || isShadow) ? 1 : 0;
                    umbra += ((!isLitP && isOpaque) || isInf || isShadow) ? 1 : 0;
                    inf += isInf ? 1 : 0;
                    float hitDist = (isLitP || isInf) ? 0.0f : h;
                    float pixelSize = PixelRadiusToWorld(c.gUnproject, c.gOrthoMode, 1.0f, viewZ);
                    maxRadius = max(GetKernelRadiusInPixels(hitDist, pixelSize), maxRadius);
                }
            bool isLitT = lit == 256, isUmbra = umbra == 256, isInfT = inf == 256;
            float4 result = float4((isLitT || isUmbra) ? 0.0f : 1.0f, saturate(maxRadius * 0.0625f), isInfT ? 1.0f : 0.0f, 0.0f);
            gOut_Tiles.Store(tx, ty, result);
        }
}

// ================================================================================================ SmoothTiles
void SmoothTiles(const PassIO& io) {
    const SigmaCB& c = *(const SigmaCB*)io.constants;
    const Tex& gIn_Tiles = io.t[0];
    Tex& gOut_Tiles = io.t[1];
    for (int y = 0; y < gOut_Tiles.H(); y++)
        for (int x = 0; x < gOut_Tiles.W(); x++) {
            float4 center = gIn_Tiles.Load(x, y);
            float blurry = 0.0f, sumw = 0.0f;
            float k = Div(1.01f, center.y + 0.01f);
            for (int j = 0; j <= 2; j++)
                for (int i = 0; i <= 2; i++) {
                    float d = length(float2(float(i), float(j)) - 1.0f);
                    float w = exp2(-k * d * d);
                    int sx = clamp(x - 1 + i, 0, c.gTilesSizeMinusOne[0]), sy = clamp(y - 1 + j, 0, c.gTilesSizeMinusOne[1]);
                    blurry += gIn_Tiles.Load(sx, sy).x * w;
                    sumw += w;
                }
            blurry = Div(blurry, sumw);
            gOut_Tiles.Store(x, y, float4(center.z, blurry, 0.0f, 0.0f));
        }
}

// ================================================================================================ Copy
void Copy(const PassIO& io) {
    const SigmaCB& c = *(const SigmaCB*)io.constants;
    const Tex& gIn_Tiles = io.t[0];
    const Tex& gIn_History = io.t[1];
    const Tex& gIn_HistoryLength = io.t[2];
    Tex& gOut_History = io.t[3];
    Tex& gOut_HistoryLength = io.t[4];
#pragma omp parallel for schedule(static)
    for (int y = 0; y < gOut_History.H(); y++)
        for (int x = 0; x < gOut_History.W(); x++) {
            float isSky = gIn_Tiles.Load(x >> 4, y >> 4).x;
            if (isSky != 0.0f && !c.gIsRectChanged)
                continue;
            gOut_History.Store(x, y, gIn_History.Load(x, y)); // R8 -> R8 / RGBA8 -> RGBA8: the stored bytes round-trip exactly
            gOut_HistoryLength.StoreUint(x, y, gIn_HistoryLength.LoadUint(x, y));
        }
}

// ================================================================================================ Blur / PostBlur
template <bool FIRST_PASS, bool TRANSLUCENT>
void Blur(const PassIO& io) {
    typedef SigmaType<TRANSLUCENT> ST;
    typedef typename ST::type S;
    constexpr bool READS_SHADOW = !FIRST_PASS || TRANSLUCENT; // SIGMA_Blur.hlsli:25
    const SigmaCB& c = *(const SigmaCB*)io.constants;
    uint32_t k = 0;
    const Tex& gIn_ViewZ = io.t[k++];
    const Tex& gIn_Normal_Roughness = io.t[k++];
    const Tex& gIn_Penumbra = io.t[k++];
    const Tex& gIn_Tiles = io.t[k++];
    const Tex* gIn_Shadow = READS_SHADOW ? &io.t[k++] : nullptr; // TEMP_1, or IN_TRANSLUCENCY in the translucent first pass
    Tex& gOut_Penumbra = io.t[k++];
    Tex& gOut_Shadow = io.t[k++];
    const int rw = c.gRectSizeMinusOne[0], rh = c.gRectSizeMinusOne[1];

#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py <= rh; py++)
        for (int px = 0; px <= rw; px++) {
            float isSky = gIn_Tiles.Load(px >> 4, py >> 4).x;
            if (isSky != 0.0f)
                continue;

            // "shared memory" with clamped coordinates
            auto sPenumbra = [&](int x, int y) { return gIn_Penumbra.Load(clamp(x, 0, rw), clamp(y, 0, rh)).x; };
            auto sViewZ = [&](int x, int y) { return UnpackViewZ(c, gIn_ViewZ.Load(clamp(x, 0, rw), clamp(y, 0, rh)).x); };
            auto sShadow = [&](int x, int y) -> S {
                x = clamp(x, 0, rw), y = clamp(y, 0, rh);
                S s;
                if (READS_SHADOW)
                    s = ST::From(gIn_Shadow->Load(x, y));
                else
                    s = S(IsLit(gIn_Penumbra.Load(x, y).x) ? 1.0f : 0.0f);
                return FIRST_PASS ? s : UnpackShadow(s);
            };

            float centerPenumbra = sPenumbra(px, py);
            float viewZ = sViewZ(px, py);
            if (viewZ > c.gDenoisingRange)
                continue;

            float2 pixelUv = float2(float(px) + 0.5f, float(py) + 0.5f) * c.gRectSizeInv;
            float tileValue = TextureCubicY(gIn_Tiles, pixelUv * c.gResolutionScale);

            if (tileValue == 0.0f || centerPenumbra == 0.0f) {
                gOut_Penumbra.Store(px, py, centerPenumbra);
                gOut_Shadow.Store(px, py, PackShadow(sShadow(px, py)));
                continue;
            }

            float3 Xv = Geometry::ReconstructViewPosition(pixelUv, c.gFrustum, viewZ, c.gOrthoMode);
            float3 N = NRD_FrontEnd_UnpackNormalAndRoughness(gIn_Normal_Roughness.Load(px, py)).xyz();
            float3 Nv = Geometry::RotateVector(c.gWorldToView, N);

            float pixelSize = PixelRadiusToWorld(c.gUnproject, c.gOrthoMode, 1.0f, viewZ);
            float frustumSize = GetFrustumSize(c.gMinRectDimMulUnproject, c.gOrthoMode, viewZ);
            float3 Vv = GetViewVectorV(c, Xv);
            float NoV = fabsf(dot(Nv, Vv));
            float2 geometryWeightParams = GetGeometryWeightParams(c.gPlaneDistSensitivity, frustumSize, Xv, Nv);

            // Dense 5x5: penumbra size estimate + shadow filter
            float sumx = 0.0f, sumy = 0.0f, penumbra = 0.0f;
            S result = S(0.0f), centerTap = S(0.0f);
            for (int j = 0; j <= BORDER * 2; j++)
                for (int i = 0; i <= BORDER * 2; i++) {
                    int x = px - BORDER + i, y = py - BORDER + j;
                    float penum = sPenumbra(x, y), zs = sViewZ(x, y);
                    S s = sShadow(x, y);

                    float w = 1.0f;
                    if (i == BORDER && j == BORDER)
                        centerTap = s;
                    else {
                        float2 uv = pixelUv + float2(float(i - BORDER), float(j - BORDER)) * c.gRectSizeInv;
                        float3 Xvs = Geometry::ReconstructViewPosition(uv, c.gFrustum, zs, c.gOrthoMode);
                        w *= ComputeWeight(dot(Nv, Xvs), geometryWeightParams.x, geometryWeightParams.y);
                        w *= AreBothLitOrUnlit(centerPenumbra, penum);
                        w *= GetGaussianWeight(length(Div(float2(float(i - BORDER), float(j - BORDER)), float(BORDER))));
                    }

                    result = result + (w == 0.0f ? S(0.0f) : s * w);
                    sumx += w;

                    w *= Div(pixelSize, pixelSize + penum);
                    w *= IsLit(penum) ? 0.0f : 1.0f;

                    penumbra += w == 0.0f ? 0.0f : penum * w;
                    sumy += w;
                }

            result = Div(result, sumx);
            sumx = 1.0f;
            penumbra = Div(penumbra, max(sumy, NRD_EPS));
            sumy = sumy != 0.0f ? 1.0f : 0.0f;

            // Avoid a blurry result if the penumbra is smaller than the dense kernel
            float penumbraInPixels = Div(penumbra, pixelSize);
            float f = Math::SmoothStep(0.0f, float(BORDER), penumbraInPixels);
            result = lerp(centerTap, result, f);

            // Sparse 8-tap blur
            f = lerp(4.0f, 1.0f, f);
            result = result * f;
            penumbra *= f;
            sumx *= f;
            sumy *= f;

            float blurRadius = GetKernelRadiusInPixels(penumbra, pixelSize, tileValue);
            float4 rotator = FIRST_PASS ? c.gRotator : c.gRotatorPost;

            float2 skew = lerp(float2(1.0f - fabsf(Nv.x), 1.0f - fabsf(Nv.y)), float2(1.0f), NoV);
            skew = Div(skew, max(skew.x, skew.y));
            skew *= c.gRectSizeInv * blurRadius;
            float4 scaledRotator = Geometry::ScaleRotator(rotator, skew);

            float invEstimatedPenumbra = Rcp(max(penumbra, NRD_EPS));

            for (int n = 0; n < 8; n++) {
                float3 offset = g_Special8[n];
                float2 uv = pixelUv + Geometry::RotateVector(scaledRotator, float2(offset.x, offset.y));
                uv = (floor(uv * c.gRectSize) + 0.5f) * c.gRectSizeInv;
                float2 uvScaled = min(uv * c.gResolutionScale, c.gResolutionScale - 0.5f * c.gResourceSizeInv);

                float penum = gIn_Penumbra.SampleNearest(uvScaled).x;
                float zs = UnpackViewZ(c, gIn_ViewZ.SampleNearest(uvScaled).x);
                S s;
                if (READS_SHADOW)
                    s = ST::From(gIn_Shadow->SampleNearest(uvScaled));
                else
                    s = S(IsLit(penum) ? 1.0f : 0.0f);
                if (!FIRST_PASS)
                    s = UnpackShadow(s);

                float3 Xvs = Geometry::ReconstructViewPosition(uv, c.gFrustum, zs, c.gOrthoMode);

                float w = IsInScreenNearest(uv);
                w *= ComputeWeight(dot(Nv, Xvs), geometryWeightParams.x, geometryWeightParams.y);
                w *= AreBothLitOrUnlit(centerPenumbra, penum);
                w *= GetGaussianWeight(offset.z);
                w *= saturate(penum * invEstimatedPenumbra); // avoid umbra leaking inside a wide penumbra

                result = result + (w == 0.0f ? S(0.0f) : s * w);
                sumx += w;

                w *= Div(pixelSize, pixelSize + penum);
                w *= IsLit(penum) ? 0.0f : 1.0f;

                penumbra += w == 0.0f ? 0.0f : penum * w;
                sumy += w;
            }

            result = Div(result, sumx);
            penumbra = sumy == 0.0f ? centerPenumbra : Div(penumbra, sumy);

            if (FIRST_PASS || c.gStabilizationStrength != 0.0f)
                gOut_Penumbra.Store(px, py, penumbra);
            gOut_Shadow.Store(px, py, PackShadow(result));
        }
}

// ================================================================================================ TemporalStabilization
inline uint32_t PackViewZAndHistoryLength(float viewZ, float historyLength) {
    uint32_t p = asuint(viewZ) & ~7u;
    uint32_t h = (uint32_t)(historyLength + 0.5f);
    p |= h < 7u ? h : 7u;
    return p;
}

template <bool TRANSLUCENT>
void TemporalStabilization(const PassIO& io) {
    typedef SigmaType<TRANSLUCENT> ST;
    typedef typename ST::type S;
    const SigmaCB& c = *(const SigmaCB*)io.constants;
    const Tex& gIn_ViewZ = io.t[0];
    const Tex& gIn_Mv = io.t[1];
    const Tex& gIn_Penumbra = io.t[2];
    const Tex& gIn_Shadow = io.t[3];
    const Tex& gIn_History = io.t[4];
    const Tex& gIn_HistoryLength = io.t[5];
    const Tex& gIn_Tiles = io.t[6];
    Tex& gOut_Shadow = io.t[7];
    Tex& gOut_HistoryLength = io.t[8];
    const int rw = c.gRectSizeMinusOne[0], rh = c.gRectSizeMinusOne[1];

#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py <= rh; py++)
        for (int px = 0; px <= rw; px++) {
            float isSky = gIn_Tiles.Load(px >> 4, py >> 4).x;
            auto sShadow = [&](int x, int y) -> S { return UnpackShadow(ST::From(gIn_Shadow.Load(clamp(x, 0, rw), clamp(y, 0, rh)))); };
            auto sPenumbra = [&](int x, int y) { return gIn_Penumbra.Load(clamp(x, 0, rw), clamp(y, 0, rh)).x; };

            float centerPenumbra = sPenumbra(px, py);
            float viewZ = UnpackViewZ(c, gIn_ViewZ.Load(px, py).x);
            if (isSky != 0.0f || viewZ > c.gDenoisingRange)
                continue;

            float2 pixelUv = float2(float(px) + 0.5f, float(py) + 0.5f) * c.gRectSizeInv;
            float tileValue = TextureCubicY(gIn_Tiles, pixelUv * c.gResolutionScale);
            bool isHardShadow = tileValue == 0.0f || centerPenumbra == 0.0f;
            if (isHardShadow) {
                gOut_Shadow.Store(px, py, PackShadow(sShadow(px, py)));
                gOut_HistoryLength.StoreUint(px, py, PackViewZAndHistoryLength(viewZ, SIGMA_MAX_ACCUM_FRAME_NUM));
                continue;
            }

            // Local variance over 5x5
            float sumw = 0.0f;
            S m1 = S(0.0f), m2 = S(0.0f), input = S(0.0f);
            for (int j = 0; j <= BORDER * 2; j++)
                for (int i = 0; i <= BORDER * 2; i++) {
                    int x = px - BORDER + i, y = py - BORDER + j;
                    S s = sShadow(x, y);
                    float w = 1.0f;
                    if (i == BORDER && j == BORDER)
                        input = s;
                    else {
                        float penum = sPenumbra(x, y);
                        w = AreBothLitOrUnlit(centerPenumbra, penum);
                        w *= GetGaussianWeight(length(Div(float2(float(i - BORDER), float(j - BORDER)), float(BORDER))));
                    }
                    m1 = Mad(s, w, m1);
                    m2 = m2 + s * s * w;
                    sumw += w;
                }
            m1 = Div(m1, sumw);
            m2 = Div(m2, sumw);
            S sigma = StdDev(m1, m2);

            // Current and previous positions
            float3 Xv = Geometry::ReconstructViewPosition(pixelUv, c.gFrustum, viewZ, c.gOrthoMode);
            float3 X = Geometry::RotateVectorInverse(c.gWorldToView, Xv);

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

            // History length: 2x2 gather of (viewZ | length) words
            Filtering::Bilinear smbBilinearFilter = Filtering::GetBilinearFilter(smbPixelUv, c.gRectSizePrev);
            int bx = (int)smbBilinearFilter.origin.x, by = (int)smbBilinearFilter.origin.y;
            uint32_t prevData[4] = {gIn_HistoryLength.FetchUintClamped(bx, by), gIn_HistoryLength.FetchUintClamped(bx + 1, by), gIn_HistoryLength.FetchUintClamped(bx, by + 1),
                gIn_HistoryLength.FetchUintClamped(bx + 1, by + 1)};
            float4 prevViewZ = float4(asfloat(prevData[0] & ~7u), asfloat(prevData[1] & ~7u), asfloat(prevData[2] & ~7u), asfloat(prevData[3] & ~7u));
            float4 prevHistoryLength = float4(float(prevData[0] & 7u), float(prevData[1] & 7u), float(prevData[2] & 7u), float(prevData[3] & 7u));

            float frustumSize = GetFrustumSize(c.gMinRectDimMulUnproject, c.gOrthoMode, viewZ);
            float disocclusionThreshold = GetDisocclusionThreshold(NRD_DISOCCLUSION_THRESHOLD, frustumSize, 1.0f);
            disocclusionThreshold *= IsInScreenNearest(smbPixelUv);
            disocclusionThreshold -= NRD_EPS;

            float3 Xvprev = Geometry::AffineTransform(c.gWorldToViewPrev, Xprev);
            float4 smbPlaneDist = abs(prevViewZ - Xvprev.z);
            float4 smbOcclusion = step(smbPlaneDist, float4(disocclusionThreshold));

            float4 smbOcclusionWeights = Filtering::GetBilinearCustomWeights(smbBilinearFilter, smbOcclusion);
            float historyLength = Filtering::ApplyBilinearCustomWeights(prevHistoryLength.x, prevHistoryLength.y, prevHistoryLength.z, prevHistoryLength.w, smbOcclusionWeights);

            // Sample history. NB: the weights sum to <= 1, so this test never passes and the fetch is always custom-weight bilinear
            // (kept exactly as in the reference, SIGMA_TemporalStabilization.hlsli:151)
            bool isCatRomAllowed = sum(smbOcclusionWeights) > 3.5f;
            HistoryFilter hf = MakeHistoryFilter(saturate(smbPixelUv) * c.gRectSizePrev, smbOcclusionWeights, isCatRomAllowed);
            S history = ST::From(FetchHistoryColor(hf, gIn_History));
            history = Saturate(history);
            history = UnpackShadow(history);

            // Clamp history
            sigma = sigma * lerp(SIGMA_TS_SIGMA_SCALE, 1.0f, Rcp(1.0f + historyLength));
            S inputMin = m1 - sigma, inputMax = m1 + sigma;
            S historyClamped = Clamp(history, inputMin, inputMax);

            // Antilag (on the shadow channel only)
            float antilag = fabsf(ST::X(historyClamped) - ST::X(history));
            antilag = Math::Sqrt01(antilag);
            antilag = saturate(1.0f - antilag);
            historyLength *= antilag;

            float historyWeight = Div(historyLength, 1.0f + historyLength);
            float streetMagic = 0.6f * historyWeight * antilag;
            historyClamped = lerp(historyClamped, history, streetMagic);

            S result = lerp(input, historyClamped, min(c.gStabilizationStrength, historyWeight));
            historyLength = min(historyLength + 1.0f, SIGMA_MAX_ACCUM_FRAME_NUM);

            gOut_Shadow.Store(px, py, PackShadow(result));
            gOut_HistoryLength.StoreUint(px, py, PackViewZAndHistoryLength(viewZ, historyLength));
        }
}

// ================================================================================================ SplitScreen
template <bool TRANSLUCENT>
void SplitScreen(const PassIO& io) {
    typedef SigmaType<TRANSLUCENT> ST;
    typedef typename ST::type S;
    const SigmaCB& c = *(const SigmaCB*)io.constants;
    uint32_t k = 0;
    const Tex& gIn_ViewZ = io.t[k++];
    const Tex& gIn_Penumbra = io.t[k++];
    const Tex* gIn_Shadow_Translucency = TRANSLUCENT ? &io.t[k++] : nullptr;
    Tex& gOut_Shadow = io.t[k++];
    for (int py = 0; py <= c.gRectSizeMinusOne[1]; py++)
        for (int px = 0; px <= c.gRectSizeMinusOne[0]; px++) {
            float2 pixelUv = float2(float(px) + 0.5f, float(py) + 0.5f) * c.gRectSizeInv;
            if (pixelUv.x > c.gSplitScreen)
                continue;
            float viewZ = UnpackViewZ(c, gIn_ViewZ.Load(px, py).x);
            S s = TRANSLUCENT ? ST::From(gIn_Shadow_Translucency->Load(px, py)) : S(IsLit(gIn_Penumbra.Load(px, py).x) ? 1.0f : 0.0f);
            gOut_Shadow.Store(px, py, s * (viewZ < c.gDenoisingRange ? 1.0f : 0.0f));
        }
}

} // namespace

const PassEntry* GetSigmaPasses(uint32_t& n) {
    static const PassEntry k[] = {
        {"SIGMA_Shadow_ClassifyTiles.cs", ClassifyTiles<false>},
        {"SIGMA_SmoothTiles.cs", SmoothTiles},
        {"SIGMA_Copy.cs", Copy},
        {"SIGMA_Shadow_Blur.cs", Blur<true, false>},
        {"SIGMA_Shadow_PostBlur.cs", Blur<false, false>},
        {"SIGMA_Shadow_TemporalStabilization.cs", TemporalStabilization<false>},
        {"SIGMA_Shadow_SplitScreen.cs", SplitScreen<false>},
        {"SIGMA_ShadowTranslucency_ClassifyTiles.cs", ClassifyTiles<true>},
        {"SIGMA_ShadowTranslucency_Blur.cs", Blur<true, true>},
        {"SIGMA_ShadowTranslucency_PostBlur.cs", Blur<false, true>},
        {"SIGMA_ShadowTranslucency_TemporalStabilization.cs", TemporalStabilization<true>},
        {"SIGMA_ShadowTranslucency_SplitScreen.cs", SplitScreen<true>},
    };
    n = sizeof(k) / sizeof(k[0]);
    return k;
}

} // namespace orc
