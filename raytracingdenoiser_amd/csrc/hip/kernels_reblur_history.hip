// REBLUR HistoryFix and TemporalStabilization as HIP kernels for gfx950.
//   HistoryFix              reference Shaders/Include/REBLUR_HistoryFix.hlsli:11-463
//   TemporalStabilization   reference Shaders/Include/REBLUR_TemporalStabilization.hlsli:11-367
//
// MI355X mapping. Both are LDS-tile stencils over a scalar luma plane plus per-pixel work:
//   HistoryFix: 36x12 LDS tiles (32x8 outputs + halo 2) of the fast-history luma (one per signal) feed the 5x5 moments;
//     the 5x5-minus-corners history reconstruction is a strided gather that only runs where fewer than
//     historyFixFrameNum frames are accumulated (disocclusions), the 9x9 anti-firefly box (off by default) reads global.
//   TemporalStabilization: 34x10 LDS tiles (halo 1) of luma = .x of the YCoCg signal, decoded once per workgroup, feed
//     the 3x3 moments and the min/max clamp; the stabilised-luma history comes through Catmull-Rom fetches of an R16F plane.
// LDS rows are padded to an odd dword count (37 / 35) so that the two rows a wave touches start in different banks.
#include "passes.h"
#include "reblur_device.h"

namespace nrdhip {

constexpr int TILE_X = 32;
constexpr int TILE_Y = 8;

static const char* CheckSupportedHistory(const ReblurCB& c) {
    if (c.gRectOrigin.x != 0 || c.gRectOrigin.y != 0) // the executor moves the rect of the guide inputs to (0, 0) and zeroes this field (executor.hip "shifted rect")
        return "internal error: a pass was handed a non-zero rectOrigin";
    if (c.gOrthoMode != 0.0f)
        return "REBLUR: orthographic projection is not supported (SURVEY.md section 8c)";
    return nullptr;
}

NRD_D bool BlockHasGeometry(const Plane& tiles, int blockX, int blockY) {
    const int tileY = (blockY * TILE_Y) >> 4, tileX0 = (blockX * TILE_X) >> 4;
    bool any = false;
    for (int t = 0; t < TILE_X / 16; t++)
        if (tileX0 + t < tiles.w && tileY < tiles.h)
            any |= LoadR8Unorm(tiles, tileX0 + t, tileY) == 0.0f;
    return any;
}

// ================================================================================================ HistoryFix
namespace hf {
constexpr int BORDER = 2;
constexpr int BUF_X = TILE_X + 2 * BORDER; // 36
constexpr int BUF_Y = TILE_Y + 2 * BORDER; // 12
constexpr int BUF_STRIDE = BUF_X + 1;      // 37
} // namespace hf

struct HfPlanes {
    Plane tiles, normalRoughness, data1, viewZ, inDiff, inSpec, inDiffFast, inSpecFast, outDiff, outSpec, outDiffFast, outSpecFast;
    Plane inDiffSh, inSpecSh, outDiffSh, outSpecSh; // SH family
    NormalRoughnessGuide decodedNR; // executor's decoded guides of IN_NORMAL_ROUGHNESS (reblur_device.h NormalRoughnessGuide)
};

struct HfPixel {
    int px, py, tx, ty;
    float viewZ, roughness, materialID, frustumSize;
    float3 N, Nv, Xv;
    float2 pixelUv;
};

// PERF = REBLUR_PERFORMANCE_MODE (reference REBLUR_HistoryFix.hlsli:88-90 / 139-141 / 292-294 / 338-340, REBLUR_Config.hlsli:236-237)
template <bool IS_SPEC, bool DIFF, bool SPEC, bool PERF, int KIND, bool SH>
NRD_D typename ReblurSignal<KIND>::type HistoryFixSignal(const ReblurCB& c, const HfPlanes& P, const HfPixel& s, typename ReblurSignal<KIND>::type sig, float frameNum, float strideBase,
    const Plane& gIn_Signal, const Plane& gIn_Fast, const Plane& gOut_Fast, const float* s_Luma, float4& sh, const Plane& gIn_Sh) { // SH: the SH1 plane rides along (specular: .xyz only)
    typedef ReblurSignal<KIND> Sig;
    typedef typename Sig::type S;
    const int rw = c.gRectSizeMinusOne.x, rh = c.gRectSizeMinusOne.y;
    const float smc = GetSpecMagicCurve(s.roughness);

    float stride = strideBase * (frameNum < c.gHistoryFixFrameNum ? 1.0f : 0.0f);
    if (IS_SPEC)
        stride *= Lerp(0.5f, 1.0f, smc);
    stride = floorf(stride);

    if (stride != 0.0f) {
        const int stridei = (int)(stride + 0.5f);
        const float nonLinearAccumSpeed = Rcp(1.0f + frameNum);
        const float r = IS_SPEC ? s.roughness : 1.0f;
        const float2 rectSizeInv = ToF2(c.gRectSizeInv);
        const float4 hitDistParams = ToF4(c.gHitDistParams);

        float normalWeightParam = GetNormalWeightParam(nonLinearAccumSpeed, c.gLobeAngleFraction, r);
        float2 geometryWeightParams = GetGeometryWeightParams(c.gPlaneDistSensitivity, s.frustumSize, s.Xv, s.Nv);
        float2 relaxedRoughnessWeightParams = GetRelaxedRoughnessWeightParams(s.roughness * s.roughness, Sqrt(c.gRoughnessFraction));

        float hitDistScale = GetHitDistanceNormalization(s.viewZ, hitDistParams, r);
        float hitDist = ExtractHitDist(sig) * hitDistScale;
        float hitDistFactor = GetHitDistFactor(hitDist, s.frustumSize);
        float2 hitDistanceWeightParams = GetHitDistanceWeightParams(hitDistFactor, nonLinearAccumSpeed, r);

        float sumw = 1.0f + frameNum;
        if (PERF)
            sumw = 1.0f + Rcp(1.0f + c.gMaxAccumulatedFrameNum) - nonLinearAccumSpeed;
        sig = sig * sumw;
        if (SH)
            sh = F4(sh.x * sumw, sh.y * sumw, sh.z * sumw, IS_SPEC ? sh.w : sh.w * sumw);

        for (int j = -2; j <= 2; j++) {
            for (int i = -2; i <= 2; i++) {
                if ((i == 0 && j == 0) || (abs(i) + abs(j) == 4))
                    continue;

                float2 uv = s.pixelUv + F2(float(i), float(j)) * stride * rectSizeInv;
                int sx = ClampI(s.px + i * stridei, 0, rw), sy = ClampI(s.py + j * stridei, 0, rh);

                float zs = UnpackViewZ(c, LoadR32F(P.viewZ, sx, sy));
                float materialIDs;
                float4 Ns = LoadDecodedNormalRoughness(P.decodedNR, sx, sy, materialIDs);

                float angle = AcosApprox(Dot(Xyz(Ns), s.N));
                float3 Xvs = ReconstructViewPosition(uv, ToF4(c.gFrustum), zs, NRD_ORTHO_MODE(c));

                float w = IsInScreenNearest(uv);
                w *= ComputeWeight(Dot(s.Nv, Xvs), geometryWeightParams.x, geometryWeightParams.y);
                w *= CompareMaterials(s.materialID, materialIDs, IS_SPEC ? c.gSpecMinMaterial : c.gDiffMinMaterial) ? 1.0f : 0.0f;
                w *= ComputeExponentialWeight(angle, normalWeightParam, 0.0f);
                if (IS_SPEC)
                    w *= ComputeExponentialWeight(Ns.w * Ns.w, relaxedRoughnessWeightParams.x, relaxedRoughnessWeightParams.y);

                if (!PERF) {
                    float2 d1 = LoadData1<DIFF, SPEC>(P.data1, sx, sy);
                    w *= 1.0f + (IS_SPEC ? d1.y : d1.x);
                }

                S smp = Sig::Load(gIn_Signal, sx, sy);
                smp = Select(w == 0.0f, Sig::Zero(), smp);

                float hs = ExtractHitDist(smp) * hitDistScale;
                float hsFactor = GetHitDistFactor(hs, s.frustumSize);
                w *= ComputeExponentialWeight(hsFactor, hitDistanceWeightParams.x, hitDistanceWeightParams.y);

                if (IS_SPEC) {
                    float d = Div(Abs(hitDist - hs), Max(hitDist, hs) + 0.001f);
                    float b = LinearStep(0.03f, 0.05f, s.roughness);
                    w *= SmoothStep(0.2f + b, 0.05f + b, d);
                }

                sumw += w;
                sig = Mad(smp, w, sig);
                if (SH) {
                    float4 t = LoadRGBA16F(gIn_Sh, sx, sy);
                    t = Select(w == 0.0f, F4(0.0f), t);
                    sh = F4(sh.x + t.x * w, sh.y + t.y * w, sh.z + t.z * w, IS_SPEC ? sh.w : sh.w + t.w * w);
                }
            }
        }

        sumw = PositiveRcp(sumw);
        sig = sig * sumw;
        if (SH)
            sh = F4(sh.x * sumw, sh.y * sumw, sh.z * sumw, IS_SPEC ? sh.w : sh.w * sumw);
    }

    // Local variance of the fast history over 5x5 from the LDS tile
    float center = s_Luma[(s.ty + hf::BORDER) * hf::BUF_STRIDE + s.tx + hf::BORDER];
    float m1 = center, m2 = center * center;

    float f = Sat(Div(frameNum, c.gHistoryFixFrameNum + NRD_EPS));
    if (IS_SPEC)
        f = Lerp(1.0f, f, smc);
    center = Lerp(GetLuma(sig), center, f);
    Sig::StoreFast(gOut_Fast, s.px, s.py, center);

#pragma unroll
    for (int j = 0; j <= 4; j++) {
#pragma unroll
        for (int i = 0; i <= 4; i++) {
            if (i == 2 && j == 2)
                continue;
            float d = s_Luma[(s.ty + j) * hf::BUF_STRIDE + s.tx + i];
            m1 += d;
            m2 += d * d;
        }
    }

    float luma = GetLuma(sig);

    // Anti-firefly: 9x9 minus the central 3x3 (off by default; compiled out in the occlusion family, REBLUR_USE_ANTIFIREFLY = 0)
    if (KIND == SIGNAL_RADIANCE && c.gAntiFirefly != 0.0f) {
        float am1 = 0.0f, am2 = 0.0f;
        const int R = PERF ? 3 : REBLUR_ANTI_FIREFLY_FILTER_RADIUS;
        for (int j = -R; j <= R; j++)
            for (int i = -R; i <= R; i++) {
                if (abs(i) <= 1 && abs(j) <= 1)
                    continue;
                float d = LoadR16F(gIn_Fast, ClampI(s.px + i, 0, rw), ClampI(s.py + j, 0, rh));
                am1 += d;
                am2 += d * d;
            }
        float invNorm = Rcp(float((R * 2 + 1) * (R * 2 + 1) - 3 * 3));
        am1 *= invNorm;
        am2 *= invNorm;
        float sigma = Sqrt(Abs(am2 - am1 * am1)) * REBLUR_ANTI_FIREFLY_SIGMA_SCALE;
        luma = Clamp(luma, am1 - sigma, am1 + sigma);
    }

    m1 = m1 * (1.0f / 25.0f);
    m2 = m2 * (1.0f / 25.0f);
    float sigma = Sqrt(Abs(m2 - m1 * m1)) * (KIND != SIGNAL_RADIANCE ? REBLUR_COLOR_CLAMPING_SIGMA_SCALE_OCCLUSION : REBLUR_COLOR_CLAMPING_SIGMA_SCALE);
    float lumaClamped = Clamp(luma, m1 - sigma, m1 + sigma);
    luma = Lerp(lumaClamped, luma, Rcp(1.0f + (c.gMaxFastAccumulatedFrameNum < c.gMaxAccumulatedFrameNum ? 1.0f : 0.0f) * frameNum * 2.0f));

    if (SH) {
        float k = GetLumaScale(Length(Xyz(sh)), luma);
        sh = F4(sh.x * k, sh.y * k, sh.z * k, sh.w);
    }
    return ChangeLuma(sig, luma);
}

#ifndef NRD_REBLUR_HF_ROTATE
#define NRD_REBLUR_HF_ROTATE 1 // (0.0857 -> 0.0658 ms at 1440p, r04_q: the young pixels this pass reconstructs are the columns entering the screen -- on ONE XCD in the striped order)
#endif
template <bool DIFF, bool SPEC, bool PERF, int KIND, bool SH>
__global__ __launch_bounds__(TILE_X* TILE_Y, SH ? 0 : NRD_WAVES_REBLUR_HF) void ReblurHistoryFixKernel(ReblurCB c, HfPlanes P, RowRange rr) {
    typedef ReblurSignal<KIND> Sig;
    typedef typename Sig::type S;
    __shared__ float s_DiffLuma[DIFF ? hf::BUF_Y * hf::BUF_STRIDE : 1];
    __shared__ float s_SpecLuma[SPEC ? hf::BUF_Y * hf::BUF_STRIDE : 1];

    const int tx = threadIdx.x % TILE_X, ty = threadIdx.x / TILE_X;
    const int blockY = BlockTileY(rr);
    const int tileX = NRD_REBLUR_HF_ROTATE ? BlockTileXRotated(rr, blockY) : BlockTileX(rr); // (A/B switch: passes.h BlockTileXRotated)
    const int px = tileX * TILE_X + tx, py = blockY * TILE_Y + ty;
    const int rw = c.gRectSizeMinusOne.x, rh = c.gRectSizeMinusOne.y;

    if (!BlockHasGeometry(P.tiles, tileX, blockY))
        return;
    // the pixel's own inputs, requested in front of the tile fill (see ReblurTemporalStabilizationKernel; this pass ran 32 % above its L1-resident time)
    const int qx = min(px, rw), qy = min(max(py, 0), rh);
    const float preTile = LoadR8Unorm(P.tiles, qx >> 4, qy >> 4);
    const float preViewZ = LoadR32F(P.viewZ, qx, qy);
    float preMaterialID;
    const float4 preNormalAndRoughness = LoadDecodedNormalRoughness(P.decodedNR, qx, qy, preMaterialID);
    const float2 preData1 = LoadData1<DIFF, SPEC>(P.data1, qx, qy);
    S preDiff = Sig::Zero(), preSpec = Sig::Zero();
    if (DIFF)
        preDiff = Sig::Load(P.inDiff, qx, qy);
    if (SPEC)
        preSpec = Sig::Load(P.inSpec, qx, qy);
    {
        const int baseX = tileX * TILE_X - hf::BORDER, baseY = blockY * TILE_Y - hf::BORDER;
        for (int i = threadIdx.x; i < hf::BUF_X * hf::BUF_Y; i += TILE_X * TILE_Y) {
            int lx = i % hf::BUF_X, ly = i / hf::BUF_X;
            int gx = ClampI(baseX + lx, 0, rw), gy = ClampI(baseY + ly, 0, rh);
            if (DIFF)
                s_DiffLuma[ly * hf::BUF_STRIDE + lx] = Sig::LoadFast(P.inDiffFast, gx, gy);
            if (SPEC)
                s_SpecLuma[ly * hf::BUF_STRIDE + lx] = Sig::LoadFast(P.inSpecFast, gx, gy);
        }
    }
    __syncthreads();

    if (px > rw || py > rh || py < rr.rowBegin || py >= rr.rowEnd)
        return;
    if (preTile != 0.0f)
        return;
    HfPixel s;
    s.viewZ = UnpackViewZ(c, preViewZ);
    if (s.viewZ > c.gDenoisingRange)
        return;

    s.px = px, s.py = py, s.tx = tx, s.ty = ty;
    s.materialID = preMaterialID;
    float4 normalAndRoughness = preNormalAndRoughness;
    s.N = Xyz(normalAndRoughness);
    s.roughness = normalAndRoughness.w;
    s.frustumSize = GetFrustumSize(c.gMinRectDimMulUnproject, NRD_ORTHO_MODE(c), s.viewZ);
    s.pixelUv = F2(float(px) + 0.5f, float(py) + 0.5f) * ToF2(c.gRectSizeInv);
    s.Xv = ReconstructViewPosition(s.pixelUv, ToF4(c.gFrustum), s.viewZ, NRD_ORTHO_MODE(c));
    s.Nv = RotateVectorInverse(c.gViewToWorld, s.N);
    float2 frameNum = preData1;
    float2 stride = F2(Div(c.gHistoryFixBasePixelStride, 2.0f + frameNum.x), Div(c.gHistoryFixBasePixelStride, 2.0f + frameNum.y));

    if (DIFF) {
        float4 diffSh = F4(0.0f);
        if (SH)
            diffSh = LoadRGBA16F(P.inDiffSh, px, py);
        S diff = HistoryFixSignal<false, DIFF, SPEC, PERF, KIND, SH>(c, P, s, preDiff, frameNum.x, stride.x, P.inDiff, P.inDiffFast, P.outDiffFast, s_DiffLuma, diffSh, P.inDiffSh);
        Sig::Store(P.outDiff, px, py, diff);
        if (SH)
            StoreRGBA16F(P.outDiffSh, px, py, diffSh);
    }
    if (SPEC) {
        float4 specSh = F4(0.0f);
        if (SH)
            specSh = LoadRGBA16F(P.inSpecSh, px, py);
        S spec = HistoryFixSignal<true, DIFF, SPEC, PERF, KIND, SH>(c, P, s, preSpec, frameNum.y, stride.y, P.inSpec, P.inSpecFast, P.outSpecFast, s_SpecLuma, specSh, P.inSpecSh);
        Sig::Store(P.outSpec, px, py, spec);
        if (SH)
            StoreRGBA16F(P.outSpecSh, px, py, specSh);
    }
}

template <bool DIFF, bool SPEC, bool PERF, int KIND, bool SH>
static const char* LaunchHistoryFix(const PassArgs& a) {
    const ReblurCB& c = *(const ReblurCB*)a.constants;
    if (const char* err = CheckSupportedHistory(c))
        return err;
    HfPlanes P = {};
    uint32_t k = 0;
    P.tiles = a.planes[k++];
    P.normalRoughness = a.planes[k++];
    if (const char* err = MakeNormalRoughnessGuide(a, P.decodedNR))
        return err;
    P.data1 = a.planes[k++];
    P.viewZ = a.planes[k++];
    if (DIFF) P.inDiff = a.planes[k++];
    if (SPEC) P.inSpec = a.planes[k++];
    if (DIFF) P.inDiffFast = a.planes[k++];
    if (SPEC) P.inSpecFast = a.planes[k++];
    if (DIFF && SH) P.inDiffSh = a.planes[k++];
    if (SPEC && SH) P.inSpecSh = a.planes[k++];
    if (DIFF) P.outDiff = a.planes[k++];
    if (SPEC) P.outSpec = a.planes[k++];
    if (DIFF) P.outDiffFast = a.planes[k++];
    if (SPEC) P.outSpecFast = a.planes[k++];
    if (DIFF && SH) P.outDiffSh = a.planes[k++];
    if (SPEC && SH) P.outSpecSh = a.planes[k++];
    if (k != a.planesNum)
        return "REBLUR history fix: unexpected resource count";
    RowGrid g = GridForRows(c.gRectSizeMinusOne.x + 1, c.gRectSizeMinusOne.y + 1, TILE_X, TILE_Y, a.rowBegin, a.rowEnd);
    LaunchPass(a, (ReblurHistoryFixKernel<DIFF, SPEC, PERF, KIND, SH>), g.grid, dim3(TILE_X * TILE_Y), c, P, MakeRowRange(g));
    return nullptr;
}

// ================================================================================================ TemporalStabilization
namespace ts {
constexpr int BORDER = 1;
constexpr int BUF_X = TILE_X + 2 * BORDER; // 34
constexpr int BUF_Y = TILE_Y + 2 * BORDER; // 10
constexpr int BUF_STRIDE = BUF_X + 1;      // 35
} // namespace ts

struct TsPlanes {
    Plane tiles, normalRoughness, viewZ, data1, data2, inDiff, inSpec, historyDiffLuma, historySpecLuma, inSpecHitDistForTracking, mv, outInternalData, outDiff, outSpec, outDiffLuma,
        outSpecLuma;
    Plane inDiffSh, inSpecSh, outDiffSh, outSpecSh; // SH family
    NormalRoughnessGuide decodedNR;
    Plane baseColorMetalness; // IN_BASECOLOR_METALNESS (RGBA8_UNORM), only read when the specular MV modification is on
};

// 3x3 luma statistics from the LDS tile: centre luma (min/max clamped), mean, sigma
template <bool PERF> // PERF: no RCRS clamp of the centre luma (reference REBLUR_TemporalStabilization.hlsli:118-135)
NRD_D void LumaStats(const ReblurCB& c, const float* s_Luma, int tx, int ty, float& luma, float& m1, float& sigma) {
    luma = s_Luma[(ty + ts::BORDER) * ts::BUF_STRIDE + tx + ts::BORDER];
    float M1 = luma, M2 = luma * luma, mn = NRD_INF, mx = -NRD_INF;
#pragma unroll
    for (int j = 0; j <= 2; j++) {
#pragma unroll
        for (int i = 0; i <= 2; i++) {
            if (i == 1 && j == 1)
                continue;
            float d = s_Luma[(ty + j) * ts::BUF_STRIDE + tx + i];
            M1 += d;
            M2 += d * d;
            mn = Min(mn, d);
            mx = Max(mx, d);
        }
    }
    M1 = M1 * (1.0f / 9.0f);
    M2 = M2 * (1.0f / 9.0f);
    m1 = M1;
    sigma = Sqrt(Abs(M2 - M1 * M1));
    if (!PERF && c.gMaxBlurRadius != 0.0f)
        luma = Clamp(luma, mn, mx);
}

// KIND: radiance or directional occlusion (the occlusion family has no stabilisation pass)
// MVMOD: the specular motion-vector modification is on this frame (CommonSettings::isBaseColorMetalnessAvailable); compiled out otherwise
template <bool DIFF, bool SPEC, bool PERF, bool SH, int KIND, bool MVMOD>
__global__ __launch_bounds__(TILE_X* TILE_Y, NRD_WAVES_REBLUR_TS) void ReblurTemporalStabilizationKernel(ReblurCB c, TsPlanes P, RowRange rr) {
    typedef ReblurSignal<KIND> Sig;
    typedef typename Sig::type S;
    __shared__ float s_DiffLuma[DIFF ? ts::BUF_Y * ts::BUF_STRIDE : 1];
    __shared__ float s_SpecLuma[SPEC ? ts::BUF_Y * ts::BUF_STRIDE : 1];

    const int tx = threadIdx.x % TILE_X, ty = threadIdx.x / TILE_X;
    const int blockY = BlockTileY(rr, NRD_ALT_TILE_ORDER != 0); // (A/B: kernels_reblur_spatial.hip)
    const int px = BlockTileX(rr) * TILE_X + tx, py = blockY * TILE_Y + ty;
    const int rw = c.gRectSizeMinusOne.x, rh = c.gRectSizeMinusOne.y;

    if (!BlockHasGeometry(P.tiles, BlockTileX(rr), blockY))
        return;
    // The pixel's own inputs do not depend on the luma tile: they are requested in FRONT of the tile fill, so that one memory latency covers both. Behind the
    // barrier (rounds 1-3) they were a second, fully exposed latency in front of a third (the history fetch at the reprojected position): this pass ran 26 % above
    // the time of a build whose loads all hit the L1 (profiles/r04_c_reblur_ds_uniform_*_kernel_stats.txt). Clamped coordinates: threads outside the rect leave later.
    const int qx = min(px, rw), qy = min(max(py, 0), rh);
    const float preTile = LoadR8Unorm(P.tiles, qx >> 4, qy >> 4);
    const float preViewZ = LoadR32F(P.viewZ, qx, qy);
    const float4 preMv = LoadRGBA16F(P.mv, qx, qy);
    float preMaterialID;
    const float4 preNormalAndRoughness = LoadDecodedNormalRoughness(P.decodedNR, qx, qy, preMaterialID);
    const float2 preData1 = LoadData1<DIFF, SPEC>(P.data1, qx, qy);
    const uint32_t preData2 = SPEC ? LoadR32U(P.data2, qx, qy) : LoadR8U(P.data2, qx, qy);
    S preDiff = Sig::Zero(), preSpec = Sig::Zero();
    if (DIFF)
        preDiff = Sig::Load(P.inDiff, qx, qy);
    if (SPEC)
        preSpec = Sig::Load(P.inSpec, qx, qy);
    {
        const int baseX = BlockTileX(rr) * TILE_X - ts::BORDER, baseY = blockY * TILE_Y - ts::BORDER;
        for (int i = threadIdx.x; i < ts::BUF_X * ts::BUF_Y; i += TILE_X * TILE_Y) {
            int lx = i % ts::BUF_X, ly = i / ts::BUF_X;
            int gx = ClampI(baseX + lx, 0, rw), gy = ClampI(baseY + ly, 0, rh);
            if (DIFF)
                s_DiffLuma[ly * ts::BUF_STRIDE + lx] = GetLuma(Sig::Load(P.inDiff, gx, gy));
            if (SPEC)
                s_SpecLuma[ly * ts::BUF_STRIDE + lx] = GetLuma(Sig::Load(P.inSpec, gx, gy));
        }
    }
    __syncthreads();

    if (px > rw || py > rh || py < rr.rowBegin || py >= rr.rowEnd)
        return;
    if (preTile != 0.0f)
        return;
    const float viewZ = UnpackViewZ(c, preViewZ);
    if (viewZ > c.gDenoisingRange)
        return;

    const float2 rectSizeInv = ToF2(c.gRectSizeInv), rectSizePrev = ToF2(c.gRectSizePrev);
    const float3 cameraDelta = ToF3(c.gCameraDelta);
    const float4 frustum = ToF4(c.gFrustum), frustumPrev = ToF4(c.gFrustumPrev);

    // Position
    float2 pixelUv = F2(float(px) + 0.5f, float(py) + 0.5f) * rectSizeInv;
    float3 Xv = ReconstructViewPosition(pixelUv, frustum, viewZ, NRD_ORTHO_MODE(c));
    float3 X = RotateVector(c.gViewToWorld, Xv);

    // Previous position and surface motion uv
    float4 inMv = preMv;
    float3 mv = F3(inMv.x, inMv.y, inMv.z) * F3(c.gMvScale.x, c.gMvScale.y, c.gMvScale.z);
    float3 Xprev = X;
    float2 smbPixelUv = pixelUv + F2(mv.x, mv.y);
    if (c.gMvScale.w == 0.0f) {
        if (c.gMvScale.z == 0.0f)
            mv.z = AffineTransform(c.gWorldToViewPrev, X).z - viewZ;
        float viewZprev = viewZ + mv.z;
        float3 Xvprevlocal = ReconstructViewPosition(smbPixelUv, frustumPrev, viewZprev, NRD_ORTHO_MODE(c));
        Xprev = RotateVectorInverse(c.gWorldToViewPrev, Xvprevlocal) + cameraDelta;
    } else {
        Xprev = Xprev + mv;
        smbPixelUv = GetScreenUv(c.gWorldToClipPrev, Xprev);
    }

    float materialID = preMaterialID;
    float4 normalAndRoughness = preNormalAndRoughness;
    float3 N = Xyz(normalAndRoughness);
    float roughness = normalAndRoughness.w;

    uint32_t bits;
    float2 data1 = preData1;
    float2 data2 = UnpackData2(preData2, bits);

    // Surface motion footprint
    Bilinear smbBilinearFilter = GetBilinearFilter(smbPixelUv, rectSizePrev);
    float4 smbOcclusion = F4((bits & 1u) ? 1.0f : 0.0f, (bits & 2u) ? 1.0f : 0.0f, (bits & 4u) ? 1.0f : 0.0f, (bits & 8u) ? 1.0f : 0.0f);
    float4 smbOcclusionWeights = GetBilinearCustomWeights(smbBilinearFilter, smbOcclusion);
    bool smbAllowCatRom = Sum(smbOcclusion) > 3.5f && !PERF; // REBLUR_USE_CATROM_FOR_SURFACE_MOTION_IN_TS
    float smbFootprintQuality = ApplyBilinearFilter(smbOcclusion.x, smbOcclusion.y, smbOcclusion.z, smbOcclusion.w, smbBilinearFilter);
    smbFootprintQuality = Sqrt01(smbFootprintQuality);

    const float2 smbSamplePos = Sat(smbPixelUv) * rectSizePrev;

    if (DIFF) {
        float diffLuma, diffLumaM1, diffLumaSigma;
        LumaStats<PERF>(c, s_DiffLuma, tx, ty, diffLuma, diffLumaM1, diffLumaSigma);

        HistoryFilter smbFilter = MakeHistoryFilter(smbSamplePos, smbOcclusionWeights, smbAllowCatRom, P.historyDiffLuma);
        float smbDiffLumaHistory = FetchHistoryR16F(smbFilter, P.historyDiffLuma);
        smbDiffLumaHistory = Max(smbDiffLumaHistory, 0.0f);

        float diffAntilag = ComputeAntilag(c, smbDiffLumaHistory, diffLumaM1, diffLumaSigma, smbFootprintQuality * data1.x);

        float2 diffTemporalAccumulationParams = GetTemporalAccumulationParams(c, smbFootprintQuality, data1.x);
        float diffHistoryWeight = diffTemporalAccumulationParams.x;
        diffHistoryWeight *= diffAntilag;
        diffHistoryWeight *= pixelUv.x >= c.gSplitScreen ? 1.0f : 0.0f;
        diffHistoryWeight *= smbPixelUv.x >= c.gSplitScreenPrev ? 1.0f : 0.0f;

        smbDiffLumaHistory = ColorClamp(diffLumaM1, diffLumaSigma * diffTemporalAccumulationParams.y, smbDiffLumaHistory);
        float diffLumaStabilized = Lerp(diffLuma, smbDiffLumaHistory, Min(diffHistoryWeight, c.gStabilizationStrength));

        S diff = preDiff;
        diff = ChangeLuma(diff, diffLumaStabilized);
        Sig::Store(P.outDiff, px, py, diff);
        StoreR16F(P.outDiffLuma, px, py, diffLumaStabilized);
        if (SH) {
            float4 diffSh = LoadRGBA16F(P.inDiffSh, px, py);
            float k = GetLumaScale(Length(Xyz(diffSh)), diffLumaStabilized);
            StoreRGBA16F(P.outDiffSh, px, py, F4(diffSh.x * k, diffSh.y * k, diffSh.z * k, diffSh.w));
        }

        data1.x += 1.0f;
        float diffMinAccumSpeed = Min(data1.x, c.gHistoryFixFrameNum);
        data1.x = Lerp(diffMinAccumSpeed, data1.x, diffAntilag);
    }

    if (SPEC) {
        float specLuma, specLumaM1, specLumaSigma;
        LumaStats<PERF>(c, s_SpecLuma, tx, ty, specLuma, specLumaM1, specLumaSigma);

        float virtualHistoryAmount = data2.x;
        float curvature = data2.y;

        S spec = preSpec;
        float hitDistForTracking = ExtractHitDist(spec) * GetHitDistanceNormalization(viewZ, ToF4(c.gHitDistParams), roughness);
        if (c.gSpecPrepassBlurRadius != 0.0f)
            hitDistForTracking = Min(hitDistForTracking, LoadR16F(P.inSpecHitDistForTracking, px, py));

        float3 V = GetViewVector(c, X);
        float3 Xvirtual = GetXvirtual(hitDistForTracking, curvature, X, Xprev, N, V, roughness);
        float2 vmbPixelUv = GetScreenUv(c.gWorldToClipPrev, Xvirtual);
        vmbPixelUv = Select(materialID == c.gCameraAttachedReflectionMaterialID, pixelUv, vmbPixelUv);

        // Modify MVs if requested (reference REBLUR_TemporalStabilization.hlsli:250-285): where the surface is mostly specular, IN_MV is bent towards
        // the motion of the reflected world. A uniform branch: x = 2 unless CommonSettings::isBaseColorMetalnessAvailable
        if (MVMOD && c.gSpecProbabilityThresholdsForMvModification.x < 1.0f) {
            float NoV = Abs(Dot(N, V));
            float4 baseColorMetalness = LoadRGBA8Unorm(P.baseColorMetalness, px, py);
            float3 albedo, Rf0;
            ConvertBaseColorMetalnessToAlbedoRf0(Xyz(baseColorMetalness), baseColorMetalness.w, albedo, Rf0);
            float3 Fenv = EnvironmentTerm_Rtg(Rf0, NoV, roughness);
            float lumSpec = Luminance(Fenv);
            float lumDiff = Luminance(F3(albedo.x * (1.0f - Fenv.x), albedo.y * (1.0f - Fenv.y), albedo.z * (1.0f - Fenv.z)));
            float specProb = Div(lumSpec, lumDiff + lumSpec + NRD_EPS);
            float f = SmoothStep(c.gSpecProbabilityThresholdsForMvModification.x, c.gSpecProbabilityThresholdsForMvModification.y, specProb);
            f *= 1.0f - GetSpecMagicCurve(roughness);
            f *= 1.0f - Sqrt01(Abs(curvature));
            if (f != 0.0f) {
                float3 specMv = Xvirtual - X; // world-space delta
                if (c.gMvScale.w == 0.0f) {
                    specMv.x = vmbPixelUv.x - pixelUv.x;
                    specMv.y = vmbPixelUv.y - pixelUv.y;
                    specMv.z = AffineTransform(c.gWorldToViewPrev, Xvirtual).z - viewZ;
                }
                // only .xy for 2D, .xyz for 2.5D and 3D MVs
                float3 newMv = F3(Div(specMv.x, c.gMvScale.x), Div(specMv.y, c.gMvScale.y), c.gMvScale.z == 0.0f ? inMv.z : Div(specMv.z, c.gMvScale.z));
                inMv.x = Lerp(inMv.x, newMv.x, f);
                inMv.y = Lerp(inMv.y, newMv.y, f);
                inMv.z = Lerp(inMv.z, newMv.z, f);
                StoreRGBA16F(P.mv, px, py, inMv);
            }
        }

        HistoryFilter smbFilter = MakeHistoryFilter(smbSamplePos, smbOcclusionWeights, smbAllowCatRom, P.historySpecLuma);
        float smbSpecLumaHistory = FetchHistoryR16F(smbFilter, P.historySpecLuma);

        Bilinear vmbBilinearFilter = GetBilinearFilter(vmbPixelUv, rectSizePrev);
        float4 vmbOcclusion = F4((bits & 16u) ? 1.0f : 0.0f, (bits & 32u) ? 1.0f : 0.0f, (bits & 64u) ? 1.0f : 0.0f, (bits & 128u) ? 1.0f : 0.0f);
        float4 vmbOcclusionWeights = GetBilinearCustomWeights(vmbBilinearFilter, vmbOcclusion);
        bool vmbAllowCatRom = Sum(vmbOcclusion) > 3.5f && !PERF; // REBLUR_USE_CATROM_FOR_VIRTUAL_MOTION_IN_TS
        float vmbFootprintQuality = ApplyBilinearFilter(vmbOcclusion.x, vmbOcclusion.y, vmbOcclusion.z, vmbOcclusion.w, vmbBilinearFilter);
        vmbFootprintQuality = Sqrt01(vmbFootprintQuality);

        HistoryFilter vmbFilter = MakeHistoryFilter(Sat(vmbPixelUv) * rectSizePrev, vmbOcclusionWeights, vmbAllowCatRom, P.historySpecLuma);
        float vmbSpecLumaHistory = FetchHistoryR16F(vmbFilter, P.historySpecLuma);

        smbSpecLumaHistory = Max(smbSpecLumaHistory, 0.0f);
        vmbSpecLumaHistory = Max(vmbSpecLumaHistory, 0.0f);

        float specLumaHistory = Lerp(smbSpecLumaHistory, vmbSpecLumaHistory, virtualHistoryAmount);

        float footprintQuality = Lerp(smbFootprintQuality, vmbFootprintQuality, virtualHistoryAmount);
        float specAntilag = ComputeAntilag(c, specLumaHistory, specLumaM1, specLumaSigma, footprintQuality * data1.y);

        float2 specTemporalAccumulationParams = GetTemporalAccumulationParams(c, footprintQuality, data1.y);
        float specHistoryWeight = specTemporalAccumulationParams.x;
        specHistoryWeight *= specAntilag;
        specHistoryWeight *= pixelUv.x >= c.gSplitScreen ? 1.0f : 0.0f;
        specHistoryWeight *= virtualHistoryAmount != 1.0f ? (smbPixelUv.x >= c.gSplitScreenPrev ? 1.0f : 0.0f) : 1.0f;
        specHistoryWeight *= virtualHistoryAmount != 0.0f ? (vmbPixelUv.x >= c.gSplitScreenPrev ? 1.0f : 0.0f) : 1.0f;

        float responsiveFactor = RemapRoughnessToResponsiveFactor(c, roughness);
        float smc = GetSpecMagicCurve(roughness);
        float acceleration = Lerp(smc, 1.0f, 0.5f + responsiveFactor * 0.5f);
        specHistoryWeight *= materialID == c.gStrandMaterialID ? 0.5f : acceleration;

        specLumaHistory = ColorClamp(specLumaM1, specLumaSigma * specTemporalAccumulationParams.y, specLumaHistory);
        float specLumaStabilized = Lerp(specLuma, specLumaHistory, Min(specHistoryWeight, c.gStabilizationStrength));

        spec = ChangeLuma(spec, specLumaStabilized);
        Sig::Store(P.outSpec, px, py, spec);
        StoreR16F(P.outSpecLuma, px, py, specLumaStabilized);
        if (SH) {
            float4 specSh = LoadRGBA16F(P.inSpecSh, px, py);
            float k = GetLumaScale(Length(Xyz(specSh)), specLumaStabilized);
            StoreRGBA16F(P.outSpecSh, px, py, F4(specSh.x * k, specSh.y * k, specSh.z * k, specSh.w));
        }

        data1.y += 1.0f;
        float specMinAccumSpeed = Min(data1.y, c.gHistoryFixFrameNum);
        data1.y = Lerp(specMinAccumSpeed, data1.y, specAntilag);
    }

    StoreR16U(P.outInternalData, px, py, PackInternalData(data1.x, data1.y, materialID));
}

template <bool DIFF, bool SPEC, bool PERF, bool SH, int KIND>
static const char* LaunchTemporalStabilization(const PassArgs& a) {
    const ReblurCB& c = *(const ReblurCB*)a.constants;
    if (const char* err = CheckSupportedHistory(c))
        return err;
    TsPlanes P = {};
    uint32_t k = 0;
    P.tiles = a.planes[k++];
    P.normalRoughness = a.planes[k++];
    if (const char* err = MakeNormalRoughnessGuide(a, P.decodedNR))
        return err;
    if (SPEC) P.baseColorMetalness = a.planes[k++]; // a dummy unless CommonSettings::isBaseColorMetalnessAvailable
    if (SPEC && c.gSpecProbabilityThresholdsForMvModification.x < 1.0f && (!P.baseColorMetalness.ptr || a.formats[k - 1] != (uint32_t)FORMAT_RGBA8_UNORM))
        return "REBLUR TemporalStabilization: IN_BASECOLOR_METALNESS must be bound as RGBA8_UNORM when isBaseColorMetalnessAvailable is set";
    P.viewZ = a.planes[k++];
    P.data1 = a.planes[k++];
    P.data2 = a.planes[k++];
    if (DIFF) P.inDiff = a.planes[k++];
    if (SPEC) P.inSpec = a.planes[k++];
    if (DIFF) P.historyDiffLuma = a.planes[k++];
    if (SPEC) P.historySpecLuma = a.planes[k++];
    if (SPEC) P.inSpecHitDistForTracking = a.planes[k++];
    if (DIFF && SH) P.inDiffSh = a.planes[k++];
    if (SPEC && SH) P.inSpecSh = a.planes[k++];
    P.mv = a.planes[k++];
    P.outInternalData = a.planes[k++];
    if (DIFF) P.outDiff = a.planes[k++];
    if (SPEC) P.outSpec = a.planes[k++];
    if (DIFF) P.outDiffLuma = a.planes[k++];
    if (SPEC) P.outSpecLuma = a.planes[k++];
    if (DIFF && SH) P.outDiffSh = a.planes[k++];
    if (SPEC && SH) P.outSpecSh = a.planes[k++];
    if (k != a.planesNum)
        return "REBLUR temporal stabilization: unexpected resource count";
    RowGrid g = GridForRows(c.gRectSizeMinusOne.x + 1, c.gRectSizeMinusOne.y + 1, TILE_X, TILE_Y, a.rowBegin, a.rowEnd);
    if (SPEC && c.gSpecProbabilityThresholdsForMvModification.x < 1.0f)
        LaunchPass(a, (ReblurTemporalStabilizationKernel<DIFF, SPEC, PERF, SH, KIND, SPEC>), g.grid, dim3(TILE_X * TILE_Y), c, P, MakeRowRange(g));
    else
        LaunchPass(a, (ReblurTemporalStabilizationKernel<DIFF, SPEC, PERF, SH, KIND, false>), g.grid, dim3(TILE_X * TILE_Y), c, P, MakeRowRange(g));
    return nullptr;
}

#define REBLUR_HISTORY_FAMILY(NAME, D, S)                                                                         \
    {"REBLUR_" NAME "_HistoryFix.cs", LaunchHistoryFix<D, S, false, false, false>},                               \
    {"REBLUR_" NAME "_TemporalStabilization.cs", LaunchTemporalStabilization<D, S, false, false, 0>},                \
    {"REBLUR_Perf_" NAME "_HistoryFix.cs", LaunchHistoryFix<D, S, true, false, false>},                           \
    {"REBLUR_Perf_" NAME "_TemporalStabilization.cs", LaunchTemporalStabilization<D, S, true, false, 0>},            \
    {"REBLUR_" NAME "Sh_HistoryFix.cs", LaunchHistoryFix<D, S, false, false, true>},                              \
    {"REBLUR_" NAME "Sh_TemporalStabilization.cs", LaunchTemporalStabilization<D, S, false, true, 0>},               \
    {"REBLUR_Perf_" NAME "Sh_HistoryFix.cs", LaunchHistoryFix<D, S, true, false, true>},                          \
    {"REBLUR_Perf_" NAME "Sh_TemporalStabilization.cs", LaunchTemporalStabilization<D, S, true, true, 0>},           \
    {"REBLUR_" NAME "Occlusion_HistoryFix.cs", LaunchHistoryFix<D, S, false, true, false>},                       \
    {"REBLUR_Perf_" NAME "Occlusion_HistoryFix.cs", LaunchHistoryFix<D, S, true, true, false>},

const PassEntry* GetReblurHistoryPasses(uint32_t& num) {
    static const PassEntry k[] = {
        REBLUR_HISTORY_FAMILY("Diffuse", true, false)
        REBLUR_HISTORY_FAMILY("Specular", false, true)
        REBLUR_HISTORY_FAMILY("DiffuseSpecular", true, true)
        {"REBLUR_DiffuseDirectionalOcclusion_HistoryFix.cs", LaunchHistoryFix<true, false, false, 2, false>},
        {"REBLUR_Perf_DiffuseDirectionalOcclusion_HistoryFix.cs", LaunchHistoryFix<true, false, true, 2, false>},
        {"REBLUR_DiffuseDirectionalOcclusion_TemporalStabilization.cs", LaunchTemporalStabilization<true, false, false, false, 2>},
        {"REBLUR_Perf_DiffuseDirectionalOcclusion_TemporalStabilization.cs", LaunchTemporalStabilization<true, false, true, false, 2>},
    };
    num = sizeof(k) / sizeof(k[0]);
    return k;
}

} // namespace nrdhip
