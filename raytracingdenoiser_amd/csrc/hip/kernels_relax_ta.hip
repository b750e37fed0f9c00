// RELAX TemporalAccumulation and HistoryClamping as HIP kernels for gfx950.
//   TemporalAccumulation   reference Shaders/Include/RELAX_TemporalAccumulation.hlsli:10-931
//   HistoryClamping        reference Shaders/Include/RELAX_HistoryClamping.hlsli:10-330
//
// MI355X mapping.
//   TemporalAccumulation: 32x8-pixel workgroups; a 34x10 LDS tile (halo 1) of (normal.xyz, specular hitT), decoded once
//     per workgroup, feeds the 3x3 normal average / min-hitT and the curvature edges. The history reads are data-dependent
//     gathers (12-tap footprint validation + Catmull-Rom as 5 bilinear fetches per RGBA16F history plane, surface AND
//     virtual motion): they go through L2; consecutive lanes reproject to neighbouring texels, so a wave's gather still
//     lands in a handful of 256-byte row segments.
//   HistoryClamping: 36x12 LDS tiles (halo 2) of responsive history in YCoCg and of the noisy input + validity, one pair
//     per signal, feed the 5x5 moments; everything else is per-pixel arithmetic.
// LDS rows are padded to an odd float4 count (35 / 37) so the rows a wave touches start in different banks.
#include "relax_device.h"
#include <climits>
#include <cstdio>

namespace nrdhip {

namespace {

constexpr int TILE_X = RELAX_TILE_X;
constexpr int TILE_Y = RELAX_TILE_Y;

// ================================================================================================ TemporalAccumulation
namespace ta {
constexpr int BORDER = 1;
constexpr int BUF_X = TILE_X + 2 * BORDER; // 34
constexpr int BUF_Y = TILE_Y + 2 * BORDER; // 10
constexpr int BUF_STRIDE = BUF_X + 1;      // 35
} // namespace ta

// The surface-motion window (MODE 1; same scheme as kernels_reblur_ta.hip): the texels of the previous frame a workgroup's pixels reproject to, staged in LDS
constexpr int WIN_W = 64; // at most 64: one lane per column when the window is filled
constexpr int WIN_H = 16;

struct TaPlanes {
    uint32_t* historyReach; // passes.h PassArgs::historyReachWord (multi-GPU hosts; nullptr otherwise)
    Plane tileFlags;      // executor scratch, one byte per workgroup tile (passes.h)
    int winMaxW, winMaxH; // largest box the window kernel accepts (NRD_HIP_TA_WINDOW_LIMIT shrinks it for tests)
    Plane tiles, mv, normalRoughness, viewZ, prevNormalRoughness, prevViewZ, prevSpecHitDist, prevHistoryLength, prevMaterialID, disocclusionThresholdMix;
    Plane outSpecHitDist, outHistoryLength, outSpecReprojectionConfidence;
    Plane decodedNR; // executor's float4 cache of normalRoughness (reblur_device.h "decoded guides")
    SignalPlanes spec, diff;
};

// MODE 0 = plain kernel. MODE 1 = window kernel: everything read at the surface-motion position (the 12 previous-depth / material taps, the bilinear previous
// normal, FOUR 12-texel Catmull-Rom history fetches, history length, previous hit distance: ~450 bytes per pixel through the L1) comes from one rectangle of
// the previous frame per workgroup, copied into LDS with coalesced row loads -- every texel once -- and read from there; bit-identical by construction (same
// texels at the same clamped coordinates, same arithmetic). A tile whose rectangle does not fit sets its byte of P.tileFlags and is done by MODE 2, the plain
// kernel behind a flag test, launched right after. The SH histories (2x2 bilinear only) and the virtual-motion fetches stay in global memory.
// MAT: material tests compiled in (the launcher picks MAT = false when both minimum materials are >= 3: IDs are 0..3, every comparison then holds; inside the unrolled
// footprint loops a run-time test is if-converted into compare + select and saves nothing)
template <bool DIFF, bool SPEC, bool SH, int MODE, bool MAT = true>
__device__ __forceinline__ void RelaxTemporalAccumulationTile(const RelaxCB& cArg, TaPlanes P, const RowRange& rows, const int tileX, const int blockY) {
    __shared__ float4 s_NormalSpecHitT[ta::BUF_Y * ta::BUF_STRIDE];
    constexpr int WIN_TEXELS = MODE == 1 ? WIN_W * WIN_H : 1;
    __shared__ float s_WinZ[WIN_TEXELS];       // packed previous viewZ
    __shared__ uint32_t s_WinN[WIN_TEXELS];    // previous normal / roughness (RGBA8)
    __shared__ uint32_t s_WinMisc[WIN_TEXELS]; // previous history length (bits 0-7), material ID (8-15), specular hit distance (fp16, 16-31)
    __shared__ uint2 s_WinDiffPrev[(MODE == 1 && DIFF) ? WIN_W * WIN_H : 1], s_WinDiffFast[(MODE == 1 && DIFF) ? WIN_W * WIN_H : 1]; // RGBA16F history texels, undecoded
    __shared__ uint2 s_WinSpecPrev[(MODE == 1 && SPEC) ? WIN_W * WIN_H : 1], s_WinSpecFast[(MODE == 1 && SPEC) ? WIN_W * WIN_H : 1];
    __shared__ int s_WinBox[4][4];
    // The constant block + up to 35 planes need far more than the 102 SGPRs there are; left alone the compiler spills scalars into
    // VGPR lanes (v_writelane / v_readlane + hazard nops on every use). The body reads the constants from an LDS copy instead
    // (uniform-address ds_read, off the VALU), re-read per phase; only the prologue touches the kernel-argument copy.
    __shared__ RelaxCB s_Constants;

    // SGPR diet (planes.h): one (w, h) for every full-resolution plane, one pitch for all RGBA16F pool planes; verified by the launcher
    {
        const Plane size = P.viewZ, rgba16 = SPEC ? P.spec.prev : P.diff.prev;
        ShareSize(P.decodedNR, size), ShareSize(P.mv, size), ShareSize(P.prevNormalRoughness, size), ShareSize(P.prevViewZ, size), ShareSize(P.prevSpecHitDist, size);
        ShareSize(P.prevHistoryLength, size), ShareSize(P.prevMaterialID, size), ShareSize(P.outSpecHitDist, size), ShareSize(P.outHistoryLength, size), ShareSize(P.outSpecReprojectionConfidence, size);
        SignalPlanes* sig[2] = {&P.spec, &P.diff};
#pragma unroll
        for (int s = 0; s < 2; s++) {
            ShareSize(sig[s]->in, size), ShareSize(sig[s]->inSh, size);
            ShareLayout(sig[s]->prev, rgba16), ShareLayout(sig[s]->prevSh, rgba16), ShareLayout(sig[s]->fast, rgba16), ShareLayout(sig[s]->fastSh, rgba16);
            ShareLayout(sig[s]->out, rgba16), ShareLayout(sig[s]->outSh, rgba16), ShareLayout(sig[s]->outFast, rgba16), ShareLayout(sig[s]->outFastSh, rgba16);
        }
    }

    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int px = tileX * TILE_X + tx, py = blockY * TILE_Y + ty;
    const int rectW = cArg.shared.gRectSize.x, rectH = cArg.shared.gRectSize.y;

    // (workgroups of the XCD-aware grid may lie beyond the frame: they have no pixels and no flag)
    uint8_t* const tileFlag = (MODE == 1 && tileX < P.tileFlags.w && blockY < P.tileFlags.h) ? P.tileFlags.ptr + (uint32_t)blockY * P.tileFlags.pitch + (uint32_t)tileX : nullptr;
    if (!RelaxBlockHasGeometry(P.tiles, tileX, blockY)) { // uniform per workgroup
        if (MODE == 1 && threadIdx.x == 0 && tileFlag)
            *tileFlag = 0;
        return;
    }

    // The pixel's own guides do not depend on the LDS tile: requested in FRONT of the fill, so that one memory latency covers both. Behind the barrier they were
    // a second exposed latency in front of the third and fourth (previous-frame footprints at the surface-motion, then at the virtual-motion position): the
    // pass ran 39 % above the time of a build whose loads all hit the L1 (profiles/r04_c_relax_ds_sh_uniform_*_kernel_stats.txt).
    const int qx = min(px, rectW - 1), qy = min(max(py, 0), rectH - 1);
    const float preTile = LoadR8Unorm(P.tiles, qx >> 4, qy >> 4);
    const float preViewZ = LoadR32F(P.viewZ, qx, qy);
    const float4 preMv = LoadRGBA16F(P.mv, qx, qy);
    float preMaterialID;
    const float4 preNormalRoughness = LoadDecodedNormalRoughness(P.decodedNR, qx, qy, preMaterialID);

    // preload (normal, specular hitT) at rect-clamped coordinates
    for (int idx = threadIdx.x; idx < ta::BUF_X * ta::BUF_Y; idx += 256) {
        int lx = idx % ta::BUF_X, ly = idx / ta::BUF_X;
        int gx = ClampI(tileX * TILE_X - ta::BORDER + lx, 0, rectW - 1), gy = ClampI(blockY * TILE_Y - ta::BORDER + ly, 0, rectH - 1);
        float4 v = LoadDecodedNormalRoughness(P.decodedNR, gx, gy);
        if (SPEC)
            v.w = LoadRGBA16F(P.spec.in, gx, gy).w;
        s_NormalSpecHitT[ly * ta::BUF_STRIDE + lx] = v;
    }
    {
        // constants: kernel-argument segment (cArg is the first argument, offset 0) -> LDS, one dword per thread
        const uint32_t __attribute__((address_space(4)))* kernarg = (const uint32_t __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
        if (threadIdx.x < sizeof(RelaxCB) / 4)
            ((uint32_t*)&s_Constants)[threadIdx.x] = kernarg[threadIdx.x];
    }
    __syncthreads();
    const RelaxCB& c = s_Constants;
#define NRD_CONSTANTS_PHASE() asm volatile("" ::: "memory")

    // MODE 1: every thread stays until the window is filled; a thread without a pixel to denoise runs the prologue on a position clamped into the rect
    // (its loads stay legal, its values are never used) and is left out of the bounding box
    const int lpx = MODE == 1 ? min(px, rectW - 1) : px, lpy = MODE == 1 ? min(py, rectH - 1) : py;
    bool active = !(px >= rectW || py >= rectH || py < rows.rowBegin || py >= rows.rowEnd);
    if (MODE != 1 && !active)
        return;
    active = active && preTile == 0.0f; // (qx, qy) = (lpx, lpy) for every thread that stays
    if (MODE != 1 && !active)
        return;
    const float currentLinearZ = RelaxUnpackViewZ(c, preViewZ);
    active = active && !(currentLinearZ > c.shared.gDenoisingRange);
    if (MODE != 1 && !active)
        return;

    auto Shared = [&](int dx, int dy) { return s_NormalSpecHitT[(ty + ta::BORDER + dy) * ta::BUF_STRIDE + (tx + ta::BORDER + dx)]; };

    float currentMaterialID = preMaterialID;
    float4 currentNormalRoughness = preNormalRoughness;
    const float3 currentNormal = Xyz(currentNormalRoughness);
    const float currentRoughness = currentNormalRoughness.w;

    const float3 currentWorldPos = GetCurrentWorldPosFromPixelPos(c, px, py, currentLinearZ);
    const float3 currentViewVector = currentWorldPos;
    const float3 V = -Normalize(currentViewVector);
    const float NoV = Abs(Dot(currentNormal, V));

    const float2 rectSize = F2(float(rectW), float(rectH));
    const float2 rectSizeInv = ToF2(c.shared.gRectSizeInv);
    const float2 rectSizePrev = ToF2(c.shared.gRectSizePrev);
    const float3 cameraDelta = ToF3(c.shared.gCameraDelta);
    const float2 resolutionScalePrev = rectSizePrev * ToF2(c.shared.gResourceSizeInvPrev);

    // previous position
    const float2 pixelUv = F2(float(px) + 0.5f, float(py) + 0.5f) * rectSizeInv;
    float4 mvRaw = preMv;
    float3 mv = Xyz(mvRaw) * ToF3(c.shared.gMvScale);
    float3 prevWorldPos = currentWorldPos;
    float2 prevUVSMB = pixelUv + F2(mv.x, mv.y);
    if (c.shared.gMvScale.w == 0.0f) {
        if (c.shared.gMvScale.z == 0.0f)
            mv.z = AffineTransform(c.shared.gWorldToViewPrev, currentWorldPos).z - currentLinearZ;
        prevWorldPos = GetPreviousWorldPosFromClipSpaceXY(c, prevUVSMB * 2.0f - 1.0f, currentLinearZ + mv.z) + cameraDelta;
    } else {
        prevWorldPos = prevWorldPos + mv;
        prevUVSMB = GetScreenUv(c.shared.gWorldToClipPrev, prevWorldPos);
    }

    TrackHistoryReach(P.historyReach, active ? HistoryReachRows(prevUVSMB.y, rectSizePrev.y, py) : 0.0f); // (multi-GPU hosts; a null word otherwise: reblur_device.h)

    // noisy inputs
    const float3 diffuseIllumination = DIFF ? Xyz(LoadRGBA16F(P.diff.in, lpx, lpy)) : F3(0.0f);
    const float4 diffuseSH = (DIFF && SH) ? LoadRGBA16F(P.diff.inSh, lpx, lpy) : F4(0.0f);
    const float4 specularIllumination = SPEC ? LoadRGBA16F(P.spec.in, lpx, lpy) : F4(0.0f);
    const float4 specularSH = (SPEC && SH) ? LoadRGBA16F(P.spec.inSh, lpx, lpy) : F4(0.0f);

    // average normal and min hit distance in 3x3
    float hitTM1 = Shared(0, 0).w;
    float minHitDist3x3 = hitTM1 == 0.0f ? NRD_INF : hitTM1;
    float3 currentNormalAveraged = currentNormal;
#pragma unroll
    for (int i = -1; i <= 1; i++)
#pragma unroll
        for (int j = -1; j <= 1; j++) {
            if (i == 0 && j == 0)
                continue;
            float4 normalSpecHitT = Shared(i, j);
            minHitDist3x3 = Min(minHitDist3x3, normalSpecHitT.w == 0.0f ? NRD_INF : normalSpecHitT.w);
            currentNormalAveraged = currentNormalAveraged + Xyz(normalSpecHitT);
        }
    currentNormalAveraged = currentNormalAveraged * (1.0f / 9.0f);

    const float currentRoughnessModified = SPEC ? GetModifiedRoughnessFromNormalVariance(currentRoughness, currentNormalAveraged) : 0.0f;

    const float specular1stMoment = Luminance(Xyz(specularIllumination));
    const float specular2ndMoment = specular1stMoment * specular1stMoment;
    const float diffuse1stMoment = Luminance(diffuseIllumination);
    const float diffuse2ndMoment = diffuse1stMoment * diffuse1stMoment;

    NRD_CONSTANTS_PHASE();
    // surface parallax
    const float smbParallaxInPixels1 = ComputeParallaxInPixels(prevWorldPos + cameraDelta, prevUVSMB, c.shared.gWorldToClipPrev, rectSize);
    const float smbParallaxInPixels2 = ComputeParallaxInPixels(prevWorldPos - cameraDelta, pixelUv, c.shared.gWorldToClip, rectSize);
    const float smbParallaxInPixelsMax = Max(smbParallaxInPixels1, smbParallaxInPixels2);
    const float smbParallaxInPixelsMin = Min(smbParallaxInPixels1, smbParallaxInPixels2);

    const float pixelSize = PixelRadiusToWorld(c.shared.gUnproject, c.shared.gOrthoMode, 1.0f, currentLinearZ);

    // disocclusion threshold
    float disocclusionThresholdMix = 0.0f;
    if (currentMaterialID == c.shared.gStrandMaterialID)
        disocclusionThresholdMix = Div(pixelSize, pixelSize + c.shared.gStrandThickness); // NRD_GetNormalizedStrandThickness (reference NRD.hlsli:1158-1161)
    if (c.shared.gHasDisocclusionThresholdMix)
        disocclusionThresholdMix = LoadR8Unorm(P.disocclusionThresholdMix, lpx, lpy);
    const float disocclusionThreshold = Lerp(c.shared.gDisocclusionThreshold, c.shared.gDisocclusionThresholdAlternate, disocclusionThresholdMix);

    NRD_CONSTANTS_PHASE();
    // ---------------------------------------------------------------- surface motion based history
    float footprintQuality, historyLength, SMBReprojectionFound;
    float hitDist = 0.0f, curvature = 0.0f, hitDistFocused = 0.0f; // SPEC: the early virtual-motion geometry (set inside the surface-motion section)
    float2 prevUVVMB = F2(0.0f, 0.0f), vmbPixelPosFloat = F2(0.0f, 0.0f), vmbOriginF = F2(0.0f, 0.0f), vmbBilinearWeights = F2(0.0f, 0.0f);
    int vbx = 0, vby = 0;
    float zQuad[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
    bool quadInterior = false;
    float4 prevDiffuseIllumAnd2ndMomentSMB = F4(0.0f), prevDiffuseSH = F4(0.0f), prevDiffuseResponsiveSH = F4(0.0f);
    float3 prevDiffuseResponsiveSMB = F3(0.0f);
    float4 prevSpecularIllumAnd2ndMomentSMB = F4(0.0f), prevSpecularSMBSH = F4(0.0f), prevSpecularSMBResponsiveSH = F4(0.0f);
    float3 prevSpecularResponsiveSMB = F3(0.0f);
    float prevReflectionHitTSMB = 0.0f;
    {
        const float3 smbNormal = Normalize(currentNormalAveraged);
        const float2 prevPixelPosFloat = prevUVSMB * rectSizePrev;
        const float2 originF = Floor(prevPixelPosFloat - 0.5f);
        const int bx = (int)originF.x, by = (int)originF.y;
        const float2 bilinearWeights = F2(Frac(prevPixelPosFloat.x - 0.5f), Frac(prevPixelPosFloat.y - 0.5f));
        const float minMaterialID = Min(c.shared.gSpecMinMaterial, c.shared.gDiffMinMaterial);
        // no material reads at all while the material test is off (IDs are 0..3: a minimum >= 3 makes every comparison hold; the library default is 4)
        const bool compareMaterials = MAT && minMaterialID < 3.0f;

        // ---- MODE 1: the window. Every surface-motion read of this pixel is a texel at a clamped coordinate within [bx - 1, bx + 2] x [by - 1, by + 2]
        int wx0 = 0, wy0 = 0;
        const int W1 = P.prevViewZ.w - 1, H1 = P.prevViewZ.h - 1; // every plane staged here has the size of prevViewZ (checked by the launcher)
        if (MODE == 1) {
            int loX = INT_MAX, loY = INT_MAX, hiX = INT_MIN, hiY = INT_MIN;
            if (active)
                loX = ClampI(bx - 1, 0, W1), hiX = ClampI(bx + 2, 0, W1), loY = ClampI(by - 1, 0, H1), hiY = ClampI(by + 2, 0, H1);
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                loX = min(loX, __shfl_xor(loX, m)), loY = min(loY, __shfl_xor(loY, m));
                hiX = max(hiX, __shfl_xor(hiX, m)), hiY = max(hiY, __shfl_xor(hiY, m));
            }
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            if (lane == 0)
                s_WinBox[wave][0] = loX, s_WinBox[wave][1] = loY, s_WinBox[wave][2] = hiX, s_WinBox[wave][3] = hiY;
            __syncthreads();
            loX = min(min(s_WinBox[0][0], s_WinBox[1][0]), min(s_WinBox[2][0], s_WinBox[3][0]));
            loY = min(min(s_WinBox[0][1], s_WinBox[1][1]), min(s_WinBox[2][1], s_WinBox[3][1]));
            hiX = max(max(s_WinBox[0][2], s_WinBox[1][2]), max(s_WinBox[2][2], s_WinBox[3][2]));
            hiY = max(max(s_WinBox[0][3], s_WinBox[1][3]), max(s_WinBox[2][3], s_WinBox[3][3]));
            const bool empty = hiX < loX; // no pixel to denoise in this tile
            const int bw = hiX - loX + 1, bh = hiY - loY + 1;
            if (empty || bw > P.winMaxW || bh > P.winMaxH) { // uniform
                if (threadIdx.x == 0 && tileFlag)
                    *tileFlag = empty ? 0 : 1; // 1: the fallback kernel (MODE 2) does this tile
                return;
            }
            if (threadIdx.x == 0 && tileFlag)
                *tileFlag = 0;
            wx0 = loX, wy0 = loY;
            // fill: one wave per row of the box, one lane per column -- coalesced row segments, every texel once
            if (lane < bw)
                for (int r = wave; r < bh; r += 4) {
                    const int x = wx0 + lane, y = wy0 + r, o = r * WIN_W + lane;
                    s_WinZ[o] = LoadR32F(P.prevViewZ, x, y);
                    s_WinN[o] = *TexelPtr<const uint32_t>(P.prevNormalRoughness, x, y);
                    uint32_t misc = *TexelPtr<const uint8_t>(P.prevHistoryLength, x, y);
                    if (compareMaterials)
                        misc |= (uint32_t)*TexelPtr<const uint8_t>(P.prevMaterialID, x, y) << 8;
                    if (SPEC)
                        misc |= LoadR16U(P.prevSpecHitDist, x, y) << 16;
                    s_WinMisc[o] = misc;
                    if (DIFF)
                        s_WinDiffPrev[o] = *TexelPtr<const uint2>(P.diff.prev, x, y), s_WinDiffFast[o] = *TexelPtr<const uint2>(P.diff.fast, x, y);
                    if (SPEC)
                        s_WinSpecPrev[o] = *TexelPtr<const uint2>(P.spec.prev, x, y), s_WinSpecFast[o] = *TexelPtr<const uint2>(P.spec.fast, x, y);
                }
            __syncthreads();
            if (!active)
                return;
        }
        // window index of the texel at the clamped coordinate
        auto Win = [&](int x, int y) { return (ClampI(y, 0, H1) - wy0) * WIN_W + (ClampI(x, 0, W1) - wx0); };

        const float frustumSize = pixelSize * float(rectW < rectH ? rectW : rectH);
        const float disocclusionThresholdSlopeScale = Rcp(Lerp(Lerp(0.05f, 1.0f, NoV), 1.0f, Sat(smbParallaxInPixelsMax * (1.0f / 30.0f))));
        float4 smbDisocclusionThreshold = F4(Sat(disocclusionThreshold * disocclusionThresholdSlopeScale) * frustumSize);
        smbDisocclusionThreshold = smbDisocclusionThreshold * IsInScreenBilinear(originF, rectSizePrev);
        smbDisocclusionThreshold = smbDisocclusionThreshold - NRD_EPS;

        const float prevViewPosZ = AffineTransform(c.shared.gWorldToViewPrev, prevWorldPos).z;
        // The 12 previous-depth taps (rows of 2 + 4 + 4 + 2 texels around the bilinear origin): four row loads when no coordinate needs clamping
        // (reblur_device.h "row-vector fetches")
        float zRows[4][4];
        const bool footprintInterior = MODE != 1 && FootprintIsInterior(P.prevViewZ, bx - 1, by - 1, 4, 4);
        if (footprintInterior) {
            const float2 r0 = LoadRowR32Fx2(P.prevViewZ, bx, by - 1), r3 = LoadRowR32Fx2(P.prevViewZ, bx, by + 2);
            const float4 r1 = LoadRowR32Fx4(P.prevViewZ, bx - 1, by), r2 = LoadRowR32Fx4(P.prevViewZ, bx - 1, by + 1);
            zRows[0][1] = r0.x, zRows[0][2] = r0.y, zRows[3][1] = r3.x, zRows[3][2] = r3.y;
            zRows[1][0] = r1.x, zRows[1][1] = r1.y, zRows[1][2] = r1.z, zRows[1][3] = r1.w;
            zRows[2][0] = r2.x, zRows[2][1] = r2.y, zRows[2][2] = r2.z, zRows[2][3] = r2.w;
            zRows[0][0] = zRows[0][3] = zRows[3][0] = zRows[3][3] = 0.0f;
        }
        // ---- early virtual-motion geometry (SPEC). Curvature, the focused hit distance and the virtual-motion position depend on THIS frame's data only, so they are computed
        // here, behind the requests of the surface-motion depth rows, and the previous-depth quad at the virtual position is requested before the surface-motion taps are
        // validated: its latency overlaps with that work instead of following the surface-motion history fetches (same expressions, same order: value-identical)
        if (SPEC) {
            hitDist = minHitDist3x3 == NRD_INF ? 0.0f : minHitDist3x3;

            // curvature along the direction of motion
            {
                float2 deltaUv = prevUVSMB - GetScreenUv(c.shared.gWorldToClipPrev, prevWorldPos + cameraDelta);
                deltaUv = deltaUv * rectSize;
                deltaUv = Div(deltaUv, Max(smbParallaxInPixels1, 1.0f / 256.0f));

                float3 n10, x10, n01, x01;
                {
                    float3 x = GetCurrentWorldPosFromClipSpaceXY(c, (pixelUv + F2(1.0f, 0.0f) * rectSizeInv) * 2.0f - 1.0f, 1.0f);
                    float3 v = Normalize(-x);
                    x10 = F3(0.0f) + Div(v * Dot(currentWorldPos - F3(0.0f), currentNormal), Dot(currentNormal, v));
                    n10 = Xyz(Shared(1, 0));
                }
                {
                    float3 x = GetCurrentWorldPosFromClipSpaceXY(c, (pixelUv + F2(0.0f, 1.0f) * rectSizeInv) * 2.0f - 1.0f, 1.0f);
                    float3 v = Normalize(-x);
                    x01 = F3(0.0f) + Div(v * Dot(currentWorldPos - F3(0.0f), currentNormal), Dot(currentNormal, v));
                    n01 = Xyz(Shared(0, 1));
                }

                float2 w = Abs(deltaUv) + 1.0f / 256.0f;
                w = Div(w, w.x + w.y);
                float3 x = x10 * w.x + x01 * w.y;
                float3 n = Normalize(n10 * w.x + n01 * w.y);

                float deltaUvLenFixed = smbParallaxInPixelsMin;
                deltaUvLenFixed *= 1.0f;
                deltaUvLenFixed *= 1.0f + c.shared.gFramerateScale * Bayer4x4((uint32_t)px, (uint32_t)py, c.shared.gFrameIndex);

                float2 motionUvHigh = pixelUv + deltaUv * deltaUvLenFixed * rectSizeInv;
                motionUvHigh = (Floor(motionUvHigh * rectSize) + 0.5f) * rectSizeInv;
                if (deltaUvLenFixed > 1.0f && IsInScreenNearest(motionUvHigh) != 0.0f) {
                    float2 uvScaled = RelaxClampUvToViewport(c, motionUvHigh) + ToF2(c.shared.gRectOffset);
                    int2 q = NearestTexel(P.viewZ, uvScaled);
                    float zHigh = RelaxUnpackViewZ(c, LoadR32F(P.viewZ, q.x, q.y));
                    float3 xHigh = GetCurrentWorldPosFromClipSpaceXY(c, motionUvHigh * 2.0f - 1.0f, zHigh);
                    float3 nHigh = Xyz(LoadDecodedNormalRoughness(P.decodedNR, q.x, q.y));
                    float zError = Abs(zHigh - currentLinearZ) * Rcp(Max(zHigh, currentLinearZ));
                    bool cmp = zError < NRD_CURVATURE_Z_THRESHOLD;
                    n = Select(cmp, nHigh, n);
                    x = Select(cmp, xHigh, x);
                }

                float3 edge = x - currentWorldPos;
                float edgeLenSq = LengthSquared(edge);
                curvature = Dot(n - currentNormal, edge) * PositiveRcp(edgeLenSq);
            }

            hitDistFocused = ApplyThinLensEquation(hitDist, curvature);

            const float3 virtualViewVector = Normalize(currentViewVector) * hitDistFocused;
            const float3 prevVirtualWorldPos = prevWorldPos + virtualViewVector;

            prevUVVMB = ScreenUvNoKill(c.shared.gWorldToClipPrev, prevVirtualWorldPos);
            prevUVVMB = Select(currentMaterialID == c.shared.gCameraAttachedReflectionMaterialID, prevUVSMB, prevUVVMB);

            vmbPixelPosFloat = prevUVVMB * rectSizePrev;
            vmbOriginF = Floor(vmbPixelPosFloat - 0.5f);
            vbx = (int)vmbOriginF.x, vby = (int)vmbOriginF.y;
            vmbBilinearWeights = F2(Frac(vmbPixelPosFloat.x - 0.5f), Frac(vmbPixelPosFloat.y - 0.5f));
            quadInterior = FootprintIsInterior(P.prevViewZ, vbx, vby, 2, 2);
            if (quadInterior) {
                const float2 r0 = LoadRowR32Fx2(P.prevViewZ, vbx, vby), r1 = LoadRowR32Fx2(P.prevViewZ, vbx, vby + 1);
                zQuad[0][0] = r0.x, zQuad[0][1] = r0.y, zQuad[1][0] = r1.x, zQuad[1][1] = r1.y;
            }
        }

        auto Valid = [&](int dx, int dy, float threshold) {
            float z = RelaxUnpackViewZ(c, MODE == 1 ? s_WinZ[Win(bx + dx, by + dy)] : footprintInterior ? zRows[dy + 1][dx + 1] : FetchClampedR32F(P.prevViewZ, bx + dx, by + dy));
            float v = Step(Abs(z - prevViewPosZ), threshold);
            if (!compareMaterials)
                return v;
            float m = (MODE == 1 ? NRD_DIV_255(float((s_WinMisc[Win(bx + dx, by + dy)] >> 8) & 0xFFu)) : FetchClampedR8Unorm(P.prevMaterialID, bx + dx, by + dy)) * 255.0f;
            return v * Cmp(CompareMaterials(currentMaterialID, m, minMaterialID));
        };
        const float3 tapsValid0 = F3(Valid(0, -1, smbDisocclusionThreshold.x), Valid(-1, 0, smbDisocclusionThreshold.x), Valid(0, 0, smbDisocclusionThreshold.x));
        const float3 tapsValid1 = F3(Valid(1, -1, smbDisocclusionThreshold.y), Valid(1, 0, smbDisocclusionThreshold.y), Valid(2, 0, smbDisocclusionThreshold.y));
        const float3 tapsValid2 = F3(Valid(-1, 1, smbDisocclusionThreshold.z), Valid(0, 1, smbDisocclusionThreshold.z), Valid(0, 2, smbDisocclusionThreshold.z));
        const float3 tapsValid3 = F3(Valid(1, 1, smbDisocclusionThreshold.w), Valid(2, 1, smbDisocclusionThreshold.w), Valid(1, 2, smbDisocclusionThreshold.w));

        const float3 tapsSum = tapsValid0 + tapsValid1 + tapsValid2 + tapsValid3;
        float bicubicFootprintValid = (tapsSum.x + tapsSum.y + tapsSum.z) > 11.5f ? 1.0f : 0.0f;
        float4 bilinearTapsValid = F4(tapsValid0.z, tapsValid1.y, tapsValid2.y, tapsValid3.x);

        float3 prevNormalFlat;
        if (MODE == 1) { // SampleLinearRGBA8Unorm on window texels (same taps, same weights, same order)
            const LinearTaps t = MakeLinearTaps(F2(float(bx) + 1.0f, float(by) + 1.0f));
            const float4 s00 = DecodeRGBA8Unorm(s_WinN[Win(t.x0, t.y0)]), s10 = DecodeRGBA8Unorm(s_WinN[Win(t.x0 + 1, t.y0)]), s01 = DecodeRGBA8Unorm(s_WinN[Win(t.x0, t.y0 + 1)]),
                         s11 = DecodeRGBA8Unorm(s_WinN[Win(t.x0 + 1, t.y0 + 1)]);
            prevNormalFlat = Xyz(UnpackPrevNormalRoughness(s00 * t.w00 + s10 * t.w10 + s01 * t.w01 + s11 * t.w11));
        } else {
            prevNormalFlat = Xyz(UnpackPrevNormalRoughness(SampleLinearRGBA8Unorm(P.prevNormalRoughness, F2(float(bx) + 1.0f, float(by) + 1.0f))));
        }
        prevNormalFlat = RotateVector(c.shared.gWorldPrevToWorld, prevNormalFlat);
        if (Dot(smbNormal, prevNormalFlat) < 0.0f) {
            bilinearTapsValid = F4(0.0f);
            bicubicFootprintValid = 0.0f;
        }

        Bilinear bilinear;
        bilinear.origin = originF;
        bilinear.weights = bilinearWeights;
        const float4 bilinearCustomWeights = GetBilinearCustomWeights(bilinear, bilinearTapsValid);
        const bool useBicubic = bicubicFootprintValid > 0.0f;

        const HistoryFilter hf = MakeHistoryFilter(prevPixelPosFloat, bilinearCustomWeights, useBicubic, SPEC ? P.spec.prev : P.diff.prev);
        auto History = [&](const Plane& tex, const uint2* win) { // the 12-texel fetch: from the window (MODE 1) or from memory
            if (MODE != 1)
                return FetchHistoryRGBA16F(hf, tex);
            HistoryTexelsRGBA16F t;
            WindowHistoryTexels(hf, win, wx0, wy0, WIN_W, t);
            return FetchHistoryRGBA16F(hf, tex, t);
        };
        if (DIFF) {
            prevDiffuseIllumAnd2ndMomentSMB = Max0(History(P.diff.prev, s_WinDiffPrev));
            prevDiffuseResponsiveSMB = Xyz(Max0(History(P.diff.fast, s_WinDiffFast)));
        }
        if (SPEC) {
            prevSpecularIllumAnd2ndMomentSMB = Max0(History(P.spec.prev, s_WinSpecPrev));
            prevSpecularResponsiveSMB = Xyz(Max0(History(P.spec.fast, s_WinSpecFast)));
        }
        if (SH) {
            if (DIFF) {
                prevDiffuseSH = BilinearWithCustomWeightsRGBA16F(P.diff.prevSh, bx, by, bilinearCustomWeights);
                prevDiffuseResponsiveSH = BilinearWithCustomWeightsRGBA16F(P.diff.fastSh, bx, by, bilinearCustomWeights);
            }
            if (SPEC) {
                prevSpecularSMBSH = BilinearWithCustomWeightsRGBA16F(P.spec.prevSh, bx, by, bilinearCustomWeights);
                prevSpecularSMBResponsiveSH = BilinearWithCustomWeightsRGBA16F(P.spec.fastSh, bx, by, bilinearCustomWeights);
            }
        }

        if (MODE == 1) {
            const uint32_t m00 = s_WinMisc[Win(bx, by)], m10 = s_WinMisc[Win(bx + 1, by)], m01 = s_WinMisc[Win(bx, by + 1)], m11 = s_WinMisc[Win(bx + 1, by + 1)];
            historyLength = 255.0f * BilinearWithCustomWeightsImmediateFloat(NRD_DIV_255(float(m00 & 0xFFu)), NRD_DIV_255(float(m10 & 0xFFu)), NRD_DIV_255(float(m01 & 0xFFu)),
                                         NRD_DIV_255(float(m11 & 0xFFu)), bilinearCustomWeights);
            if (SPEC)
                prevReflectionHitTSMB = BilinearWithCustomWeightsImmediateFloat(HalfBitsToFloat((uint16_t)(m00 >> 16)), HalfBitsToFloat((uint16_t)(m10 >> 16)), HalfBitsToFloat((uint16_t)(m01 >> 16)),
                    HalfBitsToFloat((uint16_t)(m11 >> 16)), bilinearCustomWeights);
        } else {
            historyLength = 255.0f * BilinearWithCustomWeightsImmediateFloat(FetchClampedR8Unorm(P.prevHistoryLength, bx, by), FetchClampedR8Unorm(P.prevHistoryLength, bx + 1, by),
                                         FetchClampedR8Unorm(P.prevHistoryLength, bx, by + 1), FetchClampedR8Unorm(P.prevHistoryLength, bx + 1, by + 1), bilinearCustomWeights);
            if (SPEC)
                prevReflectionHitTSMB = BilinearWithCustomWeightsImmediateFloat(FetchClampedR16F(P.prevSpecHitDist, bx, by), FetchClampedR16F(P.prevSpecHitDist, bx + 1, by),
                    FetchClampedR16F(P.prevSpecHitDist, bx, by + 1), FetchClampedR16F(P.prevSpecHitDist, bx + 1, by + 1), bilinearCustomWeights);
        }
        if (SPEC)
            prevReflectionHitTSMB = Max(0.001f, prevReflectionHitTSMB);

        SMBReprojectionFound = bicubicFootprintValid > 0.0f ? 2.0f : 1.0f;
        footprintQuality = bicubicFootprintValid > 0.0f ? 1.0f : Sum(bilinearCustomWeights);
        const bool anyValid = bilinearTapsValid.x != 0.0f || bilinearTapsValid.y != 0.0f || bilinearTapsValid.z != 0.0f || bilinearTapsValid.w != 0.0f;
        if (!anyValid) {
            SMBReprojectionFound = 0.0f;
            footprintQuality = 0.0f;
        }
    }

    NRD_CONSTANTS_PHASE();
    historyLength = historyLength + 1.0f;
    historyLength = Min(RELAX_MAX_ACCUM_FRAME_NUM, historyLength);

    // avoid footprint stretching due to the changed viewing angle
    const float3 Vprev = -Normalize(prevWorldPos - cameraDelta);
    const float NoVprev = Abs(Dot(currentNormal, Vprev));
    float sizeQuality = Div(NoVprev + 1e-3f, NoV + 1e-3f);
    sizeQuality *= sizeQuality;
    sizeQuality *= sizeQuality;
    footprintQuality *= Lerp(0.1f, 1.0f, Sat(sizeQuality + Abs(c.shared.gOrthoMode)));

    if (footprintQuality < 1.0f) {
        historyLength *= Sqrt(footprintQuality);
        historyLength = Max(historyLength, 1.0f);
    }
    historyLength = c.shared.gResetHistory != 0 ? 1.0f : historyLength;

    const float maxAccumulatedFrameNum = 1.0f + ((DIFF && SPEC) ? Max(c.shared.gDiffMaxAccumulatedFrameNum, c.shared.gSpecMaxAccumulatedFrameNum)
                                                                 : (DIFF ? c.shared.gDiffMaxAccumulatedFrameNum : c.shared.gSpecMaxAccumulatedFrameNum));
    historyLength = Min(historyLength, maxAccumulatedFrameNum);

    if (DIFF) {
        float diffMaxAccumulatedFrameNum = c.shared.gDiffMaxAccumulatedFrameNum;
        float diffMaxFastAccumulatedFrameNum = c.shared.gDiffMaxFastAccumulatedFrameNum;
        if (c.shared.gHasHistoryConfidence) {
            float inDiffConfidence = LoadR8Unorm(P.diff.confidence, px, py);
            diffMaxAccumulatedFrameNum *= inDiffConfidence;
            diffMaxFastAccumulatedFrameNum *= inDiffConfidence;
        }
        const float diffHistoryLength = historyLength;
        float diffuseAlpha = SMBReprojectionFound > 0.0f ? Max(Rcp(diffMaxAccumulatedFrameNum + 1.0f), Rcp(diffHistoryLength)) : 1.0f;
        float diffuseAlphaResponsive = SMBReprojectionFound > 0.0f ? Max(Rcp(diffMaxFastAccumulatedFrameNum + 1.0f), Rcp(diffHistoryLength)) : 1.0f;
        // checkerboard: pixels without data this frame (resolved by the pre-pass) accumulate slower (reference RELAX_TemporalAccumulation.hlsli:596-606)
        const bool diffHasData = c.shared.gDiffCheckerboard == 2u || CheckerBoard((uint32_t)px, (uint32_t)py, c.shared.gFrameIndex) == c.shared.gDiffCheckerboard;
        if (!diffHasData && diffHistoryLength > 1.0f) {
            diffuseAlpha *= 1.0f - c.shared.gCheckerboardResolveAccumSpeed;
            diffuseAlphaResponsive *= 1.0f - c.shared.gCheckerboardResolveAccumSpeed;
        }

        float4 accumulated = Lerp(prevDiffuseIllumAnd2ndMomentSMB, F4(diffuseIllumination, diffuse2ndMoment), diffuseAlpha);
        float3 accumulatedResponsive = Lerp(prevDiffuseResponsiveSMB, diffuseIllumination, diffuseAlphaResponsive);
        StoreRGBA16F(P.diff.out, px, py, accumulated);
        StoreRGBA16F(P.diff.outFast, px, py, F4(accumulatedResponsive, 0.0f));
        if (SH) {
            StoreRGBA16F(P.diff.outSh, px, py, Lerp(prevDiffuseSH, diffuseSH, diffuseAlpha));
            StoreRGBA16F(P.diff.outFastSh, px, py, Lerp(prevDiffuseResponsiveSH, diffuseSH, diffuseAlphaResponsive));
        }
    }

    StoreR8Unorm(P.outHistoryLength, px, py, historyLength * (1.0f / 255.0f));

    if (SPEC) {
        float specMaxAccumulatedFrameNum = c.shared.gSpecMaxAccumulatedFrameNum;
        float specMaxFastAccumulatedFrameNum = c.shared.gSpecMaxFastAccumulatedFrameNum;
        if (c.shared.gHasHistoryConfidence) {
            float inSpecConfidence = LoadR8Unorm(P.spec.confidence, px, py);
            specMaxAccumulatedFrameNum *= inSpecConfidence;
            specMaxFastAccumulatedFrameNum *= inSpecConfidence;
        }
        const float specHistoryLength = historyLength;
        const float specHistoryFrames = Min(specMaxAccumulatedFrameNum, specHistoryLength);
        const float specHistoryResponsiveFrames = Min(specMaxFastAccumulatedFrameNum, specHistoryLength);

        // (hitDist, curvature, hitDistFocused and the virtual-motion position were computed in front of the surface-motion fetches: "early virtual-motion geometry")

        NRD_CONSTANTS_PHASE();
        // ---------------------------------------------------------------- virtual motion based history
        float4 prevSpecularIllumAnd2ndMomentVMB = F4(0.0f), prevSpecularResponsiveVMB = F4(0.0f), prevSpecularVMBSH = F4(0.0f), prevSpecularVMBResponsiveSH = F4(0.0f);
        float3 prevNormalVMB = currentNormal;
        float prevRoughnessVMB = 0.0f, prevReflectionHitTVMB = c.shared.gDenoisingRange, VMBReprojectionFound;
        {
            const float2 prevVirtualPixelPosFloat = vmbPixelPosFloat, originF = vmbOriginF, bilinearWeights = vmbBilinearWeights;
            const int bx = vbx, by = vby;
            const float3 currentWorldPosShifted = currentWorldPos - cameraDelta;

            float4 vmbDisocclusionThreshold = F4(disocclusionThreshold * currentLinearZ);
            vmbDisocclusionThreshold = vmbDisocclusionThreshold * IsInScreenBilinear(originF, rectSizePrev);
            vmbDisocclusionThreshold = vmbDisocclusionThreshold - NRD_EPS;

            const bool compareSpecMaterials = MAT && c.shared.gSpecMinMaterial < 3.0f;
            auto TapValid = [&](int dx, int dy, float threshold) {
                float z = RelaxUnpackViewZ(c, quadInterior ? zQuad[dy][dx] : FetchClampedR32F(P.prevViewZ, bx + dx, by + dy));
                float3 prevWorldPosInTap = GetPreviousWorldPosFromPixelPos(c, bx + dx, by + dy, z);
                float3 posDiff = currentWorldPosShifted - prevWorldPosInTap;
                float maxPlaneDistance = Abs(Dot(posDiff, currentNormal));
                float valid = maxPlaneDistance > threshold ? 0.0f : 1.0f;
                if (!compareSpecMaterials)
                    return valid;
                float m = FetchClampedR8Unorm(P.prevMaterialID, bx + dx, by + dy) * 255.0f;
                return valid * Cmp(CompareMaterials(currentMaterialID, m, c.shared.gSpecMinMaterial));
            };
            const float4 bilinearTapsValid = F4(TapValid(0, 0, vmbDisocclusionThreshold.x), TapValid(1, 0, vmbDisocclusionThreshold.y), TapValid(0, 1, vmbDisocclusionThreshold.z),
                TapValid(1, 1, vmbDisocclusionThreshold.w));
            const bool anyValid = bilinearTapsValid.x != 0.0f || bilinearTapsValid.y != 0.0f || bilinearTapsValid.z != 0.0f || bilinearTapsValid.w != 0.0f;
            const bool allValid = bilinearTapsValid.x != 0.0f && bilinearTapsValid.y != 0.0f && bilinearTapsValid.z != 0.0f && bilinearTapsValid.w != 0.0f;

            if (anyValid) {
                Bilinear bilinear;
                bilinear.origin = originF;
                bilinear.weights = bilinearWeights;
                const float4 bilinearCustomWeights = GetBilinearCustomWeights(bilinear, bilinearTapsValid);
                const bool useBicubic = SMBReprojectionFound == 2.0f && allValid;

                const HistoryFilter hf = MakeHistoryFilter(prevVirtualPixelPosFloat, bilinearCustomWeights, useBicubic, P.spec.prev);
                prevSpecularIllumAnd2ndMomentVMB = Max0(FetchHistoryRGBA16F(hf, P.spec.prev));
                prevSpecularResponsiveVMB = Max0(FetchHistoryRGBA16F(hf, P.spec.fast));
                if (SH) {
                    prevSpecularVMBSH = BilinearWithCustomWeightsRGBA16F(P.spec.prevSh, bx, by, bilinearCustomWeights);
                    prevSpecularVMBResponsiveSH = BilinearWithCustomWeightsRGBA16F(P.spec.fastSh, bx, by, bilinearCustomWeights);
                }

                prevReflectionHitTVMB = SampleLinearR16F(P.prevSpecHitDist, prevUVVMB * resolutionScalePrev * F2(float(P.prevSpecHitDist.w), float(P.prevSpecHitDist.h)));
                prevReflectionHitTVMB = Max(0.001f, prevReflectionHitTVMB);

                float4 prevNormalRoughness = UnpackPrevNormalRoughness(
                    SampleLinearRGBA8Unorm(P.prevNormalRoughness, prevUVVMB * resolutionScalePrev * F2(float(P.prevNormalRoughness.w), float(P.prevNormalRoughness.h))));
                prevNormalVMB = RotateVector(c.shared.gWorldPrevToWorld, Xyz(prevNormalRoughness));
                prevRoughnessVMB = prevNormalRoughness.w;
            }
            VMBReprojectionFound = allValid ? 1.0f : 0.0f;
        }

        NRD_CONSTANTS_PHASE();
        // amount of virtual motion
        const float4 D = GetSpecularDominantDirection(currentNormal, V, currentRoughnessModified);
        float virtualHistoryAmount = VMBReprojectionFound * D.w;
        virtualHistoryAmount *= 1.0f;
        virtualHistoryAmount *= Cmp(Dot(prevNormalVMB, currentNormalAveraged) > 0.0f);

        float2 uvDiff = prevUVVMB - prevUVSMB;
        const float uvDiffLengthInPixels = Length(uvDiff * rectSize);

        float tanCurvature = Abs(curvature * pixelSize);
        tanCurvature *= Max(Div(uvDiffLengthInPixels, Max(NoV, 0.01f)), 1.0f);
        const float curvatureAngle = Atan(tanCurvature);

        const float lobeHalfAngle = Max(Atan(GetSpecLobeTanHalfAngleOld(currentRoughnessModified)), RELAX_NORMAL_ULP);
        const float normalWeight = GetEncodingAwareNormalWeightR(currentNormal, prevNormalVMB, lobeHalfAngle, curvatureAngle, RELAX_NORMAL_ULP, true);
        virtualHistoryAmount *= Lerp(1.0f - Sat(uvDiffLengthInPixels), 1.0f, normalWeight);

        const float2 relaxedRoughnessWeightParams = GetRelaxedRoughnessWeightParams(currentRoughness * currentRoughness, c.shared.gRoughnessFraction);
        float virtualRoughnessWeight = ComputeWeight(prevRoughnessVMB * prevRoughnessVMB, relaxedRoughnessWeightParams.x, relaxedRoughnessWeightParams.y);
        virtualRoughnessWeight = Lerp(1.0f - Sat(uvDiffLengthInPixels), 1.0f, virtualRoughnessWeight);
        virtualHistoryAmount *= virtualRoughnessWeight;
        float specVMBConfidence = virtualRoughnessWeight * 0.9f + 0.1f;

        // look back 1 and 2 frames
        uvDiff = uvDiff * Rsqrt(LengthSquared(uvDiff));
        uvDiff = Div(uvDiff, rectSizePrev);
        uvDiff = uvDiff * (Sat(uvDiffLengthInPixels * (1.0f / 0.1f)) + uvDiffLengthInPixels * 0.5f);
        const float2 backUV1 = prevUVVMB + uvDiff * 1.0f;
        const float2 backUV2 = prevUVVMB + uvDiff * 2.0f;
        const float2 prevNrSize = F2(float(P.prevNormalRoughness.w), float(P.prevNormalRoughness.h));
        const float4 backNormalRoughness1 = UnpackPrevNormalRoughness(SampleLinearRGBA8Unorm(P.prevNormalRoughness, backUV1 * resolutionScalePrev * prevNrSize));
        const float4 backNormalRoughness2 = UnpackPrevNormalRoughness(SampleLinearRGBA8Unorm(P.prevNormalRoughness, backUV2 * resolutionScalePrev * prevNrSize));
        const float3 backNormal1 = RotateVector(c.shared.gWorldPrevToWorld, Xyz(backNormalRoughness1));
        const float3 backNormal2 = RotateVector(c.shared.gWorldPrevToWorld, Xyz(backNormalRoughness2));
        float prevPrevNormalWeight = IsInScreenNearest(backUV1) != 0.0f ? GetEncodingAwareNormalWeightR(prevNormalVMB, backNormal1, lobeHalfAngle, curvatureAngle * 2.0f, RELAX_NORMAL_ULP, true) : 1.0f;
        prevPrevNormalWeight *= IsInScreenNearest(backUV2) != 0.0f ? GetEncodingAwareNormalWeightR(prevNormalVMB, backNormal2, lobeHalfAngle, curvatureAngle * 3.0f, RELAX_NORMAL_ULP, true) : 1.0f;
        virtualHistoryAmount *= 0.33f + 0.67f * prevPrevNormalWeight;
        specVMBConfidence *= 0.33f + 0.67f * prevPrevNormalWeight;
        float rw = ComputeWeight(backNormalRoughness1.w * backNormalRoughness1.w, relaxedRoughnessWeightParams.x, relaxedRoughnessWeightParams.y);
        rw *= ComputeWeight(backNormalRoughness2.w * backNormalRoughness2.w, relaxedRoughnessWeightParams.x, relaxedRoughnessWeightParams.y);
        virtualHistoryAmount *= rw * 0.9f + 0.1f;
        // multi-GPU hosts: the virtual-motion position and the look-back taps behind it, where they still COUNT -- everything fetched there enters the result through
        // virtualHistoryAmount, and the positions of rejected samples (grazing reflections project anywhere on the screen: 1 400 rows at 4K) would say nothing about a halo
        TrackHistoryReach(P.historyReach, virtualHistoryAmount != 0.0f ? Max(HistoryReachRows(prevUVVMB.y, rectSizePrev.y, py), HistoryReachRows(backUV2.y, rectSizePrev.y, py)) : 0.0f);

        NRD_CONSTANTS_PHASE();
        // hit distance confidence
        const float SMC = GetSpecMagicCurve(currentRoughnessModified);
        const float hitDistC = Lerp(specularIllumination.w, prevReflectionHitTSMB, SMC);
        const float hitDist1 = ApplyThinLensEquation(hitDistC, curvature);
        const float hitDist2 = ApplyThinLensEquation(prevReflectionHitTVMB, curvature);
        const float maxDist = Max(hitDist1, hitDist2);
        const float dHitT = Abs(hitDist1 - hitDist2);
        const float dHitTMultiplier = Lerp(20.0f, 0.0f, SMC);
        float virtualHistoryHitDistConfidence = 1.0f - Sat(Div(dHitTMultiplier * dHitT, currentLinearZ + maxDist));
        virtualHistoryHitDistConfidence = Lerp(virtualHistoryHitDistConfidence, 1.0f, SMC);

        // virtual UV discrepancy
        const float3 virtualWorldPos = GetXvirtual(hitDist, curvature, currentWorldPos, prevWorldPos, currentNormal, V, currentRoughness);
        const float virtualWorldPosLength = Length(virtualWorldPos);
        const float hitDistForTrackingPrev = prevSpecularResponsiveVMB.w;
        const float3 prevVirtualWorldPos2 = GetXvirtual(hitDistForTrackingPrev, curvature, currentWorldPos, prevWorldPos, currentNormal, V, currentRoughness);
        const float virtualWorldPosLengthPrev = Length(prevVirtualWorldPos2);
        float2 prevUVVMBTest = ScreenUvNoKill(c.shared.gWorldToClipPrev, prevVirtualWorldPos2);
        prevUVVMBTest = Select(currentMaterialID == c.shared.gCameraAttachedReflectionMaterialID, prevUVSMB, prevUVVMBTest);

        float lobeTanHalfAngle = GetSpecLobeTanHalfAngleOld(currentRoughness, 0.6f);
        lobeTanHalfAngle = Max(lobeTanHalfAngle, 0.5f * rectSizeInv.x);
        const float unproj1 = Div(Min(hitDist, hitDistForTrackingPrev), PixelRadiusToWorld(c.shared.gUnproject, c.shared.gOrthoMode, 1.0f, Max(virtualWorldPosLength, virtualWorldPosLengthPrev)));
        const float lobeRadiusInPixels = lobeTanHalfAngle * unproj1;
        const float deltaParallaxInPixels = Length((prevUVVMBTest - prevUVVMB) * rectSize);
        virtualHistoryHitDistConfidence *= SmoothStep(lobeRadiusInPixels + 0.25f, 0.0f, deltaParallaxInPixels);

        NRD_CONSTANTS_PHASE();
        // surface motion signal
        const float specSMBConfidence = (SMBReprojectionFound > 0.0f ? 1.0f : 0.0f) * GetEncodingAwareNormalWeightR(V, Vprev, Div(lobeHalfAngle * NoV, c.shared.gFramerateScale), 0.0f, 0.0f, false);
        float specSMBAlpha = 1.0f - specSMBConfidence;
        float specSMBResponsiveAlpha = 1.0f - specSMBConfidence;
        specSMBAlpha = Max(specSMBAlpha, Rcp(1.0f + specHistoryFrames));
        specSMBResponsiveAlpha = Max(specSMBAlpha, Rcp(1.0f + specHistoryResponsiveFrames));
        // checkerboard (reference RELAX_TemporalAccumulation.hlsli:853-862, :880-887)
        const bool specHasData = c.shared.gSpecCheckerboard == 2u || CheckerBoard((uint32_t)px, (uint32_t)py, c.shared.gFrameIndex) == c.shared.gSpecCheckerboard;
        if (!specHasData && smbParallaxInPixelsMax < 0.5f) {
            specSMBAlpha *= 1.0f - c.shared.gCheckerboardResolveAccumSpeed * (SMBReprojectionFound > 0.0f ? 1.0f : 0.0f);
            specSMBResponsiveAlpha *= 1.0f - c.shared.gCheckerboardResolveAccumSpeed * (SMBReprojectionFound > 0.0f ? 1.0f : 0.0f);
        }

        const float3 specRgb = Xyz(specularIllumination);
        const float3 accumulatedSpecularSMB = Lerp(Xyz(prevSpecularIllumAnd2ndMomentSMB), specRgb, specSMBAlpha);
        const float accumulatedSpecularSMBHitT = Lerp(prevReflectionHitTSMB, specularIllumination.w, Max(specSMBAlpha, 0.1f));
        const float accumulatedSpecularM2SMB = Lerp(prevSpecularIllumAnd2ndMomentSMB.w, specular2ndMoment, specSMBAlpha);
        const float3 accumulatedSpecularSMBResponsive = Lerp(prevSpecularResponsiveSMB, specRgb, specSMBResponsiveAlpha);

        // virtual motion signal
        float specVMBAlpha = 1.0f - specVMBConfidence;
        float specVMBResponsiveAlpha = 1.0f - specVMBConfidence * virtualHistoryHitDistConfidence;
        float specVMBHitTAlpha = specVMBResponsiveAlpha;
        specVMBAlpha = Max(specVMBAlpha, Rcp(1.0f + specHistoryFrames));
        specVMBResponsiveAlpha = Max(specVMBResponsiveAlpha, Rcp(1.0f + specHistoryResponsiveFrames));
        specVMBHitTAlpha = Max(specVMBHitTAlpha, Rcp(1.0f + specHistoryFrames));
        if (!specHasData && smbParallaxInPixelsMax < 0.5f) {
            const float k = 1.0f - c.shared.gCheckerboardResolveAccumSpeed * (VMBReprojectionFound > 0.0f ? 1.0f : 0.0f);
            specVMBAlpha *= k;
            specVMBResponsiveAlpha *= k;
            specVMBHitTAlpha *= k;
        }

        const float3 accumulatedSpecularVMB = Lerp(Xyz(prevSpecularIllumAnd2ndMomentVMB), specRgb, specVMBAlpha);
        const float accumulatedSpecularVMBHitT = Lerp(prevReflectionHitTVMB, specularIllumination.w, Max(specVMBHitTAlpha, 0.1f));
        const float accumulatedSpecularM2VMB = Lerp(prevSpecularIllumAnd2ndMomentVMB.w, specular2ndMoment, specVMBAlpha);
        const float3 accumulatedSpecularVMBResponsive = Lerp(Xyz(prevSpecularResponsiveVMB), specRgb, specVMBResponsiveAlpha);

        // fall back to surface motion if virtual motion doesn't go well
        virtualHistoryAmount *= Sat(Div(specVMBConfidence, specSMBConfidence + NRD_EPS));

        const float accumulatedReflectionHitT = Lerp(accumulatedSpecularSMBHitT, accumulatedSpecularVMBHitT, virtualHistoryAmount);
        const float3 accumulatedSpecularIllumination = Lerp(accumulatedSpecularSMB, accumulatedSpecularVMB, virtualHistoryAmount);
        const float3 accumulatedSpecularIlluminationResponsive = Lerp(accumulatedSpecularSMBResponsive, accumulatedSpecularVMBResponsive, virtualHistoryAmount);
        float accumulatedSpecular2ndMoment = Lerp(accumulatedSpecularM2SMB, accumulatedSpecularM2VMB, virtualHistoryAmount);

        if (SH) {
            const float4 accumulatedSpecularSMBSH = Lerp(prevSpecularSMBSH, specularSH, specSMBAlpha);
            const float4 accumulatedSpecularSMBResponsiveSH = Lerp(prevSpecularSMBResponsiveSH, specularSH, specSMBResponsiveAlpha);
            const float4 accumulatedSpecularVMBSH = Lerp(prevSpecularVMBSH, specularSH, specVMBAlpha);
            const float4 accumulatedSpecularVMBResponsiveSH = Lerp(prevSpecularVMBResponsiveSH, specularSH, specVMBResponsiveAlpha);
            const float4 accumulatedSpecularSH = Lerp(accumulatedSpecularSMBSH, accumulatedSpecularVMBSH, virtualHistoryAmount);
            const float4 accumulatedSpecularResponsiveSH = Lerp(accumulatedSpecularSMBResponsiveSH, accumulatedSpecularVMBResponsiveSH, virtualHistoryAmount);
            StoreRGBA16F(P.spec.outSh, px, py, F4(Xyz(accumulatedSpecularSH), currentRoughnessModified));
            StoreRGBA16F(P.spec.outFastSh, px, py, accumulatedSpecularResponsiveSH);
        }

        const float specularHistoryConfidence = Lerp(specSMBConfidence, specVMBConfidence, virtualHistoryAmount);
        if (accumulatedSpecular2ndMoment == 0.0f)
            accumulatedSpecular2ndMoment = c.shared.gSpecVarianceBoost * (1.0f - specularHistoryConfidence);

        StoreRGBA16F(P.spec.out, px, py, F4(accumulatedSpecularIllumination, accumulatedSpecular2ndMoment));
        StoreRGBA16F(P.spec.outFast, px, py, F4(accumulatedSpecularIlluminationResponsive, hitDist));
        StoreR16F(P.outSpecHitDist, px, py, accumulatedReflectionHitT);
        StoreR8Unorm(P.outSpecReprojectionConfidence, px, py, specularHistoryConfidence);
    }
}

// MODE 0 / 1: one workgroup per tile (XCD-aware order). MODE 2 (fallback behind the window kernel): one workgroup per FALLBACK_TILES tile columns, which walks
// them and runs the pass on the flagged ones (kernels_reblur_ta.hip has the measurement behind this shape)
constexpr int FALLBACK_TILES = 8;
template <bool DIFF, bool SPEC, bool SH, int MODE, bool MAT = true>
__global__ __launch_bounds__(256, NRD_WAVES_RELAX_TA) void RelaxTemporalAccumulationKernel(RelaxCB cArg, TaPlanes P, RowRange rows) {
    const int blockY = BlockTileY(rows, true);
    if (MODE != 2) {
        RelaxTemporalAccumulationTile<DIFF, SPEC, SH, MODE, MAT>(cArg, P, rows, BlockTileX(rows), blockY);
        return;
    }
    if (blockY >= P.tileFlags.h)
        return;
    // the FALLBACK_TILES flags of this workgroup with ONE memory latency: lane k of every wave reads flag k, the set bits are OR-ed across the wave
    // (read one after the other, the eight dependent loads were the whole cost of this kernel: 14 us per launch with nothing to do, r03_j)
    const int lane = threadIdx.x & 63, firstTile = (int)blockIdx.x * FALLBACK_TILES;
    int mask = 0;
    if (lane < FALLBACK_TILES && firstTile + lane < P.tileFlags.w)
        mask = P.tileFlags.ptr[(uint32_t)blockY * P.tileFlags.pitch + (uint32_t)(firstTile + lane)] != 0 ? 1 << lane : 0;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        mask |= __shfl_xor(mask, m);
    if (mask == 0)
        return; // the window kernel has done all these tiles (uniform; the usual case)
#pragma nounroll
    for (int k = 0; k < FALLBACK_TILES; k++) {
        if (!(mask & (1 << k)))
            continue;
        __syncthreads(); // the LDS tiles of the previous iteration are free
        RelaxTemporalAccumulationTile<DIFF, SPEC, SH, MODE>(cArg, P, rows, firstTile + k, blockY);
    }
}

template <bool DIFF, bool SPEC, bool SH>
const char* LaunchTemporalAccumulation(const PassArgs& a) {
    if (const char* e = CheckSupportedRelax(a))
        return e;
    PlaneCursor cur(a);
    TaPlanes P = {};
    P.historyReach = a.historyReachWord;
    P.tiles = cur.next();
    if (SPEC) P.spec.in = cur.next();
    if (DIFF) P.diff.in = cur.next();
    P.mv = cur.next();
    P.normalRoughness = cur.next();
    P.viewZ = cur.next();
    if (SPEC) P.spec.fast = cur.next();
    if (DIFF) P.diff.fast = cur.next();
    if (SPEC) P.spec.prev = cur.next();
    if (DIFF) P.diff.prev = cur.next();
    P.prevNormalRoughness = cur.next();
    P.prevViewZ = cur.next();
    if (SPEC) P.prevSpecHitDist = cur.next();
    P.prevHistoryLength = cur.next();
    P.prevMaterialID = cur.next();
    if (SPEC) P.spec.confidence = cur.next();
    if (DIFF) P.diff.confidence = cur.next();
    P.disocclusionThresholdMix = cur.next();
    if (SH && SPEC) P.spec.inSh = cur.next();
    if (SH && DIFF) P.diff.inSh = cur.next();
    if (SH && SPEC) P.spec.fastSh = cur.next();
    if (SH && DIFF) P.diff.fastSh = cur.next();
    if (SH && SPEC) P.spec.prevSh = cur.next();
    if (SH && DIFF) P.diff.prevSh = cur.next();
    if (SPEC) P.spec.out = cur.next();
    if (DIFF) P.diff.out = cur.next();
    if (SPEC) P.spec.outFast = cur.next();
    if (DIFF) P.diff.outFast = cur.next();
    if (SPEC) P.outSpecHitDist = cur.next();
    P.outHistoryLength = cur.next();
    if (SPEC) P.outSpecReprojectionConfidence = cur.next();
    if (SH && SPEC) P.spec.outSh = cur.next();
    if (SH && DIFF) P.diff.outSh = cur.next();
    if (SH && SPEC) P.spec.outFastSh = cur.next();
    if (SH && DIFF) P.diff.outFastSh = cur.next();
    P.decodedNR = a.decodedNormalRoughness;
    if (!cur.complete() || !P.decodedNR.ptr)
        return "RELAX TemporalAccumulation: unexpected resource count or missing decoded normal/roughness cache";
    {
        const Plane size = P.viewZ, rgba16 = SPEC ? P.spec.prev : P.diff.prev;
        bool ok = SameSize(P.decodedNR, size) && SameSize(P.mv, size) && SameSize(P.prevNormalRoughness, size) && SameSize(P.prevViewZ, size) && SameSize(P.prevSpecHitDist, size) &&
                  SameSize(P.prevHistoryLength, size) && SameSize(P.prevMaterialID, size) && SameSize(P.outSpecHitDist, size) && SameSize(P.outHistoryLength, size) &&
                  SameSize(P.outSpecReprojectionConfidence, size) && SameSize(rgba16, size);
        const SignalPlanes* sig[2] = {&P.spec, &P.diff};
        for (int s = 0; s < 2; s++)
            ok = ok && SameSize(sig[s]->in, size) && SameSize(sig[s]->inSh, size) && SameLayout(sig[s]->prev, rgba16) && SameLayout(sig[s]->prevSh, rgba16) && SameLayout(sig[s]->fast, rgba16) &&
                 SameLayout(sig[s]->fastSh, rgba16) && SameLayout(sig[s]->out, rgba16) && SameLayout(sig[s]->outSh, rgba16) && SameLayout(sig[s]->outFast, rgba16) && SameLayout(sig[s]->outFastSh, rgba16);
        if (!ok)
            return "RELAX TemporalAccumulation: planes of one frame must share their size (and the RGBA16F pool planes their pitch)";
    }
    RelaxCB c = LoadRelaxConstants(a);
    RowGrid g = GridForRows(c.shared.gRectSize.x, c.shared.gRectSize.y, TILE_X, TILE_Y, a.rowBegin, a.rowEnd);
    // The window kernel is opt-in here (NRD_HIP_RELAX_TA_WINDOW=1): measured at 4K it runs 753 us against the plain kernel's 766 us and the fallback launch behind
    // it costs more than that difference (profiles/r03_j_relax_ds_sh_kernel_stats.txt, r03_j_relax_ds_sh_{,nowindow_}bench.json) -- the plain kernel already
    // keeps three waves per SIMD and its SIMDs 72 % busy with VALU work, so the L1 bytes the window saves buy little. Kept for A/B runs and for larger frames.
    static const bool windowEnv = getenv("NRD_HIP_RELAX_TA_WINDOW") && atoi(getenv("NRD_HIP_RELAX_TA_WINDOW")) != 0;
    if (windowEnv) {
        if (!a.tileFlags.ptr || (uint32_t)a.tileFlags.w * TILE_X < (uint32_t)P.viewZ.w || (uint32_t)a.tileFlags.h * TILE_Y < (uint32_t)P.viewZ.h)
            return "RELAX TemporalAccumulation: the executor's tile-flag scratch is missing or too small";
        P.tileFlags = a.tileFlags;
        // both kernels of the pass only look at the flags of the rect's tile columns (dynamic resolution: columns beyond keep whatever an earlier, larger rect left)
        P.tileFlags.w = min(a.tileFlags.w, (int)((c.shared.gRectSize.x) + TILE_X - 1) / TILE_X);
        if (a.windowRegion)
            a.windowRegion[0] = P.tileFlags.w, a.windowRegion[1] = g.firstBlockY, a.windowRegion[2] = g.firstBlockY + (int)g.grid.y;
        static const char* limitEnv = getenv("NRD_HIP_TA_WINDOW_LIMIT"); // "WxH", test hook: a smaller box sends tiles to the fallback kernel (results do not change)
        int limW = WIN_W, limH = WIN_H;
        if (limitEnv && sscanf(limitEnv, "%dx%d", &limW, &limH) != 2)
            limW = WIN_W, limH = WIN_H;
        P.winMaxW = limW < WIN_W ? limW : WIN_W;
        P.winMaxH = limH < WIN_H ? limH : WIN_H;
        // window kernel (LDS-staged surface-motion reads), then the plain kernel on the tiles the first one declined
        LaunchPass(a, (RelaxTemporalAccumulationKernel<DIFF, SPEC, SH, 1>), g.grid, dim3(256), c, P, MakeRowRange(g));
        dim3 fallbackGrid = g.grid;
        fallbackGrid.x = (unsigned)((P.tileFlags.w + FALLBACK_TILES - 1) / FALLBACK_TILES);
        LaunchPass(a, (RelaxTemporalAccumulationKernel<DIFF, SPEC, SH, 2>), fallbackGrid, dim3(256), c, P, MakeRowRange(g));
        return nullptr;
    }
    if (c.shared.gSpecMinMaterial < 3.0f || c.shared.gDiffMinMaterial < 3.0f)
        LaunchPass(a, (RelaxTemporalAccumulationKernel<DIFF, SPEC, SH, 0>), g.grid, dim3(256), c, P, MakeRowRange(g));
    else
        LaunchPass(a, (RelaxTemporalAccumulationKernel<DIFF, SPEC, SH, 0, false>), g.grid, dim3(256), c, P, MakeRowRange(g));
    return nullptr;
}

// ================================================================================================ HistoryClamping
namespace hc {
constexpr int BORDER = 2;
constexpr int BUF_X = TILE_X + 2 * BORDER; // 36
constexpr int BUF_Y = TILE_Y + 2 * BORDER; // 12
constexpr int BUF_STRIDE = BUF_X + 1;      // 37
constexpr int BUF_SIZE = BUF_Y * BUF_STRIDE;
} // namespace hc

struct HcPlanes {
    Plane tiles, viewZ, historyLength, outHistoryLength;
    SignalPlanes spec, diff;
};

struct ClampOut {
    float4 slow, fast;
    float clampingFactor;
};

template <bool IS_SPEC>
NRD_D ClampOut ClampSignal(const RelaxCB& c, float3 fastM1, float3 fastM2, float3 noisyM1, float noisyM2, float4 fastCenterYCoCg, float4 slowIn, float3 noisyCenter, float historyLength) {
    const float maxFast = IS_SPEC ? c.shared.gSpecMaxFastAccumulatedFrameNum : c.shared.gDiffMaxFastAccumulatedFrameNum;
    const float maxSlow = IS_SPEC ? c.shared.gSpecMaxAccumulatedFrameNum : c.shared.gDiffMaxAccumulatedFrameNum;
    const bool isFixed = historyLength <= c.shared.gHistoryFixFrameNum; // history-fix pixel: responsive history is copied, no clamping

    float3 sigma = Sqrt3(Max3(F3(0.0f), fastM2 - fastM1 * fastM1));
    float3 colorMin = fastM1 - c.shared.gColorBoxSigmaScale * sigma;
    float3 colorMax = fastM1 + c.shared.gColorBoxSigmaScale * sigma;
    colorMin = Min3(colorMin, Xyz(fastCenterYCoCg));
    colorMax = Max3(colorMax, Xyz(fastCenterYCoCg));

    float3 slowYCoCg = RgbToYCoCg(Xyz(slowIn));
    float3 clampedYCoCg = slowYCoCg;
    if (maxFast < maxSlow)
        clampedYCoCg = Min3(Max3(slowYCoCg, colorMin), colorMax);
    float3 clamped = YCoCgToRgb(clampedYCoCg);

    float4 outSlow = F4(clamped, slowIn.w);
    float3 fastCenter = YCoCgToRgb(Xyz(fastCenterYCoCg));
    float4 outFast = F4(fastCenter, IS_SPEC ? fastCenterYCoCg.w : 0.0f);
    if (isFixed)
        outSlow = IS_SPEC ? outFast : F4(Xyz(outFast), outSlow.w);

    float clampingFactor = (clampedYCoCg.x - slowYCoCg.x) == 0.0f ? 0.0f : Sat(Div(clampedYCoCg.x - slowYCoCg.x, fastCenterYCoCg.x - slowYCoCg.x));
    if (isFixed)
        clampingFactor = 1.0f;

    float historyDifferenceL = (IS_SPEC ? 0.33f * RELAX_ANTILAG_ACCELERATION_AMOUNT_SCALE : RELAX_ANTILAG_ACCELERATION_AMOUNT_SCALE) * c.shared.gHistoryAccelerationAmount *
                               Luminance(Abs(fastCenter - Xyz(slowIn)));
    historyDifferenceL *= clampingFactor;
    if (isFixed)
        historyDifferenceL = 0.0f;

    float3 distanceToNoisy = noisyM1 - fastCenter;
    float distanceToNoisyL = Luminance(Abs(distanceToNoisy));
    float3 acceleration = distanceToNoisyL == 0.0f ? F3(0.0f) : Div(distanceToNoisy * historyDifferenceL, distanceToNoisyL);
    float accelerationL = Luminance(Abs(acceleration));
    float ratio = accelerationL == 0.0f ? 0.0f : Div(distanceToNoisyL, accelerationL);
    if (ratio < 1.0f)
        acceleration = acceleration * ratio;
    if (ratio <= 0.0f)
        acceleration = F3(0.0f);

    float3 slowRgb = Xyz(outSlow) + acceleration;
    float3 fastRgb = Xyz(outFast) + acceleration;

    float slowL = Luminance(Xyz(slowIn));
    float noisyL = Luminance(noisyM1);
    float temporalSigma = c.shared.gHistoryResetTemporalSigmaScale * Sqrt(Max(0.0f, noisyM2 - noisyL * noisyL));
    float spatialSigma = c.shared.gHistoryResetSpatialSigmaScale * sigma.x;
    float resetAmount = Div((IS_SPEC ? 0.5f * c.shared.gHistoryResetAmount : c.shared.gHistoryResetAmount) * Max(0.0f, Abs(slowL - noisyL) - spatialSigma - temporalSigma), 1.0e-6f + Max(slowL, noisyL) + spatialSigma + temporalSigma);
    resetAmount = Sat(resetAmount);
    slowRgb = Lerp(slowRgb, noisyCenter, resetAmount);
    fastRgb = Lerp(fastRgb, noisyCenter, resetAmount);

    float outL = Luminance(slowRgb);
    float momentCorrection = outL * outL - slowL * slowL;
    float a = Max(0.0f, outSlow.w + momentCorrection);

    ClampOut o;
    o.slow = F4(slowRgb, a);
    o.fast = F4(fastRgb, outFast.w);
    o.clampingFactor = clampingFactor;
    return o;
}

template <bool IS_SPEC, bool SH>
NRD_D void ResolveSignal(const RelaxCB& c, const SignalPlanes& S, const float4* s_Fast, const float4* s_Noisy, int px, int py, int lx, int ly, float historyLength, float4 inSignal, uint2 inSh,
    uint2 inFastSh) { // inSignal / inSh / inFastSh: the pixel's own texels of S.in / S.inSh / S.fastSh (requested by the caller in front of the tile fill)
    float3 fastM1 = F3(0.0f), fastM2 = F3(0.0f), noisyM1 = F3(0.0f);
    float noisyM2 = 0.0f, sum = 0.0f;
#pragma unroll
    for (int dx = -2; dx <= 2; dx++)
#pragma unroll
        for (int dy = -2; dy <= 2; dy++) {
            const int li = (ly + dy) * hc::BUF_STRIDE + (lx + dx);
            const float4 noisy = LdsFloat4(&s_Noisy[li]), fast = LdsFloat4(&s_Fast[li]); // whole texels, unconditionally: one ds_read_b128 each (planes.h "LDS texels")
            if (noisy.w != 0.0f) { // (kept as a branch: the select form of this body costs 36 % more instructions and 165 VGPRs)
                float3 sampleYCoCg = Xyz(fast);
                fastM1 = fastM1 + sampleYCoCg;
                fastM2 = fastM2 + sampleYCoCg * sampleYCoCg;
                float noisyLuminance = Luminance(Xyz(noisy));
                noisyM1 = noisyM1 + Xyz(noisy);
                noisyM2 += noisyLuminance * noisyLuminance;
                sum += noisy.w;
            }
        }
    fastM1 = Div(fastM1, sum);
    fastM2 = Div(fastM2, sum);
    noisyM1 = Div(noisyM1, sum);
    noisyM2 = Div(noisyM2, sum);

    const int lc = ly * hc::BUF_STRIDE + lx;
    ClampOut o = ClampSignal<IS_SPEC>(c, fastM1, fastM2, noisyM1, noisyM2, LdsFloat4(&s_Fast[lc]), inSignal, Xyz(LdsFloat4(&s_Noisy[lc])), historyLength);
    StoreRGBA16F(S.out, px, py, o.slow);
    StoreRGBA16F(S.outFast, px, py, o.fast);
    if (SH) {
        float4 sh = DecodeRGBA16F(inSh.x, inSh.y), shFast = DecodeRGBA16F(inFastSh.x, inFastSh.y);
        StoreRGBA16F(S.outSh, px, py, Lerp(sh, shFast, o.clampingFactor));
        StoreRGBA16F(S.outFastSh, px, py, shFast);
    }
}

template <bool DIFF, bool SPEC, bool SH>
__global__ __launch_bounds__(256, NRD_WAVES_RELAX_HC) void RelaxHistoryClampingKernel(HcPlanes P, RelaxCB c, RowRange rows) {
    __shared__ float4 s_SpecFast[SPEC ? hc::BUF_SIZE : 1], s_SpecNoisy[SPEC ? hc::BUF_SIZE : 1];
    __shared__ float4 s_DiffFast[DIFF ? hc::BUF_SIZE : 1], s_DiffNoisy[DIFF ? hc::BUF_SIZE : 1];

    const int blockY = BlockTileY(rows, true);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int px = BlockTileX(rows) * TILE_X + tx, py = blockY * TILE_Y + ty;
    const int rectW = c.shared.gRectSize.x, rectH = c.shared.gRectSize.y;

    if (!RelaxBlockHasGeometry(P.tiles, BlockTileX(rows), blockY))
        return;

    // The pixel's own inputs do not depend on the LDS tiles: requested in FRONT of the fill, one memory latency covers both (behind the barrier they were a second,
    // exposed one: the pass ran 34 % above the time of a build whose loads all hit the L1, profiles/r04_c_relax_ds_sh_uniform_*_kernel_stats.txt). Undecoded SH texels.
    const int qx = min(px, rectW - 1), qy = min(max(py, 0), rectH - 1);
    const float preTile = LoadR8Unorm(P.tiles, qx >> 4, qy >> 4);
    const float preHistoryLength = LoadR8Unorm(P.historyLength, qx, qy);
    float4 preSpec = F4(0.0f), preDiff = F4(0.0f);
    uint2 preSpecSh = make_uint2(0u, 0u), preSpecFastSh = make_uint2(0u, 0u), preDiffSh = make_uint2(0u, 0u), preDiffFastSh = make_uint2(0u, 0u);
    if (SPEC) {
        preSpec = LoadRGBA16F(P.spec.in, qx, qy);
        if (SH)
            preSpecSh = *TexelPtr<const uint2>(P.spec.inSh, qx, qy), preSpecFastSh = *TexelPtr<const uint2>(P.spec.fastSh, qx, qy);
    }
    if (DIFF) {
        preDiff = LoadRGBA16F(P.diff.in, qx, qy);
        if (SH)
            preDiffSh = *TexelPtr<const uint2>(P.diff.inSh, qx, qy), preDiffFastSh = *TexelPtr<const uint2>(P.diff.fastSh, qx, qy);
    }

    for (int idx = threadIdx.x; idx < hc::BUF_X * hc::BUF_Y; idx += 256) {
        int lx = idx % hc::BUF_X, ly = idx / hc::BUF_X;
        int gx = ClampI(BlockTileX(rows) * TILE_X - hc::BORDER + lx, 0, rectW - 1), gy = ClampI(blockY * TILE_Y - hc::BORDER + ly, 0, rectH - 1);
        float isValid = Cmp(LoadR32F(P.viewZ, gx, gy) < c.shared.gDenoisingRange); // raw viewZ as in the reference
        int li = ly * hc::BUF_STRIDE + lx;
        if (SPEC) {
            float4 f = LoadRGBA16F(P.spec.fast, gx, gy);
            s_SpecFast[li] = F4(RgbToYCoCg(Xyz(f)), f.w);
            s_SpecNoisy[li] = F4(Xyz(LoadRGBA16F(P.spec.noisy, gx, gy)), isValid);
        }
        if (DIFF) {
            float4 f = LoadRGBA16F(P.diff.fast, gx, gy);
            s_DiffFast[li] = F4(RgbToYCoCg(Xyz(f)), f.w);
            s_DiffNoisy[li] = F4(Xyz(LoadRGBA16F(P.diff.noisy, gx, gy)), isValid);
        }
    }
    __syncthreads();

    if (px >= rectW || py >= rectH || py < rows.rowBegin || py >= rows.rowEnd)
        return;
    if (preTile != 0.0f)
        return;
    const int lx = tx + hc::BORDER, ly = ty + hc::BORDER;
    const float centerValid = SPEC ? s_SpecNoisy[ly * hc::BUF_STRIDE + lx].w : s_DiffNoisy[ly * hc::BUF_STRIDE + lx].w;
    if (centerValid == 0.0f)
        return;

    const float historyLength = 255.0f * preHistoryLength;
    if (SPEC)
        ResolveSignal<true, SH>(c, P.spec, s_SpecFast, s_SpecNoisy, px, py, lx, ly, historyLength, preSpec, preSpecSh, preSpecFastSh);
    if (DIFF)
        ResolveSignal<false, SH>(c, P.diff, s_DiffFast, s_DiffNoisy, px, py, lx, ly, historyLength, preDiff, preDiffSh, preDiffFastSh);
    StoreR8Unorm(P.outHistoryLength, px, py, historyLength * (1.0f / 255.0f));
}

template <bool DIFF, bool SPEC, bool SH>
const char* LaunchHistoryClamping(const PassArgs& a) {
    if (const char* e = CheckSupportedRelax(a))
        return e;
    PlaneCursor cur(a);
    HcPlanes P = {};
    P.tiles = cur.next();
    P.viewZ = cur.next();
    if (SPEC) P.spec.noisy = cur.next();
    if (DIFF) P.diff.noisy = cur.next();
    if (SPEC) P.spec.in = cur.next();
    if (DIFF) P.diff.in = cur.next();
    if (SPEC) P.spec.fast = cur.next();
    if (DIFF) P.diff.fast = cur.next();
    P.historyLength = cur.next();
    if (SH && SPEC) P.spec.inSh = cur.next();
    if (SH && DIFF) P.diff.inSh = cur.next();
    if (SH && SPEC) P.spec.fastSh = cur.next();
    if (SH && DIFF) P.diff.fastSh = cur.next();
    if (SPEC) P.spec.out = cur.next();
    if (DIFF) P.diff.out = cur.next();
    if (SPEC) P.spec.outFast = cur.next();
    if (DIFF) P.diff.outFast = cur.next();
    P.outHistoryLength = cur.next();
    if (SH && SPEC) P.spec.outSh = cur.next();
    if (SH && DIFF) P.diff.outSh = cur.next();
    if (SH && SPEC) P.spec.outFastSh = cur.next();
    if (SH && DIFF) P.diff.outFastSh = cur.next();
    if (!cur.complete())
        return "RELAX HistoryClamping: unexpected resource count";
    RelaxCB c = LoadRelaxConstants(a);
    RowGrid g = GridForRows(c.shared.gRectSize.x, c.shared.gRectSize.y, TILE_X, TILE_Y, a.rowBegin, a.rowEnd);
    LaunchPass(a, (RelaxHistoryClampingKernel<DIFF, SPEC, SH>), g.grid, dim3(256), P, c, MakeRowRange(g));
    return nullptr;
}

} // namespace

#define RELAX_TA_VARIANT(name, D, S, H)                                                \
    {"RELAX_" name "_TemporalAccumulation.cs", LaunchTemporalAccumulation<D, S, H>},  \
    {"RELAX_" name "_HistoryClamping.cs", LaunchHistoryClamping<D, S, H>}

const PassEntry* GetRelaxTemporalPasses(uint32_t& num) {
    static const PassEntry k[] = {
        RELAX_TA_VARIANT("Diffuse", true, false, false),
        RELAX_TA_VARIANT("DiffuseSh", true, false, true),
        RELAX_TA_VARIANT("Specular", false, true, false),
        RELAX_TA_VARIANT("SpecularSh", false, true, true),
        RELAX_TA_VARIANT("DiffuseSpecular", true, true, false),
        RELAX_TA_VARIANT("DiffuseSpecularSh", true, true, true),
    };
    num = sizeof(k) / sizeof(k[0]);
    return k;
}

} // namespace nrdhip
