// RELAX a-trous wavelet passes as HIP kernels for gfx950.
//   AtrousSmem   reference Shaders/Include/RELAX_AtrousSmem.hlsli:10-455  (first iteration: 3x3 or 5x5 around the pixel)
//   Atrous       reference Shaders/Include/RELAX_Atrous.hlsli:10-240      (iterations 2..N: 8 taps at +-stepSize)
//
// MI355X mapping.
//   AtrousSmem is a dense stencil: 32x8-pixel workgroups stage a 36x12 halo tile (halo 2) of every per-pixel quantity the
//     window needs -- radiance+2nd moment per signal, SH1 per signal, decoded normal/roughness, world position + material
//     id -- in LDS once (up to 6 float4 planes = 41 KiB), so each plane is read from HBM ~1.7x instead of 25x. Rows are
//     padded to 37 float4 so the rows a wave touches start in different banks.
//   Atrous is the HBM-bound loop of the chain (BASELINE config 5: five iterations at 3840x2160): per iteration every
//     signal plane is read once and written once; the 8 taps of a pixel are +-step texels away, so the lanes of a wave
//     read three contiguous row segments per plane (rows y-step, y, y+step), and the working set of a band of rows
//     (3 rows x planes) stays in the 4 MiB L2 of the XCD that owns the band. Guides are tested first and the radiance
//     planes are only touched for taps whose geometric weight survives (as in the reference).
#include "relax_device.h"

namespace nrdhip {

namespace {

constexpr int TILE_X = RELAX_TILE_X;
constexpr int TILE_Y = RELAX_TILE_Y;

struct AtrousPlanes {
    Plane tiles, historyLength, specReprojectionConfidence, normalRoughness, viewZ;
    Plane outNormalRoughness, outMaterialID, outViewZ; // AtrousSmem only
    Plane decodedNR; // executor's float4 cache of normalRoughness (reblur_device.h "decoded guides")
    Plane worldPosViewZ; // executor's float4 guide plane: (world position, viewZ) per pixel (passes.h), read by the Atrous taps
    SignalPlanes spec, diff;
};

template <bool DIFF, bool SPEC, bool SH, bool SMEM>
bool BindAtrous(const PassArgs& a, AtrousPlanes& P) {
    PlaneCursor cur(a);
    P.tiles = cur.next();
    if (SPEC) P.spec.in = cur.next();
    if (DIFF) P.diff.in = cur.next();
    P.historyLength = cur.next();
    if (SPEC) P.specReprojectionConfidence = cur.next();
    P.normalRoughness = cur.next();
    P.viewZ = cur.next();
    if (SPEC) P.spec.confidence = cur.next();
    if (DIFF) P.diff.confidence = cur.next();
    if (SH && SPEC) P.spec.inSh = cur.next();
    if (SH && DIFF) P.diff.inSh = cur.next();
    if (SPEC) P.spec.out = cur.next();
    if (DIFF) P.diff.out = cur.next();
    if (SMEM) {
        P.outNormalRoughness = cur.next();
        P.outMaterialID = cur.next();
        P.outViewZ = cur.next();
    }
    if (SH && SPEC) P.spec.outSh = cur.next();
    if (SH && DIFF) P.diff.outSh = cur.next();
    P.decodedNR = a.decodedNormalRoughness;
    P.worldPosViewZ = a.worldPosViewZ;
    return cur.complete() && P.decodedNR.ptr && P.worldPosViewZ.ptr;
}

// per-pixel weight parameters shared by both flavours (confidence-driven relaxation included)
struct SpecParams {
    float centerLuminance, phiLIlluminationInv, luminanceWeightRelaxation, normalWeightParamSimplified;
    float2 normalWeightParams, roughnessWeightParams;
};
struct DiffParams {
    float centerLuminance, phiLIlluminationInv, luminanceWeightRelaxation, normalWeightParam;
};

// ================================================================================================ AtrousSmem
namespace sm {
constexpr int BORDER = 2;
constexpr int BUF_X = TILE_X + 2 * BORDER; // 36
constexpr int BUF_Y = TILE_Y + 2 * BORDER; // 12
constexpr int BUF_STRIDE = BUF_X + 1;      // 37
constexpr int BUF_SIZE = BUF_Y * BUF_STRIDE;
} // namespace sm

// MAT: material tests compiled in (the launcher picks the variant from the constants: material IDs are 0..3, a minimum material >= 3 -- the library default is 4 -- makes
// every comparison hold; a run-time test per tap, even a uniform one, is turned into compare + select by the compiler and saves nothing)
template <bool DIFF, bool SPEC, bool SH, bool MAT>
__global__ __launch_bounds__(256, NRD_WAVES_RELAX_ATROUS_SMEM) void RelaxAtrousSmemKernel(AtrousPlanes P, RelaxCB c, RowRange rows) {
    __shared__ float4 s_Spec[SPEC ? sm::BUF_SIZE : 1], s_SpecSH[(SPEC && SH) ? sm::BUF_SIZE : 1];
    __shared__ float4 s_Diff[DIFF ? sm::BUF_SIZE : 1], s_DiffSH[(DIFF && SH) ? sm::BUF_SIZE : 1];
    __shared__ float4 s_Normal_Roughness[sm::BUF_SIZE], s_WorldPos_MaterialID[sm::BUF_SIZE];

    const int blockY = BlockTileY(rows, true);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int px = BlockTileX(rows) * TILE_X + tx, py = blockY * TILE_Y + ty;
    const int rectW = c.shared.gRectSize.x, rectH = c.shared.gRectSize.y;
    const bool inRows = py >= rows.rowBegin && py < rows.rowEnd;

    // Every thread the reference launches (8x8 groups over the rect) forwards the guides to the "previous frame" planes,
    // sky or not; threads of all-sky tiles read unwritten group-shared memory there, which we define as zero.
    const bool inGrid = px < ((rectW + 7) & ~7) && py < ((rectH + 7) & ~7);
    const bool blockHasGeometry = RelaxBlockHasGeometry(P.tiles, BlockTileX(rows), blockY);

    // (measured and dropped, r04_g: requesting the pixel's own small inputs in front of the tile fill as the temporal passes do -- 0.327 against 0.310 ms)
    if (blockHasGeometry) {
        for (int idx = threadIdx.x; idx < sm::BUF_X * sm::BUF_Y; idx += 256) {
            int lx = idx % sm::BUF_X, ly = idx / sm::BUF_X;
            int gx = ClampI(BlockTileX(rows) * TILE_X - sm::BORDER + lx, 0, rectW - 1), gy = ClampI(blockY * TILE_Y - sm::BORDER + ly, 0, rectH - 1);
            int li = ly * sm::BUF_STRIDE + lx;
            if (SPEC) s_Spec[li] = LoadRGBA16F(P.spec.in, gx, gy);
            if (SPEC && SH) s_SpecSH[li] = LoadRGBA16F(P.spec.inSh, gx, gy);
            if (DIFF) s_Diff[li] = LoadRGBA16F(P.diff.in, gx, gy);
            if (DIFF && SH) s_DiffSH[li] = LoadRGBA16F(P.diff.inSh, gx, gy);
            float materialID;
            s_Normal_Roughness[li] = LoadDecodedNormalRoughness(P.decodedNR, gx, gy, materialID);
            float viewZ = RelaxUnpackViewZ(c, LoadR32F(P.viewZ, gx, gy));
            s_WorldPos_MaterialID[li] = F4(GetCurrentWorldPosFromPixelPos(c, gx, gy, viewZ), materialID);
        }
    }
    __syncthreads();

    if (!inGrid || !inRows)
        return;

    const int lx = tx + sm::BORDER, ly = ty + sm::BORDER, lc = ly * sm::BUF_STRIDE + lx;
    const bool tileIsSky = LoadR8UnormOrZero(P.tiles, px >> 4, py >> 4) != 0.0f;

    const float viewZpacked = LoadR32FOrZero(P.viewZ, px, py);
    if (InBounds(P.outViewZ, px, py))
        StoreR32F(P.outViewZ, px, py, viewZpacked);

    float4 normalRoughness = F4(0.0f);
    float4 centerWorldPosMaterialID = F4(0.0f);
    if (!tileIsSky) {
        normalRoughness = s_Normal_Roughness[lc];
        centerWorldPosMaterialID = s_WorldPos_MaterialID[lc];
    }
    const float centerViewZ = RelaxUnpackViewZ(c, viewZpacked);
    // (world position, viewZ) of every pixel for the taps of the following a-trous iterations is in P.worldPosViewZ already (per-frame guide plane)
    if (centerViewZ > c.shared.gDenoisingRange)
        normalRoughness = F4(1.0f / 255.0f);
    const float centerMaterialID = centerWorldPosMaterialID.w;
    if (InBounds(P.outNormalRoughness, px, py)) {
        StoreRGBA8Unorm(P.outNormalRoughness, px, py, PackPrevNormalRoughness(normalRoughness));
        if (NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_R10G10B10A2_UNORM) // reference RELAX_AtrousSmem.hlsli:139-141: the only encoding with material IDs
            StoreR8Unorm(P.outMaterialID, px, py, centerMaterialID * (1.0f / 255.0f));
    }

    if (tileIsSky || px >= rectW || py >= rectH)
        return;
    if (centerViewZ > c.shared.gDenoisingRange)
        return;

    const float3 centerWorldPos = Xyz(centerWorldPosMaterialID);
    const float3 centerNormal = Xyz(normalRoughness);
    const float centerRoughness = normalRoughness.w;
    const float historyLength = 255.0f * LoadR8Unorm(P.historyLength, px, py);
    constexpr bool compareSpecMaterials = MAT, compareDiffMaterials = MAT;

    if (historyLength >= c.shared.gHistoryThreshold) {
        // 3x3 gaussian-filtered variance
        float4 specularSum = F4(0.0f), diffuseSum = F4(0.0f);
#pragma unroll
        for (int dx = -1; dx <= 1; dx++)
#pragma unroll
            for (int dy = -1; dy <= 1; dy++) {
                const float k = (dx == 0 ? 0.5f : 0.25f) * (dy == 0 ? 0.5f : 0.25f); // {1/4, 1/8; 1/8, 1/16}
                const int li = (ly + dy) * sm::BUF_STRIDE + (lx + dx);
                if (SPEC) specularSum = specularSum + s_Spec[li] * k;
                if (DIFF) diffuseSum = diffuseSum + s_Diff[li] * k;
            }
        const float specular1stMomentV = Luminance(Xyz(specularSum));
        const float centerSpecularVar = Max(0.0f, specularSum.w - specular1stMomentV * specular1stMomentV);
        const float diffuse1stMomentV = Luminance(Xyz(diffuseSum));
        const float centerDiffuseVar = Max(0.0f, diffuseSum.w - diffuse1stMomentV * diffuse1stMomentV);

        float diffuseLobeAngleFraction = c.shared.gLobeAngleFraction;

        SpecParams sp = {};
        float roughnessModified = 0.0f;
        float3 centerV = F3(0.0f);
        if (SPEC) {
            sp.centerLuminance = Luminance(Xyz(s_Spec[lc]));
            sp.phiLIlluminationInv = Rcp(Max(1.0e-4f, c.shared.gSpecPhiLuminance * Sqrt(centerSpecularVar)));
            sp.roughnessWeightParams = GetRoughnessWeightParams(centerRoughness, c.shared.gRoughnessFraction);
            float diffuseLobeAngleFractionForSimplifiedSpecularNormalWeight = diffuseLobeAngleFraction;
            float specularLobeAngleFraction = c.shared.gLobeAngleFraction;
            const float specularReprojectionConfidence = LoadR8Unorm(P.specReprojectionConfidence, px, py);
            sp.luminanceWeightRelaxation = Lerp(1.0f, specularReprojectionConfidence, c.shared.gLuminanceEdgeStoppingRelaxation);
            if (c.shared.gHasHistoryConfidence) {
                float specConfidenceDrivenRelaxation = Sat(c.shared.gConfidenceDrivenRelaxationMultiplier * (1.0f - LoadR8Unorm(P.spec.confidence, px, py)));
                float r = Sat(specConfidenceDrivenRelaxation * c.shared.gConfidenceDrivenNormalEdgeStoppingRelaxation);
                diffuseLobeAngleFractionForSimplifiedSpecularNormalWeight = Lerp(diffuseLobeAngleFraction, 1.0f, r);
                specularLobeAngleFraction = Lerp(specularLobeAngleFraction, 1.0f, r);
                r = Sat(specConfidenceDrivenRelaxation * c.shared.gConfidenceDrivenLuminanceEdgeStoppingRelaxation);
                sp.luminanceWeightRelaxation *= 1.0f - r;
            }
            sp.normalWeightParamSimplified = GetNormalWeightParam2(1.0f, diffuseLobeAngleFractionForSimplifiedSpecularNormalWeight);
            sp.normalWeightParams = GetNormalWeightParams_ATrous(centerRoughness, historyLength, specularReprojectionConfidence, c.shared.gNormalEdgeStoppingRelaxation,
                specularLobeAngleFraction, c.shared.gSpecLobeAngleSlack);
            if (SH)
                roughnessModified = s_SpecSH[lc].w;
            centerV = -Normalize(centerWorldPos);
        }

        DiffParams dp = {};
        dp.luminanceWeightRelaxation = 1.0f;
        if (DIFF) {
            dp.centerLuminance = Luminance(Xyz(s_Diff[lc]));
            dp.phiLIlluminationInv = Rcp(Max(1.0e-4f, c.shared.gDiffPhiLuminance * Sqrt(centerDiffuseVar)));
            if (c.shared.gHasHistoryConfidence) {
                float diffConfidenceDrivenRelaxation = Sat(c.shared.gConfidenceDrivenRelaxationMultiplier * (1.0f - LoadR8Unorm(P.diff.confidence, px, py)));
                float r = Sat(diffConfidenceDrivenRelaxation * c.shared.gConfidenceDrivenNormalEdgeStoppingRelaxation);
                diffuseLobeAngleFraction = Lerp(diffuseLobeAngleFraction, 1.0f, r);
                r = Sat(diffConfidenceDrivenRelaxation * c.shared.gConfidenceDrivenLuminanceEdgeStoppingRelaxation);
                dp.luminanceWeightRelaxation = 1.0f - r;
            }
            dp.normalWeightParam = GetNormalWeightParam2(1.0f, diffuseLobeAngleFraction);
        }

        float sumWSpecular = 0.0f, sumWDiffuse = 0.0f;
        float4 sumSpecular = F4(0.0f), sumSpecularSH = F4(0.0f), sumDiffuse = F4(0.0f), sumDiffuseSH = F4(0.0f);
        const float depthThreshold = c.shared.gDepthThreshold * centerViewZ;

#pragma unroll
        for (int cx = -1; cx <= 1; cx++)
#pragma unroll
            for (int cy = -1; cy <= 1; cy++) {
                const int qx = px + cx, qy = py + cy;
                const bool isCenter = cx == 0 && cy == 0;
                const bool isInside = qx >= 0 && qy >= 0 && qx < rectW && qy < rectH;
                const float kernelW = isInside ? (cx == 0 ? 0.44198f : 0.27901f) * (cy == 0 ? 0.44198f : 0.27901f) : 0.0f;
                const int li = (ly + cy) * sm::BUF_STRIDE + (lx + cx);

                const float4 sampleNormalRoughness = s_Normal_Roughness[li];
                const float3 sampleNormal = Xyz(sampleNormalRoughness);
                const float sampleRoughness = sampleNormalRoughness.w;
                const float4 sampleWorldPosMaterialID = s_WorldPos_MaterialID[li];
                const float3 sampleWorldPos = Xyz(sampleWorldPosMaterialID);
                const float sampleMaterialID = sampleWorldPosMaterialID.w;

                float geometryW = GetPlaneDistanceWeight_Atrous(centerWorldPos, centerNormal, sampleWorldPos, depthThreshold);
                geometryW *= kernelW;

                if (SPEC) {
                    float angles = AcosApprox(Dot(centerNormal, sampleNormal));
                    float3 sampleV = -Normalize(sampleWorldPos + c.shared.gRoughnessEdgeStoppingRelaxation * centerWorldPos);
                    float normalWSpecularSimplified = ComputeWeight(angles, sp.normalWeightParamSimplified, 0.0f);
                    float normalWSpecular = GetSpecularNormalWeight_ATrous(sp.normalWeightParams, centerNormal, sampleNormal, centerV, sampleV);
                    float roughnessWSpecular = ComputeWeight(sampleRoughness, sp.roughnessWeightParams.x, sp.roughnessWeightParams.y);

                    float4 sampleSpecular = s_Spec[li];
                    float sampleSpecularLuminance = Luminance(Xyz(sampleSpecular));
                    float specularLuminanceW = Abs(sp.centerLuminance - sampleSpecularLuminance) * sp.phiLIlluminationInv;
                    specularLuminanceW = Min(c.shared.gSpecMaxLuminanceRelativeDifference, specularLuminanceW);
                    specularLuminanceW *= sp.luminanceWeightRelaxation;

                    float wSpecular = geometryW * ExpNegAbs(specularLuminanceW);
                    wSpecular *= c.shared.gRoughnessEdgeStoppingEnabled ? (normalWSpecular * roughnessWSpecular) : normalWSpecularSimplified;
                    wSpecular = isCenter ? kernelW : wSpecular;
                    if (compareSpecMaterials)
                        wSpecular *= Cmp(CompareMaterials(sampleMaterialID, centerMaterialID, c.shared.gSpecMinMaterial));

                    sumWSpecular += wSpecular;
                    sumSpecular = Mad(sampleSpecular, wSpecular, sumSpecular);
                    if (SH)
                        sumSpecularSH = Mad(s_SpecSH[li], wSpecular, sumSpecularSH);
                }
                if (DIFF) {
                    float angled = AcosApprox(Dot(centerNormal, sampleNormal));
                    float normalWDiffuse = ComputeWeight(angled, dp.normalWeightParam, 0.0f);

                    float4 sampleDiffuse = s_Diff[li];
                    float sampleDiffuseLuminance = Luminance(Xyz(sampleDiffuse));
                    float diffuseLuminanceW = Abs(dp.centerLuminance - sampleDiffuseLuminance) * dp.phiLIlluminationInv;
                    diffuseLuminanceW = Min(c.shared.gDiffMaxLuminanceRelativeDifference, diffuseLuminanceW);
                    diffuseLuminanceW *= dp.luminanceWeightRelaxation;

                    float wDiffuse = geometryW * normalWDiffuse * ExpNegAbs(diffuseLuminanceW);
                    wDiffuse = isCenter ? kernelW : wDiffuse;
                    if (compareDiffMaterials)
                        wDiffuse *= Cmp(CompareMaterials(sampleMaterialID, centerMaterialID, c.shared.gDiffMinMaterial));

                    sumWDiffuse += wDiffuse;
                    sumDiffuse = Mad(sampleDiffuse, wDiffuse, sumDiffuse);
                    if (SH)
                        sumDiffuseSH = Mad(s_DiffSH[li], wDiffuse, sumDiffuseSH);
                }
            }

        if (SPEC) {
            sumWSpecular = Max(sumWSpecular, 1e-6f);
            sumSpecular = Div(sumSpecular, sumWSpecular);
            float m1 = Luminance(Xyz(sumSpecular));
            float variance = Max(0.0f, sumSpecular.w - m1 * m1);
            StoreRGBA16F(P.spec.out, px, py, F4(Xyz(sumSpecular), variance));
            if (SH)
                StoreRGBA16F(P.spec.outSh, px, py, F4(Div(Xyz(sumSpecularSH), sumWSpecular), roughnessModified));
        }
        if (DIFF) {
            sumWDiffuse = Max(sumWDiffuse, 1e-6f);
            sumDiffuse = Div(sumDiffuse, sumWDiffuse);
            float m1 = Luminance(Xyz(sumDiffuse));
            float variance = Max(0.0f, sumDiffuse.w - m1 * m1);
            StoreRGBA16F(P.diff.out, px, py, F4(Xyz(sumDiffuse), variance));
            if (SH)
                StoreRGBA16F(P.diff.outSh, px, py, Div(sumDiffuseSH, sumWDiffuse));
        }
    } else {
        // spatial variance estimation over 5x5
        float sumWSpecular = 0.0f, sumSpecular1stMoment = 0.0f, sumSpecular2ndMoment = 0.0f;
        float3 sumSpecularIllumination = F3(0.0f);
        float4 sumSpecularSH = F4(0.0f);
        float sumWDiffuse = 0.0f, sumDiffuse1stMoment = 0.0f, sumDiffuse2ndMoment = 0.0f;
        float3 sumDiffuseIllumination = F3(0.0f);
        float4 sumDiffuseSH = F4(0.0f);

        const float diffuseNormalWeightParam = GetNormalWeightParam2(1.0f, c.shared.gLobeAngleFraction);

        for (int cx = -2; cx <= 2; cx++)
            for (int cy = -2; cy <= 2; cy++) {
                const int li = (ly + cy) * sm::BUF_STRIDE + (lx + cx);
                const float3 sampleNormal = Xyz(s_Normal_Roughness[li]);
                const float sampleMaterialID = s_WorldPos_MaterialID[li].w;

                const float depthW = 1.0f;
                float angle = AcosApprox(Dot(centerNormal, sampleNormal));
                float normalW = ComputeWeight(angle, diffuseNormalWeightParam, 0.0f);

                if (SPEC) {
                    float4 sampleSpecular = s_Spec[li];
                    float sample1stMoment = Luminance(Xyz(sampleSpecular));
                    float specularW = normalW * depthW;
                    if (compareSpecMaterials)
                        specularW *= Cmp(CompareMaterials(sampleMaterialID, centerMaterialID, c.shared.gSpecMinMaterial));
                    sumWSpecular += specularW;
                    sumSpecularIllumination = Mad(Xyz(sampleSpecular), specularW, sumSpecularIllumination);
                    sumSpecular1stMoment += sample1stMoment * specularW;
                    sumSpecular2ndMoment += sampleSpecular.w * specularW;
                    if (SH)
                        sumSpecularSH = Mad(s_SpecSH[li], specularW, sumSpecularSH);
                }
                if (DIFF) {
                    float4 sampleDiffuse = s_Diff[li];
                    float sample1stMoment = Luminance(Xyz(sampleDiffuse));
                    float diffuseW = normalW * depthW;
                    if (compareDiffMaterials)
                        diffuseW *= Cmp(CompareMaterials(sampleMaterialID, centerMaterialID, c.shared.gDiffMinMaterial));
                    sumWDiffuse += diffuseW;
                    sumDiffuseIllumination = Mad(Xyz(sampleDiffuse), diffuseW, sumDiffuseIllumination);
                    sumDiffuse1stMoment += sample1stMoment * diffuseW;
                    sumDiffuse2ndMoment += sampleDiffuse.w * diffuseW;
                    if (SH)
                        sumDiffuseSH = Mad(s_DiffSH[li], diffuseW, sumDiffuseSH);
                }
            }

        const float boost = Max(1.0f, Div(4.0f, historyLength + 1.0f));
        if (SPEC) {
            sumWSpecular = Max(sumWSpecular, 1e-6f);
            sumSpecularIllumination = Div(sumSpecularIllumination, sumWSpecular);
            sumSpecular1stMoment = Div(sumSpecular1stMoment, sumWSpecular);
            sumSpecular2ndMoment = Div(sumSpecular2ndMoment, sumWSpecular);
            float variance = Max(0.0f, sumSpecular2ndMoment - sumSpecular1stMoment * sumSpecular1stMoment);
            variance *= boost;
            StoreRGBA16F(P.spec.out, px, py, F4(sumSpecularIllumination, variance));
            if (SH) {
                float roughnessModified = s_SpecSH[lc].w;
                StoreRGBA16F(P.spec.outSh, px, py, F4(Div(Xyz(sumSpecularSH), sumWSpecular), roughnessModified));
            }
        }
        if (DIFF) {
            sumWDiffuse = Max(sumWDiffuse, 1e-6f);
            sumDiffuseIllumination = Div(sumDiffuseIllumination, sumWDiffuse);
            sumDiffuse1stMoment = Div(sumDiffuse1stMoment, sumWDiffuse);
            sumDiffuse2ndMoment = Div(sumDiffuse2ndMoment, sumWDiffuse);
            float variance = Max(0.0f, sumDiffuse2ndMoment - sumDiffuse1stMoment * sumDiffuse1stMoment);
            variance *= boost;
            StoreRGBA16F(P.diff.out, px, py, F4(sumDiffuseIllumination, variance));
            if (SH)
                StoreRGBA16F(P.diff.outSh, px, py, Div(sumDiffuseSH, sumWDiffuse));
        }
    }
}

template <bool DIFF, bool SPEC, bool SH>
const char* LaunchAtrousSmem(const PassArgs& a) {
    if (const char* e = CheckSupportedRelax(a))
        return e;
    AtrousPlanes P = {};
    if (!BindAtrous<DIFF, SPEC, SH, true>(a, P))
        return "RELAX AtrousSmem: unexpected resource count";
    RelaxCB c = LoadRelaxConstants(a);
    // the grid covers the reference's 8x8-group launch area (rect rounded up to 8)
    RowGrid g = GridForRows((c.shared.gRectSize.x + 7) & ~7, (c.shared.gRectSize.y + 7) & ~7, TILE_X, TILE_Y, a.rowBegin, a.rowEnd);
    if (a.rowEnd >= c.shared.gRectSize.y) // the owner of the last rows also owns the rounding rows below the rect
        g.rowEnd = (c.shared.gRectSize.y + 7) & ~7;
    if (c.shared.gSpecMinMaterial < 3.0f || c.shared.gDiffMinMaterial < 3.0f)
        LaunchPass(a, (RelaxAtrousSmemKernel<DIFF, SPEC, SH, true>), g.grid, dim3(256), P, c, MakeRowRange(g));
    else
        LaunchPass(a, (RelaxAtrousSmemKernel<DIFF, SPEC, SH, false>), g.grid, dim3(256), P, c, MakeRowRange(g));
    return nullptr;
}

// ================================================================================================ the arithmetic of one a-trous pixel
// Shared by every tap source of RelaxAtrousKernel (LDS tiles, LDS bands, global gathers) and by the marching kernel below: what is computed once per pixel
// (AtrousBegin), per tap (AtrousTap: reference RELAX_Atrous.hlsli:130-215) and at the end (AtrousEnd: `:217-240`). One source text, so one arithmetic.
template <bool DIFF, bool SPEC, bool SH>
struct AtrousPixel {
    float3 centerWorldPos, centerNormal, centerV;
    float centerRoughness, centerMaterialID, depthThreshold;
    SpecParams sp;
    DiffParams dp;
    float4 sumSpecular, sumSpecularSH, sumDiffuse, sumDiffuseSH;
    float sumWSpecular, sumWDiffuse, roughnessModified;
};

// the small per-pixel inputs, as UNORM8 bytes (the marching kernel requests them one step ahead) and decoded (= LoadR8Unorm of the byte)
struct AtrousPixelBytes {
    uint32_t historyLength, specReprojectionConfidence, specConfidence, diffConfidence;
};
struct AtrousPixelInputs {
    float historyLength, specReprojectionConfidence, specConfidence, diffConfidence; // in [0, 1]; the confidences only with gHasHistoryConfidence
};
template <bool DIFF, bool SPEC>
NRD_D AtrousPixelBytes LoadAtrousPixelBytes(const AtrousPlanes& P, const RelaxCB& c, int px, int py) {
    AtrousPixelBytes b = {};
    b.historyLength = LoadR8U(P.historyLength, px, py);
    if (SPEC)
        b.specReprojectionConfidence = LoadR8U(P.specReprojectionConfidence, px, py);
    if (c.shared.gHasHistoryConfidence) {
        if (SPEC)
            b.specConfidence = LoadR8U(P.spec.confidence, px, py);
        if (DIFF)
            b.diffConfidence = LoadR8U(P.diff.confidence, px, py);
    }
    return b;
}
NRD_D AtrousPixelInputs DecodeAtrousPixelBytes(const AtrousPixelBytes& b) {
    AtrousPixelInputs in;
    in.historyLength = NRD_DIV_255(float(b.historyLength));
    in.specReprojectionConfidence = NRD_DIV_255(float(b.specReprojectionConfidence));
    in.specConfidence = NRD_DIV_255(float(b.specConfidence));
    in.diffConfidence = NRD_DIV_255(float(b.diffConfidence));
    return in;
}

// centerSpecularSH / centerDiffuseSH: the decoded SH1 texels (read only with SH)
template <bool DIFF, bool SPEC, bool SH>
NRD_D void AtrousBegin(AtrousPixel<DIFF, SPEC, SH>& a, const AtrousPlanes& P, const RelaxCB& c, int px, int py, float4 centerWorldPosViewZ, float4 centerNormalRoughness,
    float centerMaterialID, float4 centerSpecular, float4 centerSpecularSH, float4 centerDiffuse, float4 centerDiffuseSH, const AtrousPixelInputs& in) {
    const float centerViewZ = centerWorldPosViewZ.w;
    a.centerMaterialID = centerMaterialID;
    a.centerNormal = Xyz(centerNormalRoughness);
    a.centerRoughness = centerNormalRoughness.w;
    const float centerRoughness = a.centerRoughness;
    const float historyLength = 255.0f * in.historyLength;

    float diffuseLobeAngleFraction = Div(c.shared.gLobeAngleFraction, Sqrt(float(c.gStepSize)));
    if (SH)
        diffuseLobeAngleFraction = Rcp(Sqrt(float(c.gStepSize)));
    diffuseLobeAngleFraction = Lerp(0.99f, diffuseLobeAngleFraction, Sat(historyLength * (1.0f / 5.0f)));

    SpecParams sp = {};
    sp.luminanceWeightRelaxation = 1.0f;
    float4 sumSpecular = F4(0.0f), sumSpecularSH = F4(0.0f);
    float sumWSpecular = 0.44198f * 0.44198f, roughnessModified = 0.0f;
    if (SPEC) {
        sp.centerLuminance = Luminance(Xyz(centerSpecular));
        const float centerSpecularVar = centerSpecular.w;
        sp.phiLIlluminationInv = Rcp(Max(1.0e-4f, c.shared.gSpecPhiLuminance * Sqrt(centerSpecularVar)));

        sp.roughnessWeightParams = GetRoughnessWeightParams(centerRoughness, c.shared.gRoughnessFraction);
        float diffuseLobeAngleFractionForSimplifiedSpecularNormalWeight = diffuseLobeAngleFraction;
        float specularLobeAngleFraction = c.shared.gLobeAngleFraction;
        const float specularReprojectionConfidence = in.specReprojectionConfidence;
        if (c.gStepSize <= 4)
            sp.luminanceWeightRelaxation = Lerp(1.0f, specularReprojectionConfidence, c.shared.gLuminanceEdgeStoppingRelaxation);
        if (c.shared.gHasHistoryConfidence) {
            float specConfidenceDrivenRelaxation = Sat(c.shared.gConfidenceDrivenRelaxationMultiplier * (1.0f - in.specConfidence));
            float r = Sat(specConfidenceDrivenRelaxation * c.shared.gConfidenceDrivenNormalEdgeStoppingRelaxation);
            diffuseLobeAngleFractionForSimplifiedSpecularNormalWeight = Lerp(diffuseLobeAngleFraction, 1.0f, r);
            specularLobeAngleFraction = Lerp(specularLobeAngleFraction, 1.0f, r);
            r = Sat(specConfidenceDrivenRelaxation * c.shared.gConfidenceDrivenLuminanceEdgeStoppingRelaxation);
            sp.luminanceWeightRelaxation *= 1.0f - r;
        }
        sp.normalWeightParamSimplified = GetNormalWeightParam2(1.0f, diffuseLobeAngleFractionForSimplifiedSpecularNormalWeight);
        sp.normalWeightParams = GetNormalWeightParams_ATrous(centerRoughness, historyLength, specularReprojectionConfidence, c.shared.gNormalEdgeStoppingRelaxation, specularLobeAngleFraction,
            c.shared.gSpecLobeAngleSlack);

        sumSpecular = centerSpecular * F4(sumWSpecular, sumWSpecular, sumWSpecular, sumWSpecular * sumWSpecular);
        if (SH) {
            sumSpecularSH = centerSpecularSH * sumWSpecular;
            roughnessModified = centerSpecularSH.w;
        }
    }

    DiffParams dp = {};
    dp.luminanceWeightRelaxation = 1.0f;
    float4 sumDiffuse = F4(0.0f), sumDiffuseSH = F4(0.0f);
    float sumWDiffuse = 0.44198f * 0.44198f;
    if (DIFF) {
        dp.centerLuminance = Luminance(Xyz(centerDiffuse));
        const float centerDiffuseVar = centerDiffuse.w;
        dp.phiLIlluminationInv = Rcp(Max(1.0e-4f, c.shared.gDiffPhiLuminance * Sqrt(centerDiffuseVar)));
        if (c.shared.gHasHistoryConfidence) {
            float diffConfidenceDrivenRelaxation = Sat(c.shared.gConfidenceDrivenRelaxationMultiplier * (1.0f - in.diffConfidence));
            float r = Sat(diffConfidenceDrivenRelaxation * c.shared.gConfidenceDrivenNormalEdgeStoppingRelaxation);
            diffuseLobeAngleFraction = Lerp(diffuseLobeAngleFraction, 1.0f, r);
            r = Sat(diffConfidenceDrivenRelaxation * c.shared.gConfidenceDrivenLuminanceEdgeStoppingRelaxation);
            dp.luminanceWeightRelaxation = 1.0f - r;
        }
        dp.normalWeightParam = GetNormalWeightParam2(1.0f, diffuseLobeAngleFraction);
        sumDiffuse = centerDiffuse * F4(sumWDiffuse, sumWDiffuse, sumWDiffuse, sumWDiffuse * sumWDiffuse);
        if (SH)
            sumDiffuseSH = centerDiffuseSH * sumWDiffuse;
    }

    a.centerWorldPos = Xyz(centerWorldPosViewZ);
    a.centerV = -Normalize(a.centerWorldPos);
    a.depthThreshold = c.shared.gDepthThreshold * centerViewZ;
    a.sp = sp, a.dp = dp;
    a.sumSpecular = sumSpecular, a.sumSpecularSH = sumSpecularSH, a.sumDiffuse = sumDiffuse, a.sumDiffuseSH = sumDiffuseSH;
    a.sumWSpecular = sumWSpecular, a.sumWDiffuse = sumWDiffuse, a.roughnessModified = roughnessModified;
}

// One tap. Branch-free: a tap outside the rect reads the clamped texel (a valid address) and gets weight 0 through isInside, exactly what the
// reference's "Load outside = 0" amounts to (its geometry weight is multiplied by isInside); a tap whose guide weight is <= 1e-4 keeps weight 0
// instead of being skipped by a divergent branch, so all loads of a pixel can be in flight together instead of one dependent wait per tap and signal.
// (0 * sample adds nothing: the history planes hold finite fp16 values by construction.)
// g0 = the tap's decoded (normal, roughness | material word) texel, rawSpecSh / rawDiffSh = its undecoded SH1 texels.
template <bool DIFF, bool SPEC, bool SH, bool RES, bool MAT>
NRD_D void AtrousTap(AtrousPixel<DIFF, SPEC, SH>& a, const RelaxCB& c, float kernelW, bool isInside, float4 g0, float4 sampleWorldPosViewZ, float4 sampleSpecular, float4 sampleDiffuse,
    uint2 rawSpecSh, uint2 rawDiffSh) {
    constexpr bool compareSpecMaterials = MAT, compareDiffMaterials = MAT; // compile-time variant (RelaxAtrousSmemKernel): IDs are 0..3, a minimum >= 3 disables the test
    const float3 centerWorldPos = a.centerWorldPos, centerNormal = a.centerNormal, centerV = a.centerV;
    const float centerMaterialID = a.centerMaterialID;
    float sampleMaterialID;
    const float4 sampleNormalRoughness = DecodedToNormalRoughness(g0, sampleMaterialID);
    const float3 sampleNormal = Xyz(sampleNormalRoughness);
    const float sampleRoughness = sampleNormalRoughness.w;
    const float sampleViewZ = sampleWorldPosViewZ.w;
    const float3 sampleWorldPos = Xyz(sampleWorldPosViewZ);

    float geometryW = GetPlaneDistanceWeight_Atrous(centerWorldPos, centerNormal, sampleWorldPos, a.depthThreshold);
    geometryW *= kernelW;
    geometryW *= Cmp(isInside && sampleViewZ < c.shared.gDenoisingRange);

    if (SPEC) {
        const SpecParams& sp = a.sp;
        float wSpecular;
        if (RES) { // gRoughnessEdgeStoppingEnabled != 0 (launcher)
            float3 sampleV = -Normalize(sampleWorldPos + c.shared.gRoughnessEdgeStoppingRelaxation * centerWorldPos);
            float normalWSpecular = GetSpecularNormalWeight_ATrous(sp.normalWeightParams, centerNormal, sampleNormal, centerV, sampleV);
            float roughnessWSpecular = ComputeWeight(sampleRoughness, sp.roughnessWeightParams.x, sp.roughnessWeightParams.y);
            wSpecular = geometryW * (normalWSpecular * roughnessWSpecular);
        } else {
            float angles = AcosApprox(Dot(centerNormal, sampleNormal));
            float normalWSpecularSimplified = ComputeWeight(angles, sp.normalWeightParamSimplified, 0.0f);
            wSpecular = geometryW * normalWSpecularSimplified;
        }
        if (compareSpecMaterials)
            wSpecular *= Cmp(CompareMaterials(sampleMaterialID, centerMaterialID, c.shared.gSpecMinMaterial));
        const bool on = wSpecular > 1e-4f;
        float sampleSpecularLuminance = Luminance(Xyz(sampleSpecular));
        float specularLuminanceW = Abs(sp.centerLuminance - sampleSpecularLuminance) * sp.phiLIlluminationInv;
        specularLuminanceW = Min(c.shared.gSpecMaxLuminanceRelativeDifference, specularLuminanceW);
        specularLuminanceW *= sp.luminanceWeightRelaxation;
        wSpecular *= ExpNegAbs(specularLuminanceW);
        wSpecular = on ? wSpecular : 0.0f;

        a.sumWSpecular += wSpecular;
        a.sumSpecular = Mad(sampleSpecular, F4(wSpecular, wSpecular, wSpecular, wSpecular * wSpecular), a.sumSpecular);
        if (SH)
            a.sumSpecularSH = Mad(DecodeRGBA16F(rawSpecSh.x, rawSpecSh.y), wSpecular, a.sumSpecularSH);
    }
    if (DIFF) {
        const DiffParams& dp = a.dp;
        float angled = AcosApprox(Dot(centerNormal, sampleNormal));
        float normalWDiffuse = ComputeWeight(angled, dp.normalWeightParam, 0.0f);
        float wDiffuse = geometryW * normalWDiffuse;
        if (compareDiffMaterials)
            wDiffuse *= Cmp(CompareMaterials(sampleMaterialID, centerMaterialID, c.shared.gDiffMinMaterial));
        const bool on = wDiffuse > 1e-4f;
        float sampleDiffuseLuminance = Luminance(Xyz(sampleDiffuse));
        float diffuseLuminanceW = Abs(dp.centerLuminance - sampleDiffuseLuminance) * dp.phiLIlluminationInv;
        diffuseLuminanceW = Min(c.shared.gDiffMaxLuminanceRelativeDifference, diffuseLuminanceW);
        diffuseLuminanceW *= dp.luminanceWeightRelaxation;
        wDiffuse *= ExpNegAbs(diffuseLuminanceW);
        wDiffuse = on ? wDiffuse : 0.0f;

        a.sumWDiffuse += wDiffuse;
        a.sumDiffuse = Mad(sampleDiffuse, F4(wDiffuse, wDiffuse, wDiffuse, wDiffuse * wDiffuse), a.sumDiffuse);
        if (SH)
            a.sumDiffuseSH = Mad(DecodeRGBA16F(rawDiffSh.x, rawDiffSh.y), wDiffuse, a.sumDiffuseSH);
    }
}

// the pixel's output texels, packed as StoreRGBA16F packs them (the marching kernel keeps them in registers until the next step: its stores then do not sit in front of a wait)
template <bool DIFF, bool SPEC, bool SH>
struct AtrousResult {
    uint2 spec, specSh, diff, diffSh;
};
NRD_D uint2 PackRGBA16F(float4 v) { return make_uint2(FloatsToHalf2Bits(v.x, v.y), FloatsToHalf2Bits(v.z, v.w)); } // = planes.h StoreRGBA16F
template <bool DIFF, bool SPEC, bool SH>
NRD_D AtrousResult<DIFF, SPEC, SH> AtrousFinish(const AtrousPixel<DIFF, SPEC, SH>& a, const RelaxCB& c) {
    AtrousResult<DIFF, SPEC, SH> r = {};
    if (SPEC) {
        const float sumWSpecular = a.sumWSpecular;
        float4 filtered = Div(a.sumSpecular, F4(sumWSpecular, sumWSpecular, sumWSpecular, sumWSpecular * sumWSpecular));
        if (SH) {
            if (c.gIsLastPass == 1)
                filtered = F4(LinearToYCoCg(Xyz(filtered)), filtered.w);
            r.specSh = PackRGBA16F(F4(Div(Xyz(a.sumSpecularSH), sumWSpecular), a.roughnessModified));
        }
        r.spec = PackRGBA16F(filtered);
    }
    if (DIFF) {
        const float sumWDiffuse = a.sumWDiffuse;
        float4 filtered = Div(a.sumDiffuse, F4(sumWDiffuse, sumWDiffuse, sumWDiffuse, sumWDiffuse * sumWDiffuse));
        if (SH) {
            if (c.gIsLastPass == 1)
                filtered = F4(LinearToYCoCg(Xyz(filtered)), filtered.w);
            r.diffSh = PackRGBA16F(Div(a.sumDiffuseSH, sumWDiffuse));
        }
        r.diff = PackRGBA16F(filtered);
    }
    return r;
}
// NT: the packed texels stored with the non-temporal hint -- a compile-time choice of the kernel, made by the launcher for frames above NRD_NT_STORE_PIXELS (planes.h)
template <bool NT>
NRD_D void AtrousStoreTexel(const Plane& p, int px, int py, uint2 v) {
    if constexpr (NT) {
        typedef uint32_t V2 __attribute__((ext_vector_type(2)));
        V2 raw;
        raw.x = v.x;
        raw.y = v.y;
        __builtin_nontemporal_store(raw, (V2*)TexelPtr<uint2>(p, px, py));
    } else {
        *TexelPtr<uint2>(p, px, py) = v;
    }
}
template <bool DIFF, bool SPEC, bool SH, bool NT = false>
NRD_D void AtrousStore(const AtrousResult<DIFF, SPEC, SH>& r, const AtrousPlanes& P, int px, int py) {
    if (SPEC) {
        if (SH)
            AtrousStoreTexel<NT>(P.spec.outSh, px, py, r.specSh);
        AtrousStoreTexel<NT>(P.spec.out, px, py, r.spec);
    }
    if (DIFF) {
        if (SH)
            AtrousStoreTexel<NT>(P.diff.outSh, px, py, r.diffSh);
        AtrousStoreTexel<NT>(P.diff.out, px, py, r.diff);
    }
}
template <bool DIFF, bool SPEC, bool SH, bool NT = false>
NRD_D void AtrousEnd(const AtrousPixel<DIFF, SPEC, SH>& a, const AtrousPlanes& P, const RelaxCB& c, int px, int py) {
    AtrousStore<DIFF, SPEC, SH, NT>(AtrousFinish(a, c), P, px, py);
}

// ================================================================================================ Atrous
// One iteration of the dilated 3x3 (reference RELAX_Atrous.hlsli:10-240): 8 taps at +-step texels, edge-stopping weights from the guides (plane distance,
// normal, roughness, material) and from the luminance difference, variance filtered with the squared weights.
//
// Two tap sources, one arithmetic:
//   STEP = 2 / 4  LDS tile. The 32x8 workgroup stages the (32 + 2 step) x (8 + 2 step) texels its taps can reach -- 36x12 / 40x16 -- ONCE: per texel one
//                 coalesced read of every plane (1.7 / 2.5 texels per pixel instead of 9 gathers per pixel through the L1), and everything that depends on the
//                 texel alone is computed at the fill instead of in each of the up-to-8 taps that visit it: the world position (re-derived from viewZ with the
//                 expression that wrote the guide plane, ~20 VALU) and the fp16 -> fp32 decode of the radiance planes. Taps are ds_read_b128 / b64.
//   STEP = 8      LDS bands (round 4; written for 8 and 16, used for 8). At these steps every pixel shifts its taps by a hashed offset of up to +-step/4 texels (reference RELAX_Atrous.hlsli:122-128), so the
//                 64 lanes of a wave-load pick 64 texels out of a 40 x 10 region: ~40 cache lines for 1 KB of useful data, six planes, eight taps -- the
//                 global variant moves ~6x its useful bytes from L2 to L1 and runs at twice the time of a build whose loads all hit the L1
//                 (profiles/r04_c_relax_ds_sh_uniform_*_kernel_stats.txt: 601 vs 301 us). The union of all tap positions does not fit the LDS (step 16: 72 x 48
//                 texels x 64 B = 216 KB), the three tap ROWS do: for yy = -1, 0, 1 the workgroup stages the band of (32 + 2 step + 2 r) x (8 + 2 r) texels
//                 (r = step / 4) its taps of that row can reach -- 52 x 12 / 72 x 16 texels x 64 B = 39 / 73 KB, every plane read once with coalesced row
//                 loads, the (world position, viewZ) texel from the per-frame guide plane --, then does the row's 3 (2) taps from LDS. Same texels, same
//                 arithmetic, same tap order as the global variant. Measured at 4K (profiles/r04_d_relax_ds_sh_kernel_stats.txt): step 8: 321 us against 391-405 of the
//                 gathers; step 16: 535 us -- three bands of 72 x 16 texels are 13.5 staged texels per pixel (step 8: 7.3), 221 KB per tile through the L1 and the
//                 LDS write port, at two workgroups per CU: worse than the gathers it replaces, so step 16 keeps them (NRD_HIP_ATROUS_BANDS=16 forces the bands).
//   STEP = 0      global gathers (steps 32 and beyond of 6..8 iterations; the cross-check of the two LDS variants).
// RES: RelaxSettings::enableRoughnessEdgeStopping, a compile-time variant picked by the launcher -- the taps then compute either the lobe-aware normal weight
// and the roughness weight, or the simplified normal weight, never both (the reference selects per tap between two fully evaluated expressions).
// The gathering taps (STEP = 0; steps 16 and beyond by default) fetch the 4-byte inputs the guide planes were made from instead of the 16-byte guide texels, and redo
// the decode: the taps are bound by what crosses the L1 (profiles/r02_c_gather_bench.txt: a scattered wave-load of 16 B per lane costs 149.7 CU cycles, of 4 B 41.7)
//   NRD_ATROUS_GUIDES_VIEWZ   viewZ + relax_device.h GetCurrentWorldPosFromPixelPos instead of the (world position, viewZ) texel (-6 %, r02_g_atrz_relax.json; the
//                             default since round 2 -- the define was lost in round 3's LDS-tile commit and restored in round 4)
//   NRD_ATROUS_GUIDES_RAW_NR  IN_NORMAL_ROUGHNESS + EncodeDecodedNormalRoughness instead of the decoded float4
#ifndef NRD_ATROUS_GUIDES_VIEWZ
#define NRD_ATROUS_GUIDES_VIEWZ 1
#endif
#ifndef NRD_ATROUS_GUIDES_RAW_NR
#define NRD_ATROUS_GUIDES_RAW_NR 1
#endif
#ifndef NRD_ATROUS_LDS_TILES
#define NRD_ATROUS_LDS_TILES 1 // 0: every iteration gathers from global memory (A/B and the emulation's cross-check)
#endif
template <bool DIFF, bool SPEC, bool SH, int STEP, bool RES, bool MAT, bool NT = false>
__global__ __launch_bounds__(256, NRD_WAVES_RELAX_ATROUS) void RelaxAtrousKernel(AtrousPlanes P, RelaxCB c, RowRange rows) {
    // one layout for the two guide planes and one for the (up to four) RGBA16F signal planes: verified by the launcher
    ShareLayout(P.worldPosViewZ, P.decodedNR);
    {
        const Plane sig = SPEC ? P.spec.in : P.diff.in;
        ShareLayout(P.spec.in, sig), ShareLayout(P.diff.in, sig), ShareLayout(P.spec.inSh, sig), ShareLayout(P.diff.inSh, sig);
    }
    constexpr bool TILED = STEP == 2 || STEP == 4, BANDED = STEP == 8 || STEP == 16;
    static_assert(TILED || BANDED || STEP == 0, "RelaxAtrousKernel: STEP");
    constexpr int TW = TILE_X + 2 * STEP, TH = TILE_Y + 2 * STEP, TS = TW + 1, TN = TILED ? TH * TS : 1; // row stride padded by one texel
    __shared__ float4 s_NR[TN], s_Pos[TN];
    __shared__ float4 s_Spec[SPEC ? TN : 1], s_Diff[DIFF ? TN : 1];
    __shared__ uint2 s_SpecSh[SPEC && SH ? TN : 1], s_DiffSh[DIFF && SH ? TN : 1];
    // one band of tap positions: hashed offsets reach R texels beyond the regular stencil; undecoded signal texels (8 B each)
    constexpr int R = STEP / 4, BW = TILE_X + 2 * STEP + 2 * R, BH = TILE_Y + 2 * R, BS = BW + 1, BN = BANDED ? BH * BS : 1;
    // BAND_RAW (step 16, round 5): the band holds the UNDECODED guides -- the packed normal (4 B) and viewZ (4 B) the gathering taps read, decoded per tap with the very functions
    // that wrote the guide planes -- so a 72 x 16-texel band is 46 KB instead of 74 (three workgroups per CU instead of two; the decoded bands lost to the gathers at step 16: r04_d)
#ifndef NRD_ATROUS_BAND_RAW_MIN_STEP
#define NRD_ATROUS_BAND_RAW_MIN_STEP 16
#endif
    constexpr bool BAND_RAW = STEP >= NRD_ATROUS_BAND_RAW_MIN_STEP;
    __shared__ float4 b_NR[BAND_RAW ? 1 : BN], b_Pos[BAND_RAW ? 1 : BN];
    __shared__ NrRaw br_NR[BAND_RAW ? BN : 1];
    __shared__ float br_Z[BAND_RAW ? BN : 1];
    __shared__ uint2 b_Spec[SPEC && BANDED ? BN : 1], b_Diff[DIFF && BANDED ? BN : 1], b_SpecSh[SPEC && SH && BANDED ? BN : 1], b_DiffSh[DIFF && SH && BANDED ? BN : 1];

    const int blockY = BlockTileY(rows, true);
    // lanes -> pixels of the 32x8 workgroup. The LDS variants: a wave = 2 rows of 32 pixels. The gathering variant (A/B, NRD_ATROUS_GATHER_WAVE_16x4): a wave = 16 x 4 pixels -- its
    // hashed taps then fall into ~22 cache lines of an 8-byte plane (11 rows x 23 texels) instead of ~31 (9 rows x 39 texels): fewer L1 requests per wave-load
#ifndef NRD_ATROUS_GATHER_WAVE_16x4
#define NRD_ATROUS_GATHER_WAVE_16x4 1
#endif
    constexpr bool WAVE_16x4 = NRD_ATROUS_GATHER_WAVE_16x4 && STEP == 0;
    const int tx = WAVE_16x4 ? (int)((threadIdx.x & 15u) | ((threadIdx.x >> 2) & 16u)) : (int)(threadIdx.x & 31u);
    const int ty = WAVE_16x4 ? (int)(((threadIdx.x >> 4) & 3u) | ((threadIdx.x >> 5) & 4u)) : (int)(threadIdx.x >> 5);
    const int blockX0 = BlockTileX(rows) * TILE_X, blockY0 = blockY * TILE_Y;
    const int px = blockX0 + tx, py = blockY0 + ty;
    const int rectW = c.shared.gRectSize.x, rectH = c.shared.gRectSize.y;
    if (TILED || BANDED) {
        // uniform early-outs: a workgroup beyond the rect, or over sky tiles only (TILE_X = 32 = two 16x16 tiles, TILE_Y = 8: one tile row)
        if (blockX0 >= rectW || blockY0 >= rectH)
            return;
        bool anyGeometry = false;
        for (int t = 0; t < TILE_X / 16; t++)
            if ((blockX0 >> 4) + t < P.tiles.w && (blockY0 >> 4) < P.tiles.h)
                anyGeometry |= LoadR8Unorm(P.tiles, (blockX0 >> 4) + t, blockY0 >> 4) == 0.0f;
        if (!anyGeometry)
            return;
    }
    if (TILED) {
        for (int i = threadIdx.x; i < TW * TH; i += 256) {
            const int lx = i % TW, ly = i / TW;
            const int cx = ClampI(blockX0 - STEP + lx, 0, P.worldPosViewZ.w - 1), cy = ClampI(blockY0 - STEP + ly, 0, P.worldPosViewZ.h - 1); // the taps' clamped texel
            const int li = ly * TS + lx;
            s_NR[li] = *(const float4*)(P.decodedNR.ptr + TexelOffset(P.decodedNR, cx, cy, 16u, true));
            const float z = RelaxUnpackViewZ(c, *(const float*)(P.viewZ.ptr + TexelOffset(P.viewZ, cx, cy, 4u, true)));
            s_Pos[li] = F4(GetCurrentWorldPosFromPixelPos(c, cx, cy, z), z);
            const uint32_t signalOffset = TexelOffset(SPEC ? P.spec.in : P.diff.in, cx, cy, 8u, true); // (the four signal planes share one layout: launcher check)
            if (SPEC) {
                const uint2 raw = *(const uint2*)(P.spec.in.ptr + signalOffset);
                s_Spec[li] = DecodeRGBA16F(raw.x, raw.y);
                if (SH)
                    s_SpecSh[li] = *(const uint2*)(P.spec.inSh.ptr + signalOffset);
            }
            if (DIFF) {
                const uint2 raw = *(const uint2*)(P.diff.in.ptr + signalOffset);
                s_Diff[li] = DecodeRGBA16F(raw.x, raw.y);
                if (SH)
                    s_DiffSh[li] = *(const uint2*)(P.diff.inSh.ptr + signalOffset);
            }
        }
        __syncthreads();
    }
    // BANDED: every thread of the workgroup stages the bands and meets the barriers of the tap loop; a thread without a pixel to filter computes on the
    // workgroup's first pixel (a valid address: the uniform early-out above) and stores nothing
    bool active = !(px >= rectW || py >= rectH || py < rows.rowBegin || py >= rows.rowEnd);
    if (!BANDED && !active)
        return;
    const int pxv = BANDED && !active ? blockX0 : px, pyv = BANDED && !active ? blockY0 : py;
#define px pxv
#define py pyv
    if (LoadR8Unorm(P.tiles, px >> 4, py >> 4) != 0.0f) {
        if (!BANDED)
            return;
        active = false;
    }
    const int lc = TILED ? (ty + STEP) * TS + tx + STEP : 0; // the pixel's own texel in the tile
    const float4 centerWorldPosViewZ = TILED ? s_Pos[lc] : LoadRGBA32F(P.worldPosViewZ, px, py);
    const float centerViewZ = centerWorldPosViewZ.w;
    if (centerViewZ > c.shared.gDenoisingRange) {
        if (!BANDED)
            return;
        active = false;
    }

    float centerMaterialID;
    const float4 centerNormalRoughness = TILED ? DecodedToNormalRoughness(s_NR[lc], centerMaterialID) : LoadDecodedNormalRoughness(P.decodedNR, px, py, centerMaterialID);
    const int stepSize = TILED || BANDED ? STEP : (int)c.gStepSize;
    AtrousPixel<DIFF, SPEC, SH> a;
    {
        float4 centerSpecular = F4(0.0f), centerSpecularSH = F4(0.0f), centerDiffuse = F4(0.0f), centerDiffuseSH = F4(0.0f);
        if (SPEC) {
            centerSpecular = TILED ? s_Spec[lc] : LoadRGBA16F(P.spec.in, px, py);
            if (SH) {
                if (TILED)
                    centerSpecularSH = DecodeRGBA16F(s_SpecSh[lc].x, s_SpecSh[lc].y);
                else
                    centerSpecularSH = LoadRGBA16F(P.spec.inSh, px, py);
            }
        }
        if (DIFF) {
            centerDiffuse = TILED ? s_Diff[lc] : LoadRGBA16F(P.diff.in, px, py);
            if (SH) {
                if (TILED)
                    centerDiffuseSH = DecodeRGBA16F(s_DiffSh[lc].x, s_DiffSh[lc].y);
                else
                    centerDiffuseSH = LoadRGBA16F(P.diff.inSh, px, py);
            }
        }
        AtrousBegin(a, P, c, px, py, centerWorldPosViewZ, centerNormalRoughness, centerMaterialID, centerSpecular, centerSpecularSH, centerDiffuse, centerDiffuseSH,
            DecodeAtrousPixelBytes(LoadAtrousPixelBytes<DIFF, SPEC>(P, c, px, py)));
    }

    // random offsets against ringing at large steps
    int offx = 0, offy = 0;
    if (!TILED && c.gStepSize > 4) { // (BANDED: STEP is 8 or 16, always true)
        RngHash rng;
        rng.Initialize((uint32_t)px, (uint32_t)py, c.shared.gFrameIndex);
        float2 rnd = rng.GetFloat2();
        offx = (int)(float(c.gStepSize) * 0.5f * (rnd.x - 0.5f));
        offy = (int)(float(c.gStepSize) * 0.5f * (rnd.y - 0.5f));
    }

    // The 8 taps (AtrousTap). Planes of one format share their layout (launcher check), so one texel offset serves the two guide planes and one the four signal planes.
#pragma unroll
    for (int yy = -1; yy <= 1; yy++) {
        const int bandX0 = blockX0 - STEP - R, bandY0 = blockY0 + yy * STEP - R; // texel (unclamped) of the band's LDS element (0, 0)
        if (BANDED) {
            if (yy != -1)
                __syncthreads(); // the taps of the previous band have been read
            // (measured and dropped, r04_e: requesting all of a thread's texels before the first LDS store -- 160 instead of 106 VGPRs, one wave per SIMD less, 376 instead of 321 us)
            // (measured and dropped, r04_j: undecoded 40-byte band texels decoded per tap + the NEXT band requested into registers before this band's taps -- 168 VGPRs,
            //  three waves: step 8 368 instead of 322 us, step 16 660 us against 414 for its gathers)
            for (int i = threadIdx.x; i < BW * BH; i += 256) {
                const int lx = i % BW, ly = i / BW;
                const int cx = ClampI(bandX0 + lx, 0, P.worldPosViewZ.w - 1), cy = ClampI(bandY0 + ly, 0, P.worldPosViewZ.h - 1); // the taps' clamped texel
                const int li = ly * BS + lx;
                if (BAND_RAW) {
                    br_NR[li] = *(const NrRaw*)(P.normalRoughness.ptr + TexelOffset(P.normalRoughness, cx, cy, NR_TEXEL_BYTES, true));
                    br_Z[li] = *(const float*)(P.viewZ.ptr + TexelOffset(P.viewZ, cx, cy, 4u, true));
                } else {
                    const uint32_t guideOffset = TexelOffset(P.decodedNR, cx, cy, 16u, true);
                    b_NR[li] = *(const float4*)(P.decodedNR.ptr + guideOffset);
                    b_Pos[li] = *(const float4*)(P.worldPosViewZ.ptr + guideOffset);
                }
                const uint32_t signalOffset = TexelOffset(SPEC ? P.spec.in : P.diff.in, cx, cy, 8u, true);
                if (SPEC) {
                    b_Spec[li] = *(const uint2*)(P.spec.in.ptr + signalOffset);
                    if (SH)
                        b_SpecSh[li] = *(const uint2*)(P.spec.inSh.ptr + signalOffset);
                }
                if (DIFF) {
                    b_Diff[li] = *(const uint2*)(P.diff.in.ptr + signalOffset);
                    if (SH)
                        b_DiffSh[li] = *(const uint2*)(P.diff.inSh.ptr + signalOffset);
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int xx = -1; xx <= 1; xx++) {
            if (xx == 0 && yy == 0)
                continue;
            const int qx = px + offx + xx * stepSize, qy = py + offy + yy * stepSize;
            const bool isInside = (uint32_t)qx < (uint32_t)rectW && (uint32_t)qy < (uint32_t)rectH && InBounds(P.worldPosViewZ, qx, qy);
            const float kernelW = (xx == 0 ? 0.44198f : 0.27901f) * (yy == 0 ? 0.44198f : 0.27901f);
            float4 g0, sampleWorldPosViewZ, sampleSpecular = F4(0.0f), sampleDiffuse = F4(0.0f);
            uint2 rawSpecSh = make_uint2(0u, 0u), rawDiffSh = make_uint2(0u, 0u);
            if (TILED) {
                const int li = lc + yy * STEP * TS + xx * STEP; // the tile holds the clamped texel of every tap position
                g0 = s_NR[li];
                sampleWorldPosViewZ = s_Pos[li];
                if (SPEC) {
                    sampleSpecular = s_Spec[li];
                    if (SH)
                        rawSpecSh = s_SpecSh[li];
                }
                if (DIFF) {
                    sampleDiffuse = s_Diff[li];
                    if (SH)
                        rawDiffSh = s_DiffSh[li];
                }
            } else if (BANDED) {
                const int li = (qy - bandY0) * BS + (qx - bandX0); // |offset| <= R: inside the band (which holds the clamped texel of every position)
                if (BAND_RAW) { // the decode of the gathering taps below, on the band's undecoded texel: same functions, same values
                    const int cx = ClampI(qx, 0, P.worldPosViewZ.w - 1), cy = ClampI(qy, 0, P.worldPosViewZ.h - 1);
                    g0 = EncodeDecodedNormalRoughness(br_NR[li]);
                    const float tapZ = RelaxUnpackViewZ(c, br_Z[li]);
                    sampleWorldPosViewZ = F4(GetCurrentWorldPosFromPixelPos(c, cx, cy, tapZ), tapZ);
                } else {
                    g0 = LdsFloat4(&b_NR[li]);
                    sampleWorldPosViewZ = LdsFloat4(&b_Pos[li]);
                }
                if (SPEC) {
                    const uint2 raw = b_Spec[li];
                    sampleSpecular = DecodeRGBA16F(raw.x, raw.y);
                    if (SH)
                        rawSpecSh = b_SpecSh[li];
                }
                if (DIFF) {
                    const uint2 raw = b_Diff[li];
                    sampleDiffuse = DecodeRGBA16F(raw.x, raw.y);
                    if (SH)
                        rawDiffSh = b_DiffSh[li];
                }
            } else {
                const int cx = ClampI(qx, 0, P.worldPosViewZ.w - 1), cy = ClampI(qy, 0, P.worldPosViewZ.h - 1);
                [[maybe_unused]] const uint32_t guideOffset = TexelOffset(P.decodedNR, cx, cy, 16u, true);
#if NRD_ATROUS_GUIDES_RAW_NR
                // same reasoning for the normal: 4 bytes of IN_NORMAL_ROUGHNESS through the L1 (41.7 cycles per scattered wave-load against 149.7 for the 16-byte decoded texel,
                // profiles/r02_c_gather_bench.txt) and the decode that wrote the guide plane (kernels_common.hip DecodeGuidesRelaxKernel) redone per tap: it IS the stored value
                g0 = EncodeDecodedNormalRoughness(*(const NrRaw*)(P.normalRoughness.ptr + TexelOffset(P.normalRoughness, cx, cy, NR_TEXEL_BYTES, true)));
#else
                g0 = *(const float4*)(P.decodedNR.ptr + guideOffset);
#endif
#if NRD_ATROUS_GUIDES_VIEWZ
                // 4 bytes of viewZ instead of the 16-byte (world position, viewZ) texel: the taps are bound by the bytes that cross the L1 (gather probe:
                // 39.6 cycles per 16-byte wave-load against 6.4 per 4-byte one); the position is re-derived with the very expression that wrote the guide
                // plane (DecodeGuidesRelaxKernel = relax_device.h GetCurrentWorldPosFromPixelPos), ~20 VALU, so it IS the stored value
                const float tapZ = RelaxUnpackViewZ(c, *(const float*)(P.viewZ.ptr + TexelOffset(P.viewZ, cx, cy, 4u, true)));
                sampleWorldPosViewZ = F4(GetCurrentWorldPosFromPixelPos(c, cx, cy, tapZ), tapZ);
#else
                sampleWorldPosViewZ = *(const float4*)(P.worldPosViewZ.ptr + guideOffset);
#endif
                const uint32_t signalOffset = TexelOffset(SPEC ? P.spec.in : P.diff.in, cx, cy, 8u, true); // (the four signal planes share one layout: launcher check)
                if (SPEC) {
                    const uint2 raw = *(const uint2*)(P.spec.in.ptr + signalOffset);
                    sampleSpecular = DecodeRGBA16F(raw.x, raw.y);
                    if (SH)
                        rawSpecSh = *(const uint2*)(P.spec.inSh.ptr + signalOffset);
                }
                if (DIFF) {
                    const uint2 raw = *(const uint2*)(P.diff.in.ptr + signalOffset);
                    sampleDiffuse = DecodeRGBA16F(raw.x, raw.y);
                    if (SH)
                        rawDiffSh = *(const uint2*)(P.diff.inSh.ptr + signalOffset);
                }
            }

            AtrousTap<DIFF, SPEC, SH, RES, MAT>(a, c, kernelW, isInside, g0, sampleWorldPosViewZ, sampleSpecular, sampleDiffuse, rawSpecSh, rawDiffSh);
        }
    }
    if (BANDED && !active)
        return;

    AtrousEnd<DIFF, SPEC, SH, NT>(a, P, c, px, py);
#undef px
#undef py
}

// ================================================================================================ Atrous, marching (round 6): steps 8 and 16
// At steps 8 and 16 every pixel shifts its eight taps by a hashed offset (reference RELAX_Atrous.hlsli:122-128), so the lanes of a wave read scattered texels. Round 4 staged the three tap
// rows of a 32x8 tile as LDS bands (step 8: 7.3 staged texels of 64 B per pixel -- the kernel is bound by what the FILL moves through the L1, 467 B per pixel), step 16 gathers from
// global memory (6 scattered loads per tap: 2.05x its issue floor, r05). This kernel removes the redundancy instead of pricing it: a workgroup owns a column STRIPE of 32 pixels and
// marches down it 16 rows at a time, keeping the rows its taps can reach in an LDS RING of undecoded texels (packed normal 4 B, viewZ 4 B, four signal planes 8 B each = 40 B):
//   step 16: ring of 56 rows x 72 columns = 161 280 B (one workgroup of 512 threads per CU);  step 8: 36 rows x 56 columns = 80 640 B (two per CU)
// Every row of the stripe is read from global memory ONCE per stripe (+ the 2.25x / 1.75x column halo) -- 2.6 / 2.0 staged texels per pixel instead of 7.3 (bands) or 8 scattered taps --
// by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B land contiguously in LDS, no VGPR round trip, no ds_write), and the rows of the NEXT step are requested as soon as the
// rows they replace are dead: after the yy = -1 taps of this step the top H rows of the ring are read by nothing any more (step 8: half of them, the other half after the yy = 0
// taps), so the fill of step n + 1 runs under the yy = 0 / +1 taps of step n. The taps read the ring with ds_read_b32 / b64 at their hashed positions and decode per tap with the
// functions that wrote the guide planes (the BAND_RAW / gathering variants above: same values bit for bit).
// A stripe is cut into SEGMENTS of m.segSteps steps (a workgroup = one segment of one stripe; the ring is filled completely at its first step with geometry and after every run of sky steps).
#ifndef NRD_LDS_DMA16 // (the CPU emulation of these sources under tests/emu substitutes a per-lane copy: shim/hip/hip_runtime.h)
// gsrc: this lane's 16 source bytes; ldsWaveBase: wave-uniform LDS byte address; the lane's bytes land at ldsWaveBase + lane * 16. The compiler neither counts nor waits for it (inline asm):
// NRD_LDS_DMA_WAIT() + __syncthreads() in front of the first read (cdna_hip_programming.md section 5.7: M0 written in the statement that reads it, s_nop 0 before the load).
#define NRD_LDS_DMA16(gsrc, ldsWaveBase)                                                                                                                        \
    do {                                                                                                                                                        \
        unsigned keepM0_;                                                                                                                                       \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keepM0_) : "v"(gsrc), "s"(ldsWaveBase) : "memory"); \
    } while (0)
#define NRD_LDS_DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define NRD_WAVE_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#define NRD_LDS_ADDRESS(p) ((uint32_t)(uintptr_t)(p)) // the low half of a flat address inside the shared aperture is the LDS byte address
#define NRD_LDS_POINTER(T, address) ((__attribute__((address_space(3))) T*)(uintptr_t)(address))
typedef uint32_t LdsAddress;
#endif
namespace march {
constexpr int W = 32, H = 16, THREADS = W * H;
constexpr int RoundUp(int v, int m) { return (v + m - 1) / m * m; }
constexpr int MinI(int a, int b) { return a < b ? a : b; }
template <int STEP>
struct Geo {
    static constexpr int R = STEP / 4;                               // the hashed offset lies in [-R, R - 1]: (int)(STEP / 2 * (rnd - 0.5)), rnd in [0, 1) (0 happens: 2^-24 per pixel)
    static constexpr int LEFT = RoundUp(STEP + R, 4);                // ring column 0 = stripe column - LEFT: a multiple of 4 texels, so the 16-byte pieces of the 4-byte planes stay aligned
    static constexpr int BW = RoundUp(LEFT + W + STEP + R - 1, 4);   // ring columns
    static constexpr int TOP = STEP + R;                             // ring row 0 of a step = its first pixel row - TOP
    static constexpr int SPAN = H + 2 * (STEP + R) - 1;              // rows the taps of a step can reach
    static constexpr int CH = STEP >= 16 ? 8 : 4;                    // rows per chunk (the unit of a fill: contiguous in LDS, never split by the ring's wrap-around)
    static constexpr int RR = RoundUp(SPAN, CH);                     // ring rows
    static constexpr int FREE1 = MinI(STEP, H);                      // rows (from the step's row 0) dead after the yy = -1 taps: neither the other taps of this step nor yy = -1 of the next read them
    static constexpr int FREE2 = MinI(2 * STEP, H) - FREE1;          // ... more after the yy = 0 taps
    static_assert(FREE1 + FREE2 == H && FREE1 % CH == 0 && FREE2 % CH == 0 && H % CH == 0, "march::Geo: the prefetch points must free the H rows of the next step in whole chunks");
    static_assert(LEFT >= STEP + R && BW % 4 == 0 && RR >= SPAN, "march::Geo");
};
struct Range { // kernel argument
    int firstY;   // pixel row of step 0 (a multiple of H)
    int numSteps, segSteps, numSegs;
    int numStripes, stripesPerXcd;
    int rowBegin, rowEnd; // rows this rank produces (multi-GPU row strips); the whole rect otherwise
};
} // namespace march

// the ring's normal plane holds 4-byte texels: the marching kernel serves the 4-byte encodings (nrdmath.h NRD_NORMAL_ENCODING 0..2) and is never launched for the others (AtrousMarchMaxStep)
NRD_D float4 EncodeDecodedRingWord(uint32_t w) {
#if NRD_NORMAL_ENCODING <= NRD_NORMAL_ENCODING_R10G10B10A2_UNORM
    return EncodeDecodedNormalRoughness(w);
#else
    return F4(0.0f);
#endif
}

// the ring, one array per plane (structure of arrays: what the DMA's lane-linear destination allows)
template <bool DIFF, bool SPEC, bool SH, int N>
struct MarchRing {
    uint32_t nr[N];
    float z[N];
    uint2 spec[SPEC ? N : 1], specSh[SPEC && SH ? N : 1], diff[DIFF ? N : 1], diffSh[DIFF && SH ? N : 1];
};

// The fill. A chunk (CH ring rows of one plane, contiguous in LDS) is cut into pieces of 16 bytes (4 texels of the 4-byte planes, 2 of the 8-byte ones) and 64 pieces make a JOB = one
// wave-wide global_load_lds_dwordx4; a chunk's JOBS jobs are dealt to the 8 waves round-robin -- job k of every chunk is always done by wave k % 8 as its slot k / 8 --, so what a lane
// needs of a job beyond the chunk's first row is a constant of the kernel: the row of its piece inside the chunk and the first texel column of its piece. A job then costs a clamp, a
// multiply-add and the 64-bit address. Pieces on or beyond the left / right plane border (first and last stripe only) are written texel by texel with the clamped addressing of the
// taps instead (a DMA piece is 16 contiguous source bytes); rows beyond the plane read the clamped row.
template <int STEP, bool DIFF, bool SPEC, bool SH>
struct MarchFiller {
    typedef march::Geo<STEP> G;
    static constexpr int P4 = G::CH * G::BW / 4, P8 = G::CH * G::BW / 2, J4 = (P4 + 63) / 64, J8 = (P8 + 63) / 64;
    static constexpr int NSIG = (SPEC ? (SH ? 2 : 1) : 0) + (DIFF ? (SH ? 2 : 1) : 0);
    static constexpr int JOBS = 2 * J4 + NSIG * J8, SLOTS = (JOBS + march::THREADS / 64 - 1) / (march::THREADS / 64);
    int rowInChunk[SLOTS]; // < 0: this lane has no piece in the job
    int x[SLOTS];          // first texel column of the piece
    int wave, lane;

    NRD_D void Init(int x0) {
        lane = (int)(threadIdx.x & 63u);
        wave = NRD_WAVE_UNIFORM((int)(threadIdx.x >> 6));
#pragma unroll
        for (int i = 0; i < SLOTS; i++) {
            const int k = wave + i * (march::THREADS / 64);
            const bool wide = k >= 2 * J4; // a signal plane (8-byte texels)
            const int part = wide ? (k - 2 * J4) % J8 : k % J4;
            const int perRow = wide ? G::BW / 2 : G::BW / 4, pieces = wide ? P8 : P4;
            const int piece = part * 64 + lane;
            const int r = piece / perRow, cp = piece - r * perRow;
            rowInChunk[i] = k < JOBS && piece < pieces ? r : -1;
            x[i] = x0 + cp * (wide ? 2 : 4);
        }
    }
    // rows [row0, row0 + numRows) -> their ring chunks (every thread of the workgroup calls it; numRows and row0 - segTop are multiples of CH)
    template <typename Ring>
    NRD_D void Fill(Ring& ring, const AtrousPlanes& P, int segTop, int row0, int numRows) const {
        const int w = P.viewZ.w, h = P.viewZ.h;
#pragma unroll
        for (int i = 0; i < SLOTS; i++) {
            const int k = wave + i * (march::THREADS / 64);
            if (k >= JOBS)
                continue;
            // the job's plane (uniform): packed normal, viewZ, then the signal planes that exist in the order spec, specSh, diff, diffSh
            const uint8_t* src;
            uint32_t pitch, part;
            LdsAddress dst;
            const bool wide = k >= 2 * J4;
            if (k < J4)
                src = P.normalRoughness.ptr, pitch = P.normalRoughness.pitch, dst = NRD_LDS_ADDRESS(ring.nr), part = (uint32_t)k;
            else if (k < 2 * J4)
                src = P.viewZ.ptr, pitch = P.viewZ.pitch, dst = NRD_LDS_ADDRESS(ring.z), part = (uint32_t)(k - J4);
            else {
                const int ks = k - 2 * J4, sig = ks / J8;
                const int which = SPEC ? (SH ? sig : (sig == 0 ? 0 : 2)) : (SH ? sig + 2 : 2);
                pitch = (SPEC ? P.spec.in : P.diff.in).pitch, part = (uint32_t)(ks - sig * J8);
                if (SPEC && which == 0)
                    src = P.spec.in.ptr, dst = NRD_LDS_ADDRESS(ring.spec);
                else if (SPEC && SH && which == 1)
                    src = P.spec.inSh.ptr, dst = NRD_LDS_ADDRESS(ring.specSh);
                else if (DIFF && which == 2)
                    src = P.diff.in.ptr, dst = NRD_LDS_ADDRESS(ring.diff);
                else
                    src = P.diff.inSh.ptr, dst = NRD_LDS_ADDRESS(ring.diffSh);
            }
            const uint32_t bpt = wide ? 8u : 4u, texels = wide ? 2u : 4u;
            const bool dma = x[i] >= 0 && x[i] + (int)texels <= w; // (all pieces of all but the first and last stripe)
            for (int row = row0; row < row0 + numRows; row += G::CH) {
                const LdsAddress chunkBase = dst + ((uint32_t)(row - segTop) % (uint32_t)G::RR) * ((uint32_t)G::BW * bpt); // LDS byte address of the chunk in this plane
                if (rowInChunk[i] >= 0) {
                    const int cy = ClampI(row + rowInChunk[i], 0, h - 1);
                    Plane pl = {const_cast<uint8_t*>(src), pitch, w, h};
                    if (dma)
                        NRD_LDS_DMA16(src + TexelOffset(pl, x[i], cy, bpt, true), NRD_WAVE_UNIFORM(chunkBase + part * 1024u));
                    else
                        for (uint32_t t = 0; t < texels; t++) {
                            const uint8_t* texel = src + TexelOffset(pl, ClampI(x[i] + (int)t, 0, w - 1), cy, bpt, true);
                            const LdsAddress at = chunkBase + (part * 64u + (uint32_t)lane) * 16u + t * bpt;
                            *NRD_LDS_POINTER(uint32_t, at) = *(const uint32_t*)texel;
                            if (wide)
                                *NRD_LDS_POINTER(uint32_t, at + 4u) = *(const uint32_t*)(texel + 4);
                        }
                }
            }
        }
    }
};

template <bool DIFF, bool SPEC, bool SH, int STEP, bool RES, bool MAT>
__global__ __launch_bounds__(march::THREADS, STEP == 8 ? 4 : 2) void RelaxAtrousMarchKernel(AtrousPlanes P, RelaxCB c, march::Range m) {
    using G = march::Geo<STEP>;
    constexpr int W = march::W, H = march::H;
    static_assert(STEP == 8 || STEP == 16, "RelaxAtrousMarchKernel: STEP");
    {
        const Plane sig = SPEC ? P.spec.in : P.diff.in;
        ShareLayout(P.spec.in, sig), ShareLayout(P.diff.in, sig), ShareLayout(P.spec.inSh, sig), ShareLayout(P.diff.inSh, sig);
        ShareSize(P.viewZ, sig), ShareSize(P.normalRoughness, sig); // (the launcher checked that the two guide inputs have the frame size)
    }
    __shared__ MarchRing<DIFF, SPEC, SH, G::RR * G::BW> ring;

    // workgroup -> (stripe, segment): blockIdx.x % 8 is the XCD (passes.h): an XCD owns stripesPerXcd adjacent stripes, whose column halos then meet in ONE L2;
    // segments are dealt bottom-up (the sky, at the top as usual, drains last)
    const int xcd = (int)(blockIdx.x & 7u), j = (int)(blockIdx.x >> 3);
    const int seg = m.numSegs - 1 - j / m.stripesPerXcd, stripe = xcd * m.stripesPerXcd + (j - (j / m.stripesPerXcd) * m.stripesPerXcd);
    if (stripe >= m.numStripes)
        return;
    const int rectW = c.shared.gRectSize.x, rectH = c.shared.gRectSize.y;
    const int blockX0 = stripe * W, x0 = blockX0 - G::LEFT;
    const int step0 = seg * m.segSteps, numSteps = (step0 + m.segSteps < m.numSteps ? step0 + m.segSteps : m.numSteps) - step0; // (<= 32: launcher)
    const int segY0 = m.firstY + step0 * H, segTop = segY0 - G::TOP;
    const int tx = (int)(threadIdx.x & 31u), ty = (int)(threadIdx.x >> 5);
    const int px = blockX0 + tx;
    const int lcol = tx + G::LEFT;
    MarchFiller<STEP, DIFF, SPEC, SH> filler;
    filler.Init(x0);
    auto fill = [&](int row0, int numRows) { filler.Fill(ring, P, segTop, row0, numRows); };

    // the tile flags of the segment, once: bit 2 i + t = tile t (of the two 16x16 tiles of a 32x16 step) of step i has geometry
    uint64_t geometry = 0;
    for (int i = 0; i < numSteps; i++) {
        const int tileY = (segY0 + i * H) >> 4;
#pragma unroll
        for (int t = 0; t < W / 16; t++)
            if ((blockX0 >> 4) + t < P.tiles.w && tileY < P.tiles.h && segY0 + i * H < rectH && LoadR8U(P.tiles, (blockX0 >> 4) + t, tileY) == 0u)
                geometry |= (uint64_t)1 << (2 * i + t);
    }
    if (geometry == 0)
        return;

    // a pixel of step i takes part when it lies in the rect, in this rank's rows and in a tile with geometry (the viewZ test follows when the ring holds its texel)
    auto isActive = [&](int i) {
        const int py = segY0 + i * H + ty;
        return i < numSteps && px < rectW && py < rectH && py >= m.rowBegin && py < m.rowEnd && ((geometry >> (2 * i + (tx >> 4))) & 1u) != 0u;
    };

    int ringFor = -1; // the step whose rows the ring holds or has in flight
    // software pipeline over the steps: the small per-pixel inputs of step i + 1 are requested during step i, the outputs of step i are stored during step i + 1 --
    // with one workgroup (step 16) or two (step 8) per CU whose waves meet at barriers, a load or a store in front of a wait is latency nobody hides
    int first = 0;
    while (((geometry >> (2 * first)) & 3u) == 0u)
        first++;
    // (step 8 runs two workgroups per CU, which hide each other's waits, inside 128 VGPRs: no pipeline there)
    constexpr bool PIPELINED = STEP == 16;
    AtrousPixelBytes bytes = {};
    if (PIPELINED && isActive(first))
        bytes = LoadAtrousPixelBytes<DIFF, SPEC>(P, c, px, segY0 + first * H + ty);
    AtrousResult<DIFF, SPEC, SH> pending = {};
    int pendingY = -1; // >= 0: `pending` waits to be stored at (px, pendingY)

    for (int i = first; i < numSteps; i++) {
        if (((geometry >> (2 * i)) & 3u) == 0u)
            continue;
        const int y0 = segY0 + i * H, top = y0 - G::TOP;
        int next = i + 1; // the next step with geometry (numSteps: none)
        while (next < numSteps && ((geometry >> (2 * next)) & 3u) == 0u)
            next++;
        const bool nextAdjacent = next == i + 1 && next < numSteps;
        if (ringFor != i) { // first step with geometry of the segment, or the step behind a run of sky steps: the whole ring
            __syncthreads(); // (the taps of an earlier step may still be reading the ring)
            fill(top, G::RR);
        }
        const int py = y0 + ty;
        bool active = isActive(i);
        if (!PIPELINED && active)
            bytes = LoadAtrousPixelBytes<DIFF, SPEC>(P, c, px, py);
        const AtrousPixelInputs in = DecodeAtrousPixelBytes(bytes); // (PIPELINED: requested one step ago)
        // the hashed offset of the pixel (reference RELAX_Atrous.hlsli:122-128)
        int offx, offy;
        {
            RngHash rng;
            rng.Initialize((uint32_t)px, (uint32_t)py, c.shared.gFrameIndex);
            float2 rnd = rng.GetFloat2();
            offx = (int)(float(c.gStepSize) * 0.5f * (rnd.x - 0.5f));
            offy = (int)(float(c.gStepSize) * 0.5f * (rnd.y - 0.5f));
        }
        NRD_LDS_DMA_WAIT(); // this wave's pieces have landed ...
        __syncthreads();    // ... and everybody else's
        ringFor = i;
        if (pendingY >= 0) // the previous step's outputs: their latency runs under this step's taps
            AtrousStore(pending, P, px, pendingY);
        pendingY = -1;
        if (PIPELINED && next < numSteps && isActive(next))
            bytes = LoadAtrousPixelBytes<DIFF, SPEC>(P, c, px, segY0 + next * H + ty);

        AtrousPixel<DIFF, SPEC, SH> a;
        if (active) {
            const int lc = (int)((uint32_t)(py - segTop) % (uint32_t)G::RR) * G::BW + lcol;
            const float centerViewZ = RelaxUnpackViewZ(c, ring.z[lc]);
            if (centerViewZ > c.shared.gDenoisingRange)
                active = false;
            else {
                const float4 centerWorldPosViewZ = F4(GetCurrentWorldPosFromPixelPos(c, px, py, centerViewZ), centerViewZ); // = the texel of the (world position, viewZ) guide plane
                float centerMaterialID;
                const float4 centerNormalRoughness = DecodedToNormalRoughness(EncodeDecodedRingWord(ring.nr[lc]), centerMaterialID); // = the texel of the decoded guide plane
                float4 centerSpecular = F4(0.0f), centerSpecularSH = F4(0.0f), centerDiffuse = F4(0.0f), centerDiffuseSH = F4(0.0f);
                if (SPEC) {
                    centerSpecular = DecodeRGBA16F(ring.spec[lc].x, ring.spec[lc].y);
                    if (SH)
                        centerSpecularSH = DecodeRGBA16F(ring.specSh[lc].x, ring.specSh[lc].y);
                }
                if (DIFF) {
                    centerDiffuse = DecodeRGBA16F(ring.diff[lc].x, ring.diff[lc].y);
                    if (SH)
                        centerDiffuseSH = DecodeRGBA16F(ring.diffSh[lc].x, ring.diffSh[lc].y);
                }
                AtrousBegin(a, P, c, px, py, centerWorldPosViewZ, centerNormalRoughness, centerMaterialID, centerSpecular, centerSpecularSH, centerDiffuse, centerDiffuseSH, in);
            }
        }
#pragma unroll
        for (int yy = -1; yy <= 1; yy++) {
            if (yy != -1) { // the rows only the finished taps read are dead: the rows of the next step go there, under the remaining taps
                const int freeRows = yy == 0 ? G::FREE1 : G::FREE2;
                if (freeRows > 0) {
                    __syncthreads();
                    if (nextAdjacent)
                        fill(top + G::RR + (yy == 0 ? 0 : G::FREE1), freeRows);
                }
            }
            if (active) {
                const int qy = py + offy + yy * STEP;
                const int rowBase = (int)((uint32_t)(qy - segTop) % (uint32_t)G::RR) * G::BW + lcol + offx;
                const int cy = ClampI(qy, 0, P.viewZ.h - 1);
#pragma unroll
                for (int xx = -1; xx <= 1; xx++) {
                    if (xx == 0 && yy == 0)
                        continue;
                    const int qx = px + offx + xx * STEP;
                    const bool isInside = (uint32_t)qx < (uint32_t)rectW && (uint32_t)qy < (uint32_t)rectH && InBounds(P.viewZ, qx, qy);
                    const float kernelW = (xx == 0 ? 0.44198f : 0.27901f) * (yy == 0 ? 0.44198f : 0.27901f);
                    const int li = rowBase + xx * STEP; // the ring holds the clamped texel of every tap position
                    const int cx = ClampI(qx, 0, P.viewZ.w - 1);
                    const float4 g0 = EncodeDecodedRingWord(ring.nr[li]);
                    const float tapZ = RelaxUnpackViewZ(c, ring.z[li]);
                    const float4 sampleWorldPosViewZ = F4(GetCurrentWorldPosFromPixelPos(c, cx, cy, tapZ), tapZ);
                    float4 sampleSpecular = F4(0.0f), sampleDiffuse = F4(0.0f);
                    uint2 rawSpecSh = make_uint2(0u, 0u), rawDiffSh = make_uint2(0u, 0u);
                    if (SPEC) {
                        const uint2 raw = ring.spec[li];
                        sampleSpecular = DecodeRGBA16F(raw.x, raw.y);
                        if (SH)
                            rawSpecSh = ring.specSh[li];
                    }
                    if (DIFF) {
                        const uint2 raw = ring.diff[li];
                        sampleDiffuse = DecodeRGBA16F(raw.x, raw.y);
                        if (SH)
                            rawDiffSh = ring.diffSh[li];
                    }
                    AtrousTap<DIFF, SPEC, SH, RES, MAT>(a, c, kernelW, isInside, g0, sampleWorldPosViewZ, sampleSpecular, sampleDiffuse, rawSpecSh, rawDiffSh);
                }
            }
        }
        if (active) {
            pending = AtrousFinish(a, c);
            if (PIPELINED)
                pendingY = py;
            else
                AtrousStore(pending, P, px, py);
        }
        if (nextAdjacent)
            ringFor = next;
    }
    if (pendingY >= 0)
        AtrousStore(pending, P, px, pendingY);
}

static bool AtrousLdsTilesEnabled() {
    static const bool v = !(getenv("NRD_HIP_ATROUS_LDS") && atoi(getenv("NRD_HIP_ATROUS_LDS")) == 0); // run-time A/B switch (results are identical)
    return NRD_ATROUS_LDS_TILES && v;
}
static int AtrousLdsBandsMaxStep() { // run-time A/B switch (results are identical): 0 = off, 8 = step 8 (default), 16 = steps 8 and 16
    static const int v = getenv("NRD_HIP_ATROUS_BANDS") ? atoi(getenv("NRD_HIP_ATROUS_BANDS")) : 8;
    return NRD_ATROUS_LDS_TILES ? v : 0;
}

// run-time A/B switches of the marching kernel (results are identical): NRD_HIP_ATROUS_MARCH = 0 (off) / 8 / 16 (steps 8 and 16: default); NRD_HIP_ATROUS_MARCH_SEG = steps of 16 rows per segment
static int AtrousMarchMaxStep() {
    static const int v = getenv("NRD_HIP_ATROUS_MARCH") ? atoi(getenv("NRD_HIP_ATROUS_MARCH")) : 0;
    return sizeof(NrRaw) == 4 ? v : 0; // (the ring's normal plane and its DMA jobs are laid out for 4-byte texels: encodings 0..2)
}
static int AtrousMarchSegSteps() {
    static const int v = getenv("NRD_HIP_ATROUS_MARCH_SEG") && atoi(getenv("NRD_HIP_ATROUS_MARCH_SEG")) > 0 ? atoi(getenv("NRD_HIP_ATROUS_MARCH_SEG")) : 8;
    return v;
}

template <bool DIFF, bool SPEC, bool SH>
const char* LaunchAtrous(const PassArgs& a) {
    if (const char* e = CheckSupportedRelax(a))
        return e;
    if (a.constantsSize < sizeof(RelaxCB))
        return "RELAX Atrous: constant block too small";
    AtrousPlanes P = {};
    if (!BindAtrous<DIFF, SPEC, SH, false>(a, P))
        return "RELAX Atrous: unexpected resource count";
    {
        const Plane sig = SPEC ? P.spec.in : P.diff.in;
        if (!SameLayout(P.worldPosViewZ, P.decodedNR) || !SameLayout(P.spec.in, sig) || !SameLayout(P.diff.in, sig) || !SameLayout(P.spec.inSh, sig) || !SameLayout(P.diff.inSh, sig) ||
            sig.w != P.decodedNR.w || sig.h != P.decodedNR.h)
            return "RELAX Atrous: the signal planes must share one layout and the frame size";
    }
    RelaxCB c = LoadRelaxConstants(a);
    RowGrid g = GridForRows(c.shared.gRectSize.x, c.shared.gRectSize.y, TILE_X, TILE_Y, a.rowBegin, a.rowEnd);
    const RowRange rr = MakeRowRange(g);
    const bool res = !SPEC || c.shared.gRoughnessEdgeStoppingEnabled != 0; // (irrelevant without a specular signal: one instantiation)
    const bool mat = c.shared.gSpecMinMaterial < 3.0f || c.shared.gDiffMinMaterial < 3.0f;
    if ((c.gStepSize == 8 || c.gStepSize == 16) && (int)c.gStepSize <= AtrousMarchMaxStep() && g.rowEnd > g.rowBegin && P.viewZ.w == P.decodedNR.w && P.viewZ.h == P.decodedNR.h &&
        P.normalRoughness.w == P.decodedNR.w && P.normalRoughness.h == P.decodedNR.h) {
        march::Range m;
        m.firstY = g.rowBegin / march::H * march::H;
        m.numSteps = (g.rowEnd - m.firstY + march::H - 1) / march::H;
        m.segSteps = AtrousMarchSegSteps() < m.numSteps ? AtrousMarchSegSteps() : m.numSteps;
        if (m.segSteps > 32) // (the kernel keeps the segment's tile flags in a 64-bit mask)
            m.segSteps = 32;
        m.numSegs = (m.numSteps + m.segSteps - 1) / m.segSteps;
        m.numStripes = (c.shared.gRectSize.x + march::W - 1) / march::W;
        m.stripesPerXcd = (m.numStripes + 7) / 8;
        m.rowBegin = g.rowBegin, m.rowEnd = g.rowEnd;
        const dim3 grid((unsigned)(8 * m.stripesPerXcd * m.numSegs));
#define NRD_LAUNCH_MARCH_M(STEP, RES, MAT) LaunchPass(a, (RelaxAtrousMarchKernel<DIFF, SPEC, SH, STEP, RES, MAT>), grid, dim3(march::THREADS), P, c, m)
#define NRD_LAUNCH_MARCH(STEP, RES) (mat ? NRD_LAUNCH_MARCH_M(STEP, RES, true) : NRD_LAUNCH_MARCH_M(STEP, RES, false))
        if (c.gStepSize == 8)
            res ? NRD_LAUNCH_MARCH(8, true) : NRD_LAUNCH_MARCH(8, SPEC ? false : true);
        else
            res ? NRD_LAUNCH_MARCH(16, true) : NRD_LAUNCH_MARCH(16, SPEC ? false : true);
#undef NRD_LAUNCH_MARCH_M
#undef NRD_LAUNCH_MARCH
        return nullptr;
    }
    const int step = (AtrousLdsTilesEnabled() && (c.gStepSize == 2 || c.gStepSize == 4)) || ((c.gStepSize == 8 || c.gStepSize == 16) && (int)c.gStepSize <= AtrousLdsBandsMaxStep()) ? (int)c.gStepSize : 0;
#define NRD_LAUNCH_ATROUS_M(STEP, RES, MAT) LaunchPass(a, (RelaxAtrousKernel<DIFF, SPEC, SH, STEP, RES, MAT>), g.grid, dim3(256), P, c, rr)
    // frames larger than the caches: the instantiation with hinted stores (planes.h NRD_NT_STORE_PIXELS; RELAX 4K: -2..3 % per iteration). Only the twin without material tests has one.
    const bool nt = !mat && (uint32_t)P.viewZ.w * (uint32_t)P.viewZ.h > NRD_NT_STORE_PIXELS;
#define NRD_LAUNCH_ATROUS(STEP, RES) (mat ? NRD_LAUNCH_ATROUS_M(STEP, RES, true) : nt ? LaunchPass(a, (RelaxAtrousKernel<DIFF, SPEC, SH, STEP, RES, false, true>), g.grid, dim3(256), P, c, rr) : NRD_LAUNCH_ATROUS_M(STEP, RES, false))
    if (step == 2)
        res ? NRD_LAUNCH_ATROUS(2, true) : NRD_LAUNCH_ATROUS(2, SPEC ? false : true);
    else if (step == 4)
        res ? NRD_LAUNCH_ATROUS(4, true) : NRD_LAUNCH_ATROUS(4, SPEC ? false : true);
    else if (step == 8)
        res ? NRD_LAUNCH_ATROUS(8, true) : NRD_LAUNCH_ATROUS(8, SPEC ? false : true);
    else if (step == 16)
        res ? NRD_LAUNCH_ATROUS(16, true) : NRD_LAUNCH_ATROUS(16, SPEC ? false : true);
    else
        res ? NRD_LAUNCH_ATROUS(0, true) : NRD_LAUNCH_ATROUS(0, SPEC ? false : true);
#undef NRD_LAUNCH_ATROUS_M
#undef NRD_LAUNCH_ATROUS
    return nullptr;
}

} // namespace

#define RELAX_ATROUS_VARIANT(name, D, S, H)                           \
    {"RELAX_" name "_AtrousSmem.cs", LaunchAtrousSmem<D, S, H>},     \
    {"RELAX_" name "_Atrous.cs", LaunchAtrous<D, S, H>}

const PassEntry* GetRelaxAtrousPasses(uint32_t& num) {
    static const PassEntry k[] = {
        RELAX_ATROUS_VARIANT("Diffuse", true, false, false),
        RELAX_ATROUS_VARIANT("DiffuseSh", true, false, true),
        RELAX_ATROUS_VARIANT("Specular", false, true, false),
        RELAX_ATROUS_VARIANT("SpecularSh", false, true, true),
        RELAX_ATROUS_VARIANT("DiffuseSpecular", true, true, false),
        RELAX_ATROUS_VARIANT("DiffuseSpecularSh", true, true, true),
    };
    num = sizeof(k) / sizeof(k[0]);
    return k;
}

} // namespace nrdhip
