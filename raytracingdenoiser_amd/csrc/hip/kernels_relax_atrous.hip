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
template <bool DIFF, bool SPEC, bool SH, int STEP, bool RES, bool MAT>
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
    constexpr bool BAND_RAW = STEP == 16;
    __shared__ float4 b_NR[BAND_RAW ? 1 : BN], b_Pos[BAND_RAW ? 1 : BN];
    __shared__ uint32_t br_NR[BAND_RAW ? BN : 1];
    __shared__ float br_Z[BAND_RAW ? BN : 1];
    __shared__ uint2 b_Spec[SPEC && BANDED ? BN : 1], b_Diff[DIFF && BANDED ? BN : 1], b_SpecSh[SPEC && SH && BANDED ? BN : 1], b_DiffSh[DIFF && SH && BANDED ? BN : 1];

    const int blockY = BlockTileY(rows, true);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
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
    const float3 centerNormal = Xyz(centerNormalRoughness);
    const float centerRoughness = centerNormalRoughness.w;
    const float historyLength = 255.0f * LoadR8Unorm(P.historyLength, px, py);
    const int stepSize = TILED || BANDED ? STEP : (int)c.gStepSize;

    float diffuseLobeAngleFraction = Div(c.shared.gLobeAngleFraction, Sqrt(float(c.gStepSize)));
    if (SH)
        diffuseLobeAngleFraction = Rcp(Sqrt(float(c.gStepSize)));
    diffuseLobeAngleFraction = Lerp(0.99f, diffuseLobeAngleFraction, Sat(historyLength * (1.0f / 5.0f)));

    SpecParams sp = {};
    sp.luminanceWeightRelaxation = 1.0f;
    float4 sumSpecular = F4(0.0f), sumSpecularSH = F4(0.0f);
    float sumWSpecular = 0.44198f * 0.44198f, roughnessModified = 0.0f;
    if (SPEC) {
        const float4 centerSpecular = TILED ? s_Spec[lc] : LoadRGBA16F(P.spec.in, px, py);
        sp.centerLuminance = Luminance(Xyz(centerSpecular));
        const float centerSpecularVar = centerSpecular.w;
        sp.phiLIlluminationInv = Rcp(Max(1.0e-4f, c.shared.gSpecPhiLuminance * Sqrt(centerSpecularVar)));

        sp.roughnessWeightParams = GetRoughnessWeightParams(centerRoughness, c.shared.gRoughnessFraction);
        float diffuseLobeAngleFractionForSimplifiedSpecularNormalWeight = diffuseLobeAngleFraction;
        float specularLobeAngleFraction = c.shared.gLobeAngleFraction;
        const float specularReprojectionConfidence = LoadR8Unorm(P.specReprojectionConfidence, px, py);
        if (c.gStepSize <= 4)
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
        sp.normalWeightParams = GetNormalWeightParams_ATrous(centerRoughness, historyLength, specularReprojectionConfidence, c.shared.gNormalEdgeStoppingRelaxation, specularLobeAngleFraction,
            c.shared.gSpecLobeAngleSlack);

        sumSpecular = centerSpecular * F4(sumWSpecular, sumWSpecular, sumWSpecular, sumWSpecular * sumWSpecular);
        if (SH) {
            float4 centerSpecularSH;
            if (TILED)
                centerSpecularSH = DecodeRGBA16F(s_SpecSh[lc].x, s_SpecSh[lc].y);
            else
                centerSpecularSH = LoadRGBA16F(P.spec.inSh, px, py);
            sumSpecularSH = centerSpecularSH * sumWSpecular;
            roughnessModified = centerSpecularSH.w;
        }
    }

    DiffParams dp = {};
    dp.luminanceWeightRelaxation = 1.0f;
    float4 sumDiffuse = F4(0.0f), sumDiffuseSH = F4(0.0f);
    float sumWDiffuse = 0.44198f * 0.44198f;
    if (DIFF) {
        const float4 centerDiffuse = TILED ? s_Diff[lc] : LoadRGBA16F(P.diff.in, px, py);
        dp.centerLuminance = Luminance(Xyz(centerDiffuse));
        const float centerDiffuseVar = centerDiffuse.w;
        dp.phiLIlluminationInv = Rcp(Max(1.0e-4f, c.shared.gDiffPhiLuminance * Sqrt(centerDiffuseVar)));
        if (c.shared.gHasHistoryConfidence) {
            float diffConfidenceDrivenRelaxation = Sat(c.shared.gConfidenceDrivenRelaxationMultiplier * (1.0f - LoadR8Unorm(P.diff.confidence, px, py)));
            float r = Sat(diffConfidenceDrivenRelaxation * c.shared.gConfidenceDrivenNormalEdgeStoppingRelaxation);
            diffuseLobeAngleFraction = Lerp(diffuseLobeAngleFraction, 1.0f, r);
            r = Sat(diffConfidenceDrivenRelaxation * c.shared.gConfidenceDrivenLuminanceEdgeStoppingRelaxation);
            dp.luminanceWeightRelaxation = 1.0f - r;
        }
        dp.normalWeightParam = GetNormalWeightParam2(1.0f, diffuseLobeAngleFraction);
        sumDiffuse = centerDiffuse * F4(sumWDiffuse, sumWDiffuse, sumWDiffuse, sumWDiffuse * sumWDiffuse);
        if (SH) {
            if (TILED)
                sumDiffuseSH = DecodeRGBA16F(s_DiffSh[lc].x, s_DiffSh[lc].y) * sumWDiffuse;
            else
                sumDiffuseSH = LoadRGBA16F(P.diff.inSh, px, py) * sumWDiffuse;
        }
    }

    const float3 centerWorldPos = Xyz(centerWorldPosViewZ);
    const float3 centerV = -Normalize(centerWorldPos);
    const float depthThreshold = c.shared.gDepthThreshold * centerViewZ;

    // random offsets against ringing at large steps
    int offx = 0, offy = 0;
    if (!TILED && c.gStepSize > 4) { // (BANDED: STEP is 8 or 16, always true)
        RngHash rng;
        rng.Initialize((uint32_t)px, (uint32_t)py, c.shared.gFrameIndex);
        float2 rnd = rng.GetFloat2();
        offx = (int)(float(c.gStepSize) * 0.5f * (rnd.x - 0.5f));
        offy = (int)(float(c.gStepSize) * 0.5f * (rnd.y - 0.5f));
    }

    // The 8 taps, branch-free: a tap outside the rect reads the clamped texel (a valid address) and gets weight 0 through isInside, exactly what the
    // reference's "Load outside = 0" amounts to (its geometry weight is multiplied by isInside); a tap whose guide weight is <= 1e-4 keeps weight 0
    // instead of being skipped by a divergent branch, so all loads of a pixel can be in flight together instead of one dependent wait per tap and signal.
    // (0 * sample adds nothing: the history planes hold finite fp16 values by construction.) Planes of one format share their layout (launcher check),
    // so one texel offset serves the two guide planes and one the four signal planes.
    constexpr bool compareSpecMaterials = MAT, compareDiffMaterials = MAT; // compile-time variant (RelaxAtrousSmemKernel): IDs are 0..3, a minimum >= 3 disables the test
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
                    br_NR[li] = *(const uint32_t*)(P.normalRoughness.ptr + TexelOffset(P.normalRoughness, cx, cy, 4u, true));
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
                const uint32_t guideOffset = TexelOffset(P.decodedNR, cx, cy, 16u, true);
#if NRD_ATROUS_GUIDES_RAW_NR
                // same reasoning for the normal: 4 bytes of IN_NORMAL_ROUGHNESS through the L1 (41.7 cycles per scattered wave-load against 149.7 for the 16-byte decoded texel,
                // profiles/r02_c_gather_bench.txt) and the decode that wrote the guide plane (kernels_common.hip DecodeGuidesRelaxKernel) redone per tap: it IS the stored value
                g0 = EncodeDecodedNormalRoughness(*(const uint32_t*)(P.normalRoughness.ptr + TexelOffset(P.normalRoughness, cx, cy, 4u, true)));
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

            float sampleMaterialID;
            const float4 sampleNormalRoughness = DecodedToNormalRoughness(g0, sampleMaterialID);
            const float3 sampleNormal = Xyz(sampleNormalRoughness);
            const float sampleRoughness = sampleNormalRoughness.w;
            const float sampleViewZ = sampleWorldPosViewZ.w;
            const float3 sampleWorldPos = Xyz(sampleWorldPosViewZ);

            float geometryW = GetPlaneDistanceWeight_Atrous(centerWorldPos, centerNormal, sampleWorldPos, depthThreshold);
            geometryW *= kernelW;
            geometryW *= Cmp(isInside && sampleViewZ < c.shared.gDenoisingRange);

            if (SPEC) {
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

                sumWSpecular += wSpecular;
                sumSpecular = Mad(sampleSpecular, F4(wSpecular, wSpecular, wSpecular, wSpecular * wSpecular), sumSpecular);
                if (SH)
                    sumSpecularSH = Mad(DecodeRGBA16F(rawSpecSh.x, rawSpecSh.y), wSpecular, sumSpecularSH);
            }
            if (DIFF) {
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

                sumWDiffuse += wDiffuse;
                sumDiffuse = Mad(sampleDiffuse, F4(wDiffuse, wDiffuse, wDiffuse, wDiffuse * wDiffuse), sumDiffuse);
                if (SH)
                    sumDiffuseSH = Mad(DecodeRGBA16F(rawDiffSh.x, rawDiffSh.y), wDiffuse, sumDiffuseSH);
            }
        }
    }
    if (BANDED && !active)
        return;

    if (SPEC) {
        float4 filtered = Div(sumSpecular, F4(sumWSpecular, sumWSpecular, sumWSpecular, sumWSpecular * sumWSpecular));
        if (SH) {
            if (c.gIsLastPass == 1)
                filtered = F4(LinearToYCoCg(Xyz(filtered)), filtered.w);
            StoreRGBA16F(P.spec.outSh, px, py, F4(Div(Xyz(sumSpecularSH), sumWSpecular), roughnessModified));
        }
        StoreRGBA16F(P.spec.out, px, py, filtered);
    }
    if (DIFF) {
        float4 filtered = Div(sumDiffuse, F4(sumWDiffuse, sumWDiffuse, sumWDiffuse, sumWDiffuse * sumWDiffuse));
        if (SH) {
            if (c.gIsLastPass == 1)
                filtered = F4(LinearToYCoCg(Xyz(filtered)), filtered.w);
            StoreRGBA16F(P.diff.outSh, px, py, Div(sumDiffuseSH, sumWDiffuse));
        }
        StoreRGBA16F(P.diff.out, px, py, filtered);
    }
#undef px
#undef py
}

static bool AtrousLdsTilesEnabled() {
    static const bool v = !(getenv("NRD_HIP_ATROUS_LDS") && atoi(getenv("NRD_HIP_ATROUS_LDS")) == 0); // run-time A/B switch (results are identical)
    return NRD_ATROUS_LDS_TILES && v;
}
static int AtrousLdsBandsMaxStep() { // run-time A/B switch (results are identical): 0 = off, 8 = step 8 (default), 16 = steps 8 and 16
    static const int v = getenv("NRD_HIP_ATROUS_BANDS") ? atoi(getenv("NRD_HIP_ATROUS_BANDS")) : 8;
    return NRD_ATROUS_LDS_TILES ? v : 0;
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
    const int step = (AtrousLdsTilesEnabled() && (c.gStepSize == 2 || c.gStepSize == 4)) || ((c.gStepSize == 8 || c.gStepSize == 16) && (int)c.gStepSize <= AtrousLdsBandsMaxStep()) ? (int)c.gStepSize : 0;
    const bool res = !SPEC || c.shared.gRoughnessEdgeStoppingEnabled != 0; // (irrelevant without a specular signal: one instantiation)
    const bool mat = c.shared.gSpecMinMaterial < 3.0f || c.shared.gDiffMinMaterial < 3.0f;
#define NRD_LAUNCH_ATROUS_M(STEP, RES, MAT) LaunchPass(a, (RelaxAtrousKernel<DIFF, SPEC, SH, STEP, RES, MAT>), g.grid, dim3(256), P, c, rr)
#define NRD_LAUNCH_ATROUS(STEP, RES) (mat ? NRD_LAUNCH_ATROUS_M(STEP, RES, true) : NRD_LAUNCH_ATROUS_M(STEP, RES, false))
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
