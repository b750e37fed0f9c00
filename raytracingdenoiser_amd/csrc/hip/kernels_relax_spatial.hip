// RELAX per-pixel passes as HIP kernels for gfx950: ClassifyTiles, PrePass, HistoryFix, SplitScreen.
//   ClassifyTiles   reference Shaders/Source/RELAX_ClassifyTiles.cs.hlsl:19-53
//   PrePass         reference Shaders/Include/RELAX_PrePass.hlsli:13-346
//   HistoryFix      reference Shaders/Include/RELAX_HistoryFix.hlsli:11-160
//   SplitScreen     reference Shaders/Include/RELAX_SplitScreen.hlsli:10-52
//
// MI355X mapping. PrePass is a sparse gather (8 Poisson taps per signal at radii up to 30-50 px) and HistoryFix a sparse
// 5x5 cross-bilateral with strides up to 14 px that only runs on freshly disoccluded pixels: neither has a stencil an LDS
// tile could cover, so taps are served by L2 / Infinity Cache. 32x8-pixel workgroups keep the centre accesses of a wave
// in two 256-byte RGBA16F row segments; the 704-byte constant block travels as a kernel argument (scalar loads); sky
// tiles leave before touching a signal plane. All variants (diffuse / specular / both, +-SH) come from one template.
// ClassifyTiles is one wave per 16x16 tile with a wave-wide vote, no LDS and no atomics.
#include "relax_device.h"

namespace nrdhip {

namespace {

constexpr int TILE_X = RELAX_TILE_X;
constexpr int TILE_Y = RELAX_TILE_Y;

__device__ __constant__ const float g_Poisson8[8][3] = { // reference Shaders/Include/Poisson.hlsli:40-50
    {-0.4706069f, -0.4427112f, +0.6461146f}, {-0.9057375f, +0.3003471f, +0.9542373f}, {-0.3487388f, +0.4037880f, +0.5335386f}, {+0.1023042f, +0.6439373f, +0.6520134f},
    {+0.5699277f, +0.3513750f, +0.6695386f}, {+0.2939128f, -0.1131226f, +0.3149309f}, {+0.7836658f, -0.4208784f, +0.8895339f}, {+0.1564120f, -0.8198990f, +0.8346850f}};

// GetGaussianWeight( g_Poisson8[i].z ) = Exp( -0.66 z^2 ) of the eight constants above as OUR Exp() returns them, baked (bit patterns 0x3f425921 0x3f0c5bdb 0x3f5426ba
// 0x3f415e51 0x3f3e6f61 0x3f6fc76c 0x3f17db60 0x3f21a332; tests/test_numerics.py re-derives them on the GPU): the tap loops are not fully unrolled, so the
// compiler evaluated two multiplications and an exponential per tap and signal
__device__ __constant__ const float g_Poisson8Gaussian[8] = {0.7591725f, 0.5482766f, 0.8287159f, 0.7553454f, 0.743887f, 0.9366367f, 0.59319115f, 0.6313964f};

// ================================================================================================ ClassifyTiles
__global__ __launch_bounds__(256) void RelaxClassifyTilesKernel(Plane viewZ, Plane tiles, float denoisingRange, int tilesPerRow, int tileRows) {
    // tilesPerRow x tileRows = the tiles of the RECT (dynamic resolution: the reference dispatches ceil(rect / 16) groups; tiles beyond stay untouched)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tileIndex = blockIdx.x * 4 + wave;
    if (tileIndex >= tilesPerRow * tileRows)
        return;
    const int tx = tileIndex % tilesPerRow, ty = tileIndex / tilesPerRow;
    const int x = tx * 16 + (lane & 3) * 4, y = ty * 16 + (lane >> 2);

    bool allSky = true;
    if (y < viewZ.h && x + 3 < viewZ.w) {
        float4 z = *(const float4*)TexelPtr<const float>(viewZ, x, y); // tile rows are 64-byte aligned
        allSky = Abs(z.x) > denoisingRange && Abs(z.y) > denoisingRange && Abs(z.z) > denoisingRange && Abs(z.w) > denoisingRange;
    } else {
        for (int i = 0; i < 4; i++)
            allSky = allSky && Abs(LoadR32FOrZero(viewZ, x + i, y)) > denoisingRange; // out-of-bounds load = 0: a partial edge tile is never sky
    }
    bool tileIsSky = __all(allSky);
    if (lane == 0)
        StoreR8Unorm(tiles, tx, ty, tileIsSky ? 1.0f : 0.0f);
}

const char* LaunchClassifyTiles(const PassArgs& a) {
    if (const char* e = CheckSupportedRelax(a))
        return e;
    const nrdc::RelaxConstants& c = *(const nrdc::RelaxConstants*)a.constants;
    const Plane& tiles = a.planes[1];
    const int tilesPerRow = (c.gRectSize.x + 15) / 16, tileRows = (c.gRectSize.y + 15) / 16;
    if (tilesPerRow > tiles.w || tileRows > tiles.h)
        return "RELAX ClassifyTiles: the rect does not fit the tile plane";
    if (a.fuseGuidesFrom.ptr) { // the frame's guide planes are due: one kernel decodes them and classifies (kernels_common.hip)
        LaunchDecodeGuidesClassifyRelax(a, a.planes[0], tiles, a.constants, tilesPerRow, tileRows);
        return nullptr;
    }
    int numTiles = tilesPerRow * tileRows;
    LaunchPass(a, RelaxClassifyTilesKernel, dim3((numTiles + 3) / 4), dim3(256), a.planes[0], tiles, c.gDenoisingRange, tilesPerRow, tileRows);
    return nullptr;
}

// ================================================================================================ HitDistReconstruction
// reference Shaders/Include/RELAX_HitDistReconstruction.hlsli:10-160; an optional pass, the 3x3 / 5x5 window is read at
// rect-clamped coordinates straight from L1/L2. Hit distances are (spec, diff). NOTE (kept): the reference feeds the CENTER
// roughness to the roughness weight, which therefore evaluates to exactly 1.
struct HitDistPlanes {
    Plane tiles, viewZ, decodedNR;
    SignalPlanes spec, diff;
};

template <bool DIFF, bool SPEC, int BORDER>
__global__ __launch_bounds__(256) void RelaxHitDistReconstructionKernel(HitDistPlanes P, RelaxCB c, RowRange rows) {
    const int blockY = BlockTileY(rows, true);
    const int px = BlockTileX(rows) * TILE_X + (threadIdx.x & 31), py = blockY * TILE_Y + (threadIdx.x >> 5);
    const int rectW = c.shared.gRectSize.x, rectH = c.shared.gRectSize.y;
    if (px >= rectW || py >= rectH || py < rows.rowBegin || py >= rows.rowEnd)
        return;
    if (LoadR8Unorm(P.tiles, px >> 4, py >> 4) != 0.0f)
        return;
    const float centerViewZ = RelaxUnpackViewZ(c, LoadR32F(P.viewZ, px, py));
    if (centerViewZ > c.shared.gDenoisingRange)
        return;

    const float2 rectSizeInv = ToF2(c.shared.gRectSizeInv);
    const float2 pixelUv = F2(float(px) + 0.5f, float(py) + 0.5f) * rectSizeInv;
    const float4 normalAndRoughness = LoadDecodedNormalRoughness(P.decodedNR, px, py);
    const float3 centerNormal = Xyz(normalAndRoughness);
    const float centerRoughness = normalAndRoughness.w;
    const float4 centerSpec = SPEC ? LoadRGBA16F(P.spec.in, px, py) : F4(0.0f), centerDiff = DIFF ? LoadRGBA16F(P.diff.in, px, py) : F4(0.0f);
    const float2 centerHitDist = F2(SPEC ? centerSpec.w : c.shared.gDenoisingRange, DIFF ? centerDiff.w : c.shared.gDenoisingRange);

    const float2 relaxedRoughnessWeightParams = GetRelaxedRoughnessWeightParams(centerRoughness * centerRoughness);
    const float specularNormalWeightParam = GetNormalWeightParam(1.0f, 1.0f, centerRoughness);
    const float diffuseNormalWeightParam = GetNormalWeightParam(1.0f, 1.0f);

    float sumSpecularWeight = 1000.0f * Cmp(centerHitDist.x != 0.0f);
    float sumSpecularHitDist = centerHitDist.x * sumSpecularWeight;
    float sumDiffuseWeight = 1000.0f * Cmp(centerHitDist.y != 0.0f);
    float sumDiffuseHitDist = centerHitDist.y * sumDiffuseWeight;

    for (int dy = 0; dy <= BORDER * 2; dy++)
        for (int dx = 0; dx <= BORDER * 2; dx++) {
            const int ix = dx - BORDER, iy = dy - BORDER;
            if (ix == 0 && iy == 0)
                continue;
            const float2 o = F2(float(ix), float(iy));
            const int sx = ClampI(px + ix, 0, rectW - 1), sy = ClampI(py + iy, 0, rectH - 1);
            const float3 sampleNormal = Xyz(LoadDecodedNormalRoughness(P.decodedNR, sx, sy));
            const float sampleViewZ = RelaxUnpackViewZ(c, LoadR32F(P.viewZ, sx, sy));
            const float cosa = Dot(centerNormal, sampleNormal);
            const float angle = AcosApprox(cosa);

            float w = IsInScreenNearest(pixelUv + o * rectSizeInv);
            w *= Cmp(sampleViewZ < c.shared.gDenoisingRange);
            w *= GetGaussianWeight(Length(o) * 0.5f);
            w *= LinearStep(0.03f, 0.0f, Abs(sampleViewZ - centerViewZ) * Rcp(Max(sampleViewZ, centerViewZ))); // GetBilateralWeight

            if (SPEC) {
                float specularWeight = w;
                specularWeight *= ComputeExponentialWeight(angle, specularNormalWeightParam, 0.0f);
                specularWeight *= ComputeExponentialWeight(normalAndRoughness.w * normalAndRoughness.w, relaxedRoughnessWeightParams.x, relaxedRoughnessWeightParams.y);
                float sampleSpecularHitDist = specularWeight == 0.0f ? 0.0f : LoadRGBA16F(P.spec.in, sx, sy).w;
                specularWeight *= Cmp(sampleSpecularHitDist != 0.0f);
                sumSpecularHitDist += sampleSpecularHitDist * specularWeight;
                sumSpecularWeight += specularWeight;
            }
            if (DIFF) {
                float diffuseWeight = w;
                diffuseWeight *= ComputeExponentialWeight(angle, diffuseNormalWeightParam, 0.0f);
                float sampleDiffuseHitDist = diffuseWeight == 0.0f ? 0.0f : LoadRGBA16F(P.diff.in, sx, sy).w;
                diffuseWeight *= Cmp(sampleDiffuseHitDist != 0.0f);
                sumDiffuseHitDist += diffuseWeight == 0.0f ? 0.0f : sampleDiffuseHitDist * diffuseWeight;
                sumDiffuseWeight += diffuseWeight;
            }
        }

    if (SPEC) {
        sumSpecularHitDist = Div(sumSpecularHitDist, Max(sumSpecularWeight, 1e-6f));
        StoreRGBA16F(P.spec.out, px, py, F4(Xyz(centerSpec), sumSpecularHitDist));
    }
    if (DIFF) {
        sumDiffuseHitDist = Div(sumDiffuseHitDist, Max(sumDiffuseWeight, 1e-6f));
        StoreRGBA16F(P.diff.out, px, py, F4(Xyz(centerDiff), sumDiffuseHitDist));
    }
}

template <bool DIFF, bool SPEC, int BORDER>
const char* LaunchHitDistReconstruction(const PassArgs& a) {
    if (const char* e = CheckSupportedRelax(a))
        return e;
    PlaneCursor cur(a);
    HitDistPlanes P = {};
    P.tiles = cur.next();
    if (SPEC) P.spec.in = cur.next();
    if (DIFF) P.diff.in = cur.next();
    cur.next(); // packed normal / roughness: read through the decoded cache
    P.viewZ = cur.next();
    if (SPEC) P.spec.out = cur.next();
    if (DIFF) P.diff.out = cur.next();
    P.decodedNR = a.decodedNormalRoughness;
    if (!cur.complete() || !P.decodedNR.ptr)
        return "RELAX HitDistReconstruction: unexpected resource count or missing decoded normal/roughness cache";
    RelaxCB c = LoadRelaxConstants(a);
    RowGrid g = GridForRows(c.shared.gRectSize.x, c.shared.gRectSize.y, TILE_X, TILE_Y, a.rowBegin, a.rowEnd);
    LaunchPass(a, (RelaxHitDistReconstructionKernel<DIFF, SPEC, BORDER>), g.grid, dim3(256), P, c, MakeRowRange(g));
    return nullptr;
}

// ================================================================================================ PrePass
struct PrePassPlanes {
    Plane tiles, normalRoughness, viewZ;
    Plane decodedNR; // executor's float4 cache of normalRoughness (reblur_device.h "decoded guides")
    Plane worldPos;  // executor's float4 guide plane (world position, viewZ; passes.h), same layout as decodedNR
    SignalPlanes spec, diff;
};

struct PrePassTap {
    float2 uv;
    int2 texel;       // nearest texel of the guides
    int2 signalTexel; // nearest texel of the noisy signal: the same one, or in the left half of the plane for a checkerboarded input
};

// CB = a checkerboard mode is on (reference RELAX_PrePass.hlsli:29-60, :140-143): a tap that lands on a pixel without data moves one pixel sideways
template <bool CB>
NRD_D PrePassTap MakeTap(const RelaxCB& c, const Plane& guide, const Plane& signal, uint32_t checkerboardMode, float2 pixelUv, float2 rectSize, float4 rotator, int i, float blurRadius) {
    PrePassTap t;
    float2 uv = pixelUv * rectSize + RotateVector(rotator, F2(g_Poisson8[i][0], g_Poisson8[i][1])) * blurRadius;
    uv = Floor(uv) + 0.5f;
    if (CB)
        uv = ApplyCheckerboardShift(uv, checkerboardMode, (uint32_t)i, c.shared.gFrameIndex);
    uv = uv * ToF2(c.shared.gRectSizeInv);
    float2 uvScaled = RelaxClampUvToViewport(c, uv);
    t.uv = uv;
    t.texel = NearestTexel(guide, uvScaled + ToF2(c.shared.gRectOffset));
    t.signalTexel = t.texel;
    if (CB)
        t.signalTexel = NearestTexel(signal, F2(uvScaled.x * (checkerboardMode != 2u ? 0.5f : 1.0f), uvScaled.y));
    return t;
}

// Guides of one pre-pass tap. FR ("full rect": rect == resource, no checkerboard; picked by the launcher) reads them from the per-frame guide planes at
// ONE texel offset -- the snapped tap position IS the texel, and the world position stored there is exactly GetCurrentWorldPosFromClipSpaceXY of that pixel
// centre -- instead of re-deriving uv, texel index and world position per tap (see kernels_reblur_spatial.hip FetchTapGuides: same values bit for bit).
struct PrePassGuides {
    float inScreen;
    float3 normal, worldPos;
    float roughness, materialID, viewZ;
    int2 signalTexel;
};
template <bool CB, bool FR>
NRD_D PrePassGuides FetchPrePassGuides(const RelaxCB& c, const PrePassPlanes& P, const Plane& signal, uint32_t checkerboardMode, float2 pixelUv, float2 rectSize, float4 rotator, int i, float blurRadius) {
    PrePassGuides g;
    if (FR) {
        float2 uv = pixelUv * rectSize + RotateVector(rotator, F2(g_Poisson8[i][0], g_Poisson8[i][1])) * blurRadius;
        uv = Floor(uv);
        const float cxf = __builtin_amdgcn_fmed3f(uv.x, 0.0f, rectSize.x - 1.0f), cyf = __builtin_amdgcn_fmed3f(uv.y, 0.0f, rectSize.y - 1.0f);
        g.inScreen = (cxf == uv.x && cyf == uv.y) ? 1.0f : 0.0f;
        g.signalTexel = make_int2((int)cxf, (int)cyf);
        // decoded normal (16 B) + viewZ (4 B); the world position is re-derived from viewZ: the taps are bound by the L1 / texture-address path, and a
        // second 16-byte guide texel per tap made this pass 13 % SLOWER (profiles/r02_c_relax_bench_fast.json vs r02_b) despite fewer instructions
        const float4 nr = LoadDecodedNormalRoughness(P.decodedNR, g.signalTexel.x, g.signalTexel.y, g.materialID);
        g.normal = Xyz(nr);
        g.roughness = nr.w;
        g.viewZ = RelaxUnpackViewZ(c, LoadR32F(P.viewZ, g.signalTexel.x, g.signalTexel.y));
        const float2 uvc = (uv + 0.5f) * ToF2(c.shared.gRectSizeInv); // centre of the snapped pixel (= the texel's centre whenever the tap counts)
        g.worldPos = GetCurrentWorldPosFromClipSpaceXY(c, uvc * 2.0f - 1.0f, g.viewZ);
        return g;
    }
    PrePassTap t = MakeTap<CB>(c, P.viewZ, signal, checkerboardMode, pixelUv, rectSize, rotator, i, blurRadius);
    const float4 nr = LoadDecodedNormalRoughness(P.decodedNR, t.texel.x, t.texel.y, g.materialID);
    g.normal = Xyz(nr);
    g.roughness = nr.w;
    g.viewZ = RelaxUnpackViewZ(c, LoadR32F(P.viewZ, t.texel.x, t.texel.y));
    g.worldPos = GetCurrentWorldPosFromClipSpaceXY(c, t.uv * 2.0f - 1.0f, g.viewZ);
    g.inScreen = IsInScreenNearest(t.uv);
    g.signalTexel = t.signalTexel;
    return g;
}

// MAT: material tests compiled in (launcher; only the full-rect variant has a MAT = false twin). Material IDs are 0..3: with a minimum material >= 3 (the default is 4) every
// comparison holds. A run-time test inside the tap loop -- even a uniform one -- is if-converted into compare + select and saves nothing (round 5: a-trous 1 340 -> 1 200 instructions).
// NT: outputs stored with the non-temporal hint (planes.h StoreRGBA16FHinted; the launcher picks it for frames above NRD_NT_STORE_PIXELS: 4K -6 % for this pass, round 6)
template <bool DIFF, bool SPEC, bool SH, bool CB, bool FR, bool MAT = true, bool NT = false>
__global__ __launch_bounds__(256, NRD_WAVES_RELAX_PREPASS) void RelaxPrePassKernel(PrePassPlanes P, RelaxCB c, RowRange rows) {
    const int blockY = BlockTileY(rows, true);
    const int px = BlockTileX(rows) * TILE_X + (threadIdx.x & 31), py = blockY * TILE_Y + (threadIdx.x >> 5);
    const int rectW = c.shared.gRectSize.x, rectH = c.shared.gRectSize.y;
    if (px >= rectW || py >= rectH || py < rows.rowBegin || py >= rows.rowEnd)
        return;
    if (LoadR8Unorm(P.tiles, px >> 4, py >> 4) != 0.0f)
        return;
    float centerViewZ = RelaxUnpackViewZ(c, LoadR32F(P.viewZ, px, py));
    if (centerViewZ > c.shared.gDenoisingRange)
        return;

    float centerMaterialID;
    float4 centerNormalRoughness = LoadDecodedNormalRoughness(P.decodedNR, px, py, centerMaterialID);
    float3 centerNormal = Xyz(centerNormalRoughness);
    float centerRoughness = centerNormalRoughness.w;
    float3 centerWorldPos = GetCurrentWorldPosFromPixelPos(c, px, py, centerViewZ);
    const float4 rotator = ToF4(c.shared.gRotatorPre);
    const float2 rectSize = F2(float(rectW), float(rectH));
    const float2 pixelUv = F2(float(px) + 0.5f, float(py) + 0.5f) * ToF2(c.shared.gRectSizeInv);
    const float minRectDim = float(rectW < rectH ? rectW : rectH);

    // Checkerboard resolve weights (reference RELAX_PrePass.hlsli:29-60)
    uint32_t checkerboard = 0;
    int cbx0 = 0, cbx1 = 0;
    float materialID0 = 0.0f, materialID1 = 0.0f;
    float2 checkerboardResolveWeights = F2(1.0f, 1.0f);
    if (CB) {
        checkerboard = CheckerBoard((uint32_t)px, (uint32_t)py, c.shared.gFrameIndex);
        cbx0 = px > 0 ? px - 1 : 0;
        cbx1 = px + 1 < rectW ? px + 1 : rectW - 1;
        const float viewZ0 = RelaxUnpackViewZ(c, LoadR32F(P.viewZ, cbx0, py)), viewZ1 = RelaxUnpackViewZ(c, LoadR32F(P.viewZ, cbx1, py));
        LoadDecodedNormalRoughness(P.decodedNR, cbx0, py, materialID0);
        LoadDecodedNormalRoughness(P.decodedNR, cbx1, py, materialID1);
        checkerboardResolveWeights = F2(LinearStep(0.03f, 0.0f, Abs(viewZ0 - centerViewZ) * Rcp(Max(viewZ0, centerViewZ))), // GetBilateralWeight
            LinearStep(0.03f, 0.0f, Abs(viewZ1 - centerViewZ) * Rcp(Max(viewZ1, centerViewZ))));
        checkerboardResolveWeights.x = (viewZ0 > c.shared.gDenoisingRange || px < 1) ? 0.0f : checkerboardResolveWeights.x;
        checkerboardResolveWeights.y = (viewZ1 > c.shared.gDenoisingRange || px > rectW - 2) ? 0.0f : checkerboardResolveWeights.y;
        cbx0 >>= 1;
        cbx1 >>= 1;
    }

    if (DIFF) {
        const bool packed = CB && c.shared.gDiffCheckerboard != 2u;
        const int dpx = packed ? px >> 1 : px;
        float4 diffuseIllumination = LoadRGBA16F(P.diff.in, dpx, py);
        float4 diffuseSH = SH ? LoadRGBA16F(P.diff.inSh, dpx, py) : F4(0.0f);
        if (packed && checkerboard != c.shared.gDiffCheckerboard) { // no data this frame: resolve from the two horizontal neighbours
            float2 wc = checkerboardResolveWeights;
            wc.x *= Cmp(CompareMaterials(centerMaterialID, materialID0, c.shared.gDiffMinMaterial));
            wc.y *= Cmp(CompareMaterials(centerMaterialID, materialID1, c.shared.gDiffMinMaterial));
            wc = wc * PositiveRcp(wc.x + wc.y);
            float4 d0 = Denanify(wc.x, LoadRGBA16F(P.diff.in, cbx0, py)), d1 = Denanify(wc.y, LoadRGBA16F(P.diff.in, cbx1, py));
            diffuseIllumination = d0 * wc.x + d1 * wc.y;
            if (SH) {
                float4 d0SH = Denanify(wc.x, LoadRGBA16F(P.diff.inSh, cbx0, py)), d1SH = Denanify(wc.y, LoadRGBA16F(P.diff.inSh, cbx1, py));
                diffuseSH = d0SH * wc.x + d1SH * wc.y;
            }
        }

        if (c.shared.gDiffBlurRadius > 0.0f) {
            float frustumSize = PixelRadiusToWorld(c.shared.gUnproject, c.shared.gOrthoMode, minRectDim, centerViewZ);
            float hitDist = diffuseIllumination.w == 0.0f ? 1.0f : diffuseIllumination.w;
            float hitDistFactor = GetHitDistFactor(hitDist, frustumSize);
            float blurRadius = c.shared.gDiffBlurRadius * hitDistFactor;
            if (diffuseIllumination.w == 0.0f)
                blurRadius = Max(blurRadius, 1.0f);

            float normalWeightParam = GetNormalWeightParam2(1.0f, 0.25f * c.shared.gLobeAngleFraction);
            float2 hitDistanceWeightParams = GetHitDistanceWeightParams(diffuseIllumination.w, 1.0f / 9.0f);
            float weightSum = 1.0f;

#pragma unroll NRD_RELAX_PREPASS_UNROLL
            for (int i = 0; i < 8; i++) {
                const PrePassGuides t = FetchPrePassGuides<CB, FR>(c, P, P.diff.in, c.shared.gDiffCheckerboard, pixelUv, rectSize, rotator, i, blurRadius);
                const float sampleMaterialID = t.materialID, sampleViewZ = t.viewZ;
                const float3 sampleNormal = t.normal, sampleWorldPos = t.worldPos;

                float sampleWeight = t.inScreen;
                sampleWeight *= Cmp(sampleViewZ < c.shared.gDenoisingRange);
                if (MAT && c.shared.gDiffMinMaterial < 3.0f) // material IDs are 0..3: a minimum >= 3 makes every comparison hold
                    sampleWeight *= Cmp(CompareMaterials(centerMaterialID, sampleMaterialID, c.shared.gDiffMinMaterial));
                sampleWeight *= GetPlaneDistanceWeight(centerWorldPos, centerNormal, centerViewZ, sampleWorldPos, c.shared.gDepthThreshold);
                float angle = AcosApprox(Dot(centerNormal, sampleNormal));
                sampleWeight *= ComputeWeight(angle, normalWeightParam, 0.0f);

                float4 sampleDiffuseIllumination = LoadDenanifiedRGBA16F(sampleWeight, P.diff.in, t.signalTexel.x, t.signalTexel.y);
                sampleWeight *= Lerp(c.shared.gMinHitDistanceWeight, 1.0f, ComputeExponentialWeight(sampleDiffuseIllumination.w, hitDistanceWeightParams.x, hitDistanceWeightParams.y));
                sampleWeight *= g_Poisson8Gaussian[i]; // GetGaussianWeight( offset.z )

                weightSum += sampleWeight;
                diffuseIllumination = Mad(sampleDiffuseIllumination, sampleWeight, diffuseIllumination);
                if (SH) {
                    float4 sampleDiffuseSH = LoadDenanifiedRGBA16F(sampleWeight, P.diff.inSh, t.signalTexel.x, t.signalTexel.y);
                    diffuseSH = Mad(sampleDiffuseSH, sampleWeight, diffuseSH);
                }
            }
            diffuseIllumination = Div(diffuseIllumination, weightSum);
            if (SH)
                diffuseSH = Div(diffuseSH, weightSum);
        }
        StoreRGBA16FHinted<NT>(P.diff.out, px, py, Clamp4(diffuseIllumination, 0.0f, NRD_FP16_MAX));
        if (SH)
            StoreRGBA16FHinted<NT>(P.diff.outSh, px, py, Clamp4(diffuseSH, -NRD_FP16_MAX, NRD_FP16_MAX));
    }

    if (SPEC) {
        const bool packed = CB && c.shared.gSpecCheckerboard != 2u;
        const int spx = packed ? px >> 1 : px;
        float4 specularIllumination = LoadRGBA16F(P.spec.in, spx, py);
        float4 specularSH = SH ? LoadRGBA16F(P.spec.inSh, spx, py) : F4(0.0f);
        if (packed && checkerboard != c.shared.gSpecCheckerboard) {
            float2 wc = checkerboardResolveWeights;
            wc.x *= Cmp(CompareMaterials(centerMaterialID, materialID0, c.shared.gSpecMinMaterial));
            wc.y *= Cmp(CompareMaterials(centerMaterialID, materialID1, c.shared.gSpecMinMaterial));
            wc = wc * PositiveRcp(wc.x + wc.y);
            float4 s0 = Denanify(wc.x, LoadRGBA16F(P.spec.in, cbx0, py)), s1 = Denanify(wc.y, LoadRGBA16F(P.spec.in, cbx1, py));
            specularIllumination = s0 * wc.x + s1 * wc.y;
            if (SH) {
                float4 s0SH = Denanify(wc.x, LoadRGBA16F(P.spec.inSh, cbx0, py)), s1SH = Denanify(wc.y, LoadRGBA16F(P.spec.inSh, cbx1, py));
                specularSH = s0SH * wc.x + s1SH * wc.y;
            }
        }
        specularIllumination.w = Max(0.0f, Min(c.shared.gDenoisingRange, specularIllumination.w));

        if (c.shared.gSpecBlurRadius > 0.0f) {
            float3 viewVector = Normalize(-centerWorldPos);
            float4 D = GetSpecularDominantDirection(centerNormal, viewVector, centerRoughness);
            float NoD = Abs(Dot(centerNormal, Xyz(D)));

            float frustumSize = PixelRadiusToWorld(c.shared.gUnproject, c.shared.gOrthoMode, minRectDim, centerViewZ);
            float hitDist = specularIllumination.w == 0.0f ? 1.0f : specularIllumination.w;
            float hitDistFactor = GetHitDistFactor(hitDist * NoD, frustumSize);

            float smc = GetSpecMagicCurve(centerRoughness);
            float blurRadius = c.shared.gSpecBlurRadius * hitDistFactor * smc;
            float lobeTanHalfAngle = GetSpecularLobeTanHalfAngle(centerRoughness, 0.75f);
            float lobeRadius = hitDist * NoD * lobeTanHalfAngle;
            float minBlurRadius = Div(lobeRadius, PixelRadiusToWorld(c.shared.gUnproject, c.shared.gOrthoMode, 1.0f, centerViewZ + hitDist * D.w));
            blurRadius = Min(blurRadius, minBlurRadius);
            if (specularIllumination.w == 0.0f)
                blurRadius = Max(blurRadius, 1.0f);

            float normalWeightParam = GetNormalWeightParam2(centerRoughness, 0.5f * c.shared.gLobeAngleFraction);
            float2 hitDistanceWeightParams = GetHitDistanceWeightParams(specularIllumination.w, 1.0f / 9.0f, centerRoughness);
            float2 roughnessWeightParams = GetRoughnessWeightParams(centerRoughness, c.shared.gRoughnessFraction);

            float specMinHitDistanceWeight = specularIllumination.w == 0.0f ? 1.0f : c.shared.gMinHitDistanceWeight * smc;
            float specularHitT = specularIllumination.w == 0.0f ? c.shared.gDenoisingRange : specularIllumination.w;
            float minHitT = specularHitT == 0.0f ? NRD_INF : specularHitT;
            float weightSum = 1.0f;
            float3 rgb = Xyz(specularIllumination);
            const float roughnessRelax = LinearStep(0.5f, 1.0f, centerRoughness);

#pragma unroll NRD_RELAX_PREPASS_UNROLL
            for (int i = 0; i < 8; i++) {
                const PrePassGuides t = FetchPrePassGuides<CB, FR>(c, P, P.spec.in, c.shared.gSpecCheckerboard, pixelUv, rectSize, rotator, i, blurRadius);
                const float sampleMaterialID = t.materialID, sampleViewZ = t.viewZ, sampleRoughness = t.roughness;
                const float3 sampleNormal = t.normal;

                float sampleWeight = t.inScreen;
                sampleWeight *= Cmp(sampleViewZ < c.shared.gDenoisingRange);
                if (MAT && c.shared.gSpecMinMaterial < 3.0f)
                    sampleWeight *= Cmp(CompareMaterials(centerMaterialID, sampleMaterialID, c.shared.gSpecMinMaterial));
                sampleWeight *= ComputeWeight(sampleRoughness, roughnessWeightParams.x, roughnessWeightParams.y);
                float angle = AcosApprox(Dot(centerNormal, sampleNormal));
                sampleWeight *= ComputeWeight(angle, normalWeightParam, 0.0f);

                const float3 sampleWorldPos = t.worldPos;
                sampleWeight *= GetPlaneDistanceWeight(centerWorldPos, centerNormal, centerViewZ, sampleWorldPos, c.shared.gDepthThreshold);

                float4 sampleSpecularIllumination = LoadDenanifiedRGBA16F(sampleWeight, P.spec.in, t.signalTexel.x, t.signalTexel.y);
                sampleWeight *= Lerp(specMinHitDistanceWeight, 1.0f, ComputeExponentialWeight(sampleSpecularIllumination.w, hitDistanceWeightParams.x, hitDistanceWeightParams.y));
                sampleWeight *= g_Poisson8Gaussian[i]; // GetGaussianWeight( offset.z )

                float d = Length(sampleWorldPos - centerWorldPos);
                float h = sampleSpecularIllumination.w;
                float tt = Div(h, specularIllumination.w + d);
                sampleWeight *= Lerp(Sat(tt), 1.0f, roughnessRelax);

                weightSum += sampleWeight;
                rgb = Mad(Xyz(sampleSpecularIllumination), sampleWeight, rgb);
                if (SH) {
                    float4 sampleSpecularSH = LoadDenanifiedRGBA16F(sampleWeight, P.spec.inSh, t.signalTexel.x, t.signalTexel.y);
                    specularSH = Mad(sampleSpecularSH, sampleWeight, specularSH);
                }
                if (sampleWeight != 0.0f)
                    minHitT = Min(minHitT, sampleSpecularIllumination.w == 0.0f ? NRD_INF : sampleSpecularIllumination.w);
            }
            rgb = Div(rgb, weightSum);
            specularIllumination = F4(rgb, minHitT == NRD_INF ? 0.0f : minHitT);
            if (SH)
                specularSH = Div(specularSH, weightSum);
        }
        StoreRGBA16FHinted<NT>(P.spec.out, px, py, Clamp4(specularIllumination, 0.0f, NRD_FP16_MAX));
        if (SH)
            StoreRGBA16FHinted<NT>(P.spec.outSh, px, py, Clamp4(specularSH, -NRD_FP16_MAX, NRD_FP16_MAX));
    }
}

template <bool DIFF, bool SPEC, bool SH>
const char* LaunchPrePass(const PassArgs& a) {
    if (const char* e = CheckSupportedRelax(a))
        return e;
    PlaneCursor cur(a);
    PrePassPlanes P = {};
    P.tiles = cur.next();
    if (SPEC) P.spec.in = cur.next();
    if (DIFF) P.diff.in = cur.next();
    P.normalRoughness = cur.next();
    P.viewZ = cur.next();
    if (SH && SPEC) P.spec.inSh = cur.next();
    if (SH && DIFF) P.diff.inSh = cur.next();
    if (SPEC) P.spec.out = cur.next();
    if (DIFF) P.diff.out = cur.next();
    if (SH && SPEC) P.spec.outSh = cur.next();
    if (SH && DIFF) P.diff.outSh = cur.next();
    P.decodedNR = a.decodedNormalRoughness;
    P.worldPos = a.worldPosViewZ;
    if (!cur.complete() || !P.decodedNR.ptr)
        return "RELAX PrePass: unexpected resource count or missing decoded normal/roughness cache";
    RelaxCB c = LoadRelaxConstants(a);
    RowGrid g = GridForRows(c.shared.gRectSize.x, c.shared.gRectSize.y, TILE_X, TILE_Y, a.rowBegin, a.rowEnd);
    const char* forceGeneric = getenv("NRD_HIP_GENERIC_TAPS");
    const bool fullRect = c.shared.gResolutionScale.x == 1.0f && c.shared.gResolutionScale.y == 1.0f && c.shared.gRectSize.x == P.decodedNR.w &&
                          c.shared.gRectSize.y == P.decodedNR.h && P.viewZ.w == P.decodedNR.w && P.viewZ.h == P.decodedNR.h && !(forceGeneric && atoi(forceGeneric) != 0);
    if ((SPEC && c.shared.gSpecCheckerboard != 2u) || (DIFF && c.shared.gDiffCheckerboard != 2u))
        LaunchPass(a, (RelaxPrePassKernel<DIFF, SPEC, SH, true, false>), g.grid, dim3(256), P, c, MakeRowRange(g));
    else if (fullRect && !(c.shared.gSpecMinMaterial < 3.0f || c.shared.gDiffMinMaterial < 3.0f) && (uint32_t)P.decodedNR.w * (uint32_t)P.decodedNR.h > NRD_NT_STORE_PIXELS)
        LaunchPass(a, (RelaxPrePassKernel<DIFF, SPEC, SH, false, true, false, true>), g.grid, dim3(256), P, c, MakeRowRange(g)); // (a frame larger than the caches: hinted stores)
    else if (fullRect && !(c.shared.gSpecMinMaterial < 3.0f || c.shared.gDiffMinMaterial < 3.0f))
        LaunchPass(a, (RelaxPrePassKernel<DIFF, SPEC, SH, false, true, false>), g.grid, dim3(256), P, c, MakeRowRange(g));
    else if (fullRect)
        LaunchPass(a, (RelaxPrePassKernel<DIFF, SPEC, SH, false, true>), g.grid, dim3(256), P, c, MakeRowRange(g));
    else
        LaunchPass(a, (RelaxPrePassKernel<DIFF, SPEC, SH, false, false>), g.grid, dim3(256), P, c, MakeRowRange(g));
    return nullptr;
}

// ================================================================================================ HistoryFix
struct HistoryFixPlanes {
    Plane tiles, historyLength, normalRoughness, viewZ;
    Plane decodedNR;
    SignalPlanes spec, diff;
};

#ifndef NRD_RELAX_HF_ROTATE
#define NRD_RELAX_HF_ROTATE 1
#endif
template <bool DIFF, bool SPEC, bool SH>
__global__ __launch_bounds__(256, NRD_WAVES_RELAX_HF) void RelaxHistoryFixKernel(HistoryFixPlanes P, RelaxCB c, RowRange rows) {
    const int blockY = BlockTileY(rows, false) /* (top-down: 0.0985 against 0.101 ms bottom-up, r04_v / r04_w) */;
    // (rotated tile order: the pixels with young history -- all this pass works on -- are the columns entering the screen and the silhouettes; passes.h BlockTileXRotated)
    const int px = (NRD_RELAX_HF_ROTATE ? BlockTileXRotated(rows, blockY) : BlockTileX(rows)) * TILE_X + (threadIdx.x & 31), py = blockY * TILE_Y + (threadIdx.x >> 5);
    const int rectW = c.shared.gRectSize.x, rectH = c.shared.gRectSize.y;
    if (px >= rectW || py >= rectH || py < rows.rowBegin || py >= rows.rowEnd)
        return;
    // the three loads that decide whether the pixel has anything to do, requested together (one memory latency instead of two: in the steady state nearly every
    // pixel leaves here, and the pass ran 60 % above the time of a build whose loads all hit the L1, profiles/r04_c_relax_ds_sh_uniform_*_kernel_stats.txt)
    const float tileFlag = LoadR8Unorm(P.tiles, px >> 4, py >> 4);
    float centerViewZ = RelaxUnpackViewZ(c, LoadR32F(P.viewZ, px, py));
    float historyLength = 255.0f * LoadR8Unorm(P.historyLength, px, py);
    if (tileFlag != 0.0f)
        return;
    if (centerViewZ > c.shared.gDenoisingRange || (historyLength > c.shared.gHistoryFixFrameNum || c.shared.gHistoryFixFrameNum == 1.0f))
        return;

    float centerMaterialID;
    float4 centerNormalRoughness = LoadDecodedNormalRoughness(P.decodedNR, px, py, centerMaterialID);
    float3 centerNormal = Xyz(centerNormalRoughness);
    float centerRoughness = centerNormalRoughness.w;
    float3 centerWorldPos = GetCurrentWorldPosFromPixelPos(c, px, py, centerViewZ);
    float3 centerV = -Normalize(centerWorldPos);
    float depthThreshold = c.shared.gDepthThreshold * centerViewZ;

    float4 diffuseSum = DIFF ? LoadRGBA16F(P.diff.in, px, py) : F4(0.0f);
    float4 diffuseSumSH = (DIFF && SH) ? LoadRGBA16F(P.diff.inSh, px, py) : F4(0.0f);
    float diffuseWSum = 1.0f;
    float4 specularSum = SPEC ? LoadRGBA16F(P.spec.in, px, py) : F4(0.0f);
    float4 specularSumSH = (SPEC && SH) ? LoadRGBA16F(P.spec.inSh, px, py) : F4(0.0f);
    float roughnessModified = specularSumSH.w;
    float specularWSum = 1.0f;
    float2 specularNormalWeightParams = SPEC ? GetNormalWeightParams_ATrous(centerRoughness, 5.0f, 1.0f, 0.0f, c.shared.gLobeAngleFraction, c.shared.gSpecLobeAngleSlack) : F2(0.0f, 0.0f);
    const float normalPower = Max(c.shared.gHistoryFixEdgeStoppingNormalPower, 0.01f);

    float r = Div(c.shared.gHistoryFixBasePixelStride, 1.0f + historyLength);
    r = floorf(r + 0.5f);

    for (int j = -2; j <= 2; j++)
        for (int i = -2; i <= 2; i++) {
            int dx = (int)(float(i) * r), dy = (int)(float(j) * r);
            int sx = px + dx, sy = py + dy;
            bool isInside = sx >= 0 && sy >= 0 && sx < rectW && sy < rectH;
            if (i == 0 && j == 0)
                continue;

            float sampleMaterialID;
            float3 sampleNormal = Xyz(LoadDecodedNormalRoughnessOrZero(P.decodedNR, sx, sy, sampleMaterialID));
            float sampleViewZ = RelaxUnpackViewZ(c, LoadR32FOrZero(P.viewZ, sx, sy));
            float3 sampleWorldPos = GetCurrentWorldPosFromPixelPos(c, sx, sy, sampleViewZ);
            float geometryWeight = GetPlaneDistanceWeight_Atrous(centerWorldPos, centerNormal, sampleWorldPos, depthThreshold);

            if (DIFF) {
                float diffuseW = geometryWeight;
                diffuseW *= Pow(Max(0.01f, Dot(centerNormal, sampleNormal)), normalPower);
                diffuseW = isInside ? diffuseW : 0.0f;
                diffuseW *= Cmp(CompareMaterials(sampleMaterialID, centerMaterialID, c.shared.gDiffMinMaterial));
                if (diffuseW > 1e-4f) {
                    diffuseSum = diffuseSum + LoadRGBA16FOrZero(P.diff.in, sx, sy) * diffuseW;
                    if (SH)
                        diffuseSumSH = diffuseSumSH + LoadRGBA16FOrZero(P.diff.inSh, sx, sy) * diffuseW;
                    diffuseWSum += diffuseW;
                }
            }
            if (SPEC) {
                float3 sampleV = -Normalize(sampleWorldPos + c.shared.gRoughnessEdgeStoppingRelaxation * centerWorldPos);
                float specularW = geometryWeight;
                specularW *= GetSpecularNormalWeight_ATrous(specularNormalWeightParams, centerNormal, sampleNormal, centerV, sampleV);
                specularW = isInside ? specularW : 0.0f;
                specularW *= Cmp(CompareMaterials(sampleMaterialID, centerMaterialID, c.shared.gSpecMinMaterial));
                if (specularW > 1e-4f) {
                    specularSum = specularSum + LoadRGBA16FOrZero(P.spec.in, sx, sy) * specularW;
                    if (SH)
                        specularSumSH = specularSumSH + LoadRGBA16FOrZero(P.spec.inSh, sx, sy) * specularW;
                    specularWSum += specularW;
                }
            }
        }

    if (DIFF) {
        StoreRGBA16F(P.diff.out, px, py, Div(diffuseSum, diffuseWSum));
        if (SH)
            StoreRGBA16F(P.diff.outSh, px, py, Div(diffuseSumSH, diffuseWSum));
    }
    if (SPEC) {
        StoreRGBA16F(P.spec.out, px, py, Div(specularSum, specularWSum));
        if (SH)
            StoreRGBA16F(P.spec.outSh, px, py, F4(Div(Xyz(specularSumSH), specularWSum), roughnessModified));
    }
}

template <bool DIFF, bool SPEC, bool SH>
const char* LaunchHistoryFix(const PassArgs& a) {
    if (const char* e = CheckSupportedRelax(a))
        return e;
    PlaneCursor cur(a);
    HistoryFixPlanes P = {};
    P.tiles = cur.next();
    if (SPEC) P.spec.in = cur.next();
    if (DIFF) P.diff.in = cur.next();
    P.historyLength = cur.next();
    P.normalRoughness = cur.next();
    P.viewZ = cur.next();
    if (SH && SPEC) P.spec.inSh = cur.next();
    if (SH && DIFF) P.diff.inSh = cur.next();
    if (SPEC) P.spec.out = cur.next();
    if (DIFF) P.diff.out = cur.next();
    if (SH && SPEC) P.spec.outSh = cur.next();
    if (SH && DIFF) P.diff.outSh = cur.next();
    P.decodedNR = a.decodedNormalRoughness;
    if (!cur.complete() || !P.decodedNR.ptr)
        return "RELAX HistoryFix: unexpected resource count or missing decoded normal/roughness cache";
    RelaxCB c = LoadRelaxConstants(a);
    RowGrid g = GridForRows(c.shared.gRectSize.x, c.shared.gRectSize.y, TILE_X, TILE_Y, a.rowBegin, a.rowEnd);
    LaunchPass(a, (RelaxHistoryFixKernel<DIFF, SPEC, SH>), g.grid, dim3(256), P, c, MakeRowRange(g));
    return nullptr;
}

// ================================================================================================ Copy / AntiFirefly
// Only dispatched with RelaxSettings::enableAntiFirefly: history -> user output planes, then a 3x3 rank-conditioned
// rank-selection filter back into the history (reference Shaders/Include/RELAX_Copy.hlsli, RELAX_AntiFirefly.hlsli).
struct CopyPlanes {
    SignalPlanes spec, diff;
};

template <bool DIFF, bool SPEC>
__global__ __launch_bounds__(256) void RelaxCopyKernel(CopyPlanes P, int gridW, int gridH) {
    const int px = blockIdx.x * TILE_X + (threadIdx.x & 31), py = blockIdx.y * TILE_Y + (threadIdx.x >> 5);
    if (px >= gridW || py >= gridH)
        return;
    if (SPEC && InBounds(P.spec.out, px, py))
        StoreRGBA16F(P.spec.out, px, py, LoadRGBA16FOrZero(P.spec.in, px, py));
    if (DIFF && InBounds(P.diff.out, px, py))
        StoreRGBA16F(P.diff.out, px, py, LoadRGBA16FOrZero(P.diff.in, px, py));
}

template <bool DIFF, bool SPEC>
const char* LaunchCopy(const PassArgs& a) {
    if (const char* e = CheckSupportedRelax(a))
        return e;
    PlaneCursor cur(a);
    CopyPlanes P = {};
    if (SPEC) P.spec.in = cur.next();
    if (DIFF) P.diff.in = cur.next();
    if (SPEC) P.spec.out = cur.next();
    if (DIFF) P.diff.out = cur.next();
    if (!cur.complete())
        return "RELAX Copy: unexpected resource count";
    RelaxCB c = LoadRelaxConstants(a);
    const int gridW = (c.shared.gRectSize.x + 7) & ~7, gridH = (c.shared.gRectSize.y + 7) & ~7; // the reference's 8x8 groups over the rect
    LaunchPass(a, (RelaxCopyKernel<DIFF, SPEC>), GridFor(gridW, gridH, TILE_X, TILE_Y), dim3(256), P, gridW, gridH);
    return nullptr;
}

struct AntiFireflyPlanes {
    Plane tiles, viewZ, decodedNR;
    SignalPlanes spec, diff;
};

template <bool IS_SPEC>
NRD_D void AntiFireflySignal(const RelaxCB& c, const SignalPlanes& S, const Plane& decodedNR, int px, int py, float centerMaterialID) {
    const int rectW = c.shared.gRectSize.x, rectH = c.shared.gRectSize.y;
    const float minMaterial = IS_SPEC ? c.shared.gSpecMinMaterial : c.shared.gDiffMinMaterial;
    const float4 center = LoadRGBA16F(S.in, px, py);
    const float centerLuminance = Luminance(Xyz(center));
    float maxLuminance = -1.0f, minLuminance = 1.0e6f;
    float3 maxValue = Xyz(center), minValue = Xyz(center);
#pragma unroll
    for (int yy = -1; yy <= 1; yy++)
#pragma unroll
        for (int xx = -1; xx <= 1; xx++) {
            const int qx = px + xx, qy = py + yy;
            if ((xx == 0 && yy == 0) || qx < 0 || qy < 0 || qx >= rectW || qy >= rectH)
                continue;
            const float3 value = Xyz(LoadRGBA16F(S.in, qx, qy));
            const float luminance = Luminance(value);
            float sampleMaterialID;
            LoadDecodedNormalRoughness(decodedNR, qx, qy, sampleMaterialID);
            if (CompareMaterials(sampleMaterialID, centerMaterialID, minMaterial)) {
                if (luminance > maxLuminance) {
                    maxLuminance = luminance;
                    maxValue = value;
                }
                if (luminance < minLuminance) {
                    minLuminance = luminance;
                    minValue = value;
                }
            }
        }
    float3 result = Xyz(center);
    if (centerLuminance > maxLuminance)
        result = maxValue;
    if (centerLuminance < minLuminance)
        result = minValue;
    StoreRGBA16F(S.out, px, py, F4(result, center.w));
}

template <bool DIFF, bool SPEC>
__global__ __launch_bounds__(256) void RelaxAntiFireflyKernel(AntiFireflyPlanes P, RelaxCB c, RowRange rows) {
    const int blockY = BlockTileY(rows, true);
    const int px = BlockTileX(rows) * TILE_X + (threadIdx.x & 31), py = blockY * TILE_Y + (threadIdx.x >> 5);
    if (px >= c.shared.gRectSize.x || py >= c.shared.gRectSize.y || py < rows.rowBegin || py >= rows.rowEnd)
        return;
    if (LoadR8Unorm(P.tiles, px >> 4, py >> 4) != 0.0f)
        return;
    if (RelaxUnpackViewZ(c, LoadR32F(P.viewZ, px, py)) > c.shared.gDenoisingRange)
        return;
    float centerMaterialID;
    LoadDecodedNormalRoughness(P.decodedNR, px, py, centerMaterialID);
    if (SPEC)
        AntiFireflySignal<true>(c, P.spec, P.decodedNR, px, py, centerMaterialID);
    if (DIFF)
        AntiFireflySignal<false>(c, P.diff, P.decodedNR, px, py, centerMaterialID);
}

template <bool DIFF, bool SPEC>
const char* LaunchAntiFirefly(const PassArgs& a) {
    if (const char* e = CheckSupportedRelax(a))
        return e;
    PlaneCursor cur(a);
    AntiFireflyPlanes P = {};
    P.tiles = cur.next();
    if (SPEC) P.spec.in = cur.next();
    if (DIFF) P.diff.in = cur.next();
    cur.next(); // packed normal / roughness: read through the decoded cache
    P.viewZ = cur.next();
    if (SPEC) P.spec.out = cur.next();
    if (DIFF) P.diff.out = cur.next();
    P.decodedNR = a.decodedNormalRoughness;
    if (!cur.complete() || !P.decodedNR.ptr)
        return "RELAX AntiFirefly: unexpected resource count or missing decoded normal/roughness cache";
    RelaxCB c = LoadRelaxConstants(a);
    RowGrid g = GridForRows(c.shared.gRectSize.x, c.shared.gRectSize.y, TILE_X, TILE_Y, a.rowBegin, a.rowEnd);
    LaunchPass(a, (RelaxAntiFireflyKernel<DIFF, SPEC>), g.grid, dim3(256), P, c, MakeRowRange(g));
    return nullptr;
}

// ================================================================================================ SplitScreen
struct SplitScreenPlanes {
    Plane viewZ;
    SignalPlanes spec, diff;
};

template <bool DIFF, bool SPEC, bool SH>
__global__ __launch_bounds__(256) void RelaxSplitScreenKernel(SplitScreenPlanes P, RelaxCB c, RowRange rows) {
    const int blockY = BlockTileY(rows, true);
    const int px = BlockTileX(rows) * TILE_X + (threadIdx.x & 31), py = blockY * TILE_Y + (threadIdx.x >> 5);
    if (px >= c.shared.gRectSize.x || py >= c.shared.gRectSize.y || py < rows.rowBegin || py >= rows.rowEnd)
        return;
    float2 pixelUv = F2(float(px) + 0.5f, float(py) + 0.5f) * ToF2(c.shared.gRectSizeInv);
    if (pixelUv.x > c.shared.gSplitScreen)
        return;
    float viewZ = RelaxUnpackViewZ(c, LoadR32F(P.viewZ, px, py));
    float keep = Cmp(viewZ < c.shared.gDenoisingRange);
    if (DIFF) {
        const int cx = c.shared.gDiffCheckerboard != 2u ? px >> 1 : px; // checkerboarded inputs live in the left half of the plane
        float4 v = LoadRGBA16F(P.diff.in, cx, py);
        if (SH)
            v = F4(LinearToYCoCg(Xyz(v)), v.w);
        StoreRGBA16F(P.diff.out, px, py, v * keep);
        if (SH)
            StoreRGBA16F(P.diff.outSh, px, py, LoadRGBA16F(P.diff.inSh, cx, py) * keep);
    }
    if (SPEC) {
        const int cx = c.shared.gSpecCheckerboard != 2u ? px >> 1 : px;
        float4 v = LoadRGBA16F(P.spec.in, cx, py);
        if (SH)
            v = F4(LinearToYCoCg(Xyz(v)), v.w);
        StoreRGBA16F(P.spec.out, px, py, v * keep);
        if (SH)
            StoreRGBA16F(P.spec.outSh, px, py, LoadRGBA16F(P.spec.inSh, cx, py) * keep);
    }
}

template <bool DIFF, bool SPEC, bool SH>
const char* LaunchSplitScreen(const PassArgs& a) {
    if (const char* e = CheckSupportedRelax(a))
        return e;
    PlaneCursor cur(a);
    SplitScreenPlanes P = {};
    P.viewZ = cur.next();
    if (DIFF) P.diff.in = cur.next();
    if (SPEC) P.spec.in = cur.next();
    if (SH && DIFF) P.diff.inSh = cur.next();
    if (SH && SPEC) P.spec.inSh = cur.next();
    if (DIFF) P.diff.out = cur.next();
    if (SPEC) P.spec.out = cur.next();
    if (SH && DIFF) P.diff.outSh = cur.next();
    if (SH && SPEC) P.spec.outSh = cur.next();
    if (!cur.complete())
        return "RELAX SplitScreen: unexpected resource count";
    RelaxCB c = LoadRelaxConstants(a);
    RowGrid g = GridForRows(c.shared.gRectSize.x, c.shared.gRectSize.y, TILE_X, TILE_Y, a.rowBegin, a.rowEnd);
    LaunchPass(a, (RelaxSplitScreenKernel<DIFF, SPEC, SH>), g.grid, dim3(256), P, c, MakeRowRange(g));
    return nullptr;
}

} // namespace

#define RELAX_SPATIAL_VARIANT(name, D, S, H)                            \
    {"RELAX_" name "_PrePass.cs", LaunchPrePass<D, S, H>},             \
    {"RELAX_" name "_HistoryFix.cs", LaunchHistoryFix<D, S, H>},       \
    {"RELAX_" name "_Copy.cs", LaunchCopy<D, S>},                      \
    {"RELAX_" name "_AntiFirefly.cs", LaunchAntiFirefly<D, S>},        \
    {"RELAX_" name "_SplitScreen.cs", LaunchSplitScreen<D, S, H>}

#define RELAX_HITDIST(name, D, S)                                                              \
    {"RELAX_" name "_HitDistReconstruction.cs", LaunchHitDistReconstruction<D, S, 1>},         \
    {"RELAX_" name "_HitDistReconstruction_5x5.cs", LaunchHitDistReconstruction<D, S, 2>}

const PassEntry* GetRelaxSpatialPasses(uint32_t& num) {
    static const PassEntry k[] = {
        {"RELAX_ClassifyTiles.cs", LaunchClassifyTiles},
        RELAX_HITDIST("Diffuse", true, false),
        RELAX_HITDIST("Specular", false, true),
        RELAX_HITDIST("DiffuseSpecular", true, true),
        RELAX_SPATIAL_VARIANT("Diffuse", true, false, false),
        RELAX_SPATIAL_VARIANT("DiffuseSh", true, false, true),
        RELAX_SPATIAL_VARIANT("Specular", false, true, false),
        RELAX_SPATIAL_VARIANT("SpecularSh", false, true, true),
        RELAX_SPATIAL_VARIANT("DiffuseSpecular", true, true, false),
        RELAX_SPATIAL_VARIANT("DiffuseSpecularSh", true, true, true),
    };
    num = sizeof(k) / sizeof(k[0]);
    return k;
}

} // namespace nrdhip
