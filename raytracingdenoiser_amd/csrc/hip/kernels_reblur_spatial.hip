// REBLUR spatial passes as HIP kernels for gfx950: ClassifyTiles, PrePass, Blur, PostBlur, SplitScreen.
//   ClassifyTiles   reference Shaders/Source/REBLUR_ClassifyTiles.cs.hlsl:20-55
//   PrePass         reference Shaders/Include/REBLUR_PrePass.hlsli:11-108
//   Blur            reference Shaders/Include/REBLUR_Blur.hlsli:11-74
//   PostBlur        reference Shaders/Include/REBLUR_PostBlur.hlsli:11-78
//   filters         reference Shaders/Include/REBLUR_Common_{Diffuse,Specular}SpatialFilter.hlsli
//   SplitScreen     reference Shaders/Include/REBLUR_SplitScreen.hlsli:11-46
//
// MI355X mapping. These are SPARSE GATHER passes (8 data-dependent taps per signal at radii up to 30-60 px), not
// stencils: LDS tiling would need (16+120)^2 halos, so taps are served by the 4 MiB L2 / 256 MiB Infinity Cache and the
// kernels are bound by the texel-request rate and by latency hiding, not by HBM streaming. What we do for the memory
// system: 32x8-pixel workgroups (one wave = 2 rows x 32 px, so the centre loads/stores of a wave are 2 x 256 B
// segments of an RGBA16F plane and 2 x 128 B of an R32F plane), the 832-byte constant block travels as a kernel
// argument (scalar loads, no constant-buffer indirection), sky tiles exit before touching any signal plane, and
// workgroups are issued in plain row-major order so neighbouring blocks share taps in L2.
// ClassifyTiles is one wave per 16x16 tile: each lane reads one 16-byte vector (4 px) and the tile verdict is a single
// wave-wide vote -- no LDS, no atomics.
#include "passes.h"
#include "reblur_device.h"

namespace nrdhip {

constexpr int TILE_X = 32;
constexpr int TILE_Y = 8;

// ================================================================================================ ClassifyTiles
__global__ __launch_bounds__(256) void ReblurClassifyTilesKernel(Plane viewZ, Plane tiles, float viewZScale, float denoisingRange, int tilesPerRow, int tileRows) {
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
        allSky = Abs(z.x * viewZScale) > denoisingRange && Abs(z.y * viewZScale) > denoisingRange && Abs(z.z * viewZScale) > denoisingRange && Abs(z.w * viewZScale) > denoisingRange;
    } else {
        for (int i = 0; i < 4; i++) {
            float z = InBounds(viewZ, x + i, y) ? LoadR32F(viewZ, x + i, y) : 0.0f; // out-of-bounds load = 0 => a partial edge tile is never sky
            allSky = allSky && Abs(z * viewZScale) > denoisingRange;
        }
    }
    bool tileIsSky = __all(allSky);
    if (lane == 0)
        StoreR8Unorm(tiles, tx, ty, tileIsSky ? 1.0f : 0.0f);
}

static const char* LaunchClassifyTiles(const PassArgs& a) {
    const ReblurCB& c = *(const ReblurCB*)a.constants;
    const Plane& tiles = a.planes[1];
    const int tilesPerRow = (c.gRectSizeMinusOne.x + 16) / 16, tileRows = (c.gRectSizeMinusOne.y + 16) / 16;
    if (tilesPerRow > tiles.w || tileRows > tiles.h)
        return "REBLUR ClassifyTiles: the rect does not fit the tile plane";
    if (a.fuseGuidesFrom.ptr) { // the frame's guide planes are due: one kernel decodes them and classifies (kernels_common.hip)
        LaunchDecodeGuidesClassify(a, a.planes[0], tiles, a.constants, tilesPerRow, tileRows);
        return nullptr;
    }
    int numTiles = tilesPerRow * tileRows;
    LaunchPass(a, ReblurClassifyTilesKernel, dim3((numTiles + 3) / 4), dim3(256), a.planes[0], tiles, c.gViewZScale, c.gDenoisingRange, tilesPerRow, tileRows);
    return nullptr;
}

// ================================================================================================ spatial filters
struct SpatialCtx {
    int px, py;
    float2 pixelUv;
    float viewZ, roughness, materialID, NoV, frustumSize;
    float3 N, Nv, Xv, Vv;
    float4 rotator;
    float2 data1;
    // plane distance of a tap, expanded once per pixel: every tap position is a pixel centre uv = (k + 0.5) * rectSizeInv and the perspective view position is
    // linear in uv, so dot(Nv, Xv(uv, z)) = z * (k.x * geo.x + k.y * geo.y + geo.z) -- 3 operations per tap instead of 13. The oracle states the same form.
    float3 geo;
    // checkerboard resolve of the pre-pass (reference REBLUR_PrePass.hlsli:43-56): neighbour columns in the half-width input + their weights
    int cbX0, cbX1;
    float2 wc;
};

// ---- one tap of the Poisson kernels: position -> texel, guides of that texel ------------------------------------------------------------------
// The reference snaps the tap to a pixel centre (floor(uv * rectSize) + 0.5), turns it back into a uv, scales / clamps it to the viewport, lets the
// nearest-clamp sampler pick the texel and reconstructs the tap's view position from that uv and the fetched viewZ. FR ("full rect": rect == resource,
// no checkerboard -- the launcher picks the variant) collapses that chain: the snapped pixel IS the texel (k = floor(uv * rectSize), clamped to the
// rect; "in screen" <=> k was inside). Same values bit for bit: a tap outside the screen has weight 0 in both formulations, inside the two agree
// on texel and uv.
struct TapGuides {
    float w;          // IsInScreenNearest
    float3 Ns, Xvs;   // world-space normal, view-space position of the tap's texel
    float NvXvs;      // dot(centre view-space normal, Xvs) (the argument of the plane-distance weight)
    float roughnessS, materialIDs, zs;
    int2 ts;          // texel of the signal planes
    float2 uv;        // generic path only
};

// the "full rect" tap: k = the snapped pixel (integer-valued floats, not yet clamped)
template <int FR, bool NEED_ROUGHNESS>
NRD_D TapGuides FetchTapGuidesFullRect(const ReblurCB& c, const SpatialCtx& s, float2 k, const NormalRoughnessGuide& gIn_Normal_Roughness, const Plane& gIn_ViewPos, bool compareMaterials) {
    TapGuides t;
    const float2 rectSize = ToF2(c.gRectSize), rectSizeInv = ToF2(c.gRectSizeInv);
    // clamp in the float domain (one v_med3_f32 per axis; the snapped coordinate is an integer-valued float): inside <=> the clamp changed nothing
    const float cxf = __builtin_amdgcn_fmed3f(k.x, 0.0f, rectSize.x - 1.0f), cyf = __builtin_amdgcn_fmed3f(k.y, 0.0f, rectSize.y - 1.0f);
    t.w = (cxf == k.x && cyf == k.y) ? 1.0f : 0.0f;
    t.ts = make_int2((int)cxf, (int)cyf);
    // Guides of the texel: the per-frame (normal, viewZ) guide plane (passes.h viewPos) makes a diffuse tap ONE 16-byte guide load; a specular tap adds the
    // 4 bytes that hold the roughness / material bits of the decoded-normal texel (from their compact copy, passes.h roughnessWord). The view position is re-derived from viewZ. Fetching it as a second
    // 16-byte guide texel instead (r02_b / r02_c A/B) saved 19 % of the instructions and bought nothing: the extra 12 bytes per tap through the L1 /
    // texture-address path cost as much (profiles/r02_c_gather_bench.txt prices a wave's 16-byte gather at 40-150 CU cycles, a 4-byte one at 6-40). Staging
    // the whole tap footprint in LDS was measured too (r02_d: 32x16 workgroups, halo 12, 76 KB) and lost 15-20 % to the fill and the lower occupancy:
    // neighbouring lanes' taps land on neighbouring texels, so the global path is better coalesced than a scattered gather.
    // FR == 2: no material test in this frame (the launcher checks the constants) -- the tap is straight-line code, so the compiler can overlap the
    // loads of neighbouring taps; a runtime test per tap (even a uniform one) ends the basic block and with it the scheduling window (r02_h: -17 % on
    // the three spatial passes, 95 -> 76 VGPRs).
    const bool materials = FR == 1 && compareMaterials;
    uint32_t bits = 0u;
    {
        const uint32_t offset = TexelOffset(gIn_ViewPos, t.ts.x, t.ts.y, 16u, true); // same layout as the decoded normals (launcher check)
        const float4 g = *(const float4*)(gIn_ViewPos.ptr + offset);
        t.Ns = Xyz(g);
        t.zs = g.w;
        if (NEED_ROUGHNESS || FR == 1)
            bits = *(const uint32_t*)(gIn_Normal_Roughness.word.ptr + (offset >> 2)); // (a quarter of the guide planes' pitch: reblur_device.h NormalRoughnessGuide)
    }
    const float2 uvc = (k + 0.5f) * rectSizeInv; // centre of the snapped pixel; equals the clamped texel's centre whenever the tap counts (t.w != 0)
    t.Xvs = ReconstructViewPosition(uvc, ToF4(c.gFrustum), t.zs, 0.0f); // perspective only (CheckSupported); dead code unless the caller needs the position
    t.NvXvs = t.zs * (k.x * s.geo.x + (k.y * s.geo.y + s.geo.z));
    t.materialIDs = 0.0f;
    if (materials)
        t.materialIDs = DecodedMaterialID(bits);
    t.roughnessS = NEED_ROUGHNESS ? DecodedRoughness(bits) : 0.0f;
    t.uv = k; // unused
    return t;
}

template <SpatialMode MODE, bool CB, int FR, bool NEED_ROUGHNESS>
NRD_D TapGuides FetchTapGuides(const ReblurCB& c, const SpatialCtx& s, float2 uv, const Plane& gIn_Signal, const Plane& gIn_ViewZ, const NormalRoughnessGuide& gIn_Normal_Roughness, const Plane& gIn_ViewPos,
    uint32_t checkerboardMode, uint32_t n, bool compareMaterials) {
    const float2 rectSize = ToF2(c.gRectSize), rectSizeInv = ToF2(c.gRectSizeInv);
    uv = Floor(uv * rectSize);
    if (FR)
        return FetchTapGuidesFullRect<FR, NEED_ROUGHNESS>(c, s, uv, gIn_Normal_Roughness, gIn_ViewPos, compareMaterials);
    TapGuides t;
    uv = uv + 0.5f;
    if (MODE == PRE_BLUR && CB)
        uv = ApplyCheckerboardShift(uv, checkerboardMode, n, c.gFrameIndex);
    const float2 k = uv - 0.5f; // the tap's pixel (exact)
    uv = uv * rectSizeInv;
    const float2 resolutionScale = ToF2(c.gResolutionScale);
    const float2 uvMax = resolutionScale - ToF2(c.gResourceSizeInv) * 0.5f;
    float2 uvScaled = F2(Min(uv.x * resolutionScale.x, uvMax.x), Min(uv.y * resolutionScale.y, uvMax.y));
    // all planes of a pass have the resource size (checked by the executor): one texel index serves the three fetches
    const int2 tz = NearestTexel(gIn_ViewZ, uvScaled);
    t.ts = tz;
    if (MODE == PRE_BLUR && CB && checkerboardMode != 2)
        t.ts = NearestTexel(gIn_Signal, F2(uvScaled.x * 0.5f, uvScaled.y));
    t.zs = UnpackViewZ(c, LoadR32F(gIn_ViewZ, tz.x, tz.y));
    float4 Ns = LoadDecodedNormalRoughness(gIn_Normal_Roughness, tz.x, tz.y, t.materialIDs);
    t.Ns = Xyz(Ns);
    t.roughnessS = Ns.w;
    t.Xvs = ReconstructViewPosition(uv, ToF4(c.gFrustum), t.zs, NRD_ORTHO_MODE(c));
    t.NvXvs = t.zs * (k.x * s.geo.x + (k.y * s.geo.y + s.geo.z));
    t.w = IsInScreenNearest(uv);
    t.uv = uv;
    return t;
}

// PERF = REBLUR_PERFORMANCE_MODE (reference REBLUR_Config.hlsli:196-238): 6 taps of g_Special6 instead of 8 of g_Special8, and
// screen-space sampling for the specular Blur / PostBlur too
template <bool PERF>
NRD_D float PoissonGaussianWeight(int n) { // = GetGaussianWeight( offset.z ), baked (reblur_device.h)
    if (PERF)
        return n < 3 ? REBLUR_GAUSSIAN_WEIGHT_Z1 : REBLUR_GAUSSIAN_WEIGHT_Z03;
    return n < 4 ? REBLUR_GAUSSIAN_WEIGHT_Z1 : REBLUR_GAUSSIAN_WEIGHT_Z05;
}

// OCC = occlusion family: the signal is the hit distance alone (REBLUR_TYPE float, R16_UNORM planes)
// SH = the *_SH denoisers: an SH1 plane (RGBA16F) rides on the same taps and weights (diffuse: all 4 components, specular: .xyz only)
// CB = a checkerboard mode is on (pre-pass only): the noisy inputs live in the left half of their planes, a tap that lands on a pixel
// without data moves one pixel sideways, and pixels the taps could not fill are resolved from the two horizontal neighbours
template <SpatialMode MODE, bool PERF, int KIND, bool SH, bool CB, int FR>
NRD_D typename ReblurSignal<KIND>::type DiffuseSpatialFilterTaps(const ReblurCB& c, const SpatialCtx& s, typename ReblurSignal<KIND>::type diff, const Plane& gIn_Diff, const Plane& gIn_ViewZ,
    const NormalRoughnessGuide& gIn_Normal_Roughness, const Plane& gIn_ViewPos, float4& diffSh, const Plane& gIn_DiffSh, float& sum) {
    typedef ReblurSignal<KIND> Sig;
    constexpr bool OCC = KIND == SIGNAL_OCCLUSION;
    typedef typename Sig::type S;

    float fractionScale = 1.0f, radiusScale = 1.0f;
    if (MODE == PRE_BLUR)
        fractionScale = REBLUR_PRE_BLUR_FRACTION_SCALE;
    else if (MODE == BLUR)
        fractionScale = REBLUR_BLUR_FRACTION_SCALE;
    else {
        radiusScale = REBLUR_POST_BLUR_RADIUS_SCALE;
        fractionScale = REBLUR_POST_BLUR_FRACTION_SCALE;
    }

    const float4 hitDistParams = ToF4(c.gHitDistParams);
    float hitDistScale = GetHitDistanceNormalization(s.viewZ, hitDistParams, 1.0f);
    float hitDist = ExtractHitDist(diff) * hitDistScale;
    float hitDistFactor = GetHitDistFactor(hitDist, s.frustumSize);

    float diffNonLinearAccumSpeed, blurRadius, areaFactor;
    if (MODE == PRE_BLUR) {
        diffNonLinearAccumSpeed = REBLUR_PRE_BLUR_NON_LINEAR_ACCUM_SPEED;
        blurRadius = c.gDiffPrepassBlurRadius;
        areaFactor = hitDistFactor;
    } else {
        float boost = 1.0f - GetFadeBasedOnAccumulatedFrames(c, s.data1.x);
        boost *= 1.0f - Pow5(s.NoV);
        diffNonLinearAccumSpeed = Rcp(1.0f + REBLUR_SAMPLES_PER_FRAME * (1.0f - boost) * s.data1.x);
        blurRadius = c.gMaxBlurRadius;
        areaFactor = hitDistFactor * diffNonLinearAccumSpeed;
    }
    blurRadius *= Sqrt01(areaFactor);
    blurRadius *= radiusScale;
    blurRadius = Max(blurRadius, c.gMinBlurRadius);

    float2 geometryWeightParams = GetGeometryWeightParams(c.gPlaneDistSensitivity, s.frustumSize, s.Xv, s.Nv);
    float normalWeightParam = Div(GetNormalWeightParam(diffNonLinearAccumSpeed, c.gLobeAngleFraction), fractionScale);
    float2 hitDistanceWeightParams = GetHitDistanceWeightParams(ExtractHitDist(diff), diffNonLinearAccumSpeed);
    float minHitDistWeight = c.gMinHitDistanceWeight * fractionScale;
    if (MODE != PRE_BLUR && !OCC)
        minHitDistWeight *= Sqrt(diffNonLinearAccumSpeed);

    const float2 rectSizeInv = ToF2(c.gRectSizeInv);

    float2 skew = F2(1.0f, 1.0f);
    if (MODE != PRE_BLUR) {
        skew = Lerp(F2(1.0f - Abs(s.Nv.x), 1.0f - Abs(s.Nv.y)), F2(1.0f, 1.0f), s.NoV);
        skew = Div(skew, Max(skew.x, skew.y));
    }
    skew = skew * (rectSizeInv * blurRadius);
    float4 scaledRotator = ScaleRotator(s.rotator, skew);
    // material IDs are 0..3: with a minimum material >= 3 every comparison holds (the library default is 4 = "off")
    const bool compareMaterials = FR != 2 && c.gDiffMinMaterial < 3.0f;

#pragma unroll
    for (int n = 0; n < (PERF ? 6 : 8); n++) {
        float3 offset = PERF ? F3(g_Special6[n][0], g_Special6[n][1], g_Special6[n][2]) : F3(g_Special8[n][0], g_Special8[n][1], g_Special8[n][2]);
        const float2 uvTap = s.pixelUv + RotateVector(scaledRotator, F2(offset.x, offset.y));
        TapGuides t = FetchTapGuides<MODE, CB, FR, false>(c, s, uvTap, gIn_Diff, gIn_ViewZ, gIn_Normal_Roughness, gIn_ViewPos, c.gDiffCheckerboard, (uint32_t)n, compareMaterials);
        const int2 ts = t.ts;

        float angle = AcosApprox(Dot(s.N, t.Ns));

        float w = t.w;
        w *= ComputeWeight(t.NvXvs, geometryWeightParams.x, geometryWeightParams.y);
        if (compareMaterials)
            w *= CompareMaterials(s.materialID, t.materialIDs, c.gDiffMinMaterial) ? 1.0f : 0.0f;
        w *= ComputeWeight(angle, normalWeightParam, 0.0f);

        S smp = Sig::LoadOrZero(gIn_Diff, ts.x, ts.y, w == 0.0f);

        w *= Lerp(minHitDistWeight, 1.0f, ComputeExponentialWeight(ExtractHitDist(smp), hitDistanceWeightParams.x, hitDistanceWeightParams.y));
        w *= PoissonGaussianWeight<PERF>(n);

        sum += w;
        diff = Mad(smp, w, diff);
        if (SH) {
            float4 sh = LoadRGBA16F(gIn_DiffSh, ts.x, ts.y);
            sh = Select(w == 0.0f, F4(0.0f), sh);
            diffSh = Mad(sh, w, diffSh);
        }
    }

    float invSum = PositiveRcp(sum);
    if (SH)
        diffSh = diffSh * invSum;
    return diff * invSum;
}

// "sum" = 1 when the centre pixel carries data, 0 for the empty pixels of a checkerboarded input
template <SpatialMode MODE, bool PERF, int KIND, bool SH, bool CB, int FR>
NRD_D typename ReblurSignal<KIND>::type DiffuseSpatialFilter(const ReblurCB& c, const SpatialCtx& s, typename ReblurSignal<KIND>::type diff, const Plane& gIn_Diff, const Plane& gIn_ViewZ,
    const NormalRoughnessGuide& gIn_Normal_Roughness, const Plane& gIn_ViewPos, float4& diffSh, const Plane& gIn_DiffSh, float sum) {
    typedef ReblurSignal<KIND> Sig;
    typedef typename Sig::type S;
    if (!(MODE == PRE_BLUR && c.gDiffPrepassBlurRadius == 0.0f))
        diff = DiffuseSpatialFilterTaps<MODE, PERF, KIND, SH, CB, FR>(c, s, diff, gIn_Diff, gIn_ViewZ, gIn_Normal_Roughness, gIn_ViewPos, diffSh, gIn_DiffSh, sum);
    if (MODE == PRE_BLUR && CB && sum == 0.0f) { // reference REBLUR_Common_DiffuseSpatialFilter.hlsli:177-199
        S s0 = Select(s.wc.x == 0.0f, Sig::Zero(), Sig::Load(gIn_Diff, s.cbX0, s.py));
        S s1 = Select(s.wc.y == 0.0f, Sig::Zero(), Sig::Load(gIn_Diff, s.cbX1, s.py));
        diff = s0 * s.wc.x + s1 * s.wc.y;
        if (SH) {
            float4 sh0 = Select(s.wc.x == 0.0f, F4(0.0f), LoadRGBA16F(gIn_DiffSh, s.cbX0, s.py));
            float4 sh1 = Select(s.wc.y == 0.0f, F4(0.0f), LoadRGBA16F(gIn_DiffSh, s.cbX1, s.py));
            diffSh = sh0 * s.wc.x + sh1 * s.wc.y;
        }
    }
    return diff;
}

template <SpatialMode MODE, bool PERF, int KIND, bool SH, bool CB, int FR>
NRD_D typename ReblurSignal<KIND>::type SpecularSpatialFilterTaps(const ReblurCB& c, const SpatialCtx& s, typename ReblurSignal<KIND>::type spec, const Plane& gIn_Spec, const Plane& gIn_ViewZ,
    const NormalRoughnessGuide& gIn_Normal_Roughness, const Plane& gIn_ViewPos, const Plane& gOut_SpecHitDistForTracking, float4& specSh, const Plane& gIn_SpecSh, float& sum) {
    typedef ReblurSignal<KIND> Sig;
    constexpr bool OCC = KIND == SIGNAL_OCCLUSION;
    typedef typename Sig::type S;
    float smc = GetSpecMagicCurve(s.roughness);

    RngHash rng;
    if (MODE == PRE_BLUR)
        rng.Initialize((uint32_t)s.px, (uint32_t)s.py, c.gFrameIndex);

    float fractionScale = 1.0f, radiusScale = 1.0f;
    if (MODE == PRE_BLUR)
        fractionScale = REBLUR_PRE_BLUR_FRACTION_SCALE;
    else if (MODE == BLUR)
        fractionScale = REBLUR_BLUR_FRACTION_SCALE;
    else {
        radiusScale = REBLUR_POST_BLUR_RADIUS_SCALE;
        fractionScale = REBLUR_POST_BLUR_FRACTION_SCALE;
    }

    const float4 hitDistParams = ToF4(c.gHitDistParams);
    float4 Dv = GetSpecularDominantDirection(s.Nv, s.Vv, s.roughness);
    float NoD = Abs(Dot(s.Nv, Xyz(Dv)));
    float hitDistScale = GetHitDistanceNormalization(s.viewZ, hitDistParams, s.roughness);
    float hitDist = ExtractHitDist(spec) * hitDistScale;
    float hitDistFactor = GetHitDistFactor(hitDist, s.frustumSize);

    float hitDistForTracking = 0.0f, specNonLinearAccumSpeed, blurRadius, areaFactor;
    if (MODE == PRE_BLUR) {
        specNonLinearAccumSpeed = REBLUR_PRE_BLUR_NON_LINEAR_ACCUM_SPEED;
        hitDistForTracking = hitDist == 0.0f ? NRD_INF : hitDist;
        blurRadius = c.gSpecPrepassBlurRadius;
        areaFactor = s.roughness * hitDistFactor;
    } else {
        float boost = 1.0f - GetFadeBasedOnAccumulatedFrames(c, s.data1.y);
        boost *= 1.0f - Pow5(s.NoV);
        boost *= smc;
        specNonLinearAccumSpeed = Rcp(1.0f + REBLUR_SAMPLES_PER_FRAME * (1.0f - boost) * s.data1.y);
        blurRadius = c.gMaxBlurRadius;
        areaFactor = s.roughness * hitDistFactor * specNonLinearAccumSpeed;
    }
    blurRadius *= Sqrt01(areaFactor);

    if (MODE == PRE_BLUR) {
        float lobeTanHalfAngle = GetSpecularLobeTanHalfAngle(s.roughness, REBLUR_MAX_PERCENT_OF_LOBE_VOLUME_FOR_PRE_PASS);
        float lobeRadius = hitDist * NoD * lobeTanHalfAngle;
        float minBlurRadius = Div(lobeRadius, PixelRadiusToWorld(c.gUnproject, NRD_ORTHO_MODE(c), 1.0f, s.viewZ + hitDist * Dv.w));
        blurRadius = Min(blurRadius, minBlurRadius);
    }
    blurRadius *= radiusScale;
    blurRadius = Max(blurRadius, c.gMinBlurRadius * smc);

    float roughnessFractionScaled = Sat(c.gRoughnessFraction * fractionScale);
    float2 geometryWeightParams = GetGeometryWeightParams(c.gPlaneDistSensitivity, s.frustumSize, s.Xv, s.Nv);
    float normalWeightParam = Div(GetNormalWeightParam(specNonLinearAccumSpeed, c.gLobeAngleFraction, s.roughness), fractionScale);
    float2 roughnessWeightParams = GetRoughnessWeightParams(s.roughness, roughnessFractionScaled);
    float2 hitDistanceWeightParams = GetHitDistanceWeightParams(ExtractHitDist(spec), specNonLinearAccumSpeed, s.roughness);
    float minHitDistWeight = c.gMinHitDistanceWeight * fractionScale * smc;
    if (MODE != PRE_BLUR && !OCC)
        minHitDistWeight *= Sqrt(specNonLinearAccumSpeed);

    const float2 rectSizeInv = ToF2(c.gRectSizeInv);
    const bool compareMaterials = FR != 2 && c.gSpecMinMaterial < 3.0f; // see the diffuse filter

    constexpr bool SCREEN_SPACE = MODE == PRE_BLUR || PERF; // REBLUR_USE_SCREEN_SPACE_SAMPLING_FOR_SPECULAR
    float4 scaledRotator = F4(0.0f);
    float3 T = F3(0.0f), B = F3(0.0f);
    if (SCREEN_SPACE) {
        float2 skew = rectSizeInv * blurRadius;
        scaledRotator = ScaleRotator(s.rotator, skew);
    } else {
        float bentFactor = Sqrt(hitDistFactor);
        float skewFactor = Lerp(0.25f + 0.75f * s.roughness, 1.0f, NoD);
        skewFactor = Lerp(skewFactor, 1.0f, specNonLinearAccumSpeed);
        skewFactor = Lerp(1.0f, skewFactor, bentFactor);
        float3 bentDv = Normalize(Lerp(s.Nv, Xyz(Dv), bentFactor));
        GetKernelBasis(bentDv, s.Nv, T, B);
        float worldRadius = PixelRadiusToWorld(c.gUnproject, NRD_ORTHO_MODE(c), blurRadius, s.viewZ);
        T = T * (worldRadius * skewFactor);
        B = B * (Div(worldRadius, skewFactor));
    }
    KernelProjection kernelProjection = {};
    if (!SCREEN_SPACE)
        kernelProjection = MakeKernelProjection(c.gViewToClip, s.Xv, T, B, s.rotator);

#pragma unroll
    for (int n = 0; n < (PERF ? 6 : 8); n++) {
        float3 offset = PERF ? F3(g_Special6[n][0], g_Special6[n][1], g_Special6[n][2]) : F3(g_Special8[n][0], g_Special8[n][1], g_Special8[n][2]);
        float2 uv;
        if (SCREEN_SPACE)
            uv = s.pixelUv + RotateVector(scaledRotator, F2(offset.x, offset.y));
        else
            uv = KernelSampleUv(kernelProjection, offset.x, offset.y);
        TapGuides t = FetchTapGuides<MODE, CB, FR, true>(c, s, uv, gIn_Spec, gIn_ViewZ, gIn_Normal_Roughness, gIn_ViewPos, c.gSpecCheckerboard, (uint32_t)n, compareMaterials);
        const int2 ts = t.ts;
        const float zs = t.zs;
        const float3 Xvs = t.Xvs;
        const float4 Ns = F4(t.Ns, t.roughnessS);

        float angle = AcosApprox(Dot(s.N, t.Ns));

        float w = t.w;
        w *= ComputeWeight(t.NvXvs, geometryWeightParams.x, geometryWeightParams.y);
        if (compareMaterials)
            w *= CompareMaterials(s.materialID, t.materialIDs, c.gSpecMinMaterial) ? 1.0f : 0.0f;
        w *= ComputeWeight(angle, normalWeightParam, 0.0f);
        w *= ComputeWeight(Ns.w, roughnessWeightParams.x, roughnessWeightParams.y);

        S smp = Sig::LoadOrZero(gIn_Spec, ts.x, ts.y, w == 0.0f);

        if (MODE == PRE_BLUR) {
            float hs = ExtractHitDist(smp) * GetHitDistanceNormalization(zs, hitDistParams, Ns.w);
            float d = Length(Xvs - s.Xv) + NRD_EPS;
            float geometryWeight = w * Sat(Div(hs, d));
            if (rng.GetFloat() < geometryWeight)
                hitDistForTracking = Min(hitDistForTracking, hs);

            w *= c.gUsePrepassNotOnlyForSpecularMotionEstimation;

            float t = Div(hs, d + hitDist);
            w *= Lerp(Sat(t), 1.0f, LinearStep(0.5f, 1.0f, s.roughness));
        }
        w *= Lerp(minHitDistWeight, 1.0f, ComputeExponentialWeight(ExtractHitDist(smp), hitDistanceWeightParams.x, hitDistanceWeightParams.y));
        w *= PoissonGaussianWeight<PERF>(n);

        sum += w;
        spec = Mad(smp, w, spec);
        if (SH) {
            float4 sh = LoadRGBA16F(gIn_SpecSh, ts.x, ts.y);
            sh = Select(w == 0.0f, F4(0.0f), sh);
            specSh.x += sh.x * w, specSh.y += sh.y * w, specSh.z += sh.z * w;
        }
    }

    float invSum = PositiveRcp(sum);
    spec = spec * invSum;
    if (SH)
        specSh.x *= invSum, specSh.y *= invSum, specSh.z *= invSum;

    if (MODE == PRE_BLUR)
        StoreR16F(gOut_SpecHitDistForTracking, s.px, s.py, hitDistForTracking == NRD_INF ? 0.0f : hitDistForTracking);
    return spec;
}

template <SpatialMode MODE, bool PERF, int KIND, bool SH, bool CB, int FR>
NRD_D typename ReblurSignal<KIND>::type SpecularSpatialFilter(const ReblurCB& c, const SpatialCtx& s, typename ReblurSignal<KIND>::type spec, const Plane& gIn_Spec, const Plane& gIn_ViewZ,
    const NormalRoughnessGuide& gIn_Normal_Roughness, const Plane& gIn_ViewPos, const Plane& gOut_SpecHitDistForTracking, float4& specSh, const Plane& gIn_SpecSh, float sum) {
    typedef ReblurSignal<KIND> Sig;
    typedef typename Sig::type S;
    if (!(MODE == PRE_BLUR && c.gSpecPrepassBlurRadius == 0.0f))
        spec = SpecularSpatialFilterTaps<MODE, PERF, KIND, SH, CB, FR>(c, s, spec, gIn_Spec, gIn_ViewZ, gIn_Normal_Roughness, gIn_ViewPos, gOut_SpecHitDistForTracking, specSh, gIn_SpecSh, sum);
    if (MODE == PRE_BLUR && CB && sum == 0.0f) { // reference REBLUR_Common_SpecularSpatialFilter.hlsli:224-246 (all 4 SH components here)
        S s0 = Select(s.wc.x == 0.0f, Sig::Zero(), Sig::Load(gIn_Spec, s.cbX0, s.py));
        S s1 = Select(s.wc.y == 0.0f, Sig::Zero(), Sig::Load(gIn_Spec, s.cbX1, s.py));
        spec = s0 * s.wc.x + s1 * s.wc.y;
        if (SH) {
            float4 sh0 = Select(s.wc.x == 0.0f, F4(0.0f), LoadRGBA16F(gIn_SpecSh, s.cbX0, s.py));
            float4 sh1 = Select(s.wc.y == 0.0f, F4(0.0f), LoadRGBA16F(gIn_SpecSh, s.cbX1, s.py));
            specSh = sh0 * s.wc.x + sh1 * s.wc.y;
        }
    }
    return spec;
}

// per-pixel context; false = early-out (sky tile, outside the rect, beyond the denoising range)
NRD_D bool MakeSpatialCtx(const ReblurCB& c, int px, int py, float viewZ, const NormalRoughnessGuide& gIn_Normal_Roughness, float4 rotator, SpatialCtx& s) {
    s.viewZ = viewZ;
    if (viewZ > c.gDenoisingRange)
        return false;
    float4 normalAndRoughness = LoadDecodedNormalRoughness(gIn_Normal_Roughness, px, py, s.materialID);
    s.px = px;
    s.py = py;
    s.N = Xyz(normalAndRoughness);
    s.Nv = RotateVectorInverse(c.gViewToWorld, s.N);
    s.roughness = normalAndRoughness.w;
    s.pixelUv = F2(float(px) + 0.5f, float(py) + 0.5f) * ToF2(c.gRectSizeInv);
    s.Xv = ReconstructViewPosition(s.pixelUv, ToF4(c.gFrustum), viewZ, NRD_ORTHO_MODE(c));
    s.Vv = GetViewVector(c, s.Xv, true);
    s.NoV = Abs(Dot(s.Nv, s.Vv));
    s.frustumSize = GetFrustumSize(c.gMinRectDimMulUnproject, NRD_ORTHO_MODE(c), viewZ);
    s.rotator = rotator;
    s.data1 = F2(0.0f, 0.0f);
    {
        const float4 f = ToF4(c.gFrustum);
        const float2 r = ToF2(c.gRectSizeInv);
        s.geo = F3(s.Nv.x * f.z * r.x, s.Nv.y * f.w * r.y, s.Nv.x * (0.5f * r.x * f.z + f.x) + s.Nv.y * (0.5f * r.y * f.w + f.y) + s.Nv.z);
    }
    return true;
}


struct SpatialPlanes {
    Plane tiles, normalRoughness, viewZ, data1;
    NormalRoughnessGuide decodedNR; // executor's decoded guides: float4 (normal, viewZ) + the roughness | material word (reblur_device.h NormalRoughnessGuide)
    Plane viewPos;                  // = decodedNR.nz: what a tap reads in one 16-byte load
    Plane inDiff, inSpec;
    Plane outDiff, outSpec;
    Plane outHitDistForTracking; // pre-pass
    Plane outViewZ;              // blur
    Plane outNormalRoughness;    // post-blur
    Plane outInternalData, outDiffCopy, outSpecCopy; // post-blur without temporal stabilization
    Plane inDiffSh, inSpecSh, outDiffSh, outSpecSh, outDiffShCopy, outSpecShCopy; // SH family
};

// FR: 0 = generic taps; 1 = rect == resource, no checkerboard, (normal, viewZ) guide plane present; 2 = 1 without material tests (FetchTapGuides)
template <SpatialMode MODE, bool DIFF, bool SPEC, bool NO_TS, bool PERF, int KIND, bool SH, bool CB, int FR>
__global__ __launch_bounds__(TILE_X* TILE_Y, NRD_WAVES_REBLUR_SPATIAL) void ReblurSpatialKernel(ReblurCB c, SpatialPlanes P, RowRange rr) {
    typedef ReblurSignal<KIND> Sig;
    constexpr bool OCC = KIND == SIGNAL_OCCLUSION;
    typedef typename Sig::type S;
    const int px = BlockTileX(rr) * TILE_X + (threadIdx.x % TILE_X);
    // NRD_ALT_TILE_ORDER (A/B, round 6): the passes of the chain alternate their walk through the tile rows as the reference's NRD_CTA_ORDER_DEFAULT / _REVERSED do
    // (Common.hlsli:92-106: "helps to reuse data already stored in caches") -- PrePass top-down, TemporalAccumulation bottom-up, HistoryFix top-down, Blur bottom-up, PostBlur
    // top-down, TemporalStabilization bottom-up: every pass starts on the rows its predecessor wrote last
    const int py = (BlockTileY(rr, NRD_ALT_TILE_ORDER && MODE == BLUR)) * TILE_Y + (threadIdx.x / TILE_X);
    if (px > c.gRectSizeMinusOne.x || py > c.gRectSizeMinusOne.y || py < rr.rowBegin || py >= rr.rowEnd)
        return;
    if (LoadR8Unorm(P.tiles, px >> 4, py >> 4) != 0.0f)
        return;

    const float viewZpacked = LoadR32F(P.viewZ, px, py);
    if (MODE == BLUR)
        StoreR32F(P.outViewZ, px, py, viewZpacked); // PREV_VIEWZ, before the denoising-range early-out

    SpatialCtx s;
    const nrdc::F4 rot = MODE == PRE_BLUR ? c.gRotatorPre : (MODE == BLUR ? c.gRotator : c.gRotatorPost);
    if (!MakeSpatialCtx(c, px, py, UnpackViewZ(c, viewZpacked), P.decodedNR, ToF4(rot), s))
        return;

    if (MODE != PRE_BLUR)
        s.data1 = LoadData1<DIFF, SPEC>(P.data1, px, py);

    uint32_t checkerboard = 0;
    if (MODE == PRE_BLUR && CB) { // checkerboard resolve weights (reference REBLUR_PrePass.hlsli:43-56)
        checkerboard = CheckerBoard((uint32_t)px, (uint32_t)py, c.gFrameIndex);
        const int x0 = px > 0 ? px - 1 : 0, x1 = px < c.gRectSizeMinusOne.x ? px + 1 : c.gRectSizeMinusOne.x;
        const float viewZ0 = UnpackViewZ(c, LoadR32F(P.viewZ, x0, py)), viewZ1 = UnpackViewZ(c, LoadR32F(P.viewZ, x1, py));
        const float thr = GetDisocclusionThreshold(NRD_DISOCCLUSION_THRESHOLD, s.frustumSize, s.NoV);
        float2 wc = F2(thr >= Abs(viewZ0 - s.viewZ) ? 1.0f : 0.0f, thr >= Abs(viewZ1 - s.viewZ) ? 1.0f : 0.0f);
        wc.x = (viewZ0 > c.gDenoisingRange || px < 1) ? 0.0f : wc.x;
        wc.y = (viewZ1 > c.gDenoisingRange || px >= c.gRectSizeMinusOne.x) ? 0.0f : wc.y;
        s.wc = wc * PositiveRcp(wc.x + wc.y);
        s.cbX0 = x0 >> 1;
        s.cbX1 = x1 >> 1;
    }

    if (MODE == POST_BLUR) {
        StoreNrRaw(P.outNormalRoughness, px, py, InToPrevNormalRoughnessTexel(LoadNrRaw(P.normalRoughness, px, py))); // the texel as read (reblur_device.h: a copy of the bits, fp16 for encoding 4)
        if (NO_TS)
            StoreR16U(P.outInternalData, px, py, PackInternalData(s.data1.x + 1.0f, s.data1.y + 1.0f, s.materialID));
    }

    if (DIFF) {
        const int pos = (MODE == PRE_BLUR && CB && c.gDiffCheckerboard != 2) ? px >> 1 : px;
        float sum = 1.0f;
        S diff = Sig::Load(P.inDiff, pos, py);
        float4 diffSh = F4(0.0f);
        if (SH)
            diffSh = LoadRGBA16F(P.inDiffSh, pos, py);
        if (MODE == PRE_BLUR && CB && c.gDiffCheckerboard != 2 && checkerboard != c.gDiffCheckerboard) {
            sum = 0.0f;
            diff = Sig::Zero();
            diffSh = F4(0.0f);
        }
        diff = DiffuseSpatialFilter<MODE, PERF, KIND, SH, CB, FR>(c, s, diff, P.inDiff, P.viewZ, P.decodedNR, P.viewPos, diffSh, P.inDiffSh, sum);
        Sig::Store(P.outDiff, px, py, diff);
        if (SH)
            StoreRGBA16F(P.outDiffSh, px, py, diffSh);
        if (MODE == POST_BLUR && NO_TS && !OCC) // the occlusion family has no separate history copy: its output is the history
            Sig::Store(P.outDiffCopy, px, py, diff);
        if (MODE == POST_BLUR && NO_TS && SH)
            StoreRGBA16F(P.outDiffShCopy, px, py, diffSh);
    }
    if (SPEC) {
        const int pos = (MODE == PRE_BLUR && CB && c.gSpecCheckerboard != 2) ? px >> 1 : px;
        float sum = 1.0f;
        S spec = Sig::Load(P.inSpec, pos, py);
        float4 specSh = F4(0.0f);
        if (SH)
            specSh = LoadRGBA16F(P.inSpecSh, pos, py);
        if (MODE == PRE_BLUR && CB && c.gSpecCheckerboard != 2 && checkerboard != c.gSpecCheckerboard) {
            sum = 0.0f;
            spec = Sig::Zero();
            specSh = F4(0.0f);
        }
        spec = SpecularSpatialFilter<MODE, PERF, KIND, SH, CB, FR>(c, s, spec, P.inSpec, P.viewZ, P.decodedNR, P.viewPos, P.outHitDistForTracking, specSh, P.inSpecSh, sum);
        Sig::Store(P.outSpec, px, py, spec);
        if (SH)
            StoreRGBA16F(P.outSpecSh, px, py, specSh);
        if (MODE == POST_BLUR && NO_TS && !OCC)
            Sig::Store(P.outSpecCopy, px, py, spec);
        if (MODE == POST_BLUR && NO_TS && SH)
            StoreRGBA16F(P.outSpecShCopy, px, py, specSh);
    }
}

// NRD_HIP_GENERIC_TAPS=1: never take the "full rect" variant (A/B runs and the test that holds the two variants against each other)
static bool ForceGenericTaps() {
    const char* v = getenv("NRD_HIP_GENERIC_TAPS");
    return v && atoi(v) != 0;
}

static const char* CheckSupported(const ReblurCB& c) {
    if (c.gRectOrigin.x != 0 || c.gRectOrigin.y != 0) // the executor moves the rect of the guide inputs to (0, 0) and zeroes this field (executor.hip "shifted rect")
        return "internal error: a pass was handed a non-zero rectOrigin";
    if (c.gOrthoMode != 0.0f)
        return "REBLUR: orthographic projection is not supported (SURVEY.md section 8c)";
    return nullptr;
}

template <SpatialMode MODE, bool DIFF, bool SPEC, bool NO_TS, bool PERF, int KIND, bool SH>
static const char* LaunchSpatial(const PassArgs& a) {
    constexpr bool OCC = KIND == SIGNAL_OCCLUSION;
    const ReblurCB& c = *(const ReblurCB*)a.constants;
    if (const char* err = CheckSupported(c))
        return err;

    SpatialPlanes P = {};
    uint32_t k = 0;
    P.tiles = a.planes[k++];
    P.normalRoughness = a.planes[k++];
    if (const char* err = MakeNormalRoughnessGuide(a, P.decodedNR))
        return err;
    P.viewPos = P.decodedNR.nz;
    if (MODE == PRE_BLUR) {
        P.viewZ = a.planes[k++];
        if (DIFF) P.inDiff = a.planes[k++];
        if (SPEC) P.inSpec = a.planes[k++];
        if (DIFF && SH) P.inDiffSh = a.planes[k++];
        if (SPEC && SH) P.inSpecSh = a.planes[k++];
        if (DIFF) P.outDiff = a.planes[k++];
        if (SPEC) P.outSpec = a.planes[k++];
        if (SPEC) P.outHitDistForTracking = a.planes[k++];
        if (DIFF && SH) P.outDiffSh = a.planes[k++];
        if (SPEC && SH) P.outSpecSh = a.planes[k++];
    } else {
        P.data1 = a.planes[k++];
        if (DIFF) P.inDiff = a.planes[k++];
        if (SPEC) P.inSpec = a.planes[k++];
        P.viewZ = a.planes[k++];
        if (DIFF && SH) P.inDiffSh = a.planes[k++];
        if (SPEC && SH) P.inSpecSh = a.planes[k++];
        if (MODE == BLUR) {
            if (DIFF) P.outDiff = a.planes[k++];
            if (SPEC) P.outSpec = a.planes[k++];
            P.outViewZ = a.planes[k++];
            if (DIFF && SH) P.outDiffSh = a.planes[k++];
            if (SPEC && SH) P.outSpecSh = a.planes[k++];
        } else {
            P.outNormalRoughness = a.planes[k++];
            if (DIFF) P.outDiff = a.planes[k++];
            if (SPEC) P.outSpec = a.planes[k++];
            if (NO_TS) {
                P.outInternalData = a.planes[k++];
                if (DIFF && !OCC) P.outDiffCopy = a.planes[k++];
                if (SPEC && !OCC) P.outSpecCopy = a.planes[k++];
                if (DIFF && SH) P.outDiffShCopy = a.planes[k++];
                if (SPEC && SH) P.outSpecShCopy = a.planes[k++];
            }
            if (DIFF && SH) P.outDiffSh = a.planes[k++];
            if (SPEC && SH) P.outSpecSh = a.planes[k++];
        }
    }
    if (k != a.planesNum)
        return "REBLUR spatial pass: unexpected resource count";

    RowGrid g = GridForRows(c.gRectSizeMinusOne.x + 1, c.gRectSizeMinusOne.y + 1, TILE_X, TILE_Y, a.rowBegin, a.rowEnd);
    const RowRange rows = MakeRowRange(g);
    if constexpr (MODE == PRE_BLUR) { // only the pre-pass reads the (possibly checkerboarded) noisy inputs
        if (c.gDiffCheckerboard != 2 || c.gSpecCheckerboard != 2) {
            LaunchPass(a, (ReblurSpatialKernel<MODE, DIFF, SPEC, NO_TS, PERF, KIND, SH, true, 0>), g.grid, dim3(TILE_X * TILE_Y), c, P, rows);
            return nullptr;
        }
    }
    // "full rect": the rect is the whole resource (no dynamic-resolution scaling) and the (normal, viewZ) guide plane of this frame exists;
    // variant 2 when neither signal tests material IDs this frame (IDs are 0..3: a minimum >= 3, the library default, makes every comparison hold)
    const bool fullRect = c.gResolutionScale.x == 1.0f && c.gResolutionScale.y == 1.0f && c.gRectSizeMinusOne.x + 1 == P.decodedNR.nz.w &&
                          c.gRectSizeMinusOne.y + 1 == P.decodedNR.nz.h && P.viewZ.w == P.decodedNR.nz.w && P.viewZ.h == P.decodedNR.nz.h && !ForceGenericTaps();
    const bool materials = (DIFF && c.gDiffMinMaterial < 3.0f) || (SPEC && c.gSpecMinMaterial < 3.0f);
    if (fullRect && !materials)
        LaunchPass(a, (ReblurSpatialKernel<MODE, DIFF, SPEC, NO_TS, PERF, KIND, SH, false, 2>), g.grid, dim3(TILE_X * TILE_Y), c, P, rows);
    else if (fullRect)
        LaunchPass(a, (ReblurSpatialKernel<MODE, DIFF, SPEC, NO_TS, PERF, KIND, SH, false, 1>), g.grid, dim3(TILE_X * TILE_Y), c, P, rows);
    else
        LaunchPass(a, (ReblurSpatialKernel<MODE, DIFF, SPEC, NO_TS, PERF, KIND, SH, false, 0>), g.grid, dim3(TILE_X * TILE_Y), c, P, rows);
    return nullptr;
}

// ================================================================================================ SplitScreen
// OCC: the occlusion family binds its R16_UNORM planes to the radiance family's split-screen pipeline (the shader only scales .x there)
template <bool DIFF, bool SPEC, int KIND>
__global__ __launch_bounds__(256) void ReblurSplitScreenKernel(ReblurCB c, Plane viewZ, Plane inDiff, Plane inSpec, Plane outDiff, Plane outSpec, Plane inDiffSh, Plane inSpecSh, Plane outDiffSh,
    Plane outSpecSh, RowRange rr) {
    const int px = BlockTileX(rr) * TILE_X + (threadIdx.x % TILE_X);
    const int py = (BlockTileY(rr)) * TILE_Y + (threadIdx.x / TILE_X);
    if (px > c.gRectSizeMinusOne.x || py > c.gRectSizeMinusOne.y || py < rr.rowBegin || py >= rr.rowEnd)
        return;
    float pixelUvX = (float(px) + 0.5f) * c.gRectSizeInv.x;
    if (pixelUvX > c.gSplitScreen)
        return;
    float z = UnpackViewZ(c, LoadR32F(viewZ, px, py));
    float keep = z < c.gDenoisingRange ? 1.0f : 0.0f;
    typedef ReblurSignal<KIND> Sig;
    const int dx = c.gDiffCheckerboard != 2 ? px >> 1 : px, sx = c.gSpecCheckerboard != 2 ? px >> 1 : px; // checkerboarded inputs: left half of the plane
    if (DIFF)
        Sig::Store(outDiff, px, py, Sig::Load(inDiff, dx, py) * keep);
    if (SPEC)
        Sig::Store(outSpec, px, py, Sig::Load(inSpec, sx, py) * keep);
    if (DIFF && inDiffSh.ptr) // SH family (uniform branch)
        StoreRGBA16F(outDiffSh, px, py, LoadRGBA16F(inDiffSh, dx, py) * keep);
    if (SPEC && inSpecSh.ptr)
        StoreRGBA16F(outSpecSh, px, py, LoadRGBA16F(inSpecSh, sx, py) * keep);
}

template <bool DIFF, bool SPEC, bool SH>
static const char* LaunchSplitScreen(const PassArgs& a) {
    const ReblurCB& c = *(const ReblurCB*)a.constants;
    if (const char* err = CheckSupported(c))
        return err;
    uint32_t k = 0;
    Plane viewZ = a.planes[k++], inDiff = {}, inSpec = {}, outDiff = {}, outSpec = {}, inDiffSh = {}, inSpecSh = {}, outDiffSh = {}, outSpecSh = {};
    if (DIFF) inDiff = a.planes[k++];
    if (SPEC) inSpec = a.planes[k++];
    if (DIFF && SH) inDiffSh = a.planes[k++];
    if (SPEC && SH) inSpecSh = a.planes[k++];
    if (DIFF) outDiff = a.planes[k++];
    if (SPEC) outSpec = a.planes[k++];
    if (DIFF && SH) outDiffSh = a.planes[k++];
    if (SPEC && SH) outSpecSh = a.planes[k++];
    if (k != a.planesNum)
        return "REBLUR split screen: unexpected resource count";
    RowGrid g = GridForRows(c.gRectSizeMinusOne.x + 1, c.gRectSizeMinusOne.y + 1, TILE_X, TILE_Y, a.rowBegin, a.rowEnd);
    const RowRange rows = MakeRowRange(g);
    // the occlusion family and directional occlusion bind their planes to the radiance family's pipeline: the codec follows the plane format
    const uint32_t format = a.formats[1];
    for (uint32_t i = 1; i < k; i++)
        if (a.formats[i] != format)
            return "REBLUR split screen: mixed signal formats";
    if (format == (uint32_t)FORMAT_RGBA16_SFLOAT)
        LaunchPass(a, (ReblurSplitScreenKernel<DIFF, SPEC, SIGNAL_RADIANCE>), g.grid, dim3(256), c, viewZ, inDiff, inSpec, outDiff, outSpec, inDiffSh, inSpecSh, outDiffSh, outSpecSh, rows);
    else if (format == (uint32_t)FORMAT_R16_UNORM)
        LaunchPass(a, (ReblurSplitScreenKernel<DIFF, SPEC, SIGNAL_OCCLUSION>), g.grid, dim3(256), c, viewZ, inDiff, inSpec, outDiff, outSpec, inDiffSh, inSpecSh, outDiffSh, outSpecSh, rows);
    else if (format == (uint32_t)FORMAT_RGBA16_SNORM)
        LaunchPass(a, (ReblurSplitScreenKernel<DIFF, SPEC, SIGNAL_DIRECTIONAL_OCCLUSION>), g.grid, dim3(256), c, viewZ, inDiff, inSpec, outDiff, outSpec, inDiffSh, inSpecSh, outDiffSh,
            outSpecSh, rows);
    else
        return "REBLUR split screen: unexpected signal format";
    return nullptr;
}

// ================================================================================================ HitDistReconstruction
// reference Shaders/Include/REBLUR_HitDistReconstruction.hlsli:10-160 (normalised hit distances; PERF drops the normal / roughness weights).
// An optional pass (HitDistanceReconstructionMode != OFF): the 3x3 / 5x5 window is read straight from L1/L2 at rect-clamped
// coordinates -- 8 / 24 taps of (decoded normal 16 B, viewZ 4 B, hit distance 2 x 8 B) -- instead of staging an LDS tile.
struct HitDistPlanes {
    Plane tiles, viewZ, inDiff, inSpec, outDiff, outSpec;
    NormalRoughnessGuide decodedNR;
};

template <bool DIFF, bool SPEC, int BORDER, bool PERF, int KIND>
__global__ __launch_bounds__(TILE_X* TILE_Y) void ReblurHitDistReconstructionKernel(ReblurCB c, HitDistPlanes P, RowRange rr) {
    typedef ReblurSignal<KIND> Sig;
    typedef typename Sig::type S;
    const int px = BlockTileX(rr) * TILE_X + (threadIdx.x % TILE_X);
    const int py = (BlockTileY(rr)) * TILE_Y + (threadIdx.x / TILE_X);
    const int rw = c.gRectSizeMinusOne.x, rh = c.gRectSizeMinusOne.y;
    if (px > rw || py > rh || py < rr.rowBegin || py >= rr.rowEnd)
        return;
    if (LoadR8Unorm(P.tiles, px >> 4, py >> 4) != 0.0f)
        return;
    const float centerZ = UnpackViewZ(c, LoadR32F(P.viewZ, px, py));
    if (centerZ > c.gDenoisingRange)
        return;

    const float4 normalAndRoughness = LoadDecodedNormalRoughness(P.decodedNR, px, py);
    const float3 N = Xyz(normalAndRoughness);
    const float roughness = normalAndRoughness.w;

    const float2 rectSizeInv = ToF2(c.gRectSizeInv);
    const float4 frustum = ToF4(c.gFrustum);
    const float2 pixelUv = F2(float(px) + 0.5f, float(py) + 0.5f) * rectSizeInv;
    const float3 Xv = ReconstructViewPosition(pixelUv, frustum, centerZ, NRD_ORTHO_MODE(c));
    const float3 Nv = RotateVectorInverse(c.gViewToWorld, N);
    const float frustumSize = GetFrustumSize(c.gMinRectDimMulUnproject, NRD_ORTHO_MODE(c), centerZ);

    const float2 geometryWeightParams = GetGeometryWeightParams(c.gPlaneDistSensitivity, frustumSize, Xv, Nv);
    const float2 relaxedRoughnessWeightParams = GetRelaxedRoughnessWeightParams(roughness * roughness);
    const float diffNormalWeightParam = GetNormalWeightParam(1.0f, 1.0f);
    const float specNormalWeightParam = GetNormalWeightParam(1.0f, 1.0f, roughness);

    const S centerDiff = DIFF ? Sig::Load(P.inDiff, px, py) : Sig::Zero(), centerSpec = SPEC ? Sig::Load(P.inSpec, px, py) : Sig::Zero();
    float2 center = F2(ExtractHitDist(centerDiff), ExtractHitDist(centerSpec));
    float2 sum = F2(center.x != 0.0f ? 1000.0f : 0.0f, center.y != 0.0f ? 1000.0f : 0.0f);
    center = center * sum;

    for (int j = 0; j <= BORDER * 2; j++)
        for (int i = 0; i <= BORDER * 2; i++) {
            const float2 o = F2(float(i - BORDER), float(j - BORDER));
            if (o.x == 0.0f && o.y == 0.0f)
                continue;
            const int sx = ClampI(px + i - BORDER, 0, rw), sy = ClampI(py + j - BORDER, 0, rh);
            float2 data = F2(DIFF ? ExtractHitDist(Sig::Load(P.inDiff, sx, sy)) : 0.0f, SPEC ? ExtractHitDist(Sig::Load(P.inSpec, sx, sy)) : 0.0f);
            const float dataZ = UnpackViewZ(c, LoadR32F(P.viewZ, sx, sy));

            float w = IsInScreenNearest(pixelUv + o * rectSizeInv);
            w *= GetGaussianWeight(Length(o) * 0.5f);

            const float2 uv = pixelUv + o * rectSizeInv;
            const float3 Xvs = ReconstructViewPosition(uv, frustum, dataZ, NRD_ORTHO_MODE(c));
            w *= ComputeWeight(Dot(Nv, Xvs), geometryWeightParams.x, geometryWeightParams.y);

            float2 ww = F2(w, w);
            if (!PERF) {
                const float4 sampleNormalAndRoughness = LoadDecodedNormalRoughness(P.decodedNR, sx, sy);
                const float cosa = Dot(N, Xyz(sampleNormalAndRoughness));
                const float angle = AcosApprox(cosa);
                ww.x *= ComputeExponentialWeight(angle, diffNormalWeightParam, 0.0f);
                ww.y *= ComputeExponentialWeight(angle, specNormalWeightParam, 0.0f);
                ww.y *= ComputeExponentialWeight(sampleNormalAndRoughness.w * sampleNormalAndRoughness.w, relaxedRoughnessWeightParams.x, relaxedRoughnessWeightParams.y);
            }

            data.x = ww.x == 0.0f ? 0.0f : data.x;
            data.y = ww.y == 0.0f ? 0.0f : data.y;
            ww = ww * F2(data.x != 0.0f ? 1.0f : 0.0f, data.y != 0.0f ? 1.0f : 0.0f);

            center = Mad(data, ww, center);
            sum = sum + ww;
        }
    center = Div(center, F2(Max(sum.x, NRD_EPS), Max(sum.y, NRD_EPS)));

    if (DIFF)
        Sig::Store(P.outDiff, px, py, Sig::WithHitDist(centerDiff, center.x));
    if (SPEC)
        Sig::Store(P.outSpec, px, py, Sig::WithHitDist(centerSpec, center.y));
}

template <bool DIFF, bool SPEC, int BORDER, bool PERF, int KIND>
static const char* LaunchHitDistReconstruction(const PassArgs& a) {
    // REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION binds its RGBA16_SNORM planes to the radiance family's reconstruction pipeline
    if (KIND == SIGNAL_RADIANCE && DIFF && !SPEC && a.planesNum == 5 && a.formats[3] == FORMAT_RGBA16_SNORM)
        return LaunchHitDistReconstruction<DIFF, SPEC, BORDER, PERF, (DIFF && !SPEC) ? SIGNAL_DIRECTIONAL_OCCLUSION : KIND>(a);
    const ReblurCB& c = *(const ReblurCB*)a.constants;
    if (const char* err = CheckSupported(c))
        return err;
    HitDistPlanes P = {};
    uint32_t k = 0;
    P.tiles = a.planes[k++];
    k++; // packed normal / roughness: read through the decoded cache
    P.viewZ = a.planes[k++];
    if (DIFF) P.inDiff = a.planes[k++];
    if (SPEC) P.inSpec = a.planes[k++];
    if (DIFF) P.outDiff = a.planes[k++];
    if (SPEC) P.outSpec = a.planes[k++];
    if (k != a.planesNum)
        return "REBLUR hit distance reconstruction: unexpected resource count";
    if (const char* err = MakeNormalRoughnessGuide(a, P.decodedNR))
        return err;
    RowGrid g = GridForRows(c.gRectSizeMinusOne.x + 1, c.gRectSizeMinusOne.y + 1, TILE_X, TILE_Y, a.rowBegin, a.rowEnd);
    LaunchPass(a, (ReblurHitDistReconstructionKernel<DIFF, SPEC, BORDER, PERF, KIND>), g.grid, dim3(TILE_X * TILE_Y), c, P, MakeRowRange(g));
    return nullptr;
}

// quality and performance ("REBLUR_Perf_*") permutations of one signal family; "Sh" = SH family (reuses the radiance family's hit-distance
// reconstruction), "Occlusion" = hit distance only (R16_UNORM), no pre-pass, post-blur always without temporal stabilisation
#define REBLUR_SPATIAL_PASSES(PREFIX, NAME, D, S, P)                                                                   \
    {PREFIX NAME "_HitDistReconstruction.cs", LaunchHitDistReconstruction<D, S, 1, P, false>},                       \
    {PREFIX NAME "_HitDistReconstruction_5x5.cs", LaunchHitDistReconstruction<D, S, 2, P, false>},                   \
    {PREFIX NAME "_PrePass.cs", LaunchSpatial<PRE_BLUR, D, S, false, P, false, false>},                              \
    {PREFIX NAME "_Blur.cs", LaunchSpatial<BLUR, D, S, false, P, false, false>},                                     \
    {PREFIX NAME "_PostBlur.cs", LaunchSpatial<POST_BLUR, D, S, false, P, false, false>},                            \
    {PREFIX NAME "_PostBlur_NoTemporalStabilization.cs", LaunchSpatial<POST_BLUR, D, S, true, P, false, false>},     \
    {PREFIX NAME "Sh_PrePass.cs", LaunchSpatial<PRE_BLUR, D, S, false, P, false, true>},                             \
    {PREFIX NAME "Sh_Blur.cs", LaunchSpatial<BLUR, D, S, false, P, false, true>},                                    \
    {PREFIX NAME "Sh_PostBlur.cs", LaunchSpatial<POST_BLUR, D, S, false, P, false, true>},                           \
    {PREFIX NAME "Sh_PostBlur_NoTemporalStabilization.cs", LaunchSpatial<POST_BLUR, D, S, true, P, false, true>},    \
    {PREFIX NAME "Occlusion_HitDistReconstruction.cs", LaunchHitDistReconstruction<D, S, 1, P, true>},               \
    {PREFIX NAME "Occlusion_HitDistReconstruction_5x5.cs", LaunchHitDistReconstruction<D, S, 2, P, true>},           \
    {PREFIX NAME "Occlusion_Blur.cs", LaunchSpatial<BLUR, D, S, false, P, true, false>},                             \
    {PREFIX NAME "Occlusion_PostBlur_NoTemporalStabilization.cs", LaunchSpatial<POST_BLUR, D, S, true, P, true, false>},
#define REBLUR_SPATIAL_FAMILY(NAME, D, S)                                                              \
    REBLUR_SPATIAL_PASSES("REBLUR_", NAME, D, S, false)                                                \
    REBLUR_SPATIAL_PASSES("REBLUR_Perf_", NAME, D, S, true)                                            \
    {"REBLUR_" NAME "_SplitScreen.cs", LaunchSplitScreen<D, S, false>},                                \
    {"REBLUR_" NAME "Sh_SplitScreen.cs", LaunchSplitScreen<D, S, true>},
// REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION: the diffuse chain on RGBA16_SNORM texels
#define REBLUR_DIRECTIONAL_OCCLUSION_SPATIAL(PREFIX, P)                                                                                       \
    {PREFIX "DiffuseDirectionalOcclusion_PrePass.cs", LaunchSpatial<PRE_BLUR, true, false, false, P, 2, false>},                             \
    {PREFIX "DiffuseDirectionalOcclusion_Blur.cs", LaunchSpatial<BLUR, true, false, false, P, 2, false>},                                    \
    {PREFIX "DiffuseDirectionalOcclusion_PostBlur.cs", LaunchSpatial<POST_BLUR, true, false, false, P, 2, false>},                           \
    {PREFIX "DiffuseDirectionalOcclusion_PostBlur_NoTemporalStabilization.cs", LaunchSpatial<POST_BLUR, true, false, true, P, 2, false>},

const PassEntry* GetReblurSpatialPasses(uint32_t& num) {
    static const PassEntry k[] = {
        {"REBLUR_ClassifyTiles.cs", LaunchClassifyTiles},
        REBLUR_SPATIAL_FAMILY("Diffuse", true, false)
        REBLUR_SPATIAL_FAMILY("Specular", false, true)
        REBLUR_SPATIAL_FAMILY("DiffuseSpecular", true, true)
        REBLUR_DIRECTIONAL_OCCLUSION_SPATIAL("REBLUR_", false)
        REBLUR_DIRECTIONAL_OCCLUSION_SPATIAL("REBLUR_Perf_", true)
    };
    num = sizeof(k) / sizeof(k[0]);
    return k;
}

} // namespace nrdhip
