// SIGMA_SHADOW and SIGMA_SHADOW_TRANSLUCENCY pass chains as HIP kernels for gfx950 (one template per pass over SIGMA_TYPE:
// float = shadow in R8, float4 = shadow + translucent colour in RGBA8; reference "#ifdef SIGMA_TRANSLUCENT").
//   ClassifyTiles           reference Shaders/Include/SIGMA_ClassifyTiles.hlsli:11-81
//   SmoothTiles             reference Shaders/Include/SIGMA_SmoothTiles.hlsli:11-48
//   Copy                    reference Shaders/Include/SIGMA_Copy.hlsli:11-24
//   Blur / PostBlur         reference Shaders/Include/SIGMA_Blur.hlsli:11-268
//   TemporalStabilization   reference Shaders/Include/SIGMA_TemporalStabilization.hlsli:11-226
//   SplitScreen             reference Shaders/Include/SIGMA_SplitScreen.hlsli:11-35
//
// MI355X mapping. The chain moves only ~68 B/px (R16F penumbra, R8 shadow, R32F viewZ), so at 1080p a frame is ~140 MB =
// tens of microseconds of HBM time: it is launch- and latency-bound, and the win is in doing little work: hard-shadow and
// fully-lit tiles leave through the bicubic tile test before any filtering. ClassifyTiles is one wave per 16x16 tile with
// wave-wide reductions (__all / shuffle-max) instead of the reference's LDS atomics. Blur / PostBlur / TS stage a 36x12
// (halo 2) LDS tile of {penumbra, viewZ, shadow} decoded once per workgroup for the dense 5x5 part; the sparse 8-tap part
// gathers from global (L2). LDS rows are padded to 37 dwords.
#include "../common/pass_constants.h"
#include "passes.h"
#include "reblur_device.h" // shared Common.hlsli helpers (weights, history fetch, clamp-addressed fetches)

namespace nrdhip {

typedef nrdc::SigmaConstants SigmaCB;

constexpr int TILE_X = 32;
constexpr int TILE_Y = 8;
constexpr int BORDER = 2;
constexpr int BUF_X = TILE_X + 2 * BORDER; // 36
constexpr int BUF_Y = TILE_Y + 2 * BORDER; // 12
constexpr int BUF_STRIDE = BUF_X + 1;      // 37

#define SIGMA_MAX_PIXEL_RADIUS 32.0f
#define SIGMA_TS_SIGMA_SCALE 3.0f
#define SIGMA_MAX_ACCUM_FRAME_NUM 7.0f

NRD_D float UnpackViewZ(const SigmaCB& c, float z) { return Abs(z * c.gViewZScale); }
NRD_D bool IsLit(float p) { return p >= NRD_FP16_MAX; }
NRD_D float PackShadow(float s) { return Sqrt01(s); }
NRD_D float UnpackShadow(float s) { return s * s; }
NRD_D float4 PackShadow(float4 s) { return F4(Sqrt01(s.x), Sqrt01(s.y), Sqrt01(s.z), Sqrt01(s.w)); }
NRD_D float4 UnpackShadow(float4 s) { return s * s; }

// SIGMA_TYPE (reference SIGMA_Config.hlsli:39-43): storage codec + the component-wise intrinsics the passes apply to it
template <bool TRANSLUCENT>
struct SigmaType;
template <>
struct SigmaType<false> {
    typedef float type;
    static NRD_D float Load(const Plane& p, int x, int y) { return LoadR8Unorm(p, x, y); }
    static NRD_D void Store(const Plane& p, int x, int y, float v) { StoreR8Unorm(p, x, y, v); }
    static NRD_D float Splat(float v) { return v; }
    static NRD_D float X(float v) { return v; }
};
template <>
struct SigmaType<true> {
    typedef float4 type;
    static NRD_D float4 Load(const Plane& p, int x, int y) { return LoadRGBA8Unorm(p, x, y); }
    static NRD_D void Store(const Plane& p, int x, int y, float4 v) { StoreRGBA8Unorm(p, x, y, v); }
    static NRD_D float4 Splat(float v) { return F4(v); }
    static NRD_D float X(float4 v) { return v.x; }
};
NRD_D float ZeroIf(bool c, float v) { return c ? 0.0f : v; }
NRD_D float4 ZeroIf(bool c, float4 v) { return Select(c, F4(0.0f), v); }
NRD_D float StdDev(float m1, float m2) { return Sqrt(Abs(m2 - m1 * m1)); }
NRD_D float4 StdDev(float4 m1, float4 m2) { return F4(StdDev(m1.x, m2.x), StdDev(m1.y, m2.y), StdDev(m1.z, m2.z), StdDev(m1.w, m2.w)); }
NRD_D float ClampV(float x, float a, float b) { return Clamp(x, a, b); }
NRD_D float4 ClampV(float4 x, float4 a, float4 b) { return F4(Clamp(x.x, a.x, b.x), Clamp(x.y, a.y, b.y), Clamp(x.z, a.z, b.z), Clamp(x.w, a.w, b.w)); }
NRD_D float SatV(float x) { return Sat(x); }
NRD_D float4 SatV(float4 x) { return F4(Sat(x.x), Sat(x.y), Sat(x.z), Sat(x.w)); }
NRD_D float GetKernelRadiusInPixels(float hitDist, float unprojectZ, float scale = 1.0f) {
    float unclampedRadius = Div(hitDist, unprojectZ);
    unclampedRadius *= scale;
    float minRadius = Min(unclampedRadius, 2.0f);
    return Clamp(unclampedRadius, minRadius, SIGMA_MAX_PIXEL_RADIUS);
}
NRD_D float AreBothLitOrUnlit(float penumbra1, float penumbra2) { return ((penumbra1 == 0.0f) == (penumbra2 == 0.0f)) ? 1.0f : 0.0f; }

// ---- bicubic lookup of channel .y of the RG8 smoothed tile map (SIGMA_Common.hlsli:45-92) -------------------------------
NRD_D float FetchClampedRG8y(const Plane& p, int x, int y) {
    uint32_t raw = *TexelPtr<const uint16_t>(p, ClampI(x, 0, p.w - 1), ClampI(y, 0, p.h - 1));
    return NRD_DIV_255(float(raw >> 8));
}
NRD_D float SampleLinearRG8y(const Plane& p, float2 pos) {
    LinearTaps t = MakeLinearTaps(pos);
    float s00 = FetchClampedRG8y(p, t.x0, t.y0), s10 = FetchClampedRG8y(p, t.x0 + 1, t.y0), s01 = FetchClampedRG8y(p, t.x0, t.y0 + 1), s11 = FetchClampedRG8y(p, t.x0 + 1, t.y0 + 1);
    return s00 * t.w00 + s10 * t.w10 + s01 * t.w01 + s11 * t.w11;
}
NRD_D void BicubicAxis(float f, float& w0, float& w1, float& wz) {
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
NRD_D float TextureCubicY(const Plane& tex, float2 uv) {
    float2 size = F2(float(tex.w), float(tex.h));
    float dx = -Rcp(size.x), dy = -Rcp(size.y);
    float2 t = uv * size - 0.5f;
    float2 f = F2(Frac(t.x), Frac(t.y));
    float xw0, xw1, xwz, yw0, yw1, ywz;
    BicubicAxis(f.x, xw0, xw1, xwz);
    BicubicAxis(f.y, yw0, yw1, ywz);
    float u10 = uv.x + 1.0f * xw0 * dx, u00 = uv.x + -1.0f * xw1 * dx;
    float v1 = uv.y + yw0 * dy, v0 = uv.y - yw1 * dy;
    float c00 = SampleLinearRG8y(tex, F2(u00, v0) * size);
    float c10 = SampleLinearRG8y(tex, F2(u10, v0) * size);
    float c01 = SampleLinearRG8y(tex, F2(u00, v1) * size);
    float c11 = SampleLinearRG8y(tex, F2(u10, v1) * size);
    float tx = ywz, ty = xwz;
    c00 = Lerp(c00, c01, tx);
    c10 = Lerp(c10, c11, tx);
    return Lerp(c00, c10, ty);
}
NRD_D float LoadTileX(const Plane& tiles, int tx, int ty) { // .x of the RG8 smoothed tile map, 0 outside
    if (!InBounds(tiles, tx, ty))
        return 0.0f;
    uint32_t raw = *TexelPtr<const uint16_t>(tiles, tx, ty);
    return NRD_DIV_255(float(raw & 0xFFu));
}

// ================================================================================================ ClassifyTiles
template <bool TRANSLUCENT>
__global__ __launch_bounds__(256) void SigmaClassifyTilesKernel(SigmaCB c, Plane viewZ, Plane penumbra, Plane translucency, Plane tiles) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // the tiles of the RECT (the reference's dispatch grid, ceil( rectSize / 16 ) groups): tiles of the plane beyond it are left alone
    const int tilesW = min(tiles.w, ((int)c.gRectSize.x + 15) >> 4), tilesH = min(tiles.h, ((int)c.gRectSize.y + 15) >> 4);
    const int tileIndex = blockIdx.x * 4 + wave;
    if (tileIndex >= tilesW * tilesH)
        return;
    const int tx = tileIndex % tilesW, ty = tileIndex / tilesW;
    const int x0 = tx * 16 + (lane & 3) * 4, y = ty * 16 + (lane >> 2);

    bool allLit = true, allUmbra = true, allInf = true;
    float maxRadius = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int x = x0 + i;
        float h = InBounds(penumbra, x, y) ? LoadR16F(penumbra, x, y) : 0.0f;
        float z = UnpackViewZ(c, InBounds(viewZ, x, y) ? LoadR32F(viewZ, x, y) : 0.0f);
        bool isInf = z > c.gDenoisingRange, isShadow = h == 0.0f, isLitP = IsLit(h);
        allLit = allLit && (isLitP || isInf || isShadow);
        bool isOpaque = true;
        if (TRANSLUCENT) {
            float4 t = InBounds(translucency, x, y) ? LoadRGBA8Unorm(translucency, x, y) : F4(0.0f);
            isOpaque = Luminance(F3(t.y, t.z, t.w)) < 0.003f;
        }
        allUmbra = allUmbra && ((!isLitP && isOpaque) || isInf || isShadow);
        allInf = allInf && isInf;
        float hitDist = (isLitP || isInf) ? 0.0f : h;
        float pixelSize = PixelRadiusToWorld(c.gUnproject, c.gOrthoMode, 1.0f, z);
        maxRadius = Max(GetKernelRadiusInPixels(hitDist, pixelSize), maxRadius);
    }
    allLit = __all(allLit);
    allUmbra = __all(allUmbra);
    allInf = __all(allInf);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        maxRadius = Max(maxRadius, __shfl_xor(maxRadius, o, 64));

    if (lane == 0) {
        uint32_t r = ToUnorm((allLit || allUmbra) ? 0.0f : 1.0f, 255.0f);
        r |= ToUnorm(Sat(maxRadius * 0.0625f), 255.0f) << 8;
        r |= ToUnorm(allInf ? 1.0f : 0.0f, 255.0f) << 16;
        *TexelPtr<uint32_t>(tiles, tx, ty) = r; // RGBA8_UNORM, .w = 0
    }
}

static const char* CheckSupportedSigma(const SigmaCB& c) {
    if (c.gRectOrigin.x != 0 || c.gRectOrigin.y != 0) // the executor moves the rect of the guide inputs to (0, 0) and zeroes this field (executor.hip "shifted rect")
        return "internal error: a pass was handed a non-zero rectOrigin";
    if (c.gOrthoMode != 0.0f)
        return "SIGMA: orthographic projection is not supported (SURVEY.md section 8c)";
    return nullptr;
}

template <bool TRANSLUCENT>
static const char* LaunchClassifyTiles(const PassArgs& a) {
    const SigmaCB& c = *(const SigmaCB*)a.constants;
    if (const char* err = CheckSupportedSigma(c))
        return err;
    if (a.planesNum != (TRANSLUCENT ? 4u : 3u))
        return "SIGMA classify tiles: unexpected resource count";
    const Plane& tiles = a.planes[a.planesNum - 1];
    int numTiles = tiles.w * tiles.h;
    LaunchPass(a, (SigmaClassifyTilesKernel<TRANSLUCENT>), dim3((numTiles + 3) / 4), dim3(256), c, a.planes[0], a.planes[1], TRANSLUCENT ? a.planes[2] : Plane{}, tiles);
    return nullptr;
}

// ================================================================================================ SmoothTiles
__global__ __launch_bounds__(256) void SigmaSmoothTilesKernel(SigmaCB c, Plane inTiles, Plane outTiles) {
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (!InBounds(outTiles, x, y))
        return;
    uint32_t raw = *TexelPtr<const uint32_t>(inTiles, x, y);
    float centerY = NRD_DIV_255(float((raw >> 8) & 0xFFu)), centerZ = NRD_DIV_255(float((raw >> 16) & 0xFFu));
    float blurry = 0.0f, sumw = 0.0f;
    float k = Div(1.01f, centerY + 0.01f);
#pragma unroll
    for (int j = 0; j <= 2; j++) {
#pragma unroll
        for (int i = 0; i <= 2; i++) {
            float d = Length(F2(float(i), float(j)) - 1.0f);
            float w = Exp2(-k * d * d);
            int sx = ClampI(x - 1 + i, 0, c.gTilesSizeMinusOne.x), sy = ClampI(y - 1 + j, 0, c.gTilesSizeMinusOne.y);
            uint32_t t = *TexelPtr<const uint32_t>(inTiles, sx, sy);
            blurry += NRD_DIV_255(float(t & 0xFFu)) * w;
            sumw += w;
        }
    }
    blurry = Div(blurry, sumw);
    StoreRG8Unorm(outTiles, x, y, F2(centerZ, blurry));
}

static const char* LaunchSmoothTiles(const PassArgs& a) {
    const SigmaCB& c = *(const SigmaCB*)a.constants;
    const Plane& out = a.planes[1];
    LaunchPass(a, SigmaSmoothTilesKernel, GridFor(out.w, out.h, 16, 16), dim3(256), c, a.planes[0], out);
    return nullptr;
}

// Multi-GPU row strips (PassArgs::rowBegin / rowEnd; round 6: SIGMA's passes are sharded like REBLUR's and RELAX's): the grid covers the tile rows that intersect the strip and the
// kernel adds blockY0 to blockIdx.y. Rows of the first / last tile row that lie outside the strip are produced as well (a superset is always correct). Unsharded: the whole rect.
struct RowBand {
    int blockY0;
    unsigned blocksY;
};
static RowBand MakeRowBand(const PassArgs& a, int h, int tileH) {
    const int rb = a.rowBegin < 0 ? 0 : (a.rowBegin > h ? h : a.rowBegin), re = (a.rowEnd > h || a.rowEnd <= a.rowBegin) ? h : a.rowEnd;
    RowBand b;
    b.blockY0 = rb / tileH;
    const int last = (re + tileH - 1) / tileH;
    b.blocksY = (unsigned)(last > b.blockY0 ? last - b.blockY0 : 1);
    return b;
}

// ================================================================================================ Copy
template <typename TEXEL> // uint8_t (R8 shadow) or uint32_t (RGBA8 shadow + translucency)
__global__ __launch_bounds__(256) void SigmaCopyKernel(SigmaCB c, Plane tiles, Plane inHistory, Plane inHistoryLength, Plane outHistory, Plane outHistoryLength, int blockY0) {
    // 4 pixels per thread: 4 (or 16) bytes of shadow history and 16 bytes of R32_UINT history length
    struct alignas(sizeof(TEXEL) * 4) Texel4 {
        TEXEL v[4];
    };
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = ((int)blockIdx.y + blockY0) * 4 + (threadIdx.x >> 6);
    if (y >= outHistory.h || x >= outHistory.w)
        return;
    if (LoadTileX(tiles, x >> 4, y >> 4) != 0.0f && !c.gIsRectChanged)
        return;
    const bool aligned = ((((uintptr_t)inHistory.ptr | (uintptr_t)outHistory.ptr) | inHistory.pitch | outHistory.pitch) & (sizeof(Texel4) - 1)) == 0 &&
                         ((((uintptr_t)inHistoryLength.ptr | (uintptr_t)outHistoryLength.ptr) | inHistoryLength.pitch | outHistoryLength.pitch) & 15u) == 0;
    if (x + 3 < outHistory.w && aligned) {
        *(Texel4*)TexelPtr<TEXEL>(outHistory, x, y) = *(const Texel4*)TexelPtr<const TEXEL>(inHistory, x, y);
        *(uint4*)TexelPtr<uint32_t>(outHistoryLength, x, y) = *(const uint4*)TexelPtr<const uint32_t>(inHistoryLength, x, y);
    } else {
        for (int i = 0; i < 4 && x + i < outHistory.w; i++) {
            *TexelPtr<TEXEL>(outHistory, x + i, y) = *TexelPtr<const TEXEL>(inHistory, x + i, y);
            StoreR32U(outHistoryLength, x + i, y, LoadR32U(inHistoryLength, x + i, y));
        }
    }
}

static const char* LaunchCopy(const PassArgs& a) {
    const SigmaCB& c = *(const SigmaCB*)a.constants;
    const Plane& out = a.planes[3];
    const RowBand band = MakeRowBand(a, out.h, 4);
    dim3 grid((unsigned)((out.w + 255) / 256), band.blocksY, 1);
    if (a.planesNum != 5 || a.bytesPerTexel[1] != a.bytesPerTexel[3])
        return "SIGMA copy: unexpected resources";
    if (a.bytesPerTexel[3] == 4)
        LaunchPass(a, SigmaCopyKernel<uint32_t>, grid, dim3(256), c, a.planes[0], a.planes[1], a.planes[2], out, a.planes[4], band.blockY0);
    else if (a.bytesPerTexel[3] == 1)
        LaunchPass(a, SigmaCopyKernel<uint8_t>, grid, dim3(256), c, a.planes[0], a.planes[1], a.planes[2], out, a.planes[4], band.blockY0);
    else
        return "SIGMA copy: unexpected history format";
    return nullptr;
}

// ================================================================================================ Blur / PostBlur
struct BlurPlanes {
    Plane viewZ, normalRoughness, penumbra, tiles, shadow, outPenumbra, outShadow;
};

template <bool FIRST_PASS, bool TRANSLUCENT>
__global__ __launch_bounds__(TILE_X* TILE_Y) void SigmaBlurKernel(SigmaCB c, BlurPlanes P, int blockY0) {
    typedef SigmaType<TRANSLUCENT> ST;
    typedef typename ST::type S;
    constexpr bool READS_SHADOW = !FIRST_PASS || TRANSLUCENT; // the translucent first pass reads IN_TRANSLUCENCY (not unpacked)
    __shared__ float s_Penumbra[BUF_Y * BUF_STRIDE];
    __shared__ float s_ViewZ[BUF_Y * BUF_STRIDE];
    __shared__ S s_Shadow[BUF_Y * BUF_STRIDE];

    const int tx = threadIdx.x % TILE_X, ty = threadIdx.x / TILE_X;
    const int blockY = (int)blockIdx.y + blockY0; // (multi-GPU row strips: RowBand)
    const int px = blockIdx.x * TILE_X + tx, py = blockY * TILE_Y + ty;
    const int rw = c.gRectSizeMinusOne.x, rh = c.gRectSizeMinusOne.y;

    {
        const int tileY = (blockY * TILE_Y) >> 4, tileX0 = (blockIdx.x * TILE_X) >> 4;
        bool anyGeometry = false;
        for (int t = 0; t < TILE_X / 16; t++)
            anyGeometry |= InBounds(P.tiles, tileX0 + t, tileY) && LoadTileX(P.tiles, tileX0 + t, tileY) == 0.0f;
        if (!anyGeometry)
            return;
        const int baseX = blockIdx.x * TILE_X - BORDER, baseY = blockY * TILE_Y - BORDER;
        for (int i = threadIdx.x; i < BUF_X * BUF_Y; i += TILE_X * TILE_Y) {
            int lx = i % BUF_X, ly = i / BUF_X;
            int gx = ClampI(baseX + lx, 0, rw), gy = ClampI(baseY + ly, 0, rh);
            float pen = LoadR16F(P.penumbra, gx, gy);
            s_Penumbra[ly * BUF_STRIDE + lx] = pen;
            s_ViewZ[ly * BUF_STRIDE + lx] = UnpackViewZ(c, LoadR32F(P.viewZ, gx, gy));
            S s;
            if (READS_SHADOW)
                s = ST::Load(P.shadow, gx, gy);
            else
                s = ST::Splat(IsLit(pen) ? 1.0f : 0.0f);
            s_Shadow[ly * BUF_STRIDE + lx] = FIRST_PASS ? s : UnpackShadow(s);
        }
    }
    __syncthreads();

    if (px > rw || py > rh)
        return;
    if (LoadTileX(P.tiles, px >> 4, py >> 4) != 0.0f)
        return;

    const int so = (ty + BORDER) * BUF_STRIDE + tx + BORDER;
    const float centerPenumbra = s_Penumbra[so];
    const float viewZ = s_ViewZ[so];
    if (viewZ > c.gDenoisingRange)
        return;

    const float2 rectSize = ToF2(c.gRectSize), rectSizeInv = ToF2(c.gRectSizeInv), resolutionScale = ToF2(c.gResolutionScale);
    const float4 frustum = ToF4(c.gFrustum);

    float2 pixelUv = F2(float(px) + 0.5f, float(py) + 0.5f) * rectSizeInv;
    float tileValue = TextureCubicY(P.tiles, pixelUv * resolutionScale);

    if (tileValue == 0.0f || centerPenumbra == 0.0f) {
        StoreR16F(P.outPenumbra, px, py, centerPenumbra);
        ST::Store(P.outShadow, px, py, PackShadow(s_Shadow[so]));
        return;
    }

    float3 Xv = ReconstructViewPosition(pixelUv, frustum, viewZ, c.gOrthoMode);
    float3 N = Xyz(UnpackNormalAndRoughness(LoadInNormalRoughnessTexel(P.normalRoughness, px, py)));
    float3 Nv = RotateVector(c.gWorldToView, N);

    float pixelSize = PixelRadiusToWorld(c.gUnproject, c.gOrthoMode, 1.0f, viewZ);
    float frustumSize = GetFrustumSize(c.gMinRectDimMulUnproject, c.gOrthoMode, viewZ);
    float3 Vv = c.gOrthoMode == 0.0f ? Normalize(-Xv) : F3(0.0f, 0.0f, -1.0f);
    float NoV = Abs(Dot(Nv, Vv));
    float2 geometryWeightParams = GetGeometryWeightParams(c.gPlaneDistSensitivity, frustumSize, Xv, Nv);

    float sumx = 0.0f, sumy = 0.0f, penumbra = 0.0f;
    S result = ST::Splat(0.0f), centerTap = ST::Splat(0.0f);
#pragma unroll
    for (int j = 0; j <= BORDER * 2; j++) {
#pragma unroll
        for (int i = 0; i <= BORDER * 2; i++) {
            const int o = (ty + j) * BUF_STRIDE + tx + i;
            float penum = s_Penumbra[o], zs = s_ViewZ[o];
            S s = s_Shadow[o];

            float w = 1.0f;
            if (i == BORDER && j == BORDER)
                centerTap = s;
            else {
                float2 uv = pixelUv + F2(float(i - BORDER), float(j - BORDER)) * rectSizeInv;
                float3 Xvs = ReconstructViewPosition(uv, frustum, zs, c.gOrthoMode);
                w *= ComputeWeight(Dot(Nv, Xvs), geometryWeightParams.x, geometryWeightParams.y);
                w *= AreBothLitOrUnlit(centerPenumbra, penum);
                w *= GetGaussianWeight(Length(Div(F2(float(i - BORDER), float(j - BORDER)), float(BORDER))));
            }

            result = result + ZeroIf(w == 0.0f, s * w);
            sumx += w;

            w *= Div(pixelSize, pixelSize + penum);
            w *= IsLit(penum) ? 0.0f : 1.0f;

            penumbra += w == 0.0f ? 0.0f : penum * w;
            sumy += w;
        }
    }

    result = Div(result, sumx);
    sumx = 1.0f;
    penumbra = Div(penumbra, Max(sumy, NRD_EPS));
    sumy = sumy != 0.0f ? 1.0f : 0.0f;

    float penumbraInPixels = Div(penumbra, pixelSize);
    float f = SmoothStep(0.0f, float(BORDER), penumbraInPixels);
    result = Lerp(centerTap, result, f);

    f = Lerp(4.0f, 1.0f, f);
    result = result * f;
    penumbra *= f;
    sumx *= f;
    sumy *= f;

    float blurRadius = GetKernelRadiusInPixels(penumbra, pixelSize, tileValue);
    float4 rotator = ToF4(FIRST_PASS ? c.gRotator : c.gRotatorPost);

    float2 skew = Lerp(F2(1.0f - Abs(Nv.x), 1.0f - Abs(Nv.y)), F2(1.0f, 1.0f), NoV);
    skew = Div(skew, Max(skew.x, skew.y));
    skew = skew * (rectSizeInv * blurRadius);
    float4 scaledRotator = ScaleRotator(rotator, skew);

    float invEstimatedPenumbra = Rcp(Max(penumbra, NRD_EPS));
    const float2 uvMax = resolutionScale - ToF2(c.gResourceSizeInv) * 0.5f;

#pragma unroll
    for (int n = 0; n < 8; n++) {
        float3 offset = F3(g_Special8[n][0], g_Special8[n][1], g_Special8[n][2]);
        float2 uv = pixelUv + RotateVector(scaledRotator, F2(offset.x, offset.y));
        uv = (Floor(uv * rectSize) + 0.5f) * rectSizeInv;
        float2 uvScaled = F2(Min(uv.x * resolutionScale.x, uvMax.x), Min(uv.y * resolutionScale.y, uvMax.y));

        const int2 t = NearestTexel(P.penumbra, uvScaled);
        float penum = LoadR16F(P.penumbra, t.x, t.y);
        float zs = UnpackViewZ(c, LoadR32F(P.viewZ, t.x, t.y));
        S s;
        if (READS_SHADOW)
            s = ST::Load(P.shadow, t.x, t.y);
        else
            s = ST::Splat(IsLit(penum) ? 1.0f : 0.0f);
        if (!FIRST_PASS)
            s = UnpackShadow(s);

        float3 Xvs = ReconstructViewPosition(uv, frustum, zs, c.gOrthoMode);

        float w = IsInScreenNearest(uv);
        w *= ComputeWeight(Dot(Nv, Xvs), geometryWeightParams.x, geometryWeightParams.y);
        w *= AreBothLitOrUnlit(centerPenumbra, penum);
        w *= n < 4 ? REBLUR_GAUSSIAN_WEIGHT_Z1 : REBLUR_GAUSSIAN_WEIGHT_Z05;
        w *= Sat(penum * invEstimatedPenumbra);

        result = result + ZeroIf(w == 0.0f, s * w);
        sumx += w;

        w *= Div(pixelSize, pixelSize + penum);
        w *= IsLit(penum) ? 0.0f : 1.0f;

        penumbra += w == 0.0f ? 0.0f : penum * w;
        sumy += w;
    }

    result = Div(result, sumx);
    penumbra = sumy == 0.0f ? centerPenumbra : Div(penumbra, sumy);

    if (FIRST_PASS || c.gStabilizationStrength != 0.0f)
        StoreR16F(P.outPenumbra, px, py, penumbra);
    ST::Store(P.outShadow, px, py, PackShadow(result));
}

template <bool FIRST_PASS, bool TRANSLUCENT>
static const char* LaunchBlur(const PassArgs& a) {
    const SigmaCB& c = *(const SigmaCB*)a.constants;
    if (const char* err = CheckSupportedSigma(c))
        return err;
    BlurPlanes P = {};
    uint32_t k = 0;
    P.viewZ = a.planes[k++];
    P.normalRoughness = a.planes[k++];
    P.penumbra = a.planes[k++];
    P.tiles = a.planes[k++];
    if (!FIRST_PASS || TRANSLUCENT)
        P.shadow = a.planes[k++];
    P.outPenumbra = a.planes[k++];
    P.outShadow = a.planes[k++];
    if (k != a.planesNum)
        return "SIGMA blur: unexpected resource count";
    dim3 grid = GridFor(c.gRectSizeMinusOne.x + 1, c.gRectSizeMinusOne.y + 1, TILE_X, TILE_Y);
    const RowBand band = MakeRowBand(a, c.gRectSizeMinusOne.y + 1, TILE_Y);
    grid.y = band.blocksY;
    LaunchPass(a, (SigmaBlurKernel<FIRST_PASS, TRANSLUCENT>), grid, dim3(TILE_X * TILE_Y), c, P, band.blockY0);
    return nullptr;
}

// ================================================================================================ TemporalStabilization
struct TsPlanes {
    Plane viewZ, mv, penumbra, shadow, history, historyLength, tiles, outShadow, outHistoryLength;
};

NRD_D uint32_t PackViewZAndHistoryLength(float viewZ, float historyLength) {
    uint32_t p = AsUint(viewZ) & ~7u;
    uint32_t h = (uint32_t)(historyLength + 0.5f);
    p |= h < 7u ? h : 7u;
    return p;
}
NRD_D uint32_t FetchClampedR32U(const Plane& p, int x, int y) { return LoadR32U(p, ClampI(x, 0, p.w - 1), ClampI(y, 0, p.h - 1)); }
template <bool TRANSLUCENT>
NRD_D typename SigmaType<TRANSLUCENT>::type FetchShadowHistory(const HistoryFilter& h, const Plane& tex) {
    typedef SigmaType<TRANSLUCENT> ST;
    return FetchHistoryGeneric<typename ST::type>(h, tex, [](const Plane& p, int x, int y) { return ST::Load(p, x, y); }, ST::Splat(0.0f));
}

template <bool TRANSLUCENT>
__global__ __launch_bounds__(TILE_X* TILE_Y) void SigmaTemporalStabilizationKernel(SigmaCB c, TsPlanes P, int blockY0, uint32_t* historyReach) {
    typedef SigmaType<TRANSLUCENT> ST;
    typedef typename ST::type S;
    __shared__ float s_Penumbra[BUF_Y * BUF_STRIDE];
    __shared__ S s_Shadow[BUF_Y * BUF_STRIDE];

    const int tx = threadIdx.x % TILE_X, ty = threadIdx.x / TILE_X;
    const int blockY = (int)blockIdx.y + blockY0; // (multi-GPU row strips: RowBand)
    const int px = blockIdx.x * TILE_X + tx, py = blockY * TILE_Y + ty;
    const int rw = c.gRectSizeMinusOne.x, rh = c.gRectSizeMinusOne.y;

    {
        const int tileY = (blockY * TILE_Y) >> 4, tileX0 = (blockIdx.x * TILE_X) >> 4;
        bool anyGeometry = false;
        for (int t = 0; t < TILE_X / 16; t++)
            anyGeometry |= InBounds(P.tiles, tileX0 + t, tileY) && LoadTileX(P.tiles, tileX0 + t, tileY) == 0.0f;
        if (!anyGeometry)
            return;
        const int baseX = blockIdx.x * TILE_X - BORDER, baseY = blockY * TILE_Y - BORDER;
        for (int i = threadIdx.x; i < BUF_X * BUF_Y; i += TILE_X * TILE_Y) {
            int lx = i % BUF_X, ly = i / BUF_X;
            int gx = ClampI(baseX + lx, 0, rw), gy = ClampI(baseY + ly, 0, rh);
            s_Shadow[ly * BUF_STRIDE + lx] = UnpackShadow(ST::Load(P.shadow, gx, gy));
            s_Penumbra[ly * BUF_STRIDE + lx] = LoadR16F(P.penumbra, gx, gy);
        }
    }
    __syncthreads();

    if (px > rw || py > rh)
        return;
    const int so = (ty + BORDER) * BUF_STRIDE + tx + BORDER;
    const float centerPenumbra = s_Penumbra[so];
    const float viewZ = UnpackViewZ(c, LoadR32F(P.viewZ, px, py));
    if (LoadTileX(P.tiles, px >> 4, py >> 4) != 0.0f || viewZ > c.gDenoisingRange)
        return;

    const float2 rectSizeInv = ToF2(c.gRectSizeInv), rectSizePrev = ToF2(c.gRectSizePrev), resolutionScale = ToF2(c.gResolutionScale);
    const float3 cameraDelta = ToF3(c.gCameraDelta);

    float2 pixelUv = F2(float(px) + 0.5f, float(py) + 0.5f) * rectSizeInv;
    float tileValue = TextureCubicY(P.tiles, pixelUv * resolutionScale);
    bool isHardShadow = tileValue == 0.0f || centerPenumbra == 0.0f;
    if (isHardShadow) {
        ST::Store(P.outShadow, px, py, PackShadow(s_Shadow[so]));
        StoreR32U(P.outHistoryLength, px, py, PackViewZAndHistoryLength(viewZ, SIGMA_MAX_ACCUM_FRAME_NUM));
        return;
    }

    float sumw = 0.0f;
    S m1 = ST::Splat(0.0f), m2 = ST::Splat(0.0f), input = ST::Splat(0.0f);
#pragma unroll
    for (int j = 0; j <= BORDER * 2; j++) {
#pragma unroll
        for (int i = 0; i <= BORDER * 2; i++) {
            const int o = (ty + j) * BUF_STRIDE + tx + i;
            S s = s_Shadow[o];
            float w = 1.0f;
            if (i == BORDER && j == BORDER)
                input = s;
            else {
                float penum = s_Penumbra[o];
                w = AreBothLitOrUnlit(centerPenumbra, penum);
                w *= GetGaussianWeight(Length(Div(F2(float(i - BORDER), float(j - BORDER)), float(BORDER))));
            }
            m1 = Mad(s, w, m1);
            m2 = m2 + s * s * w;
            sumw += w;
        }
    }
    m1 = Div(m1, sumw);
    m2 = Div(m2, sumw);
    S sigma = StdDev(m1, m2);

    float3 Xv = ReconstructViewPosition(pixelUv, ToF4(c.gFrustum), viewZ, c.gOrthoMode);
    float3 X = RotateVectorInverse(c.gWorldToView, Xv);

    float4 mvRaw = LoadRGBA16F(P.mv, px, py);
    float3 mv = F3(mvRaw.x, mvRaw.y, mvRaw.z) * F3(c.gMvScale.x, c.gMvScale.y, c.gMvScale.z);
    float3 Xprev = X;
    float2 smbPixelUv = pixelUv + F2(mv.x, mv.y);
    if (c.gMvScale.w == 0.0f) {
        if (c.gMvScale.z == 0.0f)
            mv.z = AffineTransform(c.gWorldToViewPrev, X).z - viewZ;
        float viewZprev = viewZ + mv.z;
        float3 Xvprevlocal = ReconstructViewPosition(smbPixelUv, ToF4(c.gFrustumPrev), viewZprev, c.gOrthoMode);
        Xprev = RotateVectorInverse(c.gWorldToViewPrev, Xvprevlocal) + cameraDelta;
    } else {
        Xprev = Xprev + mv;
        smbPixelUv = GetScreenUv(c.gWorldToClipPrev, Xprev);
    }

    TrackHistoryReach(historyReach, HistoryReachRows(smbPixelUv.y, rectSizePrev.y, py)); // (multi-GPU hosts; a null word otherwise: reblur_device.h)
    Bilinear smbBilinearFilter = GetBilinearFilter(smbPixelUv, rectSizePrev);
    const int bx = (int)smbBilinearFilter.origin.x, by = (int)smbBilinearFilter.origin.y;
    uint32_t d0 = FetchClampedR32U(P.historyLength, bx, by), d1 = FetchClampedR32U(P.historyLength, bx + 1, by), d2 = FetchClampedR32U(P.historyLength, bx, by + 1),
             d3 = FetchClampedR32U(P.historyLength, bx + 1, by + 1);
    float4 prevViewZ = F4(AsFloat(d0 & ~7u), AsFloat(d1 & ~7u), AsFloat(d2 & ~7u), AsFloat(d3 & ~7u));
    float4 prevHistoryLength = F4(float(d0 & 7u), float(d1 & 7u), float(d2 & 7u), float(d3 & 7u));

    float frustumSize = GetFrustumSize(c.gMinRectDimMulUnproject, c.gOrthoMode, viewZ);
    float disocclusionThreshold = GetDisocclusionThreshold(NRD_DISOCCLUSION_THRESHOLD, frustumSize, 1.0f);
    disocclusionThreshold *= IsInScreenNearest(smbPixelUv);
    disocclusionThreshold -= NRD_EPS;

    float3 Xvprev = AffineTransform(c.gWorldToViewPrev, Xprev);
    float4 smbPlaneDist = Abs(prevViewZ - Xvprev.z);
    float4 smbOcclusion = Step(smbPlaneDist, F4(disocclusionThreshold));

    float4 smbOcclusionWeights = GetBilinearCustomWeights(smbBilinearFilter, smbOcclusion);
    float historyLength = ApplyBilinearCustomWeights(prevHistoryLength.x, prevHistoryLength.y, prevHistoryLength.z, prevHistoryLength.w, smbOcclusionWeights);

    bool isCatRomAllowed = Sum(smbOcclusionWeights) > 3.5f; // never true (weights sum to <= 1): kept as in the reference
    HistoryFilter hf = MakeHistoryFilter(Sat(smbPixelUv) * rectSizePrev, smbOcclusionWeights, isCatRomAllowed, P.history);
    S history = FetchShadowHistory<TRANSLUCENT>(hf, P.history);
    history = SatV(history);
    history = UnpackShadow(history);

    sigma = sigma * Lerp(SIGMA_TS_SIGMA_SCALE, 1.0f, Rcp(1.0f + historyLength));
    S inputMin = m1 - sigma, inputMax = m1 + sigma;
    S historyClamped = ClampV(history, inputMin, inputMax);

    float antilag = Abs(ST::X(historyClamped) - ST::X(history)); // the shadow channel drives the antilag
    antilag = Sqrt01(antilag);
    antilag = Sat(1.0f - antilag);
    historyLength *= antilag;

    float historyWeight = Div(historyLength, 1.0f + historyLength);
    float streetMagic = 0.6f * historyWeight * antilag;
    historyClamped = Lerp(historyClamped, history, streetMagic);

    S result = Lerp(input, historyClamped, Min(c.gStabilizationStrength, historyWeight));
    historyLength = Min(historyLength + 1.0f, SIGMA_MAX_ACCUM_FRAME_NUM);

    ST::Store(P.outShadow, px, py, PackShadow(result));
    StoreR32U(P.outHistoryLength, px, py, PackViewZAndHistoryLength(viewZ, historyLength));
}

template <bool TRANSLUCENT>
static const char* LaunchTemporalStabilization(const PassArgs& a) {
    const SigmaCB& c = *(const SigmaCB*)a.constants;
    if (const char* err = CheckSupportedSigma(c))
        return err;
    if (a.planesNum != 9)
        return "SIGMA temporal stabilization: unexpected resource count";
    TsPlanes P = {a.planes[0], a.planes[1], a.planes[2], a.planes[3], a.planes[4], a.planes[5], a.planes[6], a.planes[7], a.planes[8]};
    dim3 grid = GridFor(c.gRectSizeMinusOne.x + 1, c.gRectSizeMinusOne.y + 1, TILE_X, TILE_Y);
    const RowBand band = MakeRowBand(a, c.gRectSizeMinusOne.y + 1, TILE_Y);
    grid.y = band.blocksY;
    LaunchPass(a, SigmaTemporalStabilizationKernel<TRANSLUCENT>, grid, dim3(TILE_X * TILE_Y), c, P, band.blockY0, a.historyReachWord);
    return nullptr;
}

// ================================================================================================ SplitScreen
template <bool TRANSLUCENT>
__global__ __launch_bounds__(256) void SigmaSplitScreenKernel(SigmaCB c, Plane viewZ, Plane penumbra, Plane translucency, Plane outShadow, int blockY0) {
    typedef SigmaType<TRANSLUCENT> ST;
    const int px = blockIdx.x * TILE_X + (threadIdx.x % TILE_X), py = ((int)blockIdx.y + blockY0) * TILE_Y + (threadIdx.x / TILE_X);
    if (px > c.gRectSizeMinusOne.x || py > c.gRectSizeMinusOne.y)
        return;
    float pixelUvX = (float(px) + 0.5f) * c.gRectSizeInv.x;
    if (pixelUvX > c.gSplitScreen)
        return;
    float z = UnpackViewZ(c, LoadR32F(viewZ, px, py));
    typename ST::type s;
    if (TRANSLUCENT)
        s = ST::Load(translucency, px, py);
    else
        s = ST::Splat(IsLit(LoadR16F(penumbra, px, py)) ? 1.0f : 0.0f);
    ST::Store(outShadow, px, py, s * (z < c.gDenoisingRange ? 1.0f : 0.0f));
}

template <bool TRANSLUCENT>
static const char* LaunchSplitScreen(const PassArgs& a) {
    const SigmaCB& c = *(const SigmaCB*)a.constants;
    if (a.planesNum != (TRANSLUCENT ? 4u : 3u))
        return "SIGMA split screen: unexpected resource count";
    dim3 grid = GridFor(c.gRectSizeMinusOne.x + 1, c.gRectSizeMinusOne.y + 1, TILE_X, TILE_Y);
    const RowBand band = MakeRowBand(a, c.gRectSizeMinusOne.y + 1, TILE_Y);
    grid.y = band.blocksY;
    LaunchPass(a, SigmaSplitScreenKernel<TRANSLUCENT>, grid, dim3(256), c, a.planes[0], a.planes[1], TRANSLUCENT ? a.planes[2] : Plane{}, a.planes[a.planesNum - 1], band.blockY0);
    return nullptr;
}

const PassEntry* GetSigmaPasses(uint32_t& num) {
    static const PassEntry k[] = {
        {"SIGMA_Shadow_ClassifyTiles.cs", LaunchClassifyTiles<false>},
        {"SIGMA_SmoothTiles.cs", LaunchSmoothTiles},
        {"SIGMA_Copy.cs", LaunchCopy},
        {"SIGMA_Shadow_Blur.cs", LaunchBlur<true, false>},
        {"SIGMA_Shadow_PostBlur.cs", LaunchBlur<false, false>},
        {"SIGMA_Shadow_TemporalStabilization.cs", LaunchTemporalStabilization<false>},
        {"SIGMA_Shadow_SplitScreen.cs", LaunchSplitScreen<false>},
        {"SIGMA_ShadowTranslucency_ClassifyTiles.cs", LaunchClassifyTiles<true>},
        {"SIGMA_ShadowTranslucency_Blur.cs", LaunchBlur<true, true>},
        {"SIGMA_ShadowTranslucency_PostBlur.cs", LaunchBlur<false, true>},
        {"SIGMA_ShadowTranslucency_TemporalStabilization.cs", LaunchTemporalStabilization<true>},
        {"SIGMA_ShadowTranslucency_SplitScreen.cs", LaunchSplitScreen<true>},
    };
    num = sizeof(k) / sizeof(k[0]);
    return k;
}

} // namespace nrdhip
