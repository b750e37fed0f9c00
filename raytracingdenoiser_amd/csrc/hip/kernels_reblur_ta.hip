// REBLUR TemporalAccumulation as a HIP kernel for gfx950.
//   reference Shaders/Include/REBLUR_TemporalAccumulation.hlsli:11-931 (the library's normal encoding: nrdmath.h NRD_NORMAL_ENCODING)
//
// MI355X mapping. 32x8-pixel workgroups (4 waves of 2 rows x 32 px). The 3x3 neighbourhood statistics (averaged
// normal, roughness variance, minimum hit distance for tracking) and the two curvature edge normals come from a
// 34x10 LDS tile holding the UNPACKED normal+roughness (float4) and the tracking hit distance, filled cooperatively
// with border-clamped loads -- every texel is decoded once per workgroup instead of 9+2 times per pixel. The row stride
// of the float4 tile is 35 slots (560 B): rows of a wave start on different 16-byte slots, so the ds_read_b128 taps
// of the two rows do not collide. Everything after that is per-pixel: reprojection, the 4x4 previous-depth footprint,
// disocclusion tests, virtual-motion tracking, Catmull-Rom history fetches (12 fp16 texels per plane) and the accumulation
// itself. The kernel is bound by VALU issue, not by HBM and (since the window kernel, MODE 1 below) not by latency either:
// ~94 B/px of compulsory traffic against ~3 700 executed VALU instructions per wave of denoised pixels, its SIMDs 91 % busy
// with them (profiles/r03_i_reblur_ds_pmc3.txt).
#include "passes.h"
#include <climits>
#include <cstdio>
#include "reblur_device.h"

namespace nrdhip {

constexpr int TILE_X = 32;
constexpr int TILE_Y = 8;
constexpr int BORDER = 1;
constexpr int BUF_X = TILE_X + 2 * BORDER; // 34
constexpr int BUF_Y = TILE_Y + 2 * BORDER; // 10
constexpr int BUF_STRIDE = BUF_X + 1;      // 35 float4 slots per row

// The surface-motion window (MODE 1, below): the texels of the previous frame a workgroup's pixels reproject to, staged in LDS
#ifndef NRD_TA_WIN_H
#define NRD_TA_WIN_W 64 // at most 64: one lane per column when the window is filled
#define NRD_TA_WIN_H 16
#endif
constexpr int WIN_W = NRD_TA_WIN_W;
constexpr int WIN_H = NRD_TA_WIN_H;

struct TaPlanes {
    uint32_t* historyReach; // passes.h PassArgs::historyReachWord (multi-GPU hosts; nullptr otherwise)
    Plane tileFlags; // executor scratch, one byte per workgroup tile (passes.h): set by the window kernel for the tiles it leaves to the fallback kernel
    int winMaxW, winMaxH; // largest box the window kernel accepts (<= WIN_W x WIN_H; smaller values exercise the fallback kernel: NRD_HIP_TA_WINDOW_LIMIT)
    Plane tiles, normalRoughness, viewZ, mv, prevViewZ, prevNormalRoughness, prevInternalData;
    NormalRoughnessGuide decodedNR; // executor's decoded guides of IN_NORMAL_ROUGHNESS (reblur_device.h NormalRoughnessGuide)
    Plane disocclusionThresholdMix, diffConfidence, specConfidence; // R8_UNORM user inputs; dummies unless the gHas* flags are set
    Plane inDiff, inSpec, historyDiff, historySpec, historyDiffFast, historySpecFast, prevSpecHitDistForTracking, inSpecHitDistForTracking;
    Plane outDiff, outSpec, outDiffFast, outSpecFast, outSpecHitDistForTracking, outData1, outData2;
    Plane inDiffSh, inSpecSh, historyDiffSh, historySpecSh, outDiffSh, outSpecSh; // SH family (RGBA16F)
};

template <typename T>
NRD_D void WindowHistoryTexels(const HistoryFilter&, const uint2*, int, int, int, T&) {} // other storage kinds have no window kernel
// the 2x2 of the Load-based bilinear path (texels outside the plane read as 0); `shift` selects the half of the window dword (0 diffuse, 16 specular)
NRD_D void WindowFastTexels(const HistoryFilter& h, const uint32_t* win, int wx0, int wy0, const Plane& dims, int shift, BilinearTexelsR16F& t) {
    const int x0 = ClampI(h.ox, 0, dims.w - 1) - wx0, x1 = ClampI(h.ox + 1, 0, dims.w - 1) - wx0;
    const int y0 = (ClampI(h.oy, 0, dims.h - 1) - wy0) * WIN_W, y1 = (ClampI(h.oy + 1, 0, dims.h - 1) - wy0) * WIN_W;
    const uint32_t f00 = InBounds(dims, h.ox, h.oy) ? (win[y0 + x0] >> shift) & 0xFFFFu : 0u, f10 = InBounds(dims, h.ox + 1, h.oy) ? (win[y0 + x1] >> shift) & 0xFFFFu : 0u;
    const uint32_t f01 = InBounds(dims, h.ox, h.oy + 1) ? (win[y1 + x0] >> shift) & 0xFFFFu : 0u, f11 = InBounds(dims, h.ox + 1, h.oy + 1) ? (win[y1 + x1] >> shift) & 0xFFFFu : 0u;
    t.r0 = f00 | (f10 << 16);
    t.r1 = f01 | (f11 << 16);
    t.loaded = true;
}
template <typename T>
NRD_D void WindowFastTexels(const HistoryFilter&, const uint32_t*, int, int, const Plane&, int, T&) {}


// PERF = REBLUR_PERFORMANCE_MODE: no Catmull-Rom history fetches (REBLUR_USE_CATROM_FOR_*_MOTION_IN_TA = 0, REBLUR_Config.hlsli:196-201)
// OCC = occlusion family (REBLUR_OCCLUSION): hit-distance-only signals in R16_UNORM, no pre-pass output to read, no DATA2, no firefly suppressor
// SH = the *_SH denoisers: the SH1 plane of every signal is accumulated with the same speeds (custom-weight bilinear history fetch)
// WAVES = waves per SIMD the register allocation aims at (__launch_bounds__): 2 for the kernels with a specular signal (3 would need scratch), 3 for
// the diffuse-only ones (NRD_HIP_TA_WAVES overrides the choice at launch for A/B runs; measurements in DESIGN.md section 3)
//
// MODE 1 = "window" kernel. Everything the pass reads at the surface-motion position -- the 4x4 previous-depth / internal-data footprint, the 2x2 previous
// normals, the 12 + 12 history texels of the two signals and their fast histories: 320 bytes per pixel through the L1, ~90 VGPRs of requests in flight, and
// the reason the kernel cannot hold a third wave -- comes from ONE rectangle of the previous frame per workgroup, because neighbouring pixels reproject to
// neighbouring texels: the bounding box of the (clamped) texel coordinates its pixels need is reduced over the workgroup, the rectangle is copied into LDS
// with coalesced row loads (each texel once: ~1.7 texels per pixel instead of 44), and the per-pixel code reads the very same texels from there, right
// where they are used. Results are bit-identical by construction (same texels, same arithmetic). A workgroup whose box does not fit WIN_W x WIN_H
// (a silhouette with large parallax, a jump of the camera) writes 1 into its byte of P.tileFlags and leaves; MODE 2, the unchanged global-memory kernel
// behind a flag test, is launched right after and processes exactly those tiles. MODE 0 = the plain kernel (all other signal kinds / the performance mode).
// The pass for one 32x8 tile (tile column tileX, tile row blockY of the frame); the kernel below is a thin wrapper
template <bool DIFF, bool SPEC, bool PERF, int KIND, bool SH, int MODE>
__device__ __forceinline__ void ReblurTemporalAccumulationTile(const ReblurCB& cArg, TaPlanes P, const RowRange& rr, const int tileX, const int blockY) {
    typedef ReblurSignal<KIND> Sig;
    constexpr bool OCC = KIND == SIGNAL_OCCLUSION;
    typedef typename Sig::type S;
    __shared__ float4 s_Normal_Roughness[BUF_Y * BUF_STRIDE];
    // The 832-byte constant block + ~20 planes need > 200 SGPRs (102 exist), which the compiler resolves by spilling scalars into
    // VGPR lanes (v_writelane / v_readlane + hazard nops on every use). The body therefore reads the constants from an LDS copy
    // (uniform-address ds_read, off the VALU); only the prologue touches the kernel-argument copy.
    __shared__ ReblurCB s_Constants;

    // SGPR diet (planes.h): one (w, h) for every full-resolution plane, one pitch per pool format; verified by the launcher
    {
        const Plane size = P.viewZ, rgba16 = DIFF ? P.historyDiff : P.historySpec, r16 = DIFF ? P.historyDiffFast : P.historySpecFast;
        ShareSize(P.decodedNR, size), ShareSize(P.mv, size), ShareSize(P.prevViewZ, size), ShareSize(P.prevNormalRoughness, size), ShareSize(P.prevInternalData, size);
        ShareSize(P.inDiff, size), ShareSize(P.inSpec, size), ShareSize(P.inSpecHitDistForTracking, size), ShareSize(P.outData1, size), ShareSize(P.outData2, size);
        if (OCC) { // the history planes are the user's OUT_*_HITDIST (own pitch); only the pool outputs share a layout
            const Plane sig = DIFF ? P.outDiff : P.outSpec;
            ShareSize(P.historyDiff, size), ShareSize(P.historySpec, size), ShareLayout(P.outDiff, sig), ShareLayout(P.outSpec, sig);
        } else {
            ShareLayout(P.historyDiff, rgba16), ShareLayout(P.historySpec, rgba16), ShareLayout(P.outDiff, rgba16), ShareLayout(P.outSpec, rgba16);
        }
        ShareLayout(P.historyDiffFast, r16), ShareLayout(P.historySpecFast, r16), ShareLayout(P.prevSpecHitDistForTracking, r16), ShareLayout(P.outDiffFast, r16), ShareLayout(P.outSpecFast, r16),
            ShareLayout(P.outSpecHitDistForTracking, r16);
    }
    __shared__ float s_HitDistForTracking[BUF_Y * BUF_STRIDE];
    constexpr int WIN_TEXELS = MODE == 1 ? WIN_W * WIN_H : 1;
    __shared__ float s_WinZ[WIN_TEXELS];       // packed previous viewZ
    __shared__ NrRaw s_WinN[WIN_TEXELS];       // packed previous normal / roughness
    __shared__ uint32_t s_WinId[WIN_TEXELS];   // previous internal data (16 bits)
    __shared__ uint32_t s_WinFast[WIN_TEXELS]; // fast histories: diffuse in the low, specular in the high half
    __shared__ uint2 s_WinDiff[WIN_TEXELS], s_WinSpec[WIN_TEXELS]; // RGBA16F history texels, undecoded
    __shared__ int s_WinBox[4][4];

    const int tx = threadIdx.x % TILE_X, ty = threadIdx.x / TILE_X;
    // (workgroups of the XCD-aware grid may lie beyond the frame: they have no pixels and no flag)
    uint8_t* const tileFlag = (MODE == 1 && tileX < P.tileFlags.w && blockY < P.tileFlags.h) ? P.tileFlags.ptr + (uint32_t)blockY * P.tileFlags.pitch + (uint32_t)tileX : nullptr;
    const int px = tileX * TILE_X + tx, py = blockY * TILE_Y + ty;
    const int rw = cArg.gRectSizeMinusOne.x, rh = cArg.gRectSizeMinusOne.y;

    // The pixel's own guides do not depend on the LDS tile: they are requested in FRONT of the fill (below, behind the uniform sky test), so that one memory latency
    // covers both; behind the barrier they were a second exposed latency in front of the window fill / the history footprints (the pass ran 19 % above the time of
    // a build whose loads all hit the L1, profiles/r04_c_reblur_ds_uniform_*_kernel_stats.txt)
    float preTile = 0.0f, preViewZ = 0.0f, preMaterialID = 0.0f;
    float4 preMv = F4(0.0f), preNormalAndRoughness = F4(0.0f);
    // ---- cooperative preload (clamped to the rect), skipped when every 16x16 tile under this block is sky
    {
        const int tileY = (blockY * TILE_Y) >> 4, tileX0 = (tileX * TILE_X) >> 4;
        bool anyGeometry = false;
        for (int t = 0; t < TILE_X / 16; t++)
            if (tileX0 + t < P.tiles.w && tileY < P.tiles.h)
                anyGeometry |= LoadR8Unorm(P.tiles, tileX0 + t, tileY) == 0.0f;
        if (!anyGeometry) {
            if (MODE == 1 && threadIdx.x == 0 && tileFlag)
                *tileFlag = 0;
            return; // uniform across the block
        }

        if (MODE == 1) { // (the plain kernels sit at their register budget: there the guides are requested behind the barrier as before)
            const int qx = min(px, rw), qy = min(max(py, 0), rh);
            preTile = LoadR8Unorm(P.tiles, qx >> 4, qy >> 4);
            preViewZ = LoadR32F(P.viewZ, qx, qy);
            preMv = LoadRGBA16F(P.mv, qx, qy);
            preNormalAndRoughness = LoadDecodedNormalRoughness(P.decodedNR, qx, qy, preMaterialID);
        }
        const int baseX = tileX * TILE_X - BORDER, baseY = blockY * TILE_Y - BORDER;
        for (int i = threadIdx.x; i < BUF_X * BUF_Y; i += TILE_X * TILE_Y) {
            int lx = i % BUF_X, ly = i / BUF_X;
            int gx = ClampI(baseX + lx, 0, rw), gy = ClampI(baseY + ly, 0, rh);
            s_Normal_Roughness[ly * BUF_STRIDE + lx] = LoadDecodedNormalRoughness(P.decodedNR, gx, gy);
            if (SPEC) {
                const int shift = (OCC && cArg.gSpecCheckerboard != 2) ? 1 : 0; // checkerboarded occlusion input: left half (reference REBLUR_TemporalAccumulation.hlsli:21-27)
                float hitDist = (OCC || cArg.gSpecPrepassBlurRadius == 0.0f) ? ExtractHitDist(Sig::Load(P.inSpec, gx >> shift, gy)) : LoadR16F(P.inSpecHitDistForTracking, gx, gy);
                s_HitDistForTracking[ly * BUF_STRIDE + lx] = hitDist == 0.0f ? NRD_INF : hitDist;
            }
        }
        // constants: kernel-argument segment (cArg is the first argument, offset 0) -> LDS, one dword per thread
        const uint32_t __attribute__((address_space(4)))* kernarg = (const uint32_t __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
        if (threadIdx.x < sizeof(ReblurCB) / 4)
            ((uint32_t*)&s_Constants)[threadIdx.x] = kernarg[threadIdx.x];
    }
    __syncthreads();
    const ReblurCB& c = s_Constants;
// Constants are re-read from LDS after each phase boundary instead of being kept in VGPRs for the whole kernel (a compiler
// memory barrier: the LDS loads cannot be hoisted above it)
#define NRD_CONSTANTS_PHASE() asm volatile("" ::: "memory")

    // MODE 1: every thread stays until the window is filled. A thread without a pixel to denoise runs the prologue on a position clamped into the rect
    // (lpx, lpy: its loads stay legal, its values are never used), stores nothing and is left out of the bounding box.
    const int lpx = MODE == 1 ? min(px, rw) : px, lpy = MODE == 1 ? min(py, rh) : py;
    bool active = !(px > rw || py > rh || py < rr.rowBegin || py >= rr.rowEnd);
    if (MODE != 1 && !active)
        return;
    if (MODE != 1) {
        preTile = LoadR8Unorm(P.tiles, lpx >> 4, lpy >> 4);
        if (preTile != 0.0f)
            return;
        preViewZ = LoadR32F(P.viewZ, lpx, lpy);
    }
    active = active && preTile == 0.0f; // (MODE 1: the prefetch position is (lpx, lpy) for every thread that stays)
    const float viewZ = UnpackViewZ(c, preViewZ);
    active = active && !(viewZ > c.gDenoisingRange);
    if (MODE != 1 && !active)
        return;

    const float2 rectSize = ToF2(c.gRectSize), rectSizeInv = ToF2(c.gRectSizeInv), rectSizePrev = ToF2(c.gRectSizePrev);
    const float3 cameraDelta = ToF3(c.gCameraDelta);
    const float4 frustum = ToF4(c.gFrustum), frustumPrev = ToF4(c.gFrustumPrev), hitDistParams = ToF4(c.gHitDistParams);

    // Current position
    float2 pixelUv = F2(float(px) + 0.5f, float(py) + 0.5f) * rectSizeInv;
    float3 Xv = ReconstructViewPosition(pixelUv, frustum, viewZ, NRD_ORTHO_MODE(c));
    float3 X = RotateVector(c.gViewToWorld, Xv);

    // 3x3: averaged normal (2x2 part), roughness moments, min hit distance for tracking
    float3 Navg = F3(0.0f);
    float hitDistForTracking = NRD_INF, roughnessM1 = 0.0f, roughnessM2 = 0.0f;
#pragma unroll
    for (int j = 0; j <= 2; j++) {
#pragma unroll
        for (int i = 0; i <= 2; i++) {
            int o = (ty + j) * BUF_STRIDE + tx + i;
            float4 nr = s_Normal_Roughness[o];
            if (i < 2 && j < 2)
                Navg = Navg + Xyz(nr);
            if (SPEC) {
                hitDistForTracking = Min(hitDistForTracking, s_HitDistForTracking[o]);
                float roughnessSq = nr.w * nr.w;
                roughnessM1 += roughnessSq;
                roughnessM2 += roughnessSq * roughnessSq;
            }
        }
    }
    Navg = Navg * 0.25f;

    if (MODE != 1)
        preNormalAndRoughness = LoadDecodedNormalRoughness(P.decodedNR, lpx, lpy, preMaterialID);
    float materialID = preMaterialID;
    float4 normalAndRoughness = preNormalAndRoughness;
    float3 N = Xyz(normalAndRoughness);
    float roughness = normalAndRoughness.w;

    float roughnessModified = 0.0f, roughnessSigma = 0.0f, hitDistNormalization = 0.0f;
    RngHash rng;
    if (SPEC) {
        roughnessModified = GetModifiedRoughnessFromNormalVariance(roughness, Navg);
        roughnessM1 = roughnessM1 * (1.0f / 9.0f);
        roughnessM2 = roughnessM2 * (1.0f / 9.0f);
        roughnessSigma = Sqrt(Abs(roughnessM2 - roughnessM1 * roughnessM1));

        rng.Initialize((uint32_t)px, (uint32_t)py, c.gFrameIndex);

        hitDistForTracking = hitDistForTracking == NRD_INF ? 0.0f : hitDistForTracking;
        hitDistNormalization = GetHitDistanceNormalization(viewZ, hitDistParams, roughness);
        hitDistForTracking *= (OCC || c.gSpecPrepassBlurRadius == 0.0f) ? hitDistNormalization : 1.0f;
        if (active)
            StoreR16F(P.outSpecHitDistForTracking, px, py, hitDistForTracking);
    }

    NRD_CONSTANTS_PHASE();
    // Previous position and surface motion uv
    float4 mvRaw = MODE == 1 ? preMv : LoadRGBA16F(P.mv, lpx, lpy);
    float3 mv = F3(mvRaw.x, mvRaw.y, mvRaw.z) * F3(c.gMvScale.x, c.gMvScale.y, c.gMvScale.z);
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

    TrackHistoryReach(P.historyReach, active ? HistoryReachRows(smbPixelUv.y, rectSizePrev.y, py) : 0.0f); // (multi-GPU hosts; a null word otherwise)

    // Previous viewZ: 4x4 footprint as four 2x2 quads in (0,0)(1,0)(0,1)(1,1) order
    float2 catromOrigin = GetCatmullRomOrigin(smbPixelUv, rectSizePrev);
    const int cx = (int)catromOrigin.x, cy = (int)catromOrigin.y;
    Bilinear smbBilinearFilter = GetBilinearFilter(smbPixelUv, rectSizePrev);
    const int bx = (int)smbBilinearFilter.origin.x, by = (int)smbBilinearFilter.origin.y;
    const float2 smbSamplePos = Sat(smbPixelUv) * rectSizePrev;
    HistoryFilter smbFilter = MakeHistoryGeometry(smbSamplePos, DIFF ? P.historyDiff : P.historySpec); // both histories share a layout (checked by the launcher)
    typename Sig::HistoryTexels smbDiffTexels, smbSpecTexels;
    typename Sig::FastTexels smbDiffFastTexels, smbSpecFastTexels;
    S diff = Sig::Zero(), spec = Sig::Zero();
    float4 smbViewZ0, smbViewZ1, smbViewZ2, smbViewZ3;
    uint32_t id0[4], id1[4], id2[4], id3[4];
    NrRaw n00, n10, n01, n11; // packed texels of the 2x2 normal footprint (0 outside the plane, as Load returns)
    int wx0 = 0, wy0 = 0;        // MODE 1: plane coordinates of the window's first texel
    if (MODE == 1) {
        // ---- the window: bounding box of the clamped texel coordinates this workgroup's pixels read at the surface-motion position
        const int W1 = P.prevViewZ.w - 1, H1 = P.prevViewZ.h - 1; // every plane staged here has the size of prevViewZ (checked by the launcher)
        int loX = INT_MAX, loY = INT_MAX, hiX = INT_MIN, hiY = INT_MIN;
        if (active) {
            loX = min(min(ClampI(cx, 0, W1), ClampI(bx, 0, W1)), min(smbFilter.x[0], ClampI(smbFilter.ox, 0, W1)));
            hiX = max(max(ClampI(cx + 3, 0, W1), ClampI(bx + 1, 0, W1)), max(smbFilter.x[3], ClampI(smbFilter.ox + 1, 0, W1)));
            loY = min(min(ClampI(cy, 0, H1), ClampI(by, 0, H1)), min(smbFilter.y[0], ClampI(smbFilter.oy, 0, H1)));
            hiY = max(max(ClampI(cy + 3, 0, H1), ClampI(by + 1, 0, H1)), max(smbFilter.y[3], ClampI(smbFilter.oy + 1, 0, H1)));
        }
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
        // ---- fill: one wave per row of the box, one lane per column -- coalesced row segments, every texel once
        if (lane < bw)
            for (int r = wave; r < bh; r += 4) {
                const int x = wx0 + lane, y = wy0 + r, o = r * WIN_W + lane;
                s_WinZ[o] = LoadR32F(P.prevViewZ, x, y);
                s_WinN[o] = LoadNrRaw(P.prevNormalRoughness, x, y);
                s_WinId[o] = LoadR16U(P.prevInternalData, x, y);
                s_WinFast[o] = (DIFF ? LoadR16U(P.historyDiffFast, x, y) : 0u) | (SPEC ? LoadR16U(P.historySpecFast, x, y) << 16 : 0u);
                if (DIFF)
                    s_WinDiff[o] = *TexelPtr<const uint2>(P.historyDiff, x, y);
                if (SPEC)
                    s_WinSpec[o] = *TexelPtr<const uint2>(P.historySpec, x, y);
            }
        __syncthreads();
        if (!active)
            return;
        // ---- the footprints, read where the plain kernel has its request batch: texel (i, j) of a footprint is the texel at the clamped coordinate
        int xo[4], yo[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
            xo[i] = ClampI(cx + i, 0, W1) - wx0, yo[i] = (ClampI(cy + i, 0, H1) - wy0) * WIN_W;
#define WIN_Z(i, j) s_WinZ[yo[j] + xo[i]]
#define WIN_ID(i, j) s_WinId[yo[j] + xo[i]]
        smbViewZ0 = F4(WIN_Z(0, 0), WIN_Z(1, 0), WIN_Z(0, 1), WIN_Z(1, 1)), smbViewZ1 = F4(WIN_Z(2, 0), WIN_Z(3, 0), WIN_Z(2, 1), WIN_Z(3, 1));
        smbViewZ2 = F4(WIN_Z(0, 2), WIN_Z(1, 2), WIN_Z(0, 3), WIN_Z(1, 3)), smbViewZ3 = F4(WIN_Z(2, 2), WIN_Z(3, 2), WIN_Z(2, 3), WIN_Z(3, 3));
        id0[0] = WIN_ID(0, 0), id0[1] = WIN_ID(1, 0), id0[2] = WIN_ID(0, 1), id0[3] = WIN_ID(1, 1);
        id1[0] = WIN_ID(2, 0), id1[1] = WIN_ID(3, 0), id1[2] = WIN_ID(2, 1), id1[3] = WIN_ID(3, 1);
        id2[0] = WIN_ID(0, 2), id2[1] = WIN_ID(1, 2), id2[2] = WIN_ID(0, 3), id2[3] = WIN_ID(1, 3);
        id3[0] = WIN_ID(2, 2), id3[1] = WIN_ID(3, 2), id3[2] = WIN_ID(2, 3), id3[3] = WIN_ID(3, 3);
#undef WIN_Z
#undef WIN_ID
        const int nx0 = ClampI(bx, 0, W1) - wx0, nx1 = ClampI(bx + 1, 0, W1) - wx0, ny0 = (ClampI(by, 0, H1) - wy0) * WIN_W, ny1 = (ClampI(by + 1, 0, H1) - wy0) * WIN_W;
        n00 = InBounds(P.prevNormalRoughness, bx, by) ? s_WinN[ny0 + nx0] : NrRawZero();
        n10 = InBounds(P.prevNormalRoughness, bx + 1, by) ? s_WinN[ny0 + nx1] : NrRawZero();
        n01 = InBounds(P.prevNormalRoughness, bx, by + 1) ? s_WinN[ny1 + nx0] : NrRawZero();
        n11 = InBounds(P.prevNormalRoughness, bx + 1, by + 1) ? s_WinN[ny1 + nx1] : NrRawZero();
    } else {
        const bool footprintInterior = FootprintIsInterior(P.prevViewZ, cx, cy, 4, 4); // the four rows as four 16-byte loads (reblur_device.h "row-vector fetches")
        // Row loads from an origin clamped into the plane: always legal (pool planes have >= 4 texels per row pitch), exact for an interior footprint. The
        // few footprints that touch the border are re-read texel by texel AFTER the batch. (Two symmetric arms -- row loads / clamped loads -- would be
        // merged by the compiler into twelve scalar loads with selected addresses, which is what the row loads are there to avoid.)
        const int fx4 = max(0, min(cx, P.prevViewZ.w - 4));
        const int fy0 = ClampI(cy, 0, P.prevViewZ.h - 1), fy1 = ClampI(cy + 1, 0, P.prevViewZ.h - 1), fy2 = ClampI(cy + 2, 0, P.prevViewZ.h - 1), fy3 = ClampI(cy + 3, 0, P.prevViewZ.h - 1);
        const float4 zr0 = LoadRowR32Fx4(P.prevViewZ, fx4, fy0), zr1 = LoadRowR32Fx4(P.prevViewZ, fx4, fy1), zr2 = LoadRowR32Fx4(P.prevViewZ, fx4, fy2), zr3 = LoadRowR32Fx4(P.prevViewZ, fx4, fy3);
        // ---- every other request that depends only on the surface-motion position is issued here, in one batch with the depth footprint: the 2x2
        // previous normals, the 4x4 previous internal data, the history texels of both signals (blended once the occlusion weights exist) and the
        // noisy inputs. A wave of this kernel lives ~27 000 cycles of which ~14 000 were spent waiting on ~14 dependent request phases at 2 waves
        // per SIMD (profiles/r02_c_reblur_ds_sq_pmc1.txt); the arithmetic in between now runs while the next phase's data is in flight.
        // same geometry as the prev-viewZ footprint (launcher: same plane size): four undecoded 8-byte rows
        const uint2 ir0 = LoadRowR16x4Raw(P.prevInternalData, fx4, fy0), ir1 = LoadRowR16x4Raw(P.prevInternalData, fx4, fy1), ir2 = LoadRowR16x4Raw(P.prevInternalData, fx4, fy2), ir3 = LoadRowR16x4Raw(P.prevInternalData, fx4, fy3);
        const bool normalsInterior = FootprintIsInterior(P.prevNormalRoughness, bx, by, 2, 2);
        {
            const int nx = max(0, min(bx, P.prevNormalRoughness.w - 2));
            LoadRowNrRawx2(P.prevNormalRoughness, nx, ClampI(by, 0, P.prevNormalRoughness.h - 1), n00, n10);
            LoadRowNrRawx2(P.prevNormalRoughness, nx, ClampI(by + 1, 0, P.prevNormalRoughness.h - 1), n01, n11);
        }
        if (DIFF) {
            Sig::PrefetchHistory(smbFilter, P.historyDiff, smbDiffTexels, !PERF && KIND != SIGNAL_DIRECTIONAL_OCCLUSION);
            Sig::PrefetchFast(smbFilter, P.historyDiffFast, smbDiffFastTexels);
            diff = Sig::Load(P.inDiff, (OCC && c.gDiffCheckerboard != 2) ? px >> 1 : px, py);
        }
        if (SPEC) {
            Sig::PrefetchHistory(smbFilter, P.historySpec, smbSpecTexels, !PERF && KIND != SIGNAL_DIRECTIONAL_OCCLUSION);
            Sig::PrefetchFast(smbFilter, P.historySpecFast, smbSpecFastTexels);
            spec = Sig::Load(P.inSpec, (OCC && c.gSpecCheckerboard != 2) ? px >> 1 : px, py);
        }

        // footprints on the border of the plane: clamped / zero-filled texel loads replace the row data
        smbViewZ0 = F4(zr0.x, zr0.y, zr1.x, zr1.y), smbViewZ1 = F4(zr0.z, zr0.w, zr1.z, zr1.w), smbViewZ2 = F4(zr2.x, zr2.y, zr3.x, zr3.y), smbViewZ3 = F4(zr2.z, zr2.w, zr3.z, zr3.w);
        id0[0] = ir0.x & 0xFFFFu, id0[1] = ir0.x >> 16, id0[2] = ir1.x & 0xFFFFu, id0[3] = ir1.x >> 16;
        id1[0] = ir0.y & 0xFFFFu, id1[1] = ir0.y >> 16, id1[2] = ir1.y & 0xFFFFu, id1[3] = ir1.y >> 16;
        id2[0] = ir2.x & 0xFFFFu, id2[1] = ir2.x >> 16, id2[2] = ir3.x & 0xFFFFu, id2[3] = ir3.x >> 16;
        id3[0] = ir2.y & 0xFFFFu, id3[1] = ir2.y >> 16, id3[2] = ir3.y & 0xFFFFu, id3[3] = ir3.y >> 16;
        if (!footprintInterior) {
    #define QUADZ(ox, oy) \
        F4(FetchClampedR32F(P.prevViewZ, cx + ox, cy + oy), FetchClampedR32F(P.prevViewZ, cx + ox + 1, cy + oy), FetchClampedR32F(P.prevViewZ, cx + ox, cy + oy + 1), FetchClampedR32F(P.prevViewZ, cx + ox + 1, cy + oy + 1))
            smbViewZ0 = QUADZ(0, 0), smbViewZ1 = QUADZ(2, 0), smbViewZ2 = QUADZ(0, 2), smbViewZ3 = QUADZ(2, 2);
    #undef QUADZ
    #define QUADU(q, ox, oy)                                                   \
        q[0] = FetchClampedR16U(P.prevInternalData, cx + ox, cy + oy);         \
        q[1] = FetchClampedR16U(P.prevInternalData, cx + ox + 1, cy + oy);     \
        q[2] = FetchClampedR16U(P.prevInternalData, cx + ox, cy + oy + 1);     \
        q[3] = FetchClampedR16U(P.prevInternalData, cx + ox + 1, cy + oy + 1);
            QUADU(id0, 0, 0) QUADU(id1, 2, 0) QUADU(id2, 0, 2) QUADU(id3, 2, 2)
    #undef QUADU
        }
        if (!normalsInterior) {
            n00 = InBounds(P.prevNormalRoughness, bx, by) ? LoadNrRaw(P.prevNormalRoughness, bx, by) : NrRawZero();
            n10 = InBounds(P.prevNormalRoughness, bx + 1, by) ? LoadNrRaw(P.prevNormalRoughness, bx + 1, by) : NrRawZero();
            n01 = InBounds(P.prevNormalRoughness, bx, by + 1) ? LoadNrRaw(P.prevNormalRoughness, bx, by + 1) : NrRawZero();
            n11 = InBounds(P.prevNormalRoughness, bx + 1, by + 1) ? LoadNrRaw(P.prevNormalRoughness, bx + 1, by + 1) : NrRawZero();
        }

    }

    float3 prevViewZ0 = F3(UnpackViewZ(c, smbViewZ0.y), UnpackViewZ(c, smbViewZ0.z), UnpackViewZ(c, smbViewZ0.w));
    float3 prevViewZ1 = F3(UnpackViewZ(c, smbViewZ1.x), UnpackViewZ(c, smbViewZ1.z), UnpackViewZ(c, smbViewZ1.w));
    float3 prevViewZ2 = F3(UnpackViewZ(c, smbViewZ2.x), UnpackViewZ(c, smbViewZ2.y), UnpackViewZ(c, smbViewZ2.w));
    float3 prevViewZ3 = F3(UnpackViewZ(c, smbViewZ3.x), UnpackViewZ(c, smbViewZ3.y), UnpackViewZ(c, smbViewZ3.z));

    // Previous normal averaged over the valid pixels of the 2x2 footprint
    float3 smbNavg;
    {
        float sumw = 0.0f;
        float w = prevViewZ0.z < c.gDenoisingRange ? 1.0f : 0.0f;
        smbNavg = Xyz(UnpackNormalAndRoughness(DecodePrevNormalRoughnessTexel(n00))) * w;
        sumw += w;
        w = prevViewZ1.y < c.gDenoisingRange ? 1.0f : 0.0f;
        smbNavg = smbNavg + Xyz(UnpackNormalAndRoughness(DecodePrevNormalRoughnessTexel(n10))) * w;
        sumw += w;
        w = prevViewZ2.y < c.gDenoisingRange ? 1.0f : 0.0f;
        smbNavg = smbNavg + Xyz(UnpackNormalAndRoughness(DecodePrevNormalRoughnessTexel(n01))) * w;
        sumw += w;
        w = prevViewZ3.x < c.gDenoisingRange ? 1.0f : 0.0f;
        smbNavg = smbNavg + Xyz(UnpackNormalAndRoughness(DecodePrevNormalRoughnessTexel(n11))) * w;
        sumw += w;
        smbNavg = Div(smbNavg, sumw == 0.0f ? 1.0f : sumw);
    }
    smbNavg = RotateVector(c.gWorldPrevToWorld, smbNavg);

    NRD_CONSTANTS_PHASE();
    // Parallax
    float smbParallaxInPixels1 = ComputeParallaxInPixels(Xprev + cameraDelta, NRD_ORTHO_MODE(c) == 0.0f ? smbPixelUv : pixelUv, c.gWorldToClipPrev, rectSize);
    float smbParallaxInPixels2 = ComputeParallaxInPixels(Xprev - cameraDelta, NRD_ORTHO_MODE(c) == 0.0f ? pixelUv : smbPixelUv, c.gWorldToClip, rectSize);
    float smbParallaxInPixelsMax = Max(smbParallaxInPixels1, smbParallaxInPixels2);
    float smbParallaxInPixelsMin = Min(smbParallaxInPixels1, smbParallaxInPixels2);

    // Disocclusion: threshold
    float pixelSize = PixelRadiusToWorld(c.gUnproject, NRD_ORTHO_MODE(c), 1.0f, viewZ);
    float frustumSize = GetFrustumSize(c.gMinRectDimMulUnproject, NRD_ORTHO_MODE(c), viewZ);

    float disocclusionThresholdMix = 0.0f;
    if (materialID == c.gStrandMaterialID)
        disocclusionThresholdMix = Div(pixelSize, pixelSize + c.gStrandThickness); // NRD_GetNormalizedStrandThickness (reference NRD.hlsli:1158-1161)
    if (c.gHasDisocclusionThresholdMix)
        disocclusionThresholdMix = LoadR8Unorm(P.disocclusionThresholdMix, px, py);
    float disocclusionThreshold = Lerp(c.gDisocclusionThreshold, c.gDisocclusionThresholdAlternate, disocclusionThresholdMix);

    float smallParallax = LinearStep(0.25f, 0.0f, smbParallaxInPixelsMax);
    disocclusionThreshold += 0.05f * smallParallax;

    float3 V = GetViewVector(c, X);
    float NoV = Abs(Dot(N, V));
    float NoVstrict = Lerp(NoV, 1.0f, Sat(smbParallaxInPixelsMax * (1.0f / 30.0f)));
    float4 smbDisocclusionThreshold = F4(GetDisocclusionThreshold(disocclusionThreshold, frustumSize, NoVstrict));
    smbDisocclusionThreshold = smbDisocclusionThreshold * (Dot(smbNavg, Navg) > REBLUR_ALMOST_ZERO_ANGLE - 0.25f * smallParallax ? 1.0f : 0.0f);
    smbDisocclusionThreshold = smbDisocclusionThreshold * IsInScreenBilinear(smbBilinearFilter.origin, rectSizePrev);
    smbDisocclusionThreshold = smbDisocclusionThreshold - NRD_EPS;

    // Disocclusion: plane distance
    float3 Xvprev = AffineTransform(c.gWorldToViewPrev, Xprev);
    float3 smbOcclusion0 = Step(Abs(prevViewZ0 - F3(Xvprev.z)), smbDisocclusionThreshold.x);
    float3 smbOcclusion1 = Step(Abs(prevViewZ1 - F3(Xvprev.z)), smbDisocclusionThreshold.y);
    float3 smbOcclusion2 = Step(Abs(prevViewZ2 - F3(Xvprev.z)), smbDisocclusionThreshold.z);
    float3 smbOcclusion3 = Step(Abs(prevViewZ3 - F3(Xvprev.z)), smbDisocclusionThreshold.w);

    // Disocclusion: materialID (the 4x4 internal-data footprint was requested with the depth footprint)
    // Material IDs are 0..3 (the 2-bit field of IN_NORMAL_ROUGHNESS; the internal data keeps them in 4 bits): with a minimum material >= 3 every comparison
    // max(m0, min) == max(m, min) holds (the library default is 4 = "off"), so the twelve unpack + compare chains (~11 instructions each, 6 % of this kernel's executed
    // instructions) sit behind a UNIFORM test of the constants -- same values, as in the spatial passes (kernels_reblur_spatial.hip "compareMaterials").
    float minMaterialID = Min(c.gSpecMinMaterial, c.gDiffMinMaterial);
    if (minMaterialID < 3.0f) {
#define MATCMP(p) (CompareMaterials(materialID, UnpackInternalData(p).z, minMaterialID) ? 1.0f : 0.0f)
        smbOcclusion0 = smbOcclusion0 * F3(MATCMP(id0[1]), MATCMP(id0[2]), MATCMP(id0[3]));
        smbOcclusion1 = smbOcclusion1 * F3(MATCMP(id1[0]), MATCMP(id1[2]), MATCMP(id1[3]));
        smbOcclusion2 = smbOcclusion2 * F3(MATCMP(id2[0]), MATCMP(id2[1]), MATCMP(id2[3]));
        smbOcclusion3 = smbOcclusion3 * F3(MATCMP(id3[0]), MATCMP(id3[1]), MATCMP(id3[2]));
#undef MATCMP
    }
    const uint32_t smbInternalData0 = id0[3], smbInternalData1 = id1[2], smbInternalData2 = id2[1], smbInternalData3 = id3[0];

    NRD_CONSTANTS_PHASE();
    // 2x2 occlusion weights
    float4 smbOcclusionWeights = GetBilinearCustomWeights(smbBilinearFilter, F4(smbOcclusion0.z, smbOcclusion1.y, smbOcclusion2.y, smbOcclusion3.x));
    float3 occSum = smbOcclusion0 + smbOcclusion1 + smbOcclusion2 + smbOcclusion3;
    bool smbAllowCatRom = (occSum.x + occSum.y + occSum.z) > 11.5f && !PERF && KIND != SIGNAL_DIRECTIONAL_OCCLUSION; // no Catmull-Rom in TA for directional occlusion (REBLUR_Config.hlsli:188-194)

    SetHistoryWeights(smbFilter, smbOcclusionWeights, smbAllowCatRom);

    float fbits = smbOcclusion0.z * 1.0f;
    fbits += smbOcclusion1.y * 2.0f;
    fbits += smbOcclusion2.y * 4.0f;
    fbits += smbOcclusion3.x * 8.0f;

    // Accumulation speed
    float3 internalData00 = UnpackInternalData(smbInternalData0), internalData10 = UnpackInternalData(smbInternalData1);
    float3 internalData01 = UnpackInternalData(smbInternalData2), internalData11 = UnpackInternalData(smbInternalData3);
    float diffAccumSpeed = 0.0f, smbSpecAccumSpeed = 0.0f;
    if (DIFF)
        diffAccumSpeed = ApplyBilinearCustomWeights(internalData00.x, internalData10.x, internalData01.x, internalData11.x, smbOcclusionWeights);
    if (SPEC)
        smbSpecAccumSpeed = ApplyBilinearCustomWeights(internalData00.y, internalData10.y, internalData01.y, internalData11.y, smbOcclusionWeights);

    // Footprint quality
    float3 smbVprev = GetViewVectorPrev(c, Xprev, cameraDelta);
    float NoVprev = Abs(Dot(N, smbVprev));
    float sizeQuality = Div(NoVprev + 1e-3f, NoV + 1e-3f);
    sizeQuality *= sizeQuality;
    sizeQuality = Lerp(0.1f, 1.0f, Sat(sizeQuality));

    float smbFootprintQuality = ApplyBilinearFilter(smbOcclusion0.z, smbOcclusion1.y, smbOcclusion2.y, smbOcclusion3.x, smbBilinearFilter);
    smbFootprintQuality = Sqrt01(smbFootprintQuality);
    smbFootprintQuality *= sizeQuality;


    NRD_CONSTANTS_PHASE();
    // Checkerboard (reference REBLUR_TemporalAccumulation.hlsli:307-321): pixels without data this frame accumulate slower; only the occlusion
    // family resolves them here (it has no pre-pass), from the two horizontal neighbours in the half-width input
    const uint32_t checkerboard = CheckerBoard((uint32_t)px, (uint32_t)py, c.gFrameIndex);
    const bool diffHasData = c.gDiffCheckerboard == 2 || checkerboard == c.gDiffCheckerboard;
    const bool specHasData = c.gSpecCheckerboard == 2 || checkerboard == c.gSpecCheckerboard;
    int cbX0 = 0, cbX1 = 0;
    float2 wc = F2(0.0f, 0.0f);
    if (OCC && (c.gDiffCheckerboard != 2 || c.gSpecCheckerboard != 2)) {
        const int x0 = px > 0 ? px - 1 : 0, x1 = px < c.gRectSizeMinusOne.x ? px + 1 : c.gRectSizeMinusOne.x;
        const float viewZ0 = UnpackViewZ(c, LoadR32F(P.viewZ, x0, py)), viewZ1 = UnpackViewZ(c, LoadR32F(P.viewZ, x1, py));
        const float thr = GetDisocclusionThreshold(NRD_DISOCCLUSION_THRESHOLD, frustumSize, NoV);
        wc = F2(thr >= Abs(viewZ0 - viewZ) ? 1.0f : 0.0f, thr >= Abs(viewZ1 - viewZ) ? 1.0f : 0.0f);
        wc.x = (viewZ0 > c.gDenoisingRange || px < 1) ? 0.0f : wc.x;
        wc.y = (viewZ1 > c.gDenoisingRange || px >= c.gRectSizeMinusOne.x) ? 0.0f : wc.y;
        wc = wc * PositiveRcp(wc.x + wc.y);
        cbX0 = x0 >> 1;
        cbX1 = x1 >> 1;
    }

    // ------------------------------------------------------------------------------------------------ diffuse
    // (before the long specular section: everything the diffuse part needs from the shared footprint dies here, not after it)
    if (DIFF) {
        if (MODE == 1) // (the plain kernel requests its inputs with the surface-motion batch)
            diff = Sig::Load(P.inDiff, px, py);
        float diffHistoryConfidence = smbFootprintQuality;
        if (c.gHasHistoryConfidence)
            diffHistoryConfidence *= LoadR8Unorm(P.diffConfidence, px, py);
        diffAccumSpeed *= Lerp(diffHistoryConfidence, 1.0f, Rcp(1.0f + diffAccumSpeed));
        diffAccumSpeed = Min(diffAccumSpeed, c.gMaxAccumulatedFrameNum);

        if (OCC && !diffHasData) {
            S d0 = Select(wc.x == 0.0f, Sig::Zero(), Sig::Load(P.inDiff, cbX0, py));
            S d1 = Select(wc.y == 0.0f, Sig::Zero(), Sig::Load(P.inDiff, cbX1, py));
            diff = d0 * wc.x + d1 * wc.y;
        }

        if (MODE == 1) {
            WindowHistoryTexels(smbFilter, s_WinDiff, wx0, wy0, WIN_W, smbDiffTexels);
            WindowFastTexels(smbFilter, s_WinFast, wx0, wy0, P.historyDiffFast, 0, smbDiffFastTexels);
        }
        S smbDiffHistory = Sig::FetchHistory(smbFilter, P.historyDiff, smbDiffTexels);
        float smbDiffFastHistory = Sig::FetchFastBilinear(smbFilter, P.historyDiffFast, smbDiffFastTexels);
        smbDiffHistory = ClampNegativeToZero(smbDiffHistory);

        float diffNonLinearAccumSpeed = Rcp(1.0f + diffAccumSpeed);
        if (!diffHasData)
            diffNonLinearAccumSpeed *= Lerp(1.0f - c.gCheckerboardResolveAccumSpeed, 1.0f, diffNonLinearAccumSpeed);
        S diffResult = MixHistoryAndCurrent(c, smbDiffHistory, diff, diffNonLinearAccumSpeed);
        float4 diffShResult = F4(0.0f);
        if (SH) {
            float4 smbDiffShHistory = FetchHistoryBilinearRGBA16F(smbFilter, P.historyDiffSh);
            diffShResult = MixHistoryAndCurrent(c, smbDiffShHistory, LoadRGBA16F(P.inDiffSh, px, py), diffNonLinearAccumSpeed);
        }

        float diffMaxRelativeIntensity = 0.0f, diffAntifireflyFactor = 0.0f;
        if (KIND == SIGNAL_RADIANCE) { // firefly suppressor (neither occlusion kind has it)
            diffMaxRelativeIntensity = c.gFireflySuppressorMinRelativeScale + Div(REBLUR_FIREFLY_SUPPRESSOR_MAX_RELATIVE_INTENSITY, diffAccumSpeed + 1.0f);
            diffAntifireflyFactor = diffAccumSpeed * c.gMaxBlurRadius * REBLUR_FIREFLY_SUPPRESSOR_RADIUS_SCALE;
            diffAntifireflyFactor = Div(diffAntifireflyFactor, 1.0f + diffAntifireflyFactor);

            float diffLumaResult = GetLuma(diffResult);
            float diffLumaClamped = Min(diffLumaResult, GetLuma(smbDiffHistory) * diffMaxRelativeIntensity);
            diffLumaClamped = Lerp(diffLumaResult, diffLumaClamped, diffAntifireflyFactor);
            diffResult = ChangeLuma(diffResult, diffLumaClamped);
            if (SH) {
                float k = GetLumaScale(Length(Xyz(diffShResult)), diffLumaClamped);
                diffShResult = F4(diffShResult.x * k, diffShResult.y * k, diffShResult.z * k, diffShResult.w);
            }
        }
        Sig::Store(P.outDiff, px, py, diffResult);
        if (SH)
            StoreRGBA16F(P.outDiffSh, px, py, diffShResult);

        float diffFastAccumSpeed = Min(diffAccumSpeed, c.gMaxFastAccumulatedFrameNum);
        float diffFastNonLinearAccumSpeed = Rcp(1.0f + diffFastAccumSpeed);
        if (!diffHasData)
            diffFastNonLinearAccumSpeed *= Lerp(1.0f - c.gCheckerboardResolveAccumSpeed, 1.0f, diffFastNonLinearAccumSpeed);
        float diffFastResult = Lerp(smbDiffFastHistory, GetLuma(diff), diffFastNonLinearAccumSpeed);
        if (KIND == SIGNAL_RADIANCE) {
            float diffFastClamped = Min(diffFastResult, GetLuma(smbDiffHistory) * diffMaxRelativeIntensity * REBLUR_FIREFLY_SUPPRESSOR_FAST_RELATIVE_INTENSITY);
            diffFastResult = Lerp(diffFastResult, diffFastClamped, diffAntifireflyFactor);
        }
        Sig::StoreFast(P.outDiffFast, px, py, diffFastResult);
    }


    // ------------------------------------------------------------------------------------------------ specular
    float specAccumSpeed = 0.0f, curvature = 0.0f, virtualHistoryAmount = 0.0f;
    if (SPEC) {
        // surface history: the window kernel blends it here, from LDS, so that the footprint geometry (~30 VGPRs) dies before the virtual-motion section;
        // the plain kernel blends the texels it requested with the surface-motion batch after that section
        S smbSpecHistory = Sig::Zero();
        float smbSpecFastHistory = 0.0f;
        float4 smbSpecShHistory = F4(0.0f);
        if (MODE == 1) {
            spec = Sig::Load(P.inSpec, px, py);
            WindowHistoryTexels(smbFilter, s_WinSpec, wx0, wy0, WIN_W, smbSpecTexels);
            WindowFastTexels(smbFilter, s_WinFast, wx0, wy0, P.historySpecFast, 16, smbSpecFastTexels);
            smbSpecHistory = Sig::FetchHistory(smbFilter, P.historySpec, smbSpecTexels);
            smbSpecFastHistory = Sig::FetchFastBilinear(smbFilter, P.historySpecFast, smbSpecFastTexels);
            if (SH)
                smbSpecShHistory = FetchHistoryBilinearRGBA16F(smbFilter, P.historySpecSh);
        }
        float specHistoryConfidence = smbFootprintQuality;
        if (c.gHasHistoryConfidence)
            specHistoryConfidence *= LoadR8Unorm(P.specConfidence, px, py);
        smbSpecAccumSpeed *= Lerp(specHistoryConfidence, 1.0f, Rcp(1.0f + smbSpecAccumSpeed));
        smbSpecAccumSpeed = Min(smbSpecAccumSpeed, c.gMaxAccumulatedFrameNum);

        if (OCC && !specHasData) {
            S s0 = Select(wc.x == 0.0f, Sig::Zero(), Sig::Load(P.inSpec, cbX0, py));
            S s1 = Select(wc.y == 0.0f, Sig::Zero(), Sig::Load(P.inSpec, cbX1, py));
            spec = s0 * wc.x + s1 * wc.y;
        }

        NRD_CONSTANTS_PHASE();
        // Curvature estimation along predicted motion
        // (measured and dropped, r04_m: requesting the high-parallax tap below in front of the surface-motion section -- 9 more live VGPRs in a kernel that sits at its budget of
        //  168, 32 B of scratch: TemporalAccumulation 0.279 ms against 0.264; the same move pays in RELAX's kernel, which has the registers: kernels_relax_ta.hip)
        {
            float2 uvForZeroParallax = Select(NRD_ORTHO_MODE(c) == 0.0f, smbPixelUv, pixelUv);
            float2 deltaUv = uvForZeroParallax - GetScreenUv(c.gWorldToClipPrev, Xprev + cameraDelta);
            deltaUv = deltaUv * rectSize;
            deltaUv = Div(deltaUv, Max(smbParallaxInPixels1, 1.0f / 256.0f));

            float3 n10, x10;
            {
                float3 xv = ReconstructViewPosition(pixelUv + F2(1.0f, 0.0f) * rectSizeInv, frustum, 1.0f, NRD_ORTHO_MODE(c));
                float3 x = RotateVector(c.gViewToWorld, xv);
                float3 v = GetViewVector(c, x);
                float3 o = Select(NRD_ORTHO_MODE(c) == 0.0f, F3(0.0f), x);
                x10 = o + Div(v * Dot(X - o, N), Dot(N, v));
                n10 = Xyz(s_Normal_Roughness[(ty + BORDER) * BUF_STRIDE + tx + BORDER + 1]);
            }
            float3 n01, x01;
            {
                float3 xv = ReconstructViewPosition(pixelUv + F2(0.0f, 1.0f) * rectSizeInv, frustum, 1.0f, NRD_ORTHO_MODE(c));
                float3 x = RotateVector(c.gViewToWorld, xv);
                float3 v = GetViewVector(c, x);
                float3 o = Select(NRD_ORTHO_MODE(c) == 0.0f, F3(0.0f), x);
                x01 = o + Div(v * Dot(X - o, N), Dot(N, v));
                n01 = Xyz(s_Normal_Roughness[(ty + BORDER + 1) * BUF_STRIDE + tx + BORDER]);
            }
            float2 w = Abs(deltaUv) + 1.0f / 256.0f;
            w = Div(w, w.x + w.y);
            float3 x = x10 * w.x + x01 * w.y;
            float3 n = Normalize(n10 * w.x + n01 * w.y);

            // High parallax: flatten the surface on fast motion
            float deltaUvLenFixed = smbParallaxInPixelsMin;
            deltaUvLenFixed *= 1.0f + c.gFramerateScale * Bayer4x4((uint32_t)px, (uint32_t)py, c.gFrameIndex);

            float2 motionUvHigh = pixelUv + deltaUv * deltaUvLenFixed * rectSizeInv;
            motionUvHigh = (Floor(motionUvHigh * rectSize) + 0.5f) * rectSizeInv;

            if (deltaUvLenFixed > 1.0f && IsInScreenNearest(motionUvHigh) != 0.0f) {
                const float2 resolutionScale = ToF2(c.gResolutionScale);
                const float2 uvMax = resolutionScale - ToF2(c.gResourceSizeInv) * 0.5f;
                float2 uvScaled = F2(Min(motionUvHigh.x * resolutionScale.x, uvMax.x), Min(motionUvHigh.y * resolutionScale.y, uvMax.y));
                int2 tz = NearestTexel(P.viewZ, uvScaled);
                float zHigh = UnpackViewZ(c, LoadR32F(P.viewZ, tz.x, tz.y));
                float3 xHigh = ReconstructViewPosition(motionUvHigh, frustum, zHigh, NRD_ORTHO_MODE(c));
                xHigh = RotateVector(c.gViewToWorld, xHigh);
                int2 tn = NearestTexel(P.normalRoughness, uvScaled);
                float3 nHigh = Xyz(LoadDecodedNormalRoughness(P.decodedNR, tn.x, tn.y));

                float zError = Abs(zHigh - viewZ) * Rcp(Max(zHigh, viewZ));
                bool cmp = zError < NRD_CURVATURE_Z_THRESHOLD;
                n = Select(cmp, nHigh, n);
                x = Select(cmp, xHigh, x);
            }

            float3 edge = x - X;
            float edgeLenSq = LengthSquared(edge);
            curvature = Dot(n - N, edge) * PositiveRcp(edgeLenSq);
        }

        NRD_CONSTANTS_PHASE();
        // Virtual motion - coordinates
        float3 Xvirtual = GetXvirtual(hitDistForTracking, curvature, X, Xprev, N, V, roughness);
        float XvirtualLength = Length(Xvirtual);

        float2 vmbPixelUv = GetScreenUv(c.gWorldToClipPrev, Xvirtual);
        vmbPixelUv = Select(materialID == c.gCameraAttachedReflectionMaterialID, smbPixelUv, vmbPixelUv);

        float2 vmbDelta = vmbPixelUv - smbPixelUv;
        float vmbPixelsTraveled = Length(vmbDelta * rectSize);

        // ---- every request that depends only on the virtual-motion position, in one batch (see the surface-motion batch above): the 2x2 footprints of
        // the previous normals / viewZ / internal data, the two stochastic normal taps, the tracking hit distance and the texels of the virtual history
        Bilinear vmbBilinearFilter = GetBilinearFilter(vmbPixelUv, rectSizePrev);
        const int vx = (int)vmbBilinearFilter.origin.x, vy = (int)vmbBilinearFilter.origin.y;
        const bool vmbInterior = FootprintIsInterior(P.prevViewZ, vx, vy, 2, 2); // one test for the three planes of this footprint (same size)
        NrRaw vq00, vq10, vq01, vq11;        // packed normal / roughness
        const int vfx = max(0, min(vx, P.prevViewZ.w - 2)), vfy0 = ClampI(vy, 0, P.prevViewZ.h - 1), vfy1 = ClampI(vy + 1, 0, P.prevViewZ.h - 1); // row loads as above
        LoadRowNrRawx2(P.prevNormalRoughness, vfx, vfy0, vq00, vq10);
        LoadRowNrRawx2(P.prevNormalRoughness, vfx, vfy1, vq01, vq11);
        const float2 vzr0 = LoadRowR32Fx2(P.prevViewZ, vfx, vfy0), vzr1 = LoadRowR32Fx2(P.prevViewZ, vfx, vfy1);
        const uint32_t vidr0 = LoadRowR16x2Raw(P.prevInternalData, vfx, vfy0), vidr1 = LoadRowR16x2Raw(P.prevInternalData, vfx, vfy1);
        // stochastic nearest taps of the bilinear footprint at the virtual position and one step back along the virtual motion (the draws keep their order)
        const float2 resolutionScalePrev = ToF2(c.gResolutionScalePrev);
        auto stochasticTexel = [&](float2 uv) {
            Bilinear f = GetBilinearFilter(uv, rectSizePrev);
            float2 rnd = rng.GetFloat2();
            f.origin = f.origin + F2(Step(rnd.x, f.weights.x), Step(rnd.y, f.weights.y));
            float2 uvs = (Div(f.origin + 0.5f, rectSizePrev)) * resolutionScalePrev;
            return NearestTexel(P.prevNormalRoughness, uvs);
        };
        // (encodings other than R10G10B10A2: no stochastic tap and no draws -- Common.hlsli:76-85 STOCHASTIC_BILINEAR_FILTER = gLinearClamp, `:359-372` StochasticBilinear( uv ) = uv --
        //  the two samples are true bilinear fetches of the encoded texels)
        constexpr bool STOCHASTIC_NORMALS = NRD_NORMAL_ENCODING == NRD_NORMAL_ENCODING_R10G10B10A2_UNORM;
        const float2 prevNormalRoughnessSize = F2(float(P.prevNormalRoughness.w), float(P.prevNormalRoughness.h));
        NrRaw stochasticRaw0 = NrRawZero(), stochasticRaw1 = NrRawZero();
        float4 vmbLinear0 = F4(0.0f), vmbLinear1 = F4(0.0f);
        if (STOCHASTIC_NORMALS) {
            const int2 st0 = stochasticTexel(vmbPixelUv);
            stochasticRaw0 = LoadNrRaw(P.prevNormalRoughness, st0.x, st0.y);
        } else
            vmbLinear0 = SampleLinearPrevNormalRoughness(P.prevNormalRoughness, vmbPixelUv * resolutionScalePrev * prevNormalRoughnessSize);
        const float stepBetweenTaps = Min(vmbPixelsTraveled * c.gFramerateScale, 2.0f) + vmbPixelsTraveled * 1.0f;
        vmbDelta = vmbDelta * Rsqrt(LengthSquared(vmbDelta));
        vmbDelta = Div(vmbDelta, rectSizePrev);
        const float2 vmbPixelUvPrevTap = vmbPixelUv + vmbDelta * 1.0f * stepBetweenTaps;
        if (STOCHASTIC_NORMALS) {
            const int2 st1 = stochasticTexel(vmbPixelUvPrevTap);
            stochasticRaw1 = LoadNrRaw(P.prevNormalRoughness, st1.x, st1.y);
        } else
            vmbLinear1 = SampleLinearPrevNormalRoughness(P.prevNormalRoughness, vmbPixelUvPrevTap * resolutionScalePrev * prevNormalRoughnessSize);
        // previous tracking hit distance: the 2x2 of the linear sample
        const LinearTaps hitDistTaps = MakeLinearTaps(vmbPixelUv * resolutionScalePrev * F2(float(P.prevSpecHitDistForTracking.w), float(P.prevSpecHitDistForTracking.h)));
        const bool hitDistInterior = FootprintIsInterior(P.prevSpecHitDistForTracking, hitDistTaps.x0, hitDistTaps.y0, 2, 2);
        const int hx = max(0, min(hitDistTaps.x0, P.prevSpecHitDistForTracking.w - 2));
        const uint32_t hdr0 = LoadRowR16x2Raw(P.prevSpecHitDistForTracking, hx, ClampI(hitDistTaps.y0, 0, P.prevSpecHitDistForTracking.h - 1));
        const uint32_t hdr1 = LoadRowR16x2Raw(P.prevSpecHitDistForTracking, hx, ClampI(hitDistTaps.y0 + 1, 0, P.prevSpecHitDistForTracking.h - 1));
        // virtual history
        HistoryFilter vmbFilter = MakeHistoryGeometry(Sat(vmbPixelUv) * rectSizePrev, P.historySpec);
        typename Sig::HistoryTexels vmbSpecTexels;
        typename Sig::FastTexels vmbSpecFastTexels;
        if (MODE != 1) { // the window kernel runs three waves per SIMD and requests these where they are blended: their ~28 VGPRs are what the third wave costs
            Sig::PrefetchHistory(vmbFilter, P.historySpec, vmbSpecTexels, !PERF && KIND != SIGNAL_DIRECTIONAL_OCCLUSION);
            Sig::PrefetchFast(vmbFilter, P.historySpecFast, vmbSpecFastTexels);
        }

        // footprints on the border of the plane: clamped texel loads replace the row data
        float vz00 = vzr0.x, vz10 = vzr0.y, vz01 = vzr1.x, vz11 = vzr1.y; // packed viewZ
        uint32_t vmbId00 = vidr0 & 0xFFFFu, vmbId10 = vidr0 >> 16, vmbId01 = vidr1 & 0xFFFFu, vmbId11 = vidr1 >> 16;
        uint32_t hd00 = hdr0 & 0xFFFFu, hd10 = hdr0 >> 16, hd01 = hdr1 & 0xFFFFu, hd11 = hdr1 >> 16;
        if (!vmbInterior) {
            const int x0 = ClampI(vx, 0, P.prevViewZ.w - 1), x1 = ClampI(vx + 1, 0, P.prevViewZ.w - 1), y0 = ClampI(vy, 0, P.prevViewZ.h - 1), y1 = ClampI(vy + 1, 0, P.prevViewZ.h - 1);
            vq00 = LoadNrRaw(P.prevNormalRoughness, x0, y0), vq10 = LoadNrRaw(P.prevNormalRoughness, x1, y0), vq01 = LoadNrRaw(P.prevNormalRoughness, x0, y1), vq11 = LoadNrRaw(P.prevNormalRoughness, x1, y1);
            vz00 = LoadR32F(P.prevViewZ, x0, y0), vz10 = LoadR32F(P.prevViewZ, x1, y0), vz01 = LoadR32F(P.prevViewZ, x0, y1), vz11 = LoadR32F(P.prevViewZ, x1, y1);
            vmbId00 = LoadR16U(P.prevInternalData, x0, y0), vmbId10 = LoadR16U(P.prevInternalData, x1, y0), vmbId01 = LoadR16U(P.prevInternalData, x0, y1), vmbId11 = LoadR16U(P.prevInternalData, x1, y1);
        }
        if (!hitDistInterior) {
            hd00 = FetchClampedR16U(P.prevSpecHitDistForTracking, hitDistTaps.x0, hitDistTaps.y0), hd10 = FetchClampedR16U(P.prevSpecHitDistForTracking, hitDistTaps.x0 + 1, hitDistTaps.y0);
            hd01 = FetchClampedR16U(P.prevSpecHitDistForTracking, hitDistTaps.x0, hitDistTaps.y0 + 1), hd11 = FetchClampedR16U(P.prevSpecHitDistForTracking, hitDistTaps.x0 + 1, hitDistTaps.y0 + 1);
        }

        // Virtual motion - roughness
        float2 relaxedRoughnessWeightParams = GetRelaxedRoughnessWeightParams(roughness * roughness, c.gRoughnessFraction, REBLUR_ROUGHNESS_SENSITIVITY_IN_TA);
        const float4 vmbRoughness = F4(PrevNormalRoughnessTexelRoughness(vq00), PrevNormalRoughnessTexelRoughness(vq10), PrevNormalRoughnessTexelRoughness(vq01), PrevNormalRoughnessTexelRoughness(vq11));
        float4 roughnessWeight;
        roughnessWeight.x = ComputeNonExponentialWeightWithSigma(vmbRoughness.x * vmbRoughness.x, relaxedRoughnessWeightParams.x, relaxedRoughnessWeightParams.y, roughnessSigma);
        roughnessWeight.y = ComputeNonExponentialWeightWithSigma(vmbRoughness.y * vmbRoughness.y, relaxedRoughnessWeightParams.x, relaxedRoughnessWeightParams.y, roughnessSigma);
        roughnessWeight.z = ComputeNonExponentialWeightWithSigma(vmbRoughness.z * vmbRoughness.z, relaxedRoughnessWeightParams.x, relaxedRoughnessWeightParams.y, roughnessSigma);
        roughnessWeight.w = ComputeNonExponentialWeightWithSigma(vmbRoughness.w * vmbRoughness.w, relaxedRoughnessWeightParams.x, relaxedRoughnessWeightParams.y, roughnessSigma);
        float jitterFriendly = SmoothStep(1.0f, 0.0f, smbParallaxInPixelsMax);
        roughnessWeight = F4(Lerp(jitterFriendly, 1.0f, roughnessWeight.x), Lerp(jitterFriendly, 1.0f, roughnessWeight.y), Lerp(jitterFriendly, 1.0f, roughnessWeight.z),
            Lerp(jitterFriendly, 1.0f, roughnessWeight.w));
        float virtualHistoryRoughnessBasedConfidence = ApplyBilinearFilter(roughnessWeight.x, roughnessWeight.y, roughnessWeight.z, roughnessWeight.w, vmbBilinearFilter);

        // Virtual motion - normal: parallax; stochastic nearest tap of the bilinear footprint
        float4 vmbNormalAndRoughness = UnpackNormalAndRoughness(STOCHASTIC_NORMALS ? DecodePrevNormalRoughnessTexel(stochasticRaw0) : vmbLinear0);
        float3 vmbN = RotateVector(c.gWorldPrevToWorld, Xyz(vmbNormalAndRoughness));
        float Dfactor = GetSpecularDominantFactor(NoV, roughness);
        float virtualHistoryNormalBasedConfidence = Rcp(1.0f + 0.5f * Dfactor * Sat(Length(N - vmbN) - REBLUR_NORMAL_ULP) * vmbPixelsTraveled);

        smbNavg = Select(smbFootprintQuality == 0.0f, vmbN, smbNavg);

        NRD_CONSTANTS_PHASE();
        // Virtual motion - disocclusion: plane distance and roughness
        float4 vmbOcclusion;
        {
            float4 vmbOcclusionThreshold = F4(disocclusionThreshold * frustumSize);
            vmbOcclusionThreshold = vmbOcclusionThreshold * Lerp(0.25f, 1.0f, NoV);
            vmbOcclusionThreshold = vmbOcclusionThreshold * (Dot(vmbN, N) > REBLUR_ALMOST_ZERO_ANGLE ? 1.0f : 0.0f);
            vmbOcclusionThreshold = vmbOcclusionThreshold * (Dot(vmbN, smbNavg) > REBLUR_ALMOST_ZERO_ANGLE ? 1.0f : 0.0f);
            vmbOcclusionThreshold = vmbOcclusionThreshold * IsInScreenBilinear(vmbBilinearFilter.origin, rectSizePrev);
            vmbOcclusionThreshold = vmbOcclusionThreshold - NRD_EPS;

            const float4 vmbViewZ = F4(UnpackViewZ(c, vz00), UnpackViewZ(c, vz10), UnpackViewZ(c, vz01), UnpackViewZ(c, vz11));
            float3 vmbVv = ReconstructViewPosition(vmbPixelUv, frustumPrev, 1.0f, 0.0f);
            float3 vmbV = RotateVectorInverse(c.gWorldToViewPrev, vmbVv);
            float NoXcurr = Dot(N, Xprev - cameraDelta);
            float4 NoXprev = (NRD_ORTHO_MODE(c) == 0.0f ? vmbViewZ : F4(NRD_ORTHO_MODE(c))) * (N.x * vmbV.x + N.y * vmbV.y) + vmbViewZ * (N.z * vmbV.z);
            float4 vmbPlaneDist = Abs(NoXprev - NoXcurr);

            vmbOcclusion = Step(vmbPlaneDist, vmbOcclusionThreshold);
            vmbOcclusion = vmbOcclusion * Step(F4(0.5f), roughnessWeight);
        }

        // Virtual motion - disocclusion: materialID
        float3 vmbInternalData00 = UnpackInternalData(vmbId00);
        float3 vmbInternalData10 = UnpackInternalData(vmbId10);
        float3 vmbInternalData01 = UnpackInternalData(vmbId01);
        float3 vmbInternalData11 = UnpackInternalData(vmbId11);
        if (c.gSpecMinMaterial < 3.0f) { // uniform; see the surface-motion footprint above
            vmbOcclusion.x *= CompareMaterials(materialID, vmbInternalData00.z, c.gSpecMinMaterial) ? 1.0f : 0.0f;
            vmbOcclusion.y *= CompareMaterials(materialID, vmbInternalData10.z, c.gSpecMinMaterial) ? 1.0f : 0.0f;
            vmbOcclusion.z *= CompareMaterials(materialID, vmbInternalData01.z, c.gSpecMinMaterial) ? 1.0f : 0.0f;
            vmbOcclusion.w *= CompareMaterials(materialID, vmbInternalData11.z, c.gSpecMinMaterial) ? 1.0f : 0.0f;
        }

        fbits += vmbOcclusion.x * 16.0f;
        fbits += vmbOcclusion.y * 32.0f;
        fbits += vmbOcclusion.z * 64.0f;
        fbits += vmbOcclusion.w * 128.0f;

        // Virtual motion - accumulation speed
        float4 vmbOcclusionWeights = GetBilinearCustomWeights(vmbBilinearFilter, vmbOcclusion);
        float vmbSpecAccumSpeed = ApplyBilinearCustomWeights(vmbInternalData00.y, vmbInternalData10.y, vmbInternalData01.y, vmbInternalData11.y, vmbOcclusionWeights);

        float vmbFootprintQuality = ApplyBilinearFilter(vmbOcclusion.x, vmbOcclusion.y, vmbOcclusion.z, vmbOcclusion.w, vmbBilinearFilter);
        vmbFootprintQuality = Sqrt01(vmbFootprintQuality);
        vmbSpecAccumSpeed *= Lerp(vmbFootprintQuality, 1.0f, Rcp(1.0f + vmbSpecAccumSpeed));

        bool vmbAllowCatRom = Sum(vmbOcclusion) > 3.5f && !PERF && KIND != SIGNAL_DIRECTIONAL_OCCLUSION;
        vmbAllowCatRom = vmbAllowCatRom && smbAllowCatRom;

        float curvatureAngleTan = pixelSize * Abs(curvature);
        curvatureAngleTan *= Max(Div(vmbPixelsTraveled, Max(NoV, 0.01f)), 1.0f);
        curvatureAngleTan *= 2.0f;
        float curvatureAngle = Atan(curvatureAngleTan);

        float percentOfVolume = Div(NRD_MAX_PERCENT_OF_LOBE_VOLUME, 1.0f + vmbSpecAccumSpeed);
        float lobeTanHalfAngle = GetSpecularLobeTanHalfAngle(roughnessModified, percentOfVolume);
        float lobeHalfAngle = Atan(lobeTanHalfAngle);
        lobeHalfAngle = Max(lobeHalfAngle, NRD_NORMAL_ENCODING_ERROR);

        float normalWeight = GetEncodingAwareNormalWeight(N, vmbN, lobeHalfAngle, curvatureAngle, REBLUR_NORMAL_ULP);
        normalWeight = Lerp(SmoothStep(1.0f, 0.0f, vmbPixelsTraveled), 1.0f, normalWeight);
        virtualHistoryNormalBasedConfidence = Min(virtualHistoryNormalBasedConfidence, normalWeight);

        virtualHistoryAmount = SmoothStep(0.05f, 0.95f, Dfactor);
        virtualHistoryAmount *= virtualHistoryNormalBasedConfidence;

        NRD_CONSTANTS_PHASE();
        // Virtual motion - virtual parallax difference
        float virtualHistoryParallaxBasedConfidence;
        {
            float hitDistForTrackingPrev = HalfBitsToFloat((uint16_t)hd00) * hitDistTaps.w00 + HalfBitsToFloat((uint16_t)hd10) * hitDistTaps.w10 + HalfBitsToFloat((uint16_t)hd01) * hitDistTaps.w01 +
                                           HalfBitsToFloat((uint16_t)hd11) * hitDistTaps.w11; // SampleLinearR16F on the texels requested above
            float3 XvirtualPrev = GetXvirtual(hitDistForTrackingPrev, curvature, X, Xprev, N, V, roughness);

            float2 vmbPixelUvPrev = GetScreenUv(c.gWorldToClipPrev, XvirtualPrev);
            vmbPixelUvPrev = Select(materialID == c.gCameraAttachedReflectionMaterialID, smbPixelUv, vmbPixelUvPrev);

            float pixelSizeAtXvirtual = PixelRadiusToWorld(c.gUnproject, NRD_ORTHO_MODE(c), 1.0f, XvirtualLength);
            float r = Div((lobeTanHalfAngle + curvatureAngle) * Min(hitDistForTracking, hitDistForTrackingPrev), pixelSizeAtXvirtual);
            float d = Length((vmbPixelUvPrev - vmbPixelUv) * rectSize);

            r = Max(r, 0.1f);
            virtualHistoryParallaxBasedConfidence = LinearStep(r, 0.0f, d);
        }

        // Virtual motion - normal & roughness prev-prev tests (1 iteration)
        relaxedRoughnessWeightParams = GetRelaxedRoughnessWeightParams(vmbNormalAndRoughness.w * vmbNormalAndRoughness.w, c.gRoughnessFraction, REBLUR_ROUGHNESS_SENSITIVITY_IN_TA);
        {
            const float i = 1.0f;
            const float2 vmbPixelUvPrev = vmbPixelUvPrevTap;
            float4 vmbNormalAndRoughnessPrev = UnpackNormalAndRoughness(STOCHASTIC_NORMALS ? DecodePrevNormalRoughnessTexel(stochasticRaw1) : vmbLinear1);

            float2 w;
            w.x = GetEncodingAwareNormalWeight(Xyz(vmbNormalAndRoughness), Xyz(vmbNormalAndRoughnessPrev), lobeHalfAngle, curvatureAngle * (1.0f + i * stepBetweenTaps), REBLUR_NORMAL_ULP);
            w.y = ComputeNonExponentialWeightWithSigma(vmbNormalAndRoughnessPrev.w * vmbNormalAndRoughnessPrev.w, relaxedRoughnessWeightParams.x, relaxedRoughnessWeightParams.y, roughnessSigma);
            if (STOCHASTIC_NORMALS) // "cures issues of StochasticBilinear" (REBLUR_TemporalAccumulation.hlsli:599-602): R10G10B10A2 only
                w = Lerp(F2(1.0f, 1.0f), w, Sat(stepBetweenTaps));
            w = Select(IsInScreenNearest(vmbPixelUvPrev) != 0.0f, w, F2(1.0f, 1.0f));

            virtualHistoryNormalBasedConfidence = Min(virtualHistoryNormalBasedConfidence, w.x);
            virtualHistoryRoughnessBasedConfidence = Min(virtualHistoryRoughnessBasedConfidence, w.y);
        }

        float virtualHistoryConfidenceForSmbRelaxation = virtualHistoryNormalBasedConfidence * virtualHistoryRoughnessBasedConfidence;
        float virtualHistoryConfidence = virtualHistoryNormalBasedConfidence * virtualHistoryRoughnessBasedConfidence * virtualHistoryParallaxBasedConfidence;
        virtualHistoryAmount *= virtualHistoryRoughnessBasedConfidence;
        // multi-GPU hosts: the virtual-motion position and the look-back tap behind it, where they still COUNT (the virtual history is blended with this weight; the positions of
        // rejected samples -- grazing reflections project anywhere on the screen -- would say nothing about a halo)
        TrackHistoryReach(P.historyReach, virtualHistoryAmount != 0.0f ? Max(HistoryReachRows(vmbPixelUv.y, rectSizePrev.y, py), HistoryReachRows(vmbPixelUvPrevTap.y, rectSizePrev.y, py)) : 0.0f);

        NRD_CONSTANTS_PHASE();
        // Sample surface history
        if (MODE != 1) {
            smbSpecHistory = Sig::FetchHistory(smbFilter, P.historySpec, smbSpecTexels);
            smbSpecFastHistory = Sig::FetchFastBilinear(smbFilter, P.historySpecFast, smbSpecFastTexels);
        }

        float surfaceHistoryConfidence;
        {
            float a = Atan(Div(smbParallaxInPixelsMax * pixelSize, Length(X)));
            float nonLinearAccumSpeed = Rcp(1.0f + smbSpecAccumSpeed);
            float h = Lerp(ExtractHitDist(smbSpecHistory), ExtractHitDist(spec), nonLinearAccumSpeed) * hitDistNormalization;

            float tana0 = GetSpecularLobeTanHalfAngle(roughnessModified, NRD_MAX_PERCENT_OF_LOBE_VOLUME);
            tana0 *= Lerp(NoV, 1.0f, roughnessModified);
            tana0 *= nonLinearAccumSpeed;
            tana0 = Div(tana0, GetHitDistFactor(h, frustumSize) + NRD_EPS);

            float a0 = Atan(tana0);
            a0 = Max(a0, NRD_NORMAL_ENCODING_ERROR);

            float f = LinearStep(a0, 0.0f, a);
            surfaceHistoryConfidence = Pow01(f, 4.0f);
        }

        // Responsive accumulation
        float2 maxResponsiveFrameNum;
        {
            float responsiveFactor = RemapRoughnessToResponsiveFactor(c, roughness);
            float smc = GetSpecMagicCurve(roughnessModified);
            float2 f = F2(Dot(N, Normalize(smbNavg)), Dot(N, vmbN));
            float e = Lerp(32.0f, 1.0f, smc) * (1.0f - responsiveFactor);
            f = F2(Pow01(f.x, e), Pow01(f.y, e)) * Lerp(smc, 1.0f, responsiveFactor);
            maxResponsiveFrameNum = F2(Max(c.gMaxAccumulatedFrameNum * f.x, c.gHistoryFixFrameNum), Max(c.gMaxAccumulatedFrameNum * f.y, c.gHistoryFixFrameNum));
        }

        float smbMaxFrameNum = c.gMaxAccumulatedFrameNum;
        smbMaxFrameNum *= surfaceHistoryConfidence;
        smbMaxFrameNum = Min(smbMaxFrameNum, maxResponsiveFrameNum.x);

        float smbBoostedMaxFrameNum = Max(smbMaxFrameNum, c.gHistoryFixFrameNum * (1.0f - virtualHistoryConfidenceForSmbRelaxation));
        float smbSpecAccumSpeedBoosted = Min(smbSpecAccumSpeed, smbBoostedMaxFrameNum);

        float vmbMaxFrameNum = c.gMaxAccumulatedFrameNum;
        vmbMaxFrameNum *= virtualHistoryConfidence;
        vmbMaxFrameNum = Min(vmbMaxFrameNum, maxResponsiveFrameNum.y);

        smbSpecAccumSpeed = Min(smbSpecAccumSpeed, smbMaxFrameNum);
        vmbSpecAccumSpeed = Min(vmbSpecAccumSpeed, vmbMaxFrameNum);

        float magic = vmbSpecAccumSpeed > smbSpecAccumSpeed ? 8.0f : 0.5f;
        virtualHistoryAmount *= 1.0f + Div(vmbSpecAccumSpeed - smbSpecAccumSpeed, magic * Max(vmbSpecAccumSpeed, smbSpecAccumSpeed) + 1.0f);
        virtualHistoryAmount = Sat(virtualHistoryAmount);

        NRD_CONSTANTS_PHASE();
        // Sample virtual history
        SetHistoryWeights(vmbFilter, vmbOcclusionWeights, vmbAllowCatRom);
        if (MODE == 1) {
            Sig::PrefetchHistory(vmbFilter, P.historySpec, vmbSpecTexels, !PERF && KIND != SIGNAL_DIRECTIONAL_OCCLUSION);
            Sig::PrefetchFast(vmbFilter, P.historySpecFast, vmbSpecFastTexels);
        }
        S vmbSpecHistory = Sig::FetchHistory(vmbFilter, P.historySpec, vmbSpecTexels);
        float vmbSpecFastHistory = Sig::FetchFastBilinear(vmbFilter, P.historySpecFast, vmbSpecFastTexels);

        smbSpecHistory = ClampNegativeToZero(smbSpecHistory);
        vmbSpecHistory = ClampNegativeToZero(vmbSpecHistory);

        float smbSpecNonLinearAccumSpeed = Rcp(1.0f + smbSpecAccumSpeed);
        float vmbSpecNonLinearAccumSpeed = Rcp(1.0f + vmbSpecAccumSpeed);
        if (!specHasData) {
            smbSpecNonLinearAccumSpeed *= Lerp(1.0f - c.gCheckerboardResolveAccumSpeed, 1.0f, smbSpecNonLinearAccumSpeed);
            vmbSpecNonLinearAccumSpeed *= Lerp(1.0f - c.gCheckerboardResolveAccumSpeed, 1.0f, vmbSpecNonLinearAccumSpeed);
        }

        S smbSpec = MixHistoryAndCurrent(c, smbSpecHistory, spec, smbSpecNonLinearAccumSpeed, roughnessModified);
        S vmbSpec = MixHistoryAndCurrent(c, vmbSpecHistory, spec, vmbSpecNonLinearAccumSpeed, roughnessModified);
        S specResult = Lerp(smbSpec, vmbSpec, virtualHistoryAmount);

        float4 specShResult = F4(0.0f);
        if (SH) {
            if (MODE != 1)
                smbSpecShHistory = FetchHistoryBilinearRGBA16F(smbFilter, P.historySpecSh);
            float4 vmbSpecShHistory = FetchHistoryBilinearRGBA16F(vmbFilter, P.historySpecSh);
            float4 specSh = LoadRGBA16F(P.inSpecSh, px, py);
            float4 smbShSpec = Lerp(smbSpecShHistory, specSh, smbSpecNonLinearAccumSpeed);
            float4 vmbShSpec = Lerp(vmbSpecShHistory, specSh, vmbSpecNonLinearAccumSpeed);
            specShResult = Lerp(smbShSpec, vmbShSpec, virtualHistoryAmount);
            specShResult.w = roughnessModified; // assists AA during the SG resolve; never blurred
        }

        specAccumSpeed = Lerp(smbSpecAccumSpeedBoosted, vmbSpecAccumSpeed, virtualHistoryAmount);
        S specHistory = Lerp(smbSpecHistory, vmbSpecHistory, virtualHistoryAmount);

        // Firefly suppressor (not in the occlusion family)
        float specMaxRelativeIntensity = 0.0f, specAntifireflyFactor = 0.0f;
        if (KIND == SIGNAL_RADIANCE) {
            specMaxRelativeIntensity = c.gFireflySuppressorMinRelativeScale + Div(REBLUR_FIREFLY_SUPPRESSOR_MAX_RELATIVE_INTENSITY, specAccumSpeed + 1.0f);
            specAntifireflyFactor = specAccumSpeed * c.gMaxBlurRadius * REBLUR_FIREFLY_SUPPRESSOR_RADIUS_SCALE;
            specAntifireflyFactor = Div(specAntifireflyFactor, 1.0f + specAntifireflyFactor);

            float specLumaResult = GetLuma(specResult);
            float specLumaClamped = Min(specLumaResult, GetLuma(specHistory) * specMaxRelativeIntensity);
            specLumaClamped = Lerp(specLumaResult, specLumaClamped, specAntifireflyFactor);
            specResult = ChangeLuma(specResult, specLumaClamped);
            if (SH) {
                float k = GetLumaScale(Length(Xyz(specShResult)), specLumaClamped);
                specShResult = F4(specShResult.x * k, specShResult.y * k, specShResult.z * k, specShResult.w);
            }
        }

        Sig::Store(P.outSpec, px, py, specResult);
        if (SH)
            StoreRGBA16F(P.outSpecSh, px, py, specShResult);

        // Fast history
        float smbSpecFastNonLinearAccumSpeed = GetNonLinearAccumSpeed(c, smbSpecAccumSpeed, c.gMaxFastAccumulatedFrameNum, surfaceHistoryConfidence, specHasData);
        float vmbSpecFastNonLinearAccumSpeed = GetNonLinearAccumSpeed(c, vmbSpecAccumSpeed, c.gMaxFastAccumulatedFrameNum, virtualHistoryConfidence, specHasData);
        float smbSpecFast = Lerp(smbSpecFastHistory, GetLuma(spec), smbSpecFastNonLinearAccumSpeed);
        float vmbSpecFast = Lerp(vmbSpecFastHistory, GetLuma(spec), vmbSpecFastNonLinearAccumSpeed);
        float specFastResult = Lerp(smbSpecFast, vmbSpecFast, virtualHistoryAmount);

        if (KIND == SIGNAL_RADIANCE) {
            float specFastClamped = Min(specFastResult, GetLuma(specHistory) * specMaxRelativeIntensity * REBLUR_FIREFLY_SUPPRESSOR_FAST_RELATIVE_INTENSITY);
            specFastResult = Lerp(specFastResult, specFastClamped, specAntifireflyFactor);
        }
        Sig::StoreFast(P.outSpecFast, px, py, specFastResult);
    }

    NRD_CONSTANTS_PHASE();
    // DATA2: occlusion bits, curvature, virtual history amount (R32_UINT; diffuse-only keeps the low byte in R8_UINT)
    if (!OCC) {
        uint32_t packed = PackData2(fbits, curvature, virtualHistoryAmount);
        if (SPEC)
            StoreR32U(P.outData2, px, py, packed);
        else
            StoreR8U(P.outData2, px, py, packed);
    }

    StoreData1<DIFF, SPEC>(P.outData1, px, py, diffAccumSpeed, specAccumSpeed);
}

// MODE 0 / 1: one workgroup per tile (XCD-aware order). MODE 2 (fallback behind the window kernel): one workgroup per FALLBACK_TILES tile columns, which walks
// them and runs the pass on the flagged ones -- normally none: 4 us per launch where one workgroup per tile with a flag test in front cost 14 us
// (profiles/r03_k_reblur_ds_kernel_stats.txt against r03_i_reblur_ds_kernel_stats.txt).
constexpr int FALLBACK_TILES = 8;
template <bool DIFF, bool SPEC, bool PERF, int KIND, bool SH, int WAVES, int MODE>
__global__ __launch_bounds__(TILE_X* TILE_Y, WAVES) void ReblurTemporalAccumulationKernel(ReblurCB cArg, TaPlanes P, RowRange rr) {
    const int blockY = BlockTileY(rr, true);
    if (MODE != 2) {
        ReblurTemporalAccumulationTile<DIFF, SPEC, PERF, KIND, SH, MODE>(cArg, P, rr, BlockTileX(rr), blockY);
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
        ReblurTemporalAccumulationTile<DIFF, SPEC, PERF, KIND, SH, MODE>(cArg, P, rr, firstTile + k, blockY);
    }
}

template <bool DIFF, bool SPEC, bool PERF, int KIND, bool SH>
static const char* LaunchTemporalAccumulation(const PassArgs& a) {
    constexpr bool OCC = KIND == SIGNAL_OCCLUSION;
    const ReblurCB& c = *(const ReblurCB*)a.constants;
    if (c.gRectOrigin.x != 0 || c.gRectOrigin.y != 0) // the executor moves the rect of the guide inputs to (0, 0) and zeroes this field (executor.hip "shifted rect")
        return "internal error: a pass was handed a non-zero rectOrigin";
    if (c.gOrthoMode != 0.0f)
        return "REBLUR: orthographic projection is not supported (SURVEY.md section 8c)";

    TaPlanes P = {};
    P.historyReach = a.historyReachWord;
    uint32_t k = 0;
    P.tiles = a.planes[k++];
    P.normalRoughness = a.planes[k++];
    if (const char* err = MakeNormalRoughnessGuide(a, P.decodedNR))
        return err;
    P.viewZ = a.planes[k++];
    P.mv = a.planes[k++];
    P.prevViewZ = a.planes[k++];
    P.prevNormalRoughness = a.planes[k++];
    P.prevInternalData = a.planes[k++];
    P.disocclusionThresholdMix = a.planes[k++];
    if (DIFF) P.diffConfidence = a.planes[k++];
    if (SPEC) P.specConfidence = a.planes[k++];
    if (DIFF) P.inDiff = a.planes[k++];
    if (SPEC) P.inSpec = a.planes[k++];
    if (DIFF) P.historyDiff = a.planes[k++];
    if (SPEC) P.historySpec = a.planes[k++];
    if (DIFF) P.historyDiffFast = a.planes[k++];
    if (SPEC) P.historySpecFast = a.planes[k++];
    if (SPEC) P.prevSpecHitDistForTracking = a.planes[k++];
    if (SPEC && !OCC) P.inSpecHitDistForTracking = a.planes[k++];
    if (DIFF && SH) P.inDiffSh = a.planes[k++];
    if (SPEC && SH) P.inSpecSh = a.planes[k++];
    if (DIFF && SH) P.historyDiffSh = a.planes[k++];
    if (SPEC && SH) P.historySpecSh = a.planes[k++];
    if (DIFF) P.outDiff = a.planes[k++];
    if (SPEC) P.outSpec = a.planes[k++];
    if (DIFF) P.outDiffFast = a.planes[k++];
    if (SPEC) P.outSpecFast = a.planes[k++];
    if (SPEC) P.outSpecHitDistForTracking = a.planes[k++];
    P.outData1 = a.planes[k++];
    if (!OCC) P.outData2 = a.planes[k++];
    if (DIFF && SH) P.outDiffSh = a.planes[k++];
    if (SPEC && SH) P.outSpecSh = a.planes[k++];
    if (k != a.planesNum)
        return "REBLUR temporal accumulation: unexpected resource count";
    {
        const Plane size = P.viewZ, rgba16 = DIFF ? P.historyDiff : P.historySpec, r16 = DIFF ? P.historyDiffFast : P.historySpecFast;
        bool ok = SameSize(P.decodedNR, size) && SameSize(P.mv, size) && SameSize(P.prevViewZ, size) && SameSize(P.prevNormalRoughness, size) && SameSize(P.prevInternalData, size) &&
                  SameSize(P.inDiff, size) && SameSize(P.inSpec, size) && SameSize(P.inSpecHitDistForTracking, size) && SameSize(P.outData1, size) && SameSize(P.outData2, size) &&
                  SameSize(rgba16, size) && SameSize(r16, size);
        if (OCC) {
            const Plane sig = DIFF ? P.outDiff : P.outSpec;
            ok = ok && SameSize(P.historyDiff, size) && SameSize(P.historySpec, size) && SameLayout(P.outDiff, sig) && SameLayout(P.outSpec, sig);
        } else {
            ok = ok && SameLayout(P.historyDiff, rgba16) && SameLayout(P.historySpec, rgba16) && SameLayout(P.outDiff, rgba16) && SameLayout(P.outSpec, rgba16);
        }
        ok = ok && SameLayout(P.historyDiffFast, r16) && SameLayout(P.historySpecFast, r16) && SameLayout(P.prevSpecHitDistForTracking, r16) && SameLayout(P.outDiffFast, r16) &&
             SameLayout(P.outSpecFast, r16) && SameLayout(P.outSpecHitDistForTracking, r16);
        if (!ok)
            return "REBLUR temporal accumulation: planes of one frame must share their size (and pool planes of one format their pitch)";
    }

    RowGrid g = GridForRows(c.gRectSizeMinusOne.x + 1, c.gRectSizeMinusOne.y + 1, TILE_X, TILE_Y, a.rowBegin, a.rowEnd);
    // register budget: with a specular signal the batched requests need ~250 VGPRs (2 waves per SIMD); the diffuse-only kernel fits 168 without scratch,
    // which keeps its third wave (r02_k: 0.135 ms at 2 waves against 0.109 before the batching)
    static const int wavesEnv = getenv("NRD_HIP_TA_WAVES") ? atoi(getenv("NRD_HIP_TA_WAVES")) : 0;
    static const bool windowEnv = !(getenv("NRD_HIP_TA_WINDOW") && atoi(getenv("NRD_HIP_TA_WINDOW")) == 0); // A/B switch
    // the window kernel pays where the pass is heaviest: radiance + hit distance with BOTH signals (0.267 + 0.002 ms against 0.298). Measured and left out: the
    // single-signal denoisers (REBLUR_DIFFUSE: 0.126 + launch of the fallback against 0.112 -- the plain kernel already runs 3 waves and is VALU-bound, r03_i)
    constexpr bool HAS_WINDOW = DIFF && SPEC && !PERF && KIND == SIGNAL_RADIANCE;
    if (HAS_WINDOW && windowEnv && !wavesEnv) {
        if (!a.tileFlags.ptr || (uint32_t)a.tileFlags.w * TILE_X < (uint32_t)P.viewZ.w || (uint32_t)a.tileFlags.h * TILE_Y < (uint32_t)P.viewZ.h)
            return "REBLUR temporal accumulation: the executor's tile-flag scratch is missing or too small";
        P.tileFlags = a.tileFlags;
        // both kernels of the pass only look at the flags of the rect's tile columns (dynamic resolution: columns beyond keep whatever an earlier, larger rect left)
        P.tileFlags.w = min(a.tileFlags.w, (int)((c.gRectSizeMinusOne.x + 1) + TILE_X - 1) / TILE_X);
        if (a.windowRegion)
            a.windowRegion[0] = P.tileFlags.w, a.windowRegion[1] = g.firstBlockY, a.windowRegion[2] = g.firstBlockY + (int)g.grid.y;
        static const char* limitEnv = getenv("NRD_HIP_TA_WINDOW_LIMIT"); // "WxH", test hook: a smaller box sends tiles to the fallback kernel (results do not change)
        int limW = WIN_W, limH = WIN_H;
        if (limitEnv && sscanf(limitEnv, "%dx%d", &limW, &limH) != 2)
            limW = WIN_W, limH = WIN_H;
        P.winMaxW = limW < WIN_W ? limW : WIN_W;
        P.winMaxH = limH < WIN_H ? limH : WIN_H;
        // window kernel (LDS-staged surface-motion footprints, 3 waves per SIMD), then the plain kernel on the tiles the first one declined
        LaunchPass(a, (ReblurTemporalAccumulationKernel<DIFF, SPEC, PERF, KIND, SH, 3, HAS_WINDOW ? 1 : 0>), g.grid, dim3(TILE_X * TILE_Y), c, P, MakeRowRange(g));
        dim3 fallbackGrid = g.grid;
        fallbackGrid.x = (unsigned)((P.tileFlags.w + FALLBACK_TILES - 1) / FALLBACK_TILES);
        LaunchPass(a, (ReblurTemporalAccumulationKernel<DIFF, SPEC, PERF, KIND, SH, SPEC ? 2 : 3, HAS_WINDOW ? 2 : 0>), fallbackGrid, dim3(TILE_X * TILE_Y), c, P, MakeRowRange(g));
        return nullptr;
    }
    const int waves = wavesEnv ? wavesEnv : (SPEC ? 2 : 3);
    if (waves >= 3)
        LaunchPass(a, (ReblurTemporalAccumulationKernel<DIFF, SPEC, PERF, KIND, SH, 3, 0>), g.grid, dim3(TILE_X * TILE_Y), c, P, MakeRowRange(g));
    else
        LaunchPass(a, (ReblurTemporalAccumulationKernel<DIFF, SPEC, PERF, KIND, SH, 2, 0>), g.grid, dim3(TILE_X * TILE_Y), c, P, MakeRowRange(g));
    return nullptr;
}

const PassEntry* GetReblurTemporalAccumulationPasses(uint32_t& num) {
    static const PassEntry k[] = {
#define REBLUR_TA_FAMILY(NAME, D, S)                                                                                 \
    {"REBLUR_" NAME "_TemporalAccumulation.cs", LaunchTemporalAccumulation<D, S, false, false, false>},            \
    {"REBLUR_Perf_" NAME "_TemporalAccumulation.cs", LaunchTemporalAccumulation<D, S, true, false, false>},        \
    {"REBLUR_" NAME "Sh_TemporalAccumulation.cs", LaunchTemporalAccumulation<D, S, false, false, true>},           \
    {"REBLUR_Perf_" NAME "Sh_TemporalAccumulation.cs", LaunchTemporalAccumulation<D, S, true, false, true>},       \
    {"REBLUR_" NAME "Occlusion_TemporalAccumulation.cs", LaunchTemporalAccumulation<D, S, false, true, false>},    \
    {"REBLUR_Perf_" NAME "Occlusion_TemporalAccumulation.cs", LaunchTemporalAccumulation<D, S, true, true, false>},
        REBLUR_TA_FAMILY("Diffuse", true, false)
        REBLUR_TA_FAMILY("Specular", false, true)
        REBLUR_TA_FAMILY("DiffuseSpecular", true, true)
        {"REBLUR_DiffuseDirectionalOcclusion_TemporalAccumulation.cs", LaunchTemporalAccumulation<true, false, false, 2, false>},
        {"REBLUR_Perf_DiffuseDirectionalOcclusion_TemporalAccumulation.cs", LaunchTemporalAccumulation<true, false, true, 2, false>},
    };
    num = sizeof(k) / sizeof(k[0]);
    return k;
}

} // namespace nrdhip
