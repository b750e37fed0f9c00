// Clear and REFERENCE passes (pure streaming kernels).
//   Clear_Float / Clear_Uint          reference Shaders/Source/Clear_{Float,Uint}.cs.hlsl:16-23
//   REFERENCE_TemporalAccumulation    reference Shaders/Source/REFERENCE_TemporalAccumulation.cs.hlsl:18-27
//   REFERENCE_Copy                    reference Shaders/Source/REFERENCE_Copy.cs.hlsl:18-26
// MI355X mapping: rows are processed as 16-byte vectors, one 256-thread block covers 256 vectors of one row segment,
// so every wave issues full 1 KiB coalesced transactions; ~48 B/px of traffic and no reuse => HBM-bound.
#include "../common/pass_constants.h"

#include <climits>
#include "nrdmath.h"
#include "passes.h"
#include "reblur_device.h"

namespace nrdhip {

struct GuideRows {
    int launchBegin, launchEnd; // rows the grid covers
    int validBegin, validEnd;   // rows that receive decoded values (the others, test hook only: NaNs)
};

// ---- decoded guides, once per frame (reblur_device.h "decoded guides") ------------------------------------------------------------------------------
// REBLUR lists: a (normal, viewZ = |z * gViewZScale|) float4 plane -- what a tap of the spatial passes needs from its texel in ONE 16-byte load -- and the
// roughness | material word at 4 B per pixel. 8 B read + 20 B written per pixel, one texel per lane: a wave's accesses are contiguous runs.
__global__ __launch_bounds__(256) void DecodeGuidesKernel(Plane packed, Plane viewZ, Plane viewPos, Plane roughnessWord, float viewZScale, GuideRows rows) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y + rows.launchBegin;
    if (x >= packed.w)
        return;
    if (y < rows.validBegin || y >= rows.validEnd) { // test hook only (NRD_HIP_POISON_GUIDES)
        // (the readers mask the word with 0x3FFFFFFF -- reblur_device.h DecodedRoughness --, so no bit pattern of it decodes to a NaN: all ones decode to a roughness of 1.9999999,
        //  far outside [0, 1], and material 3, which changes every weight that uses them; the normals next to it are NaNs. ADVICE r05: 0x7FC00000 decoded to a plausible 1.5 / material 1)
        StoreRGBA32F(viewPos, x, y, F4(__uint_as_float(0x7FC00000u)));
        StoreR32U(roughnessWord, x, y, 0xFFFFFFFFu);
        return;
    }
    const float4 d = EncodeDecodedNormalRoughness(LoadNrRaw(packed, x, y));
    StoreRGBA32F(viewPos, x, y, F4(d.x, d.y, d.z, Abs(LoadR32F(viewZ, x, y) * viewZScale)));
    StoreR32U(roughnessWord, x, y, AsUint(d.w));
}

// rows the guide kernels have to cover: PassArgs::rowBegin / rowEnd when the executor set them (multi-GPU: the strip + the reach of the passes that read the guides), else all.
// NRD_HIP_POISON_GUIDES=1 (test hook): the kernels run over the whole planes and write NaNs outside those rows, so that a pass reading a row it did not declare shows up
static GuideRows MakeGuideRows(const PassArgs& a, const Plane& p) {
    static const bool poison = getenv("NRD_HIP_POISON_GUIDES") && atoi(getenv("NRD_HIP_POISON_GUIDES")) != 0;
    GuideRows r;
    r.validBegin = a.rowEnd > a.rowBegin && a.rowBegin > 0 ? a.rowBegin : 0;
    r.validEnd = a.rowEnd > a.rowBegin && a.rowEnd < p.h ? a.rowEnd : p.h;
    r.launchBegin = poison ? 0 : r.validBegin;
    r.launchEnd = poison ? p.h : r.validEnd;
    return r;
}

void LaunchDecodeGuides(const PassArgs& a, const Plane& packed, const Plane& viewZ, const Plane& viewPos, const Plane& roughnessWord, const void* reblurConstants) {
    const nrdc::ReblurConstants& c = *(const nrdc::ReblurConstants*)reblurConstants;
    const GuideRows rows = MakeGuideRows(a, packed);
    if (rows.launchEnd <= rows.launchBegin)
        return;
    LaunchPass(a, DecodeGuidesKernel, dim3((unsigned)((packed.w + 255) / 256), (unsigned)(rows.launchEnd - rows.launchBegin), 1), dim3(256), packed, viewZ, viewPos, roughnessWord, c.gViewZScale, rows);
}

// RELAX lists: the same decode plus (world position, viewZ) of every pixel -- GetCurrentWorldPosFromPixelPos( pixel, |z * gViewZScale| ) of RELAX_Common.hlsli:
// the plane the a-trous taps read (formerly written by the first a-trous pass) and, being available from the start of the frame, the pre-pass taps too.
__global__ __launch_bounds__(256) void DecodeGuidesRelaxKernel(Plane packed, Plane viewZ, Plane decoded, Plane worldPos, float3 frustumRight, float3 frustumUp, float3 frustumForward,
    float2 rectSizeInv, float viewZScale, GuideRows rows) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y + rows.launchBegin;
    if (x >= packed.w)
        return;
    if (y < rows.validBegin || y >= rows.validEnd) { // test hook only (NRD_HIP_POISON_GUIDES)
        StoreRGBA32F(decoded, x, y, F4(__uint_as_float(0x7FC00000u), __uint_as_float(0x7FC00000u), __uint_as_float(0x7FC00000u), __uint_as_float(0xFFFFFFFFu))); // (.w: see DecodeGuidesKernel)
        StoreRGBA32F(worldPos, x, y, F4(__uint_as_float(0x7FC00000u)));
        return;
    }
    StoreRGBA32F(decoded, x, y, EncodeDecodedNormalRoughness(LoadNrRaw(packed, x, y)));
    const float z = Abs(LoadR32F(viewZ, x, y) * viewZScale);
    const float2 clip = F2(float(x) + 0.5f, float(y) + 0.5f) * rectSizeInv * 2.0f - 1.0f;
    const float3 d = frustumForward + frustumRight * clip.x - frustumUp * clip.y; // relax_device.h WorldPosFromClip, same operation order
    StoreRGBA32F(worldPos, x, y, F4(z * d.x, z * d.y, z * d.z, z));
}

void LaunchDecodeGuidesRelax(const PassArgs& a, const Plane& packed, const Plane& viewZ, const Plane& decoded, const Plane& worldPos, const void* relaxConstants) {
    const nrdc::RelaxConstants& c = *(const nrdc::RelaxConstants*)relaxConstants;
    const GuideRows rows = MakeGuideRows(a, packed);
    if (rows.launchEnd <= rows.launchBegin)
        return;
    LaunchPass(a, DecodeGuidesRelaxKernel, dim3((unsigned)((packed.w + 255) / 256), (unsigned)(rows.launchEnd - rows.launchBegin), 1), dim3(256), packed, viewZ, decoded, worldPos,
        make_float3(c.gFrustumRight.x, c.gFrustumRight.y, c.gFrustumRight.z), make_float3(c.gFrustumUp.x, c.gFrustumUp.y, c.gFrustumUp.z),
        make_float3(c.gFrustumForward.x, c.gFrustumForward.y, c.gFrustumForward.z), make_float2(c.gRectSizeInv.x, c.gRectSizeInv.y), c.gViewZScale, rows);
}

// ---- guide decode + tile classification in one launch (round 5) -------------------------------------------------------------------------------------
// The first pass of a REBLUR / RELAX list (ClassifyTiles: "all sky" per 16x16 tile over IN_VIEWZ) reads nothing the decode kernels write and streams the same
// IN_VIEWZ. When the guide planes are due for the whole frame the executor hands the packed plane to that pass's launcher (PassArgs::fuseGuidesFrom) and skips its own
// decode launch. A workgroup covers 64 x 16 pixels = four tiles side by side; wave w takes rows 4w .. 4w+3 of it, a lane one column: viewZ is read ONCE, every
// store of a wave is one contiguous run (1 KiB of a float4 plane, 256 B of the roughness-word plane), the four row loads of a lane are in flight together. The tile
// vote is a 16-lane AND (four xor-shuffles) per wave and tile, combined over the four waves through 16 ints of LDS. Values are those of the separate kernels (same
// expressions); saves a launch, a dependent-launch gap and 4 B/px.
template <bool RELAX>
__global__ __launch_bounds__(256) void DecodeGuidesClassifyKernel(Plane packed, Plane viewZ, Plane decoded, Plane guide, Plane roughnessWord, Plane tiles, float3 frustumRight, float3 frustumUp,
    float3 frustumForward, float2 rectSizeInv, float viewZScale, float denoisingRange, int tilesPerRow, int tileRows) {
    __shared__ int s_sky[4][4]; // [wave][tile of the workgroup]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane, y0 = blockIdx.y * 16 + wave * 4;
    bool allSky = true;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int y = y0 + k;
        const bool inside = x < packed.w && y < packed.h;
        const float zRaw = inside ? LoadR32F(viewZ, x, y) : 0.0f; // out-of-bounds load = 0 => a partial edge tile is never sky (as in the stand-alone classification)
        allSky = allSky && (RELAX ? Abs(zRaw) : Abs(zRaw * viewZScale)) > denoisingRange;
        if (!inside)
            continue;
        const float4 d = EncodeDecodedNormalRoughness(LoadNrRaw(packed, x, y));
        const float z = Abs(zRaw * viewZScale);
        if (RELAX) {
            StoreRGBA32F(decoded, x, y, d);
            const float2 clip = F2(float(x) + 0.5f, float(y) + 0.5f) * rectSizeInv * 2.0f - 1.0f;
            const float3 dir = frustumForward + frustumRight * clip.x - frustumUp * clip.y; // DecodeGuidesRelaxKernel, same operation order
            StoreRGBA32F(guide, x, y, F4(z * dir.x, z * dir.y, z * dir.z, z));
        } else {
            StoreRGBA32F(guide, x, y, F4(d.x, d.y, d.z, z));
            StoreR32U(roughnessWord, x, y, AsUint(d.w)); // passes.h PassArgs::roughnessWord
        }
    }
    int sky = allSky ? 1 : 0;
    for (int m = 1; m < 16; m <<= 1) // over the 16 columns of one tile (the lanes of a wave that share lane >> 4)
        sky &= __shfl_xor(sky, m);
    if ((lane & 15) == 0)
        s_sky[wave][lane >> 4] = sky;
    __syncthreads();
    if (threadIdx.x < 4) {
        const int t = threadIdx.x, tx = blockIdx.x * 4 + t, ty = blockIdx.y;
        if (tx < tilesPerRow && ty < tileRows) // the tiles of the RECT; tiles beyond stay untouched
            StoreR8Unorm(tiles, tx, ty, (s_sky[0][t] & s_sky[1][t] & s_sky[2][t] & s_sky[3][t]) ? 1.0f : 0.0f);
    }
}

void LaunchDecodeGuidesClassify(const PassArgs& a, const Plane& viewZ, const Plane& tiles, const void* reblurConstants, int tilesPerRow, int tileRows) {
    const nrdc::ReblurConstants& c = *(const nrdc::ReblurConstants*)reblurConstants;
    const Plane& packed = a.fuseGuidesFrom;
    LaunchPass(a, (DecodeGuidesClassifyKernel<false>), dim3((unsigned)((packed.w + 63) / 64), (unsigned)((packed.h + 15) / 16), 1), dim3(256), packed, viewZ, a.decodedNormalRoughness, a.viewPos, a.roughnessWord, tiles,
        make_float3(0.0f, 0.0f, 0.0f), make_float3(0.0f, 0.0f, 0.0f), make_float3(0.0f, 0.0f, 0.0f), make_float2(0.0f, 0.0f), c.gViewZScale, c.gDenoisingRange, tilesPerRow, tileRows);
}

void LaunchDecodeGuidesClassifyRelax(const PassArgs& a, const Plane& viewZ, const Plane& tiles, const void* relaxConstants, int tilesPerRow, int tileRows) {
    const nrdc::RelaxConstants& c = *(const nrdc::RelaxConstants*)relaxConstants;
    const Plane& packed = a.fuseGuidesFrom;
    LaunchPass(a, (DecodeGuidesClassifyKernel<true>), dim3((unsigned)((packed.w + 63) / 64), (unsigned)((packed.h + 15) / 16), 1), dim3(256), packed, viewZ, a.decodedNormalRoughness, a.worldPosViewZ, Plane{}, tiles,
        make_float3(c.gFrustumRight.x, c.gFrustumRight.y, c.gFrustumRight.z), make_float3(c.gFrustumUp.x, c.gFrustumUp.y, c.gFrustumUp.z),
        make_float3(c.gFrustumForward.x, c.gFrustumForward.y, c.gFrustumForward.z), make_float2(c.gRectSizeInv.x, c.gRectSizeInv.y), c.gViewZScale, c.gDenoisingRange, tilesPerRow, tileRows);
}

// ---- shifted rect (CommonSettings::rectOrigin; reference Common.hlsli:200-206 WithRectOrigin / WithRectOffset) ------------------------------------
// The reference addresses its GUIDE inputs (IN_MV, IN_NORMAL_ROUGHNESS, IN_VIEWZ, the confidence / threshold-mix inputs, IN_BASECOLOR_METALNESS) at
// rectOrigin + pixel and everything else (noisy inputs, outputs, pool planes) at the pixel itself. The executor implements that without touching a pass:
// the guides are copied once per frame into internal planes with the rect moved to (0, 0) -- 17 B/px of streaming for the usual guide set -- and the
// passes see rectOrigin = 0. dst(x, y) = src(x + ox, y + oy); with back = 1 the copy runs the other way (IN_MV is an in/out plane).
__global__ __launch_bounds__(256) void ShiftPlaneKernel(Plane user, Plane shifted, int ox, int oy, uint32_t bytesPerTexel, uint32_t back) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x + ox >= user.w || y + oy >= user.h)
        return;
    const uint8_t* src = user.ptr + (size_t)(y + oy) * user.pitch + (size_t)(x + ox) * bytesPerTexel;
    uint8_t* dst = shifted.ptr + (size_t)y * shifted.pitch + (size_t)x * bytesPerTexel;
    if (back) {
        const uint8_t* t = dst;
        dst = (uint8_t*)src;
        src = t;
    }
    if (bytesPerTexel == 8)
        *(uint2*)dst = *(const uint2*)src;
    else if (bytesPerTexel == 4)
        *(uint32_t*)dst = *(const uint32_t*)src;
    else
        for (uint32_t i = 0; i < bytesPerTexel; i++)
            dst[i] = src[i];
}

void LaunchShiftPlane(const PassArgs& a, const Plane& user, const Plane& shifted, int ox, int oy, uint32_t bytesPerTexel, bool back) {
    LaunchPass(a, ShiftPlaneKernel, dim3((unsigned)((user.w + 255) / 256), (unsigned)user.h, 1), dim3(256), user, shifted, ox, oy, bytesPerTexel, back ? 1u : 0u);
}

// ---- Clear: zero every texel of the plane (row padding is never read). User planes may have any pitch / alignment, so a 16-byte
// chunk is written with one store only when it is whole and aligned, bytewise otherwise (ragged row ends, 1-pixel-wide frames)
__global__ __launch_bounds__(256) void ClearPlaneKernel(Plane out, uint32_t rowBytes, uint32_t firstRow) {
    const uint32_t begin = (blockIdx.x * 256u + threadIdx.x) * 16u;
    if (begin >= rowBytes)
        return;
    uint8_t* p = out.ptr + (size_t)(blockIdx.y + firstRow) * out.pitch + begin;
    const uint32_t n = rowBytes - begin < 16u ? rowBytes - begin : 16u;
    if (n == 16u && ((uintptr_t)p & 15u) == 0) {
        *(uint4*)p = make_uint4(0, 0, 0, 0);
    } else {
        for (uint32_t i = 0; i < n; i++)
            p[i] = 0;
    }
}

static const char* LaunchClear(const PassArgs& a) {
    const Plane& out = a.planes[0];
    const uint32_t rowBytes = (uint32_t)out.w * a.bytesPerTexel[0];
    if (rowBytes == 0 || out.h == 0)
        return nullptr;
    // always the whole plane: clears belong to restart frames, which the sharding planners run unsharded (reach unknown), and the plane may be a
    // down-sampled one whose rows are not the frame's rows
    dim3 grid((rowBytes + 4095) / 4096, (unsigned)out.h, 1);
    LaunchPass(a, ClearPlaneKernel, grid, dim3(256), out, rowBytes, 0u);
    return nullptr;
}

// columns [0, w) and rows [r0, r1) a REFERENCE pass covers: the dispatch grid (16x16 groups) clipped to the plane and to the rank's rows
static void CoveredArea(const PassArgs& a, const Plane& plane, int& w, int& r0, int& r1) {
    const int gw = a.gridWidth ? (int)a.gridWidth * 16 : plane.w, gh = a.gridHeight ? (int)a.gridHeight * 16 : plane.h;
    w = gw < plane.w ? gw : plane.w;
    const int h = gh < plane.h ? gh : plane.h;
    r0 = a.rowBegin < 0 ? 0 : (a.rowBegin > h ? h : a.rowBegin);
    r1 = a.rowEnd > h ? h : a.rowEnd;
}

// ---- REFERENCE accumulate: history = lerp(history, input, accumSpeed), in place -------------------------------------
// The reference launches 16x16 groups over the rect (Reference.hpp NRD_DECLARE_DIMS): the pass covers [0, 16 * gridWidth) x [0, 16 * gridHeight)
// clipped to the texture, which is what (limitW, rows) carry; a rank of a sharded frame covers its rows only.
__global__ __launch_bounds__(256) void ReferenceAccumulateKernel(Plane input, Plane history, float accumSpeed, int limitW, int firstRow) {
    int x = blockIdx.x * 256 + threadIdx.x;
    int y = blockIdx.y + firstRow;
    if (x >= limitW)
        return;
    float4 in = InBounds(input, x, y) ? LoadRGBA32F(input, x, y) : F4(0.0f);
    float4 h = LoadRGBA32F(history, x, y);
    // BASELINE.json specifies this accumulator bit-exactly: the sequential fp32 running mean hist + (in - hist) * a with THREE roundings, which is what
    // tests/test_reference.py computes in numpy independently of the oracle. Hence no fused multiply-add here, whatever the file's contraction mode.
    float4 r;
    {
#pragma clang fp contract(off)
        r.x = h.x + (in.x - h.x) * accumSpeed;
        r.y = h.y + (in.y - h.y) * accumSpeed;
        r.z = h.z + (in.z - h.z) * accumSpeed;
        r.w = h.w + (in.w - h.w) * accumSpeed;
    }
    StoreRGBA32F(history, x, y, r);
}

static const char* LaunchReferenceAccumulate(const PassArgs& a) {
    const auto* c = (const nrdc::ReferenceAccumulateConstants*)a.constants;
    const Plane& history = a.planes[1];
    int w, r0, r1;
    CoveredArea(a, history, w, r0, r1);
    if (w <= 0 || r1 <= r0)
        return nullptr;
    dim3 grid((unsigned)(w + 255) / 256, (unsigned)(r1 - r0), 1);
    LaunchPass(a, ReferenceAccumulateKernel, grid, dim3(256), a.planes[0], history, c->gAccumSpeed, w, r0);
    return nullptr;
}

// ---- REFERENCE copy: out = history where pixelUv.x > splitScreen -----------------------------------------------------
__global__ __launch_bounds__(256) void ReferenceCopyKernel(Plane history, Plane out, float rectSizeInvX, float splitScreen, int limitW, int firstRow) {
    int x = blockIdx.x * 256 + threadIdx.x;
    int y = blockIdx.y + firstRow;
    if (x >= limitW || !InBounds(history, x, y))
        return;
    float pixelUvX = (float(x) + 0.5f) * rectSizeInvX;
    if (pixelUvX > splitScreen)
        StoreRGBA32F(out, x, y, LoadRGBA32F(history, x, y));
}

static const char* LaunchReferenceCopy(const PassArgs& a) {
    const auto* c = (const nrdc::ReferenceCopyConstants*)a.constants;
    const Plane& out = a.planes[1];
    int w, r0, r1;
    CoveredArea(a, out, w, r0, r1);
    if (w <= 0 || r1 <= r0)
        return nullptr;
    dim3 grid((unsigned)(w + 255) / 256, (unsigned)(r1 - r0), 1);
    LaunchPass(a, ReferenceCopyKernel, grid, dim3(256), a.planes[0], out, c->gRectSizeInv.x, c->gSplitScreen, w, r0);
    return nullptr;
}

// ---- numerics probe (include/NRDHip.h: nrdHipEvalNumerics) ---------------------------------------------------------
__global__ __launch_bounds__(256) void EvalNumericsKernel(uint32_t op, const float* in1, const float* in2, float* out, uint32_t count) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count)
        return;
    float a = in1[i], b = in2 ? in2[i] : 0.0f, r = 0.0f;
    switch (op) {
        case 0: r = Exp2(a); break;
        case 1: r = Log2(a); break;
        case 2: r = Atan(a); break;
        case 3: r = Pow(a, b); break;
        case 4: r = HalfBitsToFloat(FloatToHalfBits(a)); break;
        case 5: r = Div(a, b); break;
        case 6: r = Sqrt(a); break;
        case 7: r = Rsqrt(a); break;
        case 8: r = NRD_DIV_1023(a); break;
        case 9: r = NRD_DIV_255(a); break;
        case 10: r = NRD_DIV_63(a); break;
        case 11: r = NRD_DIV_15(a); break;
        case 12: r = NRD_DIV_3(a); break;
        case 13: r = Exp(-0.66f * a * a); break; // GetGaussianWeight
        case 14: r = NRD_DIV_65535(a); break;
        case 15: r = NRD_DIV_32767(a); break;
        // the raw transcendental instructions
        case 16: r = __builtin_amdgcn_rcpf(a); break;
        case 17: r = __builtin_amdgcn_rsqf(a); break;
        case 18: r = __builtin_amdgcn_sqrtf(a); break;
        case 19: r = AsFloat(FloatsToHalf2Bits(a, b)); break; // v_cvt_pk_f16_f32: two fp16 conversions in one instruction
        case 20: r = Rcp(a); break;
        // round 5: the one-addition forms of 2^x for x <= 0 (nrdmath.h)
        case 21: r = Exp2NonPos(a); break;
        case 22: r = SatExp2(a); break;
        case 23: r = ExpNegAbs(a); break;
        case 24: r = Pow01(a, b); break;
        default: break;
    }
    out[i] = r;
}

// ---- streaming copy probe (include/NRDHip.h: nrdHipMeasureCopyBandwidth): one 16-byte element per lane, the grid covers the buffer -- the shape the
// pass kernels have (one pixel per lane, no grid-stride loop). tools/copy_bench.hip compared the alternatives on the device (profiles/r03_e_copy_bench.txt):
// this one reaches 6.2-6.3 TB/s on 1 GiB, grid-stride loops over 1024-16384 workgroups 4.1-5.4 TB/s, hipMemcpyAsync 4.8-5.5 TB/s.
__global__ __launch_bounds__(256) void CopyProbeKernel(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i < count)
        dst[i] = src[i];
}
void LaunchCopyProbe(const void* src, void* dst, uint64_t bytes, hipStream_t stream) {
    const uint64_t count = bytes / 16u;
    hipLaunchKernelGGL(CopyProbeKernel, dim3((unsigned)((count + 255u) / 256u)), dim3(256), 0, stream, (const uint4*)src, (uint4*)dst, count);
}

void LaunchEvalNumerics(uint32_t op, const float* in1, const float* in2, float* out, uint32_t count, hipStream_t stream) {
    hipLaunchKernelGGL(EvalNumericsKernel, dim3((count + 255) / 256), dim3(256), 0, stream, op, in1, in2, out, count);
}

const PassEntry* GetCommonPasses(uint32_t& num) {
    static const PassEntry kPasses[] = {
        {"Clear_Float.cs", LaunchClear},
        {"Clear_Uint.cs", LaunchClear},
        {"REFERENCE_TemporalAccumulation.cs", LaunchReferenceAccumulate},
        {"REFERENCE_Copy.cs", LaunchReferenceCopy},
    };
    num = sizeof(kPasses) / sizeof(kPasses[0]);
    return kPasses;
}

// ================================================================================================ motion bound of a row strip (multi-GPU)
// max over the denoised pixels of rows [rowBegin, rowEnd) of | reprojected row - row |, the surface-motion reprojection of the temporal passes (reference
// REBLUR_TemporalAccumulation.hlsli:136-150, RELAX_TemporalAccumulation.hlsli:560-575: previous position from IN_MV in its three conventions). One streaming pass
// over IN_VIEWZ + IN_MV (12 B per pixel), wave reduction + one atomicMax on the float's bits (non-negative floats order like their bit patterns).
__global__ __launch_bounds__(256) void MotionRowsKernel(Plane viewZ, Plane mv, MotionParams p, int rowBegin, int rowEnd, uint32_t* outMaxBits) {
    const int px = blockIdx.x * 64 + (threadIdx.x & 63), py = rowBegin + blockIdx.y * 4 + (threadIdx.x >> 6);
    // viewZ / mv: the user's planes, already offset to the rect origin by the caller
    float rows = 0.0f;
    if (px < p.rectW && py < rowEnd && py < p.rectH) {
        const float z = Abs(LoadR32F(viewZ, px, py) * p.viewZScale);
        if (!(z > p.denoisingRange)) {
            const float2 uv = F2((float(px) + 0.5f) * p.rectSizeInv[0], (float(py) + 0.5f) * p.rectSizeInv[1]);
            const float4 m = LoadRGBA16F(mv, px, py);
            float2 uvPrev = F2(uv.x + m.x * p.mvScale[0], uv.y + m.y * p.mvScale[1]);
            if (p.mvScale[3] != 0.0f) { // world-space motion: previous world position through the previous view-projection
                float3 X;
                if (p.relaxForm) {
                    const float cx = uv.x * 2.0f - 1.0f, cy = uv.y * 2.0f - 1.0f;
                    X = F3(z * (p.frustumForward[0] + p.frustumRight[0] * cx - p.frustumUp[0] * cy), z * (p.frustumForward[1] + p.frustumRight[1] * cx - p.frustumUp[1] * cy),
                        z * (p.frustumForward[2] + p.frustumRight[2] * cx - p.frustumUp[2] * cy));
                } else {
                    X = RotateVector(p.viewToWorld, ReconstructViewPosition(uv, F4(p.frustum[0], p.frustum[1], p.frustum[2], p.frustum[3]), z, 0.0f));
                }
                uvPrev = GetScreenUv(p.worldToClipPrev, F3(X.x + m.x * p.mvScale[0], X.y + m.y * p.mvScale[1], X.z + m.z * p.mvScale[2])); // behind the previous camera: far off screen
            }
            rows = Abs(uvPrev.y - uv.y) * p.rectHeightPrev;
            rows = rows == rows ? rows : 1.0e9f; // NaN motion vectors count as unbounded
        }
    }
    for (int o = 32; o >= 1; o >>= 1)
        rows = fmaxf(rows, __shfl_xor(rows, o));
    if ((threadIdx.x & 63) == 0 && rows > 0.0f)
        atomicMax(outMaxBits, __float_as_uint(rows));
}

void LaunchMotionRows(hipStream_t stream, const Plane& viewZ, const Plane& mv, const MotionParams& p, int rowBegin, int rowEnd, uint32_t* outMaxBits) {
    if (rowEnd <= rowBegin)
        return;
    hipLaunchKernelGGL(MotionRowsKernel, dim3((unsigned)(p.rectW + 63) / 64, (unsigned)(rowEnd - rowBegin + 3) / 4), dim3(256), 0, stream, viewZ, mv, p, rowBegin, rowEnd, outMaxBits);
}

} // namespace nrdhip
