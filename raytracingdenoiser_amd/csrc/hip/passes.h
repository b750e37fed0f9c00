// Pass registry of the HIP executor: one launcher per NRD pass (= per reference shader file). The executor resolves
// DispatchDesc::pipelineIndex -> PipelineDesc::shaderFileName -> launcher once at creation.
#pragma once

#include "planes.h"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

// Tuning knobs of A/B builds (tools/build_variant.py -DNRD_WAVES_<KERNEL>=n): the waves-per-SIMD target handed to the register allocator through
// __launch_bounds__(threads, n); 0 = no hint (the allocator keeps the occupancy the kernel reaches by itself). The defaults are the measured best.
#ifndef NRD_WAVES_REBLUR_SPATIAL
#define NRD_WAVES_REBLUR_SPATIAL 0
#endif
#ifndef NRD_WAVES_REBLUR_HF
#define NRD_WAVES_REBLUR_HF 5 // 111 -> 84 VGPRs without scratch (6 would need 12 B of scratch)
#endif
#ifndef NRD_WAVES_REBLUR_TS
#define NRD_WAVES_REBLUR_TS 0
#endif
#ifndef NRD_WAVES_RELAX_ATROUS
#define NRD_WAVES_RELAX_ATROUS 0
#endif
#ifndef NRD_WAVES_RELAX_ATROUS_SMEM
#define NRD_WAVES_RELAX_ATROUS_SMEM 0
#endif
#ifndef NRD_WAVES_RELAX_PREPASS
#define NRD_WAVES_RELAX_PREPASS 0
#endif
#ifndef NRD_WAVES_RELAX_TA
#define NRD_WAVES_RELAX_TA 3
#endif
#ifndef NRD_WAVES_RELAX_HC
#define NRD_WAVES_RELAX_HC 0
#endif
#ifndef NRD_WAVES_RELAX_HF
#define NRD_WAVES_RELAX_HF 0
#endif
#ifndef NRD_RELAX_PREPASS_UNROLL // unroll factor of the 8-tap loops of the RELAX pre-pass
#define NRD_RELAX_PREPASS_UNROLL 2
#endif

namespace nrdhip {

// One kernel launch of a pass as data: what the launchers hand to the executor instead of launching when a recorder is attached.
// The executor uses it for the pre-flight of a dispatch range (nothing is launched unless every pass of the range can be) and for
// HIP-graph execution (kernel nodes built from / updated with these records: executor.hip "graph mode").
// nrdHipMeasureMotionRows (multi-GPU: the history halo must cover the frame's reprojection): what the reduction kernel needs of the shared constants of either family
struct MotionParams {
    float worldToClipPrev[16];
    float viewToWorld[16];      // REBLUR form: X = viewToWorld * ( frustum-reconstructed view position )
    float frustum[4];
    float frustumRight[4], frustumUp[4], frustumForward[4]; // RELAX form: X = viewZ * ( forward + right * ndc.x - up * ndc.y )
    float mvScale[4];
    float rectSizeInv[2];
    float rectHeightPrev, viewZScale, denoisingRange;
    int rectW, rectH, relaxForm;
};
void LaunchMotionRows(hipStream_t stream, const Plane& viewZ, const Plane& mv, const MotionParams& p, int rowBegin, int rowEnd, uint32_t* outMaxBits);

struct LaunchRecord {
    const void* func;
    dim3 grid, block;
    std::vector<uint8_t> args;     // kernel arguments, each copied to a 16-byte aligned offset
    std::vector<uint32_t> offsets; // offset of every argument inside args
    void (*launch)(const LaunchRecord&, hipStream_t); // typed launch of this very record (eager path: the launchers themselves run once, in the pre-flight)
};
struct LaunchRecorder {
    bool keep = false; // false = pre-flight only: launches are counted, not stored
    uint32_t count = 0;
    std::vector<LaunchRecord> records;
};

// numeric values of the nrd::Format entries the format-dispatching launchers look at (checked against NRDDescs.h in executor.hip)
enum : uint8_t { FORMAT_RGBA8_UNORM = 8, FORMAT_R16_UNORM = 13, FORMAT_RGBA16_SNORM = 24, FORMAT_RGBA16_SFLOAT = 27 };

struct PassArgs {
    const Plane* planes;      // DispatchDesc::resources resolved to planes, same order (inputs then outputs)
    uint32_t planesNum;
    const uint8_t* bytesPerTexel; // per plane, for the few launchers shared between storage formats (SIGMA_Copy.cs)
    const uint8_t* formats;       // per plane, nrd::Format (REBLUR hit-distance reconstruction / split screen serve three signal kinds)
    const void* constants;    // DispatchDesc::constantBufferData
    uint32_t constantsSize;
    uint32_t gridWidth, gridHeight; // DispatchDesc::gridWidth / gridHeight (the reference's thread-group counts: the area the pass covers)
    hipStream_t stream;
    // rows [rowBegin, rowEnd) this rank has to produce in this pass (multi-GPU row-strip sharding; the whole frame by default).
    // Pixels outside are left untouched; planes stay full-size, so all neighbourhood reads keep their single-GPU meaning.
    int rowBegin, rowEnd;
    // executor-internal cache of IN_NORMAL_ROUGHNESS decoded once per frame (float4 per texel: N.xyz, roughness float | material
    // bits -- reblur_device.h "decoded guides"); ptr == nullptr when the dispatch list does not bind IN_NORMAL_ROUGHNESS
    Plane decodedNormalRoughness;
    // executor-internal guide plane of the RELAX lists (float4 per pixel: world position, viewZ), written together with the decoded normals once
    // per frame; read by the taps of the pre-pass and of the a-trous iterations; same layout as decodedNormalRoughness; ptr == nullptr outside RELAX lists
    Plane worldPosViewZ;
    // executor-internal guide plane of the REBLUR lists (float4 per pixel: decoded normal, viewZ = |z * gViewZScale|), written together with the decoded
    // normals once per frame; same pitch / size as decodedNormalRoughness, so one texel offset serves both. With it a diffuse tap of the spatial passes
    // reads its guides in one 16-byte load. ptr == nullptr outside REBLUR lists.
    Plane viewPos;
    // ptr != nullptr only for the tile-classification pass of a REBLUR / RELAX list whose guide planes are due for the whole frame: the launcher writes the planes above
    // from this packed IN_NORMAL_ROUGHNESS plane in the same kernel that classifies the tiles (kernels_common.hip DecodeGuidesClassifyKernel); the executor then
    // launches no decode kernel of its own
    Plane fuseGuidesFrom;
    // executor-internal compact copy (4 B per pixel, pitch = decodedNormalRoughness.pitch / 4) of the w word of the decoded normals -- roughness float | material bits --
    // of the REBLUR lists: what a SPECULAR tap of the spatial passes needs besides its (normal, viewZ) texel. Reading it from the 16-byte texels of decodedNormalRoughness
    // pulled that whole plane (59 MB at 1440p) through the L2 for 4 bytes in 16; from here it is a quarter of that (round 5: pre-pass -5 %, Blur / PostBlur -2 %).
    Plane roughnessWord;
    // executor-internal scratch, one byte per 32x8-pixel workgroup tile of the full-resolution planes: a pass that runs as a fast kernel plus a fallback kernel
    // (REBLUR TemporalAccumulation with its LDS window) hands the tiles the fast kernel declined to the fallback through it. Written completely by the fast
    // kernel before the fallback reads it (stream order), so it needs no clearing.
    Plane tileFlags;
    // executor-owned int[3], filled by the launcher of such a split pass: {tile columns, first tile row, end tile row} the fast kernel covered in this launch --
    // the rect's tiles in this rank's rows. Only there do the flags describe the last frame (nrdHipGetTileFallbackStats); bytes outside keep stale values of a
    // larger rect / other ranks' rows, which nothing reads.
    int* windowRegion = nullptr;
    // multi-GPU (round 6): 4 bytes of device memory in which the temporal passes leave, as the bits of a non-negative float, the largest number of ROWS a pixel of theirs read last
    // frame's planes away from its own row -- surface motion, virtual (specular) motion and the look-back taps behind it (reblur_device.h TrackHistoryReach). nullptr: not tracked.
    // A row-strip host holds it against the history halo it exchanged (nrdHipSetHistoryReachWord).
    uint32_t* historyReachWord = nullptr;
    // non-null: the launcher performs all its checks and hands its launch(es) to the recorder instead of enqueueing them
    LaunchRecorder* recorder = nullptr;
};

namespace detail {
template <typename T>
inline void PackArg(LaunchRecord& r, const T& v) {
    static_assert(std::is_trivially_copyable<T>::value, "kernel arguments are PODs");
    size_t off = (r.args.size() + 15u) & ~(size_t)15u;
    r.args.resize(off + sizeof(T));
    memcpy(r.args.data() + off, &v, sizeof(T));
    r.offsets.push_back((uint32_t)off);
}
template <typename... KArgs, size_t... I>
inline void ReplayImpl(const LaunchRecord& r, hipStream_t stream, std::index_sequence<I...>) {
    hipLaunchKernelGGL((void (*)(KArgs...))r.func, r.grid, r.block, 0, stream, (*(const KArgs*)(r.args.data() + r.offsets[I]))...);
}
template <typename... KArgs>
inline void Replay(const LaunchRecord& r, hipStream_t stream) {
    ReplayImpl<KArgs...>(r, stream, std::index_sequence_for<KArgs...>());
}
} // namespace detail

// every pass launch goes through here: launcher(args) -> LaunchPass(args, kernel, grid, block, kernel arguments...)
template <typename... KArgs, typename... Args>
inline void LaunchPass(const PassArgs& a, void (*kernel)(KArgs...), dim3 grid, dim3 block, const Args&... args) {
    if (!a.recorder) {
        hipLaunchKernelGGL(kernel, grid, block, 0, a.stream, KArgs(args)...);
        return;
    }
    a.recorder->count++;
    if (!a.recorder->keep)
        return;
    LaunchRecord r;
    r.func = (const void*)kernel;
    r.grid = grid;
    r.block = block;
    r.launch = &detail::Replay<KArgs...>;
    (void)std::initializer_list<int>{(detail::PackArg<KArgs>(r, KArgs(args)), 0)...};
    a.recorder->records.push_back(std::move(r));
}

// returns nullptr on success, or a static message if the dispatch cannot be executed by this build (nothing is launched then)
typedef const char* (*PassLauncher)(const PassArgs& args);

struct PassEntry {
    const char* shaderFileName;
    PassLauncher launch;
};

// each kernels_*.hip exports its table
const PassEntry* GetCommonPasses(uint32_t& num);
const PassEntry* GetReblurPasses(uint32_t& num);
const PassEntry* GetSigmaPasses(uint32_t& num);
const PassEntry* GetRelaxPasses(uint32_t& num);
const PassEntry* GetValidationPasses(uint32_t& num);

// same, plus the REBLUR view-position guide plane from IN_VIEWZ and the frame's REBLUR constants (kernels_common.hip)
void LaunchDecodeGuides(const PassArgs& a, const Plane& packed, const Plane& viewZ, const Plane& viewPos, const Plane& roughnessWord, const void* reblurConstants);
// same for RELAX lists: the (world position, viewZ) plane from IN_VIEWZ and the frame's RELAX constants
void LaunchDecodeGuidesRelax(const PassArgs& a, const Plane& packed, const Plane& viewZ, const Plane& decoded, const Plane& worldPos, const void* relaxConstants);

// tile classification + guide decode in one launch (PassArgs::fuseGuidesFrom set): called by the ClassifyTiles launchers of REBLUR / RELAX with their pass's planes
void LaunchDecodeGuidesClassify(const PassArgs& a, const Plane& viewZ, const Plane& tiles, const void* reblurConstants, int tilesPerRow, int tileRows);
void LaunchDecodeGuidesClassifyRelax(const PassArgs& a, const Plane& viewZ, const Plane& tiles, const void* relaxConstants, int tilesPerRow, int tileRows);

// copies a user guide plane into its rect-at-origin twin (or back): kernels_common.hip "shifted rect"
void LaunchShiftPlane(const PassArgs& a, const Plane& user, const Plane& shifted, int ox, int oy, uint32_t bytesPerTexel, bool back);

inline dim3 GridFor(int w, int h, int tileW, int tileH) { return dim3((unsigned)((w + tileW - 1) / tileW), (unsigned)((h + tileH - 1) / tileH), 1); }

// Grid covering rows [rowBegin, rowEnd) clipped to [0, h) with tileH-row blocks that stay aligned to the full-frame tiling;
// firstBlockY is added to blockIdx.y inside the kernel.
// XCD-aware tile order. The dispatcher places workgroup b on XCD b % 8, each XCD has its own 4 MiB L2, and the tap loops of the spatial passes
// reach 30-60 px (a-trous: up to 20 px) around a 32x8 tile: with the plain order every XCD touches every 8th tile of a row and fetches the
// halos of all of them from HBM (counter traffic 2-4x the algorithmic bytes, profiles/r01_*). With gridDim.x a multiple of 8 and
//     tileX = (blockIdx.x % 8) * bandTiles + blockIdx.x / 8,      bandTiles = ceil(tilesX / 8)
// XCD k owns the vertical band of tile columns [k * bandTiles, (k + 1) * bandTiles): neighbouring tiles share their halos in ONE L2, all eight
// XCDs sweep the frame top to bottom together, and a horizontal sky band costs every XCD the same (the r01 experiment with bands of tile ROWS lost
// 1.46x to that imbalance). Placement is a speed matter only: results do not depend on it. NRD_HIP_XCD_BANDS=0 restores the plain order.
// NRD_HIP_XCD_BANDS: 0 = plain order; 1 = one band per XCD; n > 1 = stripes of n tile columns dealt round-robin to the XCDs (several stripes per XCD);
// default (-1) = stripes whose width is picked per launch: the widest of 4, 3, 2 tile columns that gives every XCD the same number of stripes.
// Measured (profiles/r02_b_*, r02_c_*): one band per XCD cuts the counter traffic of the sparse-tap passes from 2-4x to 1.1-1.3x the algorithmic
// bytes but runs 3-5 % SLOWER than the plain order (the passes are not HBM-bound, and a band's cost depends on what it shows); stripes of 2-3 tile
// columns are the fastest order (REBLUR 1440p 1.014 ms vs 1.020 plain vs 1.056 banded; RELAX 4K 3.32 vs 3.37 vs 3.55).
inline int XcdBandsSetting() {
    static const int v = getenv("NRD_HIP_XCD_BANDS") ? atoi(getenv("NRD_HIP_XCD_BANDS")) : -1;
    return v;
}
inline bool XcdBandsEnabled() { return XcdBandsSetting() != 0; }
struct RowGrid {
    dim3 grid;
    int firstBlockY, rowBegin, rowEnd, bandTiles, stripeTiles;
};
inline RowGrid GridForRows(int w, int h, int tileW, int tileH, int rowBegin, int rowEnd) {
    RowGrid g;
    g.rowBegin = rowBegin < 0 ? 0 : rowBegin;
    g.rowEnd = rowEnd > h ? h : rowEnd;
    if (g.rowEnd < g.rowBegin)
        g.rowEnd = g.rowBegin;
    g.firstBlockY = g.rowBegin / tileH;
    int lastBlockY = (g.rowEnd + tileH - 1) / tileH;
    int ny = lastBlockY - g.firstBlockY;
    const int tilesX = (w + tileW - 1) / tileW;
    g.bandTiles = (XcdBandsEnabled() && tilesX >= 16) ? (tilesX + 7) / 8 : 0;
    g.stripeTiles = g.bandTiles;
    int stripe = XcdBandsSetting();
    if (g.bandTiles && stripe < 0) { // automatic: equal stripe counts per XCD if possible
        stripe = 2;
        for (int b = 4; b >= 2; b--)
            if (tilesX % (8 * b) == 0) {
                stripe = b;
                break;
            }
    }
    if (g.bandTiles && stripe > 1) { // stripes: the XCD's tile columns are dealt out in groups of stripeTiles
        g.stripeTiles = stripe < g.bandTiles ? stripe : g.bandTiles;
        g.bandTiles = ((g.bandTiles + g.stripeTiles - 1) / g.stripeTiles) * g.stripeTiles;
    }
    g.grid = dim3((unsigned)(g.bandTiles ? 8 * g.bandTiles : tilesX), (unsigned)(ny > 0 ? ny : 1), 1);
    return g;
}
struct RowRange { // kernel argument
    int firstBlockY, rowBegin, rowEnd;
    int bandTiles;   // tile columns per XCD; 0 = plain tile order
    int stripeTiles; // width of a stripe (== bandTiles: one band per XCD)
};
inline RowRange MakeRowRange(const RowGrid& g) { return RowRange{g.firstBlockY, g.rowBegin, g.rowEnd, g.bandTiles, g.stripeTiles}; }
// tile column of this workgroup (may lie beyond the frame: such workgroups find all their pixels outside the rect)
// Tile row of this workgroup. bottomUp: the grid walks the tile rows from the bottom of the frame, so that -- with the sky at the top, as usual -- the workgroups dispatched
// last are the ones that leave at the tile test, and the stream of sky tiles runs in the shadow of the draining geometry tiles instead of in front of them. Measured per
// pass (profiles/r04_v_*, r04_w_*): the temporal-accumulation kernels and the RELAX passes gain 1-1.5 %, the REBLUR spatial passes lose 1-2 % (their L2 working set follows
// the dispatch front): each kernel names its order. NRD_REVERSE_TILE_ROWS = 0 / 1 forces top-down / bottom-up everywhere (A/B).
#ifndef NRD_ALT_TILE_ORDER
#define NRD_ALT_TILE_ORDER 0 // 1: REBLUR's Blur and TemporalStabilization walk the tile rows bottom-up (kernels_reblur_spatial.hip: the reference's alternating CTA order)
#endif
#ifndef NRD_REVERSE_TILE_ROWS
#define NRD_REVERSE_TILE_ROWS -1
#endif
__device__ __forceinline__ int BlockTileY(const RowRange& r, bool bottomUp = false) {
    const bool reverse = NRD_REVERSE_TILE_ROWS < 0 ? bottomUp : NRD_REVERSE_TILE_ROWS != 0;
    return r.firstBlockY + (reverse ? (int)(gridDim.y - 1u - blockIdx.y) : (int)blockIdx.y);
}
// NRD_XCD_STAIRCASE_ROWS = K > 0 (A/B switch, off): in EVERY pass the stripes move on by one XCD every K tile rows (see BlockTileXRotated below, which is that with K = 1 for the passes
// that want it unconditionally)
#ifndef NRD_XCD_STAIRCASE_ROWS
#define NRD_XCD_STAIRCASE_ROWS 0
#endif
__device__ __forceinline__ int BlockTileXPlain(const RowRange& r) {
    const unsigned bx = blockIdx.x;
    if (!r.bandTiles)
        return (int)bx;
    const unsigned xcd = bx & 7u, j = bx >> 3; // the j-th tile column of this XCD
    if (r.stripeTiles == r.bandTiles)
        return (int)(xcd * (unsigned)r.bandTiles + j);
    const unsigned stripe = j / (unsigned)r.stripeTiles, within = j - stripe * (unsigned)r.stripeTiles;
    return (int)((stripe * 8u + xcd) * (unsigned)r.stripeTiles + within);
}
__device__ __forceinline__ int BlockTileX(const RowRange& r) {
    const int t = BlockTileXPlain(r);
    if (NRD_XCD_STAIRCASE_ROWS == 0 || !r.bandTiles)
        return t;
    const unsigned blockY = (unsigned)BlockTileY(r);
    return (int)(((unsigned)t + (blockY / (unsigned)(NRD_XCD_STAIRCASE_ROWS ? NRD_XCD_STAIRCASE_ROWS : 1)) * (unsigned)r.stripeTiles) % gridDim.x);
}

// The same with the tile columns of tile row `blockY` moved on by one stripe per row: the stripes of one XCD form a staircase instead of a column. For passes whose work is
// concentrated in a vertical feature -- the columns that enter the screen under a yawing camera, where RELAX HistoryFix does all of its work -- the XCD-aware order
// puts that feature on ONE XCD (32 of the 256 CUs); rotated, every XCD gets every eighth tile row of it. Costs the L2 locality of vertical neighbours: only for passes
// that read no vertical halo worth keeping.
__device__ __forceinline__ int BlockTileXRotated(const RowRange& r, int blockY) {
    const int t = BlockTileXPlain(r);
    if (!r.bandTiles)
        return t;
    return (int)(((unsigned)t + (unsigned)blockY * (unsigned)r.stripeTiles) % gridDim.x);
}

} // namespace nrdhip
