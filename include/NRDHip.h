// HIP execution back-end of the NRD pass chain for AMD Instinct MI355X (gfx950) -- thin C-ABI.
//
// The reference library only DESCRIBES dispatches; executing them is the job of its integration layer
// (reference Integration/NRDIntegration.h:83-165, NRDIntegration.hpp). This header is what replaces that layer:
//
//   reference nrd::Integration::Initialize   (NRDIntegration.hpp:93-139, :292-454: creates the pool textures)   -> nrdHipCreateExecutor
//   reference nrd::Integration::Destroy      (NRDIntegration.hpp:805-...)                                       -> nrdHipDestroyExecutor
//   reference UserPool / Integration_SetResource (NRDIntegration.h:37-60: app textures by ResourceType slot)     -> nrdHipBindResource
//   reference nrd::Integration::Denoise      (NRDIntegration.hpp:516-623: GetComputeDispatches + loop)           -> nrdHipDenoise
//   reference nrd::Integration::Dispatch     (NRDIntegration.hpp:625-803: bind + constants + CmdDispatch)        -> nrdHipExecuteDispatches
//   reference nrd::Integration::GetTotalMemoryUsageInMb (NRDIntegration.h:120-127)                               -> nrdHipGetPoolMemoryUsage
//
// Plain C types only: device pointers are void*, the stream is a hipStream_t passed as void*, formats and resource
// slots are the numeric values of nrd::Format / nrd::ResourceType (include/NRDDescs.h). Every function returns an
// nrd::Result value as uint32_t (0 = SUCCESS). Nothing here synchronises the device: launches are enqueued on the
// executor's stream in dispatch order.
#pragma once

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct NrdHipExecutor NrdHipExecutor;

// A pitched 2D plane in device memory ("texture" of the reference). Texel (x, y) lives at
// data + y * rowPitchBytes + x * bytesPerTexel(format). rowPitchBytes must be a multiple of the texel size.
typedef struct NrdHipPlaneDesc {
    void* data;
    uint32_t rowPitchBytes;
    uint32_t format; // nrd::Format
    uint16_t width;
    uint16_t height;
} NrdHipPlaneDesc;

// Creates the executor for an nrd::Instance (include/NRD.h) and allocates its permanent + transient pool planes
// (one hipMalloc arena, 256-byte aligned rows) for textures of resourceWidth x resourceHeight.
// "instance" is an nrd::Instance*; "hipStream" is a hipStream_t (NULL = default stream).
// Size limit: planes are addressed with 32-bit byte offsets, so a plane of 16-byte texels must stay below 4 GiB -- every size up to 16384 x 16383 (or 65535 x 4095) is accepted,
// larger ones return UNSUPPORTED (as does a user plane whose row pitch reaches 16 MiB or whose pitch x height reaches 4 GiB, in nrdHipBindResource).
uint32_t nrdHipCreateExecutor(void* instance, uint16_t resourceWidth, uint16_t resourceHeight, void* hipStream, NrdHipExecutor** executor);
void nrdHipDestroyExecutor(NrdHipExecutor* executor);

// Same, but the pool arena is provided (and owned) by the caller, which is the reference's own model ("NRD allocates no GPU
// memory": reference Integration creates the pool textures from InstanceDesc). "arena" must be device memory of at least
// nrdHipGetArenaSize() bytes, 256-byte aligned; it is zero-filled on the stream. Lets a host that owns the memory (e.g. a
// tensor library) alias pool planes for collectives.
uint64_t nrdHipGetArenaSize(void* instance, uint16_t resourceWidth, uint16_t resourceHeight);
uint32_t nrdHipCreateExecutorWithArena(void* instance, uint16_t resourceWidth, uint16_t resourceHeight, void* hipStream, void* arena, uint64_t arenaSize, NrdHipExecutor** executor);

// Binds an application plane to an IN_* / OUT_* slot. The bytes are not copied; the binding persists until rebound.
// Formats accepted in this build (anything else -> UNSUPPORTED):
//   IN_MV RGBA16_SFLOAT | IN_NORMAL_ROUGHNESS R10_G10_B10_A2_UNORM (the format of the library's normal encoding, nrd::GetLibraryDesc().normalEncoding: RGBA8_UNORM /
//   RGBA8_SNORM / R10_G10_B10_A2_UNORM / RGBA16_UNORM / RGBA16_SNORM for encodings 0..4 -- a build option as in the reference) | IN_VIEWZ R32_SFLOAT
//   IN/OUT_{DIFF,SPEC}_RADIANCE_HITDIST RGBA16_SFLOAT | IN/OUT_{DIFF,SPEC}_SH0, _SH1 RGBA16_SFLOAT (REBLUR / RELAX SH variants)
//   IN/OUT_{DIFF,SPEC}_HITDIST R16_UNORM (REBLUR occlusion family) | IN/OUT_DIFF_DIRECTION_HITDIST RGBA16_SNORM | IN_PENUMBRA R16_SFLOAT | IN_TRANSLUCENCY, IN_BASECOLOR_METALNESS RGBA8_UNORM
//   OUT_SHADOW_TRANSLUCENCY R8_UNORM (SIGMA_SHADOW) or RGBA8_UNORM (an instance holding SIGMA_SHADOW_TRANSLUCENCY)
//   IN_SIGNAL / OUT_SIGNAL RGBA32_SFLOAT | IN_{DIFF,SPEC}_CONFIDENCE, IN_DISOCCLUSION_THRESHOLD_MIX R8_UNORM | OUT_VALIDATION RGBA8_UNORM
// include/NRD.hip.h has the device functions that produce / consume these encodings (the NRD.hlsli front-end and back-end).
uint32_t nrdHipBindResource(NrdHipExecutor* executor, uint32_t resourceType, const NrdHipPlaneDesc* plane);

// Describes a pool plane (resourceType = TRANSIENT_POOL or PERMANENT_POOL, index into InstanceDesc::*Pool).
// For tooling and parity tests (history inspection); the memory stays owned by the executor.
uint32_t nrdHipGetPoolPlane(NrdHipExecutor* executor, uint32_t resourceType, uint32_t indexInPool, NrdHipPlaneDesc* plane);

// Executes a dispatch list obtained from nrd::GetComputeDispatches on the executor's stream, in order.
// "dispatchDescs" is a const nrd::DispatchDesc*. All-or-nothing: the whole list is checked first (a HIP kernel exists for every pass, all
// resources are bound, every pass accepts its constants) and an error (UNSUPPORTED / INVALID_ARGUMENT + nrdHipGetLastError) is returned
// BEFORE anything is enqueued.
uint32_t nrdHipExecuteDispatches(NrdHipExecutor* executor, const void* dispatchDescs, uint32_t dispatchDescsNum);

// nrd::GetComputeDispatches(identifiers) followed by nrdHipExecuteDispatches: one denoised frame.
// nrd::SetCommonSettings / SetDenoiserSettings must have been called for this frame.
uint32_t nrdHipDenoise(NrdHipExecutor* executor, const uint32_t* identifiers, uint32_t identifiersNum);

// Multi-GPU row-strip sharding: restricts this executor to PRODUCING rows [rowBegin, rowEnd) of the final outputs and of the
// permanent (history) planes. Planes stay full-size; every pass is launched on the strip extended by the cumulative reach of the
// passes that follow it in the dispatch list (temporal stabilization 1 row, post-blur / blur 2 x their maximum radius,
// history fix 2 x stride, ...), so the owned rows are bit-identical to a single-GPU run provided the caller makes the other
// ranks' owned rows of the permanent planes available before the next frame (one in-place all-gather per plane: owned strips
// are contiguous row ranges). rowBegin = 0, rowEnd >= height restores whole-frame execution. Passes the executor cannot bound
// (anything but the REBLUR chain in this build) always run on the whole frame.
uint32_t nrdHipSetOwnedRows(NrdHipExecutor* executor, uint32_t rowBegin, uint32_t rowEnd);

// Finer-grained sharding control for a host that exchanges halos between passes (raytracingdenoiser_amd/sharding.py, HaloSharder):
//   nrdHipGetDispatchReach     reachRows[i] = how many rows above / below a pixel dispatch i reads from planes written earlier in the same
//                              frame (0 = own row only, -1 = unknown: the pass must run on the whole frame); "instance" is an nrd::Instance*
//                              (host-only: needs no device)
//   nrdHipExecuteDispatchRange executes dispatches [first, first + count) of the list; rowBegin[i] / rowEnd[i] (indexed by the absolute
//                              dispatch index; NULL or rowBegin[i] < 0 = whole frame) are the rows dispatch i has to produce. Ranges of one
//                              list must be executed in order starting at first = 0 (the per-frame caches are rebuilt there).
uint32_t nrdHipGetDispatchReach(void* instance, const void* dispatchDescs, uint32_t dispatchDescsNum, int32_t* reachRows);
uint32_t nrdHipExecuteDispatchRange(NrdHipExecutor* executor, const void* dispatchDescs, uint32_t dispatchDescsNum, uint32_t first, uint32_t count, const int32_t* rowBegin,
    const int32_t* rowEnd);

// The halo-exchange plan of one dispatch list for rank `rank` of `world` ranks owning the row strips [stripBounds[r], stripBounds[r + 1]) (host-only,
// no device needed; the C++ counterpart of raytracingdenoiser_amd/sharding.py plan_halo_exchange, for hosts that drive RCCL themselves):
//   steps[s]   dispatches [firstDispatch, firstDispatch + dispatchCount) run between two exchanges; before them the rank sends / receives, for each of
//              items[firstItem .. firstItem + itemCount), `widthRows` rows on either side of its strip boundaries to / from ranks rank - 1 and rank + 1
//              (send its own top / bottom rows, receive into the rows just above / below its strip); the first `earlyCount` dispatches of the step touch
//              none of these planes and may run while the transfers are in flight
//   rowBegin / rowEnd (length dispatchDescsNum)   the rows every dispatch has to produce, ready for nrdHipExecuteDispatchRange (-1 = whole frame)
//   info->fallback = 1   the list cannot be sharded (unknown reach, halo wider than a strip): every rank runs the whole frame, after having received
//                        the other ranks' strips of the carried-over planes if the previous frame was sharded
// Returns INVALID_ARGUMENT with info->stepsNum / itemsNum set when the capacities are too small.
typedef struct NrdHipHaloItem {
    uint32_t resourceType; // nrd::ResourceType: TRANSIENT_POOL / PERMANENT_POOL (+ indexInPool) or an OUT_* slot doubling as history
    uint32_t indexInPool;
    uint32_t widthRows;
} NrdHipHaloItem;
typedef struct NrdHipHaloStep {
    uint32_t firstDispatch, dispatchCount, earlyCount, firstItem, itemCount;
} NrdHipHaloStep;
typedef struct NrdHipHaloPlanInfo {
    uint32_t fallback, stepsNum, itemsNum;
} NrdHipHaloPlanInfo;
uint32_t nrdHipPlanHaloExchange(void* instance, const void* dispatchDescs, uint32_t dispatchDescsNum, const uint32_t* stripBounds, uint32_t world, uint32_t rank, uint32_t height,
    uint32_t maxMotionRows, uint32_t exchangeThreshold, int32_t* rowBegin, int32_t* rowEnd, NrdHipHaloStep* steps, uint32_t stepsCapacity, NrdHipHaloItem* items, uint32_t itemsCapacity,
    NrdHipHaloPlanInfo* info);

// The motion side of the sharding contract, measured on the device: *maxRows = the largest vertical distance (in rows of the previous rect) over which the
// surface-motion reprojection of the temporal passes (reference REBLUR_TemporalAccumulation.hlsli:136-150, RELAX_TemporalAccumulation.hlsli:560-575,
// SIGMA_TemporalStabilization.hlsli) moves a denoised pixel of rows [rowBegin, rowEnd) of the rect -- from the bound IN_VIEWZ / IN_MV planes and the constants
// of the given dispatch list (the list of THIS frame, before it is executed). One streaming kernel over the strip (12 B per pixel), a 4-byte read-back and
// a stream synchronisation. A rank calls it on its own strip, takes the MAX over ranks, and runs the frame unsharded when the result + 2 rows (bicubic
// footprint) does not fit the history halo it planned with (maxMotionRows below). 0 for lists without a temporal denoiser; pixels whose previous position
// lies behind the previous camera (or whose motion vector is NaN) report a huge value on purpose. The value bounds the SURFACE motion only: hosts double it for the
// virtual motion of specular reflections, which is a heuristic (a curved reflector can exceed it), not a check.
uint32_t nrdHipMeasureMotionRows(NrdHipExecutor* executor, const void* dispatchDescs, uint32_t dispatchDescsNum, uint32_t rowBegin, uint32_t rowEnd, float* maxRows);
// The same measurement without the host round trip: enqueued on the executor's stream, the result (a float) lands in 4 bytes of the CALLER'S device memory in stream order -- the
// buffer a multi-GPU host hands to its MAX all-reduce (RCCL reads it on the device); the host then synchronises once, on the reduced value.
// What the measurement above cannot bound: the temporal passes of the SPECULAR denoisers also read last frame's planes at the virtual-motion position and at look-back taps behind
// it, whose distance depends on hit distances and surface curvature. So the kernels report it: with a device word registered here, every temporal pass (REBLUR / RELAX
// TemporalAccumulation, SIGMA TemporalStabilization) leaves in it -- atomicMax on the bits of a non-negative float -- the largest number of rows one of its pixels read away from its
// own row (sample positions only: add 3 rows for the bicubic footprint). The word belongs to the caller: clear it in stream order before a frame, read it (or MAX-all-reduce it over
// the ranks) after, hold it against the history halo that frame was run with, and size the next frame's decision with it. nullptr (the default) switches the tracking off.
uint32_t nrdHipSetHistoryReachWord(NrdHipExecutor* executor, void* deviceWord);
uint32_t nrdHipMeasureMotionRowsAsync(NrdHipExecutor* executor, const void* dispatchDescs, uint32_t dispatchDescsNum, uint32_t rowBegin, uint32_t rowEnd, void* deviceMaxRows);

// Per-pass GPU timing. When enabled, every dispatch is bracketed by hipEvents on the executor's stream.
// nrdHipCollectPassTimings synchronises the stream, folds all brackets recorded since the last collect into per-pipeline
// totals and returns the number of pipelines written: pipelineIndices[i] (index into InstanceDesc::pipelines),
// milliseconds[i] (sum of durations) and launches[i] (count). One more row than there are pipelines: index == InstanceDesc::pipelinesNum is the per-frame guide
// preparation (the decode / rect-shift kernels in front of the first pass of a list; absent when the list's tile-classification kernel writes the guide planes itself --
// whole-frame decode of a REBLUR-only or RELAX-only list without a shifted rect: that time is then part of the *_ClassifyTiles.cs row). Pass capacity >= InstanceDesc::pipelinesNum + 1.
uint32_t nrdHipSetProfiling(NrdHipExecutor* executor, uint32_t enable);
uint32_t nrdHipCollectPassTimings(NrdHipExecutor* executor, uint32_t* pipelineIndices, double* milliseconds, uint32_t* launches, uint32_t capacity, uint32_t* written);

// Graph mode (the HIP-graph counterpart of the command list the reference integration records per Denoise call, reference
// Integration/NRDIntegration.hpp:516-623): with enable != 0 the kernel launches of a dispatch range are not enqueued one by one but as ONE
// hipGraph launch. The executable graph is built once per topology (the sequence of kernels of the range; ping-pong and per-frame constants do
// not change it) and on the following frames only the parameters of the nodes that changed are updated (hipGraphExecKernelNodeSetParams).
// Results are bit-identical to eager launches; per-pass profiling (nrdHipSetProfiling) falls back to eager launches while it is on.
// nrdHipGetGraphStats: graph launches, graphs built (topology misses) and node-parameter updates so far (any pointer may be NULL).
uint32_t nrdHipSetGraphMode(NrdHipExecutor* executor, uint32_t enable);
uint32_t nrdHipGetGraphStats(const NrdHipExecutor* executor, uint64_t* graphLaunches, uint64_t* graphBuilds, uint64_t* nodeUpdates);

// Diagnostics of the passes that run as a fast kernel plus a fallback kernel (REBLUR TemporalAccumulation: the surface-motion footprints of a 32x8-pixel tile come
// from one LDS-staged window of the previous frame; a tile whose window would be too large is left to the plain kernel -- DESIGN.md section 3). Reads the tile
// flags of the LAST frame back (synchronises the stream): tiles the fallback kernel processed and tiles in total. Results never depend on the split.
uint32_t nrdHipGetTileFallbackStats(NrdHipExecutor* executor, uint32_t* fallbackTiles, uint32_t* totalTiles);

// DEPRECATED (kept so that round-2 callers still link; do not use in new code). Always 0: there is one library and one arithmetic (DESIGN.md "Numerics": IEEE + - * and
// source-determined fused multiply-adds, division / sqrt / exp2 / log2 through v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 / v_exp_f32 / v_log_f32 -- bit-identical to the CPU oracle).
__attribute__((deprecated("one library, one arithmetic: the answer is always 0")))
uint32_t nrdHipGetNumericsMode(void);

// Bytes held by the pool arena (permanent, transient).
uint32_t nrdHipGetPoolMemoryUsage(const NrdHipExecutor* executor, uint64_t* permanentBytes, uint64_t* transientBytes);

// Diagnostics: evaluates one primitive of the device numerics contract (DESIGN.md "Numerics") elementwise on device
// arrays, so a harness can pin the GPU's codecs and transcendentals bit-for-bit against another implementation.
//   op: 0 exp2, 1 log2, 2 atan, 3 pow(x, y = in2), 4 fp32->fp16->fp32 round trip, 5 Div(x, in2) = x * v_rcp_f32(in2), 6 sqrt, 7 1/sqrt,
//       8..12 small-integer / {1023, 255, 63, 15, 3} (the codecs' 3-op exact division), 13 exp(-0.66 x^2), 14 small-integer / 65535, 15 int16 / 32767;
//       16 v_rcp_f32, 17 v_rsq_f32, 18 v_sqrt_f32 (the raw instructions), 19 v_cvt_pk_f16_f32(x, in2) (the packed word as float bits), 20 Rcp;
//       21 Exp2NonPos(x) = 2 * v_exp_f32(x - 1) for x <= 0, 22 SatExp2(x) = saturate(2^x), 23 ExpNegAbs(x) = e^-|x|, 24 Pow01(x, in2) = saturate(x)^in2 for in2 >= 0 (round 5)
// in2 may be NULL for unary ops. Launches on hipStream (a hipStream_t as void*, may be NULL).
uint32_t nrdHipEvalNumerics(uint32_t op, const float* in1, const float* in2, float* out, uint32_t count, void* hipStream);

// Diagnostics: the streaming bandwidth this GPU delivers to a plain 16-bytes-per-lane copy kernel (read + write, GB/s) -- the "measured copy
// bandwidth on the same device" the roofline fractions of bench.py are also quoted against. Allocates 2 * bytes of scratch
// device memory, runs `repetitions` timed copies after 3 warm-up copies (HIP events on hipStream) and frees the scratch again.
uint32_t nrdHipMeasureCopyBandwidth(uint64_t bytes, uint32_t repetitions, void* hipStream, double* gigabytesPerSecond);

// Last error text of this executor (never NULL).
const char* nrdHipGetLastError(const NrdHipExecutor* executor);

#ifdef __cplusplus
}
#endif
