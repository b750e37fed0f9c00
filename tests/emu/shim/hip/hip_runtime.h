// TEST INFRASTRUCTURE -- NOT PRODUCT CODE. A CPU stand-in for <hip/hip_runtime.h>, just large enough to compile raytracingdenoiser_amd/csrc/hip/*.hip
// as plain C++ for x86-64 and run the kernels there, one fiber per GPU thread (tests/emu/emu_runtime.cpp). Purpose: hold the DEVICE SOURCES
// against the CPU oracle bit for bit without a GPU while they are being changed (operation order, FMA contraction, division, LDS staging); the
// GPU parity tests (-m gpu) remain the gate. Nothing under raytracingdenoiser_amd/ includes or loads this; the library it builds
// (tests/emu/libNRD_emu.so) is loaded by tests/emu/ only.
//
// What is modelled: grids / blocks / 64-lane waves, __shared__ (one block at a time per OS thread), __syncthreads, __all / __shfl_xor among the
// lanes that are still running, the kernel-argument segment, and the five transcendental instructions through the same measured tables the
// oracle uses (oracle/hw_math.h). What is not: timing, memory hierarchy, graphs (the graph API reports "not supported").
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <initializer_list>
#include <tuple>
#include <type_traits>
#include <utility>

#include "hw_math.h" // oracle/hw_math.h: v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 / v_exp_f32 / v_log_f32 bit for bit

#define NRD_EMU 1

// ------------------------------------------------------------------------------------------------ qualifiers
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __constant__ static

// ------------------------------------------------------------------------------------------------ vector types
struct alignas(8) float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(8) uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct alignas(16) int4 { int x, y, z, w; };
struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline int2 make_int2(int x, int y) { return int2{x, y}; }
inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

// ------------------------------------------------------------------------------------------------ execution model
namespace emu {
struct ThreadCtx {
    dim3 tid, bid, bdim, gdim;
};
extern thread_local ThreadCtx* t_cur; // the GPU thread this OS thread is executing right now
extern thread_local const void* t_kernarg;

void SyncThreads();
int All(int predicate);
int Any(int predicate);
uint32_t ShflXorBits(uint32_t v, int laneMask, int width);
void Launch(dim3 grid, dim3 block, const void* kernarg, const std::function<void()>& thread);

template <typename T>
inline void PackKernarg(uint8_t* buf, size_t& off, const T& v) {
    off = (off + alignof(T) - 1) & ~(alignof(T) - 1);
    memcpy(buf + off, &v, sizeof(T));
    off += sizeof(T);
}
} // namespace emu

// threadIdx.x & co read the context of the fiber that is running (clang's property members, -fdeclspec, as the real HIP headers do)
#define NRD_EMU_BUILTIN_DIM(NAME, FIELD)                                    \
    struct NAME##_t {                                                       \
        __declspec(property(get = GetX)) unsigned x;                        \
        __declspec(property(get = GetY)) unsigned y;                        \
        __declspec(property(get = GetZ)) unsigned z;                        \
        unsigned GetX() const { return emu::t_cur->FIELD.x; }               \
        unsigned GetY() const { return emu::t_cur->FIELD.y; }               \
        unsigned GetZ() const { return emu::t_cur->FIELD.z; }               \
    };                                                                      \
    static const NAME##_t NAME = {};
NRD_EMU_BUILTIN_DIM(threadIdx, tid)
NRD_EMU_BUILTIN_DIM(blockIdx, bid)
NRD_EMU_BUILTIN_DIM(blockDim, bdim)
NRD_EMU_BUILTIN_DIM(gridDim, gdim)

inline void __syncthreads() { emu::SyncThreads(); }
inline int __all(int p) { return emu::All(p); }
inline int __any(int p) { return emu::Any(p); }
inline float __shfl_xor(float v, int laneMask, int width = 64) {
    uint32_t u;
    memcpy(&u, &v, 4);
    u = emu::ShflXorBits(u, laneMask, width);
    memcpy(&v, &u, 4);
    return v;
}
inline int __shfl_xor(int v, int laneMask, int width = 64) { return (int)emu::ShflXorBits((uint32_t)v, laneMask, width); }
inline uint32_t __shfl_xor(uint32_t v, int laneMask, int width = 64) { return emu::ShflXorBits(v, laneMask, width); }

// ------------------------------------------------------------------------------------------------ device intrinsics
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline uint32_t min(uint32_t a, uint32_t b) { return a < b ? a : b; }
inline uint32_t max(uint32_t a, uint32_t b) { return a > b ? a : b; }
inline float min(float a, float b) { return fminf(a, b); }
inline float max(float a, float b) { return fmaxf(a, b); }
inline uint32_t __umul24(uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
inline uint32_t atomicMax(uint32_t* p, uint32_t v) { // blocks of a grid run on several OS threads
    uint32_t old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
    return old;
}
inline uint32_t __float_as_uint(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
inline float __uint_as_float(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline float emu_fmed3f(float a, float b, float c) { return std::max(std::min(a, b), std::min(std::max(a, b), c)); }
#define __builtin_amdgcn_fmed3f(a, b, c) emu_fmed3f(a, b, c)
#define __builtin_amdgcn_rcpf(x) hwmath::HwRcp(x)
#define __builtin_amdgcn_sqrtf(x) hwmath::HwSqrt(x)
#define __builtin_amdgcn_rsqf(x) hwmath::HwRsq(x)
#define __builtin_amdgcn_exp2f(x) hwmath::HwExp2Raw(x)
#define __builtin_amdgcn_logf(x) hwmath::HwLog2Raw(x)
#define __builtin_amdgcn_kernarg_segment_ptr() ((void*)emu::t_kernarg)
// the one opaque instruction of the device sources (planes.h FloatToHalfBits): fp32 -> fp16, round to nearest even, denormals kept
#define NRD_OPAQUE_CVT_F16(h, f) (h) = hwmath::F32ToF16Bits(f)
#define NRD_OPAQUE_CVT_PK_F16(r, a, b) (r) = (uint32_t)hwmath::F32ToF16Bits(a) | ((uint32_t)hwmath::F32ToF16Bits(b) << 16)
// nrdmath.h Rcp: the device hides the argument from the constant folder; nothing to hide from here
#define NRD_OPAQUE_VALUE(x) ((void)0)
#define NRD_LDS_WHOLE_TEXEL(v) ((void)0)
// LDS-DMA of the marching a-trous kernel (kernels_relax_atrous.hip): lane l of the wave copies its 16 source bytes to ldsWaveBase + 16 l, at once (no asynchrony to model:
// the device code waits for the DMA and meets a barrier in front of every read)
#define NRD_LDS_ADDRESS(p) ((uintptr_t)(p))
#define NRD_WAVE_UNIFORM(x) (x)
#define NRD_LDS_DMA16(gsrc, ldsWaveBase) memcpy((void*)((uintptr_t)(ldsWaveBase) + (uintptr_t)(threadIdx.x & 63u) * 16u), (const void*)(gsrc), 16)
#define NRD_LDS_DMA_WAIT() ((void)0)
#define NRD_LDS_POINTER(T, address) ((T*)(uintptr_t)(address))
typedef uintptr_t LdsAddress;
// ClampI = v_med3_i32: the MEDIAN of three like the instruction, not the clamp it stands for -- with a > b (an empty plane: clamp to [0, -1]) the two differ, and
// the device computes the median. Call sites with a > b are counted (emu_med3_violations(), asserted to be 0 by tests/test_emulation.py: ADVICE r03).
namespace emu {
extern long g_Med3Violations;
inline int Med3(int x, int a, int b) {
    if (a > b)
        __atomic_add_fetch(&g_Med3Violations, 1, __ATOMIC_RELAXED);
    const int lo = x < a ? x : a, hi = x < a ? a : x; // sort (x, a)
    return b < lo ? lo : (b > hi ? hi : b);            // the middle of (lo, hi, b)
}
} // namespace emu
#define NRD_MED3_I32(r, x, a, b) ((r) = emu::Med3((x), (a), (b)))

// ------------------------------------------------------------------------------------------------ runtime API (host side of executor.hip)
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorNotSupported = 801 };
typedef struct emuStream* hipStream_t;
typedef struct emuEvent* hipEvent_t;
typedef struct emuGraph* hipGraph_t;
typedef struct emuGraphExec* hipGraphExec_t;
typedef struct emuGraphNode* hipGraphNode_t;
struct hipKernelNodeParams {
    dim3 blockDim;
    void** extra;
    void* func;
    dim3 gridDim;
    void** kernelParams;
    unsigned sharedMemBytes;
};
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };

hipError_t hipGetDeviceCount(int* n);
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipMemsetAsync(void* p, int v, size_t bytes, hipStream_t s);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGraphCreate(hipGraph_t* g, unsigned flags);
hipError_t hipGraphDestroy(hipGraph_t g);
hipError_t hipGraphAddKernelNode(hipGraphNode_t* node, hipGraph_t g, const hipGraphNode_t* deps, size_t numDeps, const hipKernelNodeParams* p);
hipError_t hipGraphInstantiate(hipGraphExec_t* exec, hipGraph_t g, hipGraphNode_t* errNode, char* log, size_t logSize);
hipError_t hipGraphExecDestroy(hipGraphExec_t exec);
hipError_t hipGraphExecKernelNodeSetParams(hipGraphExec_t exec, hipGraphNode_t node, const hipKernelNodeParams* p);
hipError_t hipGraphLaunch(hipGraphExec_t exec, hipStream_t s);

template <typename... KArgs, typename... Args>
inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t /*dynamicLds*/, hipStream_t /*stream*/, Args&&... args) {
    std::tuple<typename std::decay<KArgs>::type...> packed{KArgs(std::forward<Args>(args))...};
    alignas(16) uint8_t kernarg[4096];
    size_t off = 0;
    std::apply([&](const auto&... a) { (void)std::initializer_list<int>{(emu::PackKernarg(kernarg, off, a), 0)...}; }, packed);
    emu::Launch(grid, block, kernarg, [&]() { std::apply(kernel, packed); });
}
